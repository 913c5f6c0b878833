"""Generates tests/golden/gt_occupancy.npz by running THE REFERENCE'S OWN `OccupancyProcessor.process_frame`
(/root/reference/SOccDPT/datasets/bdd_helper.py:238-530) in this container.  The module imports cv2 only for two channel
flips (`cv2.cvtColor(..., COLOR_BGR2RGB)`); an empty stand-in module with that one function (a channel reversal) is put in
sys.modules -- nothing of the numeric path goes through it.  numpy here is 2.x: with its promotion rules
`baseline * focal_length * np.reciprocal(disparity)` is evaluated in float64 and rounded to float32 once (numpy 1.22, which the
reference's requirements.txt pins, rounds the product in float32); the oracle and the HIP kernel follow the behaviour pinned here.

    python oracle/make_golden_gt_occ.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.golden_inputs import gt_occ_inputs  # noqa: E402

cv2 = types.ModuleType("cv2")
cv2.COLOR_BGR2RGB = 4
cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
sys.modules["cv2"] = cv2
spec = importlib.util.spec_from_file_location("ref_bdd_helper", "/root/reference/SOccDPT/datasets/bdd_helper.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def main():
    disparity, seg_class, K, H, W, C = gt_occ_inputs()
    colors = {c: (40 * c + 10, 40 * c + 10, 40 * c + 10) for c in range(C)}      # grey levels: invariant under the channel flips
    class_2_color = dict(colors)
    color_2_class = {v: k for k, v in colors.items()}
    seg_frame = np.zeros((H, W, 3), dtype=np.uint8)
    for c, col in colors.items():
        seg_frame[seg_class == c] = col
    proc = ref.OccupancyProcessor(intrinsic_matrix=K, height=H, width=W, grid_size=(256, 256, 32), scale=(2.0, 2.0, 0.666), shift=(0.0, 0.0, 0.0),
                                  pc_scale=(500.0, 2500.0, 200.0), pc_shift=(100.0, 40.0, 0.0), point_count_threshold=10,
                                  class_2_color=class_2_color, color_2_class=color_2_class, num_classes=C)
    frame = dict(rgb_frame=np.zeros((H, W, 3), dtype=np.uint8), disparity_frame=disparity.copy(), seg_frame=seg_frame)
    out = proc.process_frame(frame)
    grid = np.asarray(out["occupancy_grid"])
    pts = np.asarray(out["points"])
    assert grid.dtype == bool and pts.dtype == np.float64
    rows = np.linspace(0, H * W - 1, 4096).astype(np.int64)
    path = os.path.join(REPO, "tests", "golden", "gt_occupancy.npz")
    np.savez_compressed(path, grid_bits=np.packbits(grid.reshape(-1)), grid_shape=np.array(grid.shape), depth=np.asarray(out["depth"]).astype(np.float32),
                        point_rows=rows, points_sample=pts[rows], points_sum=np.array([pts[:, k].sum() for k in range(3)]),
                        n_occupied=np.array([int(grid.sum())]))
    print("wrote", path, "occupied cells:", int(grid.sum()), "per class", [int(grid[..., c].sum()) for c in range(C)])


if __name__ == "__main__":
    main()
