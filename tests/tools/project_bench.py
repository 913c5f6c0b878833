import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import soccdpt_ref as R
from soccdpt_amd.lib import Engine, make_config
from tests.golden_inputs import proj_inputs
dev = torch.device("cuda:0")
cam, cfg = R.Camera(), R.ProjConfig()
c = make_config("swin2t16_256", 3, 256, False, True, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                cfg.grid_size, cfg.occupancy_shape(), cfg.pc_scale, cfg.pc_shift, cfg.correction_angle)
eng = Engine(c, dev)
B = 8
inv, seg = proj_inputs(seed=21, B=B)
inv, seg = inv.to(dev), seg.to(dev)
inv_up = torch.empty((B, 1080, 1920), device=dev); seg_up = torch.empty((B, 3, 1080, 1920), device=dev)
pts = torch.empty((B, 1080, 1920, 3), device=dev); bits = torch.zeros((eng.occ_words(),), dtype=torch.int32, device=dev)
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
cases = {
 "all outputs + occ": (inv_up, seg_up, pts, bits),
 "all outputs, no occ": (inv_up, seg_up, pts, None),
 "points only": (None, None, pts, None),
 "inv_up only": (inv_up, None, None, None),
 "seg_up only": (None, seg_up, None, None),
 "occ only": (None, None, None, bits),
}
for name, (a, b_, c_, d) in cases.items():
    us = bench(lambda: eng.project(inv, seg, a, b_, c_, d, clear_bits=True))
    byt = B * 1080 * 1920 * 4 * ((1 if a is not None else 0) + (3 if b_ is not None else 0) + (3 if c_ is not None else 0))
    print(f"{name:22s} {us:8.1f} us   {byt/us/1e3:8.1f} GB/s written")
