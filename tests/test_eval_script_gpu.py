"""GPU: the evaluation entry point (counterpart of scripts/eval_SOccDPT.py / scripts/eval.sh) end to end on the synthetic
validation subset: flags parse like the reference's, the FPS loop and the metric lines run, and the fp16 (`-o`) and bf16
runs agree on the metrics within the modes' tolerances."""
import pytest

pytestmark = pytest.mark.gpu


def test_eval_script_runs(gpu_device, capsys):
    from soccdpt_amd.scripts.eval_SOccDPT import build_parser, main
    args = build_parser().parse_args(["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "-d", "cuda:0", "-b", "/nonexistent"])
    r = main(args)
    out = capsys.readouterr().out
    for line in ("Model: SOccDPT_V3_dpt_swin2_tiny_256", "FPS:", "IOU:", "ABS_REL:", "RMSE:", "A3:"):
        assert line in out
    assert r["fps"] > 100 and 0.0 < r["iou"] <= 1.0 and r["rmse"] > 0
    r16 = main(build_parser().parse_args(["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "-d", "cuda:0", "-b", "/nonexistent", "-o"]))
    assert abs(r16["iou"] - r["iou"]) < 0.05 and abs(r16["rmse"] - r["rmse"]) / r["rmse"] < 0.05
