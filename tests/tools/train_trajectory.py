import os, sys, tempfile, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from soccdpt_amd.lib import PREC_F32
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
from soccdpt_amd.utils.loss import training_loss
from soccdpt_amd.utils.optim import Adam
from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
dev = torch.device("cuda:0")
for mt, B, amp in (("dpt_swin2_tiny_256", 4, False), ("dpt_swin2_tiny_256", 4, True), ("dpt_swin2_base_384", 2, False), ("dpt_hybrid_384", 2, False), ("dpt_hybrid_384", 2, True)):
    bb = MODEL_TYPE_TO_BACKBONE[mt]; S = backbone_image_size(bb)
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "c.yaml"))
    net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32, model_type=mt)
    net.drop_path_rate = 0.0   # deterministic comparisons: no stochastic depth
    net.load_state_dict(synth_state_dict(bb, alias_pretrained=True), strict=False)
    net = net.to(dev).train(); net.train_amp = amp
    ds = SyntheticDepthSegDataset(B, S)
    x, _, md, yd, ms, ys = get_batch(ds, B, B)
    x = x.to(dev, torch.float32); yd, ys = yd.to(dev, torch.float32), ys.to(dev, torch.float32); md, ms = md.to(dev, torch.bool), ms.to(dev, torch.bool)
    opt = Adam(net.parameters(), lr=3e-5)
    losses = []
    for step in range(40):
        inv, seg = net.train_forward(x, seed=step)
        out = training_loss(inv, seg, yd, md, ys, ms, 0.5, 0.5, compute_scale_and_shift=True)
        opt.zero_grad(set_to_none=True)
        net.backward(out["d_inv"], out["d_seg"])
        opt.step()
        losses.append(float(out["loss"]))
    finite = all(torch.isfinite(p).all() for p in net.parameters())
    print(mt, "amp" if amp else "f32", "loss", " ".join(f"{v:.3f}" for v in losses[::5]), "final", f"{losses[-1]:.3f}", "params finite", finite, flush=True)
    del net, opt
    torch.cuda.empty_cache()
