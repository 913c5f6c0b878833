"""GPU: the fused multi-tensor Adam step (csrc/adam.hip through soccdpt_adam_step) against torch.optim.Adam on the CPU with the
reference's hyper-parameters (scripts/train_SOccDPT.py:311-318).  Same f32 update formula; torch evaluates it as a chain of
separately rounded ATen ops, the kernel with fused multiply-adds: parameters agree to 2e-6 relative after 5 steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch(gpu_device, wd):
    from soccdpt_amd.utils.optim import Adam
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 256, 3, 3), (768,), (3, 256, 1, 1), (1,), (96, 3, 4, 4)] + [(17, 5)] * 60     # > 48 tensors: two fused launches
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    gpu_p = [torch.nn.Parameter(p.detach().clone().to(gpu_device)) for p in ref_p]
    ro = torch.optim.Adam(ref_p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=False, foreach=False)
    go = Adam(gpu_p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for step in range(5):
        for i, (a, b) in enumerate(zip(ref_p, gpu_p)):
            if i == 3 and step < 2:          # a parameter without a gradient is skipped and keeps its own step count
                a.grad, b.grad = None, None
                continue
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (i % 5 - 3))
            a.grad, b.grad = gr, gr.to(gpu_device)
        ro.step()
        go.step()
    torch.cuda.synchronize()
    for a, b in zip(ref_p, gpu_p):
        torch.testing.assert_close(b.detach().cpu(), a.detach(), rtol=2e-6, atol=1e-7)
    st = go.state[gpu_p[0]]
    torch.testing.assert_close(st["exp_avg"].cpu(), ro.state[ref_p[0]]["exp_avg"], rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(st["exp_avg_sq"].cpu(), ro.state[ref_p[0]]["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    assert go.state[gpu_p[3]]["step"] == 3
    go.zero_grad()
    assert all(p.grad is None for p in gpu_p)


def test_forward_after_adam_step_uses_updated_weights(gpu_device):
    """the fused Adam writes parameters through raw pointers.  The step must bump the tensors' version counters so that
    SOccDPT_V3 re-runs soccdpt_prepare (16-bit weight copies, BN fold, CPB tables): the forward after a step must equal the forward
    of a freshly built model holding the stepped weights, bit for bit.  Same for `p.data = ...` (new storage, no version bump) and a
    non-contiguous gradient (Adam must keep its contiguous copy alive across the launch)."""
    import os
    import tempfile
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.optim import Adam
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))

    def mk(sd):
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False)
        m.load_state_dict(sd, strict=False)
        return m.eval().to(gpu_device)
    sd = synth_state_dict(alias_pretrained=True)
    m = mk(sd)
    x = synth_input(2, seed0=77).to(gpu_device)
    inv0, seg0 = m.network(x)
    g = torch.Generator().manual_seed(5)
    params = [p for p in m.parameters()]
    opt = Adam(params, lr=1e-2)
    for i, p in enumerate(params):
        gr = torch.randn(p.shape, generator=g).to(gpu_device)
        if p.dim() == 2 and i % 2 == 0:
            gr = torch.randn(tuple(reversed(p.shape)), generator=g).to(gpu_device).t()   # non-contiguous view
            assert not gr.is_contiguous()
        p.grad = gr
    opt.step()
    inv1, seg1 = m.network(x)
    torch.cuda.synchronize()
    assert not torch.equal(inv1, inv0)
    fresh = mk({k: v.detach().cpu() for k, v in m.state_dict().items()})
    inv2, seg2 = fresh.network(x)
    torch.cuda.synchronize()
    assert torch.equal(inv1, inv2) and torch.equal(seg1, seg2)
    # p.data = ... : new storage, same version
    w = m.depth_net.scratch.output_conv[4].weight
    w.data = (w.data * 1.5).clone()
    inv3, _ = m.network(x)
    fresh = mk({k: v.detach().cpu() for k, v in m.state_dict().items()})
    inv4, _ = fresh.network(x)
    torch.cuda.synchronize()
    assert not torch.equal(inv3, inv1) and torch.equal(inv3, inv4)
