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
