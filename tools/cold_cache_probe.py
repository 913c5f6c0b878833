"""How much of a GEMM's in-network duration is cache coldness?  Runs one shape (a) back to back on warm buffers,
(b) after a 1 GiB fill that flushes L2 + MALL, (c) flushed, then only the weights re-touched, (d) flushed, then only the
activations re-touched.  Run under `rocprofv3 --kernel-trace --output-format csv`; tools/cold_cache_parse.py reads the trace."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
thrash = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
order = []
shapes = [("lin_s0_qkv", 32768, 288, 96, 1, 0), ("lin_s0_fc1", 32768, 384, 96, 1, 0), ("lin_s2_qkv", 2048, 1152, 384, 1, 0),
          ("lin_s2_fc2", 2048, 384, 1536, 1, 0), ("conv64_256", 32768, 256, 256, 9, 64), ("conv16_256", 2048, 256, 256, 9, 16)]
for name, M, N, Cin, taps, H in shapes:
    K = Cin * taps
    if taps == 9:
        B = M // (H * H)
        x = torch.zeros(B, H + 2, H + 2, Cin, dtype=torch.bfloat16, device=dev)
        x[:, 1:-1, 1:-1] = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev)
    else:
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    outf = torch.empty(M, N, dtype=torch.float32, device=dev)
    def run(f32out=False):
        kw = dict(taps=taps, H=H, W=H) if taps == 9 else dict(ldx=K)
        op_igemm(x, w, M, N, Cin, bias=bias, out_bf16=out, out_f32=outf if f32out else None, **kw)
    for mode in ("warm", "cold", "cold+w", "cold+x", "warm+f32out", "cold+f32out"):
        for _ in range(5):
            if mode.startswith("cold"):
                thrash.fill_(1)
            if mode == "cold+w":
                w.sum()
            if mode == "cold+x":
                x.sum()
            run("f32out" in mode)
            order.append([name, mode])
        torch.cuda.synchronize()
json.dump(order, open("gpurun_out/cold_order.json", "w"))
