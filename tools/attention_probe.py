"""Runs the window-attention kernel alone at the four stage shapes of dpt_swin2_tiny_256 (B = 8) for rocprofv3 --pmc / --kernel-trace passes.
usage: rocprofv3 --pmc <counters> -d gpurun_out/attn -o a --output-format csv -- python3 tools/attention_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_window_attention
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (res, ws, shift, heads) in ((64, 16, 0, 3), (64, 16, 8, 3), (32, 16, 8, 6), (16, 16, 0, 12), (8, 8, 0, 24)):
    B, C = 8, heads * 32
    qkv = torch.randn(B * res * res, 3 * C, generator=g).to(torch.bfloat16).to(dev)
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g))).to(dev)
    scale = (10.0 + 5 * torch.rand(heads, generator=g)).to(dev)
    out = torch.empty(B * res * res, C, dtype=torch.bfloat16, device=dev)
    for _ in range(10):
        op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads)
    torch.cuda.synchronize()
print("done")
