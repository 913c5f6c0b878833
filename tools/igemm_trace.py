"""Run each (shape, config) a few times; meant to be run under `rocprofv3 --kernel-trace` so that true device
durations (not host-bound launch rates) can be compared.  Prints the launch order for the parser."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [("lin_s1_qkv", 8192, 576, 192), ("lin_s1_fc1", 8192, 768, 192), ("lin_s1_fc2", 8192, 192, 768), ("lin_s1_proj", 8192, 192, 192),
          ("lin_s2_qkv", 2048, 1152, 384), ("lin_s2_proj", 2048, 384, 384), ("lin_s2_fc1", 2048, 1536, 384), ("lin_s2_fc2", 2048, 384, 1536),
          ("lin_s3_qkv", 512, 2304, 768), ("lin_s3_proj", 512, 768, 768), ("lin_s3_fc1", 512, 3072, 768), ("lin_s3_fc2", 512, 768, 3072)]
cfgs = [2, 1, 13, 10, 14, 11]
order = []
for name, M, N, K in shapes:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for t in cfgs:
        for _ in range(6):
            op_igemm(x, w, M, N, K, ldx=K, out_bf16=out, tune=t)
        torch.cuda.synchronize()
        order.append([name, t, M, N, K])
json.dump(order, open("gpurun_out/igemm_trace_order.json", "w"))
