"""Every GEMM / conv shape of the B=8 dpt_swin2_tiny_256 forward x every viable kernel configuration, a few launches each
(WARM caches: a configuration that wins here by re-reading weights more often can lose in the network, where they arrive cold --
confirm a heuristic change with bench.py before keeping it).
Run under `rocprofv3 --kernel-trace --output-format csv`; tools/igemm_tune_parse.py turns the trace into a table of DEVICE
durations (event timing from Python is host-bound below ~10 us per launch and cannot rank these kernels)."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 8
lin = [("s0_qkv", 32768, 288, 96), ("s0_fc1", 32768, 384, 96), ("s0_proj", 32768, 96, 96), ("s0_fc2", 32768, 96, 384),
       ("s1_merge", 8192, 192, 384), ("s1_qkv", 8192, 576, 192), ("s1_proj", 8192, 192, 192), ("s1_fc1", 8192, 768, 192), ("s1_fc2", 8192, 192, 768),
       ("s2_merge", 2048, 384, 768), ("s2_qkv", 2048, 1152, 384), ("s2_proj", 2048, 384, 384), ("s2_fc1", 2048, 1536, 384), ("s2_fc2", 2048, 384, 1536),
       ("s3_merge", 512, 768, 1536), ("s3_qkv", 512, 2304, 768), ("s3_proj", 512, 768, 768), ("s3_fc1", 512, 3072, 768), ("s3_fc2", 512, 768, 3072),
       ("oc8", 512, 256, 256), ("oc16", 2048, 256, 256), ("oc32", 8192, 256, 256), ("oc64", 32768, 256, 256)]
conv = [("rn4", 8, 768, 256), ("rn3", 16, 384, 256), ("rn2", 32, 192, 256), ("rn1", 64, 96, 256),
        ("rcu8", 8, 256, 256), ("rcu16", 16, 256, 256), ("rcu32", 32, 256, 256), ("rcu64", 64, 256, 256),
        ("head_d", 128, 256, 128), ("head_s", 128, 256, 256)]
if os.environ.get("TUNE_SET") == "base384":   # dpt_swin2_base_384, B = 8 (BASELINE configs[3] per-GPU share)
    lin = [("b0_qkv", 73728, 384, 128), ("b0_fc1", 73728, 512, 128), ("b0_proj", 73728, 128, 128), ("b0_fc2", 73728, 128, 512),
           ("b1_merge", 18432, 256, 512), ("b1_qkv", 18432, 768, 256), ("b1_proj", 18432, 256, 256), ("b1_fc1", 18432, 1024, 256), ("b1_fc2", 18432, 256, 1024),
           ("b2_merge", 4608, 512, 1024), ("b2_qkv", 4608, 1536, 512), ("b2_proj", 4608, 512, 512), ("b2_fc1", 4608, 2048, 512), ("b2_fc2", 4608, 512, 2048),
           ("b3_merge", 1152, 1024, 2048), ("b3_qkv", 1152, 3072, 1024), ("b3_proj", 1152, 1024, 1024), ("b3_fc1", 1152, 4096, 1024), ("b3_fc2", 1152, 1024, 4096),
           ("boc12", 1152, 256, 256), ("boc24", 4608, 256, 256), ("boc48", 18432, 256, 256), ("boc96", 73728, 256, 256)]
    conv = [("brn4", 12, 1024, 256), ("brn3", 24, 512, 256), ("brn2", 48, 256, 256), ("brn1", 96, 128, 256),
            ("brcu12", 12, 256, 256), ("brcu24", 24, 256, 256), ("brcu48", 48, 256, 256), ("brcu96", 96, 256, 256),
            ("bhead_d", 192, 256, 128), ("bhead_s", 192, 256, 256)]
K64 = [2, 1, 13, 10, 14, 11, 8, 6, 20]
ONLY = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None   # optional: comma-separated shape names
ONLY_CFG = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # optional: comma-separated configuration ids
K32 = [4, 9, 3, 19, 15, 16]
order = []
sk_part = torch.empty(2 << 20, dtype=torch.float32, device=dev)
sk_count = torch.zeros(4096, dtype=torch.int32, device=dev)
def run_splitk(name, M, N, Cin, call_sk):   # label 100 + S in the table
    if ONLY_CFG:
        return
    if Cin % 64 or (M + 31) // 32 * ((N + 63) // 64) >= 512:
        return
    for S in (2, 3, 4, 6, 8):
        if S * M * N > sk_part.numel() or S > Cin // 64 * (1 if name.startswith(("s", "oc")) else 9):
            continue
        try:
            for _ in range(4):
                call_sk(S)
        except RuntimeError:   # scratch / counter capacity exceeded for this shape: all four launches were refused
            continue
        torch.cuda.synchronize()
        order.extend([[name, 100 + S, M, N, Cin]] * 4)
def run_all(name, M, N, Cin, call):
    cfgs = ONLY_CFG if ONLY_CFG else [-1] + (K64 + K32 if Cin % 64 == 0 else K32)
    for t in cfgs:
        ok = True
        for _ in range(4):
            try:
                call(t)
            except RuntimeError:
                ok = False
                break
        torch.cuda.synchronize()
        if ok:
            order.extend([[name, t, M, N, Cin]] * 4)
for name, M, N, K in lin:
    if ONLY and name not in ONLY:
        continue
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run_all(name, M, N, K, lambda t: op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_bf16=out, tune=t))
    run_splitk(name, M, N, K, lambda S: op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_bf16=out, splitk=S, sk_part=sk_part, sk_count=sk_count))
for name, H, Cin, Cout in conv:
    if ONLY and name not in ONLY:
        continue
    M = B * H * H
    x = torch.zeros(B, H + 2, H + 2, Cin, dtype=torch.bfloat16, device=dev)
    x[:, 1:-1, 1:-1] = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(torch.bfloat16).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.bfloat16, device=dev)
    outf = torch.empty(M, Cout, device=dev)
    run_all(name, M, Cout, Cin * 9, lambda t: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_f32=outf, out_bf16=outh, out_halo=1, tune=t))
    run_splitk(name, M, Cout, Cin, lambda S: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_f32=outf, out_bf16=outh, out_halo=1,
                                                     splitk=S, sk_part=sk_part, sk_count=sk_count))
json.dump(order, open("gpurun_out/tune_order.json", "w"))
