"""Do independent small kernels on different streams overlap on MI355X (eager and hipGraph)?"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
M, N, K = 2048, 384, 384
NCH = 4
xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NCH)]
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(NCH)]
L = 100
def chain(i, n):
    for _ in range(n):
        op_igemm(xs[i], w, M, N, K, ldx=K, out_bf16=outs[i])
streams = [torch.cuda.Stream() for _ in range(NCH)]
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / it * 1e3
def serial(nch):
    with torch.cuda.stream(streams[0]):
        for i in range(nch): chain(i, L)
def parallel(nch):
    for i in range(nch):
        with torch.cuda.stream(streams[i]):
            chain(i, L)
def parallel_interleaved(nch):
    for _ in range(L):
        for i in range(nch):
            with torch.cuda.stream(streams[i]):
                chain(i, 1)
for nch in (1, 2, 4):
    print(f"eager: {nch} chains x {L}: serial {timeit(lambda: serial(nch)):.3f} ms, parallel {timeit(lambda: parallel(nch)):.3f} ms, interleaved {timeit(lambda: parallel_interleaved(nch)):.3f} ms")
# graphs
def make_graph(nch, par):
    g = torch.cuda.CUDAGraph()
    s0 = streams[0]
    with torch.cuda.graph(g, stream=s0):
        if par:
            for i in range(1, nch):
                streams[i].wait_stream(s0)
            for i in range(nch):
                with torch.cuda.stream(streams[i] if i else s0):
                    chain(i, L)
            for i in range(1, nch):
                s0.wait_stream(streams[i])
        else:
            for i in range(nch): chain(i, L)
    return g
for nch in (1, 2, 4):
    gs, gp = make_graph(nch, False), make_graph(nch, True)
    print(f"graph: {nch} chains x {L}: serial {timeit(gs.replay):.3f} ms, parallel {timeit(gp.replay):.3f} ms")
