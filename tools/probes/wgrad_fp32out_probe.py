"""Probe: dW = g^T x with an fp32 OUTPUT straight from one library GEMM (torch.mm(..., out_dtype=torch.float32), TunableOp
choosing among all solutions incl. the library's own split-K / stream-K ones) against what ops._wgrad runs now
(split-K batched bf16 GEMM + mmae_splitk_sum)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
tun.set_filename("/tmp/probe_tun2.csv")
from incomplete_multimodal_fusion_amd import ops
dev = "cuda"
shapes = [(4096, 768, 163840), (768, 2048, 163840), (1536, 768, 163840), (768, 512, 163840), (1024, 768, 164096), (4096, 768, 65536), (768, 2048, 65536)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for M, N, K in shapes:
    g = torch.randn(K, M, device=dev, dtype=torch.bfloat16); x = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    fl = 2.0 * M * N * K
    t_now = timeit(lambda: ops._wgrad(g, x, out))
    line = "dW %4d x %4d over %6d rows: split-K bmm + sum %7.1f us %5.0f TF/s |" % (M, N, K, t_now * 1e6, fl / t_now / 1e12)
    try:
        t1 = timeit(lambda: torch.mm(g.t(), x, out_dtype=torch.float32))
        line += " mm fp32-out %7.1f us %5.0f TF/s" % (t1 * 1e6, fl / t1 / 1e12)
    except Exception as e:
        line += " mm fp32-out unavailable: %s" % str(e)[:80]
    print(line, flush=True)
