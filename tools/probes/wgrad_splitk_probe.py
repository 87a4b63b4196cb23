"""Probe: weight-gradient GEMM dW = g^T x for (K x M)^T (K x N) with huge K: plain mm vs split-K through bmm + sum."""
import torch, time, sys
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
tun.set_filename("/tmp/probe_tun.csv")
dev = "cuda"
shapes = [(512, 768, 163840), (768, 1024, 164864), (1536, 768, 163840), (2048, 768, 163840), (768, 4096, 163840),
          (512, 768, 65536), (768, 512, 65536), (768, 4096, 65536), (2048, 768, 65536)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K in shapes:
    g = torch.randn(K, M, device=dev, dtype=torch.bfloat16); x = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t = timeit(lambda: torch.mm(g.t(), x))
    line = "M=%5d N=%5d K=%6d  mm %7.1f us %6.0f TF/s |" % (M, N, K, t * 1e6, fl / t / 1e12)
    ref = torch.mm(g.t(), x).float()
    for S in (4, 8, 16, 32):
        if K % S: continue
        gs = g.view(S, K // S, M); xs = x.view(S, K // S, N)
        f = lambda: torch.bmm(gs.transpose(1, 2), xs).sum(0, dtype=torch.float32)
        t = timeit(f)
        err = float((f() - ref).abs().max() / ref.abs().max())
        line += " S%-2d %6.1f us %5.0f TF (%.0e) |" % (S, t * 1e6, fl / t / 1e12, err)
    print(line, flush=True)
