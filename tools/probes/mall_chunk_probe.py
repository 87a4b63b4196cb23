#!/usr/bin/env python
"""Does chunking the FF chain by rows (so the GEGLU input is still in the 256 MB Infinity Cache when it is read) pay?
y -> GEMM1 (768->4096) -> GEGLU -> GEMM2 (2048->768), full batch vs row chunks, forward and backward-like chains."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from incomplete_multimodal_fusion_amd._lib import call, ptr, stream


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda:0"
    rows, D, F = 163840, 768, 2048
    y = torch.randn(rows, D, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(2 * F, D, device=dev) * 0.02).to(torch.bfloat16)
    w2 = (torch.randn(D, F, device=dev) * 0.02).to(torch.bfloat16)
    h = torch.empty(rows, 2 * F, device=dev, dtype=torch.bfloat16)
    g = torch.empty(rows, F, device=dev, dtype=torch.bfloat16)
    o = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    # something large in between iterations so that nothing survives in the cache from the previous repetition
    junk = torch.empty(1 << 28, device=dev, dtype=torch.float32)

    def fwd(nchunk):
        c = rows // nchunk
        junk.add_(1.0)
        for i in range(nchunk):
            s = slice(i * c, (i + 1) * c)
            torch.mm(y[s], w1.t(), out=h[s])
            call("mmae_geglu_fwd", 1, c, F, ptr(h[s]), ptr(g[s]), stream())
            torch.mm(g[s], w2.t(), out=o[s])

    def only_gemms(nchunk):
        c = rows // nchunk
        junk.add_(1.0)
        for i in range(nchunk):
            s = slice(i * c, (i + 1) * c)
            torch.mm(y[s], w1.t(), out=h[s])
            torch.mm(g[s], w2.t(), out=o[s])

    tj = timeit(lambda: junk.add_(1.0))
    for nc in (1, 2, 4, 8, 16):
        t = timeit(lambda: fwd(nc)) - tj
        tg = timeit(lambda: only_gemms(nc)) - tj
        print("chunks %2d: chain %.0f us   gemms only %.0f us   -> geglu %.0f us" % (nc, t, tg, t - tg), flush=True)


if __name__ == "__main__":
    main()
