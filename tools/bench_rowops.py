#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound row kernels at the bench shapes (one process, HIP events).  `--lib` points the binding
at another build of libmmae_hip.so for A/B runs inside one gpurun call."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, n=30, warm=5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--rows", type=int, default=163840)
    args = ap.parse_args()
    from incomplete_multimodal_fusion_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from incomplete_multimodal_fusion_amd._lib import call, ptr, stream
    dev = "cuda:0"
    rows, F = args.rows, 2048
    h = torch.randn(rows, 2 * F, device=dev).to(torch.bfloat16)
    g = torch.randn(rows, F, device=dev).to(torch.bfloat16)
    out = torch.empty(rows, F, device=dev, dtype=torch.bfloat16)
    dh = torch.empty_like(h)
    t_f = timeit(lambda: call("mmae_geglu_fwd", 1, rows, F, ptr(h), ptr(out), stream()))
    t_b = timeit(lambda: call("mmae_geglu_bwd", 1, rows, F, ptr(h), ptr(g), ptr(dh), stream()))
    bf, bb = rows * F * 2 * 3, rows * F * 2 * 5
    print("lib=%s  geglu_fwd %.1f us (%.2f TB/s)   geglu_bwd %.1f us (%.2f TB/s)" %
          (os.path.basename(_lib.LIB_PATH), t_f, bf / t_f / 1e6, t_b, bb / t_b / 1e6))
    x = torch.randn(65536, 1024, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x); gx = torch.randn_like(x); dx = torch.empty_like(x)
    t_f = timeit(lambda: call("mmae_gelu_fwd", 1, x.numel(), ptr(x), ptr(y), stream()))
    t_b = timeit(lambda: call("mmae_gelu_bwd", 1, x.numel(), ptr(x), ptr(gx), ptr(dx), stream()))
    print("            gelu_fwd %.1f us (%.2f TB/s)   gelu_bwd %.1f us (%.2f TB/s)" %
          (t_f, x.numel() * 4 / t_f / 1e6, t_b, x.numel() * 6 / t_b / 1e6))


if __name__ == "__main__":
    main()
