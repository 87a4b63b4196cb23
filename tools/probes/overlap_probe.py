#!/usr/bin/env python
"""Does an HBM-bound row kernel overlap with an MFMA-bound GEMM when they sit on two HIP streams?  (Decides whether running two
half-batches on two streams -- one in its GEMMs while the other is in its LayerNorm / GEGLU kernels -- could shorten the step.)
Times: the GEMM alone, the streaming kernel alone, both back to back on one stream, both on two streams."""
import time

import torch

dev = "cuda"
R, K, N = 163840, 768, 4096
a = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
x = torch.randn(R, N, device=dev, dtype=torch.bfloat16)          # 1.34 GB
y = torch.empty_like(x)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
IT = 20


def gemm():
    torch.mm(a, w.t(), out=out)


def stream_op():
    torch.add(x, 1.0, out=y)                                      # 2.7 GB of traffic


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def serial():
    for _ in range(IT):
        gemm(); stream_op()


def only_gemm():
    for _ in range(IT):
        gemm()


def only_stream():
    for _ in range(IT):
        stream_op()


def concurrent():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        for _ in range(IT):
            gemm()
    with torch.cuda.stream(s2):
        for _ in range(IT):
            stream_op()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


def interleaved():
    """issue order alternates between the streams, as a two-half-batch step would"""
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(IT):
        with torch.cuda.stream(s1):
            gemm()
        with torch.cuda.stream(s2):
            stream_op()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


g, s, ser, con, inter = timed(only_gemm), timed(only_stream), timed(serial), timed(concurrent), timed(interleaved)
print("per iteration: GEMM alone %.3f ms (%.0f TFLOP/s), streaming kernel alone %.3f ms (%.2f TB/s)" %
      (g / IT, 2.0 * R * K * N / (g / IT * 1e-3) / 1e12, s / IT, 2 * x.numel() * 2 / (s / IT * 1e-3) / 1e12))
print("one stream, back to back %.3f ms; two streams %.3f ms; two streams, interleaved issue %.3f ms; sum of parts %.3f ms"
      % (ser / IT, con / IT, inter / IT, (g + s) / IT))
