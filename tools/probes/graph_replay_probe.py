#!/usr/bin/env python
"""How fast does a hipGraph of N small kernels replay on this stack (ROCm 7 / torch 2.10)?  N dependent tiny launches (ctypes launches of
mmae_scale_rows + torch elementwise ops), eager enqueue time vs torch.cuda.CUDAGraph replay: host time to enqueue and GPU time per launch."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
x = torch.randn(4096, 256, device="cuda")
s = torch.full((4096,), 1.0001, device="cuda")


def body():
    y = x
    for i in range(N // 2):
        y = ops.scale_rows(y, s)          # own kernel through the C ABI
        y = y + 1e-6                      # an ATen kernel
    return y


for _ in range(2):
    body()
torch.cuda.synchronize()
t0 = time.perf_counter(); out = body(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("eager : host %.2f ms, until done %.2f ms (%d launches: %.1f us each)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, N, (t2 - t0) * 1e6 / N))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g):
    out_g = body()
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("graph : host %.2f ms, until done %.2f ms (%.1f us each)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / N))
print("same result:", bool(torch.equal(out, out_g)))
