#!/usr/bin/env python
"""Own persistent GEMM (csrc/gemm.hip) against the library GEMM (hipBLASLt through torch, TunableOp with the committed table) on every
forward / input-gradient projection shape of the ViT-B bench step, interleaved in one process (cdna_hip_programming.md rule 24).
Prints per shape: library us, own us, ratio; for the FeedForward[1] shapes the library figure includes the separate GEGLU kernel."""
import os
import shutil
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd import ops  # noqa: E402
import torch.cuda.tunable as tun  # noqa: E402

table = os.path.join(ROOT, "incomplete_multimodal_fusion_amd", "tuned", "tunableop_gfx950.csv")
work = os.path.join(tempfile.gettempdir(), "bench_gemm_tun.csv")
shutil.copyfile(table, work)
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30); tun.set_filename(work)

B, N_, P = 256, 384, 256
R, RF, RK = B * (N_ + P), B * P, B * (N_ + P) + P
SHAPES = [("qkv", R, 1536, 768), ("to_out", R, 768, 512), ("FF2", R, 768, 2048), ("kv (fusion)", RK, 1024, 768), ("q (fusion)", RF, 512, 768),
          ("to_out (fusion)", RF, 768, 512), ("FF2 (fusion)", RF, 768, 2048),
          ("qkv dgrad", R, 768, 1536), ("to_out dgrad", R, 512, 768), ("FF2 dgrad", R, 2048, 768), ("FF1 dgrad", R, 768, 4096),
          ("kv dgrad (fusion)", RK, 768, 1024), ("to_out dgrad (fusion)", RF, 512, 768), ("FF2 dgrad (fusion)", RF, 2048, 768),
          ("FF1 dgrad (fusion)", RF, 768, 4096), ("FF1 plain", R, 4096, 768)]
GEGLU = [("FF1 + GEGLU", R, 2048, 768), ("FF1 + GEGLU (fusion)", RF, 2048, 768)]


def timed(fn, it=10, rounds=3):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best * 1e3


only = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for name, M, N, K in SHAPES:
    if only and only not in name:
        continue
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    lib = lambda: torch.mm(a, w.t(), out=y)
    own = lambda: ops.gemm_nt(a, w, out=y)
    for _ in range(3):
        lib(); own()
    tl, to = [], []
    for _ in range(2):
        tl.append(timed(lib)); to.append(timed(own))
    fl = 2.0 * M * N * K
    rows.append((name, M, N, K, min(tl), min(to), fl))
    print("%-24s M %6d N %4d K %4d   library %7.1f us (%4.0f TF)   own %7.1f us (%4.0f TF)   own/lib %.3f" %
          (name, M, N, K, min(tl), fl / min(tl) / 1e6, min(to), fl / min(to) / 1e6, min(to) / min(tl)), flush=True)
for name, M, F, K in GEGLU:
    if only and only not in name:
        continue
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(2 * F, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    h = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16); g = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)

    def lib():
        torch.mm(a, w.t(), out=h)
        ops.call("mmae_geglu_fwd", ops.dt(h), M, F, ops.ptr(h), ops.ptr(g), ops.stream())
    own = lambda: ops.gemm_geglu(a, w, h, g)
    for _ in range(3):
        lib(); own()
    tl, to = [], []
    for _ in range(2):
        tl.append(timed(lib)); to.append(timed(own))
    print("%-24s M %6d F %4d K %4d   library GEMM + GEGLU kernel %7.1f us   own fused %7.1f us   own/lib %.3f" %
          (name, M, F, K, min(tl), min(to), min(to) / min(tl)), flush=True)
