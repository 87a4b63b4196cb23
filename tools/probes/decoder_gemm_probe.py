#!/usr/bin/env python
"""Library time of the decoders' projection shapes (65 536 rows, widths 256 ... 1024, bias) against their HBM / MFMA bound:
how much an own kernel for these shapes could recover.  Tuned library GEMMs (TunableOp, committed table)."""
import os, shutil, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch.cuda.tunable as tun
work = os.path.join(tempfile.gettempdir(), "dec_tun.csv")
shutil.copyfile(os.path.join(ROOT, "incomplete_multimodal_fusion_amd", "tuned", "tunableop_gfx950.csv"), work)
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30); tun.set_filename(work)
M = 65536
SHAPES = [("qkv", 768, 256, True), ("proj", 256, 256, True), ("fc1", 1024, 256, True), ("fc2", 256, 1024, True), ("out_proj s2", 768, 256, True),
          ("qkv dgrad", 256, 768, False), ("proj dgrad", 256, 256, False), ("fc1 dgrad", 256, 1024, False), ("fc2 dgrad", 1024, 256, False)]


def timed(fn, it=20, rounds=5):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best * 1e3


tot_l = tot_b = 0.0
for name, N, K, bias in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16) if bias else None
    fn = (lambda: torch.nn.functional.linear(a, w, b))
    for _ in range(3):
        fn()
    t = timed(fn)
    bytes_ = 2.0 * (M * K + M * N + N * K)
    bound = max(bytes_ / 5.5e12, 2.0 * M * N * K / 1.3e15) * 1e6
    tot_l += t; tot_b += bound
    print("%-12s N %4d K %4d  library %6.1f us   bound %5.1f us (%.0f MB)   x%.2f" % (name, N, K, t, bound, bytes_ / 1e6, t / bound), flush=True)
print("sum %.0f us library, %.0f us bound per decoder block (one of six); x6 blocks + wgrads on top" % (tot_l, tot_b))
