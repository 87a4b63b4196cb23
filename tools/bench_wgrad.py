#!/usr/bin/env python
"""Own transposing split-K weight-gradient GEMM (csrc/gemm.hip: mmae_gemm_tn) against the library path (split-K batched GEMM through torch +
mmae_splitk_sum) on the weight-gradient shapes of the ViT-B bench step, interleaved in one process."""
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
work = os.path.join(tempfile.gettempdir(), "bench_wgrad_tun.csv")
shutil.copyfile(table, work)
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30); tun.set_filename(work)
only = sys.argv[1] if len(sys.argv) > 1 else ""

B, N_, P = 256, 384, 256
R, RF, RK = B * (N_ + P), B * P, B * (N_ + P) + P
SHAPES = [("dW qkv", R, 1536, 768), ("dW to_out", R, 768, 512), ("dW FF1", R, 4096, 768), ("dW FF2", R, 768, 2048),
          ("dW kv (fusion)", RK, 1024, 768), ("dW q (fusion)", RF, 512, 768), ("dW to_out (fusion)", RF, 768, 512),
          ("dW FF1 (fusion)", RF, 4096, 768), ("dW FF2 (fusion)", RF, 768, 2048)]


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


for name, rows, N, Kin in SHAPES:
    if only and only not in name:
        continue
    pad = int(os.environ.get("WGRAD_PAD", "0"))          # elements added to both leading dimensions (power-of-two row strides or not)
    g = (torch.rand(rows, N + pad, device="cuda") * 2 - 1).to(torch.bfloat16)[:, :N]
    x = (torch.rand(rows, Kin + pad, device="cuda") * 2 - 1).to(torch.bfloat16)[:, :Kin]
    out = torch.empty(N, Kin, device="cuda")

    def lib():
        ops.OWN_GEMM = 1
        ops._wgrad(g, x, out)

    def own():
        ops.OWN_GEMM = 3
        ops._wgrad(g, x, out)
    for _ in range(3):
        lib(); own()
    tl, to = [], []
    for _ in range(2):
        tl.append(timed(lib)); to.append(timed(own))
    fl = 2.0 * rows * N * Kin
    print("%-22s rows %6d N %4d Kin %4d   library %7.1f us (%4.0f TF)   own %7.1f us (%4.0f TF)   own/lib %.3f" %
          (name, rows, N, Kin, min(tl), fl / min(tl) / 1e6, min(to), fl / min(to) / 1e6, min(to) / min(tl)), flush=True)
