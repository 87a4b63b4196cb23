#!/usr/bin/env python
"""Where the FF1 + GEGLU epilogue of the own GEMM (csrc/gemm.hip, gemm8p_kernel<1>) spends its time: the same shapes timed with the
library given by MMAE_HIP_LIB (product, the previous epilogue, or a timing-only build: `-DGM_EPI_DIAG=1` no GELU, `=2` no product store,
`=3` h written with 8-byte stores), next to the plain N = 2 F GEMM of the same kernel family (EPI 0: the epilogue-free reference).
One process per library (the library is bound at import); run them alternately:

    for l in "" prev d1 d2 d3; do MMAE_HIP_LIB=${l:+tools/probes/libmmae_epi_$l.so} python tools/probes/geglu_epi_probe.py; done

With --dump the fusion-rows h / g go to gpurun_out/epi_<library>.pt (bitwise comparison of two builds on the same inputs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd import ops  # noqa: E402

B, N_, P, D, F = 256, 384, 256, 768, 2048
R, RF = B * (N_ + P), B * P


def timed(fn, it=10, rounds=5):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best * 1e3


def main():
    tag = os.path.basename(os.environ.get("MMAE_HIP_LIB", "product"))
    out = []
    for name, M in (("rows", R), ("fusion", RF)):
        g0 = torch.Generator(device="cuda").manual_seed(5)
        y = (torch.rand(M, D, device="cuda", generator=g0) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(2 * F, D, device="cuda", generator=g0) * 2 - 1) * 0.08).to(torch.bfloat16)
        h = torch.empty(M, 2 * F, device="cuda", dtype=torch.bfloat16)
        g = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
        assert ops.own_geglu_ok(y, w, h, g)
        fused = lambda: ops.gemm_geglu(y, w, h, g)
        plain = lambda: ops.gemm_nt(y, w, out=h)
        for _ in range(3):
            fused(); plain()
        tf, tp = [], []
        for _ in range(3):
            tf.append(timed(fused)); tp.append(timed(plain))
        fl = 2.0 * M * 2 * F * D
        out.append("%s %-7s fused %7.1f us (%5.0f TF/s)  plain %7.1f us (%5.0f TF/s)  epilogue cost %6.1f us" %
                   (tag, name, min(tf), fl / min(tf) / 1e6, min(tp), fl / min(tp) / 1e6, min(tf) - min(tp)))
        if "--dump" in sys.argv and name == "fusion":
            fused(); torch.cuda.synchronize()
            torch.save({"h": h.cpu(), "g": g.cpu()}, os.path.join(ROOT, "gpurun_out", "epi_%s.pt" % tag))
    print("\n".join(out), flush=True)


if __name__ == "__main__":
    main()
