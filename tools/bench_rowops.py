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
    add_ln(args, call, ptr, stream, _lib.lib())
    modattn(args)


def add_ln(args, call, ptr, stream, lib):
    """The encoder's residual add + double LayerNorm at the bench shape (rows x 768, bf16 delta / y, fp32 stream)."""
    dev, rows, D = "cuda:0", args.rows, 768
    x = torch.randn(rows, D, device=dev); delta = torch.randn(rows, D, device=dev).to(torch.bfloat16)
    xn = torch.empty_like(x); y = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    g1 = torch.rand(D, device=dev) + 0.5; g2 = torch.rand(D, device=dev) + 0.5
    stats = torch.empty(rows, 4, device=dev)
    fwd = lambda: call("mmae_add_ln_fwd", 1, 1, rows, D, ptr(x), ptr(delta), ptr(xn), ptr(y), ptr(g1), None, 1e-5, ptr(g2), None,
                       1e-5, ptr(stats), stream())
    t_f = timeit(fwd)
    gy = torch.randn(rows, D, device=dev).to(torch.bfloat16); gup = torch.randn(rows, D, device=dev)
    gx = torch.empty_like(x); gd = torch.empty_like(delta)
    ws = torch.empty(lib.mmae_add_ln_bwd_ws_floats(rows, D), device=dev)
    dg1 = torch.zeros(D, device=dev); dg2 = torch.zeros(D, device=dev)
    bwd = lambda: call("mmae_add_ln_bwd", 1, 1, rows, D, ptr(xn), ptr(gy), ptr(gup), ptr(g1), None, ptr(g2), ptr(stats), ptr(gx),
                       ptr(gd), ptr(dg1), None, ptr(dg2), None, ptr(ws), 0, stream())
    t_b = timeit(bwd)
    bf, bb = rows * D * (4 + 2 + 4 + 2), rows * D * (4 + 2 + 4 + 4 + 2)
    print("            add_ln_fwd %.1f us (%.2f TB/s)   add_ln_bwd (gx_up, gx, gdelta) %.1f us (%.2f TB/s)" %
          (t_f, bf / t_f / 1e6, t_b, bb / t_b / 1e6))
    # dual double-LayerNorm of the modality rows (B * N = 98304 rows at the bench batch)
    rows = rows * 3 // 5
    x = x[:rows]; delta = delta[:rows]; xn = xn[:rows]; y = y[:rows]; gy = gy[:rows]; gup = gup[:rows]; gx = gx[:rows]; gd = gd[:rows]
    yb = torch.empty_like(y); gyb = torch.randn(rows, D, device=dev).to(torch.bfloat16)
    g1b = torch.rand(D, device=dev) + 0.5; g2b = torch.rand(D, device=dev) + 0.5
    stats_b = torch.empty(rows, 4, device=dev)
    dg1b = torch.zeros(D, device=dev); dg2b = torch.zeros(D, device=dev)
    fwd = lambda: call("mmae_add_ln_fwd_dual", 1, 1, rows, D, ptr(x), ptr(delta), ptr(xn), ptr(y), ptr(yb), ptr(g1), ptr(g2), ptr(g1b),
                       ptr(g2b), 1e-5, 1e-5, ptr(stats), ptr(stats_b), stream())
    t_f = timeit(fwd)
    bwd = lambda: call("mmae_add_ln_bwd_dual", 1, 1, rows, D, ptr(xn), ptr(gy), ptr(gyb), ptr(gup), ptr(g1), ptr(g2), ptr(g1b), ptr(g2b),
                       ptr(stats), ptr(stats_b), ptr(gx), ptr(gd), ptr(dg1), ptr(dg2), ptr(dg1b), ptr(dg2b), ptr(ws), 0, stream())
    t_b = timeit(bwd)
    bf, bb = rows * D * (4 + 2 + 4 + 2 + 2), rows * D * (4 + 2 + 2 + 4 + 4 + 2)
    print("            add_ln_fwd_dual %.1f us (%.2f TB/s)   add_ln_bwd_dual %.1f us (%.2f TB/s)   [%d rows]" %
          (t_f, bf / t_f / 1e6, t_b, bb / t_b / 1e6, rows))


def modattn(args):
    """Modality attention of Block_Fusion at the bench shape: B = 256, P = 256, M + 1 = 4 slots, I = 512, a slot is a kept
    token row with probability 1/2, else the patch's shared mask-embedding row."""
    from incomplete_multimodal_fusion_amd import ops
    dev, B, P, ns, H, dh = "cuda:0", 256, 256, 4, 8, 64
    I = H * dh
    BN, BP = B * 384, B * P
    g = torch.Generator(device="cpu").manual_seed(0)
    kv = torch.randn(BN + BP + P, 2 * I, device=dev).to(torch.bfloat16).requires_grad_()
    q = torch.randn(BP, I, device=dev).to(torch.bfloat16).requires_grad_()
    shared = BN + BP
    slot = torch.empty(BP, ns, dtype=torch.int32)
    tok = torch.randperm(BN, generator=g)[:BP * (ns - 1)].reshape(BP, ns - 1).to(torch.int32) if BN >= BP * (ns - 1) else \
        torch.randint(0, BN, (BP, ns - 1), generator=g, dtype=torch.int32)
    keep = torch.rand(BP, ns - 1, generator=g) < 0.5
    pidx = (torch.arange(BP) % P).to(torch.int32)[:, None]
    slot[:, :ns - 1] = torch.where(keep, tok, shared + pidx)
    slot[:, ns - 1] = BN + torch.arange(BP, dtype=torch.int32)
    slot = slot.to(dev)
    with torch.no_grad():
        t_f = timeit(lambda: ops.modattn(q, kv, slot, B, P, ns, H, dh, shared, dh ** -0.5))
    out = ops.modattn(q, kv, slot, B, P, ns, H, dh, shared, dh ** -0.5)
    go = torch.randn_like(out)
    t_b = timeit(lambda: torch.autograd.grad(out, (q, kv), go, retain_graph=True))
    print("            modattn_fwd %.1f us   modattn bwd (incl. finish) %.1f us" % (t_f, t_b))


if __name__ == "__main__":
    main()
