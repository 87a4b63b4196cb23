#!/usr/bin/env python
"""Micro-benchmark of the masked attention kernels at the bench workload's shapes (one process, interleaved variants).

    python tools/bench_attn.py [--B 256] [--variants 0,1,...]      variant = per-call kernel variant (csrc/mmae_internal.h)
"""
import argparse, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from incomplete_multimodal_fusion_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--H", type=int, default=8)
ap.add_argument("--dh", type=int, default=64)
ap.add_argument("--variants", default="0", help="0: product kernels (dh 64 forward = 32x32x16, backward = the fused dQ + dK + dV kernel), 2: round-1 "
                "16x16x32 forward, 4: 256-query tiles, 5: sample-head dQ + dK/dV pair, 50: fused backward, 51-54: its diagnostics, 55: row constants formed in the kernel (no pre-pass)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--split", default="128,128,128", help="kept tokens per modality (N = sum), P = 256 fusion tokens")
ap.add_argument("--dirichlet", type=int, default=None, metavar="SEED", help="per-sample splits of N = sum(--split) kept tokens drawn from "
                "Dirichlet(1) like the pretraining masks (samples then differ in work: what the bench step runs), instead of --split for every sample")
ap.add_argument("--lib", default=None, help="another build of libmmae_hip.so (A/B runs inside one gpurun call)")
ap.add_argument("--pair", action="store_true", help="variant 0 = the dQ + dK/dV kernel PAIR (ops.MHA_FUSED_BWD = False) instead of the fused "
                "backward the product path runs; variant 50 is the fused kernel either way")
a = ap.parse_args()
if a.pair:
    ops.MHA_FUSED_BWD = False
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
dev = "cuda"
B, H, dh, P = a.B, a.H, a.dh, 256
nm = [int(x) for x in a.split.split(",")]
N = sum(nm); S = N + P; I = H * dh
lens = torch.tensor([nm + [P]] * B, dtype=torch.int32)
if a.dirichlet is not None:
    gen = torch.Generator().manual_seed(a.dirichlet)
    pr = torch._sample_dirichlet(torch.ones(B, len(nm), dtype=torch.float64), generator=gen)
    # generate_random_masks (MM/multimae_crossattn.py:183-235): round(share * N) kept per modality, at most the P patches it has; the first N of
    # "kept tokens in random order, then masked tokens in random order" are encoded, so a shortfall is filled uniformly from the masked ones
    cnt = torch.clamp(torch.round(pr * N), max=P).to(torch.int64)
    for b in range(B):
        while int(cnt[b].sum()) > N:
            cnt[b, int(cnt[b].argmax())] -= 1
        short = N - int(cnt[b].sum())
        if short > 0:
            pool = torch.repeat_interleave(torch.arange(len(nm)), P - cnt[b])
            pick = pool[torch.randperm(len(pool), generator=gen)[:short]]
            cnt[b] += torch.bincount(pick, minlength=len(nm))
    lens[:, :len(nm)] = cnt.to(torch.int32)
st = torch.zeros_like(lens)
for b in range(B):
    off = 0
    for s_ in range(len(nm)):
        st[b, s_] = b * N + off; off += int(lens[b, s_])
    st[b, len(nm)] = B * N + b * P
seg = ops.Segments(st.to(dev), lens.to(dev), S)
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * I, device=dev).to(torch.bfloat16).requires_grad_()
g = torch.randn(B * S, I, device=dev).to(torch.bfloat16)
per_sample = (lens[:, :len(nm)].long() ** 2).sum(1) + P * S
pairs = float(per_sample.double().mean())
flops_fwd = 4.0 * dh * H * pairs * B
if a.dirichlet is not None:
    print("score cells per sample: mean %.0f  max %.0f (%.2fx)  min %.0f" % (pairs, float(per_sample.max()), float(per_sample.max()) / pairs,
                                                                        float(per_sample.min())), flush=True)
lib = _lib.lib()
has_var = True

def run(variant):
    out = ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=variant)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(a.iters):
        out = ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=variant)
    e[1].record()
    for _ in range(a.iters):
        qkv.grad = None
        out.backward(g, retain_graph=True)
    e[2].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / a.iters * 1e3, e[1].elapsed_time(e[2]) / a.iters * 1e3

variants = [int(v) for v in a.variants.split(",")]
if has_var and len(variants) > 1:       # agreement of every variant's forward output with the first one
    ref = ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=variants[0]).float()
    for v in variants[1:]:
        o = ops.mha_self(qkv, H, dh, seg, dh ** -0.5, variant=v).float()
        print("variant %d vs %d: max |diff| %.3e" % (v, variants[0], float((o - ref).abs().max())), flush=True)
res = {v: [] for v in variants}
for r in range(a.rounds):
    for v in variants:
        res[v].append(run(v))
for v in variants:
    f = sorted(x[0] for x in res[v]); bw = sorted(x[1] for x in res[v])
    fm, bm = f[len(f) // 2], bw[len(bw) // 2]
    print("variant %d: fwd %7.1f us (%6.1f TF/s mask-aware)   bwd(dq+dkdv) %7.1f us (%6.1f TF/s at 3.5x fwd flops)   [min fwd %.1f bwd %.1f]"
          % (v, fm, flops_fwd / fm / 1e6, bm, 3.5 * flops_fwd / bm / 1e6, f[0], bw[0]), flush=True)
