#!/usr/bin/env python
"""Diagnostic: the fuzz cases of tests/test_gpu_fuzz.py::test_fuzz_attention_segments run with two kernel variants on the
same inputs; prints each variant's error against the fp64 dense restatement, per case.   python tools/probes/attn_variant_diff.py 0 5"""
import random, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from incomplete_multimodal_fusion_amd import ops
from tests.test_gpu_kernels import dense_attention_ref
DEV = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [0, 5]
T = torch.bfloat16
rng = random.Random(1234)
torch.manual_seed(77)
for case in range(24):
    dh = 64
    H = rng.choice([1, 2, 3, 8]); nseg = rng.randint(1, 5); B = rng.randint(1, 4)
    empty_mode = rng.choice([0, 1]); same = rng.random() < 0.5
    pick = lambda: rng.choice([0, 0, 1, 7, 63, 64, 65, 100, 128, 129, 191, 200, 257])
    qlens = torch.tensor([[pick() for _ in range(nseg)] for _ in range(B)], dtype=torch.int32)
    klens = qlens.clone() if same else torch.tensor([[pick() for _ in range(nseg)] for _ in range(B)], dtype=torch.int32)
    if int(qlens.sum()) == 0:
        qlens[0, -1] = 5
        if same:
            klens = qlens.clone()
    I = H * dh

    def starts(lens, gap):
        st = torch.zeros_like(lens); r = 0
        for b in range(B):
            for s in range(nseg):
                st[b, s] = r; r += int(lens[b, s]) + gap
        return st, max(r, 1)
    gap = rng.choice([0, 0, 3])
    qst, nq = starts(qlens, gap)
    kst, nk = (qst, nq) if same else starts(klens, gap)
    q = torch.randn(nq, I); kv = torch.randn(nk, 2 * I); g = torch.randn(nq, I)
    qseg = ops.Segments(qst.to(DEV), qlens.to(DEV), max(int(qlens.sum(1).max()), 1), covers_all=gap == 0 and nq == int(qlens.sum()))
    kseg = ops.Segments(kst.to(DEV), klens.to(DEV), max(int(klens.sum(1).max()), 1), covers_all=gap == 0 and nk == int(klens.sum()))
    scale = dh ** -0.5
    q64 = q.to(T).double().reshape(nq, H, dh).requires_grad_()
    kv64 = kv.to(T).double()
    k64 = kv64[:, :I].reshape(nk, H, dh).clone().requires_grad_(); v64 = kv64[:, I:].reshape(nk, H, dh).clone().requires_grad_()
    ref = dense_attention_ref(q64, k64, v64, (qst, qlens), (kst, klens), scale, empty_mode)
    if ref.requires_grad:
        ref.backward(g.to(T).double().reshape(nq, H, dh))
    gq = q64.grad if q64.grad is not None else torch.zeros_like(q64)
    line = "case %2d H %d nseg %d B %d mode %d same %d q %s k %s:" % (case, H, nseg, B, empty_mode, same, qlens.tolist(), klens.tolist())
    for v in variants:
        qd = q.to(DEV, T).requires_grad_(); kvd = kv.to(DEV, T).requires_grad_()
        out = ops.mha_cross(qd, kvd, H, dh, qseg, kseg, scale, empty_mode, variant=v)
        out.backward(g.to(DEV, T))
        eo = (out.detach().cpu().double() - ref.detach().reshape(nq, I)).abs()
        eq = (qd.grad.cpu().double() - gq.reshape(nq, I)).abs()
        so = max(float(ref.abs().max()), 1e-6); sq = max(float(gq.abs().max()), 1e-6)
        line += "  [v%d out %.2e dq %.2e (row %d)]" % (v, float(eo.max()) / so, float(eq.max()) / sq, int(eq.max(1).values.argmax()))
    print(line, flush=True)
