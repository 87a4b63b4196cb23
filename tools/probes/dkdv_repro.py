#!/usr/bin/env python
"""Repro driver for the sample-head dK/dV kernel: the ragged-segment case of tests/test_gpu_kernels.py, step by step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from incomplete_multimodal_fusion_amd import ops
from tests.test_gpu_kernels import dense_attention_ref
DEV = "cuda"; T = torch.bfloat16
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
which = sys.argv[2] if len(sys.argv) > 2 else "all"
torch.manual_seed(3)
H, nseg, dh = 3, 4, 64
I = H * dh
qlens = torch.tensor([[70, 1, 130, 65], [0, 64, 63, 129], [5, 0, 0, 3]], dtype=torch.int32)
klens = torch.tensor([[70, 0, 131, 65], [3, 64, 0, 200], [0, 0, 0, 0]], dtype=torch.int32)
if which.startswith("q="):          # e.g.  q=70,1,130,65 k=70,0,131,65
    qlens = torch.tensor([[int(x) for x in which[2:].split(",")]], dtype=torch.int32)
    klens = torch.tensor([[int(x) for x in sys.argv[3][2:].split(",")]], dtype=torch.int32)
    nseg = qlens.shape[1]
elif which != "all":
    b = int(which); qlens = qlens[b:b + 1]; klens = klens[b:b + 1]
B = qlens.shape[0]
def starts(lens):
    st = torch.zeros_like(lens); r = 0
    for b in range(B):
        for s in range(nseg):
            st[b, s] = r; r += int(lens[b, s])
    return st, max(r, 1)
qst, nq = starts(qlens); kst, nk = starts(klens)
q = torch.randn(nq, I); kv = torch.randn(nk, 2 * I); g = torch.randn(nq, I)
qd = q.to(DEV, T).requires_grad_(); kvd = kv.to(DEV, T).requires_grad_()
qseg = ops.Segments(qst.to(DEV), qlens.to(DEV), int(qlens.sum(1).max()))
kseg = ops.Segments(kst.to(DEV), klens.to(DEV), max(int(klens.sum(1).max()), 1))
out = ops.mha_cross(qd, kvd, H, dh, qseg, kseg, dh ** -0.5, 0, variant=variant)
torch.cuda.synchronize(); print("forward ok", flush=True)
out.backward(g.to(DEV, T))
torch.cuda.synchronize(); print("backward ok", flush=True)
q64 = qd.detach().cpu().double().reshape(nq, H, dh).requires_grad_()
kv64 = kvd.detach().cpu().double()
k64 = kv64[:, :I].reshape(nk, H, dh).clone().requires_grad_(); v64 = kv64[:, I:].reshape(nk, H, dh).clone().requires_grad_()
ref = dense_attention_ref(q64, k64, v64, (qst, qlens), (kst, klens), dh ** -0.5, 0)
ref.backward(g.to(T).double().reshape(nq, H, dh))
for name, a, b_ in (("out", out, ref.reshape(nq, I)), ("dq", qd.grad, q64.grad.reshape(nq, I)), ("dk", kvd.grad[:, :I], k64.grad.reshape(nk, I)), ("dv", kvd.grad[:, I:], v64.grad.reshape(nk, I))):
    e = (a.detach().cpu().double() - b_.detach()).abs()
    print("%s: max err %.3e (scale %.3e) worst row %d" % (name, float(e.max()), float(b_.abs().max()), int(e.max(1).values.argmax())), flush=True)
