#!/usr/bin/env python
"""Cost of the attention-pooling head computed WITHOUT the K/V projection of every token row (timing probe, B = 256 bench shapes).

The return-token queries are the same for every sample, so  q_h . (W_k[h] x)  =  (q_h W_k[h]) . x : scores need one (rows, 768) x (768, 56)
product (7 queries x 8 heads), the weighted sums one batched (56, 640) x (640, 768) product per sample, and W_v is applied to the 56 pooled
rows -- instead of projecting all B * 640 rows to 1024 columns (forward, input gradient, weight gradient) and running the attention kernels
on 7 query rows per sample.  Prints the time of the library ops of that formulation, forward + backward, next to the projection GEMMs alone
of the current path (its attention kernels and the row gather come on top: profiles/r05_kernel_stats.md)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from incomplete_multimodal_fusion_amd import ops  # noqa: E402

B, N, P, D, H, dh, R = 256, 384, 256, 768, 8, 64, 7
S, I, HR = N + P, H * dh, H * R
dev = "cuda"


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
    torch.manual_seed(0)
    T = torch.bfloat16
    z = (torch.randn(B * S, D, device=dev) * 0.5).to(T)
    wkv = (torch.randn(2 * I, D, device=dev) * 0.03).to(T)
    wkvt = wkv.t().contiguous()
    q = (torch.randn(R, I, device=dev) * 0.5).to(T)
    allow = torch.rand(B, R, S, device=dev) > 0.4
    allow[:, 3] = True
    allow_h = allow.unsqueeze(1).expand(B, H, R, S).reshape(B, HR, S)
    dA = (torch.randn(B, R, I, device=dev) * 0.1).to(T)
    zm, zf = z[:B * N].view(B, N, D), z[B * N:].view(B, P, D)

    # ---- current path: the projection GEMMs alone
    kv = torch.empty(B * S, 2 * I, device=dev, dtype=T)
    gz = torch.empty(B * S, D, device=dev, dtype=T)
    gkv = torch.randn(B * S, 2 * I, device=dev).to(T)
    t_fwd = timed(lambda: ops.gemm_nt(z, wkv, out=kv))
    t_dg = timed(lambda: ops.gemm_nt(gkv, wkvt, out=gz))
    S_ = 8
    t_wg = timed(lambda: torch.bmm(gkv.view(S_, B * S // S_, 2 * I).transpose(1, 2), z.view(S_, B * S // S_, D)))
    print("current: kv projection %.0f us, its input gradient %.0f us, its weight gradient (8 splits, no sum) %.0f us  -> %.0f us + attention kernels"
          % (t_fwd, t_dg, t_wg, t_fwd + t_dg + t_wg), flush=True)

    # ---- low-rank formulation, forward
    wk, wv = wkv[:I].view(H, dh, D), wkv[I:].view(H, dh, D)
    qh = (q.view(R, H, dh).permute(1, 0, 2) * dh ** -0.5).contiguous()           # (H, R, dh)
    st = {}

    def fwd():
        qp = torch.zeros(64, D, device=dev, dtype=T)
        qp[:HR] = torch.bmm(qh, wk).reshape(HR, D)                                  # Q' = q_h W_k[h]
        zz = z @ qp.t()                                                            # (rows, 64)
        sc = torch.cat([zz[:B * N].view(B, N, 64), zz[B * N:].view(B, P, 64)], 1)[:, :, :HR].transpose(1, 2).float()
        pm = torch.softmax(sc.masked_fill(~allow_h, float("-inf")), -1).to(T)      # (B, HR, S)
        xb = torch.bmm(pm[:, :, :N], zm)
        xb.baddbmm_(pm[:, :, N:], zf)                                              # (B, HR, D)
        a = torch.bmm(xb.view(B, H, R, D).permute(1, 0, 2, 3).reshape(H, B * R, D), wv.transpose(1, 2))   # (H, B R, dh)
        st.update(qp=qp, pm=pm, xb=xb)
        return a
    t_f = timed(fwd)

    def bwd():
        qp, pm, xb = st["qp"], st["pm"], st["xb"]
        dah = dA.view(B, R, H, dh).permute(2, 0, 1, 3).reshape(H, B * R, dh)
        dxb = torch.bmm(dah, wv).view(H, B, R, D).permute(1, 0, 2, 3).reshape(B, HR, D)
        dwv = torch.bmm(dah.transpose(1, 2), xb.view(B, H, R, D).permute(1, 0, 2, 3).reshape(H, B * R, D))
        dp = torch.cat([torch.bmm(dxb, zm.transpose(1, 2)), torch.bmm(dxb, zf.transpose(1, 2))], 2).float()
        pf = pm.float()
        ds = pf * (dp - (dp * pf).sum(-1, keepdim=True))
        g = torch.empty(B * S, D, device=dev, dtype=T)
        torch.bmm(pm[:, :, :N].transpose(1, 2), dxb, out=g[:B * N].view(B, N, D))
        torch.bmm(pm[:, :, N:].transpose(1, 2), dxb, out=g[B * N:].view(B, P, D))
        dz = torch.zeros(B * S, 64, device=dev, dtype=T)
        dsT = ds.transpose(1, 2).to(T)                                             # (B, S, HR)
        dz[:B * N].view(B, N, 64)[:, :, :HR] = dsT[:, :N]
        dz[B * N:].view(B, P, 64)[:, :, :HR] = dsT[:, N:]
        g.addmm_(dz, qp)
        dqp = torch.bmm(dz.view(8, B * S // 8, 64).transpose(1, 2), z.view(8, B * S // 8, D)).float().sum(0)
        return g, dwv, dqp
    t_b = timed(bwd)
    print("low rank: forward %.0f us, backward %.0f us  -> %.0f us" % (t_f, t_b, t_f + t_b), flush=True)


if __name__ == "__main__":
    main()
