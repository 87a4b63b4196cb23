"""Seeded random-configuration parity (GPU tier): many small shapes per kernel, lengths crossing tile boundaries, empty
segments / dropped modalities / empty mask rows, odd widths.  Checkers: the fp64 dense restatement of the segment rule
(tests/test_gpu_kernels.py) and the CPU oracle."""
import random

import pytest
import torch

from oracle import mmae_oracle as O
from tests.test_gpu_kernels import COLSUM, DEV, GRAD, close, dense_attention_ref

pytestmark = pytest.mark.gpu


# hpb: heads one workgroup of the sample-head kernels walks (csrc/mmae_internal.h); 0 = the product's choice (1 at these batch
# sizes).  With hpb = 2 / 8 every case runs H = 8 -- the head loop and the cross-head prefetch of the bench configuration.
@pytest.mark.parametrize("T,variant,hpb", [(torch.float32, 0, 0), (torch.bfloat16, 0, 0), (torch.bfloat16, 2, 0), (torch.bfloat16, 4, 0),
                                           (torch.bfloat16, 5, 0), (torch.bfloat16, 23, 0), (torch.bfloat16, 0, 1), (torch.bfloat16, 0, 2),
                                           (torch.bfloat16, 0, 8), (torch.bfloat16, 5, 2), (torch.bfloat16, 5, 8), (torch.bfloat16, 50, 0), (torch.bfloat16, 50, 2),
                                           (torch.bfloat16, 50, 8)])
def test_fuzz_attention_segments(T, variant, hpb):
    from incomplete_multimodal_fusion_amd import ops
    rng = random.Random(1234)
    torch.manual_seed(77)
    for case in range(24):
        dh = rng.choice([32, 64]) if (variant == 0 and not hpb) else 64
        H = rng.choice([1, 2, 3, 8]); nseg = rng.randint(1, 5); B = rng.randint(1, 4)
        if hpb:
            H = 8
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
        gap = rng.choice([0, 0, 3])                       # rows outside every segment (covers_all=False): outputs / grads 0
        qst, nq = starts(qlens, gap)
        kst, nk = (qst, nq) if same else starts(klens, gap)
        q = torch.randn(nq, I); kv = torch.randn(nk, 2 * I); g = torch.randn(nq, I)
        qd = q.to(DEV, T).requires_grad_(); kvd = kv.to(DEV, T).requires_grad_()
        qseg = ops.Segments(qst.to(DEV), qlens.to(DEV), max(int(qlens.sum(1).max()), 1), covers_all=gap == 0 and nq == int(qlens.sum()))
        kseg = ops.Segments(kst.to(DEV), klens.to(DEV), max(int(klens.sum(1).max()), 1), covers_all=gap == 0 and nk == int(klens.sum()))
        scale = dh ** -0.5
        out = ops.mha_cross(qd, kvd, H, dh, qseg, kseg, scale, empty_mode, variant=variant, hpb=hpb)
        out.backward(g.to(DEV, T))
        q64 = qd.detach().cpu().double().reshape(nq, H, dh).requires_grad_()
        kv64 = kvd.detach().cpu().double()
        k64 = kv64[:, :I].reshape(nk, H, dh).clone().requires_grad_(); v64 = kv64[:, I:].reshape(nk, H, dh).clone().requires_grad_()
        ref = dense_attention_ref(q64, k64, v64, (qst, qlens), (kst, klens), scale, empty_mode)
        if ref.requires_grad:
            ref.backward(g.to(T).double().reshape(nq, H, dh))
        tol = 2e-5 if T == torch.float32 else 1e-2
        tag = "case %d (dh %d H %d nseg %d B %d mode %d)" % (case, dh, H, nseg, B, empty_mode)
        qmask = torch.ones(nq, dtype=torch.bool)
        close(out[qmask.to(DEV)], ref.reshape(nq, I)[qmask], tol, "out " + tag)
        gq = q64.grad if q64.grad is not None else torch.zeros_like(q64)
        gk = k64.grad if k64.grad is not None else torch.zeros_like(k64)
        gv = v64.grad if v64.grad is not None else torch.zeros_like(v64)
        close(qd.grad[qmask.to(DEV)], gq.reshape(nq, I)[qmask], tol * GRAD, "dq " + tag)
        close(kvd.grad[:, :I], gk.reshape(nk, I), tol * GRAD, "dk " + tag)
        close(kvd.grad[:, I:], gv.reshape(nk, I), tol * GRAD, "dv " + tag)


def test_fuzz_masks_from_draws_bit_exact():
    from incomplete_multimodal_fusion_amd import ops
    rng = random.Random(99)
    g = torch.Generator().manual_seed(5)
    for case in range(30):
        M = rng.randint(1, 4); P = rng.choice([4, 16, 49, 64, 256]); R = rng.choice([1, 2, 7])
        N = rng.randint(1, M * P)
        alpha = torch.rand(R, M, generator=g) + 0.05
        if rng.random() < 0.4:
            alpha[rng.randrange(R), rng.randrange(M)] = 0.0            # a dropped modality (sample_tasks_uniformly path)
        dirichlet = alpha / alpha.sum(1, keepdim=True)
        noise = torch.rand(R, M, P, generator=g)
        if rng.random() < 0.3:
            noise = (noise * 4).floor() / 4                             # ties: stable order required
        noise_all = torch.rand(R, M * P, generator=g)
        got = ops.masks_from_draws(dirichlet.to(DEV), noise.to(DEV), noise_all.to(DEV), N)
        exp = O.masks_from_draws(dirichlet, noise, noise_all, N)
        for a, b, nm in zip(got, exp, ("mask_all", "ids_keep", "ids_restore")):
            assert torch.equal(a.cpu(), b), (case, nm, M, P, R, N)


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16])
def test_fuzz_add_double_layernorm_parts(T):
    from incomplete_multimodal_fusion_amd import ops
    rng = random.Random(7)
    torch.manual_seed(8)
    for case in range(16):
        D = rng.choice([48, 256, 768, 1024]); nparts = rng.randint(1, 3)
        rows = [rng.choice([0, 1, 3, 64, 257, 1000]) for _ in range(nparts)]
        if sum(rows) == 0:
            rows[0] = 5
        dbl = rng.random() < 0.7; with_beta = rng.random() < 0.4
        has_delta = [rng.random() < 0.6 for _ in range(nparts)]
        if not any(has_delta):
            has_delta[0] = True
        xs = [torch.randn(r, D) for r in rows]
        nd = sum(r for r, h in zip(rows, has_delta) if h)
        delta = torch.randn(nd, D)
        offs, o = [], 0
        for r, h in zip(rows, has_delta):
            offs.append(o if h else -1); o += r if h else 0
        g1 = torch.randn(D) * 0.3 + 1; b1 = torch.randn(D) * 0.1 if with_beta else None
        g2 = torch.randn(D) * 0.3 + 1 if dbl else None; b2 = torch.randn(D) * 0.1 if (dbl and with_beta) else None
        leaf = lambda t: None if t is None else t.clone().to(DEV).requires_grad_()
        xd = [leaf(x) for x in xs]; dd = delta.to(DEV, T).requires_grad_()
        pg = [leaf(t) for t in (g1, b1, g2, b2)]
        xn, y = ops.parts_add_ln(xd, dd, offs, pg[0], pg[1], pg[2], pg[3], out_dtype=T)
        gy = torch.randn(sum(rows), D); gx = [torch.randn(r, D) for r in rows]
        (y.float() * gy.to(DEV)).sum().backward(retain_graph=True)
        sum((a * b.to(DEV)).sum() for a, b in zip(xn, gx)).backward()
        # reference in fp64 on the values the kernel saw
        x64 = [x.double().requires_grad_() for x in xs]; d64 = dd.detach().cpu().double().requires_grad_()
        r64 = [None if t is None else t.double().requires_grad_() for t in (g1, b1, g2, b2)]
        news = []
        for x, off, r in zip(x64, offs, rows):
            news.append(x + d64[off:off + r] if off >= 0 else x)
        z = torch.cat(news, 0)
        yy = torch.nn.functional.layer_norm(z, (D,), r64[0], r64[1], 1e-5)
        if dbl:
            yy = torch.nn.functional.layer_norm(yy, (D,), r64[2], r64[3], 1e-5)
        (yy * gy.double()).sum().backward(retain_graph=True)
        sum((a * b.double()).sum() for a, b in zip(news, gx)).backward()
        tol = 2e-5 if T == torch.float32 else 1e-2
        tag = "case %d D %d rows %s dbl %s" % (case, D, rows, dbl)
        close(y, yy, tol, "y " + tag)
        for a, b in zip(xn, news):
            close(a, b, 2e-5 if T == torch.float32 else 4e-3, "x_new " + tag)
        for a, b in zip(xd, x64):
            if b.shape[0]:
                close(a.grad, b.grad, tol * GRAD, "gx " + tag)
        close(dd.grad, d64.grad, tol * GRAD, "gdelta " + tag)
        for a, b, nm in zip(pg, r64, ("g1", "b1", "g2", "b2")):
            if a is not None:
                close(a.grad, b.grad, tol * COLSUM, nm + " " + tag)


def test_fuzz_masked_losses_and_patchify():
    from incomplete_multimodal_fusion_amd import ops
    rng = random.Random(21)
    torch.manual_seed(22)
    for case in range(12):
        B = rng.randint(1, 5); C = rng.choice([1, 3]); ps = rng.choice([4, 16]); nh = rng.choice([2, 4, 7])
        Hh = nh * ps; P = nh * nh
        pred = torch.randn(B, C, Hh, Hh); tgt = torch.randn(B, C, Hh, Hh)
        mask = (torch.rand(B, P) < 0.5).long()
        if rng.random() < 0.5:
            mask[rng.randrange(B)] = 0                                  # a sample with nothing masked out
        if int(mask.sum()) == 0:
            mask[0, 0] = 1
        for kind in (0, 1):
            pd = pred.to(DEV).requires_grad_()
            got = ops.masked_loss_image(pd, tgt.to(DEV), mask.to(DEV), kind, ps)
            got.backward()
            p64 = pred.double().requires_grad_()
            m = mask.view(B, 1, nh, nh).double().repeat_interleave(ps, 2).repeat_interleave(ps, 3)
            e = (p64 - tgt.double()) ** 2 if kind == 0 else (p64 - tgt.double()).abs()
            den = m.flatten(1).sum(1)
            valid = den > 0
            per = (e.mean(1, keepdim=True) * m).flatten(1).sum(1)[valid] / den[valid]
            ref = per.mean()
            ref.backward()
            close(got, ref, 1e-5, "loss %d kind %d" % (case, kind))
            close(pd.grad, p64.grad, 1e-5, "dloss %d kind %d" % (case, kind))
        # unpatchify(tokens) is the exact inverse of the (c ph pw) patch order
        tok = torch.randn(B * P, C * ps * ps)
        img = ops.unpatchify(tok.to(DEV), B, C, Hh, Hh, ps).cpu()
        ref = tok.view(B, nh, nh, C, ps, ps).permute(0, 3, 1, 4, 2, 5).reshape(B, C, Hh, Hh)
        assert torch.equal(img, ref), case


def test_fuzz_end_to_end_vs_oracle():
    """Whole step (forward, all losses, every parameter gradient) in fp32 mode on random small architectures and random
    shared masks, including dropped modalities and a single kept token."""
    from tests.test_cabi_symbols import build_model
    from tests.test_gpu_e2e import native_step
    rng = random.Random(31)
    for case in range(6):
        torch.manual_seed(100 + case)
        dh = rng.choice([32, 64]); heads = rng.choice([1, 2, 3]); dim = rng.choice([32, 48, 96])
        nh = rng.choice([2, 3, 4]); img = 16 * nh; P = nh * nh
        ddh = rng.choice([32, 64]); dheads = rng.choice([1, 2])
        cfg = dict(dim_tokens=dim, depth=rng.choice([1, 2, 3]), dim_head=dh, heads=heads, image_size=img, patch_size=16,
                   decoder_dim=ddh * dheads, decoder_depth=rng.choice([1, 2]), decoder_heads=dheads)
        channels = (("s1", 1), ("s2", 3), ("dem", 1))
        model = build_model(cfg, channels)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.requires_grad and (n.endswith("gamma") or "norm" in n and n.endswith("weight")):
                    p.add_(0.2 * torch.randn_like(p))
            model.mask_embedding.add_(0.05 * torch.randn_like(model.mask_embedding))
        B = rng.randint(1, 3)
        keep = [rng.randint(0, P) for _ in range(3)]
        if rng.random() < 0.4:
            keep[rng.randrange(3)] = 0                                  # a dropped modality
        if sum(keep) == 0:
            keep[1] = 1
        N = sum(keep)
        x = {d: torch.randn(B, c, img, img) for d, c in channels}
        masks = {}
        for (d, _), k in zip(channels, keep):
            row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
            masks[d] = row[None].repeat(B, 1)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("pos_emb") and not k.endswith("beta"))
             for k, v in state.items()}
        out_r, (tl_r, lc_r, loss_r) = O.train_step_loss(p, x, masks, N, heads, dheads, 16)
        loss_r.backward()
        model.to(DEV).train()
        xd = {k: v.to(DEV) for k, v in x.items()}; md = {k: v.to(DEV) for k, v in masks.items()}
        out, tl, lc, loss = native_step(model, xd, md, N, bool(case % 2), False)
        tag = " case %d %s keep %s B %d" % (case, cfg, keep, B)
        for d in O.DOMAINS:
            pred = out[0][d].image() if hasattr(out[0][d], "image") else out[0][d]
            close(pred, out_r[0][d], 1e-3, "pred " + d + tag)
            close(tl[d], tl_r[d], 1e-3, "loss " + d + tag)
        close(out[2], out_r[2], 1e-3, "pooled" + tag); close(out[4], out_r[4], 1e-3, "fusion" + tag)
        close(lc, lc_r, 1e-3, "contra" + tag); close(loss, loss_r, 1e-3, "loss" + tag)
        loss.backward()
        bad = []
        gscale = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None and not torch.isnan(v.grad).any())
        for n, prm in model.named_parameters():
            ref = p[n].grad
            if ref is None:
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, n
                continue
            if torch.isnan(ref).any():                                   # reference 0/0 (a sample with nothing masked)
                continue
            # a gradient that is analytically zero (e.g. the pooling query of a modality with ONE kept token: softmax
            # over one key) is rounding noise on both sides: judge it against the step's overall gradient scale
            err = float((prm.grad.detach().cpu().double() - ref.double()).abs().max())
            if err > 2e-3 * max(float(ref.abs().max()), 1e-4 * gscale):
                bad.append("%s: err %.3e ref max %.3e%s" % (n, err, float(ref.abs().max()), tag))
        assert not bad, bad[:6]
