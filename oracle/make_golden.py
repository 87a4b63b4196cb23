"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

The fixtures are pure data (inputs, weights, expected outputs/gradients, recorded random draws).
No reference source text is written anywhere.  The committed fixtures are what pins
`oracle/mmae_oracle.py` (and, through it, the HIP path) to the reference; see DESIGN.md "Oracle".

Everything is generated with fixed torch seeds on CPU, float32 (torch 2.10.0 CPU in this image).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader  # noqa: E402

OUT = os.environ.get("MMAE_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


class Bag(dict):
    def put(self, prefix, **kw):
        for k, v in kw.items():
            self[prefix + "/" + k] = npy(v)


def grads_of(out, g, tensors):
    gs = torch.autograd.grad(out, tensors, g, allow_unused=True)
    return [torch.zeros_like(t) if x is None else x for x, t in zip(gs, tensors)]


def state(mod, prefix=""):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def rand_init_(mod, gen, scale=0.2):
    """Perturb all params (LN gammas away from 1, biases away from 0) so that fixtures pin every term."""
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if not p.requires_grad:
                continue
            p.add_(scale * torch.randn(p.shape, generator=gen) * (p.abs().mean() + 0.1))


# ----------------------------------------------------------------------------------------------
_SEED = [0]


def seeded():
    """torch.manual_seed before EVERY module construction: module initialisers draw from the global RNG, and the fixtures
    must regenerate byte for byte (`python oracle/make_golden.py` twice -> identical .npz contents)."""
    _SEED[0] += 1
    torch.manual_seed(4000 + _SEED[0])


def gen_ops(ref):
    bag = Bag()
    _SEED[0] = 0
    gen = torch.Generator().manual_seed(1234)
    R = lambda *s: torch.randn(*s, generator=gen)

    # a2: sincos pos-emb (multimae_utils.py:29-45) incl. a non-square grid (pins the meshgrid quirk)
    bag.put("sincos_4x4_32", out=ref.mu.build_2d_sincos_posemb(4, 4, 32))
    bag.put("sincos_2x3_16", out=ref.mu.build_2d_sincos_posemb(2, 3, 16))

    # a7: bias-less LayerNorm (zorro_utils.py:103-110)
    seeded()
    ln = ref.zu.LayerNorm(32); rand_init_(ln, gen)
    x = R(3, 5, 32).requires_grad_(); g = R(3, 5, 32)
    y = ln(x)
    gx, gg = grads_of(y, g, [x, ln.gamma])
    bag.put("layernorm", x=x, gamma=ln.gamma, y=y, g=g, gx=gx, ggamma=gg)

    # a8: Attention, self / zorro-masked (zorro_utils.py:170-194)
    seeded()
    attn = ref.zu.Attention(dim=32, dim_head=32, heads=2); rand_init_(attn, gen)
    types = torch.tensor([0, 0, 0, 1, 1, 2, 2, 3, 3, 3])
    zmask = (types[:, None] == types[None, :]) | (types[:, None] == 3)
    x = R(2, 10, 32).requires_grad_(); g = R(2, 10, 32)
    y = attn(x, attn_mask=zmask)
    ps = [x] + list(attn.parameters())
    gs = grads_of(y, g, ps)
    bag.put("attn_self", x=x, mask=zmask, y=y, g=g, gx=gs[0], types=types,
            **{"w." + k: v for k, v in attn.state_dict().items()},
            **{"gw." + n: gi for (n, _), gi in zip(attn.named_parameters(), gs[1:])})
    # unmasked
    y = attn(x)
    gs = grads_of(y, g, ps)
    bag.put("attn_self_nomask", y=y, gx=gs[0])

    # a13: cross attention with pool mask, one return type has no tokens (fully masked row -> uniform)
    ctx_types = torch.tensor([0, 0, 0, 0, 2, 2, 3, 3, 3, 3])  # no type-1 token
    rtypes = torch.tensor([0, 1, 2, 3])
    pmask = (rtypes[:, None] == ctx_types[None, :]) | (rtypes[:, None] == 3)
    q = R(2, 4, 32).requires_grad_(); ctx = R(2, 10, 32).requires_grad_(); g = R(2, 4, 32)
    y = attn(q, context=ctx, attn_mask=pmask)
    gs = grads_of(y, g, [q, ctx] + list(attn.parameters()))
    bag.put("attn_pool", q=q, ctx=ctx, mask=pmask, ctx_types=ctx_types, y=y, g=g, gq=gs[0], gctx=gs[1],
            **{"gw." + n: gi for (n, _), gi in zip(attn.named_parameters(), gs[2:])})
    # a15: cross attention, no mask, 1 query; and with an EMPTY context (dropped modality)
    q1 = R(2, 1, 32).requires_grad_(); g1 = R(2, 1, 32)
    ctx3 = R(2, 3, 32).requires_grad_()
    y = attn(q1, context=ctx3)
    gs = grads_of(y, g1, [q1, ctx3])
    bag.put("attn_cross1", q=q1, ctx=ctx3, y=y, g=g1, gq=gs[0], gctx=gs[1])
    y = attn(q1, context=torch.zeros(2, 0, 32))
    bag.put("attn_cross_empty", y=y)

    # a9: GEGLU feed-forward (zorro_utils.py:115-128); inner = int(32*4*2/3) = 85 (odd on purpose)
    seeded()
    ff = ref.zu.FeedForward(dim=32, mult=4); rand_init_(ff, gen)
    x = R(2, 7, 32).requires_grad_(); g = R(2, 7, 32)
    y = ff(x)
    gs = grads_of(y, g, [x] + list(ff.parameters()))
    bag.put("feedforward", x=x, y=y, g=g, gx=gs[0],
            **{"w." + k: v for k, v in ff.state_dict().items()},
            **{"gw." + n: gi for (n, _), gi in zip(ff.named_parameters(), gs[1:])})

    # Mlp (zorro_utils.py:131-148)
    seeded()
    mlp = ref.zu.Mlp(in_features=32, hidden_features=128); rand_init_(mlp, gen)
    x = R(2, 4, 32).requires_grad_(); g = R(2, 4, 32)
    y = mlp(x)
    gs = grads_of(y, g, [x] + list(mlp.parameters()))
    bag.put("mlp", x=x, y=y, g=g, gx=gs[0],
            **{"w." + k: v for k, v in mlp.state_dict().items()},
            **{"gw." + n: gi for (n, _), gi in zip(mlp.named_parameters(), gs[1:])})

    # a10: Block (zorro_utils.py:227-240)
    seeded()
    blk = ref.zu.Block(dim=32, dim_head=32, heads=2, ff_mult=4, norm_layer=ref.zu.LayerNorm); rand_init_(blk, gen)
    x = R(2, 10, 32).requires_grad_(); g = R(2, 10, 32)
    y = blk(x, zmask)
    gs = grads_of(y, g, [x] + list(blk.parameters()))
    bag.put("block", x=x, mask=zmask, types=types, y=y, g=g, gx=gs[0],
            **{"w." + k: v for k, v in blk.state_dict().items()},
            **{"gw." + n: gi for (n, _), gi in zip(blk.named_parameters(), gs[1:])})

    # a12: Block_Fusion, canonical downstream copy (DSI-MM/zorro_utils.py:243-258)
    seeded()
    fus = ref.zu.Block_Fusion(dim=32, dim_head=32, heads=2, ff_mult=4, norm_layer=ref.zu.LayerNorm); rand_init_(fus, gen)
    x = R(2, 6, 4, 32).requires_grad_(); g = R(2, 6, 32)
    y = fus(x, None)
    gs = grads_of(y, g, [x] + list(fus.parameters()))
    bag.put("block_fusion", x=x, y=y, g=g, gx=gs[0],
            **{"w." + k: v for k, v in fus.state_dict().items()},
            **{"gw." + n: gi for (n, _), gi in zip(fus.named_parameters(), gs[1:])})

    # a1: PatchedInputAdapter (input_adapters.py:97-119), C=3, patch 8, 32x32 -> 16 patches, D=32
    seeded()
    pia = ref.ia.PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=8, dim_tokens=32, image_size=32)
    rand_init_(pia, gen)
    x = R(2, 3, 32, 32); g = R(2, 16, 32)
    y = pia(x)
    gs = grads_of(y, g, [pia.proj.weight, pia.proj.bias])
    bag.put("patched_input", x=x, y=y, g=g, gweight=gs[0], gbias=gs[1],
            **{"w." + k: v for k, v in pia.state_dict().items()})
    # a3: FusionInputAdapter (input_adapters.py:185-206)
    seeded()
    fia = ref.ia.FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=8, dim_tokens=32, image_size=32)
    x = R(2, 16, 32)
    bag.put("fusion_input", x=x, y=fia(x), **{"w." + k: v for k, v in fia.state_dict().items()})

    # a14: SpatialOutputAdapter (output_adapters_simple.py:146-188), 2 decoder blocks @64, 2 heads (dh 32)
    seeded()
    soa = ref.oa.SpatialOutputAdapter(num_channels=3, stride_level=1, patch_size_full=8, dim_tokens_enc=32,
                                      dim_tokens=64, depth=2, num_heads=2, image_size=32, task="s2",
                                      context_tasks=["s1", "s2", "dem"])
    rand_init_(soa, gen)
    enc = R(2, 16, 32).requires_grad_(); g = R(2, 3, 32, 32)
    info = {"image_size": (32, 32)}
    y = soa(enc, info, None, None)
    names = [n for n, p in soa.named_parameters() if p.requires_grad]
    params = [p for n, p in soa.named_parameters() if p.requires_grad]
    gs = grads_of(y, g, [enc] + params)
    bag.put("spatial_output", enc=enc, y=y, g=g, genc=gs[0],
            **{"w." + k: v for k, v in soa.state_dict().items()},
            **{"gw." + n: gi for n, gi in zip(names, gs[1:])})

    # a16: masked losses (criterion.py:85-115, 142-172)
    pred = R(3, 3, 32, 32).requires_grad_(); tgt = R(3, 3, 32, 32)
    mask = (torch.rand(3, 16, generator=gen) > 0.5).long()
    mask[2] = 0  # a sample with nothing masked -> 0/0 -> nan -> dropped by nanmean
    for name, cls in (("mse", ref.cr.MaskedMSELoss), ("l1", ref.cr.MaskedL1Loss)):
        fn = cls(patch_size=8, stride=1)
        l = fn(pred, tgt, mask=mask)
        (gp,) = grads_of(l, torch.tensor(1.0), [pred])
        bag.put("masked_" + name, pred=pred, tgt=tgt, mask=mask, loss=l, gpred=gp)
        l = fn(pred, tgt, mask=None)
        (gp,) = grads_of(l, torch.tensor(1.0), [pred])
        bag.put("masked_" + name + "_nomask", loss=l, gpred=gp)
        l0 = fn(pred, tgt, mask=torch.zeros(3, 16, dtype=torch.long))
        bag.put("masked_" + name + "_zeromask", loss=l0)  # integer tensor(0), criterion.py:101-102
        fnp = cls(patch_size=8, stride=1, norm_pix=True)
        l = fnp(pred, tgt, mask=mask)
        (gp,) = grads_of(l, torch.tensor(1.0), [pred])
        bag.put("masked_" + name + "_normpix", loss=l, gpred=gp)

    # a17: dino_loss_func (criterion.py:328-335)
    s = R(4, 32).requires_grad_(); t = R(4, 32).requires_grad_()
    l = ref.cr.dino_loss_func(s, t)
    gs_, gt_ = grads_of(l, torch.tensor(1.0), [s, t])
    bag.put("dino", student=s, teacher=t, loss=l, gstudent=gs_, gteacher=gt_)

    # a18: HardNegtive_loss (criterion.py:233-268); `.cuda()` at :242 made a no-op for the CPU run only
    seeded()
    hn = ref.cr.HardNegtive_loss()
    o1 = R(4, 32).requires_grad_(); o2 = R(4, 32).requires_grad_()
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        l = hn(o1, o2)
    finally:
        torch.Tensor.cuda = orig_cuda
    g1, g2 = grads_of(l, torch.tensor(1.0), [o1, o2])
    bag.put("hardneg", out_1=o1, out_2=o2, loss=l, g1=g1, g2=g2)

    # the rarely used siblings on the SAME inputs (no new draws from the global RNG: the fixtures after this one keep their seeds):
    # HardNegtive_loss(estimator='easy') (criterion.py:257-258), vicreg (:175-212), byol_loss_func (:319-326)
    hn_e = ref.cr.HardNegtive_loss(estimator='easy')
    a = o1.detach().clone().requires_grad_(); b = o2.detach().clone().requires_grad_()
    torch.Tensor.cuda = lambda self, *a_, **k: self
    try:
        le = hn_e(a, b)
    finally:
        torch.Tensor.cuda = orig_cuda
    ge1, ge2 = grads_of(le, torch.tensor(1.0), [a, b])
    bag.put("hardneg_easy", loss=le, g1=ge1, g2=ge2)
    a = o1.detach().clone().requires_grad_(); b = o2.detach().clone().requires_grad_()
    lv = ref.cr.vicreg(a, b)
    gv1, gv2 = grads_of(lv, torch.tensor(1.0), [a, b])
    bag.put("vicreg", loss=lv, g1=gv1, g2=gv2)
    a = o1.detach().clone().requires_grad_(); b = o2.detach().clone().requires_grad_()
    lb = ref.cr.byol_loss_func(a, b)
    (gb1,) = grads_of(lb, torch.tensor(1.0), [a])
    lb_full = ref.cr.byol_loss_func(a, b, simplified=False)
    (gb1f,) = grads_of(lb_full, torch.tensor(1.0), [a])
    bag.put("byol", loss=lb, gp=gb1, loss_full=lb_full, gp_full=gb1f)

    np.savez_compressed(os.path.join(OUT, "ops.npz"), **bag)
    print("ops.npz:", len(bag), "arrays")


# ----------------------------------------------------------------------------------------------
E2E_CFG = dict(dim_tokens=32, depth=2, dim_head=32, heads=2, image_size=64, patch_size=16,
               decoder_dim=32, decoder_depth=1, decoder_heads=1)
CHANNELS = (("s1", 1), ("s2", 3), ("dem", 1))


def harness_losses(ref, out, x, masks):
    """Transcription of the loss block of train_one_epoch (pretraining/pretrain_mmae.py:479-500),
    with NoWeightingStrategy (identity, utils/task_balancing.py:11-19)."""
    preds, _, pooled, ori, fus, s1_t, s2_t, dem_t = out
    fns = {"s1": ref.cr.MaskedMSELoss(patch_size=16, stride=1),
           "s2": ref.cr.MaskedMSELoss(patch_size=16, stride=1),
           "dem": ref.cr.MaskedL1Loss(patch_size=16, stride=1)}
    task_losses = {t: fns[t](preds[t].float(), x[t], mask=masks.get(t, None)) for t in preds}
    s1_f, s2_f, dsm_f, fus_f = [c.squeeze() for c in torch.chunk(pooled, 4, dim=1)]
    s1_t, s2_t, dem_t = s1_t.squeeze(), s2_t.squeeze(), dem_t.squeeze()
    d = ref.cr.dino_loss_func
    loss_contra = d(s1_t, s1_f) + d(s2_t, s2_f) + d(dem_t, dsm_f)
    loss = sum(task_losses.values()) + 0.3 * loss_contra
    return task_losses, loss_contra, loss


def gen_e2e(ref):
    torch.manual_seed(7)
    model = ref_loader.build_reference_model(ref, channels=CHANNELS, **E2E_CFG)
    gen = torch.Generator().manual_seed(99)
    rand_init_(model, gen, scale=0.3)
    with torch.no_grad():
        model.mask_embedding.add_(0.05 * torch.randn(model.mask_embedding.shape, generator=gen))
    model.train()
    B, P = 2, 16
    x = {d: torch.randn(B, c, 64, 64, generator=gen) for d, c in CHANNELS}
    bag = Bag()
    bag["config"] = np.array(json.dumps(dict(E2E_CFG, channels=CHANNELS, B=B)))
    for k, v in model.state_dict().items():
        bag["state/" + k] = npy(v)
    for d in x:
        bag["x/" + d] = npy(x[d])

    def mask_case(keep):
        m = {}
        for d, idx in keep.items():
            row = torch.ones(P, dtype=torch.long)
            row[torch.tensor(idx, dtype=torch.long)] = 0
            m[d] = row[None].repeat(B, 1)
        return m

    cases = {
        # uneven split, 24 of 48 kept
        "split": mask_case({"s1": [0, 3, 5, 6, 9, 10, 12, 15, 2, 7], "s2": [1, 2, 4, 8, 11, 13, 14, 0], "dem": [5, 6, 7, 9, 12, 3]}),
        # dem fully dropped (N_m = 0): fully-masked pool row + empty contrastive context
        "dropdem": mask_case({"s1": list(range(0, 16, 2)) + [1, 3], "s2": list(range(1, 16, 2)) + [0, 2, 4, 6], "dem": []}),
        # one modality only
        "onlys2": mask_case({"s1": [], "s2": list(range(16)), "dem": []}),
    }
    names = [n for n, p in model.named_parameters()]
    for cname, masks in cases.items():
        N = int(sum((m[0] == 0).sum() for m in masks.values()))
        captured = {}
        def hook_blk(mod, args):
            captured["zorro"] = args[1].clone()

        def hook_pool(mod, args, kwargs):
            if kwargs.get("attn_mask") is not None and "pool" not in captured:
                captured["pool"] = kwargs["attn_mask"].clone()

        h1 = model.blocks[0].register_forward_pre_hook(hook_blk)
        h2 = model.attn_pool.register_forward_pre_hook(hook_pool, with_kwargs=True)
        model.zero_grad()
        out = model(x, task_masks=masks, num_encoded_tokens=N)
        h1.remove(); h2.remove()
        task_losses, loss_contra, loss = harness_losses(ref, out, x, masks)
        loss.backward()
        preds, tm, pooled, ori, fus, s1_t, s2_t, dem_t = out
        pre = "case_%s/" % cname
        bag[pre + "N"] = np.array(N)
        for d in masks:
            bag[pre + "mask/" + d] = npy(masks[d])
            bag[pre + "pred/" + d] = npy(preds[d])
            bag[pre + "task_loss/" + d] = npy(task_losses[d])
        bag.put(pre[:-1], pooled=pooled, ori_tokens=ori, fusion_tokens=fus, ret_s1=s1_t, ret_s2=s2_t, ret_dem=dem_t,
                loss_contra=loss_contra, loss=loss, zorro_mask=captured["zorro"], pool_mask=captured["pool"])
        if cname != "onlys2":
            for n, p in model.named_parameters():
                if p.grad is not None:
                    bag[pre + "grad/" + n] = npy(p.grad)
    bag["param_names"] = np.array(json.dumps(names))
    np.savez_compressed(os.path.join(OUT, "e2e_tiny.npz"), **bag)
    print("e2e_tiny.npz:", len(bag), "arrays")


# ----------------------------------------------------------------------------------------------
def gen_masks(ref):
    """a4: generate_random_masks (multimae_crossattn.py:205-278) with the global-RNG draws recorded by
    replaying the same seed: Dirichlet(alphas).sample((1,)), M x rand(1,P), rand_like(mask_all)."""
    from torch.distributions.dirichlet import Dirichlet
    bag = Bag()
    torch.manual_seed(11)
    model = ref_loader.build_reference_model(ref, channels=CHANNELS, **E2E_CFG)
    cases = []
    for seed, (P, N, B) in enumerate([(16, 24, 2), (16, 24, 2), (16, 8, 1), (16, 40, 2), (64, 96, 3),
                                      (256, 384, 2), (256, 384, 2), (256, 128, 1), (256, 700, 1), (16, 48, 2)]):
        toks = {d: torch.zeros(B, P, 4) for d, _ in CHANNELS}
        torch.manual_seed(1000 + seed)
        tm, ids_keep, ids_restore = model.generate_random_masks(toks, N, alphas=1.0)
        torch.manual_seed(1000 + seed)
        dirichlet = Dirichlet(torch.Tensor([1.0] * 3)).sample((1,))
        noises = [torch.rand(1, P) for _ in range(3)]
        noise_all = torch.rand(1, 3 * P)
        pre = "mask%d" % seed
        bag.put(pre, P=np.array(P), N=np.array(N), B=np.array(B), dirichlet=dirichlet, noise=torch.cat(noises, 0),
                noise_all=noise_all, ids_keep=ids_keep, ids_restore=ids_restore,
                **{"task_mask." + d: tm[d] for d in tm})
        cases.append(pre)
    bag["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **bag)
    print("masks.npz:", len(bag), "arrays")


# ----------------------------------------------------------------------------------------------
DS_CFG = dict(dim_tokens=32, depth=4, dim_head=32, heads=2, image_size=64, patch_size=16)


def gen_downstream():
    """ViTBaseline.forward_features / forward (multimae_big_imcomplete.py:534-680): per-forward modality
    subset (absent modalities get NO slot in the modality attention), feature taps after layers `flags`, norm + necks."""
    ds = ref_loader.load_downstream()
    torch.manual_seed(17)
    model = ref_loader.build_downstream_backbone(ds, channels=CHANNELS, **DS_CFG)
    gen = torch.Generator().manual_seed(5)
    rand_init_(model, gen, scale=0.3)
    with torch.no_grad():
        model.mask_embedding.add_(0.05 * torch.randn(model.mask_embedding.shape, generator=gen))
    B, P = 2, 16
    x = {d: torch.randn(B, c, 64, 64, generator=gen) for d, c in CHANNELS}
    bag = Bag()
    bag["config"] = np.array(json.dumps(dict(DS_CFG, channels=CHANNELS, B=B, flags=list(model.flags))))
    for k, v in model.state_dict().items():
        bag["state/" + k] = npy(v)
    for d in x:
        bag["x/" + d] = npy(x[d])
    # eval: all modalities, every token kept (mask_inputs False)
    model.eval()
    with torch.no_grad():
        outs, nh, nw = model.forward_features(x)
        feats = model(x)
    for i, o in enumerate(outs):
        bag["eval_all/tap%d" % i] = npy(o)
    for i, f in enumerate(feats):
        bag["eval_all/feat%d" % i] = npy(f)
    # training-style subsets with explicit 90 % masks (int(M'*P*0.9) kept, :579); the subset itself is what the
    # reference draws with random.sample (:542-544) -- pinned here by seeding `random`
    import random
    model.train()
    for seed in (1, 2, 5):
        random.seed(seed)
        n = random.randint(1, 3); subset = random.sample(model.in_domains, n)
        present = [d for d in model.in_domains if d in subset]           # OrderedDict order of x (:560-564)
        N = int(len(present) * P * 0.9)
        masks, left = {}, N
        for j, d in enumerate(present):
            k = left if j == len(present) - 1 else min(P, max(0, N // len(present) + (j % 2) * 2 - 1))
            k = min(k, P); left -= k
            row = torch.ones(P, dtype=torch.long); row[torch.randperm(P, generator=gen)[:k]] = 0
            masks[d] = row[None].repeat(B, 1)
        assert left == 0
        random.seed(seed)
        model.zero_grad()
        outs, nh, nw = model.forward_features(x, task_masks=masks)
        assert list(model.incomplete_domains) == subset
        loss = sum((o * o).mean() for o in outs)
        loss.backward()
        pre = "train_seed%d/" % seed
        bag[pre + "present"] = np.array(json.dumps(present))
        bag[pre + "N"] = np.array(N)
        for d in present:
            bag[pre + "mask/" + d] = npy(masks[d])
        for i, o in enumerate(outs):
            bag[pre + "tap%d" % i] = npy(o)
        bag[pre + "loss"] = npy(loss)
        for n_, p_ in model.named_parameters():
            if p_.grad is not None and ("blocks.0." in n_ or "fus_blocks.3." in n_ or n_ in ("mask_embedding", "fusion_tokens")):
                bag[pre + "grad/" + n_] = npy(p_.grad)
    np.savez_compressed(os.path.join(OUT, "downstream.npz"), **bag)
    print("downstream.npz:", len(bag), "arrays")

# ----------------------------------------------------------------------------------------------
def gen_aux():
    """Pieces around the 3-modality hot path that rounds >= 2 build: the 4-modality driver's class-map modality
    (SemSegInputAdapter, MaskedCrossEntropyLoss; pretrain_mmae_my.py:67-74), the step shell's schedule and balancer
    (utils/native_scaler.py:65-82, utils/task_balancing.py:21-44), the (commented-out) DINOLoss class, and the
    normalisation constants/functions of the input staging row f3 (utils/multimodal_dfc2023.py:18-49 -- the functions that
    need neither rasterio nor cv2; load_* and the cv2 resize stay UNPINNED)."""
    aux = ref_loader.load_aux()
    bag = Bag()
    gen = torch.Generator().manual_seed(777)
    R = lambda *s: torch.randn(*s, generator=gen)

    # SemSegInputAdapter (input_adapters.py:209-328): 5 classes, emb 8, patch 8, 32x32 -> 16 patches, D=32
    torch.manual_seed(5001)
    ssa = aux.ia.SemSegInputAdapter(num_classes=5, stride_level=1, patch_size_full=8, dim_tokens=32, image_size=32,
                                    dim_class_emb=8, interpolate_class_emb=False)
    rand_init_(ssa, gen)
    x = torch.randint(0, 5, (2, 32, 32), generator=gen)
    g = R(2, 16, 32)
    y = ssa(x)
    gs = grads_of(y, g, [ssa.class_emb.weight, ssa.proj.weight, ssa.proj.bias])
    bag.put("semseg_input", x=x, y=y, g=g, gclass_emb=gs[0], gweight=gs[1], gbias=gs[2],
            **{"w." + k: v for k, v in ssa.state_dict().items()})

    # MaskedCrossEntropyLoss (criterion.py:24-58): mask with an empty row, no mask, zero mask, label smoothing
    pred = R(3, 5, 32, 32).requires_grad_(); tgt = torch.randint(0, 5, (3, 32, 32), generator=gen)
    mask = (torch.rand(3, 16, generator=gen) > 0.5).long()
    mask[2] = 0
    for name, ls in (("ce", 0.0), ("ce_smooth", 0.1)):
        fn = aux.cr.MaskedCrossEntropyLoss(patch_size=8, stride=1, label_smoothing=ls)
        l = fn(pred, tgt, mask=mask)
        (gp,) = grads_of(l, torch.tensor(1.0), [pred])
        bag.put("masked_" + name, pred=pred, tgt=tgt, mask=mask, loss=l, gpred=gp)
        l = fn(pred, tgt, mask=None)
        (gp,) = grads_of(l, torch.tensor(1.0), [pred])
        bag.put("masked_" + name + "_nomask", loss=l, gpred=gp)
        bag.put("masked_" + name + "_zeromask", loss=fn(pred, tgt, mask=torch.zeros(3, 16, dtype=torch.long)))

    # DINOLoss class (criterion.py:270-317).  Dead code in the reference: its update_center (:309-314) raises TypeError on
    # the (B, D) tensor its own forward hands it (`torch.cat(tensor)`), so the class cannot complete one call; only the loss
    # expression (:281-303) is pinned here, with update_center made a no-op on this instance.
    dl = aux.cr.DINOLoss(out_dim=32)
    dl.update_center = lambda t: None
    s1, t1 = R(4, 32).requires_grad_(), R(4, 32)
    l1 = dl(s1, t1)
    (g1,) = grads_of(l1, torch.tensor(1.0), [s1])
    bag.put("dino_class", s1=s1, t1=t1, loss1=l1, gs1=g1)

    # UncertaintyWeightingStrategy (task_balancing.py:21-44) incl. a dropped task (loss exactly 0)
    uw = aux.tb.UncertaintyWeightingStrategy(tasks=["s1", "s2", "dem"])
    with torch.no_grad():
        uw.log_vars.copy_(torch.tensor([0.3, -0.2, 0.1]))
    losses = {"s1": torch.tensor(0.7, requires_grad=True), "s2": torch.tensor(0.0, requires_grad=True),
              "dem": torch.tensor(1.3, requires_grad=True)}
    w = uw(losses)
    tot = sum(w.values())
    gl = grads_of(tot, torch.tensor(1.0), [uw.log_vars] + list(losses.values()))
    bag.put("uncertainty", log_vars=uw.log_vars, losses=torch.stack(list(losses.values())),
            weighted=torch.stack(list(w.values())), glog_vars=gl[0], glosses=torch.stack(gl[1:]))

    # cosine_scheduler (native_scaler.py:65-82)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        bag.put("cosine", a=aux.ns.cosine_scheduler(1e-3, 1e-5, 4, 7, warmup_epochs=1),
                b=aux.ns.cosine_scheduler(2e-4, 0.0, 3, 5, warmup_epochs=1, warmup_steps=3, start_warmup_value=1e-6),
                c=aux.ns.cosine_scheduler(1.0, 0.1, 2, 4))

    # f3 constants and the rasterio/cv2-free normalisation functions (multimodal_dfc2023.py:18-49)
    d = aux.dfc
    rgb = np.random.default_rng(3).uniform(0, 255, size=(3, 6, 5))
    sar = np.random.default_rng(4).uniform(-25, 0, size=(1, 6, 5))
    dem = np.random.default_rng(5).normal(5, 7, size=(1, 6, 5))
    bag.put("staging", rgb_mean=d.rgb_MEAN, rgb_std=d.rgb_STD, sar_mean=d.sar_MEAN, sar_std=d.sar_STD,
            dem_mean=d.dem_MEAN, dem_std=d.dem_STD, rgb_in=rgb, rgb_out=d.normalize_rgb(rgb.copy()),
            sar_in=sar, sar_out=d.normalize_sar(sar.copy()), dem_in=dem, dem_out=d.normalize_dem(dem.copy()),
            minmax_out=d.normalization(dem.copy()))
    # (appended last so that the draws of the cases above are unchanged) the SemSeg adapter with interpolate_class_emb=True (:288-294) and an embedding padding index (zero-gradient row)
    torch.manual_seed(5002)
    ssi = aux.ia.SemSegInputAdapter(num_classes=5, stride_level=1, patch_size_full=8, dim_tokens=32, image_size=32,
                                    dim_class_emb=8, interpolate_class_emb=True, emb_padding_idx=5)
    rand_init_(ssi, gen)
    xi = torch.randint(0, 6, (2, 32, 32), generator=gen)
    gi = R(2, 16, 32)
    yi = ssi(xi)
    gsi = grads_of(yi, gi, [ssi.class_emb.weight, ssi.proj[1].weight, ssi.proj[1].bias])
    bag.put("semseg_input_interp", x=xi, y=yi, g=gi, gclass_emb=gsi[0], gweight=gsi[1], gbias=gsi[2],
            **{"w." + k: v for k, v in ssi.state_dict().items()})

    np.savez_compressed(os.path.join(OUT, "aux.npz"), **bag)
    print("aux.npz:", len(bag), "arrays")


# ----------------------------------------------------------------------------------------------
def gen_quad():
    """The reference's own 4-modality model (multimae_quadruplet.MultiMAE, no fusion blocks) under the loss of its driver
    (pretrain_mmae_my.py:495-515: masked MSE / L1 / cross-entropy task losses, NoWeightingStrategy, no contrastive term):
    forward 5-tuple, losses and every parameter gradient on explicit masks, plus one generate_random_masks call with its
    recorded draws (the 4-modality version of masks.npz)."""
    q = ref_loader.load_quad()
    cfg = dict(dim_tokens=32, depth=2, dim_head=32, heads=2, image_size=64, patch_size=16, num_classes=9, dim_class_emb=8,
               decoder_dim=32, decoder_depth=1, decoder_heads=1)
    seeded()
    model = ref_loader.build_reference_quad_model(q, **cfg)
    gen = torch.Generator().manual_seed(1234)
    rand_init_(model, gen, scale=0.3)
    model.train()
    B, P = 2, 16
    x = {d: torch.randn(B, c, 64, 64, generator=gen) for d, c in ref_loader.QUAD_CHANNELS[:3]}
    x["dnw"] = torch.randint(0, 9, (B, 64, 64), generator=gen)
    bag = Bag()
    bag["config"] = np.array(json.dumps(dict(cfg, B=B)))
    for k, v in model.state_dict().items():
        bag["state/" + k] = npy(v)
    for d in x:
        bag["x/" + d] = npy(x[d])
    fns = {"s1": q.cr.MaskedMSELoss(patch_size=16, stride=1), "s2": q.cr.MaskedMSELoss(patch_size=16, stride=1),
           "dem": q.cr.MaskedL1Loss(patch_size=16, stride=1),
           "dnw": q.cr.MaskedCrossEntropyLoss(patch_size=16, stride=1, label_smoothing=0.0)}

    def mask_case(keep):
        m = {}
        for d, idx in keep.items():
            row = torch.ones(P, dtype=torch.long)
            row[torch.tensor(idx, dtype=torch.long)] = 0
            m[d] = row[None].repeat(B, 1)
        return m
    cases = {
        "split": mask_case({"s1": [0, 3, 5, 6, 9, 10, 12], "s2": [1, 2, 4, 8, 11], "dem": [5, 6, 7, 9, 12, 3],
                            "dnw": [0, 15, 14, 2, 7, 8]}),
        "dropdnw": mask_case({"s1": list(range(0, 16, 2)), "s2": [1, 3, 5], "dem": list(range(4, 16)), "dnw": []}),
    }
    for cname, masks in cases.items():
        N = int(sum((m[0] == 0).sum() for m in masks.values()))
        model.zero_grad()
        preds, tm, pooled, ori, fus = model(x, task_masks=masks, num_encoded_tokens=N)
        task_losses = {t: fns[t](preds[t].float(), x[t], mask=masks.get(t, None)) for t in preds}
        loss = sum(task_losses.values())
        loss.backward()
        pre = "case_%s/" % cname
        bag[pre + "N"] = np.array(N)
        for d in masks:
            bag[pre + "mask/" + d] = npy(masks[d])
            bag[pre + "pred/" + d] = npy(preds[d])
            bag[pre + "task_loss/" + d] = npy(task_losses[d])
        bag.put(pre[:-1], pooled=pooled, ori_tokens=ori, fusion_tokens=fus, loss=loss)
        for n, p in model.named_parameters():
            if p.grad is not None:
                bag[pre + "grad/" + n] = npy(p.grad)
    # generate_random_masks (multimae_quadruplet.py:177-250) with four modalities: replay the global-RNG draws
    from torch.distributions.dirichlet import Dirichlet
    toks = {d: torch.zeros(3, P, 4) for d in ("s1", "s2", "dem", "dnw")}
    for i, (n_enc, alphas) in enumerate(((20, 1.0), (7, [0.3, 2.0, 1.0, 0.5]))):
        torch.manual_seed(8800 + i)
        tm, ids_keep, ids_restore = model.generate_random_masks(toks, n_enc, alphas=alphas, sample_tasks_uniformly=False)
        torch.manual_seed(8800 + i)
        al = [alphas] * 4 if isinstance(alphas, float) else alphas
        dirichlet = Dirichlet(torch.Tensor(al)).sample((1,))
        noise = torch.stack([torch.rand(1, P) for _ in range(4)])
        noise_all = torch.rand(1, 4 * P)
        bag.put("masks_%d" % i, n_enc=np.array(n_enc), alphas=np.array(al), dirichlet=dirichlet, noise=noise,
                noise_all=noise_all, ids_keep=ids_keep, ids_restore=ids_restore, **{"mask_" + d: tm[d] for d in tm})
    np.savez_compressed(os.path.join(OUT, "quad_tiny.npz"), **bag)
    print("quad_tiny.npz:", len(bag), "arrays")


# ----------------------------------------------------------------------------------------------
# Stochastic depth (drop_path_rate > 0, pretrain_mmae.py:108,245; DSI-MM/zorro_utils.py:69-96,233,238-239; decoder Blocks
# MM/multimae_utils.py:224,230-231): one TRAINING forward + backward of the reference with a seeded global generator.  With explicit
# task_masks the DropPath modules are the only consumers of that generator (torch.rand((B, 1, 1)) per application, attention branch
# then feed-forward branch of every Block whose rate is > 0, encoder first, then the decoders in domain order), so the draws the
# reference used are reproduced by re-seeding and drawing the same shapes in the same order; they go into the fixture.
DROP_CFG = dict(dim_tokens=32, depth=3, dim_head=32, heads=2, image_size=64, patch_size=16,
                decoder_dim=32, decoder_depth=2, decoder_heads=1, drop_path_rate=0.5, decoder_drop_path_rate=0.4)


def gen_droppath(ref):
    torch.manual_seed(11)
    model = ref_loader.build_reference_model(ref, channels=CHANNELS, **DROP_CFG)
    gen = torch.Generator().manual_seed(123)
    rand_init_(model, gen, scale=0.3)
    with torch.no_grad():
        model.mask_embedding.add_(0.05 * torch.randn(model.mask_embedding.shape, generator=gen))
    model.train()
    B, P = 6, 16
    x = {d: torch.randn(B, c, 64, 64, generator=gen) for d, c in CHANNELS}
    keep = {"s1": [0, 3, 5, 6, 9, 10, 12, 15, 2, 7], "s2": [1, 2, 4, 8, 11, 13, 14, 0], "dem": [5, 6, 7, 9, 12, 3]}
    masks = {}
    for d, idx in keep.items():
        row = torch.ones(P, dtype=torch.long); row[torch.tensor(idx, dtype=torch.long)] = 0
        masks[d] = row[None].repeat(B, 1)
    N = int(sum((m[0] == 0).sum() for m in masks.values()))
    SEED = 2024
    # the draws, in consumption order: encoder layers with rate > 0 (linspace(0, rate, depth): layer 0 has none), two each; then per
    # decoder (s1, s2, dem) its Blocks with rate > 0, two each
    enc_rates = [v.item() for v in torch.linspace(0, DROP_CFG["drop_path_rate"], DROP_CFG["depth"])]
    dec_rates = [v.item() for v in torch.linspace(0, DROP_CFG["decoder_drop_path_rate"], DROP_CFG["decoder_depth"])]
    n_draws = 2 * sum(1 for r in enc_rates if r > 0) + 3 * 2 * sum(1 for r in dec_rates if r > 0)
    torch.manual_seed(SEED)
    draws = torch.stack([torch.rand((B, 1, 1)).reshape(B) for _ in range(n_draws)])
    state_after = torch.get_rng_state()
    torch.manual_seed(SEED)
    model.zero_grad()
    out = model(x, task_masks=masks, num_encoded_tokens=N)
    assert torch.equal(torch.get_rng_state(), state_after), "the forward consumed a different number of draws"
    task_losses, loss_contra, loss = harness_losses(ref, out, x, masks)
    loss.backward()
    # both outcomes must occur for the fixture to pin anything: some (sample, branch) pairs dropped, some kept
    rates = [r for r in enc_rates if r > 0 for _ in range(2)] + [r for _ in range(3) for r in dec_rates if r > 0 for _ in range(2)]
    kept = torch.stack([torch.floor((1 - r) + u) for r, u in zip(rates, draws)])
    assert 0 < int(kept.sum()) < kept.numel()
    preds, tm, pooled, ori, fus, s1_t, s2_t, dem_t = out
    bag = Bag()
    bag["config"] = np.array(json.dumps(dict(DROP_CFG, channels=CHANNELS, B=B, N=N)))
    for k, v in model.state_dict().items():
        bag["state/" + k] = npy(v)
    for d in x:
        bag["x/" + d] = npy(x[d]); bag["mask/" + d] = npy(masks[d]); bag["pred/" + d] = npy(preds[d])
        bag["task_loss/" + d] = npy(task_losses[d])
    bag["draws"] = npy(draws)
    bag.put("out", pooled=pooled, ori_tokens=ori, fusion_tokens=fus, ret_s1=s1_t, ret_s2=s2_t, ret_dem=dem_t,
            loss_contra=loss_contra, loss=loss)
    for n, p in model.named_parameters():
        if p.grad is not None:
            bag["grad/" + n] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "droppath.npz"), **bag)
    print("droppath.npz:", len(bag), "arrays;", int(kept.numel() - kept.sum()), "of", kept.numel(), "(sample, branch) pairs dropped")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)          # single-threaded reductions: the fixtures regenerate byte for byte
    ref = ref_loader.load()
    gen_ops(ref)
    gen_e2e(ref)
    gen_masks(ref)
    gen_downstream()
    gen_aux()
    gen_quad()
    gen_droppath(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
