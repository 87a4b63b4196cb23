"""Build-container-only strengthening of the oracle pin: run the ACTUAL reference (oracle/ref_loader.py) next to the
oracle on a fresh seeded configuration that is not among the committed fixtures.  Skipped wherever /root/reference is
absent (e.g. on the GPU box, which only carries tests/golden)."""
import pytest
import torch

from oracle import mmae_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")


def test_random_config_forward_backward_matches_reference():
    ref = ref_loader.load()
    torch.manual_seed(21)
    cfg = dict(dim_tokens=64, depth=3, dim_head=32, heads=2, image_size=64, patch_size=16, decoder_dim=32,
               decoder_depth=2, decoder_heads=2)
    model = ref_loader.build_reference_model(ref, **cfg)
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_(0.1 * torch.randn_like(p))
    model.train()
    B, P = 2, 16
    x = {"s1": torch.randn(B, 1, 64, 64), "s2": torch.randn(B, 3, 64, 64), "dem": torch.randn(B, 1, 64, 64)}
    masks = {}
    for d, k in (("s1", 7), ("s2", 12), ("dem", 3)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)
    N = 22
    out = model(x, task_masks=masks, num_encoded_tokens=N)
    fns = {"s1": ref.cr.MaskedMSELoss(16, 1), "s2": ref.cr.MaskedMSELoss(16, 1), "dem": ref.cr.MaskedL1Loss(16, 1)}
    tl = {t: fns[t](out[0][t].float(), x[t], mask=masks[t]) for t in out[0]}
    feats = [c.squeeze() for c in torch.chunk(out[2], 4, dim=1)]
    lc = sum(ref.cr.dino_loss_func(r.squeeze(), f) for r, f in zip(out[5:], feats))
    loss = sum(tl.values()) + 0.3 * lc
    loss.backward()

    p = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("pos_emb") and not k.endswith("beta"))
         for k, v in model.state_dict().items()}
    out_o, (tl_o, lc_o, loss_o) = O.train_step_loss(p, x, masks, N, cfg["heads"], cfg["decoder_heads"], 16)
    loss_o.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-5 * max(1.0, abs(float(loss)))
    for d in O.DOMAINS:
        assert torch.allclose(out[0][d], out_o[0][d], atol=2e-5)
    for i in (2, 3, 4, 5, 6, 7):
        assert torch.allclose(out[i], out_o[i], atol=2e-5), i
    n = 0
    for name, prm in model.named_parameters():
        if prm.grad is None:
            assert p[name].grad is None
            continue
        assert torch.allclose(prm.grad, p[name].grad, atol=3e-5, rtol=1e-4), name
        n += 1
    assert n > 100


def test_random_mask_draws_match_reference_rng_order():
    """Same seed -> the oracle's injected-draw bookkeeping reproduces generate_random_masks of the reference."""
    ref = ref_loader.load()
    from torch.distributions.dirichlet import Dirichlet
    model = ref_loader.build_reference_model(ref, dim_tokens=32, depth=1, dim_head=32, heads=1, image_size=128,
                                             decoder_dim=32, decoder_depth=1, decoder_heads=1)
    for seed in range(20):
        P, N, B = 64, 96, 2
        toks = {d: torch.zeros(B, P, 4) for d in O.DOMAINS}
        torch.manual_seed(seed)
        tm, ids_keep, ids_restore = model.generate_random_masks(toks, N, alphas=1.0)
        torch.manual_seed(seed)
        d = Dirichlet(torch.Tensor([1.0] * 3)).sample((1,))
        noise = torch.stack([torch.rand(1, P) for _ in range(3)], dim=1)
        na = torch.rand(1, 3 * P)
        mask_all, k, r = O.masks_from_draws(d, noise, na, N)
        assert torch.equal(mask_all.repeat(B, 1), torch.cat([tm[x] for x in O.DOMAINS], 1))
        assert torch.equal(k.repeat(B, 1), ids_keep) and torch.equal(r.repeat(B, 1), ids_restore)
