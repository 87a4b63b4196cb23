"""Parity at BASELINE.json's full model size: ViT-B (D768 / L12 / 8 heads of 64), 3 modalities, 256x256 tiles, N = 384 of
768 tokens kept, decoders 256/2/8.

  * B = 2, shared masks: the whole step (forward, every loss, every parameter gradient) against the CPU oracle.
  * B = 64 (configs[1]'s per-GPU batch), per-sample Dirichlet masks drawn by the product's own mask generator: individual
    samples against the oracle run on that sample alone (sample independence is the size-independent property here),
    bookkeeping invariants of the draw, and the masked losses re-derived from the predictions.
"""
import pytest
import torch

from oracle import mmae_oracle as O
from tests.test_cabi_symbols import build_model
from tests.test_gpu_kernels import DEV, close

pytestmark = pytest.mark.gpu

VITB = dict(dim_tokens=768, depth=12, dim_head=64, heads=8, image_size=256, patch_size=16, decoder_dim=256,
            decoder_depth=2, decoder_heads=8)
CHANNELS = (("s1", 1), ("s2", 3), ("dem", 1))


def _vitb(seed):
    torch.manual_seed(seed)
    model = build_model(VITB, CHANNELS)
    with torch.no_grad():                       # move gammas / mask embedding off their init so they matter
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith("gamma") or "norm" in n and n.endswith("weight")):
                p.add_(0.1 * torch.randn_like(p))
        model.mask_embedding.add_(0.05 * torch.randn_like(model.mask_embedding))
    return model


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-owngemm"])
def test_vitb_full_step_vs_oracle(mode):
    """Every output, loss and parameter gradient of one ViT-B step.  fp32: 1e-3 (gradients 2e-3).  bf16: 1e-2, or -- where a
    tensor exceeds it -- within 1.5x the error the REFERENCE arithmetic itself shows in bf16 on the same weights and inputs
    (the oracle under CPU bf16 autocast; tests/parity.py).  No hand-picked relaxation.
    bf16-owngemm: the composition the bench runs -- every supported projection on the own GEMM (gemm8p_kernel<0/1>: forward, input
    gradients through the engine's transposed bf16 shadows, FF1 + GEGLU epilogue with h consumed by geglu_bwd), the step through
    engine.FlatAdamW with gradients read from its flat buffer; the context asserts the kernels were launched."""
    from tests import parity
    own = mode.endswith("-owngemm")
    mode = mode.split("-")[0]
    model = _vitb(21)
    B, P, N = 2, 256, 384
    x = {d: torch.randn(B, c, 256, 256) for d, c in CHANNELS}
    keep = {"s1": 170, "s2": 41, "dem": 173}            # ragged 64-row tiles in every segment
    masks = {}
    for d, k in keep.items():
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    autocast = mode == "bf16"
    # (seeded construction: the library and own-GEMM cases see identical weights / inputs / masks and share the oracle's two runs)
    ref, anchor = parity.cached_oracle(("vitb_full_step", mode), state, x, masks, N, VITB["heads"], VITB["decoder_heads"], anchor=autocast)
    model.to(DEV).train()
    xd = {k: v.to(DEV) for k, v in x.items()}; md = {k: v.to(DEV) for k, v in masks.items()}
    if own:
        with parity.own_gemm_engaged():
            got = parity.native_step_flat(model, xd, md, N, autocast, engine=True)
    else:
        got = parity.native_step_flat(model, xd, md, N, autocast)
    parity.compare(got, ref, anchor, tol=1e-2 if autocast else 1e-3)


def test_vitb_batch64_bf16_step_at_product_dispatch_vs_oracle():
    """B = 64 (batch-shared masks, the headline bench's mask mode), bf16, the step through engine.FlatAdamW at the PRODUCT's own-GEMM
    threshold (ops._OWN_GEMM_MIN_TILES untouched; since round 6 from 256 output tiles on where N >= 512: at 40 960 rows every projection of the modality rows
    with N >= 512 runs on gemm8p_kernel, the fusion rows' narrow ones and the decoders on the library): every output, loss and parameter gradient against the oracle evaluated in
    chunks of 8 samples (tests/parity.chunked_oracle -- exact: every loss term is a mean over samples), anchored on the oracle's
    own bf16 arithmetic (the same chunked evaluation under CPU bf16 autocast)."""
    from tests import parity
    model = _vitb(23)
    B, P, N = 64, 256, 384
    x = {d: torch.randn(B, c, 256, 256) for d, c in CHANNELS}
    keep = {"s1": 97, "s2": 211, "dem": 76}
    masks = {}
    for d, k in keep.items():
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    xd = {k: v.to(DEV) for k, v in x.items()}; md = {k: v.to(DEV) for k, v in masks.items()}
    with parity.own_gemm_engaged(min_tiles=None):
        got = parity.native_step_flat(model, xd, md, N, autocast=True, engine=True)
    ref = parity.chunked_oracle(state, x, masks, N, VITB["heads"], VITB["decoder_heads"])
    anchor = parity.chunked_oracle(state, x, masks, N, VITB["heads"], VITB["decoder_heads"], bf16=True)
    parity.compare(got, ref, anchor, tol=1e-2)


def test_vitb_batch64_samples_are_independent_and_match_oracle():
    model = _vitb(22)
    B, P, N = 64, 256, 384
    x = {d: torch.randn(B, c, 256, 256) for d, c in CHANNELS}
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    model.per_sample_masks = True
    model.fuse_unpatchify_loss = False
    xd = {k: v.to(DEV) for k, v in x.items()}
    torch.manual_seed(5)
    with torch.no_grad():
        out = model(xd, num_encoded_tokens=N, alphas=1.0)           # fp32 mode, masks drawn on the device
    preds, tm, pooled, ori, fus, r1, r2, r3 = out
    keep = torch.stack([(tm[d] == 0).sum(1) for d in O.DOMAINS], 1).cpu()
    assert torch.equal(keep.sum(1), torch.full((B,), N)), "exactly N kept tokens per sample"
    assert len({tuple(r.tolist()) for r in keep}) > B // 2, "Dirichlet splits differ per sample"
    for d in O.DOMAINS:
        assert set(tm[d].unique().tolist()) <= {0, 1}
    # individual samples against the oracle on that sample alone, with the mask the product drew for it
    for b in (0, 29, 63):
        xb = {k: v[b:b + 1] for k, v in x.items()}
        mb = {d: tm[d][b:b + 1].cpu() for d in O.DOMAINS}
        ref = O.multimae_forward(state, xb, mb, N, VITB["heads"], VITB["decoder_heads"])
        for d in O.DOMAINS:
            close(preds[d][b:b + 1], ref[0][d], 1e-3, "pred %s sample %d" % (d, b))
        close(pooled[b:b + 1], ref[2], 1e-3, "pooled %d" % b)
        close(ori[b:b + 1], ref[3], 1e-3, "ori %d" % b)
        close(fus[b:b + 1], ref[4], 1e-3, "fusion %d" % b)
    # the same samples inside a different batch (other neighbours, other row offsets in the packed space): same result
    sel = [63, 7, 29, 0]
    with torch.no_grad():
        out2 = model({k: v[sel] for k, v in xd.items()}, task_masks={d: tm[d][sel] for d in O.DOMAINS}, num_encoded_tokens=N)
    for d in O.DOMAINS:
        close(out2[0][d], preds[d][sel], 2e-5, "re-batched pred " + d)
    close(out2[4], fus[sel], 2e-5, "re-batched fusion")
    # masked losses re-derived from the predictions (checksum of the loss kernels at full size)
    from incomplete_multimodal_fusion_amd.multimae.criterion import MaskedL1Loss, MaskedMSELoss
    mse, l1 = MaskedMSELoss(patch_size=16, stride=1), MaskedL1Loss(patch_size=16, stride=1)
    for d, c in CHANNELS:
        m = tm[d].view(B, 1, 16, 16).float().repeat_interleave(16, 2).repeat_interleave(16, 3)     # nearest resize
        diff = preds[d].float() - xd[d]
        valid = m.flatten(1).sum(1) > 0                                                            # nanmean over samples

        def per(e):
            return ((e.mean(1, keepdim=True) * m).flatten(1).sum(1) / m.flatten(1).sum(1).clamp_min(1.0))[valid].mean()
        close(mse(preds[d].float(), xd[d], mask=tm[d]), per(diff * diff), 1e-5, "mse " + d)
        close(l1(preds[d].float(), xd[d], mask=tm[d]), per(diff.abs()), 1e-5, "l1 " + d)
