"""BASELINE.json's configurations on the GPU against the oracle (round-1 verdict, weak #2): every output, loss and gradient.

  C1  the real Tiny preset (D192 / L12 / 3 heads), 64x64 tiles, B = 2, dem dropped (2 live modalities)
  C2  "Small" (D384 / L12 / 8 heads), 128x128 tiles, dem fully masked (2 live modalities), B = 8, fp32 and bf16
  C3  ViT-B with per-sample masks drawn by the product under sample_tasks_uniformly (modality dropout): samples WITH a
      dropped modality are picked and compared with the oracle run on that sample alone
  C5' ViT-L (D1024 / L24 / 8 heads), 3 modalities, B = 2 (the widest / deepest preset; the 4-modality variant of C5 is in
      tests/test_gpu_quad.py)
Tolerances and the bf16 anchor: tests/parity.py."""
import pytest
import torch

from oracle import mmae_oracle as O
from tests import parity
from tests.test_gpu_kernels import DEV, close

pytestmark = pytest.mark.gpu


def _model(name, size, seed, **kw):
    from incomplete_multimodal_fusion_amd.pretrain import get_model
    torch.manual_seed(seed)
    m = get_model(name, input_size=size, **kw)
    with torch.no_grad():                       # move gammas / mask embedding off their init so they matter
        for n, p in m.named_parameters():
            if p.requires_grad and (n.endswith("gamma") or "norm" in n and n.endswith("weight")):
                p.add_(0.1 * torch.randn_like(p))
        m.mask_embedding.add_(0.05 * torch.randn_like(m.mask_embedding))
    return m


def _masks(P, B, keep):
    masks = {}
    for d, k in keep.items():
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1)
    return masks


def _full_step(model, heads, B, size, keep, mode, key=None):
    P = (size // 16) ** 2
    x = {"s1": torch.randn(B, 1, size, size), "s2": torch.randn(B, 3, size, size), "dem": torch.randn(B, 1, size, size)}
    masks = _masks(P, B, keep)
    N = sum(keep.values())
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    own = mode.endswith("-owngemm")                 # the bench's composition: own GEMM on every supported shape + the flat engine (tests/parity.py)
    autocast = mode.split("-")[0] == "bf16"
    ref, anchor = parity.cached_oracle((key, autocast), state, x, masks, N, heads, 8, anchor=autocast)    # shared by the library / own-GEMM cases
    model.to(DEV).train()
    xd, md = {k: v.to(DEV) for k, v in x.items()}, {k: v.to(DEV) for k, v in masks.items()}
    if own:
        with parity.own_gemm_engaged():
            got = parity.native_step_flat(model, xd, md, N, autocast, engine=True)
    else:
        got = parity.native_step_flat(model, xd, md, N, autocast)
    parity.compare(got, ref, anchor, tol=1e-2 if autocast else 1e-3)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_c1_tiny_preset_64px_two_live_modalities(mode):
    _full_step(_model("tiny", 64, 31), 3, 2, 64, {"s1": 9, "s2": 7, "dem": 0}, mode, key="c1")


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-owngemm"])
def test_c2_small_128px_two_live_modalities(mode):
    _full_step(_model("small", 128, 32), 8, 8, 128, {"s1": 37, "s2": 27, "dem": 0}, mode, key="c2")


@pytest.mark.parametrize("mode", ["bf16", "bf16-owngemm"])
def test_c5_vit_large_depth24_three_modalities(mode):
    _full_step(_model("large", 256, 33), 8, 2, 256, {"s1": 150, "s2": 61, "dem": 173}, mode, key="c5")


def test_c3_vitb_per_sample_modality_dropout_vs_oracle():
    """The product's own random path: per-sample rows, uniform task pre-sampling (multimae_crossattn.py:188-203, :224-228).
    Samples in which a modality received NO token are compared with the oracle run on that sample alone."""
    model = _model("base", 256, 34)
    B, P, N = 32, 256, 384
    x = {"s1": torch.randn(B, 1, 256, 256), "s2": torch.randn(B, 3, 256, 256), "dem": torch.randn(B, 1, 256, 256)}
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    model.per_sample_masks = True
    model.fuse_unpatchify_loss = False
    xd = {k: v.to(DEV) for k, v in x.items()}
    torch.manual_seed(6)
    with torch.no_grad():
        out = model(xd, num_encoded_tokens=N, alphas=1.0, sample_tasks_uniformly=True)
    preds, tm, pooled, ori, fus, *rets = out
    per_mod = torch.stack([(tm[d] == 0).sum(1) for d in O.DOMAINS], 1).cpu()
    assert torch.equal(per_mod.sum(1), torch.full((B,), N))
    dropped = [b for b in range(B) if (per_mod[b] == 0).any()]
    full = [b for b in range(B) if (per_mod[b] > 0).all()]
    assert dropped, "uniform task pre-sampling must drop a modality in some samples"
    one_left = [b for b in dropped if (per_mod[b] == 0).sum() == 2]
    picks = dropped[:2] + one_left[:1] + full[:1]
    for b in picks:
        xb = {k: v[b:b + 1] for k, v in x.items()}
        mb = {d: tm[d][b:b + 1].cpu() for d in O.DOMAINS}
        ref = O.multimae_forward(state, xb, mb, N, 8, 8)
        for d in O.DOMAINS:
            close(preds[d][b:b + 1], ref[0][d], 1e-3, "pred %s sample %d %s" % (d, b, per_mod[b].tolist()))
        close(pooled[b:b + 1], ref[2], 1e-3, "pooled %d" % b)
        close(ori[b:b + 1], ref[3], 1e-3, "ori %d" % b)
        close(fus[b:b + 1], ref[4], 1e-3, "fusion %d" % b)
        for r, want in zip(rets, ref[5:]):
            close(r[b:b + 1], want, 1e-3, "ret %d" % b)


@pytest.mark.parametrize("own", [False, True], ids=["library", "owngemm"])
def test_c3_vitb_per_sample_dropout_bf16_step_with_gradients_vs_oracle(own):
    """BASELINE config 3 at full width, bf16, INCLUDING the backward: ViT-B, 256 x 256, the model's own per-sample draw with
    uniform task pre-sampling (some samples lose a modality), B = 4.  Every output, loss and parameter gradient of the native
    batch against the per-sample assembly of the oracle (tests/parity.per_sample_oracle), anchored on the oracle's own bf16 run."""
    from tests import parity
    model = _model("base", 256, 35)
    B, P, N = 4, 256, 384
    x = {"s1": torch.randn(B, 1, 256, 256), "s2": torch.randn(B, 3, 256, 256), "dem": torch.randn(B, 1, 256, 256)}
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    model.per_sample_masks = True
    xd = {k: v.to(DEV) for k, v in x.items()}
    for seed in range(6, 40):                                      # a draw in which at least one sample drops a modality
        torch.manual_seed(seed)
        with torch.no_grad():
            tm = model(xd, num_encoded_tokens=N, alphas=1.0, sample_tasks_uniformly=True)[1]
        per_mod = torch.stack([(tm[d] == 0).sum(1) for d in O.DOMAINS], 1).cpu()
        if (per_mod == 0).any() and (per_mod > 0).all(1).any():
            break
    assert (per_mod == 0).any(), per_mod.tolist()
    masks = {d: tm[d].cpu() for d in O.DOMAINS}
    if own:                                                        # the bench's composition (tests/parity.own_gemm_engaged)
        with parity.own_gemm_engaged():
            got = parity.native_step_flat(model, xd, {d: tm[d] for d in O.DOMAINS}, N, autocast=True, engine=True)
    else:
        got = parity.native_step_flat(model, xd, {d: tm[d] for d in O.DOMAINS}, N, autocast=True)
    ref, anchor = parity.cached_oracle("c3_per_sample", state, x, masks, N, 8, 8, fn=parity.per_sample_oracle)
    parity.compare(got, ref, anchor, tol=1e-2)


ODD_FFI = dict(dim_tokens=512, depth=2, dim_head=64, heads=8, image_size=128, patch_size=16, decoder_dim=64, decoder_depth=1, decoder_heads=2)


@pytest.mark.parametrize("own", [False, True], ids=["library", "padded-owngemm"])
def test_odd_geglu_width_padded_route_vs_oracle(own, monkeypatch):
    """A token width whose GEGLU width fits none of the own GEMM's tiles (D = 512: ffi = int(512 * 8 / 3) = 1365 -> padded to 1536; ViT-L's
    2730 -> 2816 runs in the ViT-L tests): the whole step on the padded route (ops._FeedForwardGEGLU over engine.padded_ff: FF1 + GEGLU,
    FF2 and both input-gradient GEMMs on gemm8p_kernel, weight gradients at the exact width into the flat buffer) against the oracle --
    every output, loss and gradient; the library case shares the oracle's runs."""
    from incomplete_multimodal_fusion_amd import ops
    from tests.test_cabi_symbols import build_model
    torch.manual_seed(51)
    channels = (("s1", 1), ("s2", 3), ("dem", 1))
    model = build_model(ODD_FFI, channels)
    assert model.blocks[0].mlp[3].weight.shape[1] == 1365
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith("gamma") or "norm" in n and n.endswith("weight")):
                p.add_(0.1 * torch.randn_like(p))
        model.mask_embedding.add_(0.05 * torch.randn_like(model.mask_embedding))
    B, P = 6, 64
    x = {d: torch.randn(B, c, 128, 128) for d, c in channels}
    masks = _masks(P, B, {"s1": 37, "s2": 30, "dem": 29})
    N = 96
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref, anchor = parity.cached_oracle("odd_ffi", state, x, masks, N, 8, 2)
    model.to(DEV).train()
    xd, md = {k: v.to(DEV) for k, v in x.items()}, {k: v.to(DEV) for k, v in masks.items()}
    if own:
        monkeypatch.setattr(ops, "PAD_FF_MIN_TILES", 0)
        with parity.own_gemm_engaged():
            got = parity.native_step_flat(model, xd, md, N, True, engine=True)
        assert len(model._parity_engine._pad) == 4            # 2 layers x (Block, Block_Fusion) FeedForwards registered padded copies
    else:
        got = parity.native_step_flat(model, xd, md, N, True)
    parity.compare(got, ref, anchor, tol=1e-2)
