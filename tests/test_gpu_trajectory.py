"""A multi-step TRAINING TRAJECTORY of the native path against the oracle: k steps of PretrainStep on the flat
AdamW engine -- forward, losses, backward, device-side AdamW, bf16 shadows refreshed, the next step reading them -- against k steps of
oracle.train_step_loss + torch.optim.AdamW on the CPU (the reference's step: pretraining/pretrain_mmae.py:466-517 with the optimizer of
utils/optim_factory.py:136-150: AdamW beta (0.9, 0.95), weight decay 0.05 on every parameter that has a gradient).

One step's outputs / losses / 323 gradients are pinned elsewhere (tests/test_gpu_configs.py, test_gpu_e2e.py), AdamW alone against
torch.optim.AdamW in tests/test_gpu_engine.py; what only a trajectory shows is their COMPOSITION over steps: the update applied to the
right parameter with the right per-parameter step count, nothing stale between steps (shadows, transposed / padded shadows, flat
gradient buffer zeroing, deferred split-K sums), a fresh explicit mask every step.

  fp32 mode: every step's losses within 1e-4 of the oracle's, every final weight tensor within 1e-5 (max-abs, relative to max|w|).
  bf16 mode: every step's losses within max(1e-2, 1.5 x anchor) (at this learning rate the contrastive term falls from 15 to 0.65 in five
             steps and the reference arithmetic's own bf16 trajectory is 2.3e-2 off on it by step 5); the total update  w_k - w_0  of all parameters within
             max(1e-2, 1.5 x the reference arithmetic's own bf16 trajectory error) in relative L2 (the anchor of tests/parity.py,
             here over a trajectory: the oracle stepped under CPU autocast(bfloat16))."""
import pytest
import torch

from oracle import mmae_oracle as O
from tests import parity
from tests.test_gpu_configs import _masks, _model
from tests.test_gpu_kernels import DEV

pytestmark = pytest.mark.gpu

STEPS, LR, HEADS, SIZE, B = 5, 1e-3, 3, 64, 2
KEEP = [{"s1": 9, "s2": 7, "dem": 0}, {"s1": 6, "s2": 5, "dem": 5}, {"s1": 4, "s2": 8, "dem": 4}, {"s1": 0, "s2": 10, "dem": 6},
        {"s1": 7, "s2": 2, "dem": 7}]                                          # 16 kept tokens per step, a dropped modality in two of them


def _oracle_trajectory(state, x, masks, bf16):
    p = parity.leaf_params(state)
    train = [t for t in p.values() if t.requires_grad]
    opt = torch.optim.AdamW(train, lr=LR, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8)
    losses = []
    for m in masks:
        opt.zero_grad(set_to_none=True)                                        # a parameter without a gradient is skipped, as in the engine
        _, (tl, lc, loss) = O.train_step_loss(p, x, m, 16, HEADS, 8, 16, bf16=bf16)
        loss.backward()
        opt.step()
        losses.append({"loss": float(loss.detach()), "loss_contra": float(lc.detach()), **{d + "_loss": float(v.detach()) for d, v in tl.items()}})
    return losses, {k: v.detach().double() for k, v in p.items()}


def _native_trajectory(base_state, x, masks, autocast):
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    model = _model("tiny", SIZE, 31)
    model.load_state_dict(base_state)
    model.to(DEV).train()
    opt = FlatAdamW(model.parameters(), lr=LR, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 16, autocast=autocast)
    xd = {k: v.to(DEV) for k, v in x.items()}
    losses = []
    for m in masks:
        r = step(xd, task_masks={k: v.to(DEV) for k, v in m.items()})
        losses.append({k: float(v) for k, v in r.items()})
    torch.cuda.synchronize()
    assert opt.steps == len(masks) and opt.skipped_steps() == 0
    return losses, {k: v.detach().double().cpu() for k, v in model.state_dict().items()}


def _setup():
    base = _model("tiny", SIZE, 31)
    state = {k: v.detach().clone() for k, v in base.state_dict().items()}
    torch.manual_seed(77)
    x = {"s1": torch.randn(B, 1, SIZE, SIZE), "s2": torch.randn(B, 3, SIZE, SIZE), "dem": torch.randn(B, 1, SIZE, SIZE)}
    masks = [_masks((SIZE // 16) ** 2, B, k) for k in KEEP]
    return state, x, masks


def _loss_errs(got, ref):
    worst = 0.0
    for g, r in zip(got, ref):
        for k, v in r.items():
            worst = max(worst, abs(g[k] - v) / max(abs(v), 1e-6))
    return worst


def test_training_trajectory_vs_oracle_fp32():
    state, x, masks = _setup()
    ref_losses, ref_w = _oracle_trajectory(state, x, masks, bf16=False)
    got_losses, got_w = _native_trajectory(state, x, masks, autocast=False)
    assert ref_losses[0]["loss"] != ref_losses[-1]["loss"]
    e = _loss_errs(got_losses, ref_losses)
    moved, worst, worst_name = 0, 0.0, ""
    for k, r in ref_w.items():
        if not r.dtype.is_floating_point:
            continue
        err = float((got_w[k] - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        if err > worst:
            worst, worst_name = err, k
        moved += int(float((r - state[k].double()).abs().max()) > 0)
    print("\n[trajectory fp32] %d steps: worst loss err %.2e; %d tensors moved; worst final-weight err %.2e (%s)"
          % (STEPS, e, moved, worst, worst_name))
    assert moved >= 300                                                           # the 323 gradient-receiving tensors (minus a never-live path)
    assert e <= 1e-4, (e, got_losses, ref_losses)
    assert worst <= 1e-5, (worst, worst_name)


def test_training_trajectory_vs_oracle_bf16_anchored():
    state, x, masks = _setup()
    ref_losses, ref_w = _oracle_trajectory(state, x, masks, bf16=False)
    anc_losses, anc_w = _oracle_trajectory(state, x, masks, bf16=True)
    got_losses, got_w = _native_trajectory(state, x, masks, autocast=True)
    e, ea = _loss_errs(got_losses, ref_losses), _loss_errs(anc_losses, ref_losses)
    keys = [k for k, r in ref_w.items() if r.dtype.is_floating_point and float((r - state[k].double()).abs().max()) > 0]
    cat = lambda w: torch.cat([(w[k] - state[k].double()).flatten() for k in keys])
    dref = cat(ref_w)
    d_hip = float((cat(got_w) - dref).norm() / dref.norm())
    d_anc = float((cat(anc_w) - dref).norm() / dref.norm())
    print("\n[trajectory bf16] %d steps: worst loss err %.2e (reference-bf16 %.2e); update rel L2 %.3e vs reference-bf16 %.3e (ratio %.2f)"
          % (STEPS, e, ea, d_hip, d_anc, d_hip / max(d_anc, 1e-30)))
    assert e <= max(1e-2, 1.5 * ea), (e, ea)
    assert d_hip <= max(1e-2, 1.5 * d_anc), (d_hip, d_anc)
