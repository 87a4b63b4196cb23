"""The pretraining step captured into ONE hipGraph (PretrainStep.capture / replay): the same launches in the same order, so the replays
must reproduce the eager steps bit for bit -- losses, weights, optimizer state --, including what a graph bakes in by value and the
optimizer therefore reads from the device: the step count of the bias correction, the learning rate and the weight decay."""
import copy

import pytest
import torch

from tests.test_gpu_kernels import DEV

pytestmark = pytest.mark.gpu


def _setup(seed=3):
    from incomplete_multimodal_fusion_amd.pretrain import get_model
    torch.manual_seed(seed)
    base = get_model("small", input_size=128, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    base.depth = 2; base.blocks = base.blocks[:2]; base.fus_blocks = base.fus_blocks[:2]
    B, P = 8, 64
    x = {"s1": torch.randn(B, 1, 128, 128, device=DEV), "s2": torch.randn(B, 3, 128, 128, device=DEV),
         "dem": torch.randn(B, 1, 128, 128, device=DEV)}
    masks = {}
    for d, k in (("s1", 40), ("s2", 30), ("dem", 26)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)
    return base, x, masks


def _step(base, clip=None):
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    model = copy.deepcopy(base).to(DEV).train()
    opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    return model, opt, PretrainStep(model, opt, 96, clip_grad=clip, check_finite=True)


@pytest.mark.parametrize("clip", [None, 0.5])
def test_captured_step_replays_bitwise_like_eager(clip):
    base, x, masks = _setup()
    _, opt_e, step_e = _step(base, clip)
    _, opt_g, step_g = _step(base, clip)
    sched = [(1e-3, 0.05), (1e-3, 0.05), (7e-4, 0.05), (7e-4, 0.02), (2e-4, 0.0), (2e-4, 0.0)]   # (lr, weight decay) per step

    def hyper(opt, i):
        opt.param_groups[0]["lr"], opt.param_groups[0]["weight_decay"] = sched[i]
    losses_e = []
    for i in range(6):
        hyper(opt_e, i)
        losses_e.append(step_e(x, task_masks=masks)["loss"].clone())
    hyper(opt_g, 0)                                   # capture() runs its two warm-up steps eagerly: steps 0 and 1 of the schedule
    step_g.capture(x, masks, warmup=2)
    assert opt_g.steps == 2
    for i in range(2, 6):
        hyper(opt_g, i)
        out = step_g.replay()
        assert torch.equal(out["loss"], losses_e[i]), (i, float(out["loss"]), float(losses_e[i]))
    assert opt_g.steps == opt_e.steps == 6
    assert torch.equal(opt_g.master, opt_e.master) and torch.equal(opt_g.exp_avg, opt_e.exp_avg)
    assert torch.equal(opt_g.exp_avg_sq, opt_e.exp_avg_sq) and torch.equal(opt_g.shadow, opt_e.shadow)
    assert not opt_g.last_step_skipped()
    # new explicit masks are copied into the captured mask tensors: same result as the eager step on them
    masks2 = {d: m.flip(1).contiguous() for d, m in masks.items()}
    hyper(opt_e, 5); hyper(opt_g, 5)
    le = step_e(x, task_masks=masks2)["loss"]
    lg = step_g.replay(None, masks2)["loss"]
    assert torch.equal(le, lg) and torch.equal(opt_g.master, opt_e.master)


def test_captured_step_takes_new_batches_and_fresh_mask_draws():
    """Random masks: the Dirichlet shares are drawn on the host before every replay and reach the graph through a static device tensor;
    a new batch is copied into the captured inputs.  The replays train (loss falls on a fixed batch) and differ from step to step."""
    base, x, _ = _setup(5)
    _, opt, step = _step(base)
    step.capture(x, None)
    first = float(step.replay()["loss"])
    seen = set()
    for _ in range(25):
        out = step.replay()
        seen.add(round(float(out["loss"]), 6))
    assert len(seen) > 20                              # fresh masks every replay
    assert float(out["loss"]) < first and torch.isfinite(out["loss"])
    with pytest.raises(RuntimeError):
        step(x)                                        # eager calls are refused once the step is a graph
    x2 = {k: v.flip(0).contiguous() for k, v in x.items()}
    before = {k: v.clone() for k, v in x.items()}
    step.replay(x2)
    assert all(torch.equal(x[k], x2[k]) and not torch.equal(x[k], before[k]) for k in x)     # copied into the captured tensors


def test_capture_refuses_what_it_cannot_hold():
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    base, x, masks = _setup()
    model = copy.deepcopy(base).to(DEV).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    with pytest.raises(NotImplementedError):
        PretrainStep(model, opt, 96).capture(x, masks)


def test_resume_into_a_captured_step_replays_like_eager():
    """capture -> load a checkpoint (model + optimizer state in the reference layout) -> replay: the replays must continue from the
    RESTORED step counts, learning rate and weight decay, bitwise like the eager run the checkpoint came from (the load
    used to wipe the replay counter and the lr/wd hand-over flag of the optimizer's control block)."""
    from incomplete_multimodal_fusion_amd import checkpoint
    base, x, masks = _setup(7)
    model_e, opt_e, step_e = _step(base, 0.5)
    model_g, opt_g, step_g = _step(base, 0.5)
    sched = [(1e-3, 0.05), (9e-4, 0.05), (7e-4, 0.04), (6e-4, 0.03), (4e-4, 0.02), (2e-4, 0.01)]

    def hyper(opt, i):
        opt.param_groups[0]["lr"], opt.param_groups[0]["weight_decay"] = sched[i]
    for i in range(4):
        hyper(opt_e, i)
        step_e(x, task_masks=masks)
    snap_model = {k: v.detach().clone() for k, v in model_e.state_dict().items()}
    snap_opt = checkpoint.optimizer_state_dict(opt_e, model_e)
    tail = []
    for i in (4, 5):
        hyper(opt_e, i)
        tail.append(step_e(x, task_masks=masks)["loss"].clone())
    # the captured step has seen a DIFFERENT history (two warm-up steps + one replay at other hyper-parameters)
    hyper(opt_g, 5)
    step_g.capture(x, masks, warmup=2)
    step_g.replay()
    model_g.load_state_dict(snap_model, strict=True)
    checkpoint.load_optimizer_state_dict(opt_g, model_g, snap_opt)
    assert opt_g.steps == 4
    for n, i in enumerate((4, 5)):
        hyper(opt_g, i)
        out = step_g.replay()
        assert torch.equal(out["loss"], tail[n]), (i, float(out["loss"]), float(tail[n]))
    assert opt_g.steps == opt_e.steps == 6
    assert torch.equal(opt_g.master, opt_e.master) and torch.equal(opt_g.exp_avg, opt_e.exp_avg)
    assert torch.equal(opt_g.exp_avg_sq, opt_e.exp_avg_sq) and torch.equal(opt_g.shadow, opt_e.shadow)


def test_failed_capture_leaves_the_step_eager_and_unchanged():
    """A capture that raises inside the captured region must leave the step as it was: fresh mask draws per call (not the one
    static share), the optimizer's step counts untouched by the step that never ran, eager calls still allowed."""
    base, x, _ = _setup(9)
    model, opt, step = _step(base)
    step(x)
    steps_before = opt.steps
    orig = step._optimizer_step
    calls = [0]

    def failing_step():
        calls[0] += 1
        if calls[0] == 2:                              # call 1: the eager warm-up step; call 2: inside the capture, before opt.step()
            raise RuntimeError("injected failure inside the captured region")
        return orig()
    step._optimizer_step = failing_step
    with pytest.raises(RuntimeError, match="injected failure"):
        step.capture(x, None, warmup=1)
    step._optimizer_step = orig
    torch.cuda.synchronize()
    assert getattr(step, "_graph", None) is None and model.mask_draws is None and step._draws is None
    assert opt.steps == steps_before + 1               # the ONE warm-up step ran eagerly; the failed captured step advanced nothing
    assert all(v == opt.steps for v in opt._pstep.values())
    a = float(step(x)["loss"]); b = float(step(x)["loss"])
    assert a == a and b == b and a != b                # still training, still drawing fresh masks
    step.capture(x, None)
    with pytest.raises(RuntimeError, match="already captured"):
        step.capture(x, None)


def test_capture_that_fails_after_the_optimizer_step_leaves_no_graph_state():
    """a capture that raises AFTER opt.step() was recorded used to leave _graph_params / _graph_base populated and the
    control block's replay slots dirty -- a later resume then hit 'resuming into a captured step' or bias-corrected every eager
    step with a stale replay count.  engine.abort_capture() (called by PretrainStep.capture's except path) must leave the engine
    exactly as an engine that never captured: same loss trajectory as an untouched twin, resume allowed."""
    import copy
    base, x, masks = _setup(11)
    model, opt, step = _step(base)
    model_t, opt_t, step_t = _step(copy.deepcopy(base))               # the twin never captures
    for s_ in (step, step_t):
        s_(x, task_masks=masks)
    orig = step._optimizer_step
    calls = [0]

    def step_then_fail():
        calls[0] += 1
        r = orig()
        if calls[0] == 2:                              # call 1: the eager warm-up step; call 2: inside the capture, AFTER opt.step() was recorded
            raise RuntimeError("injected failure behind the optimizer step")
        return r
    step._optimizer_step = step_then_fail
    with pytest.raises(RuntimeError, match="behind the optimizer step"):
        step.capture(x, masks, warmup=1)
    step._optimizer_step = orig
    step_t(x, task_masks=masks)                        # the twin takes the warm-up step too
    torch.cuda.synchronize()
    assert step._graph is None and opt._graph_params == [] and opt._graph_base == {} and opt._graph_lr is None
    assert float(opt._ctl[4:8].abs().max()) == 0.0
    assert opt.steps == opt_t.steps and all(v == opt.steps for v in opt._pstep.values())
    opt.load_state_dict(opt.state_dict())              # a resume must not see a captured step
    a = [float(step(x, task_masks=masks)["loss"]) for _ in range(3)]
    b = [float(step_t(x, task_masks=masks)["loss"]) for _ in range(3)]
    assert a == b, (a, b)
    assert torch.equal(opt.master, opt_t.master)


def test_captured_step_with_padded_feedforward_replays_bitwise(monkeypatch):
    """The padded-FeedForward route (odd GEGLU width on the own GEMM: engine.padded_ff) inside a captured step: the refresh of the padded
    copies (mmae_pad_copy_bf16_batched) is one of the captured launches, so the replays must track the weight updates exactly like the
    eager steps -- losses and optimizer state bitwise equal."""
    from incomplete_multimodal_fusion_amd import ops
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    from tests.test_cabi_symbols import build_model
    from tests.test_gpu_configs import ODD_FFI
    monkeypatch.setattr(ops, "_OWN_GEMM_MIN_TILES", 0)
    monkeypatch.setattr(ops, "PAD_FF_MIN_TILES", 0)
    torch.manual_seed(13)
    channels = (("s1", 1), ("s2", 3), ("dem", 1))
    base = build_model(ODD_FFI, channels)
    B, P = 8, 64
    x = {d: torch.randn(B, c, 128, 128, device=DEV) for d, c in channels}
    masks = {}
    for d, k in (("s1", 40), ("s2", 30), ("dem", 26)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)

    def make():
        model = copy.deepcopy(base).to(DEV).train()
        opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
        return model, opt, PretrainStep(model, opt, 96, clip_grad=0.5, check_finite=True)
    _, opt_e, step_e = make()
    _, opt_g, step_g = make()
    losses = [step_e(x, task_masks=masks)["loss"].clone() for _ in range(5)]
    assert len(opt_e._pad) == 4, "the padded route must be the one that ran"
    step_g.capture(x, masks, warmup=2)
    for i in range(2, 5):
        out = step_g.replay()
        assert torch.equal(out["loss"], losses[i]), (i, float(out["loss"]), float(losses[i]))
    assert torch.equal(opt_g.master, opt_e.master) and torch.equal(opt_g.shadow, opt_e.shadow)
    pe, pg = list(opt_e._pad.values()), list(opt_g._pad.values())
    assert all(torch.equal(a.w1p, b.w1p) and torch.equal(a.w2pt, b.w2pt) for a, b in zip(pe, pg))
