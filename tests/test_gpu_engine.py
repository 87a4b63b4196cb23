"""Fused AdamW engine (csrc/optim.hip + engine.py) against torch.optim.AdamW, and the engine-backed training step
against the plain one."""
import copy

import pytest
import torch

from tests.test_gpu_kernels import DEV, close

pytestmark = pytest.mark.gpu


def test_fused_adamw_matches_torch_adamw():
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    torch.manual_seed(0)
    shapes = [(512, 768), (768,), (85, 32), (1, 16, 48), (7,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    unused = torch.nn.Parameter(torch.ones(5, device=DEV))
    o_ref = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    o_mine = FlatAdamW(mine + [unused], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, exclude=[unused])
    assert all(p.data_ptr() >= o_mine.master.data_ptr() for p in mine)
    for step in range(4):
        grads = [torch.randn_like(p) for p in ref]
        o_ref.zero_grad(); o_mine.zero_grad()
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone()
            (q * g).sum().backward()            # through autograd: exercises the post-accumulate hook copy
        gn_ref = torch.norm(torch.stack([torch.norm(p.grad) for p in ref]))
        close(o_mine.grad_norm(), gn_ref, 1e-5, "grad norm")
        o_ref.step(); o_mine.step()
        for p, q in zip(ref, mine):
            close(q, p, 2e-6, "param step %d" % step)
            assert torch.equal(q._mmae_shadow, q.detach().to(torch.bfloat16))
    assert torch.equal(unused, torch.ones(5, device=DEV))


def test_engine_training_step_matches_plain_step():
    """Same model, same batch, same masks: (FlatAdamW + shadow weights + in-place flat gradients) vs
    (torch AdamW + per-use casts).  Gradients agree to bf16 GEMM rounding noise, parameters after the step likewise."""
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    torch.manual_seed(1)
    base = get_model("small", input_size=128, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    base.depth = 2; base.blocks = base.blocks[:2]; base.fus_blocks = base.fus_blocks[:2]
    B, P, N = 32, 64, 96
    x = {"s1": torch.randn(B, 1, 128, 128, device=DEV), "s2": torch.randn(B, 3, 128, 128, device=DEV),
         "dem": torch.randn(B, 1, 128, 128, device=DEV)}
    masks = {}
    for d, k in (("s1", 40), ("s2", 30), ("dem", 26)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)
    results = []
    for use_engine in (False, True):
        model = copy.deepcopy(base).to(DEV).train()
        if use_engine:
            opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05,
                            exclude=model.never_used_parameters())
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
        step = PretrainStep(model, opt, N, autocast=True)
        names = [n for n, _ in model.named_parameters()]
        out = step(x, task_masks=masks)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        out2 = step(x, task_masks=masks)
        params = {n: p.detach().clone() for n, p in model.named_parameters()}
        results.append((float(out["loss"]), float(out2["loss"]), grads, params, names, dict(model.state_dict())))
    (l0, l0b, g0, p0, n0, sd0), (l1, l1b, g1, p1, n1, sd1) = results
    assert n0 == n1 and sd0.keys() == sd1.keys()                 # the engine does not disturb the module surface
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0))               # identical forward (bf16(fp32 weight) == shadow)
    # Adam's first update is ~lr*sign(g): bf16 noise on near-zero gradients legitimately flips single elements, so the
    # second-step loss is only required to track (the exact update rule is pinned by the test above)
    assert abs(l0b - l1b) < 5e-2 * max(1.0, abs(l0b)), (l0b, l1b)
    assert g0.keys() == g1.keys()
    for n in g0:
        a, b = g1[n].double().flatten(), g0[n].double().flatten()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        assert rel < 2e-2, (n, rel)
    for n in p0:
        d = (p1[n] - p0[n]).abs().max()
        assert float(d) <= 2.1e-3, (n, float(d))                # at most one flipped lr-sized step per element


@pytest.mark.parametrize("use_engine", [False, True])
def test_training_reduces_loss(use_engine):
    """End-to-end sanity of the whole native step (kernels + autograd wiring + optimizer): 40 steps on one fixed batch with
    fixed masks must drive the loss down substantially."""
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    torch.manual_seed(2)
    model = get_model("tiny", input_size=64, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    model.depth = 2; model.blocks = model.blocks[:2]; model.fus_blocks = model.fus_blocks[:2]
    model.to(DEV).train()
    if use_engine:
        opt = FlatAdamW(model.parameters(), lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05,
                        exclude=model.never_used_parameters())
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, betas=(0.9, 0.95), weight_decay=0.05)
    B, P, N = 8, 16, 24
    x = {"s1": torch.randn(B, 1, 64, 64, device=DEV), "s2": torch.randn(B, 3, 64, 64, device=DEV),
         "dem": torch.randn(B, 1, 64, 64, device=DEV)}
    masks = {}
    for d, k in (("s1", 10), ("s2", 8), ("dem", 6)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)
    step = PretrainStep(model, opt, N, autocast=True)
    losses = [float(step(x, task_masks=masks)["loss"]) for _ in range(40)]
    assert all(l == l for l in losses)
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])


def test_transposed_shadow_and_splitk_sum():
    """engine.shadow_t_of: W^T of (row-concatenated) weights kept fresh by ONE batched-transpose launch per update;
    mmae_splitk_sum: fp32 reduction of bf16 split-K partials.  Both bit-exact against torch."""
    from incomplete_multimodal_fusion_amd import ops
    from incomplete_multimodal_fusion_amd._lib import call, ptr, stream
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW, shadow_of, shadow_t_of
    torch.manual_seed(3)
    shapes = [(512, 768), (1024, 768), (768, 2048), (85, 32), (72, 200), (256,)]
    ps = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    opt = FlatAdamW(ps, lr=1e-2)
    cat = shadow_t_of((ps[0], ps[1]), torch.bfloat16)            # [to_q | to_kv]-style concatenation
    one = shadow_t_of((ps[2],), torch.bfloat16)
    small = shadow_t_of((ps[4],), torch.bfloat16)               # ragged 64-tiles (72 x 200)
    assert shadow_t_of((ps[3],), torch.bfloat16) is None        # 85 x 32: not 8-aligned -> torch fallback in ops
    assert shadow_t_of((ps[0],), torch.bfloat16) is None        # overlaps the registered concatenation -> fallback
    assert shadow_t_of((ps[2],), torch.float32) is None
    for step in range(3):
        assert torch.equal(cat, shadow_of((ps[0], ps[1]), torch.bfloat16).t())
        assert torch.equal(one, ps[2].detach().to(torch.bfloat16).t())
        assert torch.equal(small, ps[4].detach().to(torch.bfloat16).t())
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step()
    # the data gradient of ops.linear goes through the transposed shadow
    x = torch.randn(640, 768, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    y = ops.linear(x, [ps[0], ps[1]], once=True)
    g = torch.randn_like(y)
    y.backward(g)
    want = (g.float() @ torch.cat([ps[0], ps[1]]).detach().to(torch.bfloat16).float())
    close(x.grad, want, 1e-2, "dx through the transposed shadow")
    for S, n_out, n_in in ((4, 768, 4096), (16, 768, 1024), (3, 40, 24)):
        part = torch.randn(S, n_out, n_in, device=DEV).to(torch.bfloat16)
        out = torch.empty(n_out, n_in, device=DEV)
        call("mmae_splitk_sum", S, n_out * n_in, ptr(part), ptr(out), stream())
        ref = torch.zeros(n_out, n_in, device=DEV)
        for s in range(S):
            ref += part[s].float()
        assert torch.equal(out, ref)


def _pair(shapes, **kw):
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    ref = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    o_mine = FlatAdamW(mine, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    return ref, mine, o_ref, o_mine


def _feed(ref, mine, o_ref, o_mine, grads):
    o_ref.zero_grad(); o_mine.zero_grad()
    for p, q, g in zip(ref, mine, grads):
        if g is None:
            continue
        p.grad = g.clone()
        (q * g).sum().backward()


def test_device_side_clip_matches_torch_clip_grad_norm():
    """clip_grad of NativeScaler (native_scaler.py:23-26): torch.nn.utils.clip_grad_norm_ + AdamW, without a host sync."""
    torch.manual_seed(10)
    shapes = [(256, 768), (768,), (85, 32), (7,)]
    ref, mine, o_ref, o_mine = _pair(shapes)
    for step, scale in enumerate([1.0, 30.0, 0.01, 5.0]):             # norms above and below max_norm = 3
        grads = [scale * torch.randn_like(p) for p in ref]
        _feed(ref, mine, o_ref, o_mine, grads)
        norm = torch.nn.utils.clip_grad_norm_(ref, 3.0)
        o_ref.step()
        o_mine.step(clip_grad=3.0)
        close(o_mine.last_grad_norm(), norm, 1e-5, "norm step %d" % step)
        assert not o_mine.last_step_skipped()
        for p, q in zip(ref, mine):
            close(q, p, 3e-6, "param step %d" % step)


def test_device_side_skip_and_non_finite_guard_match_not_stepping():
    """skip_grad (native_scaler.py:27-32) and GradScaler's found-inf rule: the optimizer is NOT stepped -- weights, moments
    and the step count used for bias correction stay; later steps agree with a torch optimizer that skipped the same ones."""
    torch.manual_seed(11)
    shapes = [(128, 96), (96,), (33, 8)]
    ref, mine, o_ref, o_mine = _pair(shapes)
    plan = [("ok", 1.0), ("big", 1e4), ("ok", 1.0), ("nan", 1.0), ("inf", 1.0), ("ok", 1.0)]     # |g| ~ 112 when ok, ~1e6 when big
    for step, (kind, scale) in enumerate(plan):
        grads = [scale * torch.randn_like(p) for p in ref]
        if kind == "nan":
            grads[0][3, 5] = float("nan")
        if kind == "inf":
            grads[2][0, 0] = float("inf")
        _feed(ref, mine, o_ref, o_mine, grads)
        before = [q.detach().clone() for q in mine]
        if kind == "ok":
            o_ref.step()
        o_mine.step(skip_grad=5000.0, check_finite=True)
        assert o_mine.last_step_skipped() == (kind != "ok"), (step, kind)
        if kind != "ok":
            for q, b in zip(mine, before):
                assert torch.equal(q, b), "a skipped step must not touch the weights"
        for p, q in zip(ref, mine):
            close(q, p, 3e-6, "param step %d (%s)" % (step, kind))
            assert torch.isfinite(q).all()
    assert o_mine.skipped_steps() == 3
    assert all(o_mine.param_step(q) == 3 for q in mine)
    for q, p in zip(mine, ref):                                      # torch's per-parameter step count agrees
        assert float(o_ref.state[p]["step"]) == 3.0


def test_parameters_without_gradient_are_left_alone_like_torch_adamw():
    """torch.optim.AdamW skips `p.grad is None`: no weight decay, no moment decay, its own step count (the downstream
    backbone's per-forward modality subsets do this to the patch-embedding weights)."""
    torch.manual_seed(12)
    shapes = [(64, 48), (48,), (40, 16), (16,), (9, 8)]
    ref, mine, o_ref, o_mine = _pair(shapes)
    present = [[1, 1, 1, 1, 1], [1, 0, 1, 1, 0], [0, 0, 1, 1, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1]]
    for step, pres in enumerate(present):
        grads = [torch.randn_like(p) if on else None for p, on in zip(ref, pres)]
        _feed(ref, mine, o_ref, o_mine, grads)
        o_ref.step(); o_mine.step()
        for i, (p, q) in enumerate(zip(ref, mine)):
            close(q, p, 3e-6, "param %d step %d" % (i, step))
            assert torch.equal(q._mmae_shadow, q.detach().to(torch.bfloat16))
    for i, (p, q) in enumerate(zip(ref, mine)):
        assert o_mine.param_step(q) == int(float(o_ref.state[p]["step"])) == sum(pr[i] for pr in present)


def test_second_backward_before_zero_grad_raises_for_in_place_gradients():
    """Weight gradients that producers write straight into the flat buffer are assigned, not accumulated: a second backward
    before zero_grad() must fail loudly instead of silently dropping the first gradient."""
    from incomplete_multimodal_fusion_amd import ops
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    torch.manual_seed(13)
    w = torch.nn.Parameter(torch.randn(256, 128, device=DEV))
    opt = FlatAdamW([w], lr=1e-3)
    x = torch.randn(64, 128, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    opt.zero_grad()
    ops.linear(x, w, once=True).float().sum().backward()
    with pytest.raises(RuntimeError, match="one backward per zero_grad"):
        ops.linear(x, w, once=True).float().sum().backward()
    opt.zero_grad()
    ops.linear(x, w, once=True).float().sum().backward()             # fine again after zero_grad()
    opt.step()


def test_pretrain_step_clip_and_finite_guard():
    """PretrainStep wires clip_grad / the non-finite guard to the engine: a NaN input batch leaves the model untouched."""
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    torch.manual_seed(14)
    model = get_model("tiny", input_size=64, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    model.depth = 2; model.blocks = model.blocks[:2]; model.fus_blocks = model.fus_blocks[:2]
    model.to(DEV).train()
    opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 24, autocast=True, clip_grad=0.5)
    B = 4
    x = {"s1": torch.randn(B, 1, 64, 64, device=DEV), "s2": torch.randn(B, 3, 64, 64, device=DEV),
         "dem": torch.randn(B, 1, 64, 64, device=DEV)}
    step(x)
    assert not opt.last_step_skipped() and float(opt.last_grad_norm()) > 0
    before = opt.master.clone()
    bad = {k: v.clone() for k, v in x.items()}
    bad["s2"][1, 2, 5, 5] = float("nan")
    step(bad)
    assert opt.last_step_skipped()
    assert torch.equal(opt.master, before) and torch.isfinite(opt.master).all()
    step(x)
    assert not opt.last_step_skipped() and not torch.equal(opt.master, before)


def test_loss_balancer_companion_follows_the_engine(tmp_path):
    """a trainable loss balancer (UncertaintyWeightingStrategy.log_vars) next to the flat engine is optimised by a
    companion AdamW inside PretrainStep.  (1) Its trajectory equals the second group of ONE torch AdamW built the reference way
    (create_optimizer({'model', 'balancer'}), utils/optim_factory.py:136-150).  (2) A step the engine's device-side control skips
    (non-finite gradient norm: GradScaler's guard in the reference, native_scaler.py:24-37) skips log_vars, their moments and
    their step count too -- without it one NaN loss left log_vars NaN for good.  (3) save_model / auto_load_model round trip."""
    from incomplete_multimodal_fusion_amd import checkpoint as C
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, UncertaintyWeightingStrategy, create_optimizer, get_model
    torch.manual_seed(3)
    base = get_model("small", input_size=128, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    base.depth = 2; base.blocks = base.blocks[:2]; base.fus_blocks = base.fus_blocks[:2]
    B, P, N, doms = 8, 64, 96, ["s1", "s2", "dem"]
    x = {"s1": torch.randn(B, 1, 128, 128, device=DEV), "s2": torch.randn(B, 3, 128, 128, device=DEV),
         "dem": torch.randn(B, 1, 128, 128, device=DEV)}
    masks = {}
    for d, k in (("s1", 40), ("s2", 30), ("dem", 26)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)
    lr, scale = 1e-3, 2.0
    model_r, bal_r = copy.deepcopy(base).to(DEV).train(), UncertaintyWeightingStrategy(doms).to(DEV)
    opt_r = create_optimizer(model_r, bal_r, lr=lr, balancer_lr_scale=scale)
    opt_r.param_groups[1]["lr"] = lr * scale                       # what the driver's per-step lr assignment does (pretrain_mmae.py:439-445)
    step_r = PretrainStep(model_r, opt_r, N, loss_balancer=bal_r)
    model_e, bal_e = copy.deepcopy(base).to(DEV).train(), UncertaintyWeightingStrategy(doms).to(DEV)
    opt_e = FlatAdamW(model_e.parameters(), lr=lr, betas=(0.9, 0.95), weight_decay=0.05, exclude=model_e.never_used_parameters())
    step_e = PretrainStep(model_e, opt_e, N, loss_balancer=bal_e, balancer_lr_scale=scale, check_finite=True)
    assert step_e.balancer_opt is not None and step_r.balancer_opt is None
    for it in range(3):
        step_r(x, task_masks=masks); step_e(x, task_masks=masks)
        assert float((bal_e.log_vars - bal_r.log_vars).abs().max()) < 2e-5, (it, bal_e.log_vars, bal_r.log_vars)
    assert float(bal_e.log_vars.abs().max()) > 1e-3                # they did move
    # (2) a poisoned batch: NaN loss -> the engine skips the model update; log_vars and their Adam state must sit it out too
    bad = dict(x); bad["s1"] = x["s1"].clone(); bad["s1"][0, 0, 0, 0] = float("nan")
    lv0 = bal_e.log_vars.detach().clone()
    st0 = {k: v.detach().clone() for k, v in step_e.balancer_opt.state[bal_e.log_vars].items() if torch.is_tensor(v)}
    w0 = opt_e.master.clone()
    out = step_e(bad, task_masks=masks)
    assert not torch.isfinite(out["loss"]) and opt_e.last_step_skipped()
    assert torch.equal(opt_e.master, w0) and torch.equal(bal_e.log_vars, lv0) and torch.isfinite(bal_e.log_vars).all()
    for k, v in step_e.balancer_opt.state[bal_e.log_vars].items():
        if torch.is_tensor(v):
            assert torch.equal(v, st0[k]), k                        # moments AND step count
    step_r(x, task_masks=masks); step_e(x, task_masks=masks)        # the reference-way run never saw the poisoned batch
    assert float((bal_e.log_vars - bal_r.log_vars).abs().max()) < 5e-5
    # (3) checkpoint round trip through the reference's joint two-group layout
    C.save_model(str(tmp_path), 1, model_e, opt_e, loss_balancer=bal_e, balancer_lr_scale=scale, balancer_optimizer=step_e.balancer_opt)
    model_f, bal_f = copy.deepcopy(base).to(DEV).train(), UncertaintyWeightingStrategy(doms).to(DEV)
    opt_f = FlatAdamW(model_f.parameters(), lr=9.0, betas=(0.9, 0.95), weight_decay=0.0, exclude=model_f.never_used_parameters())
    step_f = PretrainStep(model_f, opt_f, N, loss_balancer=bal_f, balancer_lr_scale=scale, check_finite=True)
    assert C.auto_load_model(str(tmp_path), model_f, opt_f, loss_balancer=bal_f, balancer_optimizer=step_f.balancer_opt,
                             map_location="cuda") == 2
    assert torch.equal(bal_f.log_vars, bal_e.log_vars) and opt_f.param_groups[0]["lr"] == lr
    step_e(x, task_masks=masks); step_f(x, task_masks=masks)
    assert float((bal_f.log_vars - bal_e.log_vars).abs().max()) < 1e-6
    model_t, bal_t = copy.deepcopy(base).to(DEV).train(), UncertaintyWeightingStrategy(doms).to(DEV)
    opt_t = create_optimizer(model_t, bal_t, lr=9.0, balancer_lr_scale=scale)
    assert C.auto_load_model(str(tmp_path), model_t, opt_t, loss_balancer=bal_t, map_location="cuda") == 2
    assert opt_t.param_groups[1]["lr"] == pytest.approx(lr * scale) and opt_t.param_groups[1]["lr_scale"] == scale



def test_failed_backward_does_not_poison_the_next_step():
    """the deferred split-K sums of weight gradients used to depend on a sticky 'callback registered' flag that only the
    end-of-backward callback cleared.  A backward() that raises after the first deferral skips autograd's final callbacks: the flag
    stayed set, later passes never flushed the sums queued after the last per-layer flush (gradients never published, weights
    silently frozen), and the failed pass's stale partial products were summed into the next pass's buffer.  Now: a step after a
    failed backward gives exactly the gradients of a step that never saw the failure, and every engine parameter is published."""
    from incomplete_multimodal_fusion_amd import ops
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
    torch.manual_seed(2)
    base = get_model("small", input_size=128, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    base.depth = 2; base.blocks = base.blocks[:2]; base.fus_blocks = base.fus_blocks[:2]
    B, P, N = 64, 64, 96                                  # 10 240 rows: the weight gradients split (S > 1) and are deferred
    x = {"s1": torch.randn(B, 1, 128, 128, device=DEV), "s2": torch.randn(B, 3, 128, 128, device=DEV),
         "dem": torch.randn(B, 1, 128, 128, device=DEV)}
    masks = {}
    for d, k in (("s1", 40), ("s2", 30), ("dem", 26)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).to(DEV)

    def make():
        model = copy.deepcopy(base).to(DEV).train()
        opt = FlatAdamW(model.parameters(), lr=0.0, betas=(0.9, 0.95), weight_decay=0.0, exclude=model.never_used_parameters())
        return model, opt, PretrainStep(model, opt, N, autocast=True, check_finite=False)
    model_a, opt_a, step_a = make()
    model_b, opt_b, step_b = make()
    step_a(x, task_masks=masks)
    clean = opt_a.grads.clone()
    assert all(p.grad is not None for p in opt_a.params)
    # model b: the first backward dies inside the second per-layer flush, i.e. after deferrals were queued and one flush ran
    real, n = ops.flush_splitk, [0]
    deferred = [0]
    real_append = ops._SPLITK_Q.append

    def dying_flush():
        n[0] += 1
        if n[0] == 2:
            deferred[0] = len(ops._SPLITK_Q)
            raise RuntimeError("injected failure inside backward")
        return real()
    ops.flush_splitk = dying_flush
    try:
        with pytest.raises(RuntimeError, match="injected failure"):
            step_b(x, task_masks=masks)
    finally:
        ops.flush_splitk = real
    assert deferred[0] > 0, "the scenario needs deferred split-K sums in flight when backward fails"
    torch.cuda.synchronize()
    step_b(x, task_masks=masks)                          # lr = 0: the weights are those of model a
    assert all(p.grad is not None for p in opt_b.params), [i for i, p in enumerate(opt_b.params) if p.grad is None][:5]
    assert not ops._SPLITK_Q
    assert torch.equal(opt_b.grads, clean)
