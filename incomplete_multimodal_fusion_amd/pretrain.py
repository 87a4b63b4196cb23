"""Host-side restatement of the reference's pretraining step on top of the native MultiMAE path.

Reference: pretraining/pretrain_mmae.py -- DOMAIN_CONF :45-72, get_model :188-248, and the step of train_one_epoch
:447-517 (H2D of the tile stack, autocast forward, per-task masked losses :479-486, 3x DINO-style contrastive loss
:489-493, weighted sum :499-500, backward, optimizer step through NativeScaler with clip_grad / skip_grad,
utils/native_scaler.py:20-40, step-level cosine schedule :65-82); the hard-negative contrastive variant of
pretrain_mmae_s2dsm.py:482-492; the 4-modality domain table of pretrain_mmae_my.py:46-81.  Logging / dataset code is out
of scope.  Device tensors only.
"""
import math
from functools import partial
from typing import Dict, Iterable, Optional

import numpy as np
import torch
from torch import nn

from . import ops
from .multimae import (FusionInputAdapter, HardNegtive_loss, MaskedL1Loss, MaskedMSELoss, PatchedInputAdapter,
                       SpatialOutputAdapter, TokenTypes, dino_loss_func)
from .multimae import multimae_crossattn as mc
from .multimae.criterion import MaskedCrossEntropyLoss
from .multimae.input_adapters import SemSegInputAdapter
from .multimae.zorro_utils import TokenTypesQuad

DOMAIN_CONF = {                                   # pretrain_mmae.py:45-72
    's1': {'channels': 1, 'stride_level': 1, 'loss': MaskedMSELoss},
    's2': {'channels': 3, 'stride_level': 1, 'loss': MaskedMSELoss},
    'dem': {'channels': 1, 'stride_level': 1, 'loss': MaskedL1Loss},
    'fusion': {'channels': 1, 'stride_level': 1},
}

DOMAIN_CONF_QUAD = {                              # pretrain_mmae_my.py:46-81 (s1-s2-dem-dnw)
    's1': {'channels': 2, 'stride_level': 1, 'loss': MaskedMSELoss},
    's2': {'channels': 4, 'stride_level': 1, 'loss': MaskedMSELoss},
    'dem': {'channels': 1, 'stride_level': 1, 'loss': MaskedL1Loss},
    'dnw': {'num_classes': 9, 'channels': 9, 'stride_level': 1, 'dim_class_emb': 64,
            'loss': partial(MaskedCrossEntropyLoss, label_smoothing=0.0)},
    'fusion': {'channels': 1, 'stride_level': 1},
}

PRESETS = {'tiny': (192, 12, 3), 'small': (384, 12, 8), 'base': (768, 12, 8), 'large': (1024, 24, 8)}


def get_model(model: str = 'base', in_domains=('s1', 's2', 'dem'), out_domains=None, patch_size: int = 16,
              input_size: int = 256, decoder_dim: int = 256, decoder_depth: int = 2, decoder_num_heads: int = 8,
              dim_head: int = 64, domain_conf: Optional[dict] = None, fusion_blocks: bool = True, drop_path_rate: float = 0.0,
              decoder_drop_path_rate: float = 0.0):
    """Adapters + model as get_model builds them (pretrain_mmae.py:193-246); `model` picks the size preset (the
    reference ignores --model and always builds the tiny factory, :239 -- SURVEY.md 0.5).  `domain_conf` defaults to the
    3-modality table, or to the 4-modality one when 'dnw' is among the domains.
    fusion_blocks=False builds the reference's own 4-modality model (multimae_quadruplet.MultiMAE, what
    pretrain_mmae_my.py:248-255 builds: Zorro-masked blocks only, 5-tuple output; pinned by tests/golden/quad_tiny.npz).
    NOTE: the reference has no M = 4 model WITH fusion blocks; fusion_blocks=True with four domains is this package's
    extension of the 3-modality algorithm to M modalities and has no reference-generated fixture."""
    out_domains = tuple(in_domains) if out_domains is None else tuple(out_domains)
    conf = domain_conf or (DOMAIN_CONF_QUAD if 'dnw' in in_domains else DOMAIN_CONF)

    def in_adapter(d):
        c = conf[d]
        if 'num_classes' in c:
            return SemSegInputAdapter(num_classes=c['num_classes'], stride_level=c['stride_level'],
                                      patch_size_full=patch_size, image_size=input_size,
                                      dim_class_emb=c.get('dim_class_emb', 64), interpolate_class_emb=False)
        return PatchedInputAdapter(num_channels=c['channels'], stride_level=c['stride_level'], patch_size_full=patch_size,
                                   image_size=input_size)
    input_adapters = {d: in_adapter(d) for d in in_domains}
    output_adapters = {
        d: SpatialOutputAdapter(num_channels=conf[d]['channels'], stride_level=conf[d]['stride_level'],
                                patch_size_full=patch_size, dim_tokens=decoder_dim, depth=decoder_depth,
                                num_heads=decoder_num_heads, use_task_queries=True, task=d,
                                context_tasks=list(in_domains), use_xattn=True, drop_path_rate=decoder_drop_path_rate)
        for d in out_domains}
    input_adapters['fusion'] = FusionInputAdapter(num_channels=1, stride_level=1, patch_size_full=patch_size,
                                                  image_size=input_size)
    D, depth, heads = PRESETS[model]
    P = (input_size // patch_size) ** 2
    M = len(in_domains)
    if tuple(in_domains) == ('s1', 's2', 'dem'):
        rtt = (TokenTypes.S1, TokenTypes.S2, TokenTypes.DEM, TokenTypes.FUSION)
    elif tuple(in_domains) == ('s1', 's2', 'dem', 'dnw'):
        rtt = tuple(TokenTypesQuad)
    else:
        from enum import Enum
        rtt = tuple(Enum('TokenTypesM', [(d.upper(), i) for i, d in enumerate(in_domains)] + [('FUSION', M)]))
    cls = mc.MultiMAE
    if not fusion_blocks:
        from .multimae import multimae_quadruplet
        cls = multimae_quadruplet.MultiMAE
    return cls(input_adapters=input_adapters, output_adapters=output_adapters, num_global_tokens=1,
               dim_tokens=D, depth=depth, dim_head=dim_head, heads=heads, ff_mult=4, num_fusion_tokens=P,
               return_token_types=rtt, drop_path_rate=drop_path_rate)     # pretrain_mmae.py:245 (args.drop_path, default 0 at :108)


def make_loss_fns(out_domains=('s1', 's2', 'dem'), patch_size: int = 16, domain_conf: Optional[dict] = None):
    conf = domain_conf or (DOMAIN_CONF_QUAD if 'dnw' in out_domains else DOMAIN_CONF)
    return {d: conf[d]['loss'](patch_size=patch_size, stride=conf[d]['stride_level']) for d in out_domains}


class NoWeightingStrategy(nn.Module):             # utils/task_balancing.py:11-19
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, task_losses):
        return task_losses


class UncertaintyWeightingStrategy(nn.Module):    # utils/task_balancing.py:21-44
    def __init__(self, tasks):
        super().__init__()
        self.tasks = tasks
        self.log_vars = nn.Parameter(torch.zeros(len(tasks)))

    def forward(self, task_losses):
        losses = torch.stack([l.float() for l in task_losses.values()])
        nonzero = (losses != 0.0)
        losses = (torch.exp(-self.log_vars) * losses + self.log_vars) * nonzero     # a dropped task (loss 0) stays 0
        out = task_losses.copy()
        out.update(zip(out, losses))
        return out


def create_optimizer(model, balancer=None, lr: float = 1e-4, weight_decay: float = 0.05, betas=(0.9, 0.95), eps: float = 1e-8,
                     balancer_lr_scale: float = 1.0, fused: bool = False):
    """The reference trainer's optimizer (utils/optim_factory.py:136-179, dict path; call site pretrain_mmae.py:351-352):
    torch AdamW over two groups -- the model's trainable parameters (lr_scale 1) and the loss balancer's
    (lr_scale balancer_lr_scale) -- with weight decay on EVERY parameter (the dict path skips the no-decay filter)."""
    groups = [{"params": [p for _, p in model.named_parameters() if p.requires_grad], "lr_scale": 1.0},
              {"params": [p for _, p in (balancer.named_parameters() if balancer is not None else []) if p.requires_grad],
               "lr_scale": balancer_lr_scale}]
    kw = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
    if fused:
        kw["fused"] = True
    return torch.optim.AdamW(groups, **kw)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Per-iteration schedule: linear warm-up then half-cosine (utils/native_scaler.py:65-82, same signature / result)."""
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    sched = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / len(iters))) for i in iters])
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


def step_losses(out, tasks_dict: Dict[str, torch.Tensor], masks: Dict[str, torch.Tensor], patch_size: int = 16,
                loss_fns=None, contra_weight: Optional[float] = None, contra: str = 'dino', loss_balancer=None,
                contra_loss=None):
    """pretrain_mmae.py:479-500 with NoWeightingStrategy (utils/task_balancing.py:11-19) by default.
    contra='dino': sum_m dino_loss_func(return_token_m, pooled_m), weight 0.3 (pretrain_mmae.py:493,:500);
    contra='hardneg': HardNegtive_loss over every pair of the pooled return tokens (modalities + fusion), weight 1
    (pretrain_mmae_s2dsm.py:482-492 written for M modalities);
    contra='none': task losses only (pretrain_mmae_my.py:514-515) -- the only choice for the 5-tuple of the quadruplet model."""
    preds, _, pooled, _, _, *rets = out
    if contra == 'dino' and not rets:
        raise ValueError("contra='dino' needs the per-modality contrastive return tokens; this model returns none "
                         "(multimae_quadruplet) -- use contra='none' or 'hardneg'")
    loss_fns = loss_fns or make_loss_fns(tuple(preds.keys()), patch_size)
    task_losses = {}
    for task in preds:
        pred = preds[task].float()
        fn = loss_fns[task]
        if isinstance(pred, mc.PredTokens):
            task_losses[task] = fn.forward_tokens(pred.tokens, tasks_dict[task], mask=masks.get(task, None))
        else:
            task_losses[task] = fn(pred, tasks_dict[task], mask=masks.get(task, None))
    feats = [f.squeeze(1) for f in torch.chunk(pooled, pooled.shape[1], dim=1)]   # :489-490 (squeeze the token axis)
    if contra == 'none':
        loss_contra, w = torch.zeros((), device=pooled.device), 0.0
    elif contra == 'dino':
        loss_contra = sum(dino_loss_func(r.squeeze(1), f) for r, f in zip(rets, feats))          # :493
        w = 0.3 if contra_weight is None else contra_weight
    elif contra == 'hardneg':
        fn = contra_loss or HardNegtive_loss()
        loss_contra = sum(fn(feats[i], feats[j]) for i in range(len(feats)) for j in range(i + 1, len(feats)))
        w = 1.0 if contra_weight is None else contra_weight
    else:
        raise ValueError("contra must be 'dino', 'hardneg' or 'none'")
    weighted = loss_balancer(task_losses) if loss_balancer is not None else task_losses
    loss = sum(weighted.values()) + w * loss_contra                                # :499-500
    return task_losses, loss_contra, loss


class PretrainStep:
    """One optimizer step on a batch of tiles already resident on the device.  No host synchronisation.

    clip_grad / skip_grad: NativeScaler's options (utils/native_scaler.py:20-40; the driver passes max_norm /
    max_skip_norm, pretrain_mmae.py:510-513).  With the flat engine they -- and the non-finite-gradient guard torch's
    GradScaler provides for the reference -- are evaluated on the device (engine.FlatAdamW.step); with a torch optimizer
    clip_grad uses torch.nn.utils.clip_grad_norm_ and skip_grad costs one host sync."""

    def __init__(self, model, optimizer, num_encoded_tokens: int, in_domains=None, alphas: float = 1.0,
                 sample_tasks_uniformly: bool = False, autocast: bool = True, patch_size: int = 16,
                 grad_reducer=None, side_stream_wgrad: bool = False, clip_grad: Optional[float] = None,
                 skip_grad: Optional[float] = None, contra: str = 'dino', contra_weight: Optional[float] = None,
                 loss_balancer=None, check_finite: bool = True, balancer_lr_scale: float = 1.0):
        self.model, self.opt = model, optimizer
        self.in_domains = tuple(in_domains) if in_domains is not None else tuple(model.domains)
        self.N, self.alphas, self.uniform = num_encoded_tokens, alphas, sample_tasks_uniformly
        self.autocast, self.patch = autocast, patch_size
        self.loss_fns = make_loss_fns(tuple(model.output_adapters.keys()), patch_size)
        self.reducer = grad_reducer
        self.clip_grad, self.skip_grad, self.check_finite = clip_grad, skip_grad, check_finite
        self.contra, self.contra_weight, self.balancer = contra, contra_weight, loss_balancer
        self.contra_loss = HardNegtive_loss() if contra == 'hardneg' else None
        # The reference optimises the loss balancer's parameters (UncertaintyWeightingStrategy.log_vars) as the second group of
        # its AdamW (utils/optim_factory.py:136-150, lr x balancer lr_scale).  The flat engine holds the MODEL's parameters
        # only, so a trainable balancer next to a FlatAdamW gets a small companion torch AdamW here -- stepped, zeroed,
        # all-reduced and checkpointed with the step (it used to be silently left un-trained with an accumulating .grad).
        from .engine import FlatAdamW
        self.balancer_lr_scale = balancer_lr_scale
        self.balancer_opt = None
        bal = [p for p in loss_balancer.parameters() if p.requires_grad] if isinstance(loss_balancer, torch.nn.Module) else []
        if bal:
            if isinstance(optimizer, FlatAdamW):
                held = {id(p) for p in optimizer.params}
                bal = [p for p in bal if id(p) not in held]
                if bal:
                    # capturable: the step counts live on the device, so a skipped step can be undone without a host sync
                    self.balancer_opt = torch.optim.AdamW(bal, lr=optimizer.param_groups[0]["lr"] * balancer_lr_scale,
                                                          betas=tuple(optimizer.betas), eps=optimizer.eps,
                                                          weight_decay=optimizer.param_groups[0]["weight_decay"],
                                                          capturable=all(p.is_cuda for p in bal))
            else:
                held = {id(p) for g in optimizer.param_groups for p in g["params"]}
                if any(id(p) not in held for p in bal):
                    raise ValueError("the loss balancer has trainable parameters that the optimizer does not hold: build it "
                                     "with create_optimizer({'model': ..., 'balancer': ...}) as the reference driver does")
        model.fuse_unpatchify_loss = True
        model.side_stream_wgrad = side_stream_wgrad

    def _optimizer_step(self):
        from .engine import FlatAdamW
        if isinstance(self.opt, FlatAdamW):
            self.opt.step(clip_grad=self.clip_grad, skip_grad=self.skip_grad, check_finite=self.check_finite)
            if self.balancer_opt is not None:
                for g in self.balancer_opt.param_groups:       # follows the engine's schedule (the driver sets lr per step)
                    g["lr"] = self.opt.param_groups[0]["lr"] * self.balancer_lr_scale
                    g["weight_decay"] = self.opt.param_groups[0]["weight_decay"]
                self._balancer_step(self.opt.skip_flag())
            return
        params = [p for g in self.opt.param_groups for p in g['params'] if p.grad is not None]
        if self.clip_grad is not None:
            torch.nn.utils.clip_grad_norm_(params, self.clip_grad)
        elif self.skip_grad is not None:
            norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in params]), 2.0)
            if float(norm) >= self.skip_grad:                                     # native_scaler.py:29-32
                return
        self.opt.step()

    def _balancer_step(self, skip):
        """Step the companion AdamW of the loss balancer under the SAME decision as the model's step: the reference holds both in
        one optimizer, so GradScaler's non-finite guard / the skip threshold skip log_vars and their moments too
        (native_scaler.py:24-37, optim_factory.py:136-150) -- otherwise one NaN loss would leave log_vars NaN for good and every
        skipped step would advance their bias correction.  `skip`: the engine's device flag (None: uncontrolled step).  The
        update is taken and, where the flag is set, undone by torch.where on the few scalars involved -- no host sync."""
        opt = self.balancer_opt
        if skip is None:
            opt.step()
            return
        snap = []
        for g in opt.param_groups:
            for p in g["params"]:
                st = opt.state.get(p, {})
                snap.append((p, p.detach().clone(), {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}))
        opt.step()
        for p, old, st_old in snap:
            sk = skip.to(p.device)
            p.data.copy_(torch.where(sk, old, p.data))
            for k, v in opt.state.get(p, {}).items():
                if torch.is_tensor(v):
                    o = st_old.get(k)
                    o = torch.zeros_like(v) if o is None else o          # first step skipped: back to the fresh state
                    v.copy_(torch.where(skip.to(v.device), o, v))

    # -- the step as one hipGraph ------------------------------------------------------------------------------------------
    def capture(self, tasks_dict: Dict[str, torch.Tensor], task_masks: Optional[Dict[str, torch.Tensor]] = None, warmup: int = 2):
        """Capture the whole step (forward, losses, backward, optimizer) on THESE input tensors into a hipGraph; afterwards
        replay() runs it with one graph launch instead of ~2500 kernel launches from Python.  The step itself is unchanged -- the
        same launches in the same order, bitwise the same results --, what goes away is the host's enqueue time (~30-45 ms a
        step): that is the whole step at the small configurations (ViT-Small / 128 px: host-bound) and nothing at the headline
        one (GPU-bound).
        The `warmup` steps that precede the capture are REAL training steps on `tasks_dict` (optimizer updates at the learning
        rate of the moment, step counts advanced): a schedule that starts after capture() starts at step `warmup`.  A step can be
        captured once; a failed capture leaves the step eager and unchanged.  The flat engine, fixed shapes; with a gradient reducer
        (dp.GradAllReducer over the engine's flat buffer, static unused set, RCCL) its bucket all-reduces are captured as graph nodes on
        RCCL's stream -- every rank captures and replays in lockstep; new batches are COPIED into the captured input
        tensors (`tasks_dict`'s, kept here) before a replay.  The host-side part of the mask draw (the Dirichlet shares) stays on
        the host: replay() draws and copies them into a static device tensor first.  Eager calls must not be mixed in after
        capture (the optimizer's device-side replay count would fall out of step with the host's count)."""
        from .engine import FlatAdamW
        if not isinstance(self.opt, FlatAdamW) or self.balancer_opt is not None:
            raise NotImplementedError("capture(): the flat engine, no companion optimizer")
        if ops._TIMER is not None or ops._TIMERS or getattr(self.model, "layer_timer", None) is not None:
            raise RuntimeError("capture(): switch the bench timers off first (HIP-event brackets cannot be captured)")
        if getattr(self, "_graph", None) is not None:
            raise RuntimeError("capture(): this step is already captured -- build a new PretrainStep to capture again")
        import torch.cuda.tunable as tun
        tuning = tun.is_enabled() and tun.tuning_is_enabled()
        self._x, self._masks = tasks_dict, task_masks
        B = next(iter(tasks_dict.values())).shape[0]
        M = len([d for d in self.in_domains if d in tasks_dict])
        dev = next(iter(tasks_dict.values())).device
        self._draw_shape = (B if self.model.per_sample_masks else 1, M)
        self._draws = torch.empty(self._draw_shape, dtype=torch.float32, device=dev) if task_masks is None else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # library handles, workspaces, TunableOp choices: settled before capture
            for _ in range(max(1, warmup)):
                self(tasks_dict, task_masks)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if self.reducer is not None:
            ok, why = self.reducer.capturable()             # (after the warm-up steps: the first one agrees the unused-parameter set)
            if not ok:
                raise NotImplementedError("capture(): the gradient reducer cannot be captured -- " + why)
            # every collective the warm-up steps issued must be complete AND retired by the process group's watchdog thread before this
            # thread starts capturing (the watchdog polls the events of earlier collectives; such a poll during a capture aborted the
            # process on this stack).  Deterministic hand-off, no fixed sleep: dp.GradAllReducer.quiesce()
            self.reducer.quiesce()
        if tuning:
            tun.tuning_enable(False)                         # an unseen shape must not start timing runs inside the capture
        self.model.mask_draws = self._draws
        self._next_draw()
        self.opt.begin_capture()
        graph = torch.cuda.CUDAGraph()
        check = getattr(self.model, "check_masks", False)
        self.model.check_masks = False                       # a host-side check of explicit masks: the warm-up steps ran it
        try:
            # with a reducer: thread-local capture mode -- the process group's watchdog thread polls events of earlier collectives,
            # which the global mode forbids to EVERY thread while one captures
            with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if self.reducer is not None else {})):
                self._static_out = self(tasks_dict, task_masks)
        except BaseException:
            # the step stays eager: back to fresh mask draws per call (not the one static share), nothing captured
            self.model.mask_draws = None
            self._draws = None
            self._graph = None
            self.opt.abort_capture()                         # (also when the failure came AFTER the captured opt.step() was recorded)
            raise
        finally:
            self.model.check_masks = check
            self.opt.end_capture()                           # (takes the step counts back only if the captured step() ran)
            if tuning:
                tun.tuning_enable(True)
        self._graph = graph
        if self.reducer is not None:
            self.reducer.captured = True                     # its stats / exposed_ms() describe eager steps only
        return self

    def _next_draw(self):
        if self._draws is not None:
            d = self.model.draw_mask_distribution(self._draw_shape[0], self._draw_shape[1], self.alphas, self.uniform)
            self._draws.copy_(d.pin_memory(), non_blocking=True)

    def replay(self, tasks_dict: Optional[Dict[str, torch.Tensor]] = None, task_masks: Optional[Dict[str, torch.Tensor]] = None):
        """One captured step.  tasks_dict / task_masks: a new batch / new explicit masks (only for a step captured WITH explicit
        masks), copied into the captured tensors (None: the tensors as they are).  Returns the captured step's result tensors
        (overwritten by the next replay)."""
        if tasks_dict is not None and tasks_dict is not self._x:
            for k, v in tasks_dict.items():
                if k in self._x:
                    self._x[k].copy_(v, non_blocking=True)
        if task_masks is not None and task_masks is not self._masks:
            if self._masks is None:
                raise ValueError("this step was captured with random masks")
            for k, v in task_masks.items():
                self._masks[k].copy_(v, non_blocking=True)
        self._next_draw()
        self.opt.replay_begin()
        self._graph.replay()
        self.opt.replay_end()
        return self._static_out

    def __call__(self, tasks_dict: Dict[str, torch.Tensor], task_masks: Optional[Dict[str, torch.Tensor]] = None):
        if getattr(self, "_graph", None) is not None:
            raise RuntimeError("this step has been captured into a hipGraph: use replay() (an eager step would reuse the static mask "
                               "shares and put the optimizer's device-side replay count out of step with the host's)")
        x = {t: v for t, v in tasks_dict.items() if t in self.in_domains}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.autocast):
            out = self.model(x, task_masks=task_masks, num_encoded_tokens=self.N, alphas=self.alphas,
                             sample_tasks_uniformly=self.uniform)
            task_losses, loss_contra, loss = step_losses(out, tasks_dict, out[1], self.patch, self.loss_fns,
                                                         self.contra_weight, self.contra, self.balancer, self.contra_loss)
        self.opt.zero_grad(set_to_none=True)               # (FlatAdamW: also clears the flat gradient buffer)
        if self.balancer_opt is not None:
            self.balancer_opt.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.prepare()
        loss.backward()
        ops.join_wgrad_stream()
        if self.reducer is not None:
            self.reducer.finish()
            if self.balancer_opt is not None and torch.distributed.is_initialized() and \
                    torch.distributed.get_world_size(getattr(self.reducer, "group", None)) > 1:
                grp = getattr(self.reducer, "group", None)       # the reducer's process group, not the default one
                for g in self.balancer_opt.param_groups:       # a handful of scalars: one tiny all-reduce each
                    for q in g["params"]:
                        if q.grad is not None:
                            torch.distributed.all_reduce(q.grad, group=grp)
                            q.grad.div_(torch.distributed.get_world_size(grp))
        self._optimizer_step()
        return {'loss': loss.detach(), 'loss_contra': loss_contra.detach(),
                **{k + '_loss': v.detach() for k, v in task_losses.items()}}
