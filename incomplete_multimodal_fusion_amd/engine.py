"""Flat-buffer optimizer engine: fp32 master weights + bf16 shadow + fused AdamW (csrc/optim.hip).

Stands in for the optimizer shell of the reference step (pretraining/utils/optim_factory.py:136-179 -- AdamW over all
parameters, weight decay applied to every group on the dict path, lr_scale 1; pretraining/utils/native_scaler.py:20-62).

Layout for MI355X: ONE contiguous fp32 buffer for all trainable weights (parameters become views of it, so
state_dict()/load_state_dict()/named_parameters() are unchanged), one for gradients (Linear weight gradients are written
into it directly by the split-K GEMM, the rest is copied in by post-accumulate-grad hooks), two for the moments, and a
bf16 shadow that the update kernel refreshes in the same pass -- the forward GEMMs read the shadow, so no per-parameter
cast kernels remain.  Parameters that never receive a gradient stay outside (torch.optim skips them as well).

torch.optim.AdamW semantics kept exactly: a parameter whose .grad is None at step() is not touched (no weight decay, no
moment decay, its own step count does not advance -- FlatAdamW.step updates only the flat ranges that received a
gradient); clipping / skipping follow native_scaler.py:20-40 but stay on the device (mmae_adamw_control).  One
restriction, enforced loudly: ONE backward per zero_grad() -- weight gradients that producers write straight into the flat
buffer are assigned, not accumulated (the reference never accumulates either: pretrain_mmae.py:510-513).
"""
import ctypes
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream

_ALIGN = 8          # elements: keeps every parameter 16-byte aligned in both the fp32 and the bf16 buffer


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, exclude: Iterable[torch.nn.Parameter] = ()):
        ex = {id(p) for p in exclude}
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and id(p) not in ex]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), \
            "FlatAdamW needs fp32 parameters on the GPU"
        dev = self.params[0].device
        self.offsets: Dict[int, int] = {}
        n = 0
        for p in self.params:
            self.offsets[id(p)] = n
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n = n
        self.master = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self._ws = torch.empty(2048, dtype=torch.float32, device=dev)
        self._norm = torch.empty(1, dtype=torch.float32, device=dev)
        # mmae_adamw_control: multiplier, skip, #skipped, norm | captured steps: [4] replays so far, [5] this replay's learning rate
        self._ctl = torch.zeros(8, dtype=torch.float32, device=dev)
        self._capturing = False            # step() is being captured into a hipGraph (begin_capture / end_capture)
        self._graph_lr = None              # (learning rate, weight decay) last written to _ctl[5:7]
        self._graph_params: List[torch.nn.Parameter] = []     # the parameters the captured step updates
        self._graph_base: Dict[int, int] = {}                 # their step counts as baked into the captured launches (count BEFORE the captured step)
        self._stepped_in_capture = False
        self._ctl_used = False
        self._pstep: Dict[int, int] = {id(p): 0 for p in self.params}    # per-parameter step counts, as torch keeps them
        self._in_place = set()             # ids of weights whose gradient a producer wrote in place since zero_grad()
        with torch.no_grad():
            for p in self.params:
                o = self.offsets[id(p)]
                view = self.master[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._mmae_shadow = self.shadow[o:o + p.numel()].view_as(p)
                p._mmae_grad = self.grads[o:o + p.numel()].view_as(p)
                p._mmae_flat = (self, o)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.steps = 0
        self._pending: Dict[int, torch.nn.Parameter] = {}
        self.ready_callbacks = []          # called with a parameter whose gradient was written in place (see below)
        # transposed bf16 shadows (W^T for the data-gradient GEMMs): registered lazily by shadow_t_of(), each at its
        # weight's own offset in a second bf16 buffer, refreshed by ONE batched-transpose launch after every update
        self.shadow_t: Optional[torch.Tensor] = None
        self._tr: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self._tr_tiles: Optional[torch.Tensor] = None
        # zero-padded copies of FeedForward weights whose width does not fit the own GEMM's tiles (padded_ff): refreshed with the shadows
        self._pad: Dict[Tuple[int, int], "PaddedFF"] = {}
        self._pad_tiles: Optional[torch.Tensor] = None
        self.refresh_shadow()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.param_groups = [{"params": self.params, "lr": lr, "weight_decay": weight_decay, "lr_scale": 1.0}]

    # -- gradients -------------------------------------------------------------------------------------------------
    def _on_grad(self, p: torch.nn.Parameter):
        # Gradients that were not written into the flat buffer by their producer (LayerNorm gammas, biases, embeddings,
        # multi-use weights: a few hundred small tensors) are moved there in batches by flush(), not one copy kernel each.
        g = p.grad
        if g is not None and g.data_ptr() != p._mmae_grad.data_ptr():
            self._pending[id(p)] = p

    def flush(self):
        """Move the pending gradients into the flat buffer (multi-tensor copy) and point .grad at their flat views.  Called
        by grad_norm() / step() and by the DP reducer before it all-reduces a range of the buffer."""
        if not self._pending:
            return
        ps = list(self._pending.values())
        self._pending.clear()
        with torch.no_grad():
            torch._foreach_copy_([p._mmae_grad for p in ps], [p.grad for p in ps])
        for p in ps:
            p.grad = p._mmae_grad

    def _flush_deferred(self):
        """Backstop: weight-gradient sums still queued by ops._wgrad (normally run and published by the end-of-backward callback,
        which autograd skips when backward() raised) are summed and published before the gradients are read."""
        from . import ops
        ops.flush_splitk()

    def zero_grad(self, set_to_none: bool = True):
        from . import ops
        ops.reset_splitk()                 # deferred split-K sums a FAILED backward left queued must not be summed into this pass's buffer
        self._pending.clear()
        self._in_place.clear()
        for p in self.params:
            p.grad = None
        self.grads.zero_()

    def grad_norm(self) -> torch.Tensor:
        """L2 norm over the flat gradient buffer (a parameter without a gradient contributes its zero-filled range, like
        get_grad_norm_ skipping it, native_scaler.py:49-62).  Device scalar, no host sync."""
        self._flush_deferred()
        self.flush()
        call("mmae_grad_norm", self.n, ptr(self.grads), ptr(self._ws), ptr(self._norm), stream())
        return self._norm[0]

    # -- update ------------------------------------------------------------------------------------------------------
    def refresh_shadow(self):
        call("mmae_shadow_bf16", self.n, ptr(self.master), ptr(self.shadow), stream())
        self._refresh_derived()

    def padded_ff(self, w1: torch.nn.Parameter, w2: torch.nn.Parameter) -> Optional["PaddedFF"]:
        """bf16 copies of a FeedForward's weights W1 (2F, D) / W2 (D, F) with the GEGLU width F zero-padded to Fp = the next multiple of
        256, plus their transposes -- the operand layouts csrc/gemm.hip needs when F fits none of its tiles (ViT-L: F = 2730 -> 2816):
            w1p (2 Fp, D): val rows [0, F), gate rows [Fp, Fp + F)      w1pt (D, 2 Fp) = w1p^T
            w2p (D, Fp)                                                  w2pt (Fp, D)  = w2p^T
        Pads are zero and never written (mathematically inert: h, g, dg, dh are zero in the pad columns).  Registered on first use,
        refreshed after every update by ONE launch (mmae_pad_copy_bf16_batched).  None if the weights are not both held by this engine."""
        if id(w1) not in self.offsets or id(w2) not in self.offsets or w1.dim() != 2 or w2.dim() != 2:
            return None
        key = (self.offsets[id(w1)], self.offsets[id(w2)])
        pf = self._pad.get(key)
        if pf is not None:
            return pf
        D, F = w2.shape
        if w1.shape != (2 * F, D):
            return None
        pf = PaddedFF(self, key[0], key[1], D, F)
        self._pad[key] = pf
        self._pad_tiles = None
        call("mmae_pad_copy_bf16_batched", ptr(pf.tiles), pf.tiles.shape[0], stream())      # first fill (later: with all the others)
        return pf

    def _refresh_padded(self):
        if not self._pad:
            return
        if self._pad_tiles is None:
            self._pad_tiles = torch.cat([pf.tiles for pf in self._pad.values()], 0).contiguous()
        call("mmae_pad_copy_bf16_batched", ptr(self._pad_tiles), self._pad_tiles.shape[0], stream())

    def transposed_shadow(self, o0: int, R: int, C: int) -> Optional[torch.Tensor]:
        """(C, R) bf16 view holding the transpose of the (R, C) weight block at flat offset o0; None if the block cannot
        be registered (odd sizes, or it overlaps a differently shaped registered block)."""
        key = (o0, R, C)
        v = self._tr.get(key)
        if v is not None:
            return v
        if R % 8 or C % 8 or o0 % 8:
            return None
        for (o, r, c) in self._tr:
            if o < o0 + R * C and o0 < o + r * c:
                return None
        if self.shadow_t is None:
            self.shadow_t = torch.empty(self.n, dtype=torch.bfloat16, device=self.master.device)
        v = self.shadow_t[o0:o0 + R * C].view(C, R)
        with torch.no_grad():
            v.copy_(self.shadow[o0:o0 + R * C].view(R, C).t())
        self._tr[key] = v
        self._tr_tiles = None
        return v

    def _refresh_transposed(self):
        if not self._tr:
            return
        if self._tr_tiles is None:
            import numpy as np
            rows = []
            for (o0, R, C) in self._tr:
                for r in range(0, R, 64):
                    for c in range(0, C, 64):
                        rows.append((o0 + r * C + c, o0 + c * R + r, C, R, min(64, R - r), min(64, C - c)))
            tab = np.zeros(len(rows), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("ld_src", "<i4"), ("ld_dst", "<i4"),
                                                      ("nr", "<i4"), ("nc", "<i4")]))
            for i, t in enumerate(rows):
                tab[i] = t
            self._tr_tiles = torch.from_numpy(tab.view(np.uint8).reshape(-1, 32).copy()).to(self.master.device)
        call("mmae_transpose_bf16_batched", ptr(self.shadow), ptr(self.shadow_t), ptr(self._tr_tiles),
             self._tr_tiles.shape[0], stream())

    def _refresh_derived(self):
        """Everything derived from the bf16 shadow, after an update: transposed shadows, padded FeedForward copies."""
        self._refresh_transposed()
        self._refresh_padded()

    def _update_ranges(self):
        """[(flat offset, length, step count)] of the runs of parameters that hold a gradient (torch.optim.AdamW skips
        `p.grad is None`), each parameter's own step count advanced; one run covering everything in the usual case."""
        runs, cur = [], None
        for p in self.params:
            if p.grad is None:
                cur = None
                continue
            st = self._pstep[id(p)] = self._pstep[id(p)] + 1
            o = self.offsets[id(p)]
            end = o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if cur is not None and cur[2] == st and cur[0] + cur[1] == o:
                cur[1] = end - cur[0]
            else:
                cur = [o, end - o, st]
                runs.append(cur)
        return runs

    def step(self, grad_scale: float = 1.0, clip_grad: Optional[float] = None, skip_grad: Optional[float] = None,
             check_finite: bool = False):
        """AdamW update of every parameter that received a gradient since zero_grad().
        clip_grad: global-norm clipping (torch.nn.utils.clip_grad_norm_); skip_grad: no update when the norm reaches it
        (native_scaler.py:27-32); check_finite: no update when the norm is inf / NaN (what torch's GradScaler does for the
        reference).  All three are decided on the device from mmae_grad_norm's result -- no host synchronisation; see
        last_step_skipped() / last_grad_norm() to read the outcome."""
        g = self.param_groups[0]
        self._flush_deferred()
        self.flush()
        self.steps += 1
        self._stepped_in_capture = self._capturing
        ctl = clip_grad is not None or skip_grad is not None or check_finite or self._capturing
        if self._capturing:
            # the launches below are replayed with THESE by-value arguments: the replay count the device keeps is added to the
            # captured step counts (hence the - 1 below: the first replay is the step this call would have been)
            call("mmae_adamw_tick", ptr(self._ctl), stream())
        if not ctl and self._ctl_used:
            # once a controlled step has run, the device holds the count of skipped steps the bias correction needs: an
            # uncontrolled step afterwards goes through the same kernel (no threshold, no guard) instead of using a host count
            # that may be too high by the skipped steps
            ctl = True
        if ctl:
            self._ctl_used = True
            call("mmae_grad_norm", self.n, ptr(self.grads), ptr(self._ws), ptr(self._norm), stream())
            # the reference clips OR skips (native_scaler.py:24-32: `if clip_grad ... elif skip_grad`): with a clip norm the skip
            # threshold is not applied; the non-finite guard follows check_finite
            thr = 0.0 if clip_grad is not None else float(skip_grad or 0.0)
            call("mmae_adamw_control", ptr(self._norm), float(clip_grad or 0.0), thr, float(grad_scale), 1 if check_finite else 0,
                 ptr(self._ctl), stream())
        lr = float(g["lr"]) * float(g.get("lr_scale", 1.0))
        for o, n, st in self._update_ranges():
            sl = slice(o, o + n)
            if ctl:
                call("mmae_adamw_step_ctl", n, ptr(self.master[sl]), ptr(self.grads[sl]), ptr(self.exp_avg[sl]),
                     ptr(self.exp_avg_sq[sl]), ptr(self.shadow[sl]), lr, self.betas[0], self.betas[1], self.eps,
                     float(g["weight_decay"]), st - 1 if self._capturing else st, ptr(self._ctl), stream())
            else:
                call("mmae_adamw_step", n, ptr(self.master[sl]), ptr(self.grads[sl]), ptr(self.exp_avg[sl]),
                     ptr(self.exp_avg_sq[sl]), ptr(self.shadow[sl]), lr, self.betas[0], self.betas[1], self.eps,
                     float(g["weight_decay"]), st, float(grad_scale), stream())
        if ctl:
            absent = [p for p in self.params if p.grad is None]
            if self._capturing:
                # a graph is static: the parameters without a gradient now have none in any replay and are never updated (what
                # the eager step does for them, step by step); their step counts stay where they are
                self._graph_params = [p for p in self.params if p.grad is not None]
                self._graph_base = {id(p): self._pstep[id(p)] - 1 for p in self._graph_params}
            elif absent and self.last_step_skipped():
                # the kernels bias-correct with (host count - skipped steps so far); a parameter that sat this skipped step
                # out must not lose a step for it.  Host sync, only when the set of used parameters varies (downstream).
                for p in absent:
                    self._pstep[id(p)] += 1
        self._refresh_derived()

    # -- hipGraph capture of the step (pretrain.PretrainStep.capture) ---------------------------------------------------
    def begin_capture(self):
        """The following step() is being captured: it always takes the device-controlled path and reads its step counts, learning
        rate and weight decay through the control block (the captured launches keep their by-value arguments).  A graph is static:
        the parameters without a gradient at capture time are never updated by the replays.  After end_capture() use
        replay_begin() / replay_end() around each graph replay; eager step() calls must not be mixed in any more (the device's
        replay count would no longer match the host's step count)."""
        self._capturing = True
        self._stepped_in_capture = False
        self._ctl[4:8].zero_()
        self._graph_lr = None

    def end_capture(self):
        """The captured step() only recorded launches: take back the host-side counts it advanced -- if it ran at all (a capture
        that raised before reaching step() has advanced nothing)."""
        self._capturing = False
        if not self._stepped_in_capture:
            return
        self._stepped_in_capture = False
        self.steps -= 1
        for p in self._graph_params:
            self._pstep[id(p)] -= 1

    def abort_capture(self):
        """A capture that raised (pretrain.PretrainStep.capture's except path), possibly AFTER the captured step() was recorded: no
        graph exists, so nothing of the capture may survive -- end_capture() has already taken the host counts back; forget the
        captured parameter set and its baked-in counts and zero the replay counter / hyper-parameter slots of the control block
        (the eager controlled kernel adds ctl[4] to its step count: a stale non-zero value would bias-correct every later eager
        step wrongly, and a later load_state_dict would try to resume into a captured step that does not exist)."""
        if self._capturing:
            self.end_capture()
        self._graph_params = []
        self._graph_base = {}
        self._graph_lr = None
        self._ctl[4:8].zero_()

    def replay_begin(self):
        """Before a graph replay: hand the current learning rate / weight decay to the device (one tiny copy, only when they changed)."""
        g = self.param_groups[0]
        hyper = (float(g["lr"]) * float(g.get("lr_scale", 1.0)), float(g["weight_decay"]))
        if hyper != self._graph_lr:
            self._ctl[5:8].copy_(torch.tensor([hyper[0], hyper[1], 1.0], dtype=torch.float32).pin_memory(), non_blocking=True)
            self._graph_lr = hyper

    def replay_end(self):
        """After a graph replay: the host's view of the step counts."""
        self.steps += 1
        for p in self._graph_params:
            self._pstep[id(p)] += 1

    def skip_flag(self) -> Optional[torch.Tensor]:
        """Device bool scalar: did the device-side control skip the most recent step()?  None when that step was not a controlled
        one (nothing can have skipped it).  No host sync -- for companions that must follow the decision (PretrainStep's loss
        balancer optimizer: the reference holds model and balancer in ONE optimizer, a skipped step skips both)."""
        return (self._ctl[1] != 0.0) if self._ctl_used else None

    def last_step_skipped(self) -> bool:
        """Host sync: did the device-side control skip the most recent controlled step()?"""
        return bool(self._ctl[1].item() != 0.0)

    def last_grad_norm(self) -> torch.Tensor:
        """Device scalar: the unscaled norm the most recent controlled step() saw (native_scaler's return value)."""
        return self._ctl[3]

    def skipped_steps(self) -> int:
        """Host sync: number of controlled steps skipped so far (they do not count towards the bias correction)."""
        return int(self._ctl[2].item())

    def param_step(self, p) -> int:
        """torch.optim.AdamW's state[p]['step'] for this parameter (host sync through skipped_steps())."""
        return max(0, self._pstep[id(p)] - self.skipped_steps())

    # -- checkpoint shell (moments per parameter name are produced by the caller from these flat views) ---------------
    def state_dict(self):
        sk = self.skipped_steps()
        return {"steps": self.steps - sk, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_steps": [max(0, self._pstep[id(p)] - sk) for p in self.params],
                "lr": self.param_groups[0]["lr"], "weight_decay": self.param_groups[0]["weight_decay"]}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        ps = sd.get("param_steps") or [self.steps] * len(self.params)
        self._pstep = {id(p): int(st) for p, st in zip(self.params, ps)}
        self._counts_restored()
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.refresh_shadow()

    def _counts_restored(self):
        """The host's step counts were replaced (checkpoint resume): the device's control block must describe the SAME counts.
        Multiplier / skip flag / skipped-step count / norm start over (the restored counts are net of skipped steps).  A captured
        step keeps its by-value step counts (those of capture time): the replay counter the kernels add to them is set so that
        the next replay bias-corrects with restored count + 1, and the learning rate / weight decay are handed over again."""
        self._ctl[0:4].zero_()
        if self._graph_params:
            d = {self._pstep[id(p)] - self._graph_base[id(p)] for p in self._graph_params}
            if len(d) != 1:
                raise RuntimeError("resuming into a captured step: the restored per-parameter step counts are not a uniform shift of the "
                                   "captured ones -- capture() the step again after loading")
            self._ctl[4:5].fill_(float(d.pop()))
            self._graph_lr = None

    def set_param_steps(self, steps: Dict[int, int]):
        """{id(parameter): step count}: restores torch-layout optimizer state (checkpoint.load_optimizer_state_dict)."""
        for p in self.params:
            self._pstep[id(p)] = int(steps.get(id(p), 0))
        self.steps = max(self._pstep.values()) if self._pstep else 0
        self._counts_restored()


class PaddedFF:
    """Zero-padded bf16 operand copies of one FeedForward (FlatAdamW.padded_ff) and the tile table that refreshes them."""

    def __init__(self, eng: "FlatAdamW", o1: int, o2: int, D: int, F: int):
        import numpy as np
        dev = eng.master.device
        Fp = (F + 255) // 256 * 256
        self.F, self.Fp, self.D = F, Fp, D
        self.w1p = torch.zeros(2 * Fp, D, dtype=torch.bfloat16, device=dev)
        self.w2p = torch.zeros(D, Fp, dtype=torch.bfloat16, device=dev)
        self.w1pt = torch.zeros(D, 2 * Fp, dtype=torch.bfloat16, device=dev)
        self.w2pt = torch.zeros(Fp, D, dtype=torch.bfloat16, device=dev)
        s1 = eng.shadow.data_ptr() + 2 * o1          # W1 (2F, D) row-major
        s2 = eng.shadow.data_ptr() + 2 * o2          # W2 (D, F)  row-major
        rows = []

        def grid(nr_all, nc_all):
            r = np.arange(0, nr_all, 64)[:, None].repeat((nc_all + 63) // 64, 1).reshape(-1)
            c = np.tile(np.arange(0, nc_all, 64), (nr_all + 63) // 64)
            return r, c, np.minimum(64, nr_all - r), np.minimum(64, nc_all - c)

        def add(src, dst, ld_src, ld_dst, nr, nc, tr):
            t = np.zeros(len(nr), dtype=_PAD_TILE)
            t["src"], t["dst"], t["ld_src"], t["ld_dst"], t["nr"], t["nc"], t["tr"] = src, dst, ld_src, ld_dst, nr, nc, tr
            rows.append(t)
        r, c, nr, nc = grid(F, D)
        for r_src0, r_dst0 in ((0, 0), (F, Fp)):                  # val block, gate block of W1
            src = s1 + 2 * ((r_src0 + r) * D + c)
            add(src, self.w1p.data_ptr() + 2 * ((r_dst0 + r) * D + c), D, D, nr, nc, 0)
            add(src, self.w1pt.data_ptr() + 2 * (c * (2 * Fp) + r_dst0 + r), D, 2 * Fp, nr, nc, 1)
        r, c, nr, nc = grid(D, F)
        src = s2 + 2 * (r * F + c)
        add(src, self.w2p.data_ptr() + 2 * (r * Fp + c), F, Fp, nr, nc, 0)
        add(src, self.w2pt.data_ptr() + 2 * (c * D + r), F, D, nr, nc, 1)
        tab = np.concatenate(rows)
        self.tiles = torch.from_numpy(tab.view(np.uint8).reshape(-1, 48).copy()).to(dev)


try:
    import numpy as _np
    _PAD_TILE = _np.dtype([("src", "<u8"), ("dst", "<u8"), ("ld_src", "<i4"), ("ld_dst", "<i4"), ("nr", "<i4"), ("nc", "<i4"),
                           ("tr", "<i4"), ("pad_", "<i4"), ("pad2_", "<i8")])
    assert _PAD_TILE.itemsize == 48
except ImportError:                                   # numpy is a hard dependency of torch anyway
    _PAD_TILE = None


def padded_ff_of(w1, w2, dtype) -> Optional[PaddedFF]:
    """The engine's zero-padded operand copies of FeedForward weights (w1 (2F, D), w2 (D, F)), or None (no engine / not bf16)."""
    if dtype != torch.bfloat16 or not hasattr(w1, "_mmae_flat") or not hasattr(w2, "_mmae_flat"):
        return None
    eng = w1._mmae_flat[0]
    if w2._mmae_flat[0] is not eng:
        return None
    return eng.padded_ff(w1, w2)


def shadow_of(ws, dtype) -> Optional[torch.Tensor]:
    """The bf16 shadow of one weight or of several weights that sit back to back in the flat buffer (row-concatenated
    view, e.g. [to_q.weight, to_kv.weight]); None when unavailable."""
    if dtype != torch.bfloat16 or not all(hasattr(w, "_mmae_flat") for w in ws):
        return None
    eng, o0 = ws[0]._mmae_flat
    o = o0
    for w in ws:
        if w._mmae_flat[0] is not eng or w._mmae_flat[1] != o or w.numel() % _ALIGN:
            return None
        o += w.numel()
    return eng.shadow[o0:o].view(sum(w.shape[0] for w in ws), ws[0].shape[1])


def shadow_t_of(ws, dtype) -> Optional[torch.Tensor]:
    """(K, sum N_i) bf16 transpose of the row-concatenated weights `ws` (kept fresh by the engine), or None."""
    if dtype != torch.bfloat16 or not all(hasattr(w, "_mmae_flat") for w in ws):
        return None
    eng, o0 = ws[0]._mmae_flat
    o = o0
    for w in ws:
        if w._mmae_flat[0] is not eng or w._mmae_flat[1] != o or w.numel() % _ALIGN or w.dim() != 2:
            return None
        o += w.numel()
    return eng.transposed_shadow(o0, sum(w.shape[0] for w in ws), ws[0].shape[1])


def grads_written_in_place(ws) -> None:
    """The producer wrote the fp32 gradient of `ws` straight into the flat buffer (ops._wgrad with grad_view_of): publish
    it WITHOUT going through autograd's AccumulateGrad -- a returned view would be cloned there and copied back at the
    next flush (two extra passes over every large weight).  Sets .grad to the flat view and runs the ready callbacks
    (the DP reducer's bucket accounting) that post-accumulate hooks would have run."""
    for w in ws:
        eng = w._mmae_flat[0]
        if id(w) in eng._in_place:
            raise RuntimeError("a second backward() wrote the in-place weight gradient of a parameter before zero_grad(): "
                               "gradient accumulation is not supported by the flat engine (one backward per zero_grad)")
        eng._in_place.add(id(w))
        eng._pending.pop(id(w), None)
        w.grad = w._mmae_grad
        for cb in eng.ready_callbacks:
            cb(w)


def grad_view_of(ws) -> Optional[torch.Tensor]:
    """fp32 gradient destination for the same (possibly concatenated) weights, or None."""
    if not all(hasattr(w, "_mmae_flat") for w in ws):
        return None
    eng, o0 = ws[0]._mmae_flat
    o = o0
    for w in ws:
        if w._mmae_flat[0] is not eng or w._mmae_flat[1] != o or w.numel() % _ALIGN:
            return None
        o += w.numel()
    return eng.grads[o0:o].view(sum(w.shape[0] for w in ws), ws[0].shape[1])
