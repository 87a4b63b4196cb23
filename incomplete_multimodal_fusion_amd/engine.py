"""Flat-buffer optimizer engine: fp32 master weights + bf16 shadow + fused AdamW (csrc/optim.hip).

Stands in for the optimizer shell of the reference step (pretraining/utils/optim_factory.py:136-179 -- AdamW over all
parameters, weight decay applied to every group on the dict path, lr_scale 1; pretraining/utils/native_scaler.py:20-62).

Layout for MI355X: ONE contiguous fp32 buffer for all trainable weights (parameters become views of it, so
state_dict()/load_state_dict()/named_parameters() are unchanged), one for gradients (Linear weight gradients are written
into it directly by the split-K GEMM, the rest is copied in by post-accumulate-grad hooks), two for the moments, and a
bf16 shadow that the update kernel refreshes in the same pass -- the forward GEMMs read the shadow, so no per-parameter
cast kernels remain.  Parameters that never receive a gradient stay outside (torch.optim skips them as well).
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

    def zero_grad(self, set_to_none: bool = True):
        self._pending.clear()
        for p in self.params:
            p.grad = None
        self.grads.zero_()

    def grad_norm(self) -> torch.Tensor:
        self.flush()
        call("mmae_grad_norm", self.n, ptr(self.grads), ptr(self._ws), ptr(self._norm), stream())
        return self._norm[0]

    # -- update ------------------------------------------------------------------------------------------------------
    def refresh_shadow(self):
        call("mmae_shadow_bf16", self.n, ptr(self.master), ptr(self.shadow), stream())
        self._refresh_transposed()

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

    def step(self, grad_scale: float = 1.0):
        g = self.param_groups[0]
        self.flush()
        self.steps += 1
        call("mmae_adamw_step", self.n, ptr(self.master), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             ptr(self.shadow), float(g["lr"]) * float(g.get("lr_scale", 1.0)), self.betas[0], self.betas[1], self.eps,
             float(g["weight_decay"]), self.steps, float(grad_scale), stream())
        self._refresh_transposed()

    # -- checkpoint shell (moments per parameter name are produced by the caller from these flat views) ---------------
    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.param_groups[0]["lr"], "weight_decay": self.param_groups[0]["weight_decay"]}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.refresh_shadow()


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
