"""torch.autograd wrappers around the C ABI (include/mmae_hip.h).  Device tensors only; no fallback.

Every Function here launches hand-written gfx950 kernels through ctypes on torch's current HIP stream.  Dense
projections between them are plain torch matmuls (hipBLASLt) chosen by the callers in multimae/.
"""
import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import call, dt, ptr, stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _rowmat(t, cols_needed):
    """A 2-D (rows, >=cols) view with unit column stride; returns (tensor, row_stride)."""
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor"
    return t, t.stride(0)


# ------------------------------------------------------------------------------------------------ kernel timing hook
class _NoAutocast:
    """`with torch.autocast("cuda", enabled=False)` for the library calls inside the autograd functions (their operands are already in
    the compute dtype), without torch.autocast's argument checking and nesting bookkeeping: ~0.5 us instead of ~6 us, 200+ times a step.
    Only ever entered inside an enclosing autocast region or none, so the weight-cast cache is left alone."""
    __slots__ = ("prev",)

    def __enter__(self):
        self.prev = torch.is_autocast_enabled("cuda")
        torch.set_autocast_enabled("cuda", False)

    def __exit__(self, *exc):
        torch.set_autocast_enabled("cuda", self.prev)
        return False


class KernelTimer:
    """HIP-event timing of one C-ABI entry point on the stream it is launched on (bench.py's roofline leg).
    `flops` accumulates the algorithmic work of the timed launches (FLOPs or bytes, a device scalar or a python number)."""

    def __init__(self, name: str):
        self.name, self.pairs, self.flops, self.kernels = name, [], None, set()      # kernels: device kernel names the launches ran
        self.seen, self.every = 0, 1

    def sample(self) -> bool:
        """True for every `every`-th call (default: each one): a bracket costs the stream ~5 us of idle time per event, so a
        timer on an entry point with many launches per step brackets a rotating subset."""
        self.seen += 1
        return self.seen % self.every == 0

    def bracket(self):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.append((a, b))
        return a, b

    def add_flops(self, f: torch.Tensor):
        self.flops = f if self.flops is None else self.flops + f

    def summary(self):
        torch.cuda.synchronize()
        if not self.pairs:
            return 0.0, 0, 0.0
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        work = 0.0 if self.flops is None else float(self.flops.item() if torch.is_tensor(self.flops) else self.flops)
        return sum(ms) / len(ms), len(ms), work


_TIMER = None          # attention forward
_TIMERS = {}           # other entry points by name


def set_kernel_timer(t):
    """t: one KernelTimer, a list of them, or None."""
    global _TIMER, _TIMERS
    ts = [] if t is None else (list(t) if isinstance(t, (list, tuple)) else [t])
    _TIMER = next((x for x in ts if x.name == "mmae_mha_fwd"), None)
    _TIMERS = {x.name: x for x in ts if x.name != "mmae_mha_fwd"}


class LayerTimer:
    """HIP-event brackets around ONE encoder layer (Block_Fusion + Block), forward and backward, on the stream the layer runs on
    (bench.py's roofline_block: the quantity north_star sets its MFMA target on).  The model calls layer_mark() at the layer's two
    boundaries while `model.layer_timer` is set; every boundary is an identity autograd node over the whole layer-crossing state
    (both residual parts + the pending delta), so its backward runs exactly when the layer's backward begins / has ended."""

    def __init__(self, layer: int):
        self.layer, self.ev = int(layer), {"f0": [], "f1": [], "b1": [], "b0": []}

    def mark(self, key):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev[key].append(e)

    def summary(self):
        """-> (mean forward ms, mean backward ms, steps)"""
        torch.cuda.synchronize()
        n = min(len(v) for v in self.ev.values())
        if n == 0:
            return 0.0, 0.0, 0
        fw = sum(a.elapsed_time(b) for a, b in zip(self.ev["f0"][:n], self.ev["f1"][:n])) / n
        bw = sum(a.elapsed_time(b) for a, b in zip(self.ev["b1"][:n], self.ev["b0"][:n])) / n
        return fw, bw, n


class _LayerMark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, timer, which, *ts):
        ctx.timer, ctx.which = timer, which
        ctx.set_materialize_grads(False)
        timer.mark("f%d" % which)
        return tuple(t.view_as(t) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        ctx.timer.mark("b%d" % ctx.which)
        return (None, None, *gs)


def layer_mark(timer, which, *ts):
    """which 0: the layer's entry, 1: its exit.  Returns the tensors (as views) to be used in place of `ts`."""
    return _LayerMark.apply(timer, which, *ts)


# ------------------------------------------------------------------------------------------------ segments
class Segments:
    """Device-side segment descriptors for the masked attention: int32 (B, nseg) start rows and lengths."""

    def __init__(self, start: torch.Tensor, length: torch.Tensor, max_rows: int, covers_all: bool = True):
        """covers_all: every row of the operand belongs to exactly one segment (true for the packed row space of the
        model).  With False the attention outputs / gradients are zero-initialised so rows outside every segment read 0."""
        assert start.dtype == torch.int32 and length.dtype == torch.int32 and start.shape == length.shape
        self.start, self.length, self.max_rows = _c(start), _c(length), int(max_rows)
        self.B, self.nseg = start.shape
        self.covers_all = bool(covers_all)

    @staticmethod
    def dense(B: int, n: int, device, nseg: int = 1, row0: int = 0):
        """Every sample: one attend-all segment of n rows (rows b*n ...)."""
        start = torch.zeros(B, nseg, dtype=torch.int32, device=device)
        length = torch.zeros(B, nseg, dtype=torch.int32, device=device)
        start[:, nseg - 1] = torch.arange(B, dtype=torch.int32, device=device) * n + row0
        length[:, nseg - 1] = n
        return Segments(start, length, n)

    @staticmethod
    def from_types(types: Sequence[int], B: int, nseg: int, device, per_sample_rows: Optional[int] = None):
        """Host helper (tests / standalone module use): token types of ONE sample, sorted ascending, shared by the
        batch; rows of sample b start at b * per_sample_rows."""
        n = len(types)
        per = n if per_sample_rows is None else per_sample_rows
        assert list(types) == sorted(types), "tokens must be grouped by type in ascending type order"
        lens = [sum(1 for t in types if t == s) for s in range(nseg)]
        starts = [sum(lens[:s]) for s in range(nseg)]
        st = torch.tensor([[b * per + s0 for s0 in starts] for b in range(B)], dtype=torch.int32, device=device)
        ln = torch.tensor([lens for _ in range(B)], dtype=torch.int32, device=device)
        return Segments(st, ln, n)


# ------------------------------------------------------------------------------------------------ dense projections
def _split_k(rows: int, n_out: int, n_in: int) -> int:
    """Split factor for the weight-gradient GEMM dW = g^T x (n_out x n_in output, contraction over `rows`).
    The output has few 256x256 tiles (6..48 at ViT-B) while the contraction is 65k..165k long, so a single GEMM leaves
    most of the 256 CUs idle (measured 290-830 TFLOP/s).  Splitting the contraction into S batched GEMMs + an fp32 sum
    fills the chip: largest power of two with tiles * S <= 256, S <= 32 (tools/probes/wgrad_splitk_probe.py: 1.3-3.2x)."""
    tiles = ((n_out + 255) // 256) * ((n_in + 255) // 256)
    s = 1
    while s < _SPLITK_MAX and tiles * s * 2 <= _SPLITK_CAP and rows % (s * 2) == 0 and rows // (s * 2) >= 2048:
        s *= 2
    return s


_SPLITK_CAP = 256     # workgroups the split aims at (256 CUs); tools/tuning_env.py overrides these module attributes for A/B runs
_SPLITK_MAX = 32
_WGRAD_STREAMS = {}
WGRAD_STREAM_PRIORITY = -1   # high priority: own HW queue
WGRAD_SIDE_STREAM = True     # weight-gradient GEMMs of single-use weights run on a side HIP stream (see _Linear.backward)


def wgrad_stream(device=None):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _WGRAD_STREAMS.get(dev)
    if s is None:
        s = _WGRAD_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=WGRAD_STREAM_PRIORITY)
    return s


def join_wgrad_stream():
    """Make the current stream wait for every weight-gradient GEMM issued so far (call before grads are consumed)."""
    s = _WGRAD_STREAMS.get(torch.cuda.current_device())
    if s is not None:
        torch.cuda.current_stream().wait_stream(s)


# ------------------------------------------------------------------------------------------------ own GEMM (csrc/gemm.hip)
OWN_GEMM = 1          # bit 0: forward / input-gradient projections of supported shapes run the hand-written persistent 256x256x64 kernel
                      # (FeedForward[1] + GEGLU its fused-epilogue form): faster than the tuned library GEMM on 16 of 17 shapes
                      # (tools/bench_gemm.py), -1.9 ms per step.  bit 1: weight gradients on its transposing split-K form (mmae_gemm_tn):
                      # correct and tested, but 5-30 % SLOWER than the library's split-K batched GEMM (tools/bench_wgrad.py; the loop is
                      # bound by the latency of its operand stream: 1546 TFLOP/s with every DMA hitting L2, 1040 from HBM at a 75 % L2 hit
                      # rate, no change from 3 to 5 staging steps in flight), +4.7 ms in the step -> OFF by default.
                      # 0: library GEMMs everywhere (tools/tuning_env.py: MMAE_OWN_GEMM)
_OWN_GEMM_MIN_TILES = 512     # from two 256 x 256 output tiles per CU on; below that the library's smaller tiles fill the chip better ...
                              # ... except where the output is at least two tiles WIDE (N >= 512): from HALF of it (one tile per CU) on.  Round 6, same-box A/Bs: at the
                              # reference's per-GPU batch (B = 64: 40 960 rows) a plain 512 sent every N = 768 / 512 projection to the library (44.94 / 45.01 ms
                              # per step against 44.80 / 44.90 with them on the own kernel); a plain 256 also took the one-tile-wide N = 256 projections of
                              # the 65 536 fusion rows at B = 256, ONE tile per persistent workgroup with nothing to overlap its prologue and epilogue with
                              # (150.27 -> 150.63 ms per step, three alternations)
CALLS = {"mmae_gemm_nt": 0, "mmae_gemm_geglu": 0, "mmae_gemm_tn": 0,      # launches of the own GEMM entry points (tests assert engagement)
         "library_matmul_nt": 0}                                          # ... and the projections matmul_nt handed to the library instead


_WARNED = set()


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import sys
        print("[mmae] " + msg, file=sys.stderr, flush=True)


def _offset_limit_note(kind, M, N, K, ld_in, ld_out):
    """The own GEMM addresses its operands with 31-bit byte offsets (csrc/gemm.hip gemm_shape_ok); a projection that is big enough for
    it but exceeds that range falls back to the library GEMM -- say so once per shape instead of silently (a slower path, not an error)."""
    mpad = (M + 255) // 256 * 256
    if mpad * ld_in * 2 >= (1 << 31) or mpad * ld_out * 2 >= (1 << 31):
        _warn_once((kind, M, N, K), "%s M=%d N=%d K=%d: operand larger than the own GEMM's 31-bit byte offsets -> library GEMM for this "
                   "shape (smaller per-GPU batch or row chunks keep it on the own kernel)" % (kind, M, N, K))


def own_gemm_ok(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> bool:
    """Can y = x @ w^T run on mmae_gemm_nt?  x (M, K), w (N, K): bf16, unit column stride, 16-byte aligned bases, N % 256 == 0,
    K % 128 == 0, K >= 384, and enough tiles to fill the 256 persistent workgroups.  out: the destination the caller wants written
    (a row-slice view of a larger matrix, say): its row stride and base alignment are part of the answer."""
    if not (OWN_GEMM & 1) or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or x.dim() != 2 or w.dim() != 2 or not x.is_cuda:
        return False
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or x.stride(1) != 1 or w.stride(1) != 1 or x.data_ptr() % 16 or w.data_ptr() % 16:
        return False
    ldc = N
    if out is not None:
        if out.dtype != torch.bfloat16 or out.dim() != 2 or out.shape != (M, N) or out.stride(1) != 1 or out.data_ptr() % 8:
            return False
        ldc = out.stride(0)
    tiles = ((M + 255) // 256) * (N // 256)
    if tiles < _OWN_GEMM_MIN_TILES and not (N >= 512 and tiles >= _OWN_GEMM_MIN_TILES // 2):
        return False
    ok = bool(_lib.lib().mmae_gemm_nt_supported(M, N, K, x.stride(0), w.stride(0), ldc))
    if not ok and N % 256 == 0 and K % 128 == 0 and K >= 384:
        _offset_limit_note("gemm_nt", M, N, K, x.stride(0), ldc)
    return ok


def gemm_nt(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y (M, N) = x (M, K) @ w (N, K)^T on the own kernel (caller checked own_gemm_ok).  No autograd."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if out is None else out
    assert y.shape == (M, N) and y.stride(1) == 1 and y.dtype == torch.bfloat16
    tm = _TIMERS.get("mmae_gemm_nt")            # bench.py's roofline: every `every`-th launch is bracketed (a bracket costs ~5 us)
    if tm is not None and not tm.sample():
        tm = None
    if tm is not None:
        tm.add_flops(2.0 * M * N * K)
        ev0, ev1 = tm.bracket()
        ev0.record()
    CALLS["mmae_gemm_nt"] += 1
    call("mmae_gemm_nt", M, N, K, ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(y), y.stride(0), stream())
    if tm is not None:
        ev1.record()
    return y


def matmul_nt(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x @ w^T: the own kernel where it applies, the library GEMM (hipBLASLt through torch) otherwise."""
    if own_gemm_ok(x, w, out):
        return gemm_nt(x, w, out)
    CALLS["library_matmul_nt"] += 1
    if out is not None:
        return torch.mm(x, w.t(), out=out)
    return torch.nn.functional.linear(x, w)


def own_geglu_ok(y: torch.Tensor, w1: torch.Tensor, h: Optional[torch.Tensor] = None, g: Optional[torch.Tensor] = None) -> bool:
    if not (OWN_GEMM & 1) or y.dtype != torch.bfloat16 or w1.dtype != torch.bfloat16 or y.dim() != 2 or not y.is_cuda:
        return False
    M, K = y.shape
    F2 = w1.shape[0]
    if F2 % 2 or w1.shape[1] != K or y.stride(1) != 1 or w1.stride(1) != 1 or y.data_ptr() % 16 or w1.data_ptr() % 16:
        return False
    F = F2 // 2
    if ((M + 255) // 256) * (F // 128) < _OWN_GEMM_MIN_TILES:
        return False
    ldh, ldg = 2 * F, F
    for t, width in ((h, 2 * F), (g, F)):
        if t is not None and (t.dtype != torch.bfloat16 or t.shape != (M, width) or t.stride(1) != 1 or t.data_ptr() % 4):
            return False
    ldh = h.stride(0) if h is not None else ldh
    ldg = g.stride(0) if g is not None else ldg
    ok = bool(_lib.lib().mmae_gemm_geglu_supported(M, F, K, y.stride(0), w1.stride(0), ldh, ldg))
    if not ok and F % 128 == 0 and K % 128 == 0 and K >= 384:
        _offset_limit_note("gemm_geglu", M, F, K, y.stride(0), ldh)
    return ok


def gemm_geglu(y: torch.Tensor, w1: torch.Tensor, h: torch.Tensor, g: torch.Tensor):
    """h (M, 2F) = y @ w1^T and g (M, F) = gelu(h[:, F:]) * h[:, :F] in one kernel (FeedForward[1] + GEGLU, zorro_utils.py:115-126)."""
    M, K = y.shape
    F = w1.shape[0] // 2
    assert h.shape == (M, 2 * F) and g.shape == (M, F) and h.stride(1) == 1 and g.stride(1) == 1
    CALLS["mmae_gemm_geglu"] += 1
    call("mmae_gemm_geglu", M, F, K, ptr(y), y.stride(0), ptr(w1), w1.stride(0), ptr(h), h.stride(0), ptr(g), g.stride(0), stream())


class _Linear(torch.autograd.Function):
    """y = x @ cat(ws)^T (+ bias) on hipBLASLt/rocBLAS through torch.  `ws` are the fp32 master weights (cast to the
    compute dtype here); the weight gradient comes back in fp32 straight from a split-K batched GEMM."""

    @staticmethod
    def forward(ctx, x, bias, flags, *ws):
        side, once = flags
        from .engine import grad_view_of, shadow_of
        T = x.dtype
        w = shadow_of(ws, T)                      # bf16 shadow kept fresh by the fused optimizer: no cast kernel
        if w is None:
            w = ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)
            w = w if w.dtype == T else w.to(T)
        b = None
        if bias is not None:
            b = bias._mmae_shadow if (T == torch.bfloat16 and hasattr(bias, "_mmae_shadow")) else \
                (bias if bias.dtype == T else bias.to(T))
        with _NoAutocast():
            x2o = x.reshape(-1, x.shape[-1]) if b is None else None
            if x2o is not None and own_gemm_ok(x2o, w):
                y = gemm_nt(x2o, w).reshape(*x.shape[:-1], w.shape[0])
            else:
                y = torch.nn.functional.linear(x, w, b)
        ctx.save_for_backward(x, w)
        # in-place flat gradient only for weights used ONCE per step: a second use would overwrite the first gradient
        ctx.gview = grad_view_of(ws) if (once and all(wi.dtype == torch.float32 for wi in ws)) else None
        ctx.ws = ws
        ctx.meta = ([wi.shape[0] for wi in ws], [wi.dtype for wi in ws], bias is not None and bias.dtype, side)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        sizes, wdt, bdt, side = ctx.meta
        g2 = g.reshape(-1, g.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        with _NoAutocast():
            # dx = g @ W as linear(g, W^T): both operands contraction-contiguous (the layout the forward GEMMs run in,
            # 10-30 % faster than the "nn" form at these shapes); transposing the small weight costs ~nothing
            gx = None
            if ctx.needs_input_grad[0]:
                from .engine import shadow_t_of
                wt = shadow_t_of(ctx.ws, w.dtype)            # engine: W^T is maintained by one batched launch per step
                gx = matmul_nt(g2, wt if wt is not None else w.t().contiguous()).reshape(x.shape)
            gws = [None] * len(sizes)
            if any(ctx.needs_input_grad[3:]):
                # The weight gradient is off the critical path (nothing in this backward pass reads it) and MFMA-bound,
                # while the kernels that follow on the main stream (GEGLU / LayerNorm / attention backward) are
                # HBM-bound: issue it on a side stream so the two classes overlap.  Only for weights used once per step
                # (their .grad is assigned, never accumulated, before join_wgrad_stream()).
                use_side = side and WGRAD_SIDE_STREAM and g2.is_cuda
                main = torch.cuda.current_stream() if use_side else None
                ss = wgrad_stream() if use_side else None
                if use_side:
                    ss.wait_stream(main)
                    g2.record_stream(ss); x2.record_stream(ss)
                publish = ctx.gview is not None and all(ctx.needs_input_grad[3:])
                with torch.cuda.stream(ss) if use_side else _NullCtx():
                    # ctx.gview: flat fp32 gradient buffer of the optimizer engine (or None); published by _wgrad when it is the destination
                    gw = _wgrad(g2, x2, ctx.gview, ctx.ws if publish else None)
                    off = 0
                    for i, n in enumerate([] if publish else sizes):
                        if ctx.needs_input_grad[3 + i]:
                            gi = gw[off:off + n]
                            gws[i] = gi if gi.dtype == wdt[i] else gi.to(wdt[i])
                            if use_side:
                                gws[i].record_stream(main)
                        off += n
            gb = None
            if bdt is not False and ctx.needs_input_grad[1]:
                gb = colsum(g2)
                gb = gb if gb.dtype == bdt else gb.to(bdt)
        return (gx, gb, None, *gws)


def colsum(x2: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of a row-major (rows, cols) matrix (bias gradients); deterministic summation order."""
    rows, cols = x2.shape
    V = 8 if x2.dtype == torch.bfloat16 else 4
    if x2.dtype not in (torch.bfloat16, torch.float32) or cols % V or x2.stride(0) % V or x2.stride(1) != 1 or \
            x2.data_ptr() % 16 or rows == 0:
        return x2.sum(0, dtype=torch.float32)          # odd widths (e.g. a 9-class head x 8x8 patch is fine; 85-wide is not)
    out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    ws = torch.empty(_lib.lib().mmae_colsum_ws_floats(rows, cols), dtype=torch.float32, device=x2.device)
    call("mmae_colsum", dt(x2), rows, cols, ptr(x2), x2.stride(0), ptr(out), ptr(ws), stream())
    return out


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def own_wgrad_ok(g2: torch.Tensor, x2: torch.Tensor) -> bool:
    if not (OWN_GEMM & 2) or g2.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16 or not g2.is_cuda:
        return False
    if g2.stride(1) != 1 or x2.stride(1) != 1 or g2.data_ptr() % 16 or x2.data_ptr() % 16:
        return False
    return bool(_lib.lib().mmae_gemm_tn_supported(g2.shape[0], g2.shape[1], x2.shape[1], g2.stride(0), x2.stride(0)))


def gemm_tn(g2: torch.Tensor, x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (N, Kin) fp32 = g2 (rows, N)^T @ x2 (rows, Kin) on the own split-K kernel (csrc/gemm.hip); caller checked own_wgrad_ok."""
    rows, n_out, n_in = g2.shape[0], g2.shape[1], x2.shape[1]
    out = torch.empty(n_out, n_in, dtype=torch.float32, device=g2.device) if out is None else out
    assert out.shape == (n_out, n_in) and out.is_contiguous() and out.dtype == torch.float32
    nws = _lib.lib().mmae_gemm_tn_ws_floats(rows, n_out, n_in)
    ws = torch.empty(nws, dtype=torch.float32, device=g2.device) if nws else None
    CALLS["mmae_gemm_tn"] += 1
    call("mmae_gemm_tn", rows, n_out, n_in, ptr(g2), g2.stride(0), ptr(x2), x2.stride(0), ptr(out), ptr(ws), stream())
    return out


# Split-K sums of weight gradients that go straight into the optimizer engine's flat buffer are DEFERRED: the partial products are kept
# and a layer's worth of sums runs as ONE multi-tensor launch (mmae_splitk_sum_multi) -- when the layer's backward has run
# (_KvQ.backward, the last projection of a layer) and, for whatever is left, when the backward pass ends (autograd engine callback).
# The gradient is PUBLISHED (engine.grads_written_in_place: .grad set, DP reducer told) by the flush, after the launch that completes it,
# so its only consumers -- optimizer step and gradient all-reduce -- stay ordered behind it on the stream.  ~140 dependent 6-us launches
# per step become ~15.  (A gradient that autograd itself consumes is never deferred: its consumer kernel is enqueued at once.)
# No sticky state: the end-of-backward callback is queued whenever the queue goes from empty to non-empty (a backward pass that raised
# never ran its callbacks; the next pass registers its own), and engine.FlatAdamW.zero_grad() drops whatever a failed pass left queued
# (reset_splitk) while step() / grad_norm() flush as a backstop.
DEFER_SPLITK = True
_SPLITK_Q = []                 # (partials, S, n, out view, weights to publish)


def reset_splitk():
    """Drop queued split-K sums without running them (their backward pass failed: engine.FlatAdamW.zero_grad)."""
    _SPLITK_Q.clear()


def flush_splitk():
    """Run the queued split-K sums (one launch per 16) and publish their gradients.  Safe to call at any time."""
    if not _SPLITK_Q:
        return
    q = list(_SPLITK_Q)
    _SPLITK_Q.clear()
    n = len(q)
    parts = (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in q])
    outs = (ctypes.c_void_p * n)(*[e[3].data_ptr() for e in q])
    Ss = (ctypes.c_int * n)(*[e[1] for e in q])
    ns = (ctypes.c_long * n)(*[e[2] for e in q])
    call("mmae_splitk_sum_multi", n, ctypes.cast(parts, ctypes.c_void_p), ctypes.cast(outs, ctypes.c_void_p),
         ctypes.cast(Ss, ctypes.c_void_p), ctypes.cast(ns, ctypes.c_void_p), stream())
    from .engine import grads_written_in_place
    for e in q:
        if e[4] is not None:
            grads_written_in_place(e[4])


def _end_of_backward_flush():
    flush_splitk()


def _wgrad(g2, x2, gview, publish=None, defer=False):
    """dW = g2^T x2 in fp32, written into `gview` when given: the own transposing split-K kernel where it applies, else a split-K
    batched library GEMM + sum.  publish: the weights whose in-place gradient this is -- the call then publishes it itself
    (engine.grads_written_in_place), at once or, for a deferred split-K sum, at the flush; the caller must not.
    defer (with gview, no publish): the split-K sum may join the deferred queue although THIS call publishes nothing -- a later call
    of the same backward node publishes the weights (one gradient assembled from several GEMMs); the caller guarantees that call."""
    rows, n_out, n_in = g2.shape[0], g2.shape[1], x2.shape[1]

    def done(out):
        if publish is not None:
            from .engine import grads_written_in_place
            grads_written_in_place(publish)
        return out
    if own_wgrad_ok(g2, x2) and (gview is None or (gview.is_contiguous() and gview.data_ptr() % 16 == 0)):
        return done(gemm_tn(g2, x2, gview))
    S = _split_k(rows, n_out, n_in)
    if S > 1:
        part = torch.bmm(g2.view(S, rows // S, n_out).transpose(1, 2), x2.view(S, rows // S, n_in))
        if part.dtype == torch.bfloat16 and (n_out * n_in) % 8 == 0:
            out = gview if gview is not None else torch.empty(n_out, n_in, dtype=torch.float32, device=g2.device)
            if (publish is not None or defer) and gview is not None and DEFER_SPLITK and _lib.raw_stream() == 0:      # the default stream
                if not _SPLITK_Q:       # empty -> non-empty: this backward pass flushes at its end (a few per pass: _KvQ flushes per layer)
                    torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_flush)
                _SPLITK_Q.append((part, S, n_out * n_in, out, publish))
                return out
            call("mmae_splitk_sum", S, n_out * n_in, ptr(part), ptr(out), stream())
            return done(out)
        return done(torch.sum(part, 0, dtype=torch.float32, out=gview) if gview is not None else part.sum(0, dtype=torch.float32))
    gw = torch.mm(g2.t(), x2)
    return done(gview.copy_(gw) if gview is not None else gw.float())


class _KvQ(torch.autograd.Function):
    """Block_Fusion projections in one node: kv = z @ Wkv^T over ALL rows, q = z[r0:r0+n] @ Wq^T over the fusion rows.
    One node instead of two Linear nodes on z and a slice of z: the backward adds the query path into the slice of dz in
    place (addmm) instead of autograd zero-filling a full-size gradient for the slice and summing two (rows, D) tensors."""

    @staticmethod
    def forward(ctx, z, r0, n, wq, wkv):
        from .engine import grad_view_of, shadow_of
        T = z.dtype
        wq_c = shadow_of((wq,), T)
        wq_c = wq_c if wq_c is not None else (wq if wq.dtype == T else wq.to(T))
        wkv_c = shadow_of((wkv,), T)
        wkv_c = wkv_c if wkv_c is not None else (wkv if wkv.dtype == T else wkv.to(T))
        with _NoAutocast():
            kv = matmul_nt(z, wkv_c)
            q = matmul_nt(z[r0:r0 + n], wq_c)
        ctx.save_for_backward(z, wq_c, wkv_c)
        ctx.wkv, ctx.wq = wkv, wq
        ctx.cfg = (r0, n, grad_view_of((wq,)) if wq.dtype == torch.float32 else None,
                   grad_view_of((wkv,)) if wkv.dtype == torch.float32 else None, wq.dtype, wkv.dtype)
        return kv, q

    @staticmethod
    def backward(ctx, gkv, gq):
        z, wq_c, wkv_c = ctx.saved_tensors
        r0, n, gvq, gvkv, dq_, dkv_ = ctx.cfg
        gkv = gkv if gkv.is_contiguous() else gkv.contiguous()
        gq = gq if gq.is_contiguous() else gq.contiguous()
        with _NoAutocast():
            from .engine import shadow_t_of
            wt = shadow_t_of((ctx.wkv,), wkv_c.dtype)
            gz = matmul_nt(gkv, wt if wt is not None else wkv_c.t().contiguous())
            zs = gz[r0:r0 + n]
            torch.addmm(zs, gq, wq_c, out=zs)
            both = gvq is not None and gvkv is not None
            gwkv = _wgrad(gkv, z, gvkv, (ctx.wkv,) if both else None)
            gwq = _wgrad(gq, z[r0:r0 + n], gvq, (ctx.wq,) if both else None)
        if both:
            flush_splitk()            # the last projections of a layer in backward order: this layer's deferred split-K sums in one launch
            return gz, None, None, None, None
        return gz, None, None, gwq if gwq.dtype == dq_ else gwq.to(dq_), gwkv if gwkv.dtype == dkv_ else gwkv.to(dkv_)


def kv_q_projections(z, r0, n, wq, wkv):
    """-> (kv over all rows of z, q over rows [r0, r0+n)).  Both weights must be used once per step (see linear())."""
    return _KvQ.apply(z, r0, n, wq, wkv)



class _KvCtx(torch.autograd.Function):
    """Pool K/V projection over ALL rows of z and the decoders' context projections over its LAST n rows in one node:
    kv = z @ Wkv^T, ctx = z[r0:r0+n] @ cat(Wc_i)^T + cat(b_i).  One GEMM serves every decoder (their proj_context weights
    row-concatenated: three N = 256 GEMMs become one N = 768), and the backward adds the context path into the tail rows
    of dz in place -- autograd otherwise zero-fills a full-size gradient for the row slice and sums one (rows, D) tensor per
    decoder.  The concatenated weights are assembled per call (they do not sit back to back in the flat buffer); their
    gradients go back through autograd as row slices of one split-K result."""

    @staticmethod
    def forward(ctx, z, r0, n, wkv, nw, *wb):
        from .engine import grad_view_of, shadow_of
        T = z.dtype
        ws, bs = wb[:nw], wb[nw:]

        def cast(w):
            c = shadow_of((w,), T)
            return c if c is not None else (w if w.dtype == T else w.to(T))
        wkv_c = cast(wkv)
        wc = torch.cat([cast(w) for w in ws], dim=0)
        bc = torch.cat([b if b.dtype == T else b.to(T) for b in bs], dim=0) if bs else None
        with _NoAutocast():
            kv = matmul_nt(z, wkv_c)
            c = torch.nn.functional.linear(z[r0:r0 + n], wc, bc)
        ctx.save_for_backward(z, wkv_c, wc)
        ctx.wkv = wkv
        ctx.cfg = (r0, n, nw, grad_view_of((wkv,)) if wkv.dtype == torch.float32 else None, wkv.dtype,
                   [w.shape[0] for w in ws], [w.dtype for w in ws], [b.dtype for b in bs])
        return kv, c

    @staticmethod
    def backward(ctx, gkv, gc):
        z, wkv_c, wc = ctx.saved_tensors
        r0, n, nw, gvkv, dkv_, sizes, wdt, bdt = ctx.cfg
        gkv = gkv if gkv.is_contiguous() else gkv.contiguous()
        gc = gc if gc.is_contiguous() else gc.contiguous()
        with _NoAutocast():
            from .engine import shadow_t_of
            gz = None
            if ctx.needs_input_grad[0]:
                wt = shadow_t_of((ctx.wkv,), wkv_c.dtype)
                gz = matmul_nt(gkv, wt if wt is not None else wkv_c.t().contiguous())
                zs = gz[r0:r0 + n]
                torch.addmm(zs, gc, wc, out=zs)
            need_kv = ctx.needs_input_grad[3]                      # a frozen weight gets no gradient (and no flat-buffer write)
            need_w = [ctx.needs_input_grad[5 + i] for i in range(nw)]
            need_b = [ctx.needs_input_grad[5 + nw + i] for i in range(len(bdt))]
            gwkv = _wgrad(gkv, z, gvkv, (ctx.wkv,) if gvkv is not None else None) if need_kv else None
            gwc = _wgrad(gc, z[r0:r0 + n], None) if any(need_w) else None
            gb = colsum(gc) if (bdt and any(need_b)) else None
        gws, gbs, off = [], [], 0
        for i, (sz, dt_) in enumerate(zip(sizes, wdt)):
            g = gwc[off:off + sz] if need_w[i] else None
            gws.append(g if (g is None or g.dtype == dt_) else g.to(dt_))
            off += sz
        off = 0
        for i, (sz, dt_) in enumerate(zip(sizes, bdt)):
            g = gb[off:off + sz] if need_b[i] else None
            gbs.append(g if (g is None or g.dtype == dt_) else g.to(dt_))
            off += sz
        if need_kv and gvkv is not None:
            gwkv = None                                     # published by _wgrad (in place in the engine's flat buffer)
        elif gwkv is not None and gwkv.dtype != dkv_:
            gwkv = gwkv.to(dkv_)
        return (gz, None, None, gwkv, None, *gws, *gbs)


def kv_ctx_projections(z, r0, n, wkv, ctx_weights, ctx_biases):
    """-> (kv over all rows of z, ctx over rows [r0, r0+n) with the row-concatenated `ctx_weights` (+ biases)).
    wkv must be used once per step (see linear())."""
    assert len(ctx_biases) in (0, len(ctx_weights))
    return _KvCtx.apply(z, r0, n, wkv, len(ctx_weights), *ctx_weights, *ctx_biases)


class _SplitColsF32(torch.autograd.Function):
    """x (rows, sum widths) in the compute dtype -> one contiguous fp32 tensor per column block (+ an optional broadcast row
    added to it).  The backward writes the blocks' gradients side by side into ONE (rows, sum widths) tensor of x's dtype
    (autograd's own slice backward would zero-fill a full-width tensor per block and add them up)."""

    @staticmethod
    def forward(ctx, x, widths, *adds):
        outs, off = [], 0
        for w, a in zip(widths, adds):
            o = x[:, off:off + w].to(torch.float32)
            if a is not None:
                if x.dtype != torch.float32:
                    o += a.reshape(1, -1).float()                   # o is a fresh tensor (the cast made it)
                else:
                    o = o + a.reshape(1, -1).float()
            outs.append(o.contiguous())
            off += w
        ctx.cfg = (x.dtype, tuple(widths), [None if a is None else (a.shape, a.dtype) for a in adds])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        xdt, widths, ameta = ctx.cfg
        rows = next(g.shape[0] for g in gs if g is not None)
        gx = torch.empty(rows, sum(widths), dtype=xdt, device=next(g.device for g in gs if g is not None))
        gadds, off = [], 0
        for w, g, am in zip(widths, gs, ameta):
            if g is None:
                gx[:, off:off + w].zero_()
                gadds.append(None)
            else:
                gx[:, off:off + w].copy_(g)
                gadds.append(None if am is None else colsum(g if g.is_contiguous() else g.contiguous()).to(am[1]).reshape(am[0]))
            off += w
        return (gx, None, *gadds)


def split_cols_f32(x, widths, adds=None):
    """-> list of fp32 (rows, w_i) tensors: x[:, block i] (+ adds[i] broadcast over the rows when given)."""
    adds = [None] * len(widths) if adds is None else list(adds)
    return list(_SplitColsF32.apply(x, tuple(widths), *adds))

def linear(x, weight, bias=None, side_wgrad=False, once=False):
    """weight: one (N, K) master weight or a list of them (row-concatenated, e.g. [to_q.weight, to_kv.weight]).
    once: these weights are used exactly once per optimizer step -- their fp32 gradient may then be written straight
    into the optimizer engine's flat buffer (engine.FlatAdamW) instead of a fresh tensor that autograd accumulates.
    side_wgrad (needs once): the gradient GEMM may run on the side stream; the caller must join_wgrad_stream() before
    the gradients are read (PretrainStep / GradAllReducer do)."""
    ws = weight if isinstance(weight, (list, tuple)) else (weight,)
    return _Linear.apply(x, bias, (bool(side_wgrad) and bool(once), bool(once)), *ws)


# ------------------------------------------------------------------------------------------------ attention
MHA_ROUTES = {0: "mha_fwd_kernel<float>", 1: "mha_bf16_fwd32_kernel", 2: "mha_sh_fwd_kernel"}


def mha_fwd_route(dtype, dh, qseg, kseg, kv_stride):
    """Which forward kernel mmae_mha_fwd launches for this call (keys of MHA_ROUTES) -- the library's own routing test."""
    r = _lib.lib().mmae_mha_fwd_route(1 if dtype == torch.bfloat16 else 0, dh, qseg.nseg, qseg.max_rows, kseg.max_rows, kv_stride, kv_stride)
    if r < 0:
        raise _lib.MmaeLibraryError("mmae_mha_fwd_route: bad arguments")
    return r


class _MHA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qt, kvt, qcol, kcol, vcol, H, dh, qseg, kseg, scale, empty_mode, variant=0):
        I = H * dh
        same = kvt is None
        kv = qt if same else kvt
        assert qt.dim() == 2 and qt.stride(1) == 1 and kv.dim() == 2 and kv.stride(1) == 1
        assert qt.dtype == kv.dtype
        new = torch.empty if qseg.covers_all else torch.zeros
        out = new(qt.shape[0], I, dtype=qt.dtype, device=qt.device)
        lse = new(H, qt.shape[0], dtype=torch.float32, device=qt.device)
        if kseg.max_rows // 64 + kseg.nseg > 80 or qseg.max_rows // 64 + qseg.nseg > 80:
            raise _lib.MmaeLibraryError("masked attention: at most 80 64-row tiles (~4.8k rows) per sample in one call "
                                        "(MAXT in csrc/mha_bf16.hip); got %d query / %d key rows" % (qseg.max_rows, kseg.max_rows))
        es = qt.element_size()
        # bench.py's roofline_attention: every launch that the library routes to the sample-head forward kernel (encoder
        # blocks; the few-query pooling calls run the tile-per-block kernel) -- asked of the library itself
        # (mmae_mha_fwd_route), so the event average is the same population as mha_sh_fwd_kernel's row in the rocprofv3 summary
        timed = False
        if _TIMER is not None and not variant and dh == 64:
            route = mha_fwd_route(qt.dtype, dh, qseg, kseg, kv.stride(0))
            timed = route == 2 or (route == 0 and qseg.max_rows >= 128 and kseg.max_rows >= 128)    # fp32: the same calls
        if timed:
            _TIMER.kernels.add(MHA_ROUTES[route])
            ql, kl = qseg.length.long(), kseg.length.long()
            pairs = (ql[:, :-1] * kl[:, :-1]).sum() + (ql[:, -1] * kl.sum(1)).sum()     # mask-aware (q, k) pairs
            _TIMER.add_flops(pairs.double() * (4.0 * dh * H))
            ev0, ev1 = _TIMER.bracket()
            ev0.record()
        call("mmae_mha_fwd_variant" if variant else "mmae_mha_fwd", dt(qt), dh, qseg.B, H, qseg.nseg,
             ctypes.c_void_p(qt.data_ptr() + qcol * es), ctypes.c_void_p(kv.data_ptr() + kcol * es),
             ctypes.c_void_p(kv.data_ptr() + vcol * es), ptr(out), ptr(lse),
             qt.stride(0), kv.stride(0), kv.stride(0), out.stride(0), qt.shape[0],
             ptr(qseg.start), ptr(qseg.length), ptr(kseg.start), ptr(kseg.length), qseg.max_rows, kseg.max_rows, scale, empty_mode,
             *((variant,) if variant else ()), stream())
        if timed:
            ev1.record()
        ctx.save_for_backward(qt, kv, out, lse)
        ctx.cfg = (qcol, kcol, vcol, H, dh, qseg, kseg, scale, empty_mode, same, variant)
        return out

    @staticmethod
    def backward(ctx, gout):
        qt, kv, out, lse = ctx.saved_tensors
        qcol, kcol, vcol, H, dh, qseg, kseg, scale, empty_mode, same, variant = ctx.cfg
        gout = _c(gout)
        I = H * dh
        # every row of q / kv belongs to exactly one segment, and the q, k, v column slices are written completely;
        # columns outside them (none for fused qkv / kv projections) must not exist.
        assert qt.shape[1] == (3 * I if same else I) and kv.shape[1] == (3 * I if same else 2 * I), \
            "attention operands must be exactly the fused projection outputs"
        # every row of q / kv belongs to exactly one segment unless the Segments say otherwise (then: zero-filled)
        gq = torch.empty_like(qt) if (qseg.covers_all and (not same or kseg.covers_all)) else torch.zeros_like(qt)
        gkv = gq if same else (torch.empty_like(kv) if kseg.covers_all else torch.zeros_like(kv))
        # workspace: delta plus, for the bf16 kernels, the row constants of the key-stationary dK/dV kernel (3 planes of (H, rows))
        lib_ = _lib.lib()
        fused = (not variant and MHA_FUSED_BWD and qt.dtype == torch.bfloat16 and dh == 64 and qseg.max_rows >= 128 and kseg.max_rows >= 128 and
                 (kseg.max_rows // 32 + kseg.nseg + 7) // 8 <= MHA_FUSED_MAX_PASSES and
                 bool(lib_.mmae_mha_bwd_fused_supported(dt(qt), dh, qseg.B, H, qseg.nseg, qseg.max_rows, kseg.max_rows)))
        if fused or (variant > 0 and 50 <= (variant & 255) <= 55):      # planes + fp32 partial dQ tiles
            nws = lib_.mmae_mha_bwd_fused_ws_floats(qseg.B, H, qseg.nseg, lse.shape[1], qseg.max_rows)
        else:
            nws = lib_.mmae_mha_bwd_ws_floats(H, lse.shape[1])
        # the fused kernel's workspace (row constants + fp32 partial dQ tiles of the tiles met in several key passes: ~0.47 GB at the bench
        # shape) is scratch that lives for ONE launch: every layer's backward on a stream reuses one buffer (launches on a stream are
        # ordered) instead of a fresh allocation per layer -- and per layer of a captured graph's private pool
        delta = _mha_ws(nws, lse.dtype, lse.device) if fused else torch.empty(nws, dtype=lse.dtype, device=lse.device)
        es = qt.element_size()
        call("mmae_mha_bwd_fused" if fused else ("mmae_mha_bwd_variant" if variant else "mmae_mha_bwd"), dt(qt), dh, qseg.B, H, qseg.nseg,
             ctypes.c_void_p(qt.data_ptr() + qcol * es), ctypes.c_void_p(kv.data_ptr() + kcol * es),
             ctypes.c_void_p(kv.data_ptr() + vcol * es), ptr(out), ptr(gout), ptr(lse), ptr(delta),
             ctypes.c_void_p(gq.data_ptr() + qcol * es), ctypes.c_void_p(gkv.data_ptr() + kcol * es),
             ctypes.c_void_p(gkv.data_ptr() + vcol * es),
             qt.stride(0), kv.stride(0), kv.stride(0), out.stride(0), gout.stride(0), gq.stride(0), gkv.stride(0),
             gkv.stride(0), qt.shape[0], ptr(qseg.start), ptr(qseg.length), ptr(kseg.start), ptr(kseg.length),
             qseg.max_rows, kseg.max_rows, scale, empty_mode, *((variant,) if variant else ()), stream())
        return gq, (None if same else gkv), None, None, None, None, None, None, None, None, None, None


_MHA_WS = {}


def _mha_ws(n: int, dtype, device) -> torch.Tensor:
    """One scratch buffer per (device, stream, dtype), grown on demand; contents are dead between launches.  Inside a graph capture
    the buffer comes from the capture's own pool (a cached tensor must not outlive or predate the graph that bakes its address in)."""
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=dtype, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream, dtype)
    buf = _MHA_WS.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=dtype, device=device)
        _MHA_WS[key] = buf
    return buf[:n]


MHA_SELF_VARIANT = 0       # variant of mha_self calls that pass none (0 = product kernels; tools/tuning_env.py sets it for A/B runs)
MHA_FUSED_BWD = True       # bf16 / dh 64 calls with >= 128 query and key rows per sample (the encoder blocks) run their backward as ONE kernel
                           # (mmae_mha_bwd_fused: dQ + dK + dV from five tile products) where its schedule tables fit; False: the dQ + dK/dV
                           # kernel pair of mmae_mha_bwd everywhere.  Same-box A/B in the step: 154.4 -> 153.1 ms (tools/tuning_env.py: MMAE_MHA_FUSED_BWD)
MHA_FUSED_MAX_PASSES = 3   # ... and only while a sample's keys fit this many passes of 256 (upper bound from max_k_rows and the segment count): every
                           # pass sweeps the fusion queries' tiles again and carries their fp32 partials through the workspace.  640 keys / 4
                           # segments (the headline, ViT-L 3-modality): 3 passes, fused wins; config 5 (768 keys, 5 segments, per-sample
                           # dropout): 4 passes, same-box A/B 146.9 (pair) vs 147.8 ms (fused) -> the pair


def _variant_word(variant: int, hpb: int) -> int:
    """csrc/mmae_internal.h: bits 0..7 of a non-negative variant name the kernel, bits 8..11 the heads one workgroup of the
    sample-head kernels walks (0: chosen from (B, H) as the product path does)."""
    assert 0 <= hpb < 16
    return variant if (variant < 0 or not hpb) else (variant | (hpb << 8))


def mha_self(qkv: torch.Tensor, H: int, dh: int, seg: Segments, scale: float, order: str = "qkv", variant: Optional[int] = None,
             hpb: int = 0) -> torch.Tensor:
    """Self attention on a fused projection output qkv (rows, 3*H*dh) laid out [q | k | v] column blocks.
    variant != 0 / hpb != 0: test / tuning kernels through csrc/mmae_internal.h (per call, no global state)."""
    variant = MHA_SELF_VARIANT if variant is None else variant
    if variant and dh != 64:
        variant = 0
    I = H * dh
    return _MHA.apply(qkv, None, 0, I, 2 * I, H, dh, seg, seg, scale, 0, _variant_word(variant, hpb))


def mha_cross(q: torch.Tensor, kv: torch.Tensor, H: int, dh: int, qseg: Segments, kseg: Segments, scale: float,
              empty_mode: int = 0, variant: int = 0, hpb: int = 0) -> torch.Tensor:
    """q (rows_q, H*dh), kv (rows_k, 2*H*dh) = [k | v]."""
    return _MHA.apply(q, kv, 0, 0, H * dh, H, dh, qseg, kseg, scale, empty_mode, _variant_word(variant, hpb))


# ------------------------------------------------------------------------------------------------ modality attention
class _ModAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, slot_row, B, P, ns, H, dh, shared_base, scale):
        I = H * dh
        assert q.shape == (B * P, I) and kv.shape[1] == 2 * I and q.stride(1) == 1 and kv.stride(1) == 1
        assert slot_row.dtype == torch.int32 and slot_row.is_contiguous()
        out = torch.empty(B * P, I, dtype=q.dtype, device=q.device)
        call("mmae_modattn_fwd", dt(q), dh, B, P, ns, I, ptr(q), q.stride(0), ptr(kv), kv.stride(0), ptr(slot_row),
             ptr(out), out.stride(0), scale, stream())
        ctx.save_for_backward(q, kv, slot_row)
        ctx.cfg = (B, P, ns, H, dh, shared_base, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        q, kv, slot_row = ctx.saved_tensors
        B, P, ns, H, dh, shared_base, scale = ctx.cfg
        gout = _c(gout)
        gq = torch.empty_like(q)
        gkv = torch.empty_like(kv)
        assert kv.shape[0] == shared_base + P, "kv rows must be [token rows | P mask-embedding rows]"
        ws = torch.empty(_lib.lib().mmae_modattn_bwd_nsplit(B) * P * 2 * H * dh, dtype=torch.float32, device=q.device)
        call("mmae_modattn_bwd", dt(q), dh, B, P, ns, H * dh, ptr(q), q.stride(0), ptr(kv), kv.stride(0),
             ptr(slot_row), ptr(gout), gout.stride(0), ptr(gq), gq.stride(0), ptr(gkv), gkv.stride(0), shared_base,
             scale, ptr(ws), stream())
        return gq, gkv, None, None, None, None, None, None, None, None


def modattn(q, kv, slot_row, B, P, ns, H, dh, shared_base, scale):
    return _ModAttn.apply(q, kv, slot_row, B, P, ns, H, dh, shared_base, scale)


# ------------------------------------------------------------------------------------------------ add + LayerNorm
class _PartsAddLN(torch.autograd.Function):
    """Several row-parts of the fp32 residual stream -> one contiguous normalised matrix.

    forward(delta, g1, b1, g2, b2, gb1, gb2, y_buf, yb_buf, cfg, *xs): part i has rows xs[i] (r_i, D) fp32;
    cfg.delta_off[i] is the row offset of its delta inside `delta` (or -1: no delta, residual unchanged).
    Returns (x_new_i for the parts that have a delta ..., y[, yb]) with y = LN2(LN1(x_new)), rows of all parts concatenated.
      y_buf / cfg.y_row0: write y into rows [y_row0, y_row0 + total) of an existing matrix instead of a new one (the matrix
        is returned, marked dirty) -- lets two calls fill one GEMM operand.
      cfg.dual = (part index, yb_row0) with gb1, gb2, yb_buf: that part is ALSO normalised with the second gamma pair into
        rows [yb_row0, ...) of yb_buf, in the same pass (mmae_add_ln_fwd_dual / _bwd_dual); bias-less double LayerNorm only.
    """

    @staticmethod
    def forward(ctx, delta, g1, b1, g2, b2, gb1, gb2, y_buf, yb_buf, cfg, *xs):
        eps1, eps2, out_dtype, delta_off, y_row0, dual, cast_copy = cfg
        ctx.set_materialize_grads(False)        # an unused output (a dropped alias, the last layer's residual) arrives as None,
                                                # not as a zero-filled tensor of its size (18 fills per step at ViT-B)
        D = xs[0].shape[1]
        rows = [x.shape[0] for x in xs]
        total = sum(rows)
        dev = xs[0].device
        if y_buf is None:
            y = torch.empty(total, D, dtype=out_dtype, device=dev)
        else:
            assert y_buf.dtype == out_dtype and y_buf.is_contiguous() and y_buf.shape[1] == D and y_row0 + total <= y_buf.shape[0]
            y = y_buf
            ctx.mark_dirty(y_buf)
        if dual is not None:
            assert b1 is None and b2 is None and g2 is not None and yb_buf.dtype == out_dtype and yb_buf.is_contiguous()
            assert dual[1] + rows[dual[0]] <= yb_buf.shape[0] and yb_buf.shape[1] == D
            ctx.mark_dirty(yb_buf)
            stats_b = torch.empty(rows[dual[0]], 4, dtype=torch.float32, device=dev)
        else:
            stats_b = None
        stats = torch.empty(total, 4, dtype=torch.float32, device=dev)
        y2 = None
        if cast_copy:                       # fp32 y plus its bf16 copy from the same pass (mmae_add_ln_fwd_cast)
            assert out_dtype == torch.float32 and dual is None and b1 is None and b2 is None and y_buf is None and D in (768, 1024)
            y2 = torch.empty(total, D, dtype=torch.bfloat16, device=dev)
        ln_in, outs = [], []
        r0 = 0
        ddt = dt(delta) if delta is not None else _lib.F32
        esz = y.element_size()
        for i, (x, off) in enumerate(zip(xs, delta_off)):
            assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == D
            if off >= 0:
                xn = torch.empty_like(x)
                dptr = ctypes.c_void_p(delta.data_ptr() + off * D * delta.element_size())
                outs.append(xn)
            else:
                xn, dptr = x, None
                if x.requires_grad:
                    # residual unchanged: hand back an alias so that the stream's gradient flows THROUGH this function
                    # (added inside the backward kernel) instead of being summed by a separate full-size autograd add
                    outs.append(x.view_as(x))
            if x.shape[0] > 0:                                  # (an empty part -- e.g. no kept token at all -- is legal)
                yp = ctypes.c_void_p(y.data_ptr() + (y_row0 + r0) * D * esz)
                sp = ctypes.c_void_p(stats.data_ptr() + r0 * 16)
                if dual is not None and i == dual[0]:
                    call("mmae_add_ln_fwd_dual", ddt, dt(out_dtype), x.shape[0], D, ptr(x), dptr, ptr(xn) if off >= 0 else None,
                         yp, ctypes.c_void_p(yb_buf.data_ptr() + dual[1] * D * esz), ptr(g1), ptr(g2), ptr(gb1), ptr(gb2),
                         eps1, eps2, sp, ptr(stats_b), stream())
                elif cast_copy:
                    call("mmae_add_ln_fwd_cast", ddt, x.shape[0], D, ptr(x), dptr, ptr(xn) if off >= 0 else None, yp,
                         ctypes.c_void_p(y2.data_ptr() + r0 * D * 2), ptr(g1), eps1, ptr(g2), eps2, sp, stream())
                else:
                    call("mmae_add_ln_fwd", ddt, dt(out_dtype), x.shape[0], D, ptr(x), dptr, ptr(xn) if off >= 0 else None,
                         yp, ptr(g1), ptr(b1), eps1, ptr(g2), ptr(b2), eps2, sp, stream())
            ln_in.append(xn)
            r0 += x.shape[0]
        ctx.save_for_backward(g1, b1, g2, gb1, gb2, stats, stats_b, *ln_in)
        ctx.meta = (rows, D, delta_off, out_dtype, None if delta is None else (delta.shape, delta.dtype),
                    b1 is not None, b2 is not None, [off >= 0 or x.requires_grad for x, off in zip(xs, delta_off)],
                    y_row0, dual, y_buf is not None, cast_copy)
        if cast_copy:
            return (*outs, y, y2)
        return (*outs, y) if dual is None else (*outs, y, yb_buf)

    @staticmethod
    def backward(ctx, *grads):
        g1, b1, g2, gb1, gb2, stats, stats_b, *ln_in = ctx.saved_tensors
        rows, D, delta_off, out_dtype, dmeta, has_b1, has_b2, has_out, y_row0, dual, y_given, cast_copy = ctx.meta
        grads = list(grads)
        gyb = grads.pop() if dual is not None else None
        if cast_copy:
            # two gradients of the same values: through the fp32 matrix and through its bf16 copy.  Usually only the copy has
            # consumers (pool / decoder projections): its gradient then IS gy, read by the kernel in bf16 -- no cast back
            gy2 = grads.pop()
            if grads[-1] is None and gy2 is not None:
                grads[-1], out_dtype = gy2, torch.bfloat16
            elif gy2 is not None:
                grads[-1] = grads[-1] + gy2.float()
        gy = grads[-1]
        ups = list(grads[:-1])          # upstream grads of the x_new / alias outputs, in part order (None: unused output)
        dev = stats.device
        if gy is None:                  # the normalised matrix itself went unused (only the residual outputs were)
            gy = torch.zeros(y_row0 + sum(rows), D, dtype=out_dtype, device=dev)
        if dual is not None and gyb is None:
            gyb = torch.zeros(dual[1] + rows[dual[0]], D, dtype=out_dtype, device=dev)
        gy = _c(gy)
        gyb = _c(gyb) if gyb is not None else None
        gdelta = torch.empty(dmeta[0], dtype=dmeta[1], device=dev) if dmeta is not None else None
        ddt = dt(dmeta[1]) if dmeta is not None else _lib.F32
        dbl = g2 is not None
        acc = [torch.empty(D, dtype=torch.float32, device=dev),
               torch.empty(D, dtype=torch.float32, device=dev) if has_b1 else None,
               torch.empty(D, dtype=torch.float32, device=dev) if dbl else None,
               torch.empty(D, dtype=torch.float32, device=dev) if (dbl and has_b2) else None]
        # the second pair's sums are ASSIGNED when the dual part is the first non-empty part (always, in the model: part 0);
        # only then may the buffers start uninitialised
        dual_first = dual is not None and all(rows[j] == 0 for j in range(dual[0])) and rows[dual[0]] > 0
        accb = [(torch.empty if dual_first else torch.zeros)(D, dtype=torch.float32, device=dev) for _ in range(2)] \
            if dual is not None else [None, None]
        ws = torch.empty(_lib.lib().mmae_add_ln_bwd_ws_floats(max(rows), D), dtype=torch.float32, device=dev)
        gxs = []
        r0 = 0
        first = True
        esz = gy.element_size()
        for i, (xn, n, off) in enumerate(zip(ln_in, rows, delta_off)):
            up = ups.pop(0) if has_out[i] else None
            need_gx = ctx.needs_input_grad[10 + i]
            if n == 0:
                gxs.append(None)
                continue
            gx = torch.empty_like(xn) if need_gx else None
            gd = ctypes.c_void_p(gdelta.data_ptr() + off * D * gdelta.element_size()) if off >= 0 else None
            gyp = ctypes.c_void_p(gy.data_ptr() + (y_row0 + r0) * D * esz)
            sp = ctypes.c_void_p(stats.data_ptr() + r0 * 16)
            if dual is not None and i == dual[0]:
                # accumulate flag: acc[0] / acc[2] follow `first` like every part; the second pair's sums start at zero, so
                # assigning (first) and adding (not first) are the same thing for them
                call("mmae_add_ln_bwd_dual", ddt, dt(out_dtype), n, D, ptr(xn), gyp,
                     ctypes.c_void_p(gyb.data_ptr() + dual[1] * D * esz), ptr(_c(up)) if up is not None else None,
                     ptr(g1), ptr(g2), ptr(gb1), ptr(gb2), sp, ptr(stats_b), ptr(gx), gd, ptr(acc[0]), ptr(acc[2]),
                     ptr(accb[0]), ptr(accb[1]), ptr(ws), 0 if first else 1, stream())
                first = False
                gxs.append(gx)
                r0 += n
                continue
            tm = _TIMERS.get("mmae_add_ln_bwd")
            if tm is not None and not (dbl and not has_b1 and out_dtype == torch.bfloat16 and D == 768 and ddt == _lib.BF16 and
                                       up is not None and gx is not None and off >= 0):
                tm = None       # time ONE template instance only: add_ln_bwd_fast_kernel<bf16, bf16, 3, double, up, gx, gdelta>
                                # (the encoder's residual passes: 59 of the 72 launches of this entry point per step at ViT-B)
            if tm is not None and not tm.sample():
                tm = None
            if tm is not None:
                # algorithmic bytes: x_new (4) + gy + [gx_up (4)] read, [gx (4)] + [gdelta] written, per element
                per = 4 + gy.element_size() + (4 if up is not None else 0) + (4 if gx is not None else 0) + \
                    (gdelta.element_size() if off >= 0 else 0)
                tm.add_flops(float(n) * D * per)
                ev0, ev1 = tm.bracket()
                ev0.record()
            call("mmae_add_ln_bwd", ddt, dt(out_dtype), n, D, ptr(xn), gyp, ptr(_c(up)) if up is not None else None,
                 ptr(g1), ptr(b1), ptr(g2), sp, ptr(gx), gd, ptr(acc[0]),
                 ptr(acc[1]), ptr(acc[2]), ptr(acc[3]), ptr(ws), 0 if first else 1, stream())
            if tm is not None:
                ev1.record()
            first = False
            gxs.append(gx)
            r0 += n
        if first:
            acc = [None if a is None else torch.zeros_like(a) for a in acc]
        return (gdelta, acc[0], acc[1], acc[2], acc[3], accb[0], accb[1], gy if y_given else None, None, None, *gxs)


def parts_add_ln(xs: List[torch.Tensor], delta: Optional[torch.Tensor], delta_off: List[int], g1, b1=None, g2=None,
                 b2=None, eps1=1e-5, eps2=1e-5, out_dtype=torch.float32, y_into=None, dual=None, cast_copy=False):
    """-> (list of updated residual parts, y).  A part with delta_off < 0 keeps its residual unchanged.
    y_into = (matrix, row0): y is written into rows [row0, ...) of `matrix` (returned in its place).
    dual = (part index, gamma1_b, gamma2_b, matrix_b, row0_b): that part is also normalised with the second gamma pair into
    `matrix_b` in the same pass; the call then returns (parts, y, matrix_b).
    cast_copy (fp32 y, bias-less, width 768 / 1024; ignored otherwise -- the caller checks the length of the result): the
    kernel also writes a bf16 copy of y; the call returns (parts, y, y_bf16)."""
    y_buf, y_row0 = (None, 0) if y_into is None else y_into
    gb1, gb2, yb_buf, dcfg = (None, None, None, None) if dual is None else (dual[1], dual[2], dual[3], (dual[0], dual[4]))
    cast_copy = bool(cast_copy) and out_dtype == torch.float32 and dual is None and y_into is None and b1 is None and b2 is None \
        and xs[0].shape[1] in (768, 1024)
    outs = list(_PartsAddLN.apply(delta, g1, b1, g2, b2, gb1, gb2, y_buf, yb_buf,
                                  (eps1, eps2, out_dtype, tuple(delta_off), y_row0, dcfg, cast_copy), *xs))
    yb = outs.pop() if (dual is not None or cast_copy) else None
    y = outs.pop()
    x_news = [outs.pop(0) if (off >= 0 or x.requires_grad) else x for x, off in zip(xs, delta_off)]
    if cast_copy:
        return x_news, y, yb                    # yb: the bf16 copy of the fp32 y
    return (x_news, y) if dual is None else (x_news, y, yb)


def layernorm(x2d: torch.Tensor, g1, b1=None, g2=None, b2=None, eps1=1e-5, eps2=1e-5, out_dtype=torch.float32):
    """Plain (double) LayerNorm of an fp32 (rows, D) matrix."""
    _, y = parts_add_ln([x2d], None, [-1], g1, b1, g2, b2, eps1, eps2, out_dtype)
    return y


# ------------------------------------------------------------------------------------------------ GEGLU / GELU
class _GEGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        h = _c(h)
        F2 = h.shape[-1]
        assert F2 % 2 == 0
        rows = h.numel() // F2
        out = torch.empty(*h.shape[:-1], F2 // 2, dtype=h.dtype, device=h.device)
        call("mmae_geglu_fwd", dt(h), rows, F2 // 2, ptr(h), ptr(out), stream())
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        g = _c(g)
        F2 = h.shape[-1]
        dh = torch.empty_like(h)
        call("mmae_geglu_bwd", dt(h), h.numel() // F2, F2 // 2, ptr(h), ptr(g), ptr(dh), stream())
        return dh


class _GELU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        call("mmae_gelu_fwd", dt(x), x.numel(), ptr(x), ptr(y), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(x)
        call("mmae_gelu_bwd", dt(x), x.numel(), ptr(x), ptr(g), ptr(dx), stream())
        return dx


geglu = _GEGLU.apply
gelu = _GELU.apply


class _ScaleRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, row_scale):
        x = _c(x)
        assert x.dim() == 2 and row_scale.dtype == torch.float32 and row_scale.numel() == x.shape[0] and row_scale.is_contiguous()
        out = torch.empty_like(x)
        call("mmae_scale_rows", dt(x), x.shape[0], x.shape[1], ptr(x), ptr(row_scale), ptr(out), stream())
        ctx.save_for_backward(row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        (row_scale,) = ctx.saved_tensors
        g = _c(g)
        gx = torch.empty_like(g)
        call("mmae_scale_rows", dt(g), g.shape[0], g.shape[1], ptr(g), ptr(row_scale), ptr(gx), stream())
        return gx, None


def scale_rows(x, row_scale):
    """x (rows, W) * row_scale (rows,) fp32, row by row: DropPath on a residual branch (zorro_utils.py:69-84)."""
    return _ScaleRows.apply(x, row_scale)


def drop_path_row_scale(u: torch.Tensor, drop_prob: float, counts):
    """Per-row DropPath factors of the packed row space from one uniform draw per sample (zorro_utils.py:79-83:
    random_tensor = floor(keep + u); output = x / keep * random_tensor).  counts: rows per sample of each row block, in the order
    the blocks are stacked (e.g. (N, P): B*N modality rows then B*P fusion rows).  -> (sum(counts) * B,) fp32."""
    keep = 1.0 - float(drop_prob)
    s = torch.floor(keep + u.reshape(-1).float()) / keep
    return torch.cat([s.repeat_interleave(int(c)) for c in counts]).contiguous()

FF_CHUNKS = 1      # see _FeedForwardGEGLU: 2 was measured, no net gain
PAD_FF = True      # a GEGLU width that fits none of the own GEMM's tiles (ViT-L: 2730) runs on zero-padded operand copies (tools/tuning_env.py: MMAE_PAD_FF)
PAD_FF_MIN_TILES = 128     # ... when FeedForward[3] has at least this many output tiles.  Lower than _OWN_GEMM_MIN_TILES: the alternative here is a
                           # library GEMM at an unaligned width (config 5, same-box A/B: 512 -> 128 is 153.2 -> 146.8 ms/step; no padding: 169.0)


def _row_chunks(rows: int, n: int):
    """n nearly equal row ranges with 256-aligned boundaries (GEMM tiles stay whole)."""
    if rows < 8192 or n <= 1:
        return [(0, rows)]
    n = min(n, rows // 4096)
    step = (rows + n - 1) // n
    step = (step + 255) // 256 * 256
    return [(a, min(rows, a + step)) for a in range(0, rows, step)]


class _FeedForwardGEGLU(torch.autograd.Function):
    """f = GEGLU(y @ W1^T) @ W2^T (FeedForward[1:4], zorro_utils.py:115-128) as ONE node, optionally evaluated in
    FF_CHUNKS row chunks (GEMM1 -> GEGLU -> GEMM2 back to back per chunk, so the (rows, 2*ffi) intermediate -- 1.3 GB at
    the bench batch -- is consumed while its tail still sits in the 256 MB Infinity Cache).  Measured: in isolation
    GEGLU drops 475 -> 359 us with two chunks (tools/probes/mall_chunk_probe.py); inside the step GEGLU gains 1.2 ms and
    the half-size GEMMs lose 1.2 ms, so the default stays at one chunk.  The backward walks the same chunks:
    dg = df @ W2 -> GEGLU' -> dy = dh @ W1; the two weight gradients are split-K GEMMs over all rows afterwards.
    Both weights must be used once per step (see linear())."""

    @staticmethod
    def forward(ctx, y, w1, w2):
        from .engine import grad_view_of, padded_ff_of, shadow_of
        T = y.dtype
        y = _c(y)
        rows, F = y.shape[0], w2.shape[1]
        assert w1.shape[0] == 2 * F and y.dim() == 2
        f32 = w1.dtype == torch.float32 and w2.dtype == torch.float32
        gv1, gv2 = (grad_view_of((w1,)), grad_view_of((w2,))) if f32 else (None, None)
        # A GEGLU width that fits none of the own GEMM's tiles (ViT-L: int(1024 * 8 / 3) = 2730) runs on the engine's zero-PADDED
        # operand copies (width Fp = 2816; pads are zero, so h, g, dg, dh are zero there and every product is unchanged) when the
        # projections are big enough for the own kernel at all; otherwise on the library GEMMs at the exact width.
        pf = None
        if PAD_FF and F % 256 and T == torch.bfloat16 and (OWN_GEMM & 1) and gv1 is not None and gv2 is not None:
            Fp = (F + 255) // 256 * 256
            D = w2.shape[0]
            lib_ = _lib.lib()
            if ((rows + 255) // 256) * (D // 256) >= min(_OWN_GEMM_MIN_TILES, PAD_FF_MIN_TILES) and y.data_ptr() % 16 == 0 and \
                    lib_.mmae_gemm_geglu_supported(rows, Fp, D, y.stride(0), D, 2 * Fp, Fp) and \
                    lib_.mmae_gemm_nt_supported(rows, D, Fp, Fp, Fp, D) and lib_.mmae_gemm_nt_supported(rows, Fp, D, D, D, Fp) and \
                    lib_.mmae_gemm_nt_supported(rows, D, 2 * Fp, 2 * Fp, 2 * Fp, D):
                pf = padded_ff_of(w1, w2, T)
        if pf is not None:
            Fp = pf.Fp
            h = torch.empty(rows, 2 * Fp, dtype=T, device=y.device)
            g = torch.empty(rows, Fp, dtype=T, device=y.device)
            f = torch.empty(rows, w2.shape[0], dtype=T, device=y.device)
            with _NoAutocast():
                gemm_geglu(y, pf.w1p, h, g)                          # h = [val | pad | gate | pad], g = [GEGLU | pad]; pads come out zero
                gemm_nt(g, pf.w2p, f)
            ctx.save_for_backward(y, h, g)
            ctx.cfg = (w1, w2, gv1, gv2, pf)
            return f

        def cast(w):
            c = shadow_of((w,), T)
            return c if c is not None else (w if w.dtype == T else w.to(T))
        w1c, w2c = cast(w1), cast(w2)
        h = torch.empty(rows, 2 * F, dtype=T, device=y.device)
        g = torch.empty(rows, F, dtype=T, device=y.device)
        f = torch.empty(rows, w2.shape[0], dtype=T, device=y.device)
        with _NoAutocast():
            for a, b in _row_chunks(rows, FF_CHUNKS):
                if own_geglu_ok(y[a:b], w1c, h[a:b], g[a:b]):
                    gemm_geglu(y[a:b], w1c, h[a:b], g[a:b])         # FeedForward[1] + GEGLU in one kernel (csrc/gemm.hip)
                else:
                    matmul_nt(y[a:b], w1c, out=h[a:b])
                    call("mmae_geglu_fwd", dt(T), b - a, F, ptr(h[a:b]), ptr(g[a:b]), stream())
                matmul_nt(g[a:b], w2c, out=f[a:b])
        ctx.save_for_backward(y, h, g, w1c, w2c)
        ctx.cfg = (w1, w2, gv1, gv2, None)
        return f

    @staticmethod
    def backward(ctx, df):
        from .engine import shadow_t_of
        w1, w2, gv1, gv2, pf = ctx.cfg
        df = _c(df)
        if pf is not None:
            y, h, g = ctx.saved_tensors
            T, rows, F, Fp = y.dtype, y.shape[0], pf.F, pf.Fp
            dh = torch.empty_like(h)
            dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
            with _NoAutocast():
                dg = gemm_nt(df, pf.w2pt)                                     # (rows, Fp); zero in the pad columns (zero rows of w2pt)
                call("mmae_geglu_bwd", dt(T), rows, Fp, ptr(h), ptr(dg), ptr(dh), stream())
                if dy is not None:
                    gemm_nt(dh, pf.w1pt, dy)                                  # contraction over 2 Fp: the pad columns of dh are zero
                # weight gradients at the EXACT width, straight into the flat fp32 buffer: column slices of the padded activations
                _wgrad(df, g[:, :F], gv2, None, defer=True)
                _wgrad(dh[:, :F], y, gv1[:F], None, defer=True)               # val rows of W1
                _wgrad(dh[:, Fp:Fp + F], y, gv1[F:], (w2, w1))                # gate rows; publishes both weights (after all three sums)
            return dy, None, None
        y, h, g, w1c, w2c = ctx.saved_tensors
        T = y.dtype
        rows, F = y.shape[0], g.shape[1]
        w1t = shadow_t_of((w1,), T)
        w1t = w1t if w1t is not None else w1c.t().contiguous()          # (D, 2F)
        w2t = shadow_t_of((w2,), T)
        w2t = w2t if w2t is not None else w2c.t().contiguous()          # (F, D)
        dh = torch.empty_like(h)
        dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        with _NoAutocast():
            chunks = _row_chunks(rows, FF_CHUNKS)
            dg = torch.empty(max(b - a for a, b in chunks), F, dtype=T, device=y.device)
            for a, b in chunks:
                matmul_nt(df[a:b], w2t, out=dg[:b - a])                   # dg = df @ W2
                call("mmae_geglu_bwd", dt(T), b - a, F, ptr(h[a:b]), ptr(dg), ptr(dh[a:b]), stream())
                if dy is not None:
                    matmul_nt(dh[a:b], w1t, out=dy[a:b])                  # dy = dh @ W1
            both = gv1 is not None and gv2 is not None
            gw2 = _wgrad(df, g, gv2, (w2,) if both else None)
            gw1 = _wgrad(dh, y, gv1, (w1,) if both else None)
        if both:
            return dy, None, None
        return dy, gw1 if gw1.dtype == w1.dtype else gw1.to(w1.dtype), gw2 if gw2.dtype == w2.dtype else gw2.to(w2.dtype)


def feedforward_geglu(y, w1, w2):
    """y (rows, D) in the compute dtype -> GEGLU(y @ w1^T) @ w2^T.  w1 (2*ffi, D), w2 (D, ffi): fp32 masters, each used
    once per optimizer step."""
    return _FeedForwardGEGLU.apply(y, w1, w2)


# ------------------------------------------------------------------------------------------------ gather rows
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, unique, filt, nfilt):
        assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32
        out = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
        call("mmae_gather_rows", dt(src), idx.numel(), src.shape[1], ptr(src), src.stride(0), ptr(idx), ptr(out),
             out.stride(0), stream())
        ctx.save_for_backward(idx, filt)
        ctx.cfg = (src.shape, unique, nfilt)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, filt = ctx.saved_tensors
        shape, unique, nfilt = ctx.cfg
        g = _c(g)
        gs = torch.zeros(shape, dtype=g.dtype, device=g.device)
        if unique:
            call("mmae_scatter_rows", dt(g), idx.numel(), shape[1], ptr(g), g.stride(0), ptr(idx), ptr(gs),
                 gs.stride(0), 0, None, 0, stream())
        else:
            # destination rows repeat across filter classes (modalities) but are unique within one: one pass per class
            for f in range(nfilt):
                call("mmae_scatter_rows", dt(g), idx.numel(), shape[1], ptr(g), g.stride(0), ptr(idx), ptr(gs),
                     gs.stride(0), 1, ptr(filt), f, stream())
        return gs, None, None, None, None


class _ForkGatherRows(torch.autograd.Function):
    """(src, gathered) with gathered[r] = src[idx[r]]: `src` handed through for its OTHER consumer plus the row gather, as one
    node.  The backward scatter-ADDS the gathered rows' gradient into the gradient that came back through the pass-through
    output, in place -- instead of scattering into a zero-filled full-size tensor that autograd then adds to the other one
    (a fill and a three-operand pass over (rows, W) for a gather that touches a fraction of the rows)."""

    @staticmethod
    def forward(ctx, src, idx, filt, nfilt, fresh_grad):
        assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32
        out = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
        call("mmae_gather_rows", dt(src), idx.numel(), src.shape[1], ptr(src), src.stride(0), ptr(idx), ptr(out),
             out.stride(0), stream())
        ctx.save_for_backward(idx, filt)
        ctx.cfg = (src.shape, nfilt, bool(fresh_grad))
        ctx.set_materialize_grads(False)
        return src.view_as(src), out

    @staticmethod
    def backward(ctx, g_src, g):
        idx, filt = ctx.saved_tensors
        shape, nfilt, fresh_grad = ctx.cfg
        if g is None:
            return g_src, None, None, None, None
        g = _c(g)
        if g_src is None:
            gs, acc = torch.zeros(shape, dtype=g.dtype, device=g.device), 0
        elif fresh_grad and g_src.is_contiguous() and g_src._base is None and g_src.dtype == g.dtype:
            gs, acc = g_src, 1              # the caller vouched: produced by the pass-through output's one consumer for this node alone
        else:
            gs, acc = g_src.to(g.dtype).contiguous().clone(), 1
        if filt is None:
            call("mmae_scatter_rows", dt(g), idx.numel(), shape[1], ptr(g), g.stride(0), ptr(idx), ptr(gs), gs.stride(0), acc, None, 0,
                 stream())
        else:
            # destination rows repeat across filter classes (modalities) but are unique within one: one pass per class, each
            # adding to what the earlier ones left
            for f in range(nfilt):
                call("mmae_scatter_rows", dt(g), idx.numel(), shape[1], ptr(g), g.stride(0), ptr(idx), ptr(gs), gs.stride(0),
                     1 if (acc or f > 0) else 0, ptr(filt), f, stream())
        return gs, None, None, None, None


def fork_gather_rows(src, idx, filt=None, nfilt=0, fresh_grad=False):
    """-> (src passed through, src[idx]); use the first result in place of `src` for its other consumer.
    fresh_grad: the caller's promise that the pass-through result has exactly ONE consumer and that this consumer's backward
    returns a tensor it allocated itself and keeps no other reference to (ops.kv_context does) -- only then does the backward
    add the gathered rows' gradient into that tensor in place; otherwise it works on a copy (an identity-like consumer would
    hand back a tensor that something else still reads)."""
    return _ForkGatherRows.apply(src, idx, filt, nfilt, fresh_grad)


def gather_rows(src, idx, unique=True, filt=None, nfilt=0):
    """out[r] = src[idx[r]] (zeros for idx < 0).  Backward scatters; when destination rows repeat, pass a class id
    per row (`filt`, values 0..nfilt-1) such that rows of one class have unique destinations."""
    return _GatherRows.apply(src, idx, unique, filt, nfilt)


# ------------------------------------------------------------------------------------------------ patches / images
def patchify_gather(images: Sequence[torch.Tensor], col_offsets: Sequence[int], onehot_offset: int, Kcat: int,
                    patch: int, tok_mod: Optional[torch.Tensor], tok_patch: Optional[torch.Tensor],
                    tokens_per_sample: int, out_dtype) -> torch.Tensor:
    """No gradient (images are data).  images[m]: (B, C_m, H, W) fp32 contiguous."""
    B, _, H, W = images[0].shape
    nmod = len(images)
    imgs = [_c(im.float()) for im in images]
    arr = (ctypes.c_void_p * nmod)(*[im.data_ptr() for im in imgs])
    ch = (ctypes.c_int * nmod)(*[im.shape[1] for im in imgs])
    co = (ctypes.c_int * nmod)(*col_offsets)
    out = torch.empty(B * tokens_per_sample, Kcat, dtype=out_dtype, device=imgs[0].device)
    call("mmae_patchify_gather", dt(out_dtype), nmod, ctypes.cast(arr, ctypes.c_void_p),
         ctypes.cast(ch, ctypes.c_void_p), ctypes.cast(co, ctypes.c_void_p), onehot_offset, Kcat, B, H, W, patch,
         ptr(tok_mod), ptr(tok_patch), tokens_per_sample, ptr(out), stream())
    return out


class _Unpatchify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, B, C, H, W, patch):
        tok = _c(tok)
        img = torch.empty(B, C, H, W, dtype=torch.float32, device=tok.device)
        call("mmae_unpatchify", dt(tok), B, C, H, W, patch, ptr(tok), ptr(img), stream())
        ctx.cfg = (tok.dtype, C, patch)
        return img

    @staticmethod
    def backward(ctx, g):
        tdtype, C, patch = ctx.cfg
        B, _, H, W = g.shape
        P = (H // patch) * (W // patch)
        gt = patchify_gather([g], [0], -1, C * patch * patch, patch, None, None, P, tdtype)
        return gt, None, None, None, None, None


def unpatchify(tok, B, C, H, W, patch):
    return _Unpatchify.apply(tok, B, C, H, W, patch)


class _MaskedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, kind, patch, is_tokens, shape):
        B, C, H, W = shape
        pred = _c(pred)
        target = _c(target.float())
        P = (H // patch) * (W // patch)
        dev = pred.device
        if mask is not None:
            mask = _c(mask.to(torch.int64))
        partial = torch.empty(B * P, dtype=torch.float32, device=dev)
        den = torch.empty(B, dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        call("mmae_masked_loss_fwd", dt(pred), int(is_tokens), kind, B, C, H, W, patch, ptr(pred), ptr(target),
             ptr(mask), ptr(partial), ptr(den), ptr(stats), stream())
        ctx.save_for_backward(pred, target, mask, den, stats)
        ctx.cfg = (kind, patch, is_tokens, shape)
        return stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, den, stats = ctx.saved_tensors
        kind, patch, is_tokens, (B, C, H, W) = ctx.cfg
        g = _c(g.float()).reshape(1)
        gp = torch.empty_like(pred)
        call("mmae_masked_loss_bwd", dt(pred), int(is_tokens), kind, B, C, H, W, patch, ptr(pred), ptr(target),
             ptr(mask), ptr(den), ptr(stats), ptr(g), ptr(gp), stream())
        return gp, None, None, None, None, None, None


def masked_loss_image(pred, target, mask, kind: int, patch: int):
    """pred/target (B,C,H,W); mask (B,P) {0,1} or None.  kind 0 MSE, 1 L1."""
    return _MaskedLoss.apply(pred.float(), target, mask, kind, patch, False, tuple(pred.shape))


def masked_loss_tokens(tok, target, mask, kind: int, patch: int):
    """Fused unpatchify+loss: tok (B*P, C*patch^2) decoder output in (c ph pw) order, target image (B,C,H,W)."""
    return _MaskedLoss.apply(tok, target, mask, kind, patch, True, tuple(target.shape))


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, smooth, patch, is_tokens, shape):
        B, C, H, W = shape
        pred = _c(pred)
        target = _c(target.to(torch.int64))
        assert target.shape == (B, H, W), "class-map target must be (B, H, W)"
        P = (H // patch) * (W // patch)
        dev = pred.device
        if mask is not None:
            mask = _c(mask.to(torch.int64))
        partial = torch.empty(B * P, dtype=torch.float32, device=dev)
        den = torch.empty(B, dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        call("mmae_masked_ce_loss_fwd", dt(pred), int(is_tokens), B, C, H, W, patch, ptr(pred), ptr(target), ptr(mask),
             float(smooth), ptr(partial), ptr(den), ptr(stats), stream())
        ctx.save_for_backward(pred, target, mask, den, stats)
        ctx.cfg = (smooth, patch, is_tokens, shape)
        return stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        pred, target, mask, den, stats = ctx.saved_tensors
        smooth, patch, is_tokens, (B, C, H, W) = ctx.cfg
        g = _c(g.float()).reshape(1)
        gp = torch.empty_like(pred)
        call("mmae_masked_ce_loss_bwd", dt(pred), int(is_tokens), B, C, H, W, patch, ptr(pred), ptr(target), ptr(mask),
             float(smooth), ptr(den), ptr(stats), ptr(g), ptr(gp), stream())
        return gp, None, None, None, None, None, None


def masked_ce_image(pred, target, mask, patch: int, label_smoothing: float = 0.0):
    """pred (B,C,H,W) logits, target (B,H,W) int64 class ids, mask (B,P) {0,1} or None."""
    return _MaskedCE.apply(pred.float(), target, mask, label_smoothing, patch, False, tuple(pred.shape))


def masked_ce_tokens(tok, target, mask, C: int, patch: int, label_smoothing: float = 0.0):
    """Fused unpatchify + cross-entropy: tok (B*P, C*patch^2) decoder logits in (c ph pw) order, target (B,H,W)."""
    B, H, W = target.shape
    return _MaskedCE.apply(tok, target, mask, label_smoothing, patch, True, (B, C, H, W))


# ------------------------------------------------------------------------------------------------ contrastive heads
class _Dino(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, teacher_temp, student_temp):
        s = _c(student.float()); t = _c(teacher.float())
        B, D = s.shape
        ws = torch.empty(B, dtype=torch.float32, device=s.device)
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        call("mmae_dino_loss_fwd", B, D, ptr(s), ptr(t), student_temp, teacher_temp, ptr(ws), ptr(loss), stream())
        ctx.save_for_backward(s, t)
        ctx.cfg = (teacher_temp, student_temp, student.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        s, t = ctx.saved_tensors
        teacher_temp, student_temp, sdtype = ctx.cfg
        g = _c(g.float()).reshape(1)
        gs = torch.empty_like(s)
        call("mmae_dino_loss_bwd", s.shape[0], s.shape[1], ptr(s), ptr(t), student_temp, teacher_temp, ptr(g), ptr(gs),
             stream())
        return gs.to(sdtype), None, None, None


def dino_loss(student, teacher, teacher_temp=0.04, student_temp=0.1):
    return _Dino.apply(student, teacher, teacher_temp, student_temp)


class _HardNeg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o1, o2, tau_plus, beta, temperature):
        a = _c(o1.float()); b = _c(o2.float())
        B, D = a.shape
        ws = torch.empty(_lib.lib().mmae_hardneg_ws_floats(B, D), dtype=torch.float32, device=a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        call("mmae_hardneg_loss_fwd", B, D, ptr(a), ptr(b), tau_plus, beta, temperature, ptr(ws), ptr(loss), stream())
        ctx.save_for_backward(a, b)
        ctx.cfg = (tau_plus, beta, temperature, o1.dtype, o2.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        tau_plus, beta, temperature, d1, d2 = ctx.cfg
        B, D = a.shape
        g = _c(g.float()).reshape(1)
        ws = torch.empty(_lib.lib().mmae_hardneg_ws_floats(B, D), dtype=torch.float32, device=a.device)
        g1 = torch.empty_like(a); g2 = torch.empty_like(b)
        call("mmae_hardneg_loss_bwd", B, D, ptr(a), ptr(b), tau_plus, beta, temperature, ptr(ws), ptr(g), ptr(g1),
             ptr(g2), stream())
        return g1.to(d1), g2.to(d2), None, None, None


def hardneg_loss(o1, o2, tau_plus=0.1, beta=1.0, temperature=0.5):
    return _HardNeg.apply(o1, o2, tau_plus, beta, temperature)


# ------------------------------------------------------------------------------------------------ mask bookkeeping
def masks_from_draws(dirichlet, noise, noise_all, N: int):
    """dirichlet (R,M) f32, noise (R,M,P) f32, noise_all (R,M*P) f32 on device ->
    mask_all (R,M*P) int64, ids_keep (R,N) int64, ids_restore (R,M*P) int64."""
    R, M, P = noise.shape
    dev = noise.device
    mask_all = torch.empty(R, M * P, dtype=torch.int64, device=dev)
    ids_keep = torch.empty(R, N, dtype=torch.int64, device=dev)
    ids_restore = torch.empty(R, M * P, dtype=torch.int64, device=dev)
    call("mmae_masks_from_draws", R, M, P, N, ptr(_c(dirichlet.float())), ptr(_c(noise.float())),
         ptr(_c(noise_all.float())), ptr(mask_all), ptr(ids_keep), ptr(ids_restore), stream())
    return mask_all, ids_keep, ids_restore


class Descriptors:
    """Device-side int32 descriptors of one step (csrc/masks.hip).  Row space: [B*N tokens | B*P fusion | P mask-emb]."""

    NAMES = ["enc_start", "enc_len", "tok_mod", "tok_patch", "tok_pe", "tok_fus", "slot_row", "pool_qstart",
             "pool_qlen", "ctr_qstart", "ctr_qlen", "ctr_kstart", "ctr_klen", "status"]

    def __init__(self, mask_all: torch.Tensor, B: int, M: int, P: int, N: int):
        R = mask_all.shape[0]
        assert mask_all.dtype == torch.int64 and mask_all.shape[1] == M * P and R in (1, B)
        off = (ctypes.c_long * 15)()
        total = _lib.lib().mmae_descriptor_layout(B, M, P, N, ctypes.cast(off, ctypes.c_void_p))
        self.buf = torch.empty(total, dtype=torch.int32, device=mask_all.device)
        call("mmae_build_descriptors", B, R, M, P, N, ptr(_c(mask_all)), ptr(self.buf), stream())
        self.B, self.M, self.P, self.N = B, M, P, N
        shapes = {"enc_start": (B, M + 1), "enc_len": (B, M + 1), "slot_row": (B * P, M + 1)}
        for i, name in enumerate(self.NAMES):
            v = self.buf[off[i]:off[i + 1]]
            if name in shapes:
                v = v.view(*shapes[name])
            elif name.startswith("pool_") or name.startswith("ctr_"):
                v = v.view(B, M + 1)
            setattr(self, name, v)
        S = N + P
        self.enc_seg = Segments(self.enc_start, self.enc_len, S)
        self.pool_q = Segments(self.pool_qstart, self.pool_qlen, M + 1)
        self.ctr_q = Segments(self.ctr_qstart, self.ctr_qlen, M)
        self.ctr_k = Segments(self.ctr_kstart, self.ctr_klen, N)
        self.fus_base = B * N
        self.shared_base = B * N + B * P

    def check(self):
        """Host sync: raises if some sample's kept-token count differs from num_encoded_tokens."""
        bad = int(self.status[0].item())
        if bad:
            raise AssertionError("%d sample(s): number of kept tokens != num_encoded_tokens" % bad)
