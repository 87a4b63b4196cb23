"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce over RCCL (xGMI) overlapped with backward.

Replaces the reference's torch DistributedDataParallel wrapper (pretraining/pretrain_mmae.py:343,
find_unused_parameters=True) and its env:// NCCL bring-up (pretraining/utils/dist.py:62-93).  The only collective on
the hot path is the gradient all-reduce; samples are independent.

Design for MI355X: few LARGE buckets (default 128 MiB: xGMI is point-to-point, 7 links per GPU, so per-collective
latency is paid per bucket while bandwidth is per link), filled in reverse registration order = the order backward
produces gradients, launched from post-accumulate-grad hooks as soon as a bucket is complete so that the transfer runs on
RCCL's stream underneath the remaining backward kernels.  Buckets are launched strictly IN ORDER (bucket k only after
0..k-1), so every rank issues the same sequence of collectives whatever the local arrival order of gradients.

Unused parameters.  static_unused=True (pretraining: the same 7 parameters never receive a gradient on any rank or step;
the reference needs find_unused_parameters for them): they are detected on the first step and then no longer waited for --
no graph walk; if one of them later DOES receive a gradient the step raises instead of silently reducing a stale bucket.
static_unused=False (downstream fine-tuning, where ViTBaseline draws a random modality subset per forward and rank): every
parameter is expected every step, incomplete buckets are sent by finish() with zeros for the absent gradients, and a tiny
MAX all-reduce of the per-parameter "has gradient" flags gives every rank the same set of updated parameters -- what DDP's
find_unused_parameters=True does with its local_used_map.
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> bool:
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (utils/dist.py:62-93).  Returns True when a
    process group with world_size > 1 is active.  MMAE_DIST_SINGLE_RANK=1 also forms the group for WORLD_SIZE = 1 (the
    one-GPU rehearsal of the RCCL path: same backend, collectives, streams and reducer as N > 1, tests/test_gpu_dp_engine.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not ("RANK" in os.environ and os.environ.get("MMAE_DIST_SINGLE_RANK") == "1"):
        return False
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL over xGMI needs it on this driver
        if backend is None:
            backend = os.environ.get("MMAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, init_method="env://")
        dist.barrier()
    return True


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "work", "expected", "wire")

    def __init__(self):
        self.params, self.offsets = [], []
        self.flat = None
        self.wire = None                                     # reduced-precision copy on the wire (grad_dtype), else None
        self.pending = 0
        self.expected = 0
        self.work = None


class GradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 128 << 20, group=None,
                 average: bool = True, engine=None, static_unused: bool = True, grad_dtype: torch.dtype = torch.float32):
        """engine: an engine.FlatAdamW whose flat gradient buffer is all-reduced in place (buckets = contiguous ranges of
        it, no staging copies); without it the reducer owns its bucket buffers.
        grad_dtype=torch.bfloat16: the buckets cross the wire in bf16 (half the xGMI bytes) -- one cast before
        the collective, one cast back after the wait; the accumulation inside the collective is then bf16 too.
        After every finish(): `stats` = {'allreduce_bytes', 'buckets', 'comm_exposed_ms'} of that step (bytes each rank
        hands to the collectives; time the compute stream spent waiting on them = communication NOT hidden by backward)."""
        self.engine = engine
        self.static_unused = bool(static_unused)
        self.grad_dtype = grad_dtype
        # a process group of ONE rank issues its collectives only under MMAE_DIST_SINGLE_RANK=1 (the one-GPU rehearsal of the RCCL path:
        # init_distributed); a single process that happens to run under a launcher keeps the plain path (no collectives, no wire-dtype
        # rounding of the gradients); without a process group there is nothing to issue them on
        self.collective = dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("MMAE_DIST_SINGLE_RANK") == "1")
        self._issued = []                                    # Work handles of the collectives not yet known complete (quiesce())
        self.captured = False                                # set by quiesce(): the following steps are graph replays
        self.stats = {"allreduce_bytes": 0, "buckets": 0, "comm_exposed_ms": 0.0}
        self._next = 0
        if engine is not None:
            self._init_flat(engine, bucket_bytes, group, average)
            return
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        self.buckets: List[_Bucket] = []
        self._where = {}
        self._unused = set()
        self._first = True
        cur, cur_bytes = _Bucket(), 0
        for p in reversed(self.params):                     # gradients arrive roughly in reverse registration order
            nbytes = p.numel() * 4
            if cur.params and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = _Bucket(), 0
            cur.offsets.append(cur_bytes // 4)
            cur.params.append(p)
            cur_bytes += nbytes
        if cur.params:
            self.buckets.append(cur)
        for bi, b in enumerate(self.buckets):
            n = b.offsets[-1] + b.params[-1].numel()
            b.flat = torch.zeros(n, dtype=torch.float32, device=b.params[0].device)
            for p, off in zip(b.params, b.offsets):
                self._where[p] = (bi, off)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _init_flat(self, engine, bucket_bytes, group, average):
        self.params = list(engine.params)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        self.buckets, self._where, self._unused, self._first = [], {}, set(), True
        # flat order == registration order; gradients arrive roughly in reverse, so buckets are cut from the end
        ranges, cur, hi, end = [], [], engine.n, engine.n
        for p in reversed(self.params):
            off = engine.offsets[id(p)]
            if cur and (hi - off) * 4 > bucket_bytes:
                ranges.append((end, hi, cur))                # [end, hi) holds the parameters collected so far
                cur, hi = [], end
            cur.append(p)
            end = off
        if cur:
            ranges.append((end, hi, cur))
        for bi, (lo, hi, ps) in enumerate(ranges):
            b = _Bucket()
            b.flat = engine.grads[lo:hi]
            b.params = ps
            b.offsets = [engine.offsets[id(p)] - lo for p in ps]
            self.buckets.append(b)
            for p, off in zip(b.params, b.offsets):
                self._where[p] = (bi, off)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        engine.ready_callbacks.append(self._on_grad)         # gradients the producers write in place bypass autograd

    # -- per step -----------------------------------------------------------------------------------------------------
    def prepare(self):
        if len(self._issued) > 4 * max(len(self.buckets), 1):    # handles are only kept until they are known complete
            self._issued = [w for w in self._issued if not w.is_completed()]
        self._seen = set()
        self._next = 0
        self._sent_bytes, self._sent_buckets = 0, 0
        for b in self.buckets:
            b.work = None
            b.wire = None
            b.pending = sum(1 for p in b.params if p not in self._unused)
            b.expected = b.pending

    def _launch(self, b: _Bucket):
        if self.engine is not None:
            self.engine.flush()                              # batched copy of the small gradients into the flat buffer
        self._sent_bytes += b.flat.numel() * torch.empty((), dtype=self.grad_dtype).element_size()
        self._sent_buckets += 1
        if self.collective:
            # RCCL averages inside the collective (ncclAvg); gloo has no AVG, finish() scales there.  Not inside a graph capture: SUM is
            # the collective every graph user runs; RCCL's one-rank ncclAvg (a pre-multiply kernel) replayed with a stale scalar on this
            # stack (round 5: losses and weights off from the second replay on, SUM or no collective bitwise), so a captured step sums
            # and finish() scales.  (Exact either way for a power-of-two world.)
            self._avg_in_op = self.average and self.world > 1 and dist.get_backend(self.group) == "nccl" and \
                not (b.flat.is_cuda and torch.cuda.is_current_stream_capturing())
            buf = b.flat
            if self.grad_dtype != torch.float32:
                b.wire = b.flat.to(self.grad_dtype)
                buf = b.wire
            b.work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg_in_op else dist.ReduceOp.SUM,
                                     group=self.group, async_op=True)
            if not (buf.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._issued.append(b.work)
        else:
            b.work = True

    def _launch_ready(self):
        """Send every complete bucket at the head of the queue (strict order: same collective sequence on all ranks)."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.pending != 0:                               # incomplete: it and everything behind it wait
                break
            self._launch(b)
            self._next += 1

    def _on_grad(self, p: torch.nn.Parameter):
        if p.grad is not None and p.grad.is_cuda:
            from . import ops
            ops.join_wgrad_stream()                          # the gradient may come from the side-stream wgrad GEMM
        seen = getattr(self, "_seen", None)
        if seen is not None:
            # a gradient published in place (engine.grads_written_in_place) reports here directly; this torch also runs
            # the post-accumulate hook for the `None` the producer then hands to autograd -- count each parameter once
            if id(p) in seen:
                return
            seen.add(id(p))
        bi, off = self._where[p]
        b = self.buckets[bi]
        if p in self._unused:
            raise RuntimeError(
                "GradAllReducer(static_unused=True): a parameter marked unused on the first step received a gradient "
                "later (shape %s).  Its bucket may already be in flight without it.  Construct the reducer with "
                "static_unused=False for models whose set of used parameters varies between steps or ranks." %
                (tuple(p.shape),))
        if self.engine is None:                              # (engine: its own hook ran first and queued the copy)
            view = b.flat[off:off + p.numel()].view_as(p)
            if p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view                                # the optimizer reads the reduced bucket in place
        b.pending -= 1
        self._launch_ready()

    def finish(self):
        """Send what backward could not complete (buckets holding parameters without a gradient), in order; wait for every
        transfer on the current stream; average.  With static_unused=False also agree on the set of parameters that got a
        gradient on ANY rank and hand those their (averaged) gradient on every rank."""
        capturing = bool(self.buckets) and self.buckets[0].flat.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing:
            ok, why = self.capturable()
            if not ok:
                raise RuntimeError("GradAllReducer.finish() inside a graph capture: " + why)
        if self.engine is not None:
            self.engine.flush()
        for b in self.buckets[self._next:]:
            for p, off in zip(b.params, b.offsets):
                if p.grad is None or p.grad.data_ptr() != b.flat[off:off + 1].data_ptr():
                    b.flat[off:off + p.numel()].zero_()      # absent gradient: this rank contributes zeros
                    if self._first and self.static_unused:
                        self._unused.add(p)
            self._launch(b)
        self._next = len(self.buckets)
        if self._first and self.static_unused and self.world > 1:
            # the static set must be the SAME on every rank: agree once.  A parameter without a gradient on every rank is
            # excluded for good; if the ranks disagree on any parameter (per-rank modality subsets, ...), a static set
            # would let replicas drift apart (one rank updates the parameter, another does not): switch to the
            # per-step agreement of static_unused=False.
            loc = torch.tensor([[0, 1] if p.grad is None else [1, 0] for p in self.params], dtype=torch.int32,
                               device=self.buckets[0].flat.device)
            dist.all_reduce(loc, op=dist.ReduceOp.MAX, group=self.group)
            any_used, any_unused = loc[:, 0].tolist(), loc[:, 1].tolist()
            if any(u and n for u, n in zip(any_used, any_unused)):
                import warnings
                warnings.warn("GradAllReducer: ranks disagree on the set of parameters that received a gradient on the "
                              "first step; switching to static_unused=False (per-step agreement)")
                self.static_unused = False
                self._unused = set()
            else:
                self._unused = {p for p, u in zip(self.params, any_used) if not u}
        used = None
        if not self.static_unused and self.world > 1:
            flags = torch.tensor([0 if p.grad is None else 1 for p in self.params], dtype=torch.int32,
                                 device=self.buckets[0].flat.device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            used = flags.tolist()                            # host sync, like DDP's local_used_map copy
        on_gpu = self.buckets and self.buckets[0].flat.is_cuda and not capturing     # (timing events cannot be read back from a graph)
        if on_gpu:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        elif not capturing:
            import time
            t0 = time.perf_counter()
        for b in self.buckets:
            if b.work is not True and b.work is not None:
                b.work.wait()
        if on_gpu:
            ev1.record()
        elif not capturing:
            self.stats["comm_exposed_ms"] = (time.perf_counter() - t0) * 1e3
        for b in self.buckets:
            if b.wire is not None:
                b.flat.copy_(b.wire)
                b.wire = None
            if self.average and self.world > 1 and not getattr(self, "_avg_in_op", False):
                b.flat.mul_(1.0 / self.world)
        self._pending_events = (ev0, ev1) if on_gpu else None
        self.stats["allreduce_bytes"], self.stats["buckets"] = self._sent_bytes, self._sent_buckets
        if self.static_unused and not self._first:
            # A parameter outside the agreed unused set that got no gradient on THIS rank this step (a branch not taken here):
            # its slot went out as zeros and came back as the average of the other ranks' gradients.  Hand it to the optimizer
            # like everywhere else -- every rank then applies the same update (replicas cannot drift) without a collective.
            for p in self.params:
                if p.grad is None and p not in self._unused:
                    bi, off = self._where[p]
                    p.grad = self.buckets[bi].flat[off:off + p.numel()].view_as(p)
        if used is not None:
            for p, u in zip(self.params, used):
                if u and p.grad is None:                     # used on another rank: take part in the update here too
                    bi, off = self._where[p]
                    p.grad = self.buckets[bi].flat[off:off + p.numel()].view_as(p)
        self._first = False

    def capturable(self):
        """-> (ok, reason).  A step with this reducer can be captured into a hipGraph (pretrain.PretrainStep.capture) once nothing in
        finish() needs the host: the static unused set is agreed (one eager step has run) and stays static.  The collectives themselves
        are captured as graph nodes on RCCL's stream; the bucket bookkeeping runs once, at capture, and the replays repeat its launches.
        Averaging: an eager step averages inside the collective (ncclAvg), a captured one sums and scales by 1 / world afterwards --
        bitwise the same for a power-of-two world, one rounding apart otherwise."""
        if not self.static_unused:
            return False, "static_unused=False agrees on the used parameters through the host every step"
        if self._first:
            return False, "run one eager step first (the unused-parameter set is agreed on it)"
        if self.collective and dist.get_backend(self.group) != "nccl":
            return False, "only RCCL collectives can be captured (backend %s)" % dist.get_backend(self.group)
        return True, ""

    def quiesce(self, timeout_s: float = 60.0) -> dict:
        """Deterministic hand-off before a hipGraph capture (pretrain.PretrainStep.capture): returns once every collective issued so far
        is (1) complete on the device -- the kept Work handles are waited for and polled with is_completed() --, (2) issued and complete
        on EVERY rank -- a barrier on the reducer's group, then a device synchronisation --, and (3) RETIRED by the process group's
        watchdog thread, which polls the end events of outstanding collectives: c10d's flight recorder lists a collective as active
        until the watchdog has seen it complete, so the wait ends when that list is empty.  Where the recorder is switched off
        (TORCH_NCCL_TRACE_BUFFER_SIZE=0) step (3) falls back to three watchdog periods AFTER (1) and (2) have established completion.
        Raises TimeoutError instead of capturing into a live poll.  -> {'handles', 'method', 'waited_s'} (method: none / gloo /
        flight_recorder / watchdog_periods).  (capture() then sets `captured`: stats / exposed_ms() describe no replayed step.)"""
        import time
        t0 = time.monotonic()
        info = {"handles": len(self._issued), "method": "none", "waited_s": 0.0}
        if not self.collective:
            return info

        def expired():
            return time.monotonic() - t0 > timeout_s
        for w in self._issued:
            w.wait()
        while not all(w.is_completed() for w in self._issued):
            if expired():
                raise TimeoutError("GradAllReducer.quiesce(): collectives of the warm-up steps did not complete in %.0f s" % timeout_s)
            time.sleep(0.001)
        self._issued = []
        dist.barrier(group=self.group)
        if self.buckets and self.buckets[0].flat.is_cuda:
            torch.cuda.synchronize(self.buckets[0].flat.device)
        if dist.get_backend(self.group) != "nccl":
            info["method"] = "gloo"
        else:
            c10d = torch._C._distributed_c10d
            dump = getattr(c10d, "_dump_nccl_trace", None)

            def entries(only_active):
                import pickle
                return len(pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=only_active)).get("entries", []))
            recorder = False
            try:
                recorder = dump is not None and entries(False) > 0      # at least the barrier above is on record when it is on
            except Exception:
                recorder = False
            if recorder:
                info["method"] = "flight_recorder"
                while entries(True) > 0:
                    if expired():
                        raise TimeoutError("GradAllReducer.quiesce(): the process group's watchdog has not retired %d completed "
                                           "collectives after %.0f s" % (entries(True), timeout_s))
                    time.sleep(0.005)
            else:
                info["method"] = "watchdog_periods"
                time.sleep(0.3)                              # 3 x the watchdog's 100 ms pass, behind established completion
        info["waited_s"] = round(time.monotonic() - t0, 4)
        self.last_quiesce = info
        return info

    def exposed_ms(self) -> Optional[float]:
        """Time the compute stream waited for the collectives in the last finish() (host sync on the GPU path).  None once the step
        is captured (quiesce()): graph replays record no events and update no stats."""
        if self.captured:
            return None
        ev = getattr(self, "_pending_events", None)
        if ev is not None:
            ev[1].synchronize()
            self.stats["comm_exposed_ms"] = ev[0].elapsed_time(ev[1])
            self._pending_events = None
        return self.stats["comm_exposed_ms"]

    def unused_parameters(self):
        return list(self._unused)

    def remove(self):
        for h in self._hooks:
            h.remove()
