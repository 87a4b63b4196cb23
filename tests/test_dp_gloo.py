"""N > 1 path on CPU: two gloo ranks run the bucketed, hook-driven gradient all-reduce (incomplete_multimodal_fusion_amd/dp.py)
on a small torch module and must end with the single-process gradient of the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 32)
        self.unused = torch.nn.Parameter(torch.ones(7))      # never receives a gradient (like return_tokens in the model)
        self.c = torch.nn.Linear(32, 4)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from incomplete_multimodal_fusion_amd import dp
    assert dp.init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = Net()
    red = dp.GradAllReducer(net.parameters(), bucket_bytes=3000)      # several buckets
    assert len(red.buckets) >= 3
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    res = []
    for step in range(3):                                              # step 0 detects the unused parameter
        net.zero_grad(set_to_none=True)
        red.prepare()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        red.finish()
        res.append({n: p.grad.tolist() for n, p in net.named_parameters() if p.grad is not None})
    unused = [n for n, p in net.named_parameters() if any(p is u for u in red.unused_parameters())]
    launched_early = sum(1 for b in red.buckets if b.expected > 0)
    q.put((rank, res, unused, launched_early))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_bucketed_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = Net()
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    ((net(X) - Y) ** 2).mean().backward()                             # mean over the global batch == average of rank means
    for rank, res, unused, launched in out:
        assert unused == ["unused"]
        for step_grads in res:
            for n, p in net.named_parameters():
                if n == "unused":
                    continue
                assert torch.allclose(torch.tensor(step_grads[n]), p.grad, atol=1e-6), (rank, n)


class Branchy(torch.nn.Module):
    """The set of parameters that receive a gradient depends on `use_side` (like ViTBaseline's random modality subset)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.side = torch.nn.Linear(16, 32)
        self.c = torch.nn.Linear(32, 4)

    def forward(self, x, use_side):
        h = self.a(x)
        if use_side:
            h = h + self.side(x)
        return self.c(torch.relu(h))


# (step, rank) -> does this rank's forward use the side branch?  step 0: nobody; step 1: rank 1 only; step 2: both
_SIDE = {(0, 0): False, (0, 1): False, (1, 0): False, (1, 1): True, (2, 0): True, (2, 1): True}


def _worker_dynamic(rank, world, port, q, static):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from incomplete_multimodal_fusion_amd import dp
    assert dp.init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = Branchy()
    red = dp.GradAllReducer(net.parameters(), bucket_bytes=1500, static_unused=static)
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    res, err = [], None
    for step in range(3):
        net.zero_grad(set_to_none=True)
        red.prepare()
        loss = ((net(xs, _SIDE[(step, rank)]) - ys) ** 2).mean()
        try:
            loss.backward()
            red.finish()
        except RuntimeError as e:
            err = str(e)
            break
        res.append({n: (None if p.grad is None else p.grad.tolist()) for n, p in net.named_parameters()})
    q.put((rank, res, err))
    if err is None:
        dist.barrier()
    dist.destroy_process_group()


def _run_dynamic(static):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dynamic, args=(r, 2, port, q, static)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
    return out


@pytest.mark.timeout(120)
def test_used_parameter_set_changes_between_steps_and_ranks():
    """static_unused=False: every step equals the average of the two ranks' local gradients (zeros where a rank did not
    use a parameter), and a parameter used on ANY rank gets its gradient on EVERY rank (replicas stay in sync)."""
    out = _run_dynamic(static=False)
    for step in range(3):
        locals_ = []
        for rank in range(2):
            torch.manual_seed(0)
            net = Branchy()
            torch.manual_seed(100)
            X = torch.randn(8, 16); Y = torch.randn(8, 4)
            xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
            ((net(xs, _SIDE[(step, rank)]) - ys) ** 2).mean().backward()
            locals_.append({n: p.grad for n, p in net.named_parameters()})
        for rank, res, err in out:
            assert err is None
            for n in locals_[0]:
                gs = [l[n] for l in locals_]
                got = res[step][n]
                if all(g is None for g in gs):
                    assert got is None, (step, rank, n)                  # unused everywhere: stays without a gradient
                    continue
                want = sum(torch.zeros_like(next(x for x in gs if x is not None)) if g is None else g for g in gs) / 2
                assert got is not None, (step, rank, n)
                assert torch.allclose(torch.tensor(got), want, atol=1e-6), (step, rank, n)


@pytest.mark.timeout(120)
def test_static_unused_raises_when_an_excluded_parameter_gets_a_gradient():
    out = _run_dynamic(static=True)
    errs = [err for _, _, err in out]
    assert any(e and "static_unused=False" in e for e in errs), errs


# first step: rank 1 uses the side branch, rank 0 does not -> a per-rank static set would let the replicas drift apart
_SIDE_MIXED = {(0, 0): False, (0, 1): True, (1, 0): True, (1, 1): False, (2, 0): True, (2, 1): True}


@pytest.mark.timeout(120)
def test_static_unused_switches_to_per_step_agreement_when_ranks_disagree_on_the_first_step():
    global _SIDE
    saved = dict(_SIDE)
    try:
        _SIDE.clear(); _SIDE.update(_SIDE_MIXED)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_mixed, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        out = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=30)
    finally:
        _SIDE.clear(); _SIDE.update(saved)
    (r0, res0, err0), (r1, res1, err1) = out
    assert err0 is None and err1 is None, (err0, err1)
    for step in range(3):
        for n in res0[step]:
            assert res0[step][n] is not None and res1[step][n] is not None, (step, n)     # used somewhere -> present everywhere
            assert torch.allclose(torch.tensor(res0[step][n]), torch.tensor(res1[step][n]), atol=1e-7), (step, n)


def _worker_mixed(rank, world, port, q):
    global _SIDE
    _SIDE.clear(); _SIDE.update(_SIDE_MIXED)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _worker_dynamic(rank, world, port, q, True)


# static set agreed on step 0 (both ranks use the side branch); on step 1 rank 0 does NOT take it: its slot goes out as zeros
_SIDE_LATE = {(0, 0): True, (0, 1): True, (1, 0): False, (1, 1): True, (2, 0): True, (2, 1): False}


def _worker_late(rank, world, port, q, grad_dtype):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from incomplete_multimodal_fusion_amd import dp
    assert dp.init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = Branchy()
    red = dp.GradAllReducer(net.parameters(), bucket_bytes=1500, static_unused=True,
                            grad_dtype=torch.bfloat16 if grad_dtype == "bf16" else torch.float32)
    torch.manual_seed(100)
    X = torch.randn(8, 16); Y = torch.randn(8, 4)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    res = []
    for step in range(3):
        net.zero_grad(set_to_none=True)
        red.prepare()
        ((net(xs, _SIDE_LATE[(step, rank)]) - ys) ** 2).mean().backward()
        red.finish()
        res.append(({n: (None if p.grad is None else p.grad.tolist()) for n, p in net.named_parameters()}, dict(red.stats),
                    red.exposed_ms()))
    q.put((rank, res, red.static_unused))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
def test_static_set_parameter_without_a_local_gradient_gets_the_average_on_every_rank(grad_dtype):
    """with the static unused set, a parameter outside it that gets no gradient on ONE rank in a later step must
    still be updated there with the all-reduced average (else the replicas drift).  Also: the per-step `stats`, and
    grad_dtype=bf16 buckets -- the averaged gradient within bf16 rounding of the fp32 one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_late, args=(r, 2, port, q, grad_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    nparams = sum(p.numel() for p in Branchy().parameters())
    for step in range(3):
        locals_ = []
        for rank in range(2):
            torch.manual_seed(0)
            net = Branchy()
            torch.manual_seed(100)
            X = torch.randn(8, 16); Y = torch.randn(8, 4)
            xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
            ((net(xs, _SIDE_LATE[(step, rank)]) - ys) ** 2).mean().backward()
            locals_.append({n: p.grad for n, p in net.named_parameters()})
        for rank, res, static in out:
            assert static is True                                       # the ranks agreed on step 0: the set stays static
            grads, stats, exposed = res[step]
            assert stats["buckets"] >= 2 and stats["allreduce_bytes"] == nparams * (2 if grad_dtype == "bf16" else 4)
            assert exposed >= 0.0
            for n in locals_[0]:
                gs = [l[n] for l in locals_]
                want = sum(torch.zeros_like(next(x for x in gs if x is not None)) if g is None else g for g in gs) / 2
                assert grads[n] is not None, (step, rank, n)            # ... also where this rank had no local gradient
                got = torch.tensor(grads[n])
                if grad_dtype == "fp32":
                    assert torch.allclose(got, want, atol=1e-6), (step, rank, n)
                else:                                                   # two bf16 roundings (cast, sum): 2^-8 relative each
                    assert torch.allclose(got, want, atol=2e-2 * float(want.abs().max()) + 1e-6), (step, rank, n)
        assert out[0][1][step][0] == out[1][1][step][0], "replicas hold different gradients"
