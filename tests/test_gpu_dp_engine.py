"""N > 1 path with the real model on the GPU: two ranks (gloo rendezvous, both on cuda:0 -- RCCL refuses two ranks on one
device) run PretrainStep with the flat AdamW engine and the in-place bucketed all-reduce of its gradient buffer.  The
reduced gradients must equal the average of the two single-process gradients, and both ranks must hold identical weights
after the update."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from tests.test_dp_gloo import ROOT, _free_port

pytestmark = pytest.mark.gpu


def _setup(rank):
    from incomplete_multimodal_fusion_amd.pretrain import get_model
    torch.manual_seed(7)                                              # same initial weights on every rank
    model = get_model("tiny", input_size=64, decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    model.depth = 2; model.blocks = model.blocks[:2]; model.fus_blocks = model.fus_blocks[:2]
    model.to("cuda:0").train()
    g = torch.Generator().manual_seed(1000 + rank)                   # different data and masks per rank
    B, P = 4, 16
    x = {"s1": torch.randn(B, 1, 64, 64, generator=g).cuda(), "s2": torch.randn(B, 3, 64, 64, generator=g).cuda(),
         "dem": torch.randn(B, 1, 64, 64, generator=g).cuda()}
    masks = {}
    for d, k in (("s1", 10), ("s2", 8), ("dem", 6)):
        row = torch.ones(P, dtype=torch.long); row[torch.randperm(P, generator=g)[:k]] = 0
        masks[d] = row[None].repeat(B, 1).cuda()
    return model, x, masks


def _grads(rank, reducer_world):
    """One fp32 PretrainStep with lr = 0 (gradients only); returns {name: grad list}."""
    from incomplete_multimodal_fusion_amd import dp
    from incomplete_multimodal_fusion_amd.engine import FlatAdamW
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    model, x, masks = _setup(rank)
    opt = FlatAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    red = dp.GradAllReducer(None, bucket_bytes=1 << 18, engine=opt) if reducer_world > 1 else None
    step = PretrainStep(model, opt, 24, autocast=False, grad_reducer=red)
    step(x, task_masks=masks)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    params = {n: p.detach().cpu() for n, p in model.named_parameters()}
    return grads, params, (len(red.buckets) if red else 0)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from incomplete_multimodal_fusion_amd import dp
    assert dp.init_distributed(backend="gloo")
    grads, params, nb = _grads(rank, world)
    q.put((rank, {n: g.tolist() for n, g in grads.items()}, {n: p.flatten()[:64].tolist() for n, p in params.items()}, nb))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_engine_flat_allreduce_matches_average_of_single_process_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    singles = [_grads(r, 1)[0] for r in range(2)]
    assert out[0][3] >= 3, "several buckets"
    for rank, grads, params, _ in out:
        assert grads.keys() == singles[0].keys()
        for n in grads:
            want = 0.5 * (singles[0][n] + singles[1][n])
            got = torch.tensor(grads[n])
            assert torch.allclose(got, want, rtol=1e-4, atol=1e-6 + 1e-4 * float(want.abs().max())), (rank, n)
    for n in out[0][2]:                                               # same update on both ranks
        assert out[0][2][n] == out[1][2][n], n


def test_bench_step_over_rccl_single_rank_matches_the_undistributed_step():
    """The N > 1 code path on real RCCL with the one GPU this box has: bench.py under torch.distributed.run with ONE rank and
    MMAE_DIST_SINGLE_RANK=1 -- backend "nccl" (= RCCL), GradAllReducer over the engine's flat gradient buffer with
    ReduceOp.AVG launched from the backward hooks on RCCL's stream, barrier + MAX-over-ranks timing -- must run and give the
    same loss trajectory as the same seeds without a process group (a world-1 average is the identity)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--legs", "none",
              "--tunable", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MMAE_DIST_BACKEND", None)
    plain = subprocess.run([sys.executable] + common, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ranked = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                             "--master-addr", "127.0.0.1", "--master-port", str(port)] + common, cwd=root,
                            env=dict(env, MMAE_DIST_SINGLE_RANK="1"), capture_output=True, text=True, timeout=600)
    assert ranked.returncode == 0, ranked.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in ranked.stdout.splitlines() if l.startswith("{")][-1])
    assert a["backend"] == "none" and b["backend"] == "nccl" and b["ranks"] == 1 and b["n_gpus"] == 1
    assert abs(a["config"]["loss"] - b["config"]["loss"]) <= 1e-3 * abs(a["config"]["loss"]), (a["config"]["loss"], b["config"]["loss"])


def test_captured_step_with_the_rccl_reducer_replays_like_eager():
    """PretrainStep.capture with a GradAllReducer: the bucket all-reduces become graph nodes on RCCL's stream (one rank on this box's one
    GPU, every collective really issued: tests/dp_capture_worker.py).  The replays must repeat the eager steps bit for bit."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), MMAE_DIST_SINGLE_RANK="1")
    env.pop("MMAE_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dp_capture_worker.py")], cwd=root, env=env, capture_output=True,
                       text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert not d["capturable_before_first_step"] and "eager step" in d["why"]
    assert d["sent_buckets"] == d["buckets"] > 2
    assert d["losses_replay"] == d["losses_eager"], d
    # the hand-off into the capture was deterministic (no fixed sleep): completion established, then the watchdog's retirement observed
    assert d["quiesce"]["method"] in ("flight_recorder", "watchdog_periods") and d["quiesce"]["handles"] > 0, d["quiesce"]
    assert d["exposed_after_capture"] is None
    assert d["master_equal"] and d["exp_avg_equal"] and d["steps"] == [6, 6]
