#!/usr/bin/env python
"""Headline benchmark: pretrain samples/s of the native fusion-token path, ViT-B / 3 modalities / 256x256 tiles.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one rank per GPU.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N`: RANK / WORLD_SIZE / MASTER_* in the environment), or -- when WORLD_SIZE is absent -- this process starts
them itself as a CHILD torch.distributed.run before it has made any GPU call (launch_ranks(); the reference does the
same job with torch.distributed.launch + utils/dist.py:62-93), forwards rank 0's JSON line and exits with the child's code.

A step = Dirichlet mask draw + patch embedding of the kept patches + 12 x (Block_Fusion + Zorro-masked Block) + final
norm + pooling + 3 decoders + masked MSE/L1 + 3 DINO-style contrastive terms + backward + gradient all-reduce (N > 1) +
AdamW, on synthetic tiles already resident in HBM (BASELINE.md / SURVEY.md 8d).  Prints ONE JSON line on rank 0.

`value` is the resident-input rate of the headline configuration (batch-shared Dirichlet masks, C4 of SURVEY 8d).  At
N = 1 the same invocation also times two secondary legs on the same model and reports them beside it (never as
`value`): `pcie_inclusive` -- every step takes a fresh RAW host batch through staging.TileStager (SURVEY 8d's step
includes the H2D leg) -- and `c3_per_sample_dropout` -- per-sample packed masks with random modality dropout
(`sample_tasks_uniformly`), the variable-token-count configuration north_star describes.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak
PROFILE_ROUNDS = ("r05", "r04", "r03", "r02", "r01")   # newest committed rocprofv3 summaries first


# ---------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv) -> int:
    """Start `args.gpus` ranks of this script under torch.distributed.run as a child process and return its exit code.
    Called only while this process is still GPU-free (nothing but `import torch` has run), so no process that has
    initialised HIP is ever replaced or forked."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC (RCCL over xGMI on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def domain_table(args):
    """(domains, channels per domain, classes per domain or None) of the run: the 3-modality table of pretrain_mmae.py:45-72,
    or the 4-modality one of pretrain_mmae_my.py:46-81 when 'dnw' is among the domains (s1 2 ch, s2 4 ch, class map)."""
    from incomplete_multimodal_fusion_amd.pretrain import DOMAIN_CONF, DOMAIN_CONF_QUAD
    doms = tuple(d for d in args.domains.split(",") if d)
    conf = DOMAIN_CONF_QUAD if "dnw" in doms else DOMAIN_CONF
    return doms, [conf[d]["channels"] for d in doms], [conf[d].get("num_classes") for d in doms]


def build(args, device):
    from incomplete_multimodal_fusion_amd.pretrain import get_model
    torch.manual_seed(1234)
    doms, _, _ = domain_table(args)
    model = get_model(args.model, in_domains=doms, input_size=args.input_size, patch_size=16, decoder_dim=256,
                      decoder_depth=2, decoder_num_heads=8, fusion_blocks=bool(args.fusion_blocks))
    return model.to(device).train()


def synthetic_tiles(args, B, size, device, seed):
    """i.i.d. N(0,1) tiles (the datasets are z-scored), a class map in [0, classes) for a class-map modality."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    doms, chans, classes = domain_table(args)
    x = {}
    for d, c, k in zip(doms, chans, classes):
        x[d] = (torch.randint(0, k, (B, size, size), generator=g) if k else torch.randn(B, c, size, size, generator=g)).to(device)
    return x


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The oracle (CPU restatement pinned to the reference by tests/golden) timed on this box's host cores: same model,
    same step definition (fwd + losses + bwd + AdamW), fp32, a bounded sample of the workload (B = 8, >= 3
    timed steps when the host manages them inside ~60 s; SURVEY 8d asks for >= 5, the default)."""
    from oracle import mmae_oracle as O
    from incomplete_multimodal_fusion_amd.pretrain import get_model
    O.set_fused_primitives(True)     # LayerNorm / GELU via the fused functionals the reference calls: with them the port costs
    try:                             # 0.98 x the actual reference step on the build container (tools/cpu_port_vs_reference.py)
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))                   # a 1-GPU box owns 16 host cores; never oversubscribe
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    model = get_model(args.model, input_size=args.input_size)
    p = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("pos_emb") and not k.endswith("beta"))
         for k, v in model.state_dict().items()}
    del model
    B, P = args.cpu_batch, (args.input_size // 16) ** 2
    N = args.num_encoded_tokens
    x = synthetic_tiles(args, B, args.input_size, "cpu", 99)
    opt = torch.optim.AdamW([t for t in p.values() if t.requires_grad], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    heads = {"tiny": 3}.get(args.model, 8)
    from torch.distributions.dirichlet import Dirichlet
    times, t_begin = [], time.perf_counter()
    for it in range(args.cpu_steps + 1):
        t0 = time.perf_counter()
        d = Dirichlet(torch.ones(3)).sample((1,)); noise = torch.rand(1, 3, P); na = torch.rand(1, 3 * P)
        mask_all, _, _ = O.masks_from_draws(d, noise, na, N)
        masks = {dom: mask_all[:, i * P:(i + 1) * P].repeat(B, 1) for i, dom in enumerate(O.DOMAINS)}
        out, (_, _, loss) = O.train_step_loss(p, x, masks, N, heads, 8, 16)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        dt_ = time.perf_counter() - t0
        print("[cpu_baseline] step %d: %.2f s (%d threads)" % (it, dt_, cores), file=sys.stderr, flush=True)
        if it > 0 or dt_ > 20.0:                     # a slow host: keep the (warm-up) step as the sample and stop
            times.append(dt_)
        if time.perf_counter() - t_begin > 60.0 and times:
            break
    t = sum(times) / len(times)
    return {"value": round(B / t, 4), "unit": "samples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
            "sample": "%s 3-mod %dx%d, B=%d, N=%d, fp32, fwd+bwd+AdamW, %d timed step(s), %.2f s/step" %
                      (args.model, args.input_size, args.input_size, B, N, len(times), t)}


def step_flops(args, per_layer=False):
    """Algorithmic FLOPs of one optimizer step per sample, SURVEY.md 8(d): MACs_fwd of the reference-dense formulation
    (Block + Block_Fusion as written + pooling heads + patch embedding + decoders), FLOPs_step = 6 * MACs_fwd (forward 2x,
    backward 4x, no recompute credit); and the same with Block_Fusion as executed here (K/V of every row once, query /
    output / FF of the fusion slot only).  Everything follows the run's arguments (model preset, domains and their channels,
    tile size, kept tokens, fusion blocks, contrastive head).  ViT-B 3-modality defaults: 539.8 GF reference-dense."""
    D = {"tiny": 192, "small": 384, "base": 768, "large": 1024}[args.model]
    L = {"tiny": 12, "small": 12, "base": 12, "large": 24}[args.model]
    h = {"tiny": 3}.get(args.model, 8)
    doms, C, classes = domain_table(args)
    M = len(doms)
    I, ffi, Dd, Ld, p = 64 * h, int(D * 8 / 3), 256, 2, 16
    P = (args.input_size // p) ** 2
    N = args.num_encoded_tokens
    S = N + P
    block = L * (4 * S * D * I + 2 * S * S * I + 3 * S * D * ffi)
    fus_dense = L * (4 * P * (M + 1) * D * I + 2 * P * (M + 1) ** 2 * I + 3 * P * D * ffi) if args.fusion_blocks else 0
    fus_exec = L * (2 * S * D * I + 2 * P * D * I + 2 * P * (M + 1) * I + 3 * P * D * ffi) if args.fusion_blocks else 0
    pool = (M + 1) * 2 * D * I + 2 * S * D * I + 2 * (M + 1) * S * I + 8 * (M + 1) * D * D
    ctr = (M * (2 * D * I + 8 * D * D) + 2 * N * D * I) if (args.contra == "dino" and args.fusion_blocks) else 0
    embed = sum(P * (64 if k else c) * p * p * D for c, k in zip(C, classes))     # class map: 64-wide class embedding per pixel
    dec = sum(P * D * Dd + Ld * (4 * P * Dd * Dd + 2 * P * P * Dd + 8 * P * Dd * Dd) + P * Dd * c * p * p for c in C)
    rest = pool + ctr + embed + dec
    if per_layer:                                   # one encoder layer (Block_Fusion + Block), forward + backward: bench's roofline_block
        return 6.0 * (block + fus_dense) / L, 6.0 * (block + fus_exec) / L
    return 6.0 * (block + fus_dense + rest), 6.0 * (block + fus_exec + rest)


def workload_string(args, mask_desc):
    D = {"tiny": 192, "small": 384, "base": 768, "large": 1024}[args.model]
    L = {"tiny": 12, "small": 12, "base": 12, "large": 24}[args.model]
    h = {"tiny": 3}.get(args.model, 8)
    doms, C, _ = domain_table(args)
    losses = {"dino": "MSE+L1+0.3*DINO", "hardneg": "task losses + hard-negative contrastive head", "none": "task losses only"}[args.contra]
    return ("ViT-%s (D%d/L%d/h%dx64) %d-modality (%s) %dx%d tiles, patch 16, N=%d of %d tokens kept (%s Dirichlet alpha=1 masks "
            "per step), %s, decoders 256/2/8, %s, fwd+bwd+AdamW"
            % (args.model, D, L, h, len(doms), "+".join("%s:%dch" % (d, c) for d, c in zip(doms, C)), args.input_size,
               args.input_size, args.num_encoded_tokens, len(doms) * (args.input_size // 16) ** 2, mask_desc,
               "fusion blocks" if args.fusion_blocks else "no fusion blocks (multimae_quadruplet)", losses))


def replayed_from(stem, this_ms):
    """Provenance of a counter figure that cannot be collected in-process and is therefore REPLAYED from a committed rocprofv3
    summary of this command: the file, the commit whose tree the profiled run measured (stamped by tools/stamp_profiles.py when the
    summary was copied into profiles/), the kernel time per step of the profiled run -- and whether it still describes THIS run:
    `valid` is False when the two step times differ by more than 5 % (the replayed fields are then printed as null)."""
    js, rnd = _profile_json(stem)
    if js is None:
        return None, False
    prof_ms = (js.get("step") or {}).get("kernel_ms_per_step") or js.get("kernel_ms_per_step")
    ok = bool(prof_ms) and abs(prof_ms - this_ms) <= 0.05 * this_ms
    return {"file": "profiles/%s_%s.json" % (rnd, stem), "git": js.get("git"), "profiled_ms_per_step": prof_ms,
            "this_run_ms_per_step": round(this_ms, 2), "valid": ok}, ok


def gemm_roofline():
    """The dominant kernel of the step BY TIME is a library GEMM (hipBLASLt / Tensile through torch).  Its name, share of the
    step, MFMA-busy % and held clock come from the committed rocprofv3 summaries of this command (profiles/rNN_sq_step.json:
    the counters cannot be collected in-process); None when absent."""
    js, rnd = _profile_json("sq_step")
    try:
        ks = {n: k for n, k in js["kernels"].items() if n.startswith(("Cijk", "Custom_Cijk")) or "gemm8p_kernel" in n or "gemm_tn8p" in n}
        name, top = max(ks.items(), key=lambda kv: kv[1]["ms"])
        steps, step_ms = js["step"]["steps"], js["step"]["kernel_ms_per_step"]
        gemm_ms = sum(k["ms"] for k in ks.values())
        busy = sum(k["mfma_busy_pct"] * k["ms"] for k in ks.values()) / max(gemm_ms, 1e-9)
        clock = top.get("clock_ghz") or 2.4
        return {"kernel": name[:80], "bound": "mfma", "dominant_ms_per_step": round(top["ms"] / steps, 2),
                "dominant_mfma_busy_pct": round(top["mfma_busy_pct"], 1), "dominant_clock_ghz": clock,
                "achieved": round(top["mfma_busy_pct"] / 100.0 * MFMA_BF16_PEAK_TF * clock / 2.4, 1), "peak": MFMA_BF16_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(top["mfma_busy_pct"] / 100.0 * clock / 2.4, 3),
                "all_gemms_ms_per_step": round(gemm_ms / steps, 2), "all_gemms_share_of_step": round(gemm_ms / steps / step_ms, 3),
                "all_gemms_mfma_busy_pct": round(busy, 1), "source": "profiles/%s_sq_step.json" % rnd}
    except Exception:
        return None


def _profile_json(stem):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, stem))
        if os.path.isfile(path):
            try:
                return json.load(open(path)), rnd
            except Exception:
                pass
    return None, None


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the roofline kernel from the committed PMC summary of THIS command (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied; see profiles/rNN_pmc_hbm.json) -- the counters
    cannot be collected from inside the process.  None when the summary is absent."""
    js, _ = _profile_json("pmc_hbm")
    try:
        cand = [v for name, v in js["kernels"].items() if kernel_substr in name]
        if cand:                                            # several template instances: the one the step launches most
            v = max(cand, key=lambda c: c["launches"] * (c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]))
            return round(v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"])
    except Exception:
        pass
    return None


def mfma_busy():
    """Step-level MFMA-busy % (the second half of BASELINE.json's metric) from the committed SQ counter summary of this
    command (profiles/rNN_sq_step.json, tools/prof_summary.py); None when absent."""
    js, rnd = _profile_json("sq_step")
    try:
        return {"pct": js["step"]["mfma_busy_pct"], "source": "profiles/%s_sq_step.json" % rnd}
    except Exception:
        return None


def _counter_pass(args, counters, timeout_s):
    """One child `rocprofv3 --pmc <counters> -- python3 bench.py --steps 2 --warmup 2 --legs none` of this command (its own process, run BEFORE
    this process touches the GPU): -> (rows of counter_collection.csv, seconds).  Raises on failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.isfile("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        raise FileNotFoundError("rocprofv3")
    out_dir = tempfile.mkdtemp(prefix="mmae_pmc_", dir="/tmp")
    cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "--",
           sys.executable or "python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--legs", "none",
           "--block-timer", "0", "--batch", str(args.batch)]
    t0 = time.perf_counter()
    try:
        # its own process group: on a timeout the profiler AND the bench process under it are killed (an orphaned pass would share the card
        # with the timed region)
        proc = subprocess.Popen(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                start_new_session=True)
        try:
            _, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            raise RuntimeError("rocprofv3 pass did not finish in %d s (killed)" % timeout_s)
        if proc.returncode != 0:
            raise RuntimeError("rocprofv3 pass exited with %d: %s" % (proc.returncode, (err or "")[-200:]))
        files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise RuntimeError("rocprofv3 pass wrote no counter_collection.csv")
        return list(csv.DictReader(open(files[0]))), time.perf_counter() - t0
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


PROFILED_STEPS = 4          # a counter pass runs 2 warm-up + 2 timed steps; every dispatch of it is counted


def mfma_busy_live(args, timeout_s=100):
    """MFMA-busy % and HBM traffic of THIS command measured on THIS box (the replayed figures come from a committed
    profile): three child counter passes of `python3 bench.py --steps 2 --warmup 2 --legs none` under rocprofv3 -- SQ_VALU_MFMA_BUSY_CYCLES +
    GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) -- BEFORE this process touches the GPU
    (sequential: nothing shares the card with the timed region), condensed exactly as tools/prof_summary.py does: busy = sum of busy cycles /
    (1024 SIMDs x sum of GRBM_GUI_ACTIVE / 8 XCDs) over every dispatch; HBM bytes per launch = 2 x FETCH_SIZE KiB (the gfx950 half-count
    rule) + WRITE_SIZE KiB, averaged over a kernel's launches.  Returns None when rocprofv3 is not on this machine; an {"error": ...} object --
    never an exception -- when the first pass fails or times out (a failed traffic pass only leaves `traffic` to the replayed figures)."""
    import collections
    import shutil
    if not (shutil.which("rocprofv3") or os.path.isfile("/opt/rocm/bin/rocprofv3")):
        return None
    try:
        rows, secs = _counter_pass(args, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], timeout_s)
    except Exception as e:                                  # never lose the headline line over this leg
        return {"error": ("%s: %s" % (type(e).__name__, e))[:300]}
    tot, gemm = collections.defaultdict(float), collections.defaultdict(float)
    seen, ns = set(), 0
    for x in rows:
        tot[x["Counter_Name"]] += float(x["Counter_Value"])
        if "gemm8p_kernel<0>" in x["Kernel_Name"]:
            gemm[x["Counter_Name"]] += float(x["Counter_Value"])
        if x["Dispatch_Id"] not in seen:
            seen.add(x["Dispatch_Id"])
            ns += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    simds = 1024.0 / 8.0
    pct = lambda c: round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * c["GRBM_GUI_ACTIVE"]), 2) if c.get("GRBM_GUI_ACTIVE") else None
    out = {"pct": pct(tot), "gemm8p_kernel<0>_pct": pct(gemm), "kernel_ms_per_profiled_step": round(ns / 1e6 / PROFILED_STEPS, 2), "pass_s": round(secs, 1),
           "source": "live: child `rocprofv3 --pmc ... -- python3 bench.py --steps 2 --warmup 2 --legs none` passes on this box, before the timed "
                     "region (all 4 steps of a pass counted)"}
    try:                                                    # HBM traffic per launch of the three roofline kernels
        if secs > 40:                                       # (a pass takes ~7 s: a slow first pass says the profiler is struggling here -- bound the leg at ~160 s)
            raise RuntimeError("first counter pass took %.0f s: traffic passes skipped" % secs)
        per = {}
        for counter, scale in (("FETCH_SIZE", 2 * 1024.0), ("WRITE_SIZE", 1024.0)):
            rws, s2 = _counter_pass(args, [counter], 60)
            out["pass_s"] = round(out["pass_s"] + s2, 1)
            acc = collections.defaultdict(list)
            for x in rws:
                if x["Counter_Name"] == counter:
                    acc[x["Kernel_Name"]].append(float(x["Counter_Value"]))
            for k, v in acc.items():
                e = per.setdefault(k, {"bytes": 0.0, "launches": 0})
                e["bytes"] += scale * sum(v) / len(v)
                e["launches"] = max(e["launches"], len(v))
        traffic = {}
        for key in ("gemm8p_kernel<0>", "add_ln_bwd_fast_kernel", "mha_sh_fwd"):
            cand = [v for name, v in per.items() if key in name]
            if cand:                                        # several template instances: the one that moves the most bytes in the step
                traffic[key] = round(max(cand, key=lambda c: c["launches"] * c["bytes"])["bytes"])
        out["traffic_bytes_per_launch"] = traffic
    except Exception as e:
        out["traffic_error"] = ("%s: %s" % (type(e).__name__, e))[:200]
    return out


def event_bracket_overhead_ms(device, n=96):
    """What a HIP-event bracket measures around NOTHING on a busy stream: the two event packets are each processed after
    the preceding work drains, so every bracket of ops.KernelTimer carries this constant on top of the kernel's own
    duration (rocprofv3's kernel-trace duration has no such term).  Calibrated in place -- a ~0.1 ms streaming copy of the
    library (mmae_debug_stream_copy, 256 MiB; not a GEMM: TunableOp would tune the new shape) keeps the queue busy, then an
    empty bracket -- and subtracted from the roofline legs' averages; both figures are printed.  (In a rocprofv3 summary of
    bench.py these n launches are the `calib_copy_kernel` row, outside every timed step.)"""
    from incomplete_multimodal_fusion_amd._lib import call, ptr, stream
    a = torch.zeros(256 << 20, device=device, dtype=torch.uint8)
    b = torch.empty_like(a)
    pairs = []
    for _ in range(n):
        call("mmae_debug_stream_copy", a.numel(), ptr(a), ptr(b), stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize(device)
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in pairs[n // 4:])
    return ms[len(ms) // 2]


# ---------------------------------------------------------------------------------------------------------- dry run
def dry_main(args):
    """Launcher rehearsal without a GPU (tests/test_bench_launch.py): same rendezvous, barrier / max-over-ranks timing and
    JSON assembly as the real run, gloo backend, but the 'step' is a small torch CPU module through dp.GradAllReducer --
    NOT the product path (which has no CPU form) and NOT a measurement: the line is labelled dry_run and carries no
    roofline."""
    import torch.distributed as dist
    from incomplete_multimodal_fusion_amd import dp
    os.environ.setdefault("MMAE_DIST_BACKEND", "gloo")
    distributed = dp.init_distributed()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    assert world == args.gpus, "launched %d ranks for --gpus %d" % (world, args.gpus)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.GELU(), torch.nn.Linear(256, 64))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    red = dp.GradAllReducer(net.parameters(), bucket_bytes=32 << 10,
                            grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32) if distributed else None
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 64, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        if red is not None:
            red.prepare()
        loss = (net(x) - x).pow(2).mean()
        loss.backward()
        if red is not None:
            red.finish()
        opt.step()
        return loss.detach()
    for _ in range(args.warmup):
        step()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if distributed:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    tmax = torch.tensor([dt_local], dtype=torch.float64)
    wsum = torch.cat([p.detach().flatten() for p in net.parameters()]).double().sum().reshape(1)
    wall = [torch.zeros_like(wsum) for _ in range(world)]
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_gather(wall, wsum)
    else:
        wall = [wsum]
    diag = {}
    if red is not None:
        ex = torch.tensor([red.exposed_ms()], dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        mine = torch.tensor([dt_local], dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [1e3 * float(v) / args.steps for v in allv]
        diag = {"dp": {"ranks_reported_by_backend": dist.get_world_size(), "allreduce_bytes_per_step": red.stats["allreduce_bytes"],
                       "buckets_per_step": red.stats["buckets"], "grad_wire_dtype": args.grad_dtype, "captured": False,
                       "comm_exposed_ms_last_step_max_over_ranks": round(float(ex.item()), 3),
                       "rank_ms_per_step_min": round(min(per_rank), 3), "rank_ms_per_step_max": round(max(per_rank), 3)}}
    if rank == 0:
        dt = float(tmax.item())
        print(json.dumps({**diag, "metric": "launcher_dry_run", "dry_run": True, "value": round(args.batch * world * args.steps / dt, 2),
                          "unit": "rows/s (stand-in CPU module, not the product path)", "n_gpus": world, "ranks": world,
                          "backend": dist.get_backend() if distributed else "none", "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "replicas_in_sync": bool(all(float(w) == float(wall[0]) for w in wall)),
                          "loss": round(float(loss), 6)}), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------- the run
def timed_region(step, x, steps, distributed, device):
    """barrier + synchronize | K steps | synchronize + barrier; MAX over ranks (contract of the driver).
    Also returns this rank's per-step durations (HIP events recorded behind every step, read after the region: no extra
    synchronisation inside it), the ranks' own wall times (what a bad scaling curve is made of) and the host's enqueue time per step."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        losses = step(x)
        marks[i + 1].record()
    enqueue_ms = (time.perf_counter() - t0) * 1e3 / steps     # the host's share: how far it runs ahead of the GPU
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    ranks_dt = [dt_local]
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mine = torch.tensor([dt_local], dtype=torch.float64, device=device)
        allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, mine)
        ranks_dt = [float(v.item()) for v in allv]
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return float(tmax.item()), losses, per_step, ranks_dt, enqueue_ms


def dp_diagnostics(reducer, world, steps, ranks_dt, device, distributed):
    """N > 1: what explains a scaling curve -- ranks the backend reports, bytes and buckets of one step's gradient exchange,
    the time the compute stream WAITED for the collectives in the last step (communication not hidden under backward; MAX
    over ranks), and the spread of the ranks' own step times (step = slowest rank)."""
    import torch.distributed as dist
    if reducer is None:
        return {}
    exposed = torch.tensor([reducer.exposed_ms()], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
    per_rank = [1e3 * t / steps for t in ranks_dt]
    return {"dp": {"ranks_reported_by_backend": dist.get_world_size() if distributed else 1,
                   "allreduce_bytes_per_step": reducer.stats["allreduce_bytes"], "buckets_per_step": reducer.stats["buckets"],
                   "grad_wire_dtype": "bf16" if reducer.grad_dtype == torch.bfloat16 else "fp32",
                   "captured": False,        # the N > 1 region is eager by design: the captured-DP step is an opt-in, experimental leg
                   "comm_exposed_ms_last_step_max_over_ranks": round(float(exposed.item()), 3),
                   "rank_ms_per_step_min": round(min(per_rank), 3), "rank_ms_per_step_max": round(max(per_rank), 3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="base", choices=["tiny", "small", "base", "large"])
    ap.add_argument("--domains", default="s1,s2,dem", help="input / output modalities: s1,s2,dem (3-modality table) or "
                    "s1,s2,dem,dnw (4-modality table: s1 2 ch, s2 4 ch, dem, dnw class map)")
    ap.add_argument("--contra", default="dino", choices=["dino", "hardneg", "none"], help="contrastive head of the step")
    ap.add_argument("--fusion-blocks", dest="fusion_blocks", type=int, default=1, help="0: the reference's multimae_quadruplet "
                    "model (Zorro-masked blocks only, task losses: use --contra none)")
    ap.add_argument("--grad-dtype", dest="grad_dtype", default="fp32", choices=["fp32", "bf16"], help="N > 1: wire dtype of the "
                    "gradient buckets")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5's per-GPU shape: --model large --domains "
                    "s1,s2,dem,dnw --num-encoded-tokens 512 --contra hardneg --per-sample-masks --dropout --batch 64 --clip-grad 1")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--input-size", dest="input_size", type=int, default=256)
    ap.add_argument("--num-encoded-tokens", dest="num_encoded_tokens", type=int, default=384)
    ap.add_argument("--fp32", action="store_true", help="fp32 compute instead of bf16 autocast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", dest="cpu_batch", type=int, default=8)
    ap.add_argument("--cpu-steps", dest="cpu_steps", type=int, default=5)
    ap.add_argument("--bucket-mb", type=int, default=128)
    ap.add_argument("--engine", type=int, default=1, help="1: flat-buffer fused AdamW + bf16 shadow weights (csrc/optim.hip); "
                    "0: torch.optim.AdamW(fused=True) on the fp32 parameters")
    ap.add_argument("--side-wgrad", type=int, default=0, help="1: weight-gradient GEMMs on a side stream (measured: no gain)")
    ap.add_argument("--tunable", type=int, default=1, help="1: torch TunableOp picks the hipBLASLt/rocBLAS solution per GEMM "
                    "shape (pre-tuned table in incomplete_multimodal_fusion_amd/tuned/, unseen shapes are tuned during warm-up)")
    ap.add_argument("--staging", type=int, default=0, help="1: the MAIN timed region runs in PCIe-inclusive mode (every step "
                    "takes a fresh RAW host batch through staging.TileStager).  Default 0: inputs resident in HBM (the "
                    "contract's `value`); the PCIe-inclusive rate is then timed as a secondary leg")
    ap.add_argument("--per-sample-masks", dest="per_sample", action="store_true",
                    help="main region: every sample draws its own mask row (packed variable-length segments)")
    ap.add_argument("--dropout", action="store_true", help="main region: random modality dropout (sample_tasks_uniformly: a "
                    "uniformly drawn non-empty modality subset per mask row, the others get no tokens)")
    ap.add_argument("--legs", default="auto", help="secondary legs timed after the main region: comma list of pcie,c3,graph (+ mfma: the live "
                    "rocprofv3 MFMA-busy pass in a child process BEFORE the timed region); 'auto' = all at N = 1 with default main settings, none "
                    "otherwise; 'none'")
    ap.add_argument("--clip-grad", dest="clip_grad", type=float, default=0.0, help="> 0: device-side global-norm clipping")
    ap.add_argument("--block-timer", dest="block_timer", type=int, default=1, help="1: HIP-event brackets around encoder layer 6 "
                    "(four events per step) for roofline_block")
    ap.add_argument("--tuning-env", dest="tuning_env", action="store_true", help="A/B runs (tools/ab_bench.sh): map MMAE_* tuning "
                    "variables onto the implementation switches (tools/tuning_env.py); the default run reads none")
    ap.add_argument("--dry-run", dest="dry_run", action="store_true", help="launcher rehearsal on CPU/gloo (see dry_main)")
    ap.add_argument("--tune-out", default="", help="write the TunableOp table here on exit (to refresh the committed table)")
    args = ap.parse_args()
    if args.config5:
        args.model, args.domains, args.num_encoded_tokens, args.contra = "large", "s1,s2,dem,dnw", 512, "hardneg"
        args.per_sample, args.dropout, args.batch, args.clip_grad = True, True, 64, 1.0
    if not args.fusion_blocks:
        args.contra = "none"                                  # the reference's quadruplet driver trains on the task losses only

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))           # nothing here has touched the GPU yet
    if args.dry_run:
        return dry_main(args)

    import torch.distributed as dist
    from incomplete_multimodal_fusion_amd import dp, ops
    from incomplete_multimodal_fusion_amd.pretrain import PretrainStep
    # the live MFMA-busy pass: only for the default invocation at N = 1 (`--legs auto`), and before anything here touches the GPU
    headline_cfg = args.domains == "s1,s2,dem" and args.fusion_blocks and args.contra == "dino" and args.model == "base" and \
        not (args.staging or args.per_sample or args.dropout or args.fp32) and args.batch == 256
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)   # (never nest profilers)
    live_busy = mfma_busy_live(args) if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not profiled and
                                         (args.legs == "auto" and headline_cfg or "mfma" in args.legs.split(","))) else None
    if args.tuning_env:
        from tools import tuning_env
        tuning_env.apply()
    distributed = dp.init_distributed()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    assert world == args.gpus, "launched %d ranks for --gpus %d" % (world, args.gpus)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)   # (% only matters for the
    torch.cuda.set_device(local)                                                          #  single-GPU gloo rehearsal)
    device = torch.device("cuda", local)

    if args.tunable:
        import shutil
        import tempfile
        import torch.cuda.tunable as tun
        table = os.path.join(ROOT, "incomplete_multimodal_fusion_amd", "tuned", "tunableop_gfx950.csv")
        work = args.tune_out or os.path.join(tempfile.gettempdir(), "mmae_tunableop_rank%d.csv" % rank)
        if os.path.isfile(table) and os.path.abspath(table) != os.path.abspath(work):
            shutil.copyfile(table, work)                       # never write into the tracked table implicitly
        tun.enable(True)
        tun.tuning_enable(world == 1)                          # N > 1 runs the same per-GPU shapes: table only
        tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
        tun.set_filename(work)
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(bool(args.tune_out))
    model = build(args, device)
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    lr = 1e-4 * args.batch * world / 256                                  # pretrain_mmae.py:334-335
    if args.engine:
        from incomplete_multimodal_fusion_amd.engine import FlatAdamW
        opt = FlatAdamW(model.parameters(), lr=lr, betas=(0.9, 0.95), weight_decay=0.05,
                        exclude=model.never_used_parameters())
        gdt = torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32
        reducer = dp.GradAllReducer(None, bucket_bytes=args.bucket_mb << 20, engine=opt, grad_dtype=gdt) if distributed else None
    else:
        gdt = torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32
        opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.95), weight_decay=0.05, fused=True)
        reducer = dp.GradAllReducer(model.parameters(), bucket_bytes=args.bucket_mb << 20, grad_dtype=gdt) if distributed else None
    model.per_sample_masks = bool(args.per_sample)
    resident_step = PretrainStep(model, opt, args.num_encoded_tokens, autocast=not args.fp32, grad_reducer=reducer,
                                 side_stream_wgrad=bool(args.side_wgrad), sample_tasks_uniformly=bool(args.dropout),
                                 clip_grad=args.clip_grad if args.clip_grad > 0 else None, contra=args.contra)
    x = synthetic_tiles(args, args.batch, args.input_size, device, 1234 + rank)
    torch.manual_seed(4321 + rank)

    def make_staged_step():
        import numpy as np
        from incomplete_multimodal_fusion_amd import staging
        g = np.random.default_rng(1234 + rank)
        n = args.input_size
        raw = {'s1': g.gamma(2.0, 0.1, size=(args.batch, 1, n, n)).astype(np.float32),
               's2': g.integers(0, 256, size=(args.batch, 3, n, n), dtype=np.uint8),
               'dem': g.normal(5.0, 7.0, size=(args.batch, 1, n, n)).astype(np.float32)}
        stager = staging.TileStager(device, image_size=n, slots=2)
        stager.submit(raw)

        def staged(_x):
            xb = stager.get()
            stager.submit(raw)                   # the following batch: host copy + H2D + staging under this step
            return resident_step(xb)
        return staged

    step = make_staged_step() if args.staging else resident_step

    prof = ops.KernelTimer("mmae_mha_fwd")
    prof_ln = ops.KernelTimer("mmae_add_ln_bwd")
    prof_ln.every = 3                         # 59 launches per step of the timed instance: every 3rd (rows alternate, 59 is odd: all shapes)
    prof_gemm = ops.KernelTimer("mmae_gemm_nt")
    prof_gemm.every = 11                      # 182 = 2 x 7 x 13 launches per step: a stride CO-PRIME to it visits every launch position of the
                                              # step once per 11 steps (round 4's every-7th was phase-locked: the same 26 launches each step)
    for _ in range(args.warmup):
        losses = step(x)
    calls0 = dict(ops.CALLS)
    ops.set_kernel_timer([prof, prof_ln, prof_gemm])
    # roofline_block: HIP events around ONE encoder layer (Block_Fusion + Block) of the middle of the stack, forward and backward
    block_layer = min(6, model.depth - 1)
    model.layer_timer = ops.LayerTimer(block_layer) if (model.depth > 1 and args.block_timer) else None
    dt, losses, per_step_ms, ranks_dt, host_enqueue_ms = timed_region(step, x, args.steps, distributed, device)
    ops.set_kernel_timer(None)
    calls_per_step = {k: round((ops.CALLS[k] - calls0[k]) / max(args.steps, 1), 1) for k in calls0}
    layer_timer, model.layer_timer = model.layer_timer, None
    loss_val = float(losses["loss"])
    assert loss_val == loss_val, "non-finite loss"

    # ---- secondary legs (same model / optimizer state, continuing the run; never `value`) -----------------------------
    default_doms = args.domains == "s1,s2,dem" and args.fusion_blocks and args.contra == "dino"
    default_main = not (args.staging or args.per_sample or args.dropout or args.fp32) and default_doms
    dpdiag = dp_diagnostics(reducer, world, args.steps, ranks_dt, device, distributed)
    legs = ("pcie,c3,graph" if (world == 1 and default_main) else "") if args.legs == "auto" else \
        ("" if args.legs == "none" else args.legs)
    legs = [l for l in legs.split(",") if l]
    leg_out = {}
    if "pcie" in legs and not args.staging:
        st = make_staged_step()
        for _ in range(2):
            st(x)
        d2, l2, _, _, _ = timed_region(st, x, args.steps, distributed, device)
        leg_out["pcie_inclusive"] = {
            "value": round(args.batch * world * args.steps / d2, 2), "unit": "samples/s", "ms_per_step": round(1e3 * d2 / args.steps, 3),
            "inputs": "fresh RAW host batch per step (fp32 SAR, uint8 RGB, fp32 DSM; 0.72 MB/sample): pageable->pinned ring, "
                      "async H2D + normalisation kernels (csrc/staging.hip) overlapped with the previous step",
            "loss": round(float(l2["loss"]), 4)}
    if "c3" in legs:
        model.per_sample_masks = True
        resident_step.uniform = True
        for _ in range(2):
            resident_step(x)
        d3, l3, _, _, _ = timed_region(resident_step, x, args.steps, distributed, device)
        model.per_sample_masks = bool(args.per_sample)
        resident_step.uniform = bool(args.dropout)
        leg_out["c3_per_sample_dropout"] = {
            "value": round(args.batch * world * args.steps / d3, 2), "unit": "samples/s", "ms_per_step": round(1e3 * d3 / args.steps, 3),
            "masks": "one Dirichlet draw PER SAMPLE over a uniformly drawn non-empty modality subset (sample_tasks_uniformly: "
                     "each modality dropped w.p. 1/2, never all), N=%d kept tokens per sample in variable-length packed "
                     "segments" % args.num_encoded_tokens,
            "loss": round(float(l3["loss"]), 4)}

    if "graph" in legs and args.engine and not distributed:
        # the same step as ONE hipGraph (PretrainStep.capture): one graph launch instead of ~2500 launches from Python.  Last leg:
        # after capture the step must only be replayed.  The roofline timers cannot be captured (HIP-event brackets), so the
        # headline region above stays eager; this leg shows what the host's enqueue time costs at this configuration.
        try:
            ops.set_kernel_timer(None)
            model.layer_timer = None
            t_cap = time.perf_counter()
            resident_step.capture(x, None)
            for _ in range(2):
                resident_step.replay()
            torch.cuda.synchronize()
            t_cap = time.perf_counter() - t_cap
            d4, l4, _, _, enq4 = timed_region(resident_step.replay, x, args.steps, distributed, device)
            leg_out["graph_replay"] = {
                "value": round(args.batch * world * args.steps / d4, 2), "unit": "samples/s", "ms_per_step": round(1e3 * d4 / args.steps, 3),
                "host_enqueue_ms_per_step": round(enq4, 3), "capture_s": round(t_cap, 2),
                "what": "the headline step captured into one hipGraph (forward, losses, backward, AdamW; mask shares drawn on the host "
                        "before each replay), bitwise the eager step", "loss": round(float(l4["loss"]), 4)}
        except Exception as e:                                   # never lose the headline line over the secondary leg
            leg_out["graph_replay"] = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = args.batch * world * args.steps / dt
        ev_ms = event_bracket_overhead_ms(device)
        raw_ms, n_launch, flops = prof.summary()
        avg_ms = max(raw_ms - ev_ms, 1e-6)
        # roofline of the fused attention kernel: mask-aware algorithmic FLOPs per launch
        # 4 * dh * h * sum_b (sum_m N_m^2 + P * S)   (SURVEY.md 8d), accumulated on the device per launch
        ach = flops / n_launch / (avg_ms * 1e-3) / 1e12 if n_launch else 0.0
        ln_raw_ms, ln_n, ln_bytes = prof_ln.summary()
        ln_ms = max(ln_raw_ms - ev_ms, 1e-6)
        ln_gbs = ln_bytes / ln_n / (ln_ms * 1e-3) / 1e9 if ln_n else 0.0
        mask_desc = ("per-sample" if args.per_sample else "batch-shared") + (" + modality dropout" if args.dropout else "")
        # counter figures are REPLAYED from the committed rocprofv3 summaries of this very command (they cannot be collected
        # in-process): only for the profiled configuration, and only while the profiled step time is within 5 % of this run's
        replay_ok = not args.fp32 and args.batch == 256 and default_doms and not (args.per_sample or args.dropout or args.staging)
        rep_sq, sq_ok = replayed_from("sq_step", ms) if replay_ok else (None, False)
        rep_hbm, hbm_ok = replayed_from("pmc_hbm", ms) if replay_ok else (None, False)
        live_traffic = (live_busy or {}).get("traffic_bytes_per_launch") or {}        # measured on this box by mfma_busy_live's FETCH / WRITE passes
        out = {
            "metric": "pretrain_samples_per_sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "median_ms_per_step": round(sorted(per_step_ms)[len(per_step_ms) // 2], 3),
            "host_enqueue_ms_per_step": round(host_enqueue_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": workload_string(args, mask_desc),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "inputs": "host, PCIe inclusive" if args.staging else "resident in HBM",
                       "trainable_params": n_params, "loss": round(loss_val, 4)},
            "ranks": world, "backend": (dist.get_backend() if distributed else "none"),
            # the dominant kernel of the step by total time (profiles/rNN_kernel_stats.md): the own persistent GEMM gemm8p_kernel<0> (every
            # forward / input-gradient projection of the encoder), MFMA-bound.  Filled in below from the live HIP-event brackets.
            "roofline": None,
            # dominant HBM-bound hand-written kernel: the fused residual-add + double-LayerNorm backward.
            # achieved = algorithmic bytes of the launch / its HIP-event time.
            "roofline_hbm": {"kernel": "add_ln_bwd_fast_kernel<bf16,bf16,3,double,up,gx,gdelta>" if not args.fp32 else "add_ln_bwd_fast_kernel<f32,...>",
                         "bound": "hbm", "achieved": round(ln_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ln_gbs / HBM_PEAK_GBS, 4),
                         "traffic": live_traffic.get("add_ln_bwd_fast_kernel") or (pmc_traffic("add_ln_bwd_fast_kernel") if (replay_ok and hbm_ok) else None),
                         "algorithmic_bytes_per_launch": round(ln_bytes / ln_n) if ln_n else 0,
                         "avg_launch_ms": round(ln_ms, 4), "avg_bracket_ms": round(ln_raw_ms, 4),
                         "event_bracket_overhead_ms": round(ev_ms, 4), "launches": ln_n,
                         "sampling": "every 3rd launch of this kernel instance"},
            # the flagship MFMA kernel of the path (north_star: fusion-attention block), mask-aware algorithmic FLOPs
            "roofline_attention": {"kernel": "+".join(sorted(prof.kernels)) or None,   # as routed by the library (mmae_mha_fwd_route)
                         "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_BF16_PEAK_TF, 4),
                         "traffic": live_traffic.get("mha_sh_fwd") or (pmc_traffic("mha_sh_fwd") if (replay_ok and hbm_ok) else None),
                         "algorithmic_flops_per_launch": round(flops / n_launch) if n_launch else 0,
                         "algorithmic_bytes_per_launch": args.batch * (args.num_encoded_tokens + (args.input_size // 16) ** 2) * 4 * 64 * {"tiny": 3}.get(args.model, 8) * 2,
                         "avg_launch_ms": round(avg_ms, 4), "avg_bracket_ms": round(raw_ms, 4),
                         "event_bracket_overhead_ms": round(ev_ms, 4), "launches": n_launch},
        }
        dense, executed = step_flops(args)
        # whole-step view: FLOPs per sample x samples/s against the dense bf16 MFMA peak of the job.  `achieved`
        # / `frac` count the FLOPs this implementation EXECUTES (fusion-slot shortcut of Block_Fusion); the reference-dense
        # formulation SURVEY 8d defines is kept beside it as a labelled secondary figure.
        peak = MFMA_BF16_PEAK_TF * world
        out["roofline_step"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak,
                                "achieved": round(value * executed / 1e12, 1), "frac": round(value * executed / 1e12 / peak, 4),
                                "flops_per_sample_executed": round(executed),
                                "flops_per_sample_reference_dense": round(dense),
                                "achieved_reference_dense": round(value * dense / 1e12, 1),
                                "frac_reference_dense": round(value * dense / 1e12 / peak, 4)}
        if layer_timer is not None:
            fw_ms, bw_ms, nst = layer_timer.summary()
            ldense, lexec = step_flops(args, per_layer=True)
            tot_ms = fw_ms + bw_ms
            ach_b = args.batch * lexec / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            # the quantity north_star sets its >= 50 % MFMA target on: one fusion-attention block (Block_Fusion + Zorro-masked
            # Block of one encoder layer), forward + backward incl. its weight gradients, measured live by HIP events on the
            # stream it runs on; FLOPs = SURVEY 8(d)'s per-layer terms x 6 (executed: fusion-slot shortcut; reference-dense beside it)
            out["roofline_block"] = {"what": "encoder layer %d: Block_Fusion + Block, forward + backward (HIP events, %d steps)" % (block_layer, nst),
                                     "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TF,
                                     "forward_ms": round(fw_ms, 3), "backward_ms": round(bw_ms, 3),
                                     "flops_per_sample_executed": round(lexec), "flops_per_sample_reference_dense": round(ldense),
                                     "achieved": round(ach_b, 1), "frac": round(ach_b / MFMA_BF16_PEAK_TF, 4),
                                     "achieved_reference_dense": round(args.batch * ldense / (tot_ms * 1e-3) / 1e12, 1) if tot_ms > 0 else 0.0,
                                     "frac_reference_dense": round(args.batch * ldense / (tot_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4) if tot_ms > 0 else 0.0}
        # `roofline`: the dominant kernel of the step BY TIME -- the own persistent GEMM (csrc/gemm.hip, gemm8p_kernel<0>: every forward /
        # input-gradient projection of the encoder) -- measured live: HIP events on the stream it runs on around every 11th launch of
        # mmae_gemm_nt (11 is co-prime to the 182 launches of a step, so over the timed region every launch position -- every shape --
        # is sampled equally often), algorithmic FLOPs 2 M N K of exactly the bracketed launches.  The committed rocprofv3 summary of
        # this command (profiles/rNN_kernel_stats.md) must agree: FLOPs per step / its ms per step for this kernel.
        g_raw, g_n, g_fl = prof_gemm.summary()
        if g_n:
            g_ms = max(g_raw - ev_ms, 1e-6)
            g_tf = g_fl / g_n / (g_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "gemm8p_kernel<0> (own persistent 256x256x64 8-phase bf16 GEMM: forward + input-gradient projections)",
                               "bound": "mfma", "achieved": round(g_tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                               "frac": round(g_tf / MFMA_BF16_PEAK_TF, 4),
                               "traffic": live_traffic.get("gemm8p_kernel<0>") or (pmc_traffic("gemm8p_kernel<0>") if (replay_ok and hbm_ok) else None),
                               "algorithmic_flops_per_launch": round(g_fl / g_n), "avg_launch_ms": round(g_ms, 4),
                               "avg_bracket_ms": round(g_raw, 4), "event_bracket_overhead_ms": round(ev_ms, 4),
                               "launches": g_n, "launches_per_step": round(prof_gemm.seen / max(args.steps, 1), 1),
                               "sampling": "every 11th launch of mmae_gemm_nt (stride co-prime to the launches per step)"}
        else:                                      # configurations the own GEMM does not serve (fp32, small batches): the HBM leg is the line's roofline
            out["roofline"] = out["roofline_hbm"]
        if replay_ok:
            # replayed (not measured in this process): null when the profiled step no longer matches this run
            rg = gemm_roofline()
            out["roofline_gemm"] = rg if (rg is not None and sq_ok) else None
            mb = mfma_busy()
            out["mfma_busy_pct"] = mb["pct"] if (mb is not None and sq_ok) else None
            out["mfma_busy_source"] = mb["source"] if mb is not None else None
            out["replayed_from"] = {"sq_step": rep_sq, "pmc_hbm": rep_hbm,
                                    "fields": "mfma_busy_pct (unless measured live), roofline_gemm <- sq_step; roofline.traffic, roofline_hbm.traffic, roofline_attention.traffic <- pmc_hbm"}
        if live_busy is not None:
            # measured on THIS box by a counter pass of this command (mfma_busy_live); when it succeeded it IS the line's MFMA-busy figure,
            # the replayed one is kept beside it
            out["mfma_busy_live"] = live_busy
            if live_busy.get("pct") is not None:
                out["mfma_busy_pct_replayed"] = out.get("mfma_busy_pct")
                out["mfma_busy_pct"] = live_busy["pct"]
                out["mfma_busy_source"] = "live (mfma_busy_live)"
            if live_traffic:
                out["traffic_source"] = "live (mfma_busy_live: FETCH_SIZE / WRITE_SIZE passes of this command on this box)"
        # which projections ran where (the own GEMM takes a projection from _OWN_GEMM_MIN_TILES output tiles on), and how the sample-head
        # attention kernels split a sample's heads over workgroups (csrc/mha_sh.hip sh_heads_per_block: >= 256 workgroups when B allows)
        heads = {"tiny": 3}.get(args.model, 8)
        hpb = heads
        while hpb > 1 and args.batch * (heads // hpb) < 256:
            hpb = max(d for d in range(1, hpb) if heads % d == 0)
        out["dispatch"] = {"launches_per_step": calls_per_step,
                           "own_gemm_min_tiles": "%d (%d where N >= 512)" % (ops._OWN_GEMM_MIN_TILES, ops._OWN_GEMM_MIN_TILES // 2),
                           "attention_heads_per_workgroup": hpb, "attention_workgroups": args.batch * (heads // hpb)}
        out.update(dpdiag)
        out.update(leg_out)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args) if default_doms else None      # the CPU leg is the headline configuration's
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
