#!/usr/bin/env python
"""profiles/rNN_batch64.md from two bench lines: `python bench.py --batch 64 --legs graph --no-cpu-baseline` (the reference's per-GPU scale,
pretraining/pretrain_mmae.py:79) and the headline line at B = 256 of the same round.
    python tools/batch64_table.py gpurun_out/b64.json gpurun_out/round/bench.json > profiles/r06_batch64.md"""
import json
import sys


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def main():
    b64, b256 = last_json(sys.argv[1]), last_json(sys.argv[2])
    d64, d256 = b64.get("dispatch", {}), b256.get("dispatch", {})
    r = b64["value"] / b256["value"]
    print("# Round 06 - the reference's per-GPU batch (64) next to the headline batch (256)\n")
    print("Commands: `python bench.py --batch 64 --legs graph --no-cpu-baseline` and the default `python bench.py` (separate boxes of one round: +-2.5 % box spread).\n")
    print("| | B = 64 | B = 256 |\n|---|---|---|")
    print("| samples/s | %.1f | %.1f |" % (b64["value"], b256["value"]))
    print("| ms per step (mean / median) | %.2f / %.2f | %.2f / %.2f |" % (b64["ms_per_step"], b64["median_ms_per_step"], b256["ms_per_step"], b256["median_ms_per_step"]))
    print("| host enqueue ms per step (share of the step) | %.1f (%.0f %%) | %.1f (%.0f %%) |" % (
        b64["host_enqueue_ms_per_step"], 100 * b64["host_enqueue_ms_per_step"] / b64["ms_per_step"],
        b256["host_enqueue_ms_per_step"], 100 * b256["host_enqueue_ms_per_step"] / b256["ms_per_step"]))
    for k, label in (("mmae_gemm_nt", "projections on the own GEMM (`mmae_gemm_nt`) per step"), ("mmae_gemm_geglu", "FF1 + GEGLU on the own GEMM per step"),
                     ("library_matmul_nt", "projections handed to the library by `matmul_nt` per step"), ("mmae_gemm_tn", "own weight-gradient GEMM per step")):
        print("| %s | %s | %s |" % (label, d64.get("launches_per_step", {}).get(k, "-"), d256.get("launches_per_step", {}).get(k, "-")))
    print("| attention heads per workgroup (workgroups per launch) | %s (%s) | %s (%s) |" % (
        d64.get("attention_heads_per_workgroup", "-"), d64.get("attention_workgroups", "-"),
        d256.get("attention_heads_per_workgroup", "-"), d256.get("attention_workgroups", "-")))
    for name in ("roofline", "roofline_hbm", "roofline_attention", "roofline_block", "roofline_step"):
        a, b = b64.get(name) or {}, b256.get(name) or {}
        print("| `%s.frac` | %s | %s |" % (name, a.get("frac", "-"), b.get("frac", "-")))
    g = b64.get("graph_replay") or {}
    if "value" in g:
        print("| the step as one hipGraph (`graph_replay`): samples/s, ms per step | %.1f, %.2f | (see the headline line) |" % (g["value"], g["ms_per_step"]))
    print("\nPer-sample rate at B = 64: **%.0f %%** of the B = 256 rate (VERDICT r5 item 6: at or below 85 %% calls for the A/B of `_OWN_GEMM_MIN_TILES`; threshold of this line: %s output tiles)." % (
        100 * r, d64.get("own_gemm_min_tiles", "512")))
    if "value" in g:
        print("Captured into one hipGraph (no host enqueue time at all) the B = 64 step runs at %.0f %% of the B = 256 rate (eager: %.0f %%; the host needs %.1f ms "
              "to enqueue a step that takes %.1f ms -- on a box with a slower host the eager step is host-bound, and `PretrainStep.capture` is the remedy).  What "
              "separates the captured step from the B = 256 rate is the kernels at a quarter of the rows: fewer tiles per persistent workgroup in the own GEMM "
              "(`roofline.frac`), shorter streaming launches (`roofline_hbm.frac`)." % (100 * g["value"] / b256["value"], 100 * r, b64["host_enqueue_ms_per_step"], b64["ms_per_step"]))


if __name__ == "__main__":
    main()
