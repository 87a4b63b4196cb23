#!/usr/bin/env python
"""gpurun_out/parity_log.jsonl (one JSON line per tests/parity.compare() call of a `pytest -m gpu` run) -> the tracked table
profiles/rNN_parity.md: per end-to-end parity test, how many outputs / gradient tensors sit above the plain tolerance, the worst
single-tensor HIP/anchor ratio and the aggregate gradient rel-L2 of the HIP run and of the anchor (the oracle under CPU bf16
autocast).  `strict` = the largest max-abs error among the loss scalars and the largest rel-L2 error among the prediction images:
both must meet the plain tolerance whatever the anchor says; the predictions' max-abs error is listed beside the anchor's."""
import json
import sys


def main(path):
    rows = [json.loads(l) for l in open(path) if l.strip()]
    print("# Parity verdicts of the GPU end-to-end tests (tests/parity.py)\n")
    print("One row per `compare()` call of the `pytest -m gpu` run of this round.  fp32 mode: every tensor within tol (gradients 2 x tol), "
          "max-abs relative.  bf16-anchored mode: outputs within max(1e-2, 1.5 x anchor), gradient tensors (rel L2) within max(1e-2, "
          "2.5 x anchor), all gradients together within max(1e-2, 1.5 x anchor); STRICT (no anchor): loss scalars within the plain 1e-2 "
          "(max-abs relative), prediction images within the plain 1e-2 in relative L2.\n")
    print("| test | mode | tol | outputs > tol | max output err | strict: loss max-abs | strict: pred rel L2 (anchor) | pred max-abs (anchor) | "
          "grad tensors > tol | max grad-tensor err | all-grads rel L2 HIP | anchor | worst ratio (tensor) | median ratio | failed |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    f = lambda v, fmt="%.2e": "-" if v is None else fmt % v
    for r in rows:
        pl2 = "-" if "pred_rel_l2_max" not in r else "%.2e (%.2e)" % (r["pred_rel_l2_max"], r["pred_rel_l2_max_anchor"])
        pma = "-" if "pred_maxabs_max" not in r else "%.2e (%.2e)" % (r["pred_maxabs_max"], r["pred_maxabs_max_anchor"])
        print("| `%s` | %s | %.0e | %d / %d | %s | %s | %s | %s | %d / %d | %s | %s | %s | %s | %s | %d |" % (
            r.get("test", "").replace("tests/", ""), r["mode"], r["tol"], r["outputs_above_tol"], r["outputs"], f(r["max_output_err"]),
            f(r["strict_max_err"]), pl2, pma, r["grad_tensors_above_tol"], r["grad_tensors"], f(r["max_grad_err"]),
            f(r.get("grad_rel_l2_hip")), f(r.get("grad_rel_l2_anchor")),
            ("%.2f (`%s`)" % (r["worst_ratio"], r.get("worst_tensor", ""))) if "worst_ratio" in r else "-",
            f(r.get("median_ratio"), "%.2f"), r["failed"]))
    exc = [r for r in rows if r.get("pred_l2_tol") is not None and r["pred_l2_tol"] > r["tol"] * (1 + 1e-9)]
    if exc:
        print("\nEXCEPTIONS to the strict prediction clause (a test that raises `pred_l2_tol` above the plain tolerance is listed here with "
              "the reference arithmetic's own figure, the anchor):\n")
        for r in exc:
            print("* `%s`: prediction images held to rel L2 %.0e instead of %.0e -- measured %s, the oracle's own bf16 run (anchor) %s: the "
                  "reference's arithmetic in bf16 does not meet the plain figure on this fixture either."
                  % (r.get("test", "").replace("tests/", ""), r["pred_l2_tol"], r["tol"], f(r.get("pred_rel_l2_max")), f(r.get("pred_rel_l2_max_anchor"))))
    bad = sum(1 for r in rows if r["failed"])
    bf = [r for r in rows if r["mode"] != "fp32"]
    print("\n%d compare() calls, %d with a failing tensor; strict clause worst cases over the bf16 runs: loss scalar %s, prediction rel L2 %s "
          "(prediction max-abs %s, anchor %s)" % (len(rows), bad, f(max((r["strict_max_err"] for r in bf), default=None)),
                                                   f(max((r.get("pred_rel_l2_max", 0.0) for r in bf), default=None)),
                                                   f(max((r.get("pred_maxabs_max", 0.0) for r in bf), default=None)),
                                                   f(max((r.get("pred_maxabs_max_anchor", 0.0) for r in bf), default=None))))


if __name__ == "__main__":
    main(sys.argv[1])
