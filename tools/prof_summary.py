#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

  python tools/prof_summary.py stats  <kernel_stats.csv> <steps> > profiles/rNN_kernel_stats.md
  python tools/prof_summary.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_hbm.md
  python tools/prof_summary.py json   <fetch counter_collection.csv> <write counter_collection.csv> [steps] > profiles/rNN_pmc_hbm.json
  python tools/prof_summary.py sq     <SQ counter_collection.csv> <steps> [json] > profiles/rNN_sq_step.md | .json

PMC correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read, so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void at::native::", "at::").replace("(anonymous namespace)::", "")
    return n[:86]


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in rows[:40]:
        print("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    print("\nall kernels: %.1f ms over the profiled run = %.2f ms per step (%d steps incl. warm-up)" % (tot / 1e6, tot / 1e6 / steps, steps))


def pmc(fetch, write):
    def agg(path, name):
        d = collections.defaultdict(list)
        for x in csv.DictReader(open(path)):
            if x["Counter_Name"] == name:
                d[x["Kernel_Name"]].append(float(x["Counter_Value"]))
        return d
    f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    print("| kernel | launches | HBM read MB/launch (2 x FETCH_SIZE KiB) | HBM write MB/launch | total MB/launch |\n|---|---|---|---|---|")
    keys = sorted(set(f) | set(w), key=lambda k: -(2 * sum(f.get(k, [0])) + sum(w.get(k, [0]))))
    for k in keys[:30]:
        n = max(len(f.get(k, [])), len(w.get(k, [])))
        rd = 2 * sum(f.get(k, [0])) / max(len(f.get(k, [1])), 1) * 1024 / 1e6
        wr = sum(w.get(k, [0])) / max(len(w.get(k, [1])), 1) * 1024 / 1e6
        print("| `%s` | %d | %.1f | %.1f | %.1f |" % (short(k), n, rd, wr, rd + wr))


def pmc_json(fetch, write, steps=0):
    """Machine-readable form of `pmc` (bench.py reads it for roofline.traffic): corrected HBM bytes per launch per kernel, and the
    kernel time per step of the FETCH pass (bench.py prints the traffic only while that is within 5 % of its own step time)."""
    import json
    d = {}
    seen, ns = set(), 0
    for x in csv.DictReader(open(fetch)):
        if x["Dispatch_Id"] not in seen:
            seen.add(x["Dispatch_Id"])
            ns += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    for path, name, key, scale in ((fetch, "FETCH_SIZE", "hbm_read_bytes_per_launch", 2 * 1024.0),
                                   (write, "WRITE_SIZE", "hbm_write_bytes_per_launch", 1024.0)):
        acc = collections.defaultdict(list)
        for x in csv.DictReader(open(path)):
            if x["Counter_Name"] == name:
                acc[x["Kernel_Name"]].append(float(x["Counter_Value"]))
        for k, v in acc.items():
            e = d.setdefault(k, {"hbm_read_bytes_per_launch": 0.0, "hbm_write_bytes_per_launch": 0.0, "launches": 0})
            e[key] = scale * sum(v) / len(v)
            e["launches"] = max(e["launches"], len(v))
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 "
                                "--warmup 1 --no-cpu-baseline`, B=256; bytes = 2*FETCH_SIZE*1024 (gfx950 half-count rule) + "
                                "WRITE_SIZE*1024", "kernel_ms_per_step": round(ns / 1e6 / steps, 2) if steps else None, "steps": steps,
                      "kernels": d}, indent=1))


def sq(path, steps, as_json=False):
    """Per-kernel and whole-step SQ view of `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python3 bench.py ...`.
    MFMA-busy % of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): BUSY_CYCLES is summed over the
    chip's 256 x 4 SIMDs (it equals 16 x the 16x16x32 / 32 x the 32x32x16 MFMA count, MI355X_MICROARCH.md cycle constants --
    checked on this data: BUSY / SQ_INSTS_MFMA = 16.0 for the 16x16x32 kernels); rocprofv3 reports GRBM_GUI_ACTIVE summed over
    the 8 XCDs (GUI_ACTIVE / dispatch ns = 16-18.7, i.e. 8 x 2.0-2.33 GHz), so GUI/8 is the dispatch's length in shader clocks.
    Whole step: sum of busy cycles / (1024 x sum of GUI/8) over every dispatch of the profiled steps (idle gaps between kernels
    are not in the denominator).  The percentage is of ACTUAL shader cycles: TFLOP/s = busy % x 2.5 PF x (held clock / 2.4 GHz)."""
    import json
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for x in csv.DictReader(open(path)):
        k = x["Kernel_Name"]
        acc[k][x["Counter_Name"]] += float(x["Counter_Value"])
        if (x["Dispatch_Id"], k) not in seen:
            seen.add((x["Dispatch_Id"], k))
            calls[k] += 1
            acc[k]["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    SIMDS = 1024.0 / 8.0          # per unit of the 8-XCD-summed GRBM_GUI_ACTIVE
    tot = collections.defaultdict(float)
    rows = []
    for k, c in acc.items():
        for n, v in c.items():
            tot[n] += v
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        rows.append(dict(kernel=k, calls=calls[k], ms=c["ns"] / 1e6,
                         mfma_busy_pct=100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * gui) if gui else 0.0,
                         valu_per_mfma=c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else None,
                         # per 16 MFMA-busy cycles = per 16x16x32-equivalent (a 32x32x16 instruction is two of them)
                         valu_per_mfma16=16.0 * c.get("SQ_INSTS_VALU", 0.0) / c["SQ_VALU_MFMA_BUSY_CYCLES"] if c.get("SQ_VALU_MFMA_BUSY_CYCLES") else None,
                         lds_conflict_pct=100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
                         wait_pct=100.0 * c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                         clock_ghz=gui / 8.0 / c["ns"] if c["ns"] else None,
                         mfma_busy_cycles=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), gui_active=gui))
    rows.sort(key=lambda r: -r["mfma_busy_cycles"])
    step = dict(mfma_busy_pct=round(100.0 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * tot["GRBM_GUI_ACTIVE"]), 2) if tot["GRBM_GUI_ACTIVE"] else None,
                kernel_ms_per_step=round(tot["ns"] / 1e6 / steps, 2), steps=steps,
                insts_valu=tot["SQ_INSTS_VALU"], insts_mfma=tot["SQ_INSTS_MFMA"])
    if as_json:
        print(json.dumps({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT "
                                    "SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 2 "
                                    "--no-cpu-baseline --legs none (B=256); MFMA-busy = BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)",
                          "step": step, "kernels": {r["kernel"]: {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "kernel"}
                                                    for r in rows[:40]}}, indent=1))
        return
    print("whole step (all dispatches of %d profiled steps): MFMA-busy %.1f %% of the chip's SIMD cycles while a kernel runs; "
          "%.1f ms of kernels per step; %.3g VALU / %.3g MFMA instructions" % (steps, step["mfma_busy_pct"], step["kernel_ms_per_step"],
                                                                              tot["SQ_INSTS_VALU"], tot["SQ_INSTS_MFMA"]))
    print("\n| kernel | calls | total ms | MFMA busy % | VALU / MFMA instr | VALU / 16 MFMA-busy cycles (16x16x32-equivalents) | LDS cycles conflicted % | wave cycles parked (s_waitcnt / barrier) % | clock GHz |\n|---|---|---|---|---|---|---|---|---|")
    f = lambda v, fmt="%.1f": "-" if v is None else fmt % v
    for r in rows[:24]:
        print("| `%s` | %d | %.2f | %.1f | %s | %s | %s | %s | %s |" % (short(r["kernel"]), r["calls"], r["ms"], r["mfma_busy_pct"], f(r["valu_per_mfma"]), f(r["valu_per_mfma16"]),
                                                                    f(r["lds_conflict_pct"]), f(r["wait_pct"]), f(r["clock_ghz"], "%.2f")))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "json":
        pmc_json(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], int(sys.argv[3]), len(sys.argv) > 4 and sys.argv[4] == "json")
    else:
        pmc(sys.argv[2], sys.argv[3])
