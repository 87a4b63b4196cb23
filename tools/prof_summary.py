#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

  python tools/prof_summary.py stats  <kernel_stats.csv> <steps> > profiles/rNN_kernel_stats.md
  python tools/prof_summary.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_hbm.md
  python tools/prof_summary.py json   <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_pmc_hbm.json

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


def pmc_json(fetch, write):
    """Machine-readable form of `pmc` (bench.py reads it for roofline.traffic): corrected HBM bytes per launch per kernel."""
    import json
    d = {}
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
                                "WRITE_SIZE*1024", "kernels": d}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "json":
        pmc_json(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3])
