#!/usr/bin/env python
"""GPU idle time inside the bench step, from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`: *_kernel_trace.csv).

    python tools/gap_summary.py <kernel_trace.csv> [first_fraction_to_skip=0.5]

Takes the second half of the run (steady-state steps), merges the kernels' [start, end) intervals over all queues and reports: span, busy
time (union), idle time, and the idle time grouped by the pair (kernel that ended before the gap -> kernel that started after it)."""
import collections
import csv
import sys


def short(n):
    return n.replace("void at::native::", "at::").replace("(anonymous namespace)::", "")[:70]


path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]
span = rows[-1][1] - rows[0][0]
busy, cur_end, last_name = 0, rows[0][0], None
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
for s, e, n in rows:
    if s > cur_end:
        g = s - cur_end
        k = (short(last_name or "-"), short(n))
        gaps[k][0] += g; gaps[k][1] += 1
        hist[min(int(g / 1000), 50)] += g
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
        last_name = n
idle = span - busy
print("kernels %d   span %.2f ms   busy %.2f ms   idle %.2f ms (%.2f %%)" % (len(rows), span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span))
print("idle time by gap length: " + ", ".join("%s us: %.2f ms" % (("%d-%d" % (k, k + 1)) if k < 50 else ">=50", v / 1e6) for k, v in sorted(hist.items())))
print("\n| ended -> started | gaps | total idle ms | mean us |\n|---|---|---|---|")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:30]:
    print("| `%s` -> `%s` | %d | %.3f | %.1f |" % (k[0], k[1], c, t / 1e6, t / c / 1e3))

# exclusive time: the part of a kernel's [start, end) during which NO other kernel runs -- what removing that launch could give back
ev = []
for idx, (s_, e_, n_) in enumerate(rows):
    ev.append((s_, 1, idx)); ev.append((e_, 0, idx))
ev.sort()
active, excl, last_t = set(), collections.Counter(), ev[0][0]
for t, kind, idx in ev:
    if len(active) == 1 and t > last_t:
        excl[next(iter(active))] += t - last_t
    last_t = t
    if kind:
        active.add(idx)
    else:
        active.discard(idx)
by = collections.defaultdict(lambda: [0, 0, 0])
for idx, (s_, e_, n_) in enumerate(rows):
    b = by[short(n_)]
    b[0] += 1; b[1] += e_ - s_; b[2] += excl[idx]
small = [(k, v) for k, v in by.items() if v[1] / v[0] < 15000]
print("\nkernels under 15 us on average: %d launches, %.2f ms of duration, %.2f ms of it exclusive (nothing else running)" %
      (sum(v[0] for _, v in small), sum(v[1] for _, v in small) / 1e6, sum(v[2] for _, v in small) / 1e6))
print("\n| kernel (< 15 us) | launches | duration ms | exclusive ms |\n|---|---|---|---|")
for k, v in sorted(small, key=lambda kv: -kv[1][2])[:16]:
    print("| `%s` | %d | %.3f | %.3f |" % (k, v[0], v[1] / 1e6, v[2] / 1e6))
