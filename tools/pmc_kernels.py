#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv, plus the ratios used in DESIGN.md.
    python tools/pmc_kernels.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    d = len(n[k])
    print("%s  (%d dispatches)" % (k[:100], d))
    print("   " + "  ".join("%s %.3g" % (a, b / d) for a, b in sorted(c.items())))
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        out = ["%s %.1f %%" % (a, 100 * c[a] / wc) for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                           "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC") if a in c]
        print("   of wave cycles: " + ", ".join(out))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print("   MFMA busy %.1f %% of SIMD cycles; kernel length %.0f shader cycles" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), c["GRBM_GUI_ACTIVE"] / 8 / d))
