#!/usr/bin/env python
"""Stamp the JSON summaries copied into profiles/ with the commit whose tree the profiled run measured:

    python tools/stamp_profiles.py r04            (run in the build container right after copying gpurun_out/round/profiles/*,
                                                   with the profiled tree committed: HEAD is recorded)

bench.py replays `mfma_busy_pct`, `roofline_gemm` and the `traffic` fields from these files (the counters cannot be collected
in-process) and prints their provenance -- file, this commit, the profiled step time -- under `replayed_from`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "incomplete_multimodal_fusion_amd", "bench.py"]).decode().strip())
for stem in ("sq_step", "pmc_hbm"):
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, stem))
    if not os.path.isfile(path):
        continue
    js = json.load(open(path))
    js["git"] = head + ("+uncommitted" if dirty else "")
    json.dump(js, open(path, "w"), indent=1)
    print("stamped", path, js["git"])
