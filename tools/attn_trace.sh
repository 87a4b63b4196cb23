#!/bin/bash
# per-kernel times of the attention micro-benchmark:  gpurun -- 'bash tools/attn_trace.sh "0" [split]'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=${1:-0}; SPLIT=${2:-128,128,128}
O=gpurun_out/r3; mkdir -p $O; rm -rf $O/trace_raw
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_raw -- python3 tools/bench_attn.py --variants $V --split $SPLIT --rounds 2 --iters 10 > $O/trace.log 2>&1
f=$(ls $O/trace_raw/*/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mha" in r["Name"]:
        print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/trace_raw
