#!/bin/bash
# Same-box A/B of two bench configurations (box-to-box spread is +-1.5 %, larger than most single changes):
#   gpurun -- 'A_ENV="MMAE_HIP_LIB=tools/probes/libmmae_prev.so MMAE_FUSED_CTX=0" B_ENV="" bash tools/ab_bench.sh [rounds]'
# alternates A and B `rounds` times (default 3); prints samples/s and the median step of each run.
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab; mkdir -p $OUT
R=${1:-3}
ARGS="--steps ${AB_STEPS:-20} --warmup 4 --no-cpu-baseline --legs none --tuning-env ${AB_ARGS:-}"
for i in $(seq 1 $R); do
  for side in A B; do
    if [ $side = A ]; then E="$A_ENV"; else E="$B_ENV"; fi
    env $E timeout -k 10 300 python bench.py $ARGS > $OUT/$side$i.json 2> $OUT/$side$i.err
    rc=$?; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "run $side$i killed (rc $rc): stopping"; exit $rc; fi
    python - $OUT/$side$i.json $side$i "$E" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s  %8.1f samples/s  mean %.3f ms  median %.3f ms   [%s]" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("median_ms_per_step", 0), sys.argv[3]))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
  done
done
