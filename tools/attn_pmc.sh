#!/bin/bash
# SQ counter pass over the attention micro-benchmark:  gpurun -- 'bash tools/attn_pmc.sh "0,5" [split]'
# Writes gpurun_out/r3/pmc_<tag>.txt (per-kernel averages and ratios, tools/pmc_kernels.py).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=${1:-0}; SPLIT=${2:-128,128,128}; TAG=${3:-a}
O=gpurun_out/r3; mkdir -p $O; rm -rf $O/pmc_raw
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_raw -- python3 tools/bench_attn.py --variants $V --split $SPLIT --rounds 1 --iters 2 > $O/pmc_$TAG.log 2>&1
rc=$?
python tools/pmc_kernels.py "$(ls $O/pmc_raw/*/*counter_collection.csv | head -1)" mha > $O/pmc_$TAG.txt
cat $O/pmc_$TAG.txt
rm -rf $O/pmc_raw
exit $rc
