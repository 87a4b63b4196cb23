#!/bin/bash
# instruction-mix counter pass over the attention micro-benchmark:  gpurun -- 'bash tools/attn_pmc2.sh "5" [split] [tag]'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=${1:-0}; SPLIT=${2:-128,128,128}; TAG=${3:-b}
O=gpurun_out/r3; mkdir -p $O; rm -rf $O/pmc_raw
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_raw -- python3 tools/bench_attn.py --variants $V --split $SPLIT --rounds 1 --iters 2 > $O/pmc2_$TAG.log 2>&1
rc=$?
python tools/pmc_kernels.py "$(ls $O/pmc_raw/*/*counter_collection.csv | head -1)" mha > $O/pmc2_$TAG.txt
cat $O/pmc2_$TAG.txt
rm -rf $O/pmc_raw
exit $rc
