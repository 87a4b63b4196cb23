#!/bin/bash
# Condense the newest rocprofv3 outputs of tools/gpu_round.sh (steps: stats sq hbm) into profiles/<round>_*.
#   bash tools/make_profiles.sh r02 [steps] [dest dir]      (tools/gpu_round.sh step `profiles` runs it ON the GPU box into
#   gpurun_out/round/profiles and drops the raw CSVs, which can exceed what gpurun copies back)
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/round; N=${1:?round tag, e.g. r02}; STEPS=${2:-5}; DEST=${3:-profiles}
mkdir -p $DEST
newest() { ls -t $1 | head -1; }
S=$(newest "$R/stats/*/*kernel_stats.csv"); Q=$(newest "$R/sq/*/*counter_collection.csv")
F=$(newest "$R/fetch/*/*counter_collection.csv"); W=$(newest "$R/write/*/*counter_collection.csv")
CMD="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --legs none"
hdr() { echo "# $1"; echo; echo "Command (on the MI355X box, through tools/gpu_round.sh): \`$2 -- $CMD\`"; echo; }
{ hdr "Round ${N#r} - kernel time of the bench step" "rocprofv3 --kernel-trace --stats --output-format csv"; python tools/prof_summary.py stats $S $STEPS; } > $DEST/${N}_kernel_stats.md
cp $S $DEST/${N}_kernel_stats.csv
{ hdr "Round ${N#r} - HBM traffic per launch (two passes)" "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE --output-format csv"; python tools/prof_summary.py pmc $F $W; } > $DEST/${N}_pmc_hbm.md
python tools/prof_summary.py json $F $W $STEPS > $DEST/${N}_pmc_hbm.json
{ hdr "Round ${N#r} - SQ counters of the bench step, per kernel and whole step" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv"; python tools/prof_summary.py sq $Q $STEPS; } > $DEST/${N}_sq_step.md
python tools/prof_summary.py sq $Q $STEPS json > $DEST/${N}_sq_step.json
# parity verdicts of the GPU tests of this call (tests/parity.py appends one JSON line per compare()), as a table
if [ -f gpurun_out/parity_log.jsonl ]; then python tools/parity_table.py gpurun_out/parity_log.jsonl > $DEST/${N}_parity.md; fi
echo "$S $Q $F $W"
