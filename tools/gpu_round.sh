#!/bin/bash
# One gpurun call: GPU tests, the default bench line, kernel-trace stats and SQ / HBM counter passes of the same command.
# A step that is killed or times out (exit 124 / >= 128) ends the call: no further GPU step is started after it.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh [tests] [bench] [stats] [sq] [hbm] [attn]'
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
steps="${@:-tests bench stats sq}"
guard() { rc=$1; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[gpu_round] step '$2' was killed (rc $rc): stopping" | tee -a $OUT/status; exit $rc; fi; echo "[gpu_round] $2 rc=$rc" | tee -a $OUT/status; }
BENCH_PROF="bench.py --steps 3 --warmup 2 --no-cpu-baseline --legs none"
for s in $steps; do
  case $s in
    tests) timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/tests.log 2>&1; guard $? tests; tail -5 $OUT/tests.log ;;
    testsall) timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $OUT/tests.log 2>&1; guard $? tests; tail -15 $OUT/tests.log ;;
    bench) timeout -k 10 400 python bench.py --steps 12 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err; guard $? bench; cat $OUT/bench.json ;;
    stats) rm -rf $OUT/stats; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $BENCH_PROF > $OUT/stats.log 2>&1; guard $? stats ;;
    sq) rm -rf $OUT/sq; timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $BENCH_PROF > $OUT/sq.log 2>&1; guard $? sq ;;
    hbm) rm -rf $OUT/fetch $OUT/write
         timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $BENCH_PROF > $OUT/fetch.log 2>&1; guard $? fetch
         timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $BENCH_PROF > $OUT/write.log 2>&1; guard $? write ;;
    attn) timeout -k 10 300 python tools/bench_attn.py --variants ${ATTN_VARIANTS:-0} > $OUT/attn.log 2>&1; guard $? attn; cat $OUT/attn.log
          timeout -k 10 300 python tools/bench_attn.py --variants ${ATTN_VARIANTS:-0} --split 201,64,119 >> $OUT/attn.log 2>&1; guard $? attn2; tail -3 $OUT/attn.log ;;
    attnx) for sp in ${ATTN_SPLITS:-384 64,64,64}; do timeout -k 10 300 python tools/bench_attn.py --variants ${ATTN_VARIANTS:-0} --split $sp --rounds 3 > $OUT/attnx_$sp.log 2>&1; guard $? attnx; echo "split $sp"; grep "^variant" $OUT/attnx_$sp.log; done ;;
    attnsq) rm -rf $OUT/attnsq; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/attnsq -- python3 tools/bench_attn.py --variants ${ATTN_VARIANTS:-0} --rounds 1 --iters 3 > $OUT/attnsq.log 2>&1; guard $? attnsq ;;
    profiles) du -sh $OUT/stats $OUT/sq $OUT/fetch $OUT/write 2>/dev/null; bash tools/make_profiles.sh ${PROFILE_TAG:-r02} 5 $OUT/profiles; guard $? profiles; rm -rf $OUT/stats $OUT/sq $OUT/fetch $OUT/write ;;
    pick) eval "timeout -k 10 1000 python -m pytest $PYTEST_ARGS -q --timeout 600" > $OUT/pick.log 2>&1; guard $? pick; tail -25 $OUT/pick.log ;;   # eval: PYTEST_ARGS may carry a quoted -k expression
    smoke) timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; guard $? smoke; tail -2 $OUT/smoke.log ;;
    *) echo "unknown step $s" ;;
  esac
done
find $OUT -name "*.csv" | head -20
