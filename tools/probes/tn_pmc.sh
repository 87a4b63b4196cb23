#!/bin/bash
# Memory-path counters of the own transposing weight-gradient GEMM against the library's and against the own NT kernel (FF1 shapes):
# address translation (UTCL1), TCP -> TCC latency / stalls, TCC -> fabric latency / stalls.  One rocprofv3 --pmc pass per group of three
# (more TCP counters than that in one pass: 'exceeds the capabilities of the hardware').
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r4/tnpmc; mkdir -p $OUT
CSETS=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
        "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
        "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"
        "TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum")
i=0
for G in "${CSETS[@]}"; do
  i=$((i+1))
  for prog in "tools/bench_wgrad.py FF1" "tools/bench_gemm.py FF1"; do
    tag=$(echo $prog | tr '/ .' '___')
    rm -rf $OUT/p$i$tag
    timeout -k 10 90 rocprofv3 --pmc $G GRBM_GUI_ACTIVE --output-format csv -d $OUT/p$i$tag -- python3 $prog > $OUT/p$i$tag.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "killed rc $rc"; exit $rc; fi
    echo "== group $i  $prog (rc $rc)"
    python tools/probes/pmc_fetch.py $OUT/p$i$tag | grep -i "gemm\|Cijk" | cut -c1-900
    rm -rf $OUT/p$i$tag
  done
done
