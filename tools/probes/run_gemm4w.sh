#!/bin/bash
# gpurun -- 'bash tools/probes/run_gemm4w.sh'   builds the 4-wave GEMM probe (both DMA placements) and the 8-wave probe, runs them on one box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value"
$H tools/probes/gemm4w_probe.hip -o /tmp/g4w_a || exit 1
$H -DDMA_FIRST=0 tools/probes/gemm4w_probe.hip -o /tmp/g4w_b || exit 1
$H tools/probes/gemm8p_probe.hip -o /tmp/g8p || exit 1
LOG=gpurun_out/g4w.log; : > $LOG
run() { echo "== $*" >> $LOG; timeout -k 10 100 "$@" >> $LOG 2>&1; rc=$?; echo "rc=$rc" >> $LOG; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then cat $LOG; exit $rc; fi; }
run /tmp/g4w_a 163840 4096 0
run /tmp/g8p 163840 4096 1 0
run /tmp/g4w_b 163840 4096 0
run /tmp/g4w_a 163840 4096 1
run /tmp/g4w_a 163840 4096 2
run /tmp/g4w_a 163840 4096 3
run /tmp/g8p 163840 4096 1 3
run /tmp/g4w_a 163840 768 0
run /tmp/g8p 163840 768 1 0
run /tmp/g4w_a 163840 4096 0
run /tmp/g8p 163840 4096 1 0
cat $LOG
