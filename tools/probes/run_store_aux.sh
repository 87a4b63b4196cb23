#!/bin/bash
# gpurun -- 'bash tools/probes/run_store_aux.sh'   the 8-wave GEMM probe with every cache policy of its C stores (aux bits: 1 sc0, 2 nt, 16 sc1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value"
LOG=gpurun_out/store_aux.log; : > $LOG
for a in 2 0 1 3 16 17 18 19; do $H -DSTORE_AUX=$a tools/probes/gemm8p_probe.hip -o /tmp/g8p_$a || exit 1; done
for r in 1 2; do for a in 2 0 1 3 16 17 18 19; do echo "== aux $a" >> $LOG; timeout -k 10 100 /tmp/g8p_$a 163840 4096 1 0 >> $LOG 2>&1; rc=$?; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then cat $LOG; exit $rc; fi; done; done
grep -E "^== aux|mean" $LOG | paste - - | sed -E 's/own 256x256x64 8-phase GEMM \(persistent, rolling epilogue\)  M 163840 N 4096 K 768 ://'
