#!/bin/bash
# gpurun -- 'bash tools/probes/run_dma_group.sh'   8-wave GEMM probe: DMA issue by all waves (product) vs by the leading wave group only
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value"
LOG=gpurun_out/dma_group.log; : > $LOG
$H tools/probes/gemm8p_probe.hip -o /tmp/g8p_base || exit 1
$H -DDMA_GROUP_A tools/probes/gemm8p_probe.hip -o /tmp/g8p_ga || exit 1
for r in 1 2 3; do for v in base ga; do for n in 4096 768; do echo "== $v N $n" >> $LOG; timeout -k 10 100 /tmp/g8p_$v 163840 $n 1 0 >> $LOG 2>&1; rc=$?; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then cat $LOG; exit $rc; fi; done; done; done
echo "== ga diag2" >> $LOG; /tmp/g8p_ga 163840 4096 1 2 >> $LOG 2>&1
echo "== base diag2" >> $LOG; /tmp/g8p_base 163840 4096 1 2 >> $LOG 2>&1
grep -E "^== |mean|check" $LOG | paste - - - | sed -E 's/own 256x256x64 8-phase GEMM \(persistent, rolling epilogue\) //; s/\(gate 1150; random \[-1,1\) operands\)//'
