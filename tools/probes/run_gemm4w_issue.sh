#!/bin/bash
# gpurun -- 'bash tools/probes/run_gemm4w_issue.sh'   (builds and runs the 4-wave issue-cost probe; output in gpurun_out/g4i.log)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probes/gemm4w_issue_probe.hip -o /tmp/g4i || exit 1
timeout -k 10 120 /tmp/g4i > gpurun_out/g4i.log 2>&1; rc=$?
cat gpurun_out/g4i.log
exit $rc
