#!/bin/bash
# device assembly + resource usage of one csrc file:  tools/asm_of.sh mha_sh.hip [/tmp/out.s]
cd "$(dirname "$0")/../incomplete_multimodal_fusion_amd/csrc"
out=${2:-/tmp/$(basename $1 .hip).s}
extra=""
case $1 in mha_bf16.hip|mha_sh.hip) extra="-fno-honor-nans -mno-amdgpu-ieee";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I. $extra -S --cuda-device-only $1 -o $out -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" 
echo "asm: $out"
