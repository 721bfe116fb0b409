#!/bin/bash
# Compile one csrc/*.hip for gfx950 and print the compiler's per-kernel resource usage (registers, spills, scratch).
# usage: tools/hip_resources.sh 3dhumangan_amd/csrc/field_x3t.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Wno-inline-asm \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /tmp/$(basename "$src").o 2>&1 |
  grep -E "error|warning:|Function Name|VGPRs:|AGPRs|SGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" |
  sed 's/.*remark: *//; s/ *\[-Rpass-analysis=kernel-resource-usage\]//' | awk '/Function Name/{printf "\n%s ", $3; next} {printf "| %s ", $0}'; echo
