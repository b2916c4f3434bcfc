#!/bin/bash
# Register, scratch and LDS use per kernel of one source file: tools/regs.sh vrt_kernels [flags...]
f=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage -c $root/raytracing_amd/csrc/$f.hip -o /tmp/regs_$f.o 2>&1 |
  grep -E "Function Name|  VGPRs:|ScratchSize|LDS Size|SGPRs Spill|VGPRs Spill" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' |
  paste - - - - - - | sed 's/Function Name: //' | while read -r n rest; do echo "$(echo $n | c++filt | cut -c1-70) | $rest"; done
