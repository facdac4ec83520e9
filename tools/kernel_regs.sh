#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> [name filter]  -> VGPR / AGPR / occupancy / spills per kernel
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Idisenlink_amd/csrc -c "$f" -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//; s/^[^ ]*:[0-9]*:[0-9]*: *//' \
 | awk '/Function Name/{if(line)print line; line=$3; next}{gsub(/^ +/,""); line=line" | "$0}END{print line}' | c++filt | grep -E "$pat" | sed 's/(float const.*)//' | cut -c1-200
