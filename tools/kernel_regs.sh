#!/bin/bash
# VGPRs / scratch / occupancy / LDS of every kernel of one csrc file, from the compiler's own remarks (no GPU needed):
#   tools/kernel_regs.sh pipeline.hip [extra hipcc flags]
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -w "$@" -x hip --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c dabstar_amd/csrc/$F -o /dev/null 2>&1 |
  grep -E "remark: +(Function )?Name:|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Name:/ {if (l) print l; l=$NF; next} {gsub(/ \[[^]]*\]/, ""); l=l"  "$0} END {print l}' | sed -E 's/_ZN4dabx[0-9]+//; s/ENS_9EngineDev[A-Za-z0-9_]*//'
