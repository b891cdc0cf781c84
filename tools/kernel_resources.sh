#!/bin/bash
# Register / spill / LDS / occupancy figures of every kernel of one translation unit, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage), with the product's flags.  usage: tools/kernel_resources.sh b32_setup.hip [extra -D flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" \
  -c bonnie-32_amd/csrc/$src -o /tmp/kr_$$.o 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | \
  awk '/Function Name:/ {name=$NF} / TotalSGPRs:/ {sg=$NF} / VGPRs:/ {v=$NF} / AGPRs:/ {a=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /VGPRs Spill/ {sp=$NF} /LDS Size/ {print name, "sgpr="sg, "vgpr="v, "agpr="a, "scratch="sc, "vspill="sp, "occ="oc, "lds="$NF}' | c++filt | sort
rm -f /tmp/kr_$$.o
