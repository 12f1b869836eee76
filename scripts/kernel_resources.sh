#!/bin/bash
# Registers, scratch and occupancy of every kernel in a libhvrt source file, as the compiler reports them
# (no GPU needed):  scripts/kernel_resources.sh hvrt_trace.hip [filter on the demangled name] [extra hipcc flags]
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
src=${1:-hvrt_trace.hip}; filt=${2:-.}; shift 2
extra=""; [ "$src" = hvrt_trace.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I../../include $extra "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kr_$$.o 2>&1 |
  grep -E "Function Name|TotalSGPRs|  VGPRs:|ScratchSize|Occupancy" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - |
  while IFS=$'\t' read -r name sg vg sc oc; do
    n=$(echo "$name" | sed -E 's/Function Name: //' | c++filt | sed -E 's/hvrt::\(anonymous namespace\):://; s/\(hvrt::.*//; s/^void //')
    echo "$n | $sg | $vg | $sc | $oc"
  done | grep -E "$filt"
rm -f /tmp/kr_$$.o
