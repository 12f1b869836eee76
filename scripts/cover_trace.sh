#!/bin/bash
# kernel trace of scripts/cover_stats.py <workload>: per-kernel durations with the tile cover on
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/cov_t -o cov --output-format csv -- python3 /root/repo/scripts/cover_stats.py ${1:-c2} ${2:-10} > /root/repo/gpurun_out/cov_t.log 2>&1 || exit 1
grep -v amdgpu.ids /root/repo/gpurun_out/cov_t.log | tail -3
head -7 /root/repo/gpurun_out/cov_t/cov_kernel_stats.csv | sed 's/(hvrt::SceneDev, hvrt::RenderArgs)//; s/hvrt::(anonymous namespace):://' | cut -c1-150
