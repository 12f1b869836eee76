#!/bin/bash
# kernel trace of scripts/time_render.py <workload> <launches>: the timeline of its last launches (scripts/trace_timeline.py)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/ren_t -o ren --output-format csv -- python3 /root/repo/scripts/time_render.py ${1:-c2} ${2:-8} > /root/repo/gpurun_out/ren_t.log 2>&1 || exit 1
grep "render " /root/repo/gpurun_out/ren_t.log
