#!/bin/bash
# kernel trace of an orbit (scripts/time_moving.py): the timeline of its last frame
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/mov_t -o mov --output-format csv -- python3 /root/repo/scripts/time_moving.py 1 ${1:-8} > /root/repo/gpurun_out/mov_t.log 2>&1 || exit 1
grep "moving" /root/repo/gpurun_out/mov_t.log
