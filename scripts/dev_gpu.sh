#!/bin/bash
# Developer loop on the GPU box: render parity tests, then kernel timings of configs 2, 4 and 3.
#   gpurun -- scripts/dev_gpu.sh [tag]
tag=${1:-dev}
timeout -k 10 600 python -u -m pytest tests/test_gpu_render.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/$tag.test.log 2>&1
rc=$?; tail -3 gpurun_out/$tag.test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python scripts/time_render.py c2 20 2>&1 | grep -v amdgpu.ids && timeout -k 10 120 python scripts/time_render.py c4 5 2>&1 | grep render && timeout -k 10 120 python scripts/time_render.py c3 10 2>&1 | grep render
