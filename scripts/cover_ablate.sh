#!/bin/bash
# k_cover's duration with parts of it switched off (experiment library libhvrt_x.so; HVRT_COVER_EXPERIMENT: 1 no marks, 2 small rectangles only)
cd /tmp && export TMPDIR=/tmp
export HVRT_LIB=/root/repo/raytracer_amd/lib/libhvrt_x.so
for x in 0 1 2 3; do
  export HVRT_COVER_EXPERIMENT=$x
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/cov_x$x -o cov --output-format csv -- python3 /root/repo/scripts/cover_stats.py ${1:-c2} 5 > /dev/null 2>&1 || exit 1
  echo "experiment $x: $(grep k_cover\( /root/repo/gpurun_out/cov_x$x/cov_kernel_stats.csv | cut -d, -f4-9)"
done
