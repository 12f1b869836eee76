#!/bin/bash
# A/B of experiment libraries (make -C raytracer_amd/csrc variant NAME=...): the same timings with each
#   gpurun -- scripts/ab_libs.sh "la0 la4" "c2 c3b c4"
libs=${1:-"la0"}; loads=${2:-"c2 c3b c4"}
for l in $libs; do
  if [ "$l" = default ]; then unset HVRT_LIB; else export HVRT_LIB=$GRAFT_REPO_ROOT/raytracer_amd/lib/libhvrt_$l.so; fi
  for w in $loads; do
    n=20; [ $w = c4 ] && n=5; [ $w = c3b ] && n=10
    echo -n "$l $w: "; timeout -k 10 150 python scripts/time_render.py $w $n 2>&1 | grep "render " || exit 1
  done
done
