#!/bin/bash
# A/B of experiment libraries on launches without history (HVRT_TILE_ORDER=0, HVRT_TILE_COVER_CACHE=0) at several frame sizes
libs=${1:-"la0 default"}; sizes=${2:-"800x600 1920x1080 3840x2160"}; loads=${3:-"c2"}
export HVRT_TILE_ORDER=0 HVRT_TILE_COVER_CACHE=0
for l in $libs; do
  if [ "$l" = default ]; then unset HVRT_LIB; else export HVRT_LIB=$GRAFT_REPO_ROOT/raytracer_amd/lib/libhvrt_$l.so; fi
  for w in $loads; do for wh in $sizes; do
    echo -n "$l $w $wh: "; HVRT_TIME_WH=$wh timeout -k 10 150 python scripts/time_render.py $w 10 2>&1 | grep "render " || exit 1
  done; done
done
