#!/bin/bash
# A/B of experiment libraries over the cases that matter: with history (c2, c3b, c4), without (800x600, 4K), an orbit
libs=${1:-"sr0 default"}
for l in $libs; do
  if [ "$l" = default ]; then unset HVRT_LIB; else export HVRT_LIB=$GRAFT_REPO_ROOT/raytracer_amd/lib/libhvrt_$l.so; fi
  for w in c2 c3b c4; do n=20; [ $w = c4 ] && n=5; [ $w = c3b ] && n=10
    echo -n "$l hist $w: "; timeout -k 10 150 python scripts/time_render.py $w $n 2>&1 | grep "render " || exit 1; done
  for w in c2 c3b; do for wh in 800x600 3840x2160; do
    echo -n "$l cold $w $wh: "; HVRT_TILE_ORDER=0 HVRT_TILE_COVER_CACHE=0 HVRT_TIME_WH=$wh timeout -k 10 150 python scripts/time_render.py $w 10 2>&1 | grep "render " || exit 1; done; done
  echo -n "$l orbit: "; timeout -k 10 150 python scripts/time_moving.py 1 20 2>&1 | grep -i "moving" | head -2 | tr "\n" " "; echo
done
