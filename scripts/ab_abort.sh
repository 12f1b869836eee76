#!/bin/bash
# launches without history under HVRT_TILE_ABORT_US
export HVRT_TILE_ORDER=0 HVRT_TILE_COVER_CACHE=0
for us in ${1:-"500 350"}; do for w in ${2:-"c2 c3b c4"}; do
  n=10; [ $w = c4 ] && n=4
  echo -n "abort $us $w: "; HVRT_TILE_ABORT_US=$us timeout -k 10 150 python scripts/time_render.py $w $n 2>&1 | grep "render " || exit 1
done; done
