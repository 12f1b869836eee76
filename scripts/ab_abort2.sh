#!/bin/bash
# with history: the launch list under HVRT_TILE_ABORT_US (a tile given up on stays split: the time is also a split bar)
for us in ${1:-"500 300"}; do for w in ${2:-"c2 c3b c4"}; do
  n=12; [ $w = c4 ] && n=5
  echo -n "hist abort $us $w: "; HVRT_TIME_LIST=1 HVRT_TILE_ABORT_US=$us timeout -k 10 150 python scripts/time_render.py $w $n 2>&1 | grep -E "launches" || exit 1
done; done
