#!/bin/bash
# with history (the default path): launch list under HVRT_FEW_TILES
for f in ${1:-"0 2048"}; do for w in c2 c3b; do for wh in 800x600 1280x720 1920x1080; do
  echo -n "hist few $f $w $wh: "; HVRT_FEW_TILES=$f HVRT_TIME_LIST=1 HVRT_TIME_WH=$wh timeout -k 10 150 python scripts/time_render.py $w 6 2>&1 | grep -E "launches" || exit 1
done; done; done
