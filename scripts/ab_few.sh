#!/bin/bash
# launches without history (HVRT_TILE_ORDER=0, HVRT_TILE_COVER_CACHE=0) under HVRT_FEW_TILES, at several frame sizes
fews=${1:-"0 1024"}; sizes=${2:-"800x600 1920x1080"}; loads=${3:-"c2"}
export HVRT_TILE_ORDER=0 HVRT_TILE_COVER_CACHE=0
for f in $fews; do
  export HVRT_FEW_TILES=$f
  for w in $loads; do for wh in $sizes; do
    echo -n "few $f $w $wh: "; HVRT_TIME_WH=$wh timeout -k 10 150 python scripts/time_render.py $w 10 2>&1 | grep "render " || exit 1
  done; done
done
