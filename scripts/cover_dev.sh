#!/bin/bash
# developer loop for the tile cover: parity tests, then configs 2 / 4 / 3 with the cover cached (a camera that stands) and computed per launch
bash scripts/dev_gpu.sh ${1:-cov} || exit 1
export HVRT_TILE_COVER_CACHE=0
echo "== cover computed by every launch"
timeout -k 10 120 python scripts/time_render.py c2 20 2>&1 | grep render && timeout -k 10 120 python scripts/time_render.py c4 5 2>&1 | grep render && timeout -k 10 120 python scripts/time_render.py c3 10 2>&1 | grep render
