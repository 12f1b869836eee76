#!/bin/bash
for f in ${FACTORS:-15 20 30}; do
  echo "factor x10 = $f"
  HVRT_SPLIT_FACTOR_X10=$f timeout -k 10 120 python scripts/time_series.py ${1:-c4} 10 2>&1 | grep "^c" || exit 1
done
