#!/bin/bash
# Counter passes for the hvrt kernels on config 2 (each pass its own rocprofv3 run, as gpurun requires).
# usage: scripts/pmc_passes.sh <outdir-name under gpurun_out>
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 scripts/prof_render.py > $O/p$i.log 2>&1 || { echo "pass $i ($P) failed"; tail -3 $O/p$i.log; }
done
python3 scripts/parse_pmc.py $O > $O/summary.json
python3 - <<PY
import json
r=json.load(open("$O/summary.json"))
for k,v in r.items():
    if "render" in k or "sort" in k: print(k, {c:round(x["mean"]/1e6,2) for c,x in v.items()})
PY
