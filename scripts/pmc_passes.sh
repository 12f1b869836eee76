#!/bin/bash
# Counter passes for the hvrt kernels (each pass its own rocprofv3 run, as gpurun requires; limits per pass: 8 SQ, 4 TCC
# with FETCH_SIZE = 3 and WRITE_SIZE = 2, 4 TCP, 2 TA, 2 TD, 2 GRBM).
#   scripts/pmc_passes.sh <outdir-name under gpurun_out> render   -> config 2 traversal (scripts/prof_render.py)
#   scripts/pmc_passes.sh <outdir-name under gpurun_out> render_c3 | render_c4   -> the same kernel on configs 3 and 4 (fewer passes)
#   scripts/pmc_passes.sh <outdir-name under gpurun_out> build | build_c2   -> config 5 / config 2 voxel build (scripts/prof_build.py c5|c2 3)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1
what=${2:-render}
mkdir -p $O
cd /tmp
i=0
if [ "$what" = render ]; then
  prog="scripts/prof_render.py"
  passes=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
          "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
          "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
          "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY")
elif [ "$what" = render_c3 ] || [ "$what" = render_c4 ]; then
  prog="scripts/prof_render.py ${what#render_}"
  passes=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE")
elif [ "$what" = build_c2 ]; then
  prog="scripts/prof_build.py c2 3"
  passes=("FETCH_SIZE" "WRITE_SIZE" "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
else
  prog="scripts/prof_build.py c5 3"
  passes=("FETCH_SIZE" "WRITE_SIZE" "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
fi
for P in "${passes[@]}"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $GRAFT_REPO_ROOT/$prog > $O/p$i.log 2>&1 || { echo "pass $i ($P) failed"; tail -3 $O/p$i.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/scripts/parse_pmc.py $O > $O/summary.json
echo "$what: $i passes done"
