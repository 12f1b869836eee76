#!/bin/bash
# Everything profiles/ is installed from, in two GPU calls (scripts/install_profiles.py runs in between and after):
#   gpurun -- scripts/collect_profiles.sh <tag> pmc      counter passes: config 2 traversal, config 5 build
#   gpurun -- scripts/collect_profiles.sh <tag> bench    bench lines (c2 default, c3, c4, c5) and kernel-trace summaries
tag=$1; step=${2:-pmc}
O=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
if [ "$step" = pmc ]; then
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render render || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_build build || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_build_c2 build_c2 || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render_c4 render_c4 || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render_c3 render_c3 || exit 1
  # the traversal kernel with every tile walked: where it is bound by throughput
  export HVRT_TILE_COVER=0
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render_no_cover render || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render_c4_no_cover render_c4 || exit 1
  $GRAFT_REPO_ROOT/scripts/pmc_passes.sh $tag/pmc_render_c3_no_cover render_c3 || exit 1
  exit 0
fi
rm -rf $O/kt_bench $O/kt_c3 $O/kt_c4 $O/kt_c5
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || { tail -5 $O/bench_c2.err; exit 1; }
timeout -k 10 300 python bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err || { tail -5 $O/bench_c3.err; exit 1; }
timeout -k 10 300 python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err || { tail -5 $O/bench_c4.err; exit 1; }
timeout -k 10 300 python bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err || { tail -5 $O/bench_c5.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --moving-camera > $O/bench_c2_moving_camera.json 2> $O/bench_c2_mov.err || { tail -5 $O/bench_c2_mov.err; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/kt_bench.err || { tail -5 $O/kt_bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline > $O/kt_c3.json 2> $O/kt_c3.err || { tail -5 $O/kt_c3.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c4 --no-cpu-baseline > $O/kt_c4.json 2> $O/kt_c4.err || { tail -5 $O/kt_c4.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu-baseline > $O/kt_c5.json 2> $O/kt_c5.err || { tail -5 $O/kt_c5.err; exit 1; }
echo "bench + kernel traces done"
