#!/bin/bash
# Developer tool (GPU box, repo root): kernel stats + last-step timeline of the bench workloads into gpurun_out/<tag>/
# (the whole run is traced, the one-off setup kernels of torch included: the timeline is cut out of the last step)
TAG=${1:-trace}; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for w in square512 cube64 cube136 tissue3d_97_24_w1; do
  D=$R/gpurun_out/$TAG/prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $R/bench.py --workload $w --large none --no-cpu-baseline --no-repeat --steps 6 --warmup 2 --class-steps 0 --set amg_setup=host > $R/gpurun_out/$TAG/prof_$w.log 2>&1
  T=$(find $D -name "*kernel_trace.csv" | head -1); S=$(find $D -name "*kernel_stats.csv" | head -1)
  python $R/tools/trace_step.py $T k_hh_update $([ $w = square512 ] && echo --seq) > $R/gpurun_out/$TAG/timeline_$w.txt
  cp $S $R/gpurun_out/$TAG/kernel_stats_$w.csv
  rm -rf $D
done
