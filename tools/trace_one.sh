#!/bin/bash
# Developer tool (GPU box, repo root): kernel stats + last-step timeline (with the launch sequence) of ONE bench workload.
#   tools/trace_one.sh <tag> <workload> [bench args...]
TAG=$1; W=$2; shift 2; R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/$TAG/prof_$W
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $R/bench.py --workload $W --large none --no-cpu-baseline --no-repeat --steps 6 --warmup 2 --class-steps 0 "$@" > $R/gpurun_out/$TAG/prof_$W.log 2>&1
T=$(find $D -name "*kernel_trace.csv" | head -1); S=$(find $D -name "*kernel_stats.csv" | head -1)
python $R/tools/trace_step.py $T k_hh_update --seq > $R/gpurun_out/$TAG/timeline_$W.txt
cp $S $R/gpurun_out/$TAG/kernel_stats_$W.csv
rm -rf $D
