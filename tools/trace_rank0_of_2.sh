#!/bin/bash
# Developer tool (GPU box, repo root): two ranks on the one GPU, rank 0 under rocprofv3 --kernel-trace: the kernels a rank runs per
# timestep on the distributed path (durations of the exchange kernels include waiting for the other rank, which shares the GPU)
W=${1:-square512}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/dist_trace_$W; mkdir -p $O
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 KNP_DIST_BACKEND=gloo
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 2 --workload $W --large none --no-cpu-baseline --no-repeat --steps 6 --warmup 2"
RANK=1 LOCAL_RANK=1 python $R/bench.py $ARGS > $O/rank1.log 2>&1 &
P1=$!
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py $ARGS > $O/rank0.log 2>&1
wait $P1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $T > $O/timeline.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
cat $O/timeline.txt
