#!/bin/bash
# Developer tool (GPU box, repo root): the unprofiled bench lines and the PMC passes behind profiles/r03_*  ->  gpurun_out/<tag>/
TAG=${1:-refresh}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_square512.json 2> $O/bench_square512.err
python bench.py --workload cube64 --large none > $O/bench_cube64.json 2> $O/bench_cube64.err
python bench.py --workload cube64 --large none --no-cpu-baseline --profile-all > $O/bench_cube64_profall.json 2> /dev/null
KNP_ASM_FULL=1 python bench.py --workload cube64 --large none --no-cpu-baseline --profile-all > $O/bench_cube64_asmfull.json 2> /dev/null
python bench.py --workload cube136 --large none --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_cube136.json 2> /dev/null
python bench.py --workload tissue3d_97_24_w1 --large none --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_tissue97.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for w in square512 cube64 cube136; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python $R/bench.py --workload $w --no-cpu-baseline --large none --no-repeat --steps 3 --warmup 1 --class-steps 0 > $O/pmc_${w}_$c.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py $O/pmc_traffic.json square512=$O/pmc_square512_FETCH_SIZE,$O/pmc_square512_WRITE_SIZE cube64=$O/pmc_cube64_FETCH_SIZE,$O/pmc_cube64_WRITE_SIZE cube136=$O/pmc_cube136_FETCH_SIZE,$O/pmc_cube136_WRITE_SIZE
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
tail -c 400 $O/bench_square512.json
