#!/bin/bash
# Developer tool: average duration of the kernels matching a pattern under rocprofv3 (run on a GPU box from the repo root).
#   tools/kernel_time.sh <workload> <pattern> [bench args...]      (environment variables are inherited by the bench)
R=${GRAFT_REPO_ROOT:-$PWD}; W=$1; PAT=$2; shift 2
OUT=$R/gpurun_out/kt_$$; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --workload $W --large none --no-cpu-baseline --no-repeat --steps 4 --warmup 2 --set amg_setup=host "$@" > $OUT/log.txt 2>&1
S=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$S" "$PAT" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f'{float(r["AverageNs"])/1e3:10.2f} us x {r["Calls"]:>6}  {r["Name"][:110]}')
PY
rm -rf $OUT
