#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_kstats.sh <tag> [bench args]
# kernel-trace stats of bench.py -> gpurun_out/prof/<tag>_kernel_stats.csv (+ top lines on stdout)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof/${tag}_bench.log 2>&1
f=$(ls gpurun_out/prof/$tag/*/*_kernel_stats.csv | head -1)
cp "$f" gpurun_out/prof/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls=%4s avg_us=%9.1f tot_ms=%8.2f %5s%%"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof/${tag}_bench.log
