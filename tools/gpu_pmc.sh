#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> <COUNTER> [bench args]  -> per-kernel max of the counter
tag=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/prof/${tag}_$ctr -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 "$@" > gpurun_out/prof/${tag}_$ctr.log 2>&1
f=$(ls gpurun_out/prof/${tag}_$ctr/*/*_counter_collection.csv | head -1)
python3 - "$f" $ctr <<'PY'
import csv,sys,re
from collections import defaultdict
mx=defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]!=sys.argv[2]: continue
    n=re.sub(r"\(anonymous namespace\)::","",re.sub(r"^void ","",r["Kernel_Name"]))[:60]
    mx[n]=max(mx[n],float(r["Counter_Value"]))
for k,v in sorted(mx.items(), key=lambda kv:-kv[1])[:14]:
    print("%-62s %s=%.1f"%(k,sys.argv[2],v))
PY
rm -rf gpurun_out/prof/${tag}_$ctr
