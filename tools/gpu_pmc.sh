#!/bin/bash
# usage (GPU box, repo root): tools/gpu_pmc.sh <tag> "<CTR1 CTR2 ...>" [bench args]
# one rocprofv3 --pmc pass (counters only, no tracing) of a short bench run; prints, per kernel, the value of every
# counter for the launch with the largest first counter (= the main launch of that kernel)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/prof/${tag}_pmc -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-other-configs --profile-steps 1 "$@" > gpurun_out/prof/${tag}_pmc.log 2>&1
f=$(ls gpurun_out/prof/${tag}_pmc/*/*_counter_collection.csv | head -1)
python3 - "$f" $ctrs <<'PY'
import csv,sys,re
from collections import defaultdict
ctrs=sys.argv[2:]
disp=defaultdict(dict); name={}
for r in csv.DictReader(open(sys.argv[1])):
    d=r["Dispatch_Id"]; disp[d][r["Counter_Name"]]=float(r["Counter_Value"])
    name[d]=re.sub(r"\(anonymous namespace\)::","",re.sub(r"^void ","",r["Kernel_Name"]))[:48]
best={}
for d,c in disp.items():
    n=name[d]
    if n.startswith("at::") or "rocclr" in n: continue
    if n not in best or c.get(ctrs[0],0)>best[n].get(ctrs[0],0): best[n]=c
for n,c in sorted(best.items(), key=lambda kv:-kv[1].get(ctrs[0],0))[:12]:
    print("%-50s "%n+" ".join("%s=%.4g"%(k,c.get(k,float('nan'))) for k in ctrs))
PY
rm -rf gpurun_out/prof/${tag}_pmc
