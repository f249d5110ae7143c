cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_VALU_FLOPS_FP64"; do
rm -rf gpurun_out/prof/nnpmc
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/prof/nnpmc -- python3 tools/exp_nn.py 512 1e7 > gpurun_out/prof/nnpmc.log 2>&1
f=$(ls gpurun_out/prof/nnpmc/*/*_counter_collection.csv | head -1)
python3 - "$f" $ctrs <<'PY'
import csv,sys,re
from collections import defaultdict
ctrs=sys.argv[2:]
disp=defaultdict(dict); name={}
for r in csv.DictReader(open(sys.argv[1])):
    d=r["Dispatch_Id"]; disp[d][r["Counter_Name"]]=float(r["Counter_Value"]); name[d]=r["Kernel_Name"]
best=None
for d,c in disp.items():
    if "nn_query" in name[d] and (best is None or c.get(ctrs[0],0)>best.get(ctrs[0],0)): best=c
print(" ".join("%s=%.4g"%(k,best.get(k,float('nan'))) for k in ctrs))
PY
done
rm -rf gpurun_out/prof/nnpmc
