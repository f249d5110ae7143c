#!/bin/bash
# Collects the judged profile artefacts for one round on the GPU box (run from the repo root):
#   tools/gpu_profile_round.sh r01
# -> gpurun_out/prof/<tag>_kernel_stats.csv, <tag>_pmc_fetch.csv, <tag>_pmc_write.csv, <tag>_bench.json
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
python3 bench.py > gpurun_out/prof/${tag}_bench.json 2> gpurun_out/prof/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${tag}_kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof/${tag}_kt.log 2>&1
cp $(ls gpurun_out/prof/${tag}_kt/*/*_kernel_stats.csv | head -1) gpurun_out/prof/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/${tag}_pf -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > gpurun_out/prof/${tag}_pf.log 2>&1
cp $(ls gpurun_out/prof/${tag}_pf/*/*_counter_collection.csv | head -1) gpurun_out/prof/${tag}_pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/${tag}_pw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > gpurun_out/prof/${tag}_pw.log 2>&1
cp $(ls gpurun_out/prof/${tag}_pw/*/*_counter_collection.csv | head -1) gpurun_out/prof/${tag}_pmc_write.csv
rm -rf gpurun_out/prof/${tag}_kt gpurun_out/prof/${tag}_pf gpurun_out/prof/${tag}_pw
head -c 1500 gpurun_out/prof/${tag}_bench.json; echo
head -12 gpurun_out/prof/${tag}_kernel_stats.csv | cut -c1-150
