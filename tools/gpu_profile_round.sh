#!/bin/bash
# Collects the judged profile artefacts of a round on the GPU box (run from the repo root):
#   tools/gpu_profile_round.sh r03
# -> gpurun_out/prof/<tag>_bench.json          the JSON line of the default `python3 bench.py` (C4 headline + C2, C3)
#    gpurun_out/prof/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the SAME command
#    gpurun_out/prof/<tag>_pmc_fetch.csv / _pmc_write.csv / _pmc_valu.csv   three separate --pmc passes (counters only)
# then: python3 tools/pmc_summary.py gpurun_out/prof/<tag> profiles/<tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
python3 bench.py > gpurun_out/prof/${tag}_bench.json 2> gpurun_out/prof/${tag}_bench.err || exit 1
echo "bench done"
# (--no-full-check: the full-size check re-runs the step with shells widened to the corners of the k cube -- no row cut -- and
#  computes float64 moments with torch: neither belongs into the per-kernel averages or the traffic of the timed step)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${tag}_kt -- python3 bench.py --no-cpu-baseline --no-full-check > gpurun_out/prof/${tag}_kt.log 2>&1 || exit 1
cp $(ls gpurun_out/prof/${tag}_kt/*/*_kernel_stats.csv | head -1) gpurun_out/prof/${tag}_kernel_stats.csv
echo "kernel trace done"
for c in FETCH_SIZE:fetch WRITE_SIZE:write "VALUBusy LDSBankConflict MemUnitStalled":valu; do
  ctr=${c%%:*}; short=${c##*:}
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/prof/${tag}_p$short -- python3 bench.py --steps 1 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-parity --no-full-check > gpurun_out/prof/${tag}_p$short.log 2>&1 || exit 1
  cp $(ls gpurun_out/prof/${tag}_p$short/*/*_counter_collection.csv | head -1) gpurun_out/prof/${tag}_pmc_$short.csv
  rm -rf gpurun_out/prof/${tag}_p$short
  echo "pmc $short done"
done
rm -rf gpurun_out/prof/${tag}_kt
python3 tools/pmc_summary.py gpurun_out/prof/${tag} gpurun_out/prof/${tag}
head -c 1200 gpurun_out/prof/${tag}_bench.json; echo
head -14 gpurun_out/prof/${tag}_kernel_stats.csv | cut -c1-160
