export VPS_BENCH_NOCHECK=1
for r in 1 2; do for v in base contig; do
  if [ $v = base ]; then unset VPS_LIB_PATH; else export VPS_LIB_PATH=$PWD/tools/exp_libs/lib_$v.so; fi
  python bench.py --no-cpu-baseline > gpurun_out/v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/v.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done; done
