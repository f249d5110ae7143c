#!/bin/bash
# GPU box, repo root: tools/exp_run.sh <what> <lib name> [<lib name> ...]
#   what: x (x pass at 2048, 3 components) | p (pencil kernel at C4) | y (z / y / x passes of one 2048^3 field) | t (thin-slab + binning parity tests) -- letters combine
# every tools/exp_libs/lib_<name>.so (tools/build_variant.sh; "tree" = the in-tree library) is timed through VPS_LIB_PATH;
# full output under gpurun_out/exp/<name>_<what>.log, the timing lines also in gpurun_out/exp/summary.log
what=$1; shift
ulimit -c 0          # (a faulting kernel variant must not fill the box's disk with a core file)
mkdir -p gpurun_out/exp
export TMPDIR=${TMPDIR:-/tmp}
for n in "$@"; do
  lib=$PWD/tools/exp_libs/lib_$n.so
  [ "$n" = tree ] && lib=$PWD/large-velocity-power-spectrum_amd/vpower/libvps_hip.so
  echo "== $n" | tee -a gpurun_out/exp/summary.log
  case $what in *x*) VPS_LIB_PATH=$lib timeout -k 10 300 python3 tools/time_xpass.py 2048 3 > gpurun_out/exp/${n}_x.log 2>&1; rc=$?
                     grep "N=" gpurun_out/exp/${n}_x.log | tee -a gpurun_out/exp/summary.log; [ $rc -ne 0 ] && { tail -5 gpurun_out/exp/${n}_x.log; df -h /tmp | tail -1; exit 1; };; esac
  case $what in *p*) VPS_LIB_PATH=$lib timeout -k 10 400 python3 tools/time_pencil.py > gpurun_out/exp/${n}_p.log 2>&1; rc=$?
                     grep "N=" gpurun_out/exp/${n}_p.log | tee -a gpurun_out/exp/summary.log; [ $rc -ne 0 ] && { tail -5 gpurun_out/exp/${n}_p.log; df -h /tmp | tail -1; exit 1; };; esac
  case $what in *y*) VPS_LIB_PATH=$lib timeout -k 10 300 python3 tools/time_fft_passes.py 2048 2048 > gpurun_out/exp/${n}_y.log 2>&1; rc=$?
                     grep "N=" gpurun_out/exp/${n}_y.log | tee -a gpurun_out/exp/summary.log; [ $rc -ne 0 ] && { tail -5 gpurun_out/exp/${n}_y.log; exit 1; };; esac
  case $what in *t*) VPS_LIB_PATH=$lib timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "thin_slab or binning or fused_z_images or rfft3 or power" > gpurun_out/exp/${n}_t.log 2>&1; rc=$?
                     tail -3 gpurun_out/exp/${n}_t.log | tee -a gpurun_out/exp/summary.log; [ $rc -ne 0 ] && { tail -30 gpurun_out/exp/${n}_t.log; exit 1; };; esac
done
if grep -l "Memory access fault" gpurun_out/exp/*.log 2>/dev/null; then echo "GPU FAULT in the logs above"; exit 3; fi
exit 0
