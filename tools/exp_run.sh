#!/bin/bash
# GPU box, repo root: tools/exp_run.sh <what> <lib name> [<lib name> ...]
#   what: x (x pass at 2048, 3 components) | p (pencil kernel at C4) | xp (both) | t (thin-slab + binning parity tests) -- letters combine
# every tools/exp_libs/lib_<name>.so (tools/build_variant.sh) is timed through VPS_LIB_PATH; logs under gpurun_out/exp/
what=$1; shift
mkdir -p gpurun_out/exp
for n in "$@"; do
  lib=$PWD/tools/exp_libs/lib_$n.so
  [ "$n" = tree ] && lib=$PWD/large-velocity-power-spectrum_amd/vpower/libvps_hip.so
  echo "== $n" | tee -a gpurun_out/exp/summary.log
  case $what in *x*) VPS_LIB_PATH=$lib timeout -k 10 300 python3 tools/time_xpass.py 2048 3 2>&1 | grep "N=" | tee -a gpurun_out/exp/summary.log || exit 1;; esac
  case $what in *p*) VPS_LIB_PATH=$lib timeout -k 10 400 python3 tools/time_pencil.py 2>&1 | grep "N=" | tee -a gpurun_out/exp/summary.log || exit 1;; esac
  case $what in *t*) VPS_LIB_PATH=$lib timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "thin_slab or binning or fused_z_images or rfft3 or power" 2>&1 | tail -3 | tee -a gpurun_out/exp/summary.log || exit 1;; esac
done
