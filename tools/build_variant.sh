#!/bin/bash
# tools/build_variant.sh <name> <fft source (abs path or csrc-relative)> [extra hipcc flags]
# -> tools/exp_libs/lib_<name>.so: the in-tree library with fft.o replaced by a variant build (tuning experiments;
#    bench.py / tests pick it up through VPS_LIB_PATH).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/large-velocity-power-spectrum_amd/csrc
name=$1; src=$2; shift 2
mkdir -p $root/tools/exp_libs /tmp/vps_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$csrc -I$root/include "$@" -c -x hip $src -o /tmp/vps_variants/fft_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/exp_libs/lib_$name.so $csrc/build/api.o $csrc/build/deposit.o \
  $csrc/build/hist.o $csrc/build/nn.o $csrc/build/preprocess.o /tmp/vps_variants/fft_$name.o
