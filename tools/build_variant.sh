#!/bin/bash
# tools/build_variant.sh <name> <unit> <source> [extra hipcc flags]
#   unit: which object of the library to replace (fft | fft0..fft3 | deposit | nn | hist | api | comm | preprocess)
#         fft = the whole of fft.hip as ONE unit (minutes); fftK = only the part with that family of line lengths
# -> tools/exp_libs/lib_<name>.so: the in-tree library with one object rebuilt from <source> with extra flags
#    (tuning experiments; bench.py / tests pick it up through VPS_LIB_PATH).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/large-velocity-power-spectrum_amd/csrc
name=$1; unit=$2; src=$3; shift 3
mkdir -p $root/tools/exp_libs /tmp/vps_variants
part=""
case $unit in fft[0-3]) part="-DVPS_FFT_PART=${unit#fft}";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$csrc -I$root/include $part "$@" -c -x hip $src -o /tmp/vps_variants/${unit}_$name.o
objs=""
for u in api comm deposit fft hist nn preprocess; do
  if [ $u = $unit ]; then objs="$objs /tmp/vps_variants/${unit}_$name.o";      # (a variant fft.hip is ONE unit: VPS_FFT_PART = -1)
  elif [ $u = fft ]; then
    for k in 0 1 2 3; do
      if [ fft$k = $unit ]; then objs="$objs /tmp/vps_variants/${unit}_$name.o"; else objs="$objs $csrc/build/fft_p$k.o"; fi
    done
  else objs="$objs $csrc/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/exp_libs/lib_$name.so $objs -ldl
