#!/bin/bash
# Builds build/libgrb_hip_<name>.so variants of ONE source file of the library for same-box A/B timing
# (GRB_HIP_LIB=build/libgrb_hip_<name>.so python tools/...).  The variants are git-ignored; delete them when done --
# everything under build/ ships to the GPU box.
# usage: tools/build_variants.sh <file.hip> name "-DFLAG=1 ..." [name flags ...]
set -e
cd "$(dirname "$0")/../graphblast_amd/csrc"
make -s
src=$1; shift
stem=${src%.hip}
mkdir -p ../../build
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. $flags -c $src -o /tmp/${stem}_$name.o
  objs=$(ls *.o | grep -v "^$stem.o\$" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libgrb_hip_$name.so $objs /tmp/${stem}_$name.o -ldl
  echo built $name
done
