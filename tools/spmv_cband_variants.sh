#!/bin/bash
# Builds libgrb_hip_<name>.so variants of the column-sorted SpMV for A/B timing (tools/spmv_cband_quick.py with
# GRB_HIP_LIB=build/libgrb_hip_<name>.so).  usage: tools/spmv_cband_variants.sh name "-DGRB_CB_UNROLL=4 ..." [...]
set -e
cd "$(dirname "$0")/../graphblast_amd/csrc"
make -s
mkdir -p ../../build
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. $flags -c spmv.hip -o /tmp/spmv_$name.o
  objs=$(ls *.o | grep -v '^spmv.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libgrb_hip_$name.so $objs /tmp/spmv_$name.o -ldl
  echo built $name
done
