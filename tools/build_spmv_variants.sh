#!/bin/bash
# Builds libgrb_hip_<name>.so variants of the hub-packed SpMV for A/B timing (tools/spmv_ab.py
# with GRB_HIP_LIB=...).  usage: tools/build_spmv_variants.sh name "-DGRB_HUB_TILE=1024 ..." [...]
set -e
cd "$(dirname "$0")/../graphblast_amd/csrc"
make -s
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. $flags -c spmv.hip -o /tmp/spmv_$name.o
  objs=$(ls *.o | grep -v '^spmv.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libgrb_hip_$name.so $objs /tmp/spmv_$name.o
  echo built $name
done
