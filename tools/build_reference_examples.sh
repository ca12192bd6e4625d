#!/bin/bash
# Drop-in check (build container only: needs /root/reference): compiles the reference's OWN
# example/g{bfs,sssp,pr,cc,tc,mis,gc,gc_cusparse,lgc,diameter}.cu (all ten) -- unchanged, together with its graphblas/algorithm/*.hpp and
# test/test.hpp -- against include/graphblas/graphblas.hpp and links them to libgrb_hip.so.
# Outputs go to build/refcheck/ (git-ignored; travels to the GPU box, where
# tests/test_gpu_dropin.py runs them on the reference's data/small graphs).
set -e
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/example" ] || { echo "no reference tree, skipping"; exit 0; }
mkdir -p "$ROOT/build/refcheck"
for ex in gbfs gsssp gpr gcc gtc gmis ggc ggc_cusparse glgc gdiameter; do
  # -O0: the reference's CC helpers are non-void functions without a return statement
  # (graphblas/algorithm/cc.hpp:138-152, test_cc.hpp:14-95) -- undefined behaviour that an
  # optimising g++ turns into a fall-through crash
  g++ -std=c++11 -O0 -fpermissive -w -x c++ -I"$ROOT/include" -I"$REF" "$REF/example/$ex.cu" \
      -L"$ROOT/graphblast_amd" -lgrb_hip -Wl,-rpath,'$ORIGIN/../../graphblast_amd' \
      -o "$ROOT/build/refcheck/${ex}_ref"
  echo "built build/refcheck/${ex}_ref"
done
