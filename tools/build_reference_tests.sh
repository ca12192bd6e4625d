#!/bin/bash
# Drop-in check, part 2 (build container only: needs /root/reference): compiles the reference's OWN
# unit tests and micro-benchmark mains under test/ -- unchanged -- against
# include/graphblas/graphblas.hpp and links them to libgrb_hip.so.  Boost.Test and
# Boost.ProgramOptions are absent from the image; include/boost/ holds the few macros / classes
# these sources touch.  Outputs go to build/refcheck/tests/ (git-ignored; travels to the GPU box,
# where tests/test_gpu_reftests.py runs them).
#
# Left out, and why:
#   gassign gvector gmmio gutil gsparsematrix  stale: they do not compile against the reference's
#                                              own headers either (assign / convert / readMtx /
#                                              countUnique signatures have moved on)
#   gbuild gspgemm grandbfs                    call the CUDA runtime / cuSPARSE / cuda_profiler_api
#                                              directly
#   matrix.cpp dense.cpp mmio.cpp spgemm.cpp util.cpp   the dead sequential (MKL) backend
set -e
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$REF/test" ] || { echo "no reference tree, skipping"; exit 0; }
mkdir -p "$ROOT/build/refcheck/tests"
for t in gvxm greduce gewiseadd gewisemult gtrace gdensevector gsparsevector gdescriptor gbinaryop \
         gspmspv gpush gpull gpushbench gpullbench gspmvbench gspmspvbench; do
  g++ -std=c++11 -O0 -fpermissive -w -x c++ -I"$ROOT/include" -I"$REF" "$REF/test/$t.cu" \
      -L"$ROOT/graphblast_amd" -lgrb_hip -Wl,-rpath,'$ORIGIN/../../../graphblast_amd' \
      -o "$ROOT/build/refcheck/tests/${t}_ref"
  echo "built build/refcheck/tests/${t}_ref"
done
