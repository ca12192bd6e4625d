#!/bin/bash
# band size against waves per CU in the band SpMV kernel: the variants of tools/build_variants.sh spmv.hip cbA ... (see
# docs/experiments.md R6.6), valued and pattern, RMAT-22   [GPU box]
cd $GRAFT_REPO_ROOT
for v in "" $(ls build/libgrb_hip_cb*.so 2>/dev/null); do
  GRB_HIP_LIB=$v timeout 300 python tools/spmv_cband_quick.py 2>&1 | tail -1
  ISO=1 GRB_HIP_LIB=$v timeout 300 python tools/spmv_cband_quick.py 2>&1 | tail -1
done
