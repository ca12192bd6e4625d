#!/bin/bash
# per-kernel durations of a matrix's FIRST traversal (the once-per-matrix preparation): tools/bfs_prep_trace.sh [outdir]   [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06/prep}
mkdir -p $out
cat > /tmp/prep_one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
v = g.Vector(n)
torch.cuda.synchronize(); t0 = time.perf_counter()
g.bfs(v, A, 5, desc, fused=True)
torch.cuda.synchronize(); print("first traversal ms", (time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
g.bfs(v, A, 5, desc, fused=True)
torch.cuda.synchronize(); print("second traversal ms", (time.perf_counter() - t0) * 1e3)
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o b -- python /tmp/prep_one.py > $out/stdout.log 2>&1
f=$(find $out/kt -name "b_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    if r and any(k in r[0] for k in ("bfs_", "oc_", "ptr_differ", "scan_", "empty_rows", "degree_class")):
        name = r[0].replace("void grb::", "").split("(")[0]
        print("   %-60s calls %4s total %8.3f ms" % (name[:60], r[1], int(r[2]) / 1e6))
PY
grep "traversal ms" $out/stdout.log
rm -rf $out/kt
