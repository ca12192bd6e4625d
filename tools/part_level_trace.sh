#!/bin/bash
# durations of the partitioned traversal's launches (one per level) for two sources, next to the one-launch kernel's own
# per-level clock for the same sources: tools/part_level_trace.sh [outdir]   [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06/part_trace}
mkdir -p $out
cat > /tmp/part_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
from graphblast_amd.dist import Partition1D
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 3, seed=0)
part = Partition1D(n, tptr.long(), tind.long(), 0, 1, dev, edgeswitch=0.08, levels_per_launch=1)
for s in srcs: part.bfs(s)
torch.cuda.synchronize()
print("MARK")
for s in srcs[:2]:
    r = part.bfs(s, want_trace=True)
    print("part src", s, "levels", r["levels"], "launches", r["launches"], "device_ms", round(r["device_ms"], 4))
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
v = g.Vector(n)
for s in srcs[:2]:
    g.bfs(v, A, s, desc, fused=True)
    r = g.bfs(v, A, s, desc, fused=True, profile=1)[1]
    print("one-launch src", s, "event ms", round(r["tight_ms"], 4), " | ".join("%s nf=%d %.1fus" % (L["direction"][:2], L["frontier"], L["ms"] * 1e3) for L in r["per_level"]))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o b -- python /tmp/part_one.py > $out/stdout.log 2>&1
f=$(find $out/kt -name "b_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bfs_part_level_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
# the last two traversals' launches (those after MARK): print the tail
print("last 20 bfs_part_level_kernel launches: duration us (gap to the next):")
for i in range(max(0, len(d) - 20), len(d)):
    print("  %7.1f  (%s)" % (d[i], "%.1f" % gaps[i] if i < len(gaps) else "-"))
PY
grep -E "part src|one-launch src" $out/stdout.log
rm -rf $out/kt
