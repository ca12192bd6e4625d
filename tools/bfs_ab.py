"""Same-box A/B of the one-launch BFS: GRB_HIP_LIB=build/libgrb_hip_<variant>.so python tools/bfs_ab.py [scale]
Prints the mean / median kernel time over the 64 bench sources (min of 3 runs each, the kernel's own clock), the level
records of the hub source and of the slowest source, and a checksum of all 64 label vectors (equal across variants
or the variant is wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=float(os.environ.get("EDGESWITCH", "0.08")))
v = g.Vector(n)
for s in srcs[:4]: g.bfs(v, A, s, desc, fused=True)
wt = (torch.arange(n, device=dev, dtype=torch.int64) % 1000003) + 1
rows, check, ev = [], 0, []
for s in srcs:
    t = min(g.bfs(v, A, s, desc, fused=True)[1]["tight_ms"] for _ in range(3))
    lab = torch.from_numpy(v.extractTuples()[1]).to(dev).to(torch.int64)
    check = (check * 31 + int((lab * wt).sum().item())) % (1 << 61)
    info, r = g.bfs(v, A, s, desc, fused=True, profile=1)
    ev.append(min(r["tight_ms"], g.bfs(v, A, s, desc, fused=True, profile=1)[1]["tight_ms"]))   # HIP events around the launch
    rows.append((t, s, r["per_level"]))
ts = np.array([r[0] for r in rows])
print("LIB %s  mean %.4f median %.4f min %.4f max %.4f ms  by HIP events mean %.4f  checksum %d" % (os.path.basename(os.environ.get("GRB_HIP_LIB", "default")), ts.mean(), np.median(ts), ts.min(), ts.max(), float(np.mean(ev)), check))
def show(t, s, lv):
    print("  src %8d tight %.4f  " % (s, t) + " | ".join("%s nf=%d %.1fus" % (L["direction"][:2], L["frontier"], L["ms"] * 1e3) for L in lv))
show(*rows[0])
show(*max(rows, key=lambda x: x[0]))
