"""Where the one-launch traversal's time goes over the 64 bench sources, by kind of level (the kernel's own per-level
clock): python tools/bfs_level_mix.py [scale]"""
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
kinds = {}
tot = 0.0
for s in srcs:
    best = None
    for _ in range(3):
        info, r = g.bfs(v, A, s, desc, fused=True, profile=1)
        t = sum(L["ms"] for L in r["per_level"])
        if best is None or t < best[0]: best = (t, r)
    t, r = best
    lv = r["per_level"]
    for i, L in enumerate(lv):
        pull = L["direction"] == "pull"
        if pull:
            k = "pull dense" if i == 0 or lv[i - 1]["direction"] == "push" else "pull sparse"
        elif L["frontier"] == 1: k = "push source/one vertex"
        elif L["frontier_edges"] >= 262144: k = "push heavy (owner-computes)"
        elif L["frontier"] <= 4096 and L["frontier_edges"] <= 65536: k = "push tiny"
        else: k = "push other"
        a = kinds.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += L["ms"] * 1e3
    tot += t * 1e3
print("per traversal (mean over %d sources): %.1f us in levels" % (len(srcs), tot / len(srcs)))
for k, (c, us) in sorted(kinds.items(), key=lambda x: -x[1][1]):
    print("  %-30s %5.2f levels  %6.1f us each  %6.1f us per traversal" % (k, c / len(srcs), us / c, us / len(srcs)))
