"""Experiment: ONE traversal on m workgroups of 512 threads per CU (the instance built for six waves per SIMD) against the
1024-thread kernel.  GRB_HIP_LIB=build/libgrb_hip_lean512.so GRB_BFS_SINGLE_WIDE=3 python tools/bfs_single_wide_probe.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
s_, d_, n = rmat_edges(22, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
v = g.Vector(n)
ref = {}
for s in srcs:
    assert g.bfs(v, A, s, desc, fused=True)[0] == 0
    ref[s] = v.extractTuples()[1]
g.bfs_set_coschedule(2)          # tickets are deferred; a wait on a lone ticket launches it (wide when GRB_BFS_SINGLE_WIDE is set)
for s in srcs[:4]:
    g.bfs_wait(g.bfs_enqueue(v, A, s, desc)[1])
clock = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in srcs:
    info, r = g.bfs_wait(g.bfs_enqueue(v, A, s, desc)[1])
    assert info == 0
    clock.append(r["tight_ms"])
torch.cuda.synchronize(); el = time.perf_counter() - t0
bad = 0
for s in srcs[:8]:
    g.bfs_wait(g.bfs_enqueue(v, A, s, desc)[1])
    bad += 0 if np.array_equal(v.extractTuples()[1], ref[s]) else 1
print(json.dumps({"lib": os.environ.get("GRB_HIP_LIB", "default"), "single_wide": os.environ.get("GRB_BFS_SINGLE_WIDE"),
                  "kernel_clock_ms_mean": round(float(np.mean(clock)), 4), "wall_ms_per_traversal": round(el / len(srcs) * 1e3, 4), "label_mismatches": bad}))
