"""Throughput of K queued traversals by the number of traversals per launch (grb_bfs_set_coschedule):
python tools/bfs_co_bench.py [K] [scale] [widths, e.g. 1,2,3,4]
Every width's labels are compared with width 1's (all K vectors), which the test suite compares with the oracle."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 22
widths = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4]
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
nv = min(K, 64)
vs = [g.Vector(n) for _ in range(nv)]
def run(count):
    out = []
    for i0 in range(0, count, nv):                       # a vector is queued again only after its ticket has been waited for
        ts = [g.bfs_enqueue(vs[i % nv], A, srcs[i % 64], desc)[1] for i in range(i0, min(count, i0 + nv))]
        out += [g.bfs_wait(t)[1] for t in ts]
    return out
ref = ref_lab = None
for w in widths:
    g.bfs_set_coschedule(w)
    run(2 * max(w, 1))
    best = None
    for rep in range(3):
        g.bfs_host_times(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = run(K)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        best = el if best is None or el < best else best
    sig = [(r["reached"], r["edges_traversed"], r["levels"]) for r in res]
    lab = [v.extractTuples()[1] for v in vs]
    if ref is None: ref, ref_lab = sig, lab
    assert sig == ref, "result blocks differ from one traversal per launch"
    assert all(np.array_equal(a, b) for a, b in zip(lab, ref_lab)), "labels differ from one traversal per launch"
    print(json.dumps({"per_launch": w, "K": K, "ms_per_traversal": round(best / K * 1e3, 4), "TEPS": sum(r["edges_traversed"] for r in res) / best,
                      "kernel_clock_ms_mean": round(float(np.mean([r["tight_ms"] for r in res])), 4),
                      "host_enqueue_us": round(g.bfs_host_times()["enqueue_us"] / K, 2), "labels": "equal to per_launch 1 (all %d vectors)" % nv}), flush=True)
g.bfs_set_coschedule(1)
