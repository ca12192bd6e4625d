"""Throughput of K queued traversals by the number of lanes (grb_bfs_set_lanes): python tools/bfs_lanes_bench.py [K] [scale]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 22
s_, d_, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]; nnz = gr["nnz"]
tval = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), keep=(tptr, tind, tval)) == 0
ptr = tptr.cpu().numpy()
srcs = [int(np.argmax(np.diff(ptr)))] + random_sources(ptr, 63, seed=0)
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, edgeswitch=0.08)
vs = [g.Vector(n) for _ in range(min(K, 64))]
def run(count):
    ts = [g.bfs_enqueue(vs[i % len(vs)], A, srcs[i % 64], desc)[1] for i in range(count)]
    return [g.bfs_wait(t)[1] for t in ts]
ref = None
for lanes in (1, 2, 4, 8):
    g.bfs_set_lanes(lanes)
    run(2 * lanes)
    g.bfs_host_times(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = run(K)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    sig = [(r["reached"], r["edges_traversed"], r["levels"]) for r in res]
    if ref is None: ref = sig
    assert sig == ref
    print(json.dumps({"lanes": lanes, "K": K, "ms_per_traversal": round(el / K * 1e3, 4), "TEPS": sum(r["edges_traversed"] for r in res) / el,
                      "kernel_clock_ms_mean": round(float(np.mean([r["tight_ms"] for r in res])), 4),
                      "host_enqueue_us": round(g.bfs_host_times()["enqueue_us"] / K, 2)}))
g.bfs_set_lanes(1)
