"""One-launch BFS on a road-like graph (side^2 grid, 40 % of the edges removed): long diameter, a wave front a few
thousand vertices wide.  python tools/grid_bfs_bench.py [side]   (GPU box)"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import grid_edges, finalize_edges

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
es, ed, n = grid_edges(side, keep=0.6, seed=3)
gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
ptr, ind = gg["csr"]
nnz = gg["nnz"]
val = torch.ones(nnz, dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), val.data_ptr(),
                          keep=(ptr, ind, val)) == 0
hp = ptr.cpu().numpy()
src = int(np.nonzero(np.diff(hp))[0][len(hp) // 3])
desc = g.Descriptor()
assert desc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
v = g.Vector(n)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, res = g.bfs(v, A, src, desc, fused=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print("n %d nnz %d: info %d, %d levels, reached %d, %.2f ms wall (%.2f us per level), tight %.2f ms"
          % (n, nnz, info, res["levels"], res["reached"], dt, dt * 1e3 / max(1, res["levels"]), res["tight_ms"]))
