"""Config 3's stand-in (4896^2 thinned grid, weights 1..64): the synchronous rounds against the near / far order.
python tools/sssp_nearfar_bench.py [side]   (GPU box; GRB_SSSP_DELTA=<factor> changes the bucket width)"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import grid_edges, finalize_edges

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4896
dev = torch.device("cuda", 0)
es, ed, n = grid_edges(side, keep=0.6, seed=3)
gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
gptr, gind = gg["csr"]
nnz = gg["nnz"]
grow = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (gptr[1:] - gptr[:-1]).long())
lo, hi_ = torch.minimum(grow, gind.long()), torch.maximum(grow, gind.long())
gw = ((((lo * 1000003) ^ hi_) * 2654435761 >> 7) % 64 + 1).to(torch.float32)
G = g.Matrix(n, n)
assert G.build_device_csr(gptr.data_ptr(), gind.data_ptr(), gw.data_ptr(), nnz, gptr.data_ptr(), gind.data_ptr(),
                          gw.data_ptr(), keep=(gptr, gind, gw)) == 0
hp = gptr.cpu().numpy()
src = int(np.nonzero(np.diff(hp))[0][len(hp) // 3])
desc = g.Descriptor()
assert desc.loadArgs(mxvmode=0, timing=0) == 0
v = g.Vector(n)
out = {}
for mode in (-1, 0) if len(sys.argv) < 3 else (-1,):
    g.sssp_set_nearfar(mode)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info, res = g.sssp(v, G, src, desc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    out[mode] = v.extractTuples()[1].copy()
    order = g.sssp_last_order()
    print("mode %2d: info %d, %d rounds reported, %.1f ms wall, %.1f ms tight, %s"
          % (mode, info, res["iterations"], dt, res["tight_ms"],
             "near / far, %d passes, %.1f us per pass" % (order, res["tight_ms"] * 1e3 / order) if order else "synchronous rounds"))
if len(out) == 2:
    print("distances identical:", bool(np.array_equal(out[-1], out[0])))
