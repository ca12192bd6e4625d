"""Where config 3 (SSSP on the road-like 4896^2 grid) spends its time: the per-round log of the one-launch
kernel (rounds, frontier sizes, microseconds) summarised.  python tools/sssp_rounds.py [side]   (GPU box)"""
import ctypes as C
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd import _lib
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
assert desc.loadArgs(mxvmode=0, timing=1) == 0
v = g.Vector(n)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info, res = g.sssp(v, G, src, desc)
    torch.cuda.synchronize()
    print("n %d nnz %d: info %d, %d rounds, %.1f ms wall, %.1f ms tight" % (n, nnz, info, res["iterations"], (time.perf_counter() - t0) * 1e3, res["tight_ms"]))


class Iter(C.Structure):
    _fields_ = [("iteration", C.c_int), ("direction", C.c_int), ("value", C.c_double), ("ms", C.c_float), ("reserved", C.c_int)]


cap = 1 << 16
buf = (Iter * cap)()
cnt = C.c_int(0)
_lib.call("grb_descriptor_iter_log", desc._h if hasattr(desc, "_h") else desc.handle, buf, cap, C.byref(cnt))
k = min(cnt.value, cap)
ms = np.array([buf[i].ms for i in range(k)])
fr = np.array([buf[i].value for i in range(k)])
rs = np.array([buf[i].reserved for i in range(k)], dtype=np.int64) & 0xffffffff
t_copy, t_relax = (rs >> 16) * 0.1, (rs & 0xffff) * 0.1       # us, workgroup 0's clock
print("logged rounds %d: total %.1f ms; per round us: median %.1f mean %.1f p90 %.1f max %.1f" % (k, ms.sum(), np.median(ms) * 1e3, ms.mean() * 1e3, np.percentile(ms, 90) * 1e3, ms.max() * 1e3))
print("improved per round: median %.0f mean %.0f max %.0f" % (np.median(fr), fr.mean(), fr.max()))
for lo_, hi2 in ((0, 100), (100, 1000), (1000, 10000), (10000, 100000), (100000, 10**9)):
    m = (fr >= lo_) & (fr < hi2)
    if m.any():
        print("  improved in [%d, %d): %5d rounds, mean %.1f us (copy forward %.1f, relax %.1f, barrier + totals %.1f), share of time %.2f"
              % (lo_, hi2, m.sum(), ms[m].mean() * 1e3, t_copy[m].mean(), t_relax[m].mean(),
                 ms[m].mean() * 1e3 - t_copy[m].mean() - t_relax[m].mean(), ms[m].sum() / ms.sum()))
