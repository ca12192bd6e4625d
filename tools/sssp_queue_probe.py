"""The queue-based synchronous-rounds SSSP of csrc/sssp_part_run.hip (one rank, many rounds per launch) on the
config-3 stand-in (4896^2 thinned grid), next to grb_sssp's default (near / far) and its synchronous rounds."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import grid_edges, finalize_edges
from graphblast_amd.dist import Partition1D
dev = torch.device("cuda", 0)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4896
es, ed, n = grid_edges(side, keep=0.6, seed=3)
gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
gptr, gind = gg["csr"]; nnz = gg["nnz"]
grow = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (gptr[1:] - gptr[:-1]).long())
lo, hi_ = torch.minimum(grow, gind.long()), torch.maximum(grow, gind.long())
gw = ((((lo * 1000003) ^ hi_) * 2654435761 >> 7) % 64 + 1).to(torch.float32)
del grow, lo, hi_
hp = gptr.cpu().numpy()
src = int(np.nonzero(np.diff(hp))[0][len(hp) // 3])
G = g.Matrix(n, n)
assert G.build_device_csr(gptr.data_ptr(), gind.data_ptr(), gw.data_ptr(), nnz, gptr.data_ptr(), gind.data_ptr(), gw.data_ptr(), keep=(gptr, gind, gw)) == 0
desc = g.Descriptor(); desc.loadArgs(mxvmode=0, timing=0)
v = g.Vector(n)
g.sssp(v, G, src, desc)
torch.cuda.synchronize(); t0 = time.perf_counter()
info, res = g.sssp(v, G, src, desc)
torch.cuda.synchronize(); t_def = time.perf_counter() - t0
want = v.extractTuples()[1].copy()
w = g.sssp_last_work()
out = {"n": n, "nnz": nnz, "work": [int(x) for x in w], "default_ms": round(t_def * 1e3, 2), "rounds": res["iterations"], "passes": g.sssp_last_order()}
print(json.dumps(out))
if len(sys.argv) > 2 and sys.argv[2] == "part":
    out = {}
    part = Partition1D(n, gptr.long(), gind.long(), 0, 1, dev)
    for rpl in (1 << 20,):
        d, inf = part.sssp(gw, src, rounds_per_launch=rpl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d, inf = part.sssp(gw, src, rounds_per_launch=rpl)
        torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        out["queue_rounds_per_launch_%d" % rpl] = {"ms": round(t1 * 1e3, 2), "device_ms": round(inf["device_ms"], 2), "iterations": inf["iterations"],
                                                  "launches": inf["launches"], "same_distances": bool(np.array_equal(d.cpu().numpy(), want))}
    print(json.dumps(out))
