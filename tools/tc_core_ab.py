"""The dense core of the triangle count on the config-5 stand-in (RMAT ef 28, symmetrised, lower triangle): AND + popcount
per mask entry against v_mfma_i32_16x16x64_i8 on the same bit rows (grb_tc_dense_core), for a range of core sizes.
python tools/tc_core_ab.py [scale] [K,K,...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2048, 4096, 8192, 16384, 32768]
s_, d_, n = rmat_edges(scale, 28, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
tptr, tind = gr["csr"]
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), (tptr[1:] - tptr[:-1]).to(torch.int64))
keep = tind < rows
lj = tind[keep].contiguous()
cnt = torch.bincount(rows[keep].to(torch.int64), minlength=n)
lptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
lptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
lval = torch.ones(lj.numel(), dtype=torch.int32, device=dev)
del rows, keep, tind, tptr
L = g.Matrix(n, n, np.int32)
assert L.build_device_csr(lptr.data_ptr(), lj.data_ptr(), lval.data_ptr(), lj.numel(), keep=(lptr, lj, lval)) == 0
print(json.dumps({"graph": "rmat%d ef 28 symmetrised, lower triangle" % scale, "n": n, "nnz_L": int(lj.numel())}), flush=True)
for K in ks:
    row = {"k_want": K}
    ref = None
    for name, method, dense_from in (("popcount", 0, 0), ("mfma", 1, 0), ("mfma_from_2048", 2, 2048), ("mfma_from_4096", 2, 4096), ("mfma_from_8192", 2, 8192)):
        best = None
        for rep in range(3):
            info, r = g.tc_dense_core(L, K, method, dense_from)
            assert info == 0, info
            best = r if best is None or r["product_ms"] < best["product_ms"] else best
        if ref is None:
            ref = best
            row.update(core_rows=best["core_rows"], min_row_length=best["min_row_length"], core_entries=best["core_entries"],
                       count=best["count"], tiles=best["tiles"], tiles_by_density_tenths=best["tiles_by_density"],
                       build_ms=round(best["build_ms"], 3))
        assert (best["count"], best["checksum"], best["core_entries"]) == (ref["count"], ref["checksum"], ref["core_entries"]), name
        row[name] = {"product_ms": round(best["product_ms"], 4), "tiles_mfma": best["tiles_mfma"]}
    print(json.dumps(row), flush=True)
