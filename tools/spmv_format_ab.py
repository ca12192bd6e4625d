"""SpMV on RMAT-<scale>: the CSR kernel against the column-sorted band format (csrc/spmv_cband.hpp), with random
values and with an iso (all-ones) matrix, PlusMultiplies f32 and MinimumPlus f32; results checked against a float64
torch matvec.  Prints one JSON line.   python tools/spmv_format_ab.py [scale] [edge_factor]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
import graphblast_amd as g                                      # noqa: E402
g.spmv_set_reuse_threshold(0)   # measurement scripts: the band format at the first product (the library waits for 48 by default)
from graphblast_amd.graphgen import rmat_edges, finalize_edges  # noqa: E402

dev = torch.device("cuda", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s_, d_, n = rmat_edges(scale, ef, seed=1, device=dev)
gr = finalize_edges(s_, d_, n, symmetrize=True)
del s_, d_
tptr, tind = gr["csr"]
nnz = gr["nnz"]
x = torch.rand(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
rows = torch.repeat_interleave(torch.arange(n, device=dev), (tptr[1:] - tptr[:-1]).long())
out = {"graph": "rmat%d_ef%d_sym" % (scale, ef), "n": n, "nnz": nnz}
for vname, tval in (("random", torch.rand(nnz, dtype=torch.float32, device=dev)),
                    ("iso", torch.ones(nnz, dtype=torch.float32, device=dev))):
    ref = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, tval.double() * x.double()[tind.long()])
    for fmt, fname in ((0, "csr"), (1, "cband")):
        g.spmv_set_format(fmt)
        A = g.Matrix(n, n)
        assert A.build_device_csr(tptr.data_ptr(), tind.data_ptr(), tval.data_ptr(), nnz, keep=(tptr, tind, tval)) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr()) == 0
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) * 1e3
        for _ in range(3):
            g.k_spmv(A, 0, "PlusMultiplies", x.data_ptr(), None, 0, 0, y.data_ptr())
        torch.cuda.synchronize()
        err = ((y.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        rec = {"first_call_ms": round(prep_ms, 1), "max_rel_err": err}
        for op in ("PlusMultiplies", "MinimumPlus"):
            g.timer_start()
            for _ in range(20):
                g.k_spmv(A, 0, op, x.data_ptr(), None, 0, 0, y.data_ptr())
            ms = g.timer_stop() / 20
            rec[op] = {"ms": round(ms, 4), "algorithmic_GBs_csr_bytes": round(g.k_spmv_bytes(A, 0) / ms / 1e6, 1)}
        info = g.spmv_format_info(A, 0)
        if info["in_use"]:
            rec["format"] = info
            rec["PlusMultiplies"]["GBs_own_bytes"] = round(info["bytes_per_launch"] / rec["PlusMultiplies"]["ms"] / 1e6, 1)
        out["%s_%s" % (vname, fname)] = rec
        del A
g.spmv_set_format(1)
print(json.dumps(out))
