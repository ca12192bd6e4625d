#!/usr/bin/env python3
"""grb_spmm on the bench graph, k right-hand sides.  python tools/spmm_bench.py [scale] [k]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
src, dst, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(src, dst, n, symmetrize=True)
tptr, tind = gr["csr"]
nnz = gr["nnz"]
tval = torch.rand(nnz, dtype=torch.float32, device=dev)
B = torch.rand((n, k), dtype=torch.float32, device=dev)
C = torch.empty((n, k), dtype=torch.float32, device=dev)
A = g.Matrix(n, n)
assert A.build_csr(tptr.cpu().numpy(), tind.cpu().numpy(), tval.cpu().numpy()) == 0
for _ in range(2):
    assert g.spmm("PlusMultiplies", A, B.data_ptr(), C.data_ptr(), k) == 0
torch.cuda.synchronize()
reps = 5
g.timer_start()
for _ in range(reps):
    g.spmm("PlusMultiplies", A, B.data_ptr(), C.data_ptr(), k)
ms = g.timer_stop() / reps
alg = 8.0 * nnz + 4.0 * (n + 1) + 2 * 4.0 * n * k            # matrix + B once + C once
print("rmat%d k=%d: %.3f ms  %.1f GFLOP/s  %.2f TB/s algorithmic (%.2f TB/s of B gathers)"
      % (scale, k, ms, 2.0 * nnz * k / ms / 1e6, alg / ms / 1e9, 4.0 * k * nnz / ms / 1e9))
