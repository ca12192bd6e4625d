#!/usr/bin/env python3
"""Would sparse x dense mxm gain from column blocks sized to the memory-side cache?  
B at k = 64 is 1 GB and every nonzero gathers one of its 256-byte rows; with the columns cut into nb blocks a block's
rows of B (1 GB / nb) can stay in the 256 MB cache while that block's entries are multiplied.  No library change is
needed to find out: the matrix is sliced by column range here (the slices keep their global column ids), grb_spmm
runs on each slice into its OWN output, and the nb calls are timed together -- what an accumulating form of the
product would cost, give or take the read of C.  The slices' outputs are summed once and compared with the whole
product.   python tools/spmm_colblock_probe.py [scale] [k] [nb ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nbs = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
dev = torch.device("cuda", 0)
src, dst, n = rmat_edges(scale, 16, seed=1, device=dev)
gr = finalize_edges(src, dst, n, symmetrize=True)
tptr, tind = gr["csr"]
nnz = gr["nnz"]
tval = torch.rand(nnz, dtype=torch.float32, device=dev)
B = torch.rand((n, k), dtype=torch.float32, device=dev)
C = torch.empty((n, k), dtype=torch.float32, device=dev)


def build(ptr, ind, val):
    A = g.Matrix(n, n)
    m = int(ind.numel())
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), m, keep=(ptr, ind, val)) == 0
    return A


def timed(calls, reps=5):
    for _ in range(2):
        for A_, C_ in calls:
            assert g.spmm("PlusMultiplies", A_, B.data_ptr(), C_.data_ptr(), k) == 0
    torch.cuda.synchronize()
    g.timer_start()
    for _ in range(reps):
        for A_, C_ in calls:
            g.spmm("PlusMultiplies", A_, B.data_ptr(), C_.data_ptr(), k)
    return g.timer_stop() / reps


A = build(tptr.to(torch.int32).contiguous(), tind.to(torch.int32).contiguous(), tval)
ms0 = timed([(A, C)])
ref = C.clone()
print("rmat%d k=%d whole matrix: %.3f ms  (%.2f TB/s of B gathers)" % (scale, k, ms0, 4.0 * k * nnz / ms0 / 1e9))
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (tptr[1:] - tptr[:-1]).to(torch.int64))
for nb in nbs:
    cuts = [int(round(n * b / nb)) for b in range(nb + 1)]
    calls, outs, sizes = [], [], []
    for b in range(nb):
        keep = (tind >= cuts[b]) & (tind < cuts[b + 1])
        cnt = torch.bincount(rows[keep], minlength=n)
        ptr_b = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(cnt, 0, out=ptr_b[1:])
        ind_b = tind[keep].to(torch.int32).contiguous()
        val_b = tval[keep].contiguous()
        if ind_b.numel() == 0:
            continue
        Cb = torch.zeros((n, k), dtype=torch.float32, device=dev)
        calls.append((build(ptr_b.to(torch.int32).contiguous(), ind_b, val_b), Cb))
        outs.append(Cb)
        sizes.append(int(ind_b.numel()))
    ms = timed(calls)
    tot = torch.zeros_like(ref)
    for Cb in outs:
        tot += Cb
    err = float(((tot - ref).abs() / ref.abs().clamp_min(1e-20)).max().item())
    print("  %d column blocks (%s entries): %.3f ms for the %d products together = %.2f x the whole product; sum of the "
          "outputs against it: max rel diff %.1e" % (nb, "/".join("%.0fM" % (s / 1e6) for s in sizes), ms, len(calls), ms / ms0, err))
    del calls, outs
