#!/usr/bin/env python3
"""Triangle count on the bench's stand-in graph as it is labelled and relabelled by descending degree (row i of L =
the neighbours of HIGHER degree: no long lists).  The count is a graph invariant; the time is not.
python tools/tc_relabel_probe.py [scale] [edge factor]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 28
dev = torch.device("cuda", 0)
s, d, n = rmat_edges(scale, ef, seed=6, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]
del s, d
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (ptr[1:] - ptr[:-1]).to(torch.int64))
cols = ind.to(torch.int64)
deg = (ptr[1:] - ptr[:-1]).to(torch.int64)


def lower(rank, name):
    r, c = rank[rows], rank[cols]
    keep = c <= r
    key = torch.sort(r[keep] * n + c[keep]).values
    r, c = (key // n), (key % n)
    lp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    lp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    lp_h, li_h = lp.to(torch.int32).cpu().numpy(), c.to(torch.int32).cpu().numpy()
    dl = np.diff(lp_h)
    L = g.Matrix(n, n, np.int32)
    assert L.build_csr(lp_h, li_h, np.ones(li_h.size, dtype=np.int32)) == 0
    B = g.Matrix(n, n, np.int32)
    ms = []
    for i in range(4):
        dd = g.Descriptor()
        dd.loadArgs()
        info, ntri, res = g.tc(L, B, dd)
        assert info == 0
        ms.append(res["tight_ms"])
    print("%-28s nnz(L) %d longest list %d sum dL^2 %.3e: triangles %d, ms %s" % (name, li_h.size, dl.max(), float((dl.astype(np.float64) ** 2).sum()), ntri, ["%.1f" % m for m in ms]), flush=True)
    return ntri


ids = torch.arange(n, device=dev, dtype=torch.int64)
t0 = lower(ids, "as labelled")
order = torch.sort(deg * n + (n - 1 - ids), descending=True).values  # by degree descending, ties by id ascending
byrank = (n - 1 - (order % n))
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[byrank] = ids
t1 = lower(rank, "by descending degree")
rank2 = torch.empty(n, dtype=torch.int64, device=dev)
rank2[byrank] = n - 1 - ids
t2 = lower(rank2, "by ascending degree")
assert t0 == t1 == t2
