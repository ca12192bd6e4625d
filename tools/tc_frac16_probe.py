import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from graphblast_amd.graphgen import rmat_edges, finalize_edges
dev = torch.device("cuda", 0)
s, d, n = rmat_edges(22, 28, seed=6, device=dev)
gr = finalize_edges(s, d, n, symmetrize=True)
ptr, ind = gr["csr"]
deg = (ptr[1:] - ptr[:-1]).to(torch.int64)
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg)
cols = ind.to(torch.int64)
ids = torch.arange(n, device=dev, dtype=torch.int64)
key = deg * n + (n - 1 - ids)
number = torch.empty(n, dtype=torch.int64, device=dev)
number[torch.sort(key, descending=True).indices] = ids
r2, c2 = number[rows], number[cols]
up = c2 < r2
lo, hi = r2[up], c2[up]
dL = torch.bincount(lo, minlength=n)
for bits in (14, 15, 16, 17, 18):
    small = (hi < (1 << bits))
    c16 = torch.bincount(lo[small], minlength=n)
    piv_is_lo = dL[hi] <= dL[lo]
    par = torch.where(piv_is_lo, hi, lo)
    tot = float(dL[par].sum()); s16 = float(c16[par].sum())
    print("numbers below 2^%d: %.4f of the list entries, %.4f of the streamed elements" % (bits, float(small.float().mean()), s16 / tot))
# elements of a partner's list that are not below the pivot's number cannot be in the pivot's list (it holds numbers below
# the pivot): how much of the stream belongs to pivots numbered below 2^b, and how much of THAT lies below 2^b
piv_is_lo = dL[hi] <= dL[lo]
piv = torch.where(piv_is_lo, lo, hi)
par = torch.where(piv_is_lo, hi, lo)
tot = float(dL[par].sum())
for bits in (14, 15, 16, 17, 18):
    small = (hi < (1 << bits))
    c16 = torch.bincount(lo[small], minlength=n)
    sel = piv < (1 << bits)
    print("pivots numbered below 2^%d: %.4f of the edges, %.4f of the streamed elements (%.4f of the total lies in their partners' parts below 2^%d)"
          % (bits, float(sel.float().mean()), float(dL[par][sel].sum()) / tot, float(c16[par][sel].sum()) / tot, bits))
# a partner's entries that are not below the PIVOT's number cannot hit either: with sorted lists and a cut per (pivot, partner)
# only the entries below the pivot's number would be streamed -- how much of the stream is that?
keys = torch.sort(lo * n + hi).values                       # list entries, owner-major, ascending inside a list
start = torch.zeros(n + 1, dtype=torch.int64, device=dev)
start[1:] = torch.cumsum(dL, 0)
useful = torch.searchsorted(keys, par * n + piv) - start[par]
print("entries of the streamed lists that are below their pivot's number: %.4f of the streamed elements" % (float(useful.sum()) / tot))
for bits in (16,):
    sel = piv < (1 << bits)
    print("   of pivots numbered below 2^16: %.4f of their stream" % (float(useful[sel].sum()) / float(dL[par][sel].sum())))
