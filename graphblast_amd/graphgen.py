"""Seeded synthetic inputs for tests and bench; no files, no network.

rmat_edges      Graph500-style Kronecker generator, (a,b,c,d) = (0.57,0.19,0.19,0.05),
                vertex labels permuted.
finalize_edges  the reference loader's post-processing (graphblas/util.hpp:263-329):
                optional symmetrise, sort by (row, col), drop self loops and duplicates;
                returns CSR (and CSC).  Runs in numpy on the host, or in torch on the GPU
                for the full-size bench graphs (input plumbing, outside every timed region).
"""
import numpy as np


def rmat_edges(scale, edge_factor=16, seed=1, a=0.57, b=0.19, c=0.19, device=None):
    n = 1 << scale
    m = n * edge_factor
    if device is None:
        rng = np.random.default_rng(seed)
        src = np.zeros(m, dtype=np.int64)
        dst = np.zeros(m, dtype=np.int64)
        for _ in range(scale):
            r = rng.random(m)
            sbit = r >= (a + b)
            dbit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
            src = (src << 1) | sbit
            dst = (dst << 1) | dbit
        perm = np.random.default_rng(seed + 1).permutation(n)
        return perm[src].astype(np.int64), perm[dst].astype(np.int64), n
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, generator=g, device=device)
        sbit = (r >= (a + b)).to(torch.int64)
        dbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
        src = (src << 1) | sbit
        dst = (dst << 1) | dbit
    g2 = torch.Generator(device=device)
    g2.manual_seed(seed + 1)
    perm = torch.randperm(n, generator=g2, device=device)
    return perm[src], perm[dst], n


def grid_edges(side, keep=0.6, seed=3):
    """4-neighbour side x side grid with a fraction of edges kept (road-like stand-in)."""
    idx = np.arange(side * side, dtype=np.int64).reshape(side, side)
    right = np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1)
    down = np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()], 1)
    e = np.concatenate([right, down])
    rng = np.random.default_rng(seed)
    e = e[rng.random(e.shape[0]) < keep]
    return e[:, 0], e[:, 1], side * side


def finalize_edges(src, dst, n, symmetrize=True, want_csc=True):
    """-> dict(n, nnz, csr=(ptr, ind), csc=(ptr, ind)); int32 arrays (numpy or torch)."""
    is_torch = not isinstance(src, np.ndarray)
    if is_torch:
        import torch
        if symmetrize:
            src, dst = torch.cat([src, dst]), torch.cat([dst, src])
        keep = src != dst
        key = torch.unique(src[keep] * n + dst[keep])          # sorted, duplicates dropped
        row = torch.div(key, n, rounding_mode="floor")
        col = key - row * n
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
        ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
        out = dict(n=n, nnz=int(key.numel()), csr=(ptr.to(torch.int32), col.to(torch.int32)))
        if want_csc:
            if symmetrize:
                out["csc"] = out["csr"]
            else:
                tkey = torch.sort(col * n + row).values
                tcol = torch.div(tkey, n, rounding_mode="floor")
                trow = tkey - tcol * n
                tptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
                tptr[1:] = torch.cumsum(torch.bincount(tcol, minlength=n), 0)
                out["csc"] = (tptr.to(torch.int32), trow.to(torch.int32))
        return out
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    if symmetrize:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    keep = src != dst
    key = np.unique(src[keep] * n + dst[keep])
    row, col = key // n, key % n
    ptr = np.zeros(n + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(row, minlength=n))
    out = dict(n=n, nnz=int(key.size), csr=(ptr.astype(np.int32), col.astype(np.int32)))
    if want_csc:
        if symmetrize:
            out["csc"] = out["csr"]
        else:
            tkey = np.sort(col * n + row)
            tcol, trow = tkey // n, tkey % n
            tptr = np.zeros(n + 1, dtype=np.int64)
            tptr[1:] = np.cumsum(np.bincount(tcol, minlength=n))
            out["csc"] = (tptr.astype(np.int32), trow.astype(np.int32))
    return out


def random_sources(ptr, count, seed=0):
    """Sources with nonzero out-degree (mirrors test/grandbfs.cu:95-96, which draws from
    std::mt19937(0); the stream itself is libstdc++-specific, so only the policy is kept)."""
    ptr = np.asarray(ptr)
    deg = np.diff(ptr)
    rng = np.random.default_rng(seed)
    out = []
    n = deg.size
    while len(out) < count:
        s = int(rng.integers(0, n))
        if deg[s] > 0:
            out.append(s)
    return out
