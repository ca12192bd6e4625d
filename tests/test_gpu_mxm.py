"""Masked SpGEMM (csrc/mxm.hip: the pivot-driven kernels and the entry-driven one) against the oracle's restatement of
spgemmMasked (oracle/ops.py: mxm_masked, spgemm.hpp:22-110): every semiring, stored values, zero mask values, A != B,
and the long-pivot / arena paths on hub-rich RMAT graphs (sampled mask entries, the reference loop per entry)."""
import numpy as np
import pytest

from backends import HipBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


def _rand_csr(rng, n, m):
    from graphblast_amd.graphgen import finalize_edges
    gr = finalize_edges(rng.integers(0, n, m), rng.integers(0, n, m), n, symmetrize=False)
    return gr["csr"]


def _entry(sr, arow, aval, bcol, bval):
    """one mask entry as mxm_masked forms it: add(mul(a, b), acc) over the common columns in ascending order"""
    common, ia, ib = np.intersect1d(arow, bcol, return_indices=True)
    acc = sr.identity()
    for x, y in zip(aval[ia], bval[ib]):
        acc = sr.add_op(sr.mul_op(x, y), acc)[()]
    return acc


@pytest.mark.parametrize("dt", [np.int32, np.float32])
def test_every_semiring_with_values_and_zero_mask_entries(hb, dt):
    from oracle import ops as oops
    from oracle.semiring import Semiring, SEMIRINGS
    g = hb.g
    rng = np.random.default_rng(5)
    n = 300
    (ap, ai), (bp, bi), (mp, mi) = _rand_csr(rng, n, 6000), _rand_csr(rng, n, 6000), _rand_csr(rng, n, 4000)
    av = rng.integers(1, 5, ai.size).astype(dt)
    bv = rng.integers(1, 5, bi.size).astype(dt)
    mv = (rng.random(mi.size) < 0.8).astype(dt)               # a fifth of the mask's stored values are 0: skipped
    A, B, M = g.Matrix(n, n, dt), g.Matrix(n, n, dt), g.Matrix(n, n, dt)
    assert A.build_csr(ap, ai, av) == 0 and B.build_csr(bp, bi, bv) == 0 and M.build_csr(mp, mi, mv) == 0
    Ao, Bo, Mo = oops.Matrix(n, n, dt), oops.Matrix(n, n, dt), oops.Matrix(n, n, dt)
    Ao.build_csr(ap, ai, av); Bo.build_csr(bp, bi, bv); Mo.build_csr(mp, mi, mv)
    for tran_b in (True, False):
        d = hb.descriptor()
        do = oops.Descriptor(); do.loadArgs()
        if tran_b:
            assert d.toggle(g.GrB_INP1) == 0
            do.toggle(oops.GrB_INP1)
        for name in SEMIRINGS:
            Cm = g.Matrix(n, n, dt)
            assert g.mxm(Cm, M, None, name, A, B, d) == 0, name
            cp, ci, cv = Cm.host_csr()
            want = oops.mxm_masked(Mo, Semiring(name, dt), Ao, Bo, do)
            assert np.array_equal(cp, mp) and np.array_equal(ci, mi)
            if dt == np.float32 and name == "PlusDivides":
                # quotients are not integers: a float sum folded in another order than the reference's loop differs
                # in the last bit (the pivot kernels fold a row's products as a tree); everything else here is exact
                assert np.allclose(cv, want, rtol=4e-7, atol=0), (name, tran_b)
            else:
                assert np.array_equal(cv, want), (name, tran_b, int((cv != want).sum()))


@pytest.mark.parametrize("scale,values", [(15, "ones"), (15, "int"), (18, "ones"), (18, "int")])
def test_long_pivots_on_rmat_sampled_entries(hb, scale, values):
    """hub rows of thousands (scale 15: the workgroup kernel's LDS tables) and tens of thousands of entries (scale 18:
    the 128 KiB tables and the global-memory arena); stored values take the key + value slots, ones the key-only ones"""
    import torch
    from oracle.semiring import Semiring
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    g = hb.g
    s, d_, n = rmat_edges(scale, 16, seed=2, device=torch.device("cuda", 0))
    gr = finalize_edges(s, d_, n, symmetrize=True)
    ptr, ind = (x.cpu().numpy() for x in gr["csr"])
    rng = np.random.default_rng(scale)
    dt = np.int32
    vals = np.ones(ind.size, dtype=dt) if values == "ones" else rng.integers(1, 4, ind.size).astype(dt)
    A = g.Matrix(n, n, dt)
    assert A.build_csr(ptr, ind, vals) == 0
    d = hb.descriptor()
    L, Cm = g.Matrix(n, n, dt), g.Matrix(n, n, dt)
    assert g.tril(L, A, d) == 0
    lp, li, lv = L.host_csr()
    assert d.toggle(g.GrB_INP1) == 0                          # C<L> = L x L^T, as tc.hpp calls it
    deg = np.diff(lp)
    hub = int(np.argmax(deg))
    assert deg[hub] > (2000 if scale == 15 else 16384)       # beyond the 128 KiB key-only table: the arena
    for name in ("PlusMultiplies", "MinimumPlus"):
        assert g.mxm(Cm, L, None, name, L, L, d) == 0
        cp, ci, cv = Cm.host_csr()
        assert np.array_equal(cp, lp) and np.array_equal(ci, li)
        sr = Semiring(name, dt)
        rows = np.repeat(np.arange(n), deg)
        # entries of the hub row and of rows that meet the hub's column (its partners in both passes), plus a random sample
        pick = np.concatenate([np.arange(lp[hub], lp[hub + 1])[:: max(1, deg[hub] // 300)],
                               np.nonzero(li == hub)[0][:300], rng.integers(0, li.size, 1500)])
        for e in pick:
            i, j = rows[e], li[e]
            want = _entry(sr, li[lp[i]:lp[i + 1]], lv[lp[i]:lp[i + 1]], li[lp[j]:lp[j + 1]], lv[lp[j]:lp[j + 1]])
            assert cv[e] == want, (name, int(e), int(i), int(j), int(cv[e]), int(want))
    if values == "ones":                                       # the triangle count is the sum of the PlusMultiplies product
        assert g.mxm(Cm, L, None, "PlusMultiplies", L, L, d) == 0
        total = int(Cm.host_csr()[2].astype(np.int64).sum())
        T = g.Matrix(n, n, dt)
        info, ntris, _ = g.tc(L, T, hb.descriptor())
        assert info == 0 and ntris == total


def test_result_matrix_reused_and_multiplied(hb):
    """The output matrix of one product is handed to the next (its arrays stay when the shape allows: a second mask with
    the same row lengths but other columns must still give ITS structure and values), and a product's result is usable in
    mxv -- its SpMV plan is built when first asked for, not by mxm."""
    from oracle import ops as oops
    from oracle.semiring import Semiring
    g = hb.g
    dt = np.float32
    rng = np.random.default_rng(9)
    n = 700
    (ap, ai), (bp, bi), (mp, mi) = _rand_csr(rng, n, 15000), _rand_csr(rng, n, 15000), _rand_csr(rng, n, 9000)
    av = rng.integers(1, 4, ai.size).astype(dt)
    bv = rng.integers(1, 4, bi.size).astype(dt)
    # the second mask: every row's columns mirrored (n - 1 - c), re-sorted per row: same pointers, other columns
    mi2 = np.concatenate([np.sort(n - 1 - mi[mp[r]:mp[r + 1]]) for r in range(n)]).astype(mi.dtype)
    A, B = g.Matrix(n, n, dt), g.Matrix(n, n, dt)
    assert A.build_csr(ap, ai, av) == 0 and B.build_csr(bp, bi, bv) == 0
    Ao, Bo = oops.Matrix(n, n, dt), oops.Matrix(n, n, dt)
    Ao.build_csr(ap, ai, av); Bo.build_csr(bp, bi, bv)
    d = hb.descriptor()
    assert d.toggle(g.GrB_INP1) == 0
    do = oops.Descriptor(); do.loadArgs(); do.toggle(oops.GrB_INP1)
    Cm = g.Matrix(n, n, dt)
    u = rng.integers(0, 5, n).astype(dt)
    for cols in (mi, mi2, mi):
        M, Mo = g.Matrix(n, n, dt), oops.Matrix(n, n, dt)
        ones = np.ones(cols.size, dtype=dt)
        assert M.build_csr(mp, cols, ones) == 0
        Mo.build_csr(mp, cols, ones)
        assert g.mxm(Cm, M, None, "PlusMultiplies", A, B, d) == 0
        cp, ci, cv = Cm.host_csr()
        want = oops.mxm_masked(Mo, Semiring("PlusMultiplies", dt), Ao, Bo, do)
        assert np.array_equal(cp, mp) and np.array_equal(ci, cols) and np.array_equal(cv, want)
        # w = C u through the generic SpMV (small integers: exact in any order)
        uv, w = g.Vector(n, dt), g.Vector(n, dt)
        assert uv.build(u, n) == 0
        # (pull: the result has no CSC side -- as little as the reference's C->dup has one -- so a pushed product on it
        # reports GrB_INVALID_OBJECT, and parseArgs' default mxvmode is push)
        info = g.mxv(w, None, None, "PlusMultiplies", Cm, uv, hb.descriptor(mxvmode=2))
        assert info == 0, info
        assert g.mxv(g.Vector(n, dt), None, None, "PlusMultiplies", Cm, uv, hb.descriptor(mxvmode=1)) == 12      # GrB_INVALID_OBJECT
        dense = np.zeros((n, n), dtype=np.float64)
        dense[np.repeat(np.arange(n), np.diff(cp)), ci] = cv
        assert np.array_equal(hb.dense_values(w).astype(np.float64), dense @ u.astype(np.float64))



@pytest.mark.parametrize("scale,k_want", [(11, 200), (13, 700), (14, 2100)])
def test_dense_core(hb, scale, k_want):
    """grb_tc_dense_core (csrc/mxm_core.hip): the product C<L> = L (+.x) L^T restricted to the longest rows of L, as bit
    rows -- by AND + popcount per mask entry, by v_mfma_i32_16x16x64_i8 on expanded 0/1 bytes with the mask applied to
    the finished tile, and by both (MFMA for the denser tiles).  All three against numpy's dense product on the same
    core: the count, the number of entries, and a position-weighted checksum of the per-entry results."""
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    g = hb.g
    s, d, n = rmat_edges(scale, 24, seed=11)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    rows = np.repeat(np.arange(n), np.diff(ptr))
    keep = ind < rows
    li, lj = rows[keep], ind[keep]
    lptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(li, minlength=n), out=lptr[1:])
    L = g.Matrix(n, n, np.int32)
    assert L.build_csr(lptr, lj.astype(np.int32), np.ones(lj.size, dtype=np.int32)) == 0
    got = {}
    for method, dense_from in ((0, 0), (1, 0), (2, 600), (2, 3000)):
        info, r = g.tc_dense_core(L, k_want, method, dense_from)
        assert info == 0, (method, info)
        got[(method, dense_from)] = r
    r0 = got[(0, 0)]
    K, theta = r0["core_rows"], r0["min_row_length"]
    dl = np.diff(lptr)
    core = np.nonzero(dl >= theta)[0]
    assert K == core.size and 2 <= K <= k_want
    assert np.count_nonzero(dl >= theta - 1) > k_want or theta == 1       # the threshold is the lowest that fits
    rank = np.full(n, -1)
    rank[core] = np.arange(K)
    H = np.zeros((K, K), dtype=np.int64)
    cc = (rank[li] >= 0) & (rank[lj] >= 0)
    H[rank[li[cc]], rank[lj[cc]]] = 1
    P = (H @ H.T) * H                                                      # the mask: L's own entries between core rows
    ent_r, ent_c = np.nonzero(H)                                           # row-major = the order of the stored results
    vals = P[ent_r, ent_c].astype(np.uint64)
    pos = np.arange(vals.size, dtype=np.uint64)
    weight = ((pos * np.uint64(2654435761)) & np.uint64(0xffffffff)) | np.uint64(1)
    want_sum = int((vals * weight).sum() & np.uint64(0xffffffffffffffff))
    for key, r in got.items():
        assert r["core_entries"] == int(H.sum()), key
        assert r["count"] == int(P.sum()), key
        assert r["checksum"] == want_sum, key
        assert sum(r["tiles_by_density"]) == r["tiles"]
    assert got[(1, 0)]["tiles_mfma"] == got[(1, 0)]["tiles"] and got[(0, 0)]["tiles_mfma"] == 0
    assert 0 <= got[(2, 3000)]["tiles_mfma"] <= got[(2, 600)]["tiles_mfma"] <= got[(1, 0)]["tiles"]


@pytest.mark.parametrize("mfma_from", ["0", "1", "600"])
def test_triangle_count_with_the_dense_core(hb, mfma_from):
    """grb_mxm's dense-core path (C<L> = L (+.x) L^T through grb_tc): the entries between the longest rows come from the
    bit rows (popcount only, MFMA only, both) plus the pivot passes on the lists without the core vertices; everything
    else from the passes over the whole mask with those entries switched off.  Every stored value of the product equals
    the product computed with the core off, and the count the reference's SimpleReferenceTc."""
    import os
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    from oracle import simple_reference as sr
    g = hb.g
    s, d, n = rmat_edges(14, 24, seed=3)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    A = g.Matrix(n, n, np.int32)
    assert A.build_csr(ptr, ind, np.ones(ind.size, dtype=np.int32)) == 0
    L = g.Matrix(n, n, np.int32)
    assert g.tril(L, A, hb.descriptor()) == 0
    lp, li, lv = L.host_csr()
    saved = {k: os.environ.get(k) for k in ("GRB_TC_CORE_K", "GRB_TC_CORE_MIN_NVALS", "GRB_TC_CORE_MFMA_FROM")}
    was = g.tc_set_product(1)                                   # (the product in B is what this test is about)
    try:
        os.environ["GRB_TC_CORE_K"] = "0"
        B0 = g.Matrix(n, n, np.int32)
        info, want_n, _ = g.tc(L, B0, hb.descriptor())
        assert info == 0 and want_n == sr.tc(lp, li)[0] and want_n > 0
        _, _, want = B0.host_csr()
        for k in ("300", "1500"):
            os.environ.update(GRB_TC_CORE_K=k, GRB_TC_CORE_MIN_NVALS="0", GRB_TC_CORE_MFMA_FROM=mfma_from)
            B1 = g.Matrix(n, n, np.int32)
            info, got_n, _ = g.tc(L, B1, hb.descriptor())
            assert info == 0 and got_n == want_n, (k, got_n, want_n)
            bp, bi, bv = B1.host_csr()
            assert np.array_equal(bp, lp) and np.array_equal(bi, li)
            assert np.array_equal(bv, want), (k, int(np.count_nonzero(bv != want)))
            info, got_n, _ = g.tc(L, B1, hb.descriptor())             # the same output matrix again (its arrays are kept)
            assert info == 0 and got_n == want_n
    finally:
        g.tc_set_product(was)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _lower(ptr, ind, n, strict=True):
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
    keep = ind < rows if strict else ind <= rows
    lp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[keep], minlength=n), out=lp[1:])
    return lp, ind[keep].astype(np.int32)


def _tc_graphs():
    from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges
    rng = np.random.default_rng(11)
    out = []
    s, d, n = rmat_edges(13, 24, seed=5)                                    # hubs: lists of several hundred
    out.append(("rmat13", finalize_edges(s, d, n, symmetrize=True)))
    s, d, n = grid_edges(60, keep=0.7, seed=2)                              # road-like: nothing but short lists
    out.append(("grid", finalize_edges(s, d, n, symmetrize=True)))
    n = 900                                                                 # dense: every list long, ties in the degrees
    out.append(("dense", finalize_edges(rng.integers(0, n, 300000), rng.integers(0, n, 300000), n, symmetrize=True)))
    n = 640                                                                 # complete graph: the ranking is the ids alone
    a, b = np.meshgrid(np.arange(n), np.arange(n))
    out.append(("complete", finalize_edges(a.ravel(), b.ravel(), n, symmetrize=True)))
    n = 3000                                                                # a star, a path and isolated vertices
    src = np.concatenate([np.zeros(1500, dtype=np.int64), np.arange(1600, 2500)])
    dst = np.concatenate([np.arange(1, 1501), np.arange(1601, 2501)])
    out.append(("star+path", finalize_edges(src, dst, n, symmetrize=True)))
    return out


@pytest.mark.parametrize("bitmap_upto", [None, "100"])
def test_triangle_count_without_the_product(hb, bitmap_upto):
    """grb_tc on a strictly lower triangle of ones counts on the degree-ordered orientation (csrc/tc_count.hip) and leaves
    the buffer matrix alone; the number is SimpleReferenceTc's and the product path's, for hub graphs, road-like graphs,
    dense graphs (bitmap kernel; with GRB_TC_BITMAP_UPTO=100 the hash-table kernel of the pivots numbered beyond the
    bitmap), f32 and i32.  Whatever is not such a matrix -- a diagonal entry, a value that is not 1, a transposed first
    operand -- goes through the reference's two calls, as does everything after grb_tc_set_product(1)."""
    import os
    from oracle import simple_reference as sr
    g = hb.g
    saved = os.environ.get("GRB_TC_BITMAP_UPTO")
    if bitmap_upto is not None:
        os.environ["GRB_TC_BITMAP_UPTO"] = bitmap_upto
    was = g.tc_set_product(2)                                   # the count wherever it is a count (0 would send the first count on a hub-free graph through the product)
    try:
        seen = [0, 0, 0]
        for name, gr in _tc_graphs():
            ptr, ind = np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1])
            n = gr["n"]
            lp, li = _lower(ptr, ind, n)
            want = sr.tc(lp, li)[0]
            for dt in (np.int32, np.float32):
                L, B = g.Matrix(n, n, dt), g.Matrix(n, n, dt)
                assert L.build_csr(lp, li, np.ones(li.size, dtype=dt)) == 0
                for rep in range(2):                                        # the second count finds the orientation with the matrix
                    info, ntris, _ = g.tc(L, B, hb.descriptor())
                    last = g.tc_last()[1]
                    assert info == 0 and ntris == want, (name, dt, rep, ntris, want)
                    assert last["path"] == 1 and (rep == 0 or last["prep_ms"] < 1.0), (name, last)
                    assert B.nvals() == 0                                   # the buffer matrix was not touched
                for k in range(3):
                    seen[k] += last["tasks"][k]
                if dt == np.int32:                                          # released: the next count prepares it again
                    assert g.tc_release(L) == 0
                    info, ntris, _ = g.tc(L, B, hb.descriptor())
                    assert info == 0 and ntris == want and g.tc_last()[1]["prep_ms"] > g.tc_last()[1]["count_ms"] * 0.05
                # the reference's two calls on the same matrix: the same number, the product in B
                g.tc_set_product(1)
                info, ntris, _ = g.tc(L, B, hb.descriptor())
                assert info == 0 and g.tc_last()[1]["path"] == 0
                # (the reduction of an f32 product is an f32 sum: exact up to 2^24 only; the count path's is an integer)
                assert ntris == want if dt == np.int32 else abs(ntris - want) <= 2e-7 * want, (name, dt, ntris, want)
                assert int(B.host_csr()[2].astype(np.int64).sum()) == want
                g.tc_set_product(2)
                # values changed in place: the kept orientation goes with them (4 = 2 x 2 per common neighbour now)
                if name == "rmat13" and dt == np.int32:
                    assert g.apply(L, None, None, "bind_second", L, hb.descriptor(), binop="multiplies", scalar=2) == 0
                    info, ntris, _ = g.tc(L, B, hb.descriptor())
                    assert info == 0 and g.tc_last()[1]["path"] == 0 and ntris == 4 * want
        assert seen[0] > 0 and (seen[1] > 0 if bitmap_upto is None else seen[2] > 0), seen
        # not such a matrix
        name, gr = _tc_graphs()[0]
        ptr, ind = np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1])
        n = gr["n"]
        lp, li = _lower(ptr, ind, n)
        want = sr.tc(lp, li)[0]
        vals = np.ones(li.size, dtype=np.int32)
        vals[li.size // 2] = 3
        L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert L.build_csr(lp, li, vals) == 0
        info, ntris, _ = g.tc(L, B, hb.descriptor())
        assert info == 0 and g.tc_last()[1]["path"] == 0 and ntris == int(B.host_csr()[2].astype(np.int64).sum()) != want
        info, again, _ = g.tc(L, B, hb.descriptor())                        # (the verdict is kept with the matrix too)
        assert info == 0 and again == ntris and g.tc_last()[1]["path"] == 0
        dp, di = _lower(ptr, np.asarray(ind), n)                            # a diagonal entry: row 5 gets (5, 5)
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(dp))
        r2, c2 = np.concatenate([rows, [5]]), np.concatenate([di, [5]])
        o = np.lexsort((c2, r2))
        p2 = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(r2, minlength=n), out=p2[1:])
        L2, B2 = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert L2.build_csr(p2, c2[o].astype(np.int32), np.ones(c2.size, dtype=np.int32)) == 0
        info, ntris, _ = g.tc(L2, B2, hb.descriptor())
        assert info == 0 and g.tc_last()[1]["path"] == 0 and ntris == sr.tc(p2, c2[o].astype(np.int32))[0]
        # a strictly UPPER triangle of ones: the same count, on the orientation as well
        up_rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(lp))
        o = np.lexsort((up_rows, li))
        upp = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(li, minlength=n), out=upp[1:])
        U, BU = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert U.build_csr(upp, up_rows[o].astype(np.int32), np.ones(li.size, dtype=np.int32)) == 0
        info, ntris, _ = g.tc(U, BU, hb.descriptor())
        assert info == 0 and ntris == want and g.tc_last()[1]["path"] == 1
        g.tc_set_product(1)
        info, ntris, _ = g.tc(U, BU, hb.descriptor())
        g.tc_set_product(2)
        assert info == 0 and ntris == want and g.tc_last()[1]["path"] == 0
        # the first operand transposed: grb_mxm's business
        L3, B3 = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert L3.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
        d = hb.descriptor()
        d.toggle(g.GrB_INP0)
        info, ntris, _ = g.tc(L3, B3, d)
        assert info == 0 and g.tc_last()[1]["path"] == 0
    finally:
        g.tc_set_product(was)
        if saved is None:
            os.environ.pop("GRB_TC_BITMAP_UPTO", None)
        else:
            os.environ["GRB_TC_BITMAP_UPTO"] = saved


@pytest.mark.parametrize("bitmap_upto", [None, "66000"])
def test_triangle_count_numbers_beyond_16_bits(hb, bitmap_upto):
    """The oriented lists keep the numbers below 65 535 as 16-bit entries and the others as 32-bit entries, nobody is numbered
    65 535, and a pivot numbered below 65 536 does not stream its partners' 32-bit parts: graphs with more than 65 536
    vertices whose low-ranked vertices have long lists -- a sparse one (every pivot a wave's, most of them beyond 65 536)
    and a dense one (pivots beyond 65 536 with lists of several hundred entries: the bitmap kernel's wide path, and with
    GRB_TC_BITMAP_UPTO=66000 the 512-thread hash-table kernel's).  The count equals the product path's sum, and the CPU
    reference's on the sparse graph."""
    import os
    from graphblast_amd.graphgen import finalize_edges
    from oracle import simple_reference as sr
    g = hb.g
    saved = os.environ.get("GRB_TC_BITMAP_UPTO")
    if bitmap_upto is not None:
        os.environ["GRB_TC_BITMAP_UPTO"] = bitmap_upto
    was = g.tc_set_product(2)
    try:
        rng = np.random.default_rng(23)
        seen = [0, 0, 0]
        for name, n, m in (("sparse", 150000, 1500000), ("dense", 67000, 14000000)):
            gr = finalize_edges(rng.integers(0, n, m), rng.integers(0, n, m), n, symmetrize=True)
            lp, li = _lower(np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1]), n)
            del gr
            L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
            assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
            info, ntris, _ = g.tc(L, B, hb.descriptor())
            last = g.tc_last()[1]
            assert info == 0 and last["path"] == 1 and B.nvals() == 0, (name, last)
            for k in range(3):
                seen[k] += last["tasks"][k]
            if name == "sparse":
                assert ntris == sr.tc(lp, li)[0], name
            else:
                assert last["longest_list"] > 256, last
            g.tc_set_product(1)
            info, want, _ = g.tc(L, B, hb.descriptor())
            g.tc_set_product(2)
            assert info == 0 and g.tc_last()[1]["path"] == 0 and ntris == want > 0, (name, ntris, want)
        assert seen[0] > 0 and (seen[1] > 0 if bitmap_upto is None else seen[2] > 0), seen
    finally:
        g.tc_set_product(was)
        if saved is None:
            os.environ.pop("GRB_TC_BITMAP_UPTO", None)
        else:
            os.environ["GRB_TC_BITMAP_UPTO"] = saved


def test_triangle_count_random_graphs(hb):
    """Forty random graphs -- a single vertex, no edges, paths, cliques, power-law and uniform ones, up to 20 000 vertices --
    counted on the orientation and through the product: both equal SimpleReferenceTc."""
    from graphblast_amd.graphgen import finalize_edges
    from oracle import simple_reference as sr
    g = hb.g
    rng = np.random.default_rng(77)
    was = g.tc_set_product(0)
    try:
        paths = [0, 0]
        for trial in range(40):
            n = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 2000, 20000]))
            m = int(rng.choice([0, 1, n, 4 * n, 30 * n]))
            if trial % 4 == 0:                                   # skewed endpoints: a few hubs
                s_ = (rng.random(m) ** 3 * n).astype(np.int64)
                d_ = rng.integers(0, n, m)
            else:
                s_, d_ = rng.integers(0, n, m), rng.integers(0, n, m)
            gr = finalize_edges(s_, d_, n, symmetrize=True)
            lp, li = _lower(np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1]), n)
            want = sr.tc(lp, li)[0] if li.size else 0
            L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
            assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
            for product in (2, 1, 0, 0):
                g.tc_set_product(product)
                info, ntris, _ = g.tc(L, B, hb.descriptor())
                assert info == 0 and ntris == want, (trial, n, m, product, ntris, want)
                paths[g.tc_last()[1]["path"]] += 1
        assert paths[1] >= 20, paths
    finally:
        g.tc_set_product(was)


def test_triangle_count_first_count_on_a_matrix_without_long_rows(hb):
    """The default (grb_tc_set_product(0)): a matrix whose rows are all short is cheap for the product and the orientation's
    preparation is not -- its FIRST count goes through the reference's two calls, the second prepares the orientation, the
    third finds it with the matrix; a hub graph prepares at once.  The same number every time."""
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    from oracle import simple_reference as sr
    g = hb.g
    rng = np.random.default_rng(3)
    was = g.tc_set_product(0)
    try:
        n = 30000
        gr = finalize_edges(rng.integers(0, n, 300000), rng.integers(0, n, 300000), n, symmetrize=True)
        lp, li = _lower(np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1]), n)
        want = sr.tc(lp, li)[0]
        L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
        trace = []
        for rep in range(3):
            info, ntris, _ = g.tc(L, B, hb.descriptor())
            assert info == 0 and ntris == want
            trace.append((g.tc_last()[1]["path"], g.tc_last()[1]["prep_ms"] > 0.05))
        assert trace == [(0, False), (1, True), (1, False)], trace
        s_, d_, n = rmat_edges(14, 24, seed=3)
        gr = finalize_edges(s_, d_, n, symmetrize=True)
        lp, li = _lower(np.asarray(gr["csr"][0]), np.asarray(gr["csr"][1]), n)
        L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert L.build_csr(lp, li, np.ones(li.size, dtype=np.int32)) == 0
        info, ntris, _ = g.tc(L, B, hb.descriptor())
        assert info == 0 and ntris == sr.tc(lp, li)[0] and g.tc_last()[1]["path"] == 1
    finally:
        g.tc_set_product(was)
