"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle on
the same seeded inputs.  Integer-valued float data keeps every semiring exact, so the
comparison is bit-exact; one test uses real-valued floats with the 1e-5 relative bar."""
import numpy as np
import pytest

from backends import HipBackend, OracleBackend
import ref_cases

pytestmark = pytest.mark.gpu
F = np.float32
FLT_MAX = np.finfo(np.float32).max

ALL_SR = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusDivides", "PlusGreater",
          "GreaterPlus", "PlusMinus", "PlusLess", "CustomLessPlus", "MinimumMultiplies", "MultipliesMultiplies",
          "NotEqualToPlus", "MinimumSelectSecond", "PlusNotEqualTo", "CustomLessLess", "MinimumNotEqualTo"]
# order-independent additive monoids (results do not depend on the reduction order)
ORDER_FREE = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusGreater", "PlusMinus",
              "PlusLess", "MinimumMultiplies", "MinimumSelectSecond", "PlusNotEqualTo", "MinimumNotEqualTo"]


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


@pytest.fixture(scope="module")
def ob():
    return OracleBackend()


def rand_graph(n, m, seed, hub=0):
    from graphblast_amd.graphgen import finalize_edges
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    if hub:                      # one row/column far longer than an SpMV tile (2048) and a slice (8192)
        hs = np.full(hub, n // 3)
        hd = rng.integers(0, n, hub)
        src, dst = np.concatenate([src, hs, hd]), np.concatenate([dst, hd, hs])
    g = finalize_edges(src, dst, n, symmetrize=False)
    ptr, ind = g["csr"]
    val = rng.integers(1, 5, ind.size).astype(F)
    return n, ptr, ind, val


def both(hb, ob, n, ptr, ind, val, dtype=np.float32):
    return hb.matrix_from_csr(n, ptr, ind, val.astype(dtype), dtype), ob.matrix_from_csr(n, ptr, ind, val.astype(dtype), dtype)


def same(got, want, exact=True, what=""):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if exact:
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (what, bad[:10], got[bad[:10]], want[bad[:10]])
    else:
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6), what


def test_device_is_gfx950(hip):
    info = hip.device_info()
    assert info.startswith("gfx950"), info


def test_reference_unit_cases_on_hip(hb):
    """test/gvxm.cu, gewiseadd.cu, gewisemult.cu, greduce.cu literal cases, exact."""
    results = ref_cases.run_all(hb)
    assert len(results) > 40
    for label, got, cor in results:
        same(got, cor, what=label)


@pytest.mark.parametrize("n,m,hub", [(1, 0, 0), (63, 200, 0), (65, 300, 0), (1000, 6000, 0), (5000, 20000, 12000)])
def test_mxv_vxm_pull_all_semirings(hb, ob, n, m, hub):
    """Generic SpMV (dense u): every semiring, no mask / mask / scmp / accum, CSR and CSC
    orientation, ragged sizes, empty rows, a hub row (long-row path)."""
    n, ptr, ind, val = rand_graph(n, m, 11 + n, hub)
    rng = np.random.default_rng(n)
    uvals = rng.integers(0, 4, n).astype(F)
    maskv = (rng.random(n) < 0.5).astype(F)
    wprev = rng.integers(0, 3, n).astype(F)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    for sr in ALL_SR:
        exact_sr = sr in ORDER_FREE
        for opname in ("mxv", "vxm"):
            for use_mask, scmp, accum in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1)):
                if not exact_sr and n > 100:
                    continue            # order-dependent "monoids": tiny cases only (single-entry rows vary)
                outs = []
                for be, A in ((hb, A_h), (ob, A_o)):
                    d = be.descriptor(mxvmode=2, fusedmask=0)
                    if scmp:
                        be.set(d, 0, 0)
                    u = be.vector(n); be.build_dense(u, uvals)
                    w = be.vector(n); be.build_dense(w, wprev)
                    mk = None
                    if use_mask:
                        mk = be.vector(n); be.build_dense(mk, maskv)
                    acc = "accum" if accum else None
                    if opname == "mxv":
                        info = be.mxv(w, mk, acc, sr, A, u, d)
                    else:
                        info = be.vxm(w, mk, acc, sr, u, A, d)
                    assert info == 0
                    assert be.lastmxv(d) == 12
                    outs.append(be.dense_values(w))
                if exact_sr:
                    same(outs[0], outs[1], what=(sr, opname, use_mask, scmp, accum, n))


def test_spmv_float_tolerance(hb, ob):
    """Real-valued PlusMultiplies / MinimumPlus: within 1e-5 relative of the sequential fold."""
    n, ptr, ind, _ = rand_graph(3000, 30000, 5, hub=9000)
    rng = np.random.default_rng(1)
    val = rng.random(ind.size).astype(F)
    uvals = rng.random(n).astype(F)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    for sr in ("PlusMultiplies", "MinimumPlus"):
        outs = []
        for be, A in ((hb, A_h), (ob, A_o)):
            d = be.descriptor(mxvmode=2)
            u = be.vector(n); be.build_dense(u, uvals)
            w = be.vector(n)
            assert be.mxv(w, None, None, sr, A, u, d) == 0
            outs.append(be.dense_values(w))
        assert np.allclose(outs[0], outs[1], rtol=1e-5, atol=0), sr


@pytest.mark.parametrize("n,m", [(64, 300), (777, 4000), (4096, 30000)])
def test_masked_or_pull_all_variants(hb, ob, n, m):
    """Boolean fused-mask pull: 8 <scmp, earlyexit, opreuse> variants (kernels/spmv.hpp:10-59)."""
    n, ptr, ind, val = rand_graph(n, m, 3 * n, hub=300 if n > 1000 else 0)
    rng = np.random.default_rng(n)
    visited = (rng.random(n) < 0.4).astype(F) * rng.integers(1, 4, n).astype(F)
    frontier = ((visited != 0) & (rng.random(n) < 0.5)).astype(F)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    for scmp in (0, 1):
        for ee in (0, 1):
            for orr in (0, 1):
                outs = []
                for be, A in ((hb, A_h), (ob, A_o)):
                    d = be.descriptor(mxvmode=2, earlyexit=ee, opreuse=orr, fusedmask=1)
                    if scmp:
                        be.set(d, 0, 0)
                    u = be.vector(n); be.build_dense(u, frontier)
                    mk = be.vector(n); be.build_dense(mk, visited)
                    w = be.vector(n)
                    assert be.vxm(w, mk, None, "LogicalOrAnd", u, A, d) == 0
                    outs.append(be.dense_values(w))
                same(outs[0], outs[1], what=(scmp, ee, orr))


@pytest.mark.parametrize("n,m,nf", [(50, 200, 1), (1000, 8000, 37), (1000, 8000, 1000), (6000, 40000, 2500)])
def test_push_spmspv(hb, ob, n, m, nf):
    """SpMSpV push: structure-only and key-value, unmasked / mask / scmp, sorted unique
    output, zero-degree frontier entries, a hub in the frontier."""
    n, ptr, ind, val = rand_graph(n, m, 9 + nf, hub=9000 if n >= 6000 else 0)
    rng = np.random.default_rng(nf)
    fidx = np.sort(rng.choice(n, nf, replace=False)).astype(np.int32)
    if n >= 6000:
        fidx = np.unique(np.append(fidx, n // 3)).astype(np.int32)
    fval = rng.integers(0, 4, fidx.size).astype(F)      # zeros exercise the identity short-circuit + prune
    maskv = (rng.random(n) < 0.5).astype(F)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    for sr in ("LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "MinimumSelectSecond"):
        for struc in (0, 1):
            for use_mask, scmp in ((0, 0), (1, 0), (1, 1)):
                for opname in ("vxm", "mxv"):
                    outs = []
                    for be, A in ((hb, A_h), (ob, A_o)):
                        d = be.descriptor(mxvmode=1, struconly=struc)
                        if scmp:
                            be.set(d, 0, 0)
                        u = be.vector(n); be.build_sparse(u, fidx, fval)
                        mk = None
                        if use_mask:
                            mk = be.vector(n); be.build_dense(mk, maskv)
                        w = be.vector(n)
                        info = be.vxm(w, mk, None, sr, u, A, d) if opname == "vxm" else be.mxv(w, mk, None, sr, A, u, d)
                        assert info == 0
                        assert be.storage(w) == 1 and be.lastmxv(d) == 11
                        i, v = be.sparse_tuples(w)
                        outs.append((i, v, be.nvals(w)))
                    what = (sr, struc, use_mask, scmp, opname)
                    assert outs[0][2] == outs[1][2], what
                    same(outs[0][0], outs[1][0], what=what)
                    if not struc:
                        same(outs[0][1], outs[1][1], what=what)


def test_vxm_error_behaviour(hb):
    g = hb.g
    n, ptr, ind, val = rand_graph(100, 400, 1)
    A = hb.matrix_from_csr(n, ptr, ind, val)
    d = hb.descriptor()
    u, w = g.Vector(n), g.Vector(n)
    assert g.vxm(w, None, None, "PlusMultiplies", u, A, d) == g.GrB_UNINITIALIZED_OBJECT   # u.nvals == 0
    assert g.vxm(None, None, None, "PlusMultiplies", u, A, d) == g.GrB_UNINITIALIZED_OBJECT
    u2 = g.Vector(n + 1); u2.fill(1.0)
    assert g.vxm(w, None, None, "PlusMultiplies", u2, A, d) == g.GrB_DIMENSION_MISMATCH
    u.fill(1.0)
    g.Descriptor.toggle(d, g.GrB_INP0)
    assert g.vxm(w, None, None, "PlusMultiplies", u, A, d) == g.GrB_INVALID_VALUE
    e = g.Vector(n)
    assert e.build([1, 2], [1.0, 2.0], 2, None) == 0
    assert e.build([3], [1.0], 1, None) == g.GrB_OUTPUT_NOT_EMPTY
    info, _ = e.extractTuples(n=5)
    assert info == g.GrB_INSUFFICIENT_SPACE
    f1, f2 = g.Vector(n), g.Vector(n)
    f1.fill(0.0); f2.build([1], [1.0], 1, None)
    assert f1.swap(f2) == g.GrB_INVALID_OBJECT            # different storage, vector.hpp:430-434


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1024, 1025, 50000])
def test_vector_conversions(hb, ob, n):
    """dense2sparse / sparse2dense / convert heuristic / fill / nvals at ragged sizes."""
    rng = np.random.default_rng(n)
    dv = (rng.random(n) < 0.3).astype(F) * rng.integers(1, 9, n).astype(F)
    for struc in (0, 1):
        outs = []
        for be in (hb, ob):
            d = be.descriptor(struconly=struc)
            v = be.vector(n); be.build_dense(v, dv)
            assert v.dense2sparse(0.0, d) == 0
            i, x = be.sparse_tuples(v)
            nv = be.nvals(v)
            v.sparse2dense(0.0, d)
            outs.append((i, x, nv, be.dense_values(v)))
        assert outs[0][2] == outs[1][2] == int(np.count_nonzero(dv))
        same(outs[0][0], outs[1][0])
        if not struc:
            same(outs[0][1], outs[1][1])
            same(outs[0][3], dv)
        else:
            same(outs[0][3], (dv != 0).astype(F))
    # convert(): both directions and the ratio_ memory (vector.hpp:291-323)
    seq = []
    for be in (hb, ob):
        d = be.descriptor()
        v = be.vector(n); be.build_dense(v, dv)
        trace = []
        for sp in (0.9, 0.9, 0.2, 0.2, 0.9):
            v.convert(0.0, sp, d)
            trace.append((be.storage(v), be.nvals(v)))
        seq.append(trace)
    assert seq[0] == seq[1], seq


def test_assign_reduce(hb, ob):
    n = 5000
    rng = np.random.default_rng(4)
    base = rng.integers(0, 9, n).astype(F)
    maskv = (rng.random(n) < 0.5).astype(F)
    midx = np.nonzero(maskv)[0].astype(np.int32)
    for scmp in (0, 1):
        for mask_kind in ("dense", "sparse"):
            for w_kind in ("dense", "sparse"):
                outs = []
                for be in (hb, ob):
                    d = be.descriptor()
                    if scmp:
                        be.set(d, 0, 0)
                    w = be.vector(n)
                    if w_kind == "dense":
                        be.build_dense(w, base)
                    else:
                        wi = np.nonzero(base)[0].astype(np.int32)
                        be.build_sparse(w, wi, base[wi])
                    mk = be.vector(n)
                    if mask_kind == "dense":
                        be.build_dense(mk, maskv)
                    else:
                        be.build_sparse(mk, midx, np.ones(midx.size, dtype=F))
                    assert be.assign(w, mk, 7.0, d) == 0
                    if w_kind == "dense":
                        outs.append((be.dense_values(w),))
                    else:
                        i, v = be.sparse_tuples(w)
                        outs.append((i, v))
                for a, b in zip(outs[0], outs[1]):
                    same(a, b, what=(scmp, mask_kind, w_kind))
    from oracle.semiring import MONOIDS
    for dtype in (np.float32, np.int32):
        vals = rng.integers(0, 4, n).astype(dtype)
        for mono in MONOIDS:
            if mono in ("Greater", "CustomLess", "NotEqualTo", "Multiplies"):
                continue                         # order-dependent / overflowing folds
            outs = []
            for be in (hb, ob):
                d = be.descriptor()
                u = be.vector(n, dtype); be.build_dense(u, vals)
                info, r = be.reduce(mono, u, d)
                assert info == 0
                outs.append(float(r))
                s = be.vector(n, dtype); be.build_sparse(s, midx, vals[midx])
                outs.append(float(be.reduce(mono, s, d)[1]))
                d2 = be.descriptor(struconly=1)
                outs.append(float(be.reduce(mono, s, d2)[1]))          # == nvals
            assert outs[:3] == outs[3:], (mono, dtype, outs)
    # empty sparse vector -> identity (reduce.hpp:25-28)
    g = hb.g
    e = g.Vector(n); e.setStorage(g.GrB_SPARSE)
    assert g.reduce(None, "Minimum", e, hb.descriptor())[1] == FLT_MAX
    assert g.reduce(None, "Maximum", e, hb.descriptor())[1] == 0.0


def test_ewise_random(hb, ob):
    """eWiseAdd / eWiseMult: every storage combination, aliasing, masks, scalar, both dtypes."""
    n = 3000
    rng = np.random.default_rng(8)
    for dtype in (np.float32, np.int32):
        a = rng.integers(0, 5, n).astype(dtype)
        b = rng.integers(0, 5, n).astype(dtype)
        si = np.sort(rng.choice(n, 400, replace=False)).astype(np.int32)
        sv = rng.integers(1, 5, si.size).astype(dtype)
        mi = np.sort(rng.choice(n, 500, replace=False)).astype(np.int32)
        mv = rng.integers(0, 2, mi.size).astype(dtype)
        md = rng.integers(0, 2, n).astype(dtype)
        for sr in ("PlusMultiplies", "MinimumPlus", "CustomLessPlus", "PlusMinus", "MultipliesMultiplies",
                   "MinimumNotEqualTo", "PlusGreater", "LogicalOrAnd"):
            for opname in ("add", "mult"):
                for combo in ("dd", "sd", "ds", "dd_alias_u", "sd_alias_v", "sd_alias_u", "ss_alias_u",
                              "dd_mdense", "dd_msparse", "sd_mdense", "sd_msparse", "ds_msparse"):
                    if opname == "add" and "_m" in combo:
                        continue
                    outs = []
                    for be in (hb, ob):
                        d = be.descriptor()
                        def mk(kind, dense_vals):
                            v = be.vector(n, dtype)
                            if kind == "d":
                                be.build_dense(v, dense_vals)
                            else:
                                be.build_sparse(v, si, sv)
                            return v
                        u = mk(combo[0], a)
                        v = mk(combo[1], b)
                        w = be.vector(n, dtype)
                        if "alias_u" in combo:
                            w = u
                        if "alias_v" in combo:
                            w = v
                        mask = None
                        if combo.endswith("mdense"):
                            mask = be.vector(n, dtype); be.build_dense(mask, md)
                        if combo.endswith("msparse"):
                            mask = be.vector(n, dtype); be.build_sparse(mask, mi, mv)
                        fn = be.eWiseAdd if opname == "add" else be.eWiseMult
                        info = fn(w, mask, None, sr, u, v, d)
                        st = be.storage(w)
                        if st == 2:
                            outs.append((info, st, be.dense_values(w)))
                        else:
                            i, x = be.sparse_tuples(w)
                            outs.append((info, st, i, x))
                    what = (dtype.__name__, sr, opname, combo)
                    assert outs[0][:2] == outs[1][:2], what
                    if combo == "ss_alias_u" and opname == "mult":
                        continue     # sparse x sparse eWiseMult reads an uninitialised dense side in the reference
                    for x, y in zip(outs[0][2:], outs[1][2:]):
                        same(x, y, what=what)
            outs = []
            for be in (hb, ob):
                d = be.descriptor()
                for kind in ("d", "s"):
                    u = be.vector(n, dtype)
                    if kind == "d":
                        be.build_dense(u, a)
                    else:
                        be.build_sparse(u, si, sv)
                    w = be.vector(n, dtype)
                    assert be.eWiseAdd(w, None, None, sr, u, 3, d) == 0
                    outs.append(be.dense_values(w))
            same(outs[0], outs[2], what=(sr, "scalar d"))
            same(outs[1], outs[3], what=(sr, "scalar s"))


def test_reduce_rows_random(hb, ob):
    n, ptr, ind, val = rand_graph(4000, 30000, 21, hub=5000)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    for mono in ("Plus", "Minimum", "Maximum"):
        outs = []
        for be, A in ((hb, A_h), (ob, A_o)):
            w = be.vector(n)
            assert be.reduce_rows(w, mono, A, be.descriptor(load=False)) == 0
            outs.append(be.dense_values(w))
        same(outs[0], outs[1], what=mono)


def test_matrix_build_matches_oracle_loader(hb, ob):
    """COO build (coo2csr + coo2csc) on unsorted input with the reference's data files."""
    for name in ("chesapeake.mtx", "test_cc.mtx", "test_sgm.mtx", "test_pr.mtx", "small.mtx"):
        A_h, A_o = hb.matrix_from_mtx(name), ob.matrix_from_mtx(name)
        for x, y in zip(A_h.host_csr(), ob.host_csr(A_o)):
            same(x, y, what=name)
        for x, y in zip(A_h.host_csc(), (A_o.cscColPtr, A_o.cscRowInd, A_o.cscVal)):
            same(x, y, what=name)


@pytest.mark.gpu
def test_device_build_and_ingest():
    """build() sorts and compresses on the device: CSR and CSC equal a stable host sort of the
    same coordinate list (duplicates kept, ties in input order).  ingest_device applies the
    loader's options (reverse edges, self loops, duplicates) and must equal the oracle loader's
    result on the same list."""
    import torch
    import graphblast_amd as g
    rng = np.random.default_rng(11)
    for nr, nc, m in ((1, 1, 0), (5, 7, 3), (300, 300, 5000), (70000, 65000, 400000), (1 << 20, 1 << 20, 3000000)):
        r = rng.integers(0, nr, m).astype(np.int32)
        c = rng.integers(0, nc, m).astype(np.int32)
        v = rng.integers(1, 1000, m).astype(np.float32)
        A = g.Matrix(nr, nc)
        assert A.build(r, c, v, m, None) == 0
        order = np.lexsort((c, r))                       # stable: ties keep input order
        ptr = np.zeros(nr + 1, np.int64); np.add.at(ptr, r + 1, 1); ptr = np.cumsum(ptr)
        hp, hi, hv = A.host_csr()
        assert np.array_equal(hp, ptr) and np.array_equal(hi, c[order]) and np.array_equal(hv, v[order])
        order2 = np.lexsort((r, c))
        cp = np.zeros(nc + 1, np.int64); np.add.at(cp, c + 1, 1); cp = np.cumsum(cp)
        tp, ti, tv = A.host_csc()
        assert np.array_equal(tp, cp) and np.array_equal(ti, r[order2])
        # equal (col, row) pairs keep CSR order, which is input order
        assert np.array_equal(tv, v[order2])
    # loader semantics on the device
    dev = torch.device("cuda", 0)
    for n, m, sym in ((50, 400, True), (50, 400, False), (100000, 1500000, True)):
        r = rng.integers(0, n, m).astype(np.int32)
        c = rng.integers(0, n, m).astype(np.int32)
        tr, tc = torch.as_tensor(r).to(dev), torch.as_tensor(c).to(dev)
        A = g.Matrix(n, n)
        assert A.ingest_device(tr.data_ptr(), tc.data_ptr(), None, m, symmetrize=sym, keep=(tr, tc)) == 0
        rr, cc = (np.concatenate([r, c]), np.concatenate([c, r])) if sym else (r, c)
        keep = rr != cc
        key = np.unique(rr[keep].astype(np.int64) * n + cc[keep])
        wr, wc = key // n, key % n
        ptr = np.zeros(n + 1, np.int64); np.add.at(ptr, wr + 1, 1); ptr = np.cumsum(ptr)
        hp, hi, hv = A.host_csr()
        assert np.array_equal(hp, ptr) and np.array_equal(hi, wc) and np.all(hv == 1.0)
        assert A.nvals() == key.size


@pytest.mark.gpu
def test_matrix_ewisemult_scalar_and_vector():
    """Matrix (x) scalar and matrix (x) broadcast vector (operations.hpp:206-267) update both
    orientations in place: C(i,j) = A(i,j) (x) s, A(i,j) (x) B(i), and A(i,j) (x) B(j) with
    GrB_INP1 = GrB_TRAN."""
    import graphblast_amd as g
    from graphblast_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    nr, nc, m = 300, 200, 4000
    r = rng.integers(0, nr, m).astype(np.int32); c = rng.integers(0, nc, m).astype(np.int32)
    key = np.unique(r.astype(np.int64) * nc + c)
    r, c = (key // nc).astype(np.int32), (key % nc).astype(np.int32)
    v = rng.integers(1, 50, r.size).astype(np.float32)
    A = g.Matrix(nr, nc)
    assert A.build(r, c, v, r.size, None) == 0
    d = g.Descriptor(); d.loadArgs()
    from graphblast_amd.api import _semiring_id
    assert lib.grb_matrix_eWiseMult_scalar(A._h, _semiring_id("PlusMultiplies"), A._h, 0.5) == 0
    want = v * np.float32(0.5)
    hp, hi, hv = A.host_csr()
    assert np.array_equal(hv, want)
    tp, ti, tv = A.host_csc()
    order = np.lexsort((r, c))
    assert np.array_equal(tv, want[order])
    # by row
    b = rng.integers(1, 9, nr).astype(np.float32)
    B = g.Vector(nr); assert B.build(b, nr) == 0
    assert lib.grb_matrix_eWiseMult_vector(A._h, _semiring_id("PlusMultiplies"), A._h, B._h, d._h) == 0
    want = want * b[r]
    assert np.array_equal(A.host_csr()[2], want) and np.array_equal(A.host_csc()[2], want[order])
    # by column (GrB_INP1 = GrB_TRAN), division through the semiring's multiply
    bc = rng.integers(1, 9, nc).astype(np.float32)
    Bc = g.Vector(nc); assert Bc.build(bc, nc) == 0
    d.toggle(g.GrB_INP1)
    assert lib.grb_matrix_eWiseMult_vector(A._h, _semiring_id("PlusDivides"), A._h, Bc._h, d._h) == 0
    want = want / bc[c]
    assert np.array_equal(A.host_csr()[2], want) and np.array_equal(A.host_csc()[2], want[order])


def test_vector_resize(hb):
    """Vector::resize (vector.hpp:230-237): the active representation keeps its first
    min(nsize, nvals) entries, size() follows (test/gdensevector.cu:168-186, gsparsevector.cu:188-210)."""
    g = hb.g
    for new in (15, 6, 10):
        v = g.Vector(10, np.int32)
        vals = np.arange(1, 11, dtype=np.int32)
        assert v.build(vals, 10) == 0
        assert v.resize(new) == 0
        assert v.size() == new and v.nvals() == new
        got = v.extractTuples()[1]
        k = min(new, 10)
        assert np.array_equal(got[:k], vals[:k]) and not got[k:].any()
        s = g.Vector(12, np.int32)
        idx = np.array([1, 2, 3, 4, 5, 6, 7, 8, 0, 2], dtype=np.int32)
        assert s.build(idx, idx, 10, None) == 0
        assert s.resize(new) == 0
        assert s.size() == new and s.nvals() == min(new, 10)
        info, gi, gv = s.extractTuples(sparse=True)
        assert info == 0 and np.array_equal(gi, idx[:min(new, 10)]) and np.array_equal(gv, idx[:min(new, 10)])
    u = g.Vector(4)
    assert u.resize(8) == 1                                   # GrB_UNINITIALIZED_OBJECT: no storage yet


def test_trace_mxm_transpose(hb):
    """traceMxmTranspose against the restated kernel (oracle/ops.py), including the Index-typed
    temporary B's value passes through; test/gtrace.cu's two literal cases (trace 91)."""
    from oracle import ops
    from oracle.semiring import Semiring
    g = hb.g
    d = hb.descriptor()
    for rows, cols in (([0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 6, 6]),
                       ([0, 0, 2, 3, 4, 5, 6, 7], [0, 6, 2, 3, 4, 5, 6, 6])):
        A = g.Matrix(8, 8)
        assert A.build(rows, cols, np.array([0, 1, 2, 3, 4, 5, 6, 0], dtype=np.float32), 8, None) == 0
        assert g.traceMxmTranspose("PlusMultiplies", A, A, d) == (0, 91.0)
    rng = np.random.default_rng(5)
    n = 300
    for dt in (np.float32, np.int32):
        mats = []
        for _ in range(2):
            r, c = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
            key = np.unique(r * n + c)
            r, c = (key // n).astype(np.int32), (key % n).astype(np.int32)
            v = (rng.integers(-6, 7, r.size) * (0.5 if dt == np.float32 else 1)).astype(dt)
            A = g.Matrix(n, n, dt)
            assert A.build(r, c, v, r.size, None) == 0
            OA = ops.Matrix(n, n, dt)
            OA.build(r, c, v)
            mats.append((A, OA))
        for srn in ("PlusMultiplies", "MaximumMultiplies", "PlusMinus"):        # identities an Index can hold
            info, got = g.traceMxmTranspose(srn, mats[0][0], mats[1][0], d)
            want = ops.trace_mxm_transpose(Semiring(srn, dt), mats[0][1], mats[1][1])
            assert info == 0 and got == float(want), (dt, srn, got, want)


def test_push_state_survives_other_operations(hb, ob):
    """The push path keeps an n-bit bitmap and an n-entry accumulator clean BETWEEN calls instead of
    clearing them per call; nothing else may write there.  (PageRank's partial sums, the masked SpGEMM's
    row table and the trace result once used the same scratch slots: the first push after any of them
    returned stray vertices.)  Each of them is followed by a key-value and a structure-only push."""
    g = hb.g
    n, ptr, ind, val = rand_graph(3000, 20000, 77)
    A_h, A_o = both(hb, ob, n, ptr, ind, val)
    rng = np.random.default_rng(3)
    fidx = np.sort(rng.choice(n, 40, replace=False)).astype(np.int32)
    fval = rng.integers(1, 4, fidx.size).astype(F)

    def check_push(tag):
        for struc in (0, 1):
            outs = []
            for be, A in ((hb, A_h), (ob, A_o)):
                d = be.descriptor(mxvmode=1, struconly=struc)
                u = be.vector(n); be.build_sparse(u, fidx, fval)
                w = be.vector(n)
                assert be.vxm(w, None, None, "PlusMultiplies", u, A, d) == 0
                i, v = be.sparse_tuples(w)
                outs.append((i, v))
            same(outs[0][0], outs[1][0], what=(tag, struc))
            if not struc:
                same(outs[0][1], outs[1][1], what=(tag, struc))

    check_push("fresh")
    deg = np.maximum(np.diff(ptr), 1).astype(F)
    P = hb.matrix_from_csr(n, ptr, ind, (F(0.85) / deg[np.repeat(np.arange(n), np.diff(ptr))]).astype(F))
    p = g.Vector(n)
    assert g.pr(p, P, 0.85, 0.0, hb.descriptor(mxvmode=2, max_niter=5))[0] == 0
    check_push("after pr")
    sym = rand_graph(3000, 20000, 78)
    from graphblast_amd.graphgen import finalize_edges
    gs = finalize_edges(rng.integers(0, n, 20000), rng.integers(0, n, 20000), n, symmetrize=True)
    Ai = g.Matrix(n, n, np.int32)
    assert Ai.build_csr(gs["csr"][0], gs["csr"][1], np.ones(gs["csr"][1].size, dtype=np.int32)) == 0
    L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
    d = hb.descriptor()
    assert g.tril(L, Ai, d) == 0
    assert g.tc(L, B, d)[0] == 0
    check_push("after tc")
    assert g.traceMxmTranspose("PlusMultiplies", A_h, A_h, d)[0] == 0
    check_push("after trace")


def test_load_mtx_on_device(hb, tmp_path):
    """grb_matrix_load_mtx (MatrixMarket text parsed on the device) == the restated readMtx + coo2csr /
    coo2csc on every data/small file (both `directed` conventions) and on generated pattern / integer /
    real files with ragged whitespace, signs, exponents, a missing final newline and duplicate entries."""
    import os
    from backends import GOLDEN
    from oracle import loader
    g = hb.g

    def check(path, dtype, directed, valued=True):
        A = g.Matrix.from_mtx(path, dtype, directed)
        r, c, v, nr, nc, nv = loader.read_mtx(path, directed, dtype)
        assert (A.nrows(), A.ncols(), A.nvals()) == (nr, nc, nv), path
        ptr, ind, val = loader.coo2csr(r, c, v, nr, nc)
        hp, hi, hv = A.host_csr()
        assert np.array_equal(hp, ptr) and np.array_equal(hi, ind), path
        cp, ci, cv = loader.coo2csc(r, c, v, nr, nc)
        tp, ti, tv = A.host_csc()
        assert np.array_equal(tp, cp) and np.array_equal(ti, ci), path
        # values: where the loader dropped entries of a VALUED file the reference leaves its values array
        # unshifted (util.hpp:311-323; restated in oracle/loader.py) while the device loader keeps each
        # survivor's own value -- compared only when nothing was dropped or every value is 1
        with open(path) as fh:
            code = loader.read_banner(fh.readline())
            line = fh.readline()
            while line.startswith("%"):
                line = fh.readline()
            raw = np.array(fh.read().split()[: (2 if code[2] == "P" else 3) * int(line.split()[2])], dtype=np.float64)
        raw = raw.reshape(-1, 2 if code[2] == "P" else 3)
        undirected = (code[3] == "S" or directed == 2) and directed != 1
        total = raw.shape[0] + (int(np.count_nonzero(raw[:, 0] != raw[:, 1])) if undirected else 0)
        if valued and (code[2] == "P" or total == nv):
            assert np.array_equal(hv, val) and np.array_equal(tv, cv), path

    data = os.path.join(GOLDEN, "data")
    for f in sorted(x for x in os.listdir(data) if x.endswith(".mtx") and not x.startswith(".")):
        for directed in (0, 1, 2):
            check(os.path.join(data, f), np.float32, directed)
    check(os.path.join(data, "chesapeake.mtx"), np.int32, 0)
    rng = np.random.default_rng(8)
    n, m = 5000, 60000
    key = np.unique(rng.integers(0, n, m) * n + rng.integers(0, n, m))       # no duplicates: values stay aligned
    key = key[(key // n) != (key % n)]                                        # no self loops either
    r, c = key // n + 1, key % n + 1
    p = tmp_path / "pattern.mtx"
    with open(p, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate pattern general\n% a comment\n%another\n")
        fh.write("%d %d %d\n" % (n, n, key.size))
        fh.write("\n".join("%d\t  %d " % (a, b) for a, b in zip(r, c)))       # no final newline, mixed blanks
    check(str(p), np.float32, 0)
    check(str(p), np.float32, 2)
    p = tmp_path / "integer.mtx"
    vals = rng.integers(-50, 50, key.size)
    with open(p, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate integer general\n")
        fh.write("%d %d %d\n" % (n, n, key.size))
        fh.write("".join("%d %d %d\n" % t for t in zip(r, c, vals)))
    check(str(p), np.float32, 1)
    check(str(p), np.int32, 1)
    p = tmp_path / "real.mtx"
    fv = np.round(rng.normal(0, 100, key.size), 3)
    with open(p, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate real general\n")
        fh.write("%d %d %d\n" % (n, n, key.size))
        for k, (a, b, x) in enumerate(zip(r, c, fv)):
            fh.write(("%d %d %.3f\n" if k % 3 else "%d %d %.4e\n") % (a, b, x))
    A = g.Matrix.from_mtx(str(p), np.float32, 1)
    rr, cc, vv, nr, nc, nv = loader.read_mtx(str(p), 1, np.float32)
    ptr, ind, val = loader.coo2csr(rr, cc, vv, nr, nc)
    hp, hi, hv = A.host_csr()
    assert np.array_equal(hp, ptr) and np.array_equal(hi, ind)
    np.testing.assert_allclose(hv, val, rtol=1e-6)                           # last-bit differences of decimal parsing
    # duplicates and self loops in a pattern file: dropped as by removeSelfloop
    p = tmp_path / "dups.mtx"
    with open(p, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate pattern symmetric\n6 6 7\n1 1\n2 1\n2 1\n3 2\n6 6\n5 4\n5 4\n")
    check(str(p), np.float32, 0)
    with pytest.raises(RuntimeError):
        g.Matrix.from_mtx(str(tmp_path / "missing.mtx"))
