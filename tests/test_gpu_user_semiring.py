"""GPU suite (-m gpu): semirings an application registers itself (REGISTER_SEMIRING over REGISTER_MONOID,
graphblas/stddef.hpp:140-191; grb_semiring_register in the C ABI).  A composition that is one of the reference's 17
runs their compiled kernels; any other runs the same kernels with run-time operators -- checked here against a numpy
evaluation of the operation's definition (oracle/ops.py's, with the operators swapped in)."""
import numpy as np
import pytest
import torch

from backends import HipBackend

pytestmark = pytest.mark.gpu
F = np.float32

OPS = {
    "plus": lambda a, b: a + b, "minus": lambda a, b: a - b, "multiplies": lambda a, b: a * b,
    "minimum": np.minimum, "maximum": np.maximum, "first": lambda a, b: a + 0 * b, "second": lambda a, b: b + 0 * a,
    "greater": lambda a, b: (a > b).astype(a.dtype), "less": lambda a, b: (a < b).astype(a.dtype),
    "logical_or": lambda a, b: ((a != 0) | (b != 0)).astype(a.dtype),
    "logical_and": lambda a, b: ((a != 0) & (b != 0)).astype(a.dtype),
    "not_equal_to": lambda a, b: (a != b).astype(a.dtype),
}

CASES = [("maximum", 0.0, "plus"),            # tropical (max, +)
         ("maximum", -1000.0, "multiplies"),  # a user monoid with its own identity
         ("minimum", float(np.finfo(np.float32).max), "maximum"),   # (min, max): bottleneck paths
         ("plus", 0.0, "minimum"),
         ("logical_or", 0.0, "greater")]


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


def pull_ref(add, ident, mul, ptr, ind, val, u, dtype):
    n = ptr.size - 1
    w = np.full(n, ident, dtype=dtype)
    for i in range(n):
        acc = np.array(ident, dtype=dtype)
        for p in range(ptr[i], ptr[i + 1]):
            acc = OPS[add](acc, OPS[mul](np.array(val[p], dtype=dtype), np.array(u[ind[p]], dtype=dtype))).astype(dtype)
        w[i] = acc
    return w


def test_registered_semirings_in_every_operation(hb):
    from graphblast_amd.graphgen import finalize_edges
    g = hb.g
    rng = np.random.default_rng(21)
    n, m = 400, 3000
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    src[:600] = 9                                      # a long row: the sliced path of the SpMV / SpMM kernels
    gr = finalize_edges(src, dst, n, symmetrize=False)
    ptr, ind = gr["csr"]
    cptr, cind = gr["csc"]
    for dtype in (np.float32, np.int32):
        val = rng.integers(1, 6, ind.size).astype(dtype)
        A = g.Matrix(n, n, dtype)
        assert A.build_csr(ptr, ind, val) == 0
        _, _, cval = A.host_csc()
        for add, ident, mul in CASES:
            sid = g.register_semiring(add, ident, mul)
            assert sid >= 64 and g.register_semiring(add, ident, mul) == sid
            idv = dtype(ident) if dtype == np.float32 else dtype(min(max(ident, -2**31), 2**31 - 1))
            x = rng.integers(0, 7, n).astype(dtype)
            # ---- mxv, dense input (pull: the SpMV kernel)
            u, w = g.Vector(n, dtype), g.Vector(n, dtype)
            assert u.build(x, n) == 0
            d = hb.descriptor(mxvmode=2)
            assert g.mxv(w, None, None, sid, A, u, d) == 0
            assert np.array_equal(hb.dense_values(w), pull_ref(add, idv, mul, ptr, ind, val, x, dtype)), (add, mul, dtype)
            # ---- vxm, sparse input (push: SpMSpV with the monoid's atomic combine), against the pull result on A^T
            idx = np.sort(rng.choice(n, 40, replace=False)).astype(np.int32)
            xs = rng.integers(1, 7, idx.size).astype(dtype)
            if add in ("plus", "maximum", "minimum", "logical_or"):
                us, ws = g.Vector(n, dtype), g.Vector(n, dtype)
                assert us.build(idx, xs, idx.size, None) == 0
                dp = hb.descriptor(mxvmode=1)
                assert g.vxm(ws, None, None, sid, us, A, dp) == 0 and dp.lastmxv_ == g.GrB_PUSHONLY
                got_i, got_v = hb.sparse_tuples(ws)
                # definition of the push product: over the frontier's out-edges, products == identity are skipped
                acc = {}
                for k_, i in enumerate(idx):
                    for p in range(ptr[i], ptr[i + 1]):
                        a_, x_ = dtype(val[p]), dtype(xs[k_])
                        prod = idv if (a_ == idv or x_ == idv) else OPS[mul](np.array(a_), np.array(x_)).astype(dtype)
                        j = int(ind[p])
                        acc[j] = OPS[add](np.array(acc[j]), np.array(prod)).astype(dtype) if j in acc else prod
                want_i = np.array(sorted(acc), dtype=np.int32)
                assert np.array_equal(got_i, want_i), (add, mul, dtype)
                assert np.array_equal(got_v, np.array([acc[j] for j in want_i], dtype=dtype)), (add, mul, dtype)
            # ---- eWiseAdd (the semiring's add) / eWiseMult (its multiply), dense operands
            y = rng.integers(0, 7, n).astype(dtype)
            a_, b_, c_ = g.Vector(n, dtype), g.Vector(n, dtype), g.Vector(n, dtype)
            assert a_.build(x, n) == 0 and b_.build(y, n) == 0
            assert g.eWiseAdd(c_, None, None, sid, a_, b_, d) == 0
            assert np.array_equal(hb.dense_values(c_), OPS[add](x, y).astype(dtype)), (add, dtype)
            assert g.eWiseMult(c_, None, None, sid, a_, b_, d) == 0
            wantm = np.where((x == idv) | (y == idv), idv, OPS[mul](x, y)).astype(dtype)   # kernels/ewisemult.hpp:22-25
            assert np.array_equal(hb.dense_values(c_), wantm), (mul, dtype)
            # ---- sparse x dense mxm, 5 right-hand sides
            B = rng.integers(0, 7, (n, 5)).astype(dtype)
            dev = torch.device("cuda", 0)
            tB = torch.as_tensor(B).to(dev).contiguous()
            tC = torch.empty((n, 5), dtype=tB.dtype, device=dev)
            torch.cuda.synchronize()
            assert g.spmm(sid, A, tB.data_ptr(), tC.data_ptr(), 5) == 0
            torch.cuda.synchronize()
            got = tC.cpu().numpy()
            for c in range(5):
                assert np.array_equal(got[:, c], pull_ref(add, idv, mul, ptr, ind, val, B[:, c], dtype)), (add, mul, c)


def test_registered_composition_of_a_builtin_uses_its_kernels(hb):
    """(plus, 0, multiplies) registered by hand == PlusMultiplies, bit for bit on random floats; and an interleaving of
    a registered and a built-in semiring does not leak operator state from one launch into the next."""
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    g = hb.g
    s, d, n = rmat_edges(13, 16, seed=4)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    rng = np.random.default_rng(5)
    val = rng.random(ind.size).astype(F)
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, val) == 0
    x = rng.random(n).astype(F)
    u = g.Vector(n)
    assert u.build(x, n) == 0
    d2 = hb.descriptor(mxvmode=2)
    sid = g.register_semiring("plus", 0.0, "multiplies")
    maxplus = g.register_semiring("maximum", 0.0, "plus")
    outs = []
    for op in ("PlusMultiplies", sid, maxplus, "PlusMultiplies", maxplus, sid):
        w = g.Vector(n)
        assert g.mxv(w, None, None, op, A, u, d2) == 0
        outs.append(hb.dense_values(w))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[3]) and np.array_equal(outs[0], outs[5])
    assert np.array_equal(outs[2], outs[4]) and not np.array_equal(outs[0], outs[2])
    rows = np.repeat(np.arange(n), np.diff(ptr))
    want = np.zeros(n, F)
    np.maximum.at(want, rows, val + x[ind])
    assert np.allclose(outs[2], want, rtol=1e-6)


# ---- apply on the device (grb_vector_apply / grb_matrix_apply) -----------------------------------
def _unary_ref(name, x, binop=None, scalar=0):
    s = np.array(scalar, dtype=x.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        if name == "identity": return x.copy()
        if name == "ainv": return (0 - x).astype(x.dtype)
        if name == "minv":
            if x.dtype == np.int32:
                return np.where(x == 0, 0, (1 / np.where(x == 0, 1, x)).astype(np.int64)).astype(np.int32)   # C: 1 / x truncated
            return (np.float32(1) / x).astype(x.dtype)
        if name == "abs": return np.abs(x)
        if name == "lnot": return (x == 0).astype(x.dtype)
        a, b = (np.broadcast_to(s, x.shape), x) if name == "bind_first" else (x, np.broadcast_to(s, x.shape))
        if binop == "divides":
            if x.dtype == np.int32:
                q = np.where(b == 0, 0, (a.astype(np.int64) / np.where(b == 0, 1, b)).astype(np.int64))   # truncation, /0 -> 0
                return q.astype(np.int32)
            return (a / b).astype(x.dtype)
        table = dict(OPS, equal=lambda p, q: (p == q).astype(p.dtype), greater_equal=lambda p, q: (p >= q).astype(p.dtype),
                     less_equal=lambda p, q: (p <= q).astype(p.dtype),
                     logical_xor=lambda p, q: ((p != 0) != (q != 0)).astype(p.dtype))
        return np.asarray(table[binop](a.copy(), b.copy())).astype(x.dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.int32])
def test_apply_on_the_device_vectors_and_matrix_values(hb, dtype):
    """every unary operator of the C ABI, every binary operator bound to a scalar on either side, dense and sparse
    vectors (in place too), and the stored values of a matrix (both orientations; the next product sees them) --
    against the reference's host loop w[i] = op(u[i]) (backend/cuda/apply.hpp:39-41, :106-108) in numpy"""
    import graphblast_amd as g
    from graphblast_amd.api import UNARY_OPS, BINARY_OPS
    rng = np.random.default_rng(5)
    n = 5000
    x = rng.integers(-6, 7, n).astype(dtype)
    d = g.Descriptor(); d.loadArgs()
    cases = [(k, None, 0) for k in UNARY_OPS[:5]]
    cases += [(k, b, s) for k in ("bind_first", "bind_second") for b in BINARY_OPS for s in (3, 0, -2)]
    for unary, binop, scalar in cases:
        u, w = g.Vector(n, dtype), g.Vector(n, dtype)
        assert u.build(x) == 0
        assert g.apply(w, None, None, unary, u, d, binop=binop, scalar=scalar) == 0
        want = _unary_ref(unary, x, binop, scalar)
        got = w.extractTuples()[1]
        assert np.array_equal(got, want, equal_nan=True), (unary, binop, scalar)
        assert np.array_equal(u.extractTuples()[1], x)                      # the input is untouched
    # in place, and a sparse vector: only the stored values change, the indices stay
    u = g.Vector(n, dtype)
    assert u.build(x) == 0
    assert g.apply(u, None, None, "bind_second", u, d, binop="multiplies", scalar=3) == 0
    assert np.array_equal(u.extractTuples()[1], (x * 3).astype(dtype))
    idx = np.sort(rng.choice(n, 700, replace=False)).astype(np.int32)
    sv = rng.integers(1, 9, idx.size).astype(dtype)
    us, ws = g.Vector(n, dtype), g.Vector(n, dtype)
    assert us.build(idx, sv, idx.size, None) == 0
    assert g.apply(ws, None, None, "ainv", us, d) == 0
    info, gi, gv = ws.extractTuples(sparse=True)
    assert info == 0 and np.array_equal(gi, idx) and np.array_equal(gv, (0 - sv).astype(dtype))
    # a mask is not implemented, as in the reference
    assert g.apply(ws, us, None, "ainv", us, d) == 9                        # GrB_NOT_IMPLEMENTED
    # matrix values: both orientations, then a product
    from graphblast_amd.graphgen import finalize_edges
    src, dst = rng.integers(0, 300, 4000), rng.integers(0, 300, 4000)
    gr = finalize_edges(src, dst, 300, symmetrize=False)
    ptr, ind = gr["csr"]
    val = rng.integers(1, 6, ind.size).astype(dtype)
    A = g.Matrix(300, 300, dtype)
    assert A.build_csr(ptr, ind, val) == 0
    csc_before = A.host_csc()[2].copy()
    assert g.apply(A, None, None, "bind_second", A, d, binop="plus", scalar=10) == 0
    assert np.array_equal(A.host_csr()[2], (val + 10).astype(dtype))
    assert np.array_equal(A.host_csc()[2], (csc_before + 10).astype(dtype))
    xv = rng.integers(0, 3, 300).astype(dtype)
    vu, vw = g.Vector(300, dtype), g.Vector(300, dtype)
    assert vu.build(xv) == 0
    d2 = g.Descriptor(); d2.loadArgs(mxvmode=2)
    assert g.mxv(vw, None, None, "PlusMultiplies", A, vu, d2) == 0
    rows = np.repeat(np.arange(300), np.diff(ptr))
    want = np.bincount(rows, weights=(val + 10).astype(np.float64) * xv[ind], minlength=300).astype(dtype)
    assert np.array_equal(vw.extractTuples()[1], want)


def test_apply_through_the_cpp_frontend(tmp_path):
    """graphblas::apply in the drop-in header: the device operators run on the GPU under any GrB_BACKEND, a stateful
    host functor keeps the reference's host loop in index order (tests/tools/apply_device.cpp)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "apply_device")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "tools", "apply_device.cpp"),
                           "-L" + os.path.join(root, "graphblast_amd"), "-lgrb_hip",
                           "-Wl,-rpath," + os.path.join(root, "graphblast_amd"), "-o", out])
    lines = [[float(t) for t in ln.split()] for ln in subprocess.check_output([out]).decode().strip().split("\n")]
    u = np.arange(10, dtype=np.float32) - 4
    assert lines[0] == list(u * 2.5)
    assert lines[1] == list(np.abs(u))
    u2 = 10 - u
    assert lines[2] == list(u2)
    assert lines[3] == list(u2 + np.arange(10))                               # the k-th call adds k: index order
    assert lines[4] == [float(i + 1 + 100) for i in range(10)]                # row i holds one entry, value i + 1 + 100
