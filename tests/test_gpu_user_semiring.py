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
