"""GPU suite (-m gpu): grb_spmm -- mxm with a dense right-hand side, the product the reference declares and
leaves a stub (backend/cuda/operations.hpp:52-70, spmm.hpp:15-27).  Checked column by column against the
oracle's SpMV definition (oracle/ops.py follows backend/cuda/spmv.hpp), all 17 semirings, and against the
library's own mxv column by column."""
import os

import numpy as np
import pytest
import torch

from backends import HipBackend, OracleBackend

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


def dense_ref(sr, ptr, ind, val, B, dtype):
    """w[i, c] = (+)_p mul(val[p], B[ind[p], c]) in CSR order, identity for empty rows."""
    n = ptr.size - 1
    k = B.shape[1]
    out = np.empty((n, k), dtype=dtype)
    for i in range(n):
        acc = np.full(k, sr.identity(), dtype=dtype)
        for p in range(ptr[i], ptr[i + 1]):
            acc = sr.add_op(acc, sr.mul_op(np.full(k, val[p], dtype=dtype), B[ind[p]])).astype(dtype)
        out[i] = acc
    return out


def run_spmm(g, A, op, B, nrows, tran=False):
    dev = torch.device("cuda", 0)
    tB = torch.as_tensor(B).to(dev).contiguous()
    tC = torch.empty((nrows, B.shape[1]), dtype=tB.dtype, device=dev)
    torch.cuda.synchronize()
    assert g.spmm(op, A, tB.data_ptr(), tC.data_ptr(), B.shape[1], None, tran=tran) == 0
    torch.cuda.synchronize()
    return tC.cpu().numpy()


def test_spmm_all_semirings_small(hb):
    from oracle.semiring import Semiring, SEMIRINGS
    from graphblast_amd.graphgen import finalize_edges
    g = hb.g
    rng = np.random.default_rng(3)
    n, m = 300, 2500
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    src[:700] = 5                                       # one long row (> 512 entries): the sliced path
    dst[:700] = rng.permutation(n)[:300].tolist() + rng.integers(0, n, 400).tolist()
    gr = finalize_edges(src, dst, n, symmetrize=False)
    ptr, ind = gr["csr"]
    assert np.diff(ptr).max() > 200
    for dtype in (np.float32, np.int32):
        val = rng.integers(1, 4, ind.size).astype(dtype)
        A = g.Matrix(n, n, dtype)
        assert A.build_csr(ptr, ind, val) == 0
        for k in (1, 5, 16, 64, 70):
            B = rng.integers(0, 3, (n, k)).astype(dtype)
            for name in SEMIRINGS:
                sr = Semiring(name, dtype)
                want = dense_ref(sr, ptr, ind, val, B, dtype)
                got = run_spmm(g, A, name, B, n)
                assert np.array_equal(got, want), (name, dtype, k)
                if k in (5, 64):
                    cp, ci, cv = A.host_csc()
                    want_t = dense_ref(sr, cp, ci, cv, B, dtype)
                    got_t = run_spmm(g, A, name, B, n, tran=True)
                    assert np.array_equal(got_t, want_t), (name, dtype, k, "tran")


def test_spmm_columns_equal_spmv(hb):
    """Every column of the SpMM result equals the library's own mxv on that column (RMAT-14, float values):
    bit-exact -- both form the row sums in CSR order."""
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    g = hb.g
    s, d, n = rmat_edges(14, 16, seed=8)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    rng = np.random.default_rng(9)
    val = rng.random(ind.size).astype(F)
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, val) == 0
    k = 16
    B = rng.random((n, k)).astype(F)
    for op in ("PlusMultiplies", "MinimumPlus", "MaximumMultiplies"):
        got = run_spmm(g, A, op, B, n)
        desc = hb.descriptor(mxvmode=2)
        for c in (0, 7, 15):
            u, w = g.Vector(n), g.Vector(n)
            assert u.build(np.ascontiguousarray(B[:, c]), n) == 0
            assert g.mxv(w, None, None, op, A, u, desc) == 0
            col = hb.dense_values(w)
            if op == "PlusMultiplies":
                assert np.allclose(got[:, c], col, rtol=1e-5, atol=0), (op, c)     # mxv sums long rows slice-wise too
            else:
                assert np.array_equal(got[:, c], col), (op, c)
