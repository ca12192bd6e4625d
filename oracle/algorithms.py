"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's algorithm drivers restated over oracle/ops.py, call for call:
  bfs   <- graphblas/algorithm/bfs.hpp:14-89
  sssp  <- graphblas/algorithm/sssp.hpp:15-103
  pr    <- graphblas/algorithm/pr.hpp:15-94
Each returns the result vector plus a per-iteration trace (direction taken by vxm,
frontier size) so the GPU path's decisions can be compared, not only its output.
"""
import numpy as np

from . import ops
from .semiring import Semiring, Monoid

FLT_MAX = np.finfo(np.float32).max


def bfs(A, s, desc):
    n = A.nrows_
    v = ops.Vector(n)
    v.fill(0.0)
    f1, f2 = ops.Vector(n), ops.Vector(n)
    if desc.get(ops.GrB_MXVMODE) == ops.GrB_PULLONLY:
        f1.fill(0.0)
        f1.setElement(1.0, s)
    else:
        f1.build_sparse([s], [1.0])
    sr = Semiring("LogicalOrAnd", np.float32)
    plus = Monoid("Plus", np.float32)
    trace = []
    it = 1
    while it <= desc.max_niter_:
        ops.assign(v, f1, None, np.float32(it), desc)
        desc.toggle(ops.GrB_MASK)
        ops.vxm(f2, v, None, sr, f1, A, desc)
        desc.toggle(ops.GrB_MASK)
        f2.swap(f1)
        succ = ops.reduce_vector(plus, f1, desc)
        trace.append(("push" if desc.lastmxv_ == ops.GrB_PUSHONLY else "pull", float(succ)))
        if succ == 0:
            break
        it += 1
    return v.extractTuples_dense(), trace


def sssp(A, s, desc):
    n = A.nrows_
    v = ops.Vector(n)
    v.fill(FLT_MAX)
    v.setElement(0.0, s)
    f1, f2, m = ops.Vector(n), ops.Vector(n), ops.Vector(n)
    if desc.get(ops.GrB_MXVMODE) == ops.GrB_PULLONLY:
        f1.fill(FLT_MAX)
        f1.setElement(0.0, s)
    else:
        f1.build_sparse([s], [0.0])
    minplus = Semiring("MinimumPlus", np.float32)
    lessplus = Semiring("CustomLessPlus", np.float32)
    plus = Monoid("Plus", np.float32)
    trace = []
    it = 1
    while it <= desc.max_niter_:
        ops.vxm(f2, None, None, minplus, f1, A, desc)
        ops.eWiseAdd(m, None, None, lessplus, f2, v, desc)
        ops.eWiseAdd(v, None, None, minplus, v, f2, desc)
        desc.toggle(ops.GrB_MASK)
        ops.assign(f2, m, None, FLT_MAX, desc)
        desc.toggle(ops.GrB_MASK)
        f2.swap(f1)
        f1_nvals = f1.nvals()
        succ = ops.reduce_vector(plus, m, desc)
        trace.append(("push" if desc.lastmxv_ == ops.GrB_PUSHONLY else "pull", int(f1_nvals), float(succ)))
        if f1_nvals == 0 or succ == 0:
            break
        it += 1
    return v.extractTuples_dense(), trace


def pr_setup(A, alpha):
    """example/gpr.cu:67-90: values = 1, A = A * alpha, A = A / outdeg (row-wise)."""
    outdeg = np.diff(A.csrRowPtr).astype(np.float32)
    rows_csr = np.repeat(np.arange(A.nrows_), np.diff(A.csrRowPtr))
    A.csrVal = (np.ones(A.nvals_, dtype=np.float32) * np.float32(alpha)) / outdeg[rows_csr]
    A.cscVal = (np.ones(A.nvals_, dtype=np.float32) * np.float32(alpha)) / outdeg[A.cscRowInd]
    return A


def pr(A, alpha, eps, desc):
    n = A.nrows_
    p = ops.Vector(n)
    p.clear()
    p.fill(np.float32(1.0) / np.float32(n))
    p_prev, p_swap, r, r_temp = ops.Vector(n), ops.Vector(n), ops.Vector(n), ops.Vector(n)
    r.fill(1.0)
    pm = Semiring("PlusMultiplies", np.float32)
    pminus = Semiring("PlusMinus", np.float32)
    mm = Semiring("MultipliesMultiplies", np.float32)
    plus = Monoid("Plus", np.float32)
    error = np.float32(1.0)
    it = 1
    trace = []
    while error > eps and it <= desc.max_niter_:
        p_prev.dup(p)
        ops.vxm(p_swap, None, None, pm, p_prev, A, desc)
        ops.eWiseAdd_scalar(p, None, None, pm, p_swap, (np.float32(1.0) - np.float32(alpha)) / np.float32(n), desc)
        ops.eWiseMult(r, None, None, pminus, p, p_prev, desc)
        ops.eWiseAdd(r_temp, None, None, mm, r, r, desc)
        error = np.sqrt(np.float32(ops.reduce_vector(plus, r_temp, desc)))
        trace.append(float(error))
        it += 1
    return p.extractTuples_dense(), trace


def cc(A, desc):
    """algorithm::cc (graphblas/algorithm/cc.hpp:17-136), FastSV on int vectors."""
    n = A.nrows_
    I = np.int32
    mk = lambda: ops.Vector(n, I)
    diff, parent, parent_temp, grandparent, grandparent_temp, mnp, mnp_temp = (mk() for _ in range(7))
    parent.fillAscending()
    for v in (mnp, mnp_temp, grandparent, grandparent_temp):
        v.dup(parent)
    mss = Semiring("MinimumSelectSecond", I)
    mp = Semiring("MinimumPlus", I)
    mne = Semiring("MinimumNotEqualTo", I)
    plus = Monoid("Plus", I)
    it = 1
    while it <= desc.max_niter_:
        parent_temp.dup(parent)
        ops.mxv(mnp_temp, None, None, mss, A, grandparent, desc)
        ops.eWiseAdd(mnp, None, None, mss, mnp, mnp_temp, desc)
        ops.assignScatter(parent, None, None, mnp, parent_temp, desc)
        ops.eWiseAdd(parent, None, None, mp, parent, mnp, desc)
        ops.eWiseAdd(parent, None, None, mp, parent, parent_temp, desc)
        ops.extractGather(grandparent, None, None, parent, parent, desc)
        ops.eWiseMult(diff, None, None, mne, grandparent_temp, grandparent, desc)
        succ = ops.reduce_vector(plus, diff, desc)
        if succ == 0:
            break
        grandparent_temp.dup(grandparent)
        desc.toggle(ops.GrB_MASK)
        ops.assign(grandparent, diff, None, np.iinfo(I).max, desc)
        desc.toggle(ops.GrB_MASK)
        it += 1
    return parent.extractTuples_dense(), it


def tc(L, desc):
    """algorithm::tc (graphblas/algorithm/tc.hpp:15-54) on L = tril(A)."""
    I = np.int32
    desc.toggle(ops.GrB_INP1)
    vals = ops.mxm_masked(L, Semiring("PlusMultiplies", I), L, L, desc)
    return int(ops.reduce_matrix(Monoid("Plus", I), vals, L.nvals_, desc))
