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


# ---- MIS / graph colouring / LGC / diameter ---------------------------
def _mis_inner(v, w, f, m, A, desc):
    """algorithm::misInner (graphblas/algorithm/mis.hpp:22-111).  v <- 1 on the members of a
    maximal independent set among the candidates (nonzeros of the weight vector w); w, f, m
    are clobbered exactly as there."""
    I = np.int32
    n = A.nrows_
    v.fill(0)
    maxmul = Semiring("MaximumMultiplies", I)
    gplus = Semiring("GreaterPlus", I)
    lor = Semiring("LogicalOrAnd", I)
    plus = Monoid("Plus", I)
    rounds = 0
    while True:
        rounds += 1
        ops.vxm(m, w, None, maxmul, w, A, desc)              # max weight among candidate neighbours
        ops.eWiseAdd(f, None, None, gplus, w, m, desc)       # local maxima
        ops.assign(v, f, None, I(1), desc)
        ops.assign(w, f, None, I(0), desc)
        succ = ops.reduce_vector(plus, f, desc)
        if succ == 0:
            break
        ops.vxm(m, w, None, lor, f, A, desc)                 # neighbours of the new members
        ops.assign(w, m, None, I(0), desc)
        if rounds > n + 1:
            raise RuntimeError("misInner did not terminate")
    return rounds


def mis(A, weights, desc):
    """algorithm::mis (mis.hpp:114-141) with the weight vector given (the reference draws it
    on the host: apply(set_random) under GrB_SEQUENTIAL = srand(seed), rand() per element,
    algorithm/common.hpp:8-20)."""
    I = np.int32
    n = A.nrows_
    v, w, f, m = (ops.Vector(n, I) for _ in range(4))
    v.fill(0)
    w.build_dense(np.asarray(weights, dtype=I))
    rounds = _mis_inner(v, w, f, m, A, desc)
    return v.extractTuples_dense(), rounds


def gc_is(A, weights, desc):
    """algorithm::gcIS (gc.hpp:43-149): one colour per round of local maxima."""
    I = np.int32
    n = A.nrows_
    v, f, w, m = (ops.Vector(n, I) for _ in range(4))
    v.fill(0)
    w.build_dense(np.asarray(weights, dtype=I))
    maxmul = Semiring("MaximumMultiplies", I)
    gplus = Semiring("GreaterPlus", I)
    plus = Monoid("Plus", I)
    it = 1
    while True:
        ops.vxm(m, None, None, maxmul, w, A, desc)
        ops.eWiseAdd(f, None, None, gplus, w, m, desc)
        succ = ops.reduce_vector(plus, f, desc)
        if succ == 0:
            break
        ops.assign(v, f, None, I(it), desc)
        ops.assign(w, f, None, I(0), desc)
        it += 1
        if it > desc.max_niter_:
            break
    return v.extractTuples_dense(), it


def gc_mis(A, weights, desc):
    """algorithm::gcMIS (gc.hpp:152-255): one colour per maximal independent set."""
    I = np.int32
    n = A.nrows_
    v, f, w, temp_w, m, nn = (ops.Vector(n, I) for _ in range(6))
    v.fill(0)
    w.build_dense(np.asarray(weights, dtype=I))
    plus = Monoid("Plus", I)
    it = 1
    while True:
        temp_w.dup(w)
        _mis_inner(f, temp_w, nn, m, A, desc)
        succ = ops.reduce_vector(plus, f, desc)
        if succ == 0:
            break
        ops.assign(v, f, None, I(it), desc)
        ops.assign(w, f, None, I(0), desc)
        it += 1
        if it > desc.max_niter_:
            break
    return v.extractTuples_dense(), it


def gc_jp(A, weights, max_colors, desc):
    """algorithm::gcJP (gc.hpp:258-421): Jones-Plassmann -- each round's local maxima all take
    the smallest colour unused in the union of their coloured neighbourhoods."""
    I = np.int32
    n = A.nrows_
    v, f, w, m, nn = (ops.Vector(n, I) for _ in range(5))
    d, ascending, min_array = (ops.Vector(max_colors, I) for _ in range(3))
    v.fill(0)
    w.build_dense(np.asarray(weights, dtype=I))
    ascending.fillAscending()
    maxmul = Semiring("MaximumMultiplies", I)
    gplus = Semiring("GreaterPlus", I)
    lor = Semiring("LogicalOrAnd", I)
    pmul = Semiring("PlusMultiplies", I)
    minplus = Semiring("MinimumPlus", I)
    plus = Monoid("Plus", I)
    mn = Monoid("Minimum", I)
    it = 1
    while True:
        ops.vxm(m, w, None, maxmul, w, A, desc)
        ops.eWiseAdd(f, None, None, gplus, w, m, desc)
        succ = ops.reduce_vector(plus, f, desc)
        if succ == 0:
            break
        ops.vxm(m, v, None, lor, f, A, desc)                 # coloured neighbours of the frontier
        ops.eWiseMult(nn, None, None, pmul, m, v, desc)      # their colours
        d.fill(0)
        ops.scatter(d, None, nn, I(max_colors), desc)
        ops.eWiseMult(min_array, None, None, minplus, d, ascending, desc)
        min_array.setElement(I(max_colors), 0)
        min_color = ops.reduce_vector(mn, min_array, desc)
        ops.assign(v, f, None, I(min_color), desc)
        ops.assign(w, f, None, I(0), desc)
        it += 1
        if it > desc.max_niter_:
            break
    return v.extractTuples_dense(), it


def lgc(A, s, alpha, eps, desc):
    """algorithm::lgc (graphblas/algorithm/lgc.hpp:14-176): local graph clustering by
    approximate personalised PageRank pushes; float vectors, the constants enter as the
    float images of the doubles (Vector<float>::fill(double))."""
    F = np.float32
    n = A.nrows_
    p, degrees, r, r2, eps_vector, degrees_eps, f, alpha_vector, alpha_vector2 = (ops.Vector(n, F) for _ in range(9))
    ops.reduce_matrix_rows(degrees, Monoid("Plus", F), A, desc)
    p.fill(0.0)
    r2.fill(0.0)
    if desc.get(ops.GrB_MXVMODE) == ops.GrB_PULLONLY:
        r.fill(0.0)
        r.setElement(1.0, s)
    else:
        r.build_sparse([s], [1.0])
    pm = Semiring("PlusMultiplies", F)
    pdiv = Semiring("PlusDivides", F)
    pgt = Semiring("PlusGreater", F)
    plus = Monoid("Plus", F)
    eps_vector.fill(F(eps))
    ops.eWiseMult(degrees_eps, None, None, pm, degrees, eps_vector, desc)
    f.build_sparse([s], [1.0])
    alpha_vector.fill(F(alpha))
    alpha_vector2.fill(F((1.0 - alpha) / 2.0))
    it = 1
    trace = []
    while True:
        desc.toggle(ops.GrB_MASK)
        ops.eWiseMult(r2, f, None, pm, r, alpha_vector, desc)
        desc.toggle(ops.GrB_MASK)
        ops.eWiseAdd(p, None, None, pm, p, r2, desc)
        ops.eWiseMult(r, f, None, pm, r, alpha_vector2, desc)
        desc.toggle(ops.GrB_MASK)
        ops.eWiseMult(r2, f, None, pdiv, r, degrees, desc)
        desc.toggle(ops.GrB_MASK)
        ops.mxv(r, None, True, pm, A, r2, desc)              # accum present: r = r + A r2
        ops.eWiseMult(f, None, None, pgt, r, degrees_eps, desc)
        succ = ops.reduce_vector(plus, f, desc)
        trace.append(float(succ))
        it += 1
        if it > desc.max_niter_:
            break
        if not succ > 0:
            break
    return p.extractTuples_dense(), trace


def diameter(A, s_start, s_end, desc):
    """algorithm::diameter (graphblas/algorithm/diameter.hpp:14-59): BFS eccentricity of the
    sources s_start..s_end-1; returns (max eccentricity, the last source attaining it)."""
    F = np.float32
    n = A.nrows_
    v, q1, q2 = (ops.Vector(n, F) for _ in range(3))
    lor = Semiring("LogicalOrAnd", F)
    plus = Monoid("Plus", F)
    dmax, dind = 0, -1
    for s in range(s_start, s_end):
        v.fill(0.0)
        q1.build_sparse([s], [1.0])
        it = 1
        while True:
            ops.assign(v, q1, None, F(it), desc)
            desc.toggle(ops.GrB_MASK)
            ops.vxm(q2, v, None, lor, q1, A, desc)
            desc.toggle(ops.GrB_MASK)
            q2.swap(q1)
            succ = ops.reduce_vector(plus, q1, desc)
            it += 1
            if not succ > 0:
                break
        dmax = max(dmax, it - 2)
        if it - 2 == dmax:
            dind = s
    return dmax, dind
