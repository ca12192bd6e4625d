"""ORACLE -- TEST INFRASTRUCTURE ONLY. ctypes wrapper over oracle/liboracle.so
(oracle/simple_reference.c)."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "simple_reference.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-shared", "-fPIC",
                               "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for name in ("oracle_bfs", "oracle_sssp", "oracle_pr", "oracle_cc", "oracle_tc", "oracle_mis", "oracle_gc",
                     "oracle_lgc"):
            getattr(_LIB, name).restype = ctypes.c_double
        _LIB.oracle_cc_verify.restype = ctypes.c_int
        _LIB.oracle_mis_verify.restype = ctypes.c_int
        _LIB.oracle_gc_verify.restype = ctypes.c_int
        _LIB.oracle_bfs_do_stats.restype = ctypes.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def bfs(row_ptr, col_ind, src, stop=10000):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    n = rp.size - 1
    depth = np.zeros(n, dtype=np.float32)
    sd = ctypes.c_int(0)
    ms = lib().oracle_bfs(n, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(depth, ctypes.c_float),
                          int(src), int(stop), ctypes.byref(sd))
    return depth, sd.value, ms


def sssp(row_ptr, col_ind, val, src):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    v = np.ascontiguousarray(val, dtype=np.float32)
    n = rp.size - 1
    dist = np.zeros(n, dtype=np.float32)
    sd = ctypes.c_int(0)
    ms = lib().oracle_sssp(n, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(v, ctypes.c_float),
                           _p(dist, ctypes.c_float), int(src), ctypes.byref(sd))
    return dist, sd.value, ms


def pr(row_ptr, col_ind, alpha=0.85, eps=1e-8, max_niter=10):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    n = rp.size - 1
    rank = np.zeros(n, dtype=np.float32)
    it = ctypes.c_int(0)
    res = ctypes.c_float(0)
    ms = lib().oracle_pr(n, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(rank, ctypes.c_float),
                         ctypes.c_float(alpha), ctypes.c_float(eps), int(max_niter),
                         ctypes.byref(it), ctypes.byref(res))
    return rank, it.value, res.value, ms


def cc(row_ptr, col_ind):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    n = rp.size - 1
    label = np.zeros(n, dtype=np.int32)
    nc = ctypes.c_int(0)
    ms = lib().oracle_cc(n, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(label, ctypes.c_int),
                         ctypes.byref(nc))
    return label, nc.value, ms


def cc_verify(row_ptr, col_ind, label):
    rp, ci, lb = _i32(row_ptr), _i32(col_ind), _i32(label)
    nd = ctypes.c_int(0)
    err = lib().oracle_cc_verify(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int),
                                 _p(lb, ctypes.c_int), ctypes.byref(nd))
    return err, nd.value


def cc_canonical(label):
    """Canonicalise a component labelling to 'min vertex id in component'."""
    label = np.asarray(label)
    first = {}
    out = np.empty(label.size, dtype=np.int32)
    order = np.argsort(label, kind="stable")
    lab_sorted = label[order]
    starts = np.r_[0, np.nonzero(np.diff(lab_sorted))[0] + 1]
    mins = np.minimum.reduceat(order, starts)
    out[order] = np.repeat(mins, np.diff(np.r_[starts, label.size]))
    return out


def tc(row_ptr, col_ind, nrows=None):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    nt = ctypes.c_longlong(0)
    ms = lib().oracle_tc(rp.size - 1 if nrows is None else int(nrows), _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), ctypes.byref(nt))
    return nt.value, ms


def mis(row_ptr, col_ind, order):
    """SimpleReferenceMis over the given vertex order (the reference shuffles with mt19937)."""
    rp, ci, od = _i32(row_ptr), _i32(col_ind), _i32(order)
    out = np.zeros(rp.size - 1, dtype=np.int32)
    ms = lib().oracle_mis(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(od, ctypes.c_int),
                          _p(out, ctypes.c_int))
    return out, ms


def mis_verify(row_ptr, col_ind, mis_vec):
    """SimpleVerifyMis: (number of errors, set size); 0 errors is the reference's CORRECT."""
    rp, ci, mv = _i32(row_ptr), _i32(col_ind), _i32(mis_vec)
    size = ctypes.c_int(0)
    err = lib().oracle_mis_verify(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(mv, ctypes.c_int),
                                  ctypes.byref(size))
    return err, size.value


def gc(row_ptr, col_ind, order, max_colors=10000):
    rp, ci, od = _i32(row_ptr), _i32(col_ind), _i32(order)
    out = np.zeros(rp.size - 1, dtype=np.int32)
    ms = lib().oracle_gc(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(od, ctypes.c_int),
                         int(max_colors), _p(out, ctypes.c_int))
    return out, ms


def gc_verify(row_ptr, col_ind, color):
    """SimpleVerifyGc: (conflicting stored edges, colours used, uncoloured vertices)."""
    rp, ci, cv = _i32(row_ptr), _i32(col_ind), _i32(color)
    mx, un = ctypes.c_int(0), ctypes.c_int(0)
    err = lib().oracle_gc_verify(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(cv, ctypes.c_int),
                                 ctypes.byref(mx), ctypes.byref(un))
    return err, mx.value, un.value


def lgc(row_ptr, col_ind, src, alpha, eps, max_niter):
    rp, ci = _i32(row_ptr), _i32(col_ind)
    out = np.zeros(rp.size - 1, dtype=np.float32)
    ms = lib().oracle_lgc(rp.size - 1, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(out, ctypes.c_float),
                          int(src), ctypes.c_double(alpha), ctypes.c_double(eps), int(max_niter))
    return out, ms


_LIB_OMP = None


def bfs_all_cores(row_ptr, col_ind, src, nthreads=0):
    """NOT the reference (its oracles are sequential): the same labels on all host cores, for bench.py's
    context line.  Symmetric graphs only (the bottom-up step reads out-neighbours as in-neighbours).
    -> (depth, ms, threads used)"""
    global _LIB_OMP
    if _LIB_OMP is None:
        so = os.path.join(_HERE, "liboracle_omp.so")
        src_c = os.path.join(_HERE, "simple_reference_omp.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src_c):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", so, src_c])
        _LIB_OMP = ctypes.CDLL(so)
        _LIB_OMP.oracle_bfs_all_cores.restype = ctypes.c_double
    rp, ci = _i32(row_ptr), _i32(col_ind)
    n = rp.size - 1
    depth = np.zeros(n, dtype=np.float32)
    used = ctypes.c_int(0)
    ms = _LIB_OMP.oracle_bfs_all_cores(n, _p(rp, ctypes.c_int), _p(ci, ctypes.c_int), _p(depth, ctypes.c_float),
                                       int(src), int(nthreads), ctypes.byref(used))
    return depth, ms, used.value


def bfs_do_stats(csr_ptr, csr_ind, csc_ptr, csc_ind, src, mxvmode=10, switchpoint=0.01,
                 max_niter=10000, max_levels=100000, edgeswitch=0.0):
    a, b, c, d = _i32(csr_ptr), _i32(csr_ind), _i32(csc_ptr), _i32(csc_ind)
    n = a.size - 1
    depth = np.zeros(n, dtype=np.float32)
    stats = np.zeros((max_levels, 6), dtype=np.int64)
    lv = lib().oracle_bfs_do_stats(n, _p(a, ctypes.c_int), _p(b, ctypes.c_int), _p(c, ctypes.c_int),
                                   _p(d, ctypes.c_int), int(src), int(mxvmode),
                                   ctypes.c_float(switchpoint), int(max_niter),
                                   _p(depth, ctypes.c_float), _p(stats, ctypes.c_longlong),
                                   int(max_levels), ctypes.c_float(edgeswitch))
    return depth, stats[:min(lv, max_levels)].copy()
