"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper over oracle/_ref/libsimple_ref.so and
libsimple_ref_nr.so: the REFERENCE's own SimpleReference{Bfs,Sssp,Pr,Cc,Tc,Lgc,Mis,Gc},
SimpleVerify{Cc,Mis,Gc}, readMtx, coo2csr / coo2csc and cache-name rule, compiled from
/root/reference by `make -C oracle ref` (oracle/ref_simple_lib.cpp).  The .so files are
built in the build container and travel to the GPU box with the snapshot; nothing here
reads /root/reference at run time.  `available()` says whether they are present."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_A = os.path.join(_HERE, "_ref", "libsimple_ref.so")
_B = os.path.join(_HERE, "_ref", "libsimple_ref_nr.so")
_LIBS = None

_I = ctypes.POINTER(ctypes.c_int)
_F = ctypes.POINTER(ctypes.c_float)


def available():
    return os.path.exists(_A) and os.path.exists(_B)


def _libs():
    global _LIBS
    if _LIBS is None:
        a, b = ctypes.CDLL(_A), ctypes.CDLL(_B)
        a.ref_bfs.argtypes = [ctypes.c_int, _I, _I, _F, ctypes.c_int, ctypes.c_int]
        a.ref_sssp.argtypes = [ctypes.c_int, _I, _I, _F, _F, ctypes.c_int, ctypes.c_int]
        a.ref_pr.argtypes = [ctypes.c_int, _I, _I, _F, _F, ctypes.c_float, ctypes.c_float, ctypes.c_int]
        a.ref_tc.argtypes = [ctypes.c_int, _I, _I]
        a.ref_lgc.argtypes = [ctypes.c_int, _I, _I, _F, _F, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                              ctypes.c_int, ctypes.c_int]
        a.ref_lgc.restype = None
        a.ref_read_mtx.argtypes = [ctypes.c_char_p, ctypes.c_int, _I, _I, _I]
        a.ref_read_mtx_copy.argtypes = [_I, _I, _F]
        a.ref_read_mtx_copy.restype = None
        a.ref_coo2csr.argtypes = [_I, _I, _F]
        a.ref_coo2csc.argtypes = [_I, _I, _F]
        a.ref_cache_name.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        a.ref_sssp_weights.argtypes = [_F, ctypes.c_int, ctypes.c_int]
        b.ref_cc.argtypes = [ctypes.c_int, _I, _I, _I]
        b.ref_cc_verify.argtypes = [ctypes.c_int, _I, _I, _I, ctypes.c_int]
        b.ref_mis.argtypes = [ctypes.c_int, _I, _I, _I, ctypes.c_int]
        b.ref_mis_verify.argtypes = [ctypes.c_int, _I, _I, _I]
        b.ref_gc.argtypes = [ctypes.c_int, _I, _I, _I, ctypes.c_int, ctypes.c_int]
        b.ref_gc_verify.argtypes = [ctypes.c_int, _I, _I, _I, ctypes.c_int]
        _LIBS = (a, b)
    return _LIBS


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pi(a):
    return a.ctypes.data_as(_I)


def _pf(a):
    return a.ctypes.data_as(_F)


def bfs(row_ptr, col_ind, src, stop=10000):
    """SimpleReferenceBfs<float> (test_bfs.hpp:11-61): depth labels from 1, 0 = unreached."""
    rp, ci = _i(row_ptr), _i(col_ind)
    depth = np.zeros(rp.size - 1, dtype=np.float32)
    sd = _libs()[0].ref_bfs(rp.size - 1, _pi(rp), _pi(ci), _pf(depth), int(src), int(stop))
    return depth, sd


def sssp(row_ptr, col_ind, val, src, stop=10000):
    """SimpleReferenceSssp<float> (test_sssp.hpp:15-79): FLT_MAX = unreached."""
    rp, ci, v = _i(row_ptr), _i(col_ind), _f(val).copy()
    dist = np.zeros(rp.size - 1, dtype=np.float32)
    sd = _libs()[0].ref_sssp(rp.size - 1, _pi(rp), _pi(ci), _pf(v), _pf(dist), int(src), int(stop))
    return dist, sd


def pr(row_ptr, col_ind, alpha=0.85, eps=1e-8, max_niter=10):
    """SimpleReferencePr<float> (test_pr.hpp:15-80)."""
    rp, ci = _i(row_ptr), _i(col_ind)
    val = np.ones(ci.size, dtype=np.float32)
    rank = np.zeros(rp.size - 1, dtype=np.float32)
    it = _libs()[0].ref_pr(rp.size - 1, _pi(rp), _pi(ci), _pf(val), _pf(rank), alpha, eps, int(max_niter))
    return rank, it


def tc(row_ptr, col_ind, nrows=None):
    """SimpleReferenceTc<int> (test_tc.hpp:41-87).  nrows: only the first rows' wedges (its loop bound)."""
    rp, ci = _i(row_ptr), _i(col_ind)
    return _libs()[0].ref_tc(rp.size - 1 if nrows is None else int(nrows), _pi(rp), _pi(ci))


def lgc(row_ptr, col_ind, val, src, alpha, eps, max_niter, dense=False):
    rp, ci, v = _i(row_ptr), _i(col_ind), _f(val)
    out = np.zeros(rp.size - 1, dtype=np.float32)
    _libs()[0].ref_lgc(rp.size - 1, _pi(rp), _pi(ci), _pf(v), _pf(out), int(src), alpha, eps, int(max_niter),
                       int(bool(dense)))
    return out


def cc(row_ptr, col_ind):
    """SimpleReferenceCc (test_cc.hpp:14-56): labels from 1 in discovery order."""
    rp, ci = _i(row_ptr), _i(col_ind)
    label = np.zeros(rp.size - 1, dtype=np.int32)
    _libs()[1].ref_cc(rp.size - 1, _pi(rp), _pi(ci), _pi(label))
    return label


def cc_verify(row_ptr, col_ind, label, suppress_zero=False):
    rp, ci, lb = _i(row_ptr), _i(col_ind), _i(label)
    return _libs()[1].ref_cc_verify(rp.size - 1, _pi(rp), _pi(ci), _pi(lb), int(suppress_zero))


def mis(row_ptr, col_ind, seed):
    rp, ci = _i(row_ptr), _i(col_ind)
    out = np.zeros(rp.size - 1, dtype=np.int32)
    _libs()[1].ref_mis(rp.size - 1, _pi(rp), _pi(ci), _pi(out), int(seed))
    return out


def mis_verify(row_ptr, col_ind, mis_):
    rp, ci, m = _i(row_ptr), _i(col_ind), _i(mis_)
    return _libs()[1].ref_mis_verify(rp.size - 1, _pi(rp), _pi(ci), _pi(m))


def gc(row_ptr, col_ind, seed, max_colors):
    rp, ci = _i(row_ptr), _i(col_ind)
    out = np.zeros(rp.size - 1, dtype=np.int32)
    _libs()[1].ref_gc(rp.size - 1, _pi(rp), _pi(ci), _pi(out), int(seed), int(max_colors))
    return out


def gc_verify(row_ptr, col_ind, colour, suppress_zero=False):
    rp, ci, c = _i(row_ptr), _i(col_ind), _i(colour)
    return _libs()[1].ref_gc_verify(rp.size - 1, _pi(rp), _pi(ci), _pi(c), int(suppress_zero))


def read_mtx(path, directed=0):
    """readMtx<float> + coo2csr + coo2csc (util.hpp:363-430, 501-572).  Returns a dict with
    nrows, ncols, nvals (as readMtx reports it), the coordinate lists and CSR / CSC triples."""
    a = _libs()[0]
    nr, nc, nv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if a.ref_read_mtx(os.fsencode(path), int(directed), ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nv)):
        raise FileNotFoundError(path)
    k = a.ref_read_mtx_size()
    row, col, val = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.float32)
    a.ref_read_mtx_copy(_pi(row), _pi(col), _pf(val))
    # coo2csr's arrays are nvals long: what readMtx reports after removeSelfloop
    csr = (np.zeros(nr.value + 1, np.int32), np.zeros(max(k, 1), np.int32), np.zeros(max(k, 1), np.float32))
    csc = (np.zeros(nc.value + 1, np.int32), np.zeros(max(k, 1), np.int32), np.zeros(max(k, 1), np.float32))
    a.ref_coo2csr(_pi(csr[0]), _pi(csr[1]), _pf(csr[2]))
    a.ref_coo2csc(_pi(csc[0]), _pi(csc[1]), _pf(csc[2]))
    return dict(nrows=nr.value, ncols=nc.value, nvals=nv.value, row=row, col=col, val=val,
                csr=(csr[0], csr[1][:k], csr[2][:k]), csc=(csc[0], csc[1][:k], csc[2][:k]))


def cache_name(path, is_undirected):
    buf = ctypes.create_string_buffer(512)
    _libs()[0].ref_cache_name(os.fsencode(path), int(bool(is_undirected)), buf, 512)
    return buf.value.decode()


def sssp_weights(nvals, seed):
    w = np.zeros(nvals, np.float32)
    _libs()[0].ref_sssp_weights(_pf(w), int(nvals), int(seed))
    return w
