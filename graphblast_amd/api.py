"""Host-side mirror of the reference frontend over the C ABI.

Same names, argument order and error behaviour as `graphblas::Vector`,
`graphblas::Matrix`, `graphblas::Descriptor` (graphblas/{vector,matrix,descriptor}.hpp)
and the free functions of graphblas/operations.hpp, so a test written against the
reference reads the same here.  Operations return the `graphblas::Info` code (they do
not raise) exactly like the reference's functions; container constructors raise on
allocation failure.  All compute happens in libgrb_hip.so -- this file moves pointers.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import BfsLevel, BfsResult, AlgoResult

# graphblas/types.hpp
GrB_NULL = None
GrB_ALL = None
(GrB_SUCCESS, GrB_UNINITIALIZED_OBJECT, GrB_NULL_POINTER, GrB_INVALID_VALUE, GrB_INVALID_INDEX,
 GrB_DOMAIN_MISMATCH, GrB_DIMENSION_MISMATCH, GrB_OUTPUT_NOT_EMPTY, GrB_NO_VALUE,
 GrB_NOT_IMPLEMENTED, GrB_OUT_OF_MEMORY, GrB_INSUFFICIENT_SPACE, GrB_INVALID_OBJECT,
 GrB_INDEX_OUT_OF_BOUNDS, GrB_PANIC) = range(15)
GrB_UNKNOWN, GrB_SPARSE, GrB_DENSE = 0, 1, 2
(GrB_MASK, GrB_OUTP, GrB_INP0, GrB_INP1, GrB_MODE, GrB_TA, GrB_TB, GrB_NT, GrB_MXVMODE, GrB_TOL,
 GrB_BACKEND) = range(11)
GrB_SCMP, GrB_REPLACE, GrB_TRAN, GrB_DEFAULT = 0, 1, 2, 3
GrB_PUSHPULL, GrB_PUSHONLY, GrB_PULLONLY = 10, 11, 12

F32, I32 = 0, 1
_NP = {F32: np.float32, I32: np.int32}

# graphblas/stddef.hpp:159-172 / :195-213 (order == include/grb_hip.h enums)
MONOIDS = ["Plus", "Multiplies", "Minimum", "Maximum", "LogicalOr", "LogicalAnd", "Greater",
           "CustomLess", "NotEqualTo"]
SEMIRINGS = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusDivides",
             "PlusGreater", "GreaterPlus", "PlusMinus", "PlusLess", "CustomLessPlus",
             "MinimumMultiplies", "MultipliesMultiplies", "NotEqualToPlus", "MinimumSelectSecond",
             "PlusNotEqualTo", "CustomLessLess", "MinimumNotEqualTo"]


def _monoid_id(op):
    if isinstance(op, int):
        return op
    name = op[:-len("Monoid")] if op.endswith("Monoid") else op
    return MONOIDS.index(name)


def _semiring_id(op):
    if isinstance(op, int):
        return op
    name = op[:-len("Semiring")] if op.endswith("Semiring") else op
    return SEMIRINGS.index(name)


BINARY_OPS = ["logical_or", "logical_and", "logical_xor", "equal", "not_equal_to", "greater", "less", "greater_equal",
              "less_equal", "first", "second", "minimum", "maximum", "plus", "minus", "multiplies", "divides"]


def register_semiring(add_op, add_identity, mul_op):
    """REGISTER_SEMIRING over REGISTER_MONOID (graphblas/stddef.hpp:140-191): a monoid (binary operator by name +
    identity) with any binary operator as the multiply.  Returns an id usable wherever a semiring name is."""
    out = C.c_int(0)
    _lib.call("grb_semiring_register", BINARY_OPS.index(add_op), float(add_identity), BINARY_OPS.index(mul_op),
              C.byref(out))
    return out.value


def _h(obj):
    return None if obj is None else obj._h


def _accum(accum):
    return 0 if accum is None else 1


def _dtype_code(dtype):
    if dtype in (F32, I32):
        return dtype
    dt = np.dtype(dtype)
    if dt == np.float32:
        return F32
    if dt == np.int32 or dt == np.bool_:
        return I32          # Vector<bool> of the reference maps to 4-byte ints
    raise TypeError("graphblast_amd supports float32 and int32 vectors/matrices")


class Descriptor:
    """graphblas::Descriptor (descriptor.hpp:17-39)."""

    def __init__(self):
        self._h = C.c_void_p()
        _lib.call("grb_descriptor_new", C.byref(self._h))

    def __del__(self):
        try:
            if self._h:
                _lib.load().grb_descriptor_free(self._h)
                self._h = None
        except Exception:
            pass

    def set(self, field, value):
        return _lib.load().grb_descriptor_set(self._h, field, value)

    def get(self, field):
        out = C.c_int(0)
        _lib.call("grb_descriptor_get", self._h, field, C.byref(out))
        return out.value

    def toggle(self, field):
        return _lib.load().grb_descriptor_toggle(self._h, field)

    def loadArgs(self, **vm):
        """Descriptor::loadArgs: parseArgs defaults (util.hpp:39-132), then the given flags."""
        info = _lib.load().grb_descriptor_load_defaults(self._h)
        for k, val in vm.items():
            if info != 0:
                break
            info = _lib.load().grb_descriptor_set_arg(self._h, k.encode(), float(val))
        return info

    def arg(self, name):
        out = C.c_double(0)
        _lib.call("grb_descriptor_get_arg", self._h, name.encode(), C.byref(out))
        return out.value

    @property
    def lastmxv_(self):
        out = C.c_int(0)
        _lib.call("grb_descriptor_lastmxv", self._h, C.byref(out))
        return out.value


class Vector:
    """graphblas::Vector<T> (vector.hpp:12-66); T in {float32, int32}."""

    def __init__(self, nsize, dtype=np.float32):
        self.dtype_code = _dtype_code(dtype)
        self.np_dtype = _NP[self.dtype_code]
        self._h = C.c_void_p()
        _lib.call("grb_vector_new", C.byref(self._h), self.dtype_code, int(nsize))
        self._keep = []

    def __del__(self):
        try:
            if self._h:
                _lib.load().grb_vector_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- C API methods --------------------------------------------------------
    def dup(self, rhs):
        return _lib.load().grb_vector_dup(self._h, rhs._h)

    def clear(self):
        return _lib.load().grb_vector_clear(self._h)

    def size(self):
        out = C.c_int(0)
        _lib.call("grb_vector_size", self._h, C.byref(out))
        return out.value

    def nvals(self):
        out = C.c_int(0)
        _lib.call("grb_vector_nvals", self._h, C.byref(out))
        return out.value

    def build(self, *args):
        """build(indices, values, nvals, dup)  -> sparse;  build(values, nvals) -> dense."""
        lib = _lib.load()
        if len(args) >= 3 and args[1] is not None and not np.isscalar(args[1]):
            idx = np.ascontiguousarray(args[0], dtype=np.int32)
            val = np.ascontiguousarray(args[1], dtype=self.np_dtype)
            n = int(args[2])
            return lib.grb_vector_build_sparse(self._h, idx.ctypes.data, val.ctypes.data, n)
        val = np.ascontiguousarray(args[0], dtype=self.np_dtype)
        n = int(args[1]) if len(args) > 1 else val.size
        return lib.grb_vector_build_dense(self._h, val.ctypes.data, n)

    def build_device(self, d_values_ptr, nvals, d_indices_ptr=None):
        """build(T* values, nvals) / build(Index*, T*, nvals): adopt device pointers."""
        lib = _lib.load()
        if d_indices_ptr is None:
            return lib.grb_vector_adopt_dense(self._h, d_values_ptr, int(nvals))
        return lib.grb_vector_adopt_sparse(self._h, d_indices_ptr, d_values_ptr, int(nvals))

    def resize(self, nsize):
        return _lib.load().grb_vector_resize(self._h, int(nsize))

    def setElement(self, val, index):
        return _lib.load().grb_vector_set_element(self._h, float(val), int(index))

    def extractElement(self, index):
        out = C.c_double(0)
        info = _lib.load().grb_vector_extract_element(self._h, C.byref(out), int(index))
        return info, self.np_dtype(out.value)

    def extractTuples(self, n=None, sparse=False):
        """extractTuples(values, n) (dense; densifies a sparse vector with fill 0) or, with
        sparse=True, extractTuples(indices, values, n).  Returns (info, ...arrays)."""
        lib = _lib.load()
        if sparse:
            n = self.nvals() if n is None else int(n)
            idx = np.zeros(max(n, 1), dtype=np.int32)
            val = np.zeros(max(n, 1), dtype=self.np_dtype)
            cn = C.c_int(n)
            info = lib.grb_vector_extract_tuples_sparse(self._h, idx.ctypes.data, val.ctypes.data, C.byref(cn))
            return info, idx[:n], val[:n]
        n = self.size() if n is None else int(n)
        val = np.zeros(max(n, 1), dtype=self.np_dtype)
        cn = C.c_int(n)
        info = lib.grb_vector_extract_tuples_dense(self._h, val.ctypes.data, C.byref(cn))
        return info, val[:n]

    def fill(self, val):
        return _lib.load().grb_vector_fill(self._h, float(val))

    def fillAscending(self, nvals=0):
        return _lib.load().grb_vector_fill_ascending(self._h, int(nvals))

    def getStorage(self):
        out = C.c_int(0)
        _lib.call("grb_vector_get_storage", self._h, C.byref(out))
        return out.value

    def setStorage(self, storage):
        return _lib.load().grb_vector_set_storage(self._h, int(storage))

    def swap(self, rhs):
        return _lib.load().grb_vector_swap(self._h, rhs._h)

    # backend::Vector methods the reference's tests reach through `#define private public`
    def convert(self, identity, switchpoint, desc):
        return _lib.load().grb_vector_convert(self._h, float(identity), float(switchpoint), desc._h)

    def sparse2dense(self, identity, desc=None):
        return _lib.load().grb_vector_sparse2dense(self._h, float(identity), _h(desc))

    def dense2sparse(self, identity, desc):
        return _lib.load().grb_vector_dense2sparse(self._h, float(identity), desc._h)

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.call("grb_vector_device_ptrs", self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


class Matrix:
    """graphblas::Matrix<T> (matrix.hpp:13-84)."""

    def __init__(self, nrows, ncols, dtype=np.float32):
        self.dtype_code = _dtype_code(dtype)
        self.np_dtype = _NP[self.dtype_code]
        self._h = C.c_void_p()
        self._nrows, self._ncols = int(nrows), int(ncols)
        _lib.call("grb_matrix_new", C.byref(self._h), self.dtype_code, self._nrows, self._ncols)
        self._keep = []

    def __del__(self):
        try:
            if self._h:
                _lib.load().grb_matrix_free(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def from_mtx(cls, path, dtype=np.float32, directed=0):
        """readMtx + build with the MatrixMarket text parsed on the device (grb_matrix_load_mtx);
        `directed` as readMtx: 0 = symmetric iff the banner says so, 1 = directed, 2 = undirected."""
        self = cls.__new__(cls)
        self.dtype_code = _dtype_code(dtype)
        self.np_dtype = _NP[self.dtype_code]
        self._h = C.c_void_p()
        self._keep = []
        dims = (C.c_int * 3)()
        info = _lib.load().grb_matrix_load_mtx(C.byref(self._h), str(path).encode(), self.dtype_code, int(directed), dims)
        if info != 0:
            self._h = None
            raise RuntimeError("grb_matrix_load_mtx(%s): Info %d" % (path, info))
        self._nrows, self._ncols = int(dims[0]), int(dims[1])
        return self

    def build(self, row_indices, col_indices, values, nvals=None, dup=None):
        r = np.ascontiguousarray(row_indices, dtype=np.int32)
        c = np.ascontiguousarray(col_indices, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=self.np_dtype)
        n = r.size if nvals is None else int(nvals)
        return _lib.load().grb_matrix_build(self._h, r.ctypes.data, c.ctypes.data, v.ctypes.data, n)

    def build_csr(self, row_ptr, col_ind, values, csc=None):
        p = np.ascontiguousarray(row_ptr, dtype=np.int32)
        i = np.ascontiguousarray(col_ind, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=self.np_dtype)
        if csc is None:
            return _lib.load().grb_matrix_build_csr(self._h, p.ctypes.data, i.ctypes.data, v.ctypes.data,
                                                    i.size, None, None, None)
        cp = np.ascontiguousarray(csc[0], dtype=np.int32)
        ci = np.ascontiguousarray(csc[1], dtype=np.int32)
        cv = np.ascontiguousarray(csc[2], dtype=self.np_dtype)
        return _lib.load().grb_matrix_build_csr(self._h, p.ctypes.data, i.ctypes.data, v.ctypes.data, i.size,
                                                cp.ctypes.data, ci.ctypes.data, cv.ctypes.data)

    def build_device_csr(self, d_ptr, d_ind, d_val, nvals, d_cptr=None, d_cind=None, d_cval=None, keep=()):
        """build(row_ptr, col_ind, values, nvals) adopting device pointers; `keep` holds
        whatever owns that memory (e.g. torch tensors) alive."""
        self._keep = list(keep)
        return _lib.load().grb_matrix_adopt_device_csr(self._h, d_ptr, d_ind, d_val, int(nvals), d_cptr, d_cind,
                                                       d_cval)

    def ingest_device(self, d_rows, d_cols, d_vals, nvals, symmetrize=False, drop_selfloops=True,
                      drop_duplicates=True, keep=()):
        """readMtx's loader semantics + build on a DEVICE coordinate list (graphblas/util.hpp:197-329,
        363-430): optionally add the reverse of every off-diagonal entry, drop self loops, drop
        duplicates (first wins); d_vals None = pattern."""
        self._keep = list(keep)
        flags = (1 if symmetrize else 0) | (2 if drop_selfloops else 0) | (4 if drop_duplicates else 0)
        return _lib.load().grb_matrix_ingest_device(self._h, d_rows, d_cols, d_vals, int(nvals), flags)

    def nrows(self):
        return self._nrows

    def ncols(self):
        return self._ncols

    def nvals(self):
        out = C.c_int(0)
        _lib.call("grb_matrix_nvals", self._h, C.byref(out))
        return out.value

    def _host(self, which):
        p, i, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.call(which, self._h, C.byref(p), C.byref(i), C.byref(v))
        n = self._nrows if which.endswith("csr") else self._ncols
        nv = self.nvals()
        ptr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n + 1,)).copy()
        if nv == 0:                      # e.g. data/small/test_sgm.mtx: only self loops -> empty
            return ptr, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=self.np_dtype)
        ind = np.ctypeslib.as_array(C.cast(i, C.POINTER(C.c_int32)), shape=(nv,)).copy()
        raw = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_uint32)), shape=(nv,)).copy()
        return ptr, ind, raw.view(self.np_dtype)

    def host_csr(self):
        """h_csrRowPtr_, h_csrColInd_, h_csrVal_ (what the CPU oracles read)."""
        return self._host("grb_matrix_host_csr")

    def host_csc(self):
        return self._host("grb_matrix_host_csc")

    def write_cache(self, path):
        """The reference's binary CSR cache (sparse_matrix.hpp:328-348): int32 nrows, nvals, rowptr, colind."""
        return _lib.load().grb_matrix_write_cache(self._h, os.fsencode(str(path)))

    def build_cache(self, path):
        """Matrix::build(dat_name) (sparse_matrix.hpp:355-407): values = 1, CSC made on the device."""
        info = _lib.load().grb_matrix_build_cache(self._h, os.fsencode(str(path)))
        if info == 0:
            n = C.c_int(0)
            _lib.call("grb_matrix_nrows", self._h, C.byref(n))
            self._nrows = self._ncols = n.value
        return info

    def set_values(self, csr_values):
        v = np.ascontiguousarray(csr_values, dtype=self.np_dtype)
        return _lib.load().grb_matrix_set_values(self._h, v.ctypes.data)


def cache_name(mtx_path, is_undirected):
    """util.hpp:340-357 (convert): `<dir>/.<file>.<ud|d>.<nosl|sl>.bin`."""
    buf = C.create_string_buffer(1024)
    _lib.call("grb_cache_name", os.fsencode(str(mtx_path)), int(bool(is_undirected)), buf, 1024)
    return buf.value.decode()


# ---- graphblas/operations.hpp ------------------------------------------------------
def vxm(w, mask, accum, op, u, A, desc):
    return _lib.load().grb_vxm(_h(w), _h(mask), _accum(accum), _semiring_id(op), _h(u), _h(A), _h(desc))


def mxv(w, mask, accum, op, A, u, desc):
    return _lib.load().grb_mxv(_h(w), _h(mask), _accum(accum), _semiring_id(op), _h(A), _h(u), _h(desc))


def eWiseMult(w, mask, accum, op, u, v, desc):
    return _lib.load().grb_eWiseMult(_h(w), _h(mask), _accum(accum), _semiring_id(op), _h(u), _h(v), _h(desc))


def eWiseAdd(w, mask, accum, op, u, v, desc):
    """eWiseAdd(w, mask, accum, op, u, v | scalar, desc)."""
    if isinstance(v, Vector):
        return _lib.load().grb_eWiseAdd(_h(w), _h(mask), _accum(accum), _semiring_id(op), _h(u), _h(v), _h(desc))
    return _lib.load().grb_eWiseAdd_scalar(_h(w), _h(mask), _accum(accum), _semiring_id(op), _h(u), float(v),
                                           _h(desc))


def reduce(accum, op, u, desc, w=None, mask=None):
    """reduce(&val, accum, MonoidT, Vector u, desc) -> (info, val)  or, with w given and u a
    Matrix, reduce(Vector w, mask, accum, MonoidT, Matrix A, desc) -> info."""
    if isinstance(u, Matrix):
        return _lib.load().grb_reduce_matrix_rows(_h(w), _h(mask), _accum(accum), _monoid_id(op), _h(u), _h(desc))
    out = C.c_double(0)
    info = _lib.load().grb_reduce_vector(C.byref(out), _accum(accum), _monoid_id(op), _h(u), _h(desc))
    return info, out.value


def assign(w, mask, accum, val, indices, nindices, desc):
    """assign(w, mask, accum, val, GrB_ALL, n, desc)."""
    if indices is not None:
        return GrB_NOT_IMPLEMENTED
    return _lib.load().grb_assign(_h(w), _h(mask), _accum(accum), float(val), _h(desc))


UNARY_OPS = ["identity", "ainv", "minv", "abs", "lnot", "bind_first", "bind_second"]      # grb_unary_op
BINARY_OPS = ["logical_or", "logical_and", "logical_xor", "equal", "not_equal_to", "greater", "less", "greater_equal",
              "less_equal", "first", "second", "minimum", "maximum", "plus", "minus", "multiplies", "divides"]


def apply(w, mask, accum, unary, u, desc, binop=None, scalar=0.0):
    """apply on the device (grb_vector_apply / grb_matrix_apply): w = f(u) on every stored element.  unary: a name of
    UNARY_OPS; the two bind kinds take binop (a name of BINARY_OPS) and scalar."""
    k = UNARY_OPS.index(unary)
    b = BINARY_OPS.index(binop) if binop is not None else 0
    fn = "grb_matrix_apply" if isinstance(w, Matrix) else "grb_vector_apply"
    return getattr(_lib.load(), fn)(_h(w), _h(mask), _accum(accum), k, b, float(scalar), _h(u), _h(desc))


def mxm(Cm, mask, accum, op, A, B, desc):
    return _lib.load().grb_mxm(_h(Cm), _h(mask), _accum(accum), _semiring_id(op), _h(A), _h(B), _h(desc))


def tril(Cm, A, desc):
    return _lib.load().grb_matrix_tril(_h(Cm), _h(A), _h(desc))


def reduce_matrix(accum, op, A, desc):
    out = C.c_double(0)
    info = _lib.load().grb_reduce_matrix_scalar(C.byref(out), _accum(accum), _monoid_id(op), _h(A), _h(desc))
    return info, out.value


def assignScatter(w, mask, accum, u, indices, desc):
    return _lib.load().grb_assignScatter(_h(w), _h(mask), _accum(accum), _h(u), _h(indices), _h(desc))


def extractGather(w, mask, accum, u, indices, desc):
    return _lib.load().grb_extractGather(_h(w), _h(mask), _accum(accum), _h(u), _h(indices), _h(desc))


# ---- graphblas/algorithm/*.hpp -----------------------------------------------------
def bfs(v, A, s, desc, fused=False, profile=False, max_levels=4096):
    """algorithm::bfs. Returns (info, result dict). fused=True runs the device-resident loop."""
    res = BfsResult()
    if not fused:
        info = _lib.load().grb_bfs(_h(v), _h(A), int(s), _h(desc), C.byref(res))
        return info, dict(levels=res.levels, tight_ms=res.tight_ms)
    if not profile:                     # no per-level table wanted: nothing to allocate or copy back
        info = _lib.load().grb_bfs_fused(_h(v), _h(A), int(s), _h(desc), C.byref(res), None, 0, 0)
        levels = []
    else:
        lv = (BfsLevel * max_levels)()
        info = _lib.load().grb_bfs_fused(_h(v), _h(A), int(s), _h(desc), C.byref(res), lv, max_levels,
                                         int(profile))
        n = min(res.levels, max_levels)
        levels = [dict(direction="pull" if lv[i].direction else "push", frontier=lv[i].frontier,
                       frontier_edges=lv[i].frontier_edges, discovered=lv[i].discovered, ms=lv[i].ms)
                  for i in range(n)]
    return info, dict(levels=res.levels, tight_ms=res.tight_ms, edges_traversed=res.edges_traversed,
                      reached=res.reached, per_level=levels)


def bfs_enqueue(v, A, s, desc):
    """grb_bfs_fused_enqueue: queue the traversal on the library's stream and return (info, ticket) without waiting.
    v, A and desc must stay alive until bfs_wait(ticket)."""
    t = C.c_int64(0)
    info = _lib.load().grb_bfs_fused_enqueue(_h(v), _h(A), int(s), _h(desc), C.byref(t))
    return info, t.value


def bfs_wait(ticket):
    """grb_bfs_wait: (info, result dict) of a queued traversal; what bfs(..., fused=True) would have returned."""
    res = BfsResult()
    info = _lib.load().grb_bfs_wait(C.c_int64(ticket), C.byref(res))
    return info, dict(levels=res.levels, tight_ms=res.tight_ms, edges_traversed=res.edges_traversed,
                      reached=res.reached, per_level=[])


def bfs_set_lanes(n=-1):
    """grb_bfs_set_lanes: traversals in flight at once for bfs_enqueue (n < 1 only queries); returns the previous value."""
    return int(_lib.load().grb_bfs_set_lanes(int(n)))


def bfs_set_coschedule(k=-1):
    """grb_bfs_set_coschedule: traversals per launch for bfs_enqueue (k < 1 only queries); returns the previous value."""
    return int(_lib.load().grb_bfs_set_coschedule(int(k)))


def bfs_coschedule_profile(on):
    """grb_bfs_coschedule_profile: on=True starts HIP events around the launches of several traversals; on=False stops and
    returns dict(ms_total, launches, traversals)."""
    ms, nl, nt = C.c_double(0), C.c_int(0), C.c_int(0)
    info = _lib.load().grb_bfs_coschedule_profile(1 if on else 0, C.byref(ms), C.byref(nl), C.byref(nt))
    assert info == 0, info
    return dict(ms_total=ms.value, launches=nl.value, traversals=nt.value)


def bfs_host_times(reset=False):
    """Host microseconds spent queueing / waiting inside the one-launch traversal since the last reset, and the calls."""
    e, w, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
    _lib.call("grb_bfs_host_times", C.byref(e), C.byref(w), C.byref(n), int(bool(reset)))
    return dict(enqueue_us=e.value, wait_us=w.value, calls=n.value)


def spmm(op, A, d_B, d_C, k, desc=None, tran=False):
    """C = A (+).(x) B with dense row-major device arrays B [ncols x k], C [nrows x k] (grb_spmm)."""
    return _lib.load().grb_spmm(_semiring_id(op), _h(A), int(bool(tran)), d_B, d_C, int(k), _h(desc))


def bfs_batch(vs, A, sources, desc):
    """Up to 64 traversals at once (grb_bfs_batch): vs[i] receives the depth labels of sources[i]."""
    k = len(vs)
    assert k == len(sources)
    handles = (C.c_void_p * k)(*[_h(v) for v in vs])
    src = np.ascontiguousarray(sources, dtype=np.int32)
    res = BfsResult()
    info = _lib.load().grb_bfs_batch(handles, k, _h(A), src.ctypes.data, _h(desc), C.byref(res))
    return info, dict(levels=res.levels, tight_ms=res.tight_ms, edges_traversed=res.edges_traversed,
                      reached=res.reached)


def bfs_batch_set_tail(edges=-1):
    """Out-edge limit of the levels grb_bfs_batch runs inside its one light-level launch (grb_bfs_batch_set_tail):
    0 never, < 0 only queries; returns the previous limit."""
    return int(_lib.load().grb_bfs_batch_set_tail(int(edges)))


def sssp(v, A, s, desc):
    res = AlgoResult()
    info = _lib.load().grb_sssp(_h(v), _h(A), int(s), _h(desc), C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms, succ=res.last_value)


def pr(p, A, alpha, eps, desc):
    res = AlgoResult()
    info = _lib.load().grb_pr(_h(p), _h(A), float(alpha), float(eps), _h(desc), C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms, error=res.last_value)


def cc(v, A, seed, desc):
    res = AlgoResult()
    info = _lib.load().grb_cc(_h(v), _h(A), int(seed), _h(desc), C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms, succ=res.last_value)


def tc(A, B, desc):
    res = AlgoResult()
    n = C.c_int64(0)
    info = _lib.load().grb_tc(C.byref(n), _h(A), _h(B), _h(desc), C.byref(res))
    return info, n.value, dict(tight_ms=res.tight_ms)


def tc_set_product(on):
    """grb_tc_set_product: 0 = grb_tc counts without the product where that is a count and pays (default), 1 = always the
    product in B, 2 = the count wherever it is a count; < 0 queries."""
    return _lib.load().grb_tc_set_product(int(on))


def tc_release(A):
    """grb_tc_release: drops the orientation the matrix keeps after its first count."""
    return _lib.load().grb_tc_release(_h(A))


def tc_last():
    """grb_tc_last: which way the last tc went and what it cost."""
    t = _lib.TcInfo()
    info = _lib.load().grb_tc_last(C.byref(t))
    return info, dict(path=t.path, prep_ms=t.prep_ms, count_ms=t.count_ms, longest_list=t.longest_list, tasks=list(t.tasks))


def tc_dense_core(L, k_want, method=0, dense_from=0):
    """grb_tc_dense_core: the product C<L> = L (+.x) L^T restricted to the k_want longest rows of L, as K x K bit rows.
    method 0 popcount, 1 MFMA (v_mfma_i32_16x16x64_i8), 2 MFMA for tiles of >= dense_from mask entries.  Returns (info, dict)."""
    res = _lib.TcCoreResult()
    info = _lib.load().grb_tc_dense_core(_h(L), int(k_want), int(method), int(dense_from), C.byref(res))
    return info, dict(core_rows=res.core_rows, min_row_length=res.min_row_length, core_entries=res.core_entries, count=res.count,
                      checksum=res.checksum, build_ms=res.build_ms, product_ms=res.product_ms, tiles=res.tiles,
                      tiles_mfma=res.tiles_mfma, tiles_by_density=list(res.tiles_by_density))


def traceMxmTranspose(op, A, B, desc):
    out = C.c_double(0)
    info = _lib.load().grb_trace_mxm_transpose(C.byref(out), _semiring_id(op), _h(A), _h(B), _h(desc))
    return info, out.value


def scatter(w, mask, u, val, desc):
    return _lib.load().grb_scatter(_h(w), _h(mask), _h(u), float(val), _h(desc))


def graph_color(w, A, desc):
    n = C.c_int(0)
    info = _lib.load().grb_graph_color(_h(w), _h(A), _h(desc), C.byref(n))
    return info, n.value


def mis(v, A, seed, desc, weights=None):
    """algorithm::mis; `weights` (an int Vector) replaces the host-drawn srand(seed)/rand() vector."""
    res = AlgoResult()
    info = _lib.load().grb_mis(_h(v), _h(A), int(seed), _h(weights), _h(desc), C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms)


def gc(v, A, seed, max_colors, algo, desc, weights=None):
    """algorithm::gcJP (algo 0) / gcMIS (1) / gcIS (2)."""
    res = AlgoResult()
    info = _lib.load().grb_gc(_h(v), _h(A), int(seed), _h(weights), int(max_colors), int(algo), _h(desc),
                              C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms, succ=res.last_value)


def lgc(p, A, s, alpha, eps, desc):
    res = AlgoResult()
    info = _lib.load().grb_lgc(_h(p), _h(A), int(s), float(alpha), float(eps), _h(desc), C.byref(res))
    return info, dict(iterations=res.iterations, tight_ms=res.tight_ms, succ=res.last_value)


def diameter(v, A, s_start, s_end, desc):
    dmax, dind = C.c_int(0), C.c_int(-1)
    info = _lib.load().grb_diameter(_h(v), _h(A), int(s_start), int(s_end), _h(desc), C.byref(dmax), C.byref(dind))
    return info, dmax.value, dind.value


# ---- raw kernels / timing -----------------------------------------------------------
def k_spmv(A, tran, op, d_u, d_mask, scmp, accum, d_w):
    return _lib.load().grb_k_spmv(_h(A), int(tran), _semiring_id(op), d_u, d_mask, int(scmp), int(accum), d_w)


def sssp_set_nearfar(mode=-2):
    """-1 auto (default), 0 synchronous rounds only, 1 near / far whenever eligible; -2 queries (grb_sssp_set_nearfar)."""
    return int(_lib.load().grb_sssp_set_nearfar(int(mode)))


def sssp_last_work():
    """(vertices expanded, out-edges relaxed, vertices marked) over all passes of the last near / far sssp()"""
    out = (C.c_int64 * 3)()
    _lib.load().grb_sssp_last_work(out)
    return tuple(int(x) for x in out)


def sssp_last_order():
    """0: the last sssp() ran the reference's synchronous rounds; else the passes of the near / far order."""
    return int(_lib.load().grb_sssp_last_order())


def spmv_set_bands(k=0):
    """LDS prefixes SpMV plans prepared from now on may use (grb_spmv_set_bands); 0 only queries."""
    return int(_lib.load().grb_spmv_set_bands(int(k)))


def spmv_plan_info(A, tran=0, warm=False):
    """{"bands", "band_nnz", "pieces", "nhot"} of the SpMV plan of this orientation (grb_spmv_plan_info)."""
    bands, nhot = C.c_int(0), C.c_int(0)
    bn, pc = C.c_int64(0), C.c_int64(0)
    _lib.call("grb_spmv_plan_info", _h(A), int(bool(tran)), int(bool(warm)), C.byref(bands), C.byref(bn), C.byref(pc),
              C.byref(nhot))
    return {"bands": bands.value, "band_nnz": bn.value, "pieces": pc.value, "nhot": nhot.value}


def cc_set_fused(on=-1):
    """1: FastSV's element-wise tail in one launch (default); 0: the reference's call sequence; < 0 queries"""
    return int(_lib.load().grb_cc_set_fused(int(on)))


def spmv_set_format(fmt=-1):
    """matrix format of the generic SpMV (grb_spmv_set_format): 0 CSR only, 1 auto, 2 column-sorted bands wherever
    the monoid allows; < 0 only queries"""
    return int(_lib.load().grb_spmv_set_format(int(fmt)))


def set_lazy(on=-1):
    """the queue of element-wise calls (grb_set_lazy): 1 queue and fuse (default), 0 run every call at once; < 0 queries.
    Returns the previous setting."""
    return int(_lib.load().grb_set_lazy(int(on)))


def lazy_pending():
    """element-wise calls waiting in the queue (grb_lazy_pending; does not flush)"""
    return int(_lib.load().grb_lazy_pending())


def lazy_fused_reductions():
    """reductions that ran inside a chain's launch so far (grb_lazy_fused_reductions)"""
    return int(_lib.load().grb_lazy_fused_reductions())


def spmv_set_reuse_threshold(launches=-1):
    """CSR-kernel products after which `auto` prepares the column-sorted format for an orientation
    (grb_spmv_set_reuse_threshold; 0 = at once); < 0 only queries.  Returns the previous value."""
    return int(_lib.load().grb_spmv_set_reuse_threshold(int(launches)))


def spmv_format_info(A, tran=0):
    """the column-sorted copy of this orientation, if prepared (grb_spmv_format_info)"""
    used, bands, items, hub, iso = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    groups, nbytes = C.c_int64(0), C.c_int64(0)
    _lib.call("grb_spmv_format_info", _h(A), int(bool(tran)), C.byref(used), C.byref(groups), C.byref(bands),
              C.byref(items), C.byref(hub), C.byref(iso), C.byref(nbytes))
    return {"in_use": used.value, "groups": groups.value, "bands": bands.value, "items": items.value,
            "hub_rows": hub.value, "iso": iso.value, "bytes_per_launch": nbytes.value}


def k_spmv_bytes(A, tran):
    return int(_lib.load().grb_k_spmv_bytes(_h(A), int(tran)))


def timer_start():
    _lib.call("grb_timer_start")


def timer_stop():
    out = C.c_float(0)
    _lib.call("grb_timer_stop", C.byref(out))
    return out.value


def set_stream(ptr):
    _lib.call("grb_set_stream", ptr)


def device_info():
    buf = C.create_string_buffer(256)
    _lib.call("grb_device_info", buf, 256)
    return buf.value.decode()
