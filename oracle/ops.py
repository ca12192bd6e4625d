"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the reference's mxv/vxm hot path, one level below the
frontend: the `graphblas::backend` containers and operations, with the reference's
dispatch rules and quirks (SURVEY.md 8(a) "semantic quirks" 1-10).

  Descriptor      <- backend/cuda/descriptor.hpp:14-287 (+ util.hpp:39-132 defaults)
  Vector          <- backend/cuda/vector.hpp, sparse_vector.hpp, dense_vector.hpp
  Matrix          <- backend/cuda/sparse_matrix.hpp:289-351 (build: coo2csr + coo2csc)
  vxm / mxv       <- graphblas/operations.hpp:59-127 + backend/cuda/operations.hpp:80-327
  _spmv           <- backend/cuda/spmv.hpp:20-236 + kernels/spmv.hpp:10-59
  _spmspv_merge   <- backend/cuda/spmspv.hpp:15-257 + spmspv_inner.hpp:62-320
  assign          <- backend/cuda/operations.hpp:822-860 + assign.hpp:14-241
  reduce          <- backend/cuda/operations.hpp:953-1059 + reduce.hpp:13-145
  eWiseAdd        <- backend/cuda/operations.hpp:567-699 + ewiseadd.hpp
  eWiseMult       <- backend/cuda/operations.hpp:331-410 + ewisemult.hpp:32-270

Float `plus` reductions are folded sequentially in stored order; the reference's
GPU order (merge-path partials, CUB trees) is unpinned, so callers compare those
with a relative tolerance (1e-5) and everything else bit-exactly.
"""
import numpy as np

from .semiring import Semiring, Monoid
from . import loader

# graphblas/types.hpp:21-78
GrB_UNKNOWN, GrB_SPARSE, GrB_DENSE = 0, 1, 2
(GrB_SUCCESS, GrB_UNINITIALIZED_OBJECT, GrB_NULL_POINTER, GrB_INVALID_VALUE,
 GrB_INVALID_INDEX, GrB_DOMAIN_MISMATCH, GrB_DIMENSION_MISMATCH, GrB_OUTPUT_NOT_EMPTY,
 GrB_NO_VALUE, GrB_NOT_IMPLEMENTED, GrB_OUT_OF_MEMORY, GrB_INSUFFICIENT_SPACE,
 GrB_INVALID_OBJECT, GrB_INDEX_OUT_OF_BOUNDS, GrB_PANIC) = range(15)
(GrB_MASK, GrB_OUTP, GrB_INP0, GrB_INP1, GrB_MODE, GrB_TA, GrB_TB, GrB_NT,
 GrB_MXVMODE, GrB_TOL, GrB_BACKEND, GrB_NDESCFIELD) = range(12)
GrB_SCMP, GrB_REPLACE, GrB_TRAN, GrB_DEFAULT = 0, 1, 2, 3
GrB_FIXEDROW = 6
GrB_PUSHPULL, GrB_PUSHONLY, GrB_PULLONLY = 10, 11, 12
GrB_CUDA = 14


class Descriptor:
    def __init__(self):
        self.desc_ = [GrB_DEFAULT] * 4 + [GrB_FIXEDROW, 32, 32, 128, GrB_PUSHPULL, 16, GrB_CUDA]
        # default-constructed flags are all zero (descriptor.hpp:17-25) ...
        self.struconly_ = False
        self.opreuse_ = False
        self.earlyexit_ = False
        self.fusedmask_ = False
        self.sort_ = False
        self.switchpoint_ = 0.0
        self.memusage_ = 0.0
        self.max_niter_ = 0
        self.lastmxv_ = GrB_PUSHONLY
        self._loaded = False

    def loadArgs(self, **vm):
        """parseArgs defaults (util.hpp:39-132) then descriptor.hpp:207-287."""
        d = dict(mxvmode=1, switchpoint=0.01, struconly=False, opreuse=False,
                 earlyexit=True, fusedmask=True, sort=True, memusage=1.0,
                 max_niter=10000, nthread=128)
        d.update(vm)
        self.struconly_ = bool(d["struconly"])
        self.opreuse_ = bool(d["opreuse"])
        self.earlyexit_ = bool(d["earlyexit"])
        self.fusedmask_ = bool(d["fusedmask"])
        self.sort_ = bool(d["sort"])
        self.switchpoint_ = float(np.float32(d["switchpoint"]))
        self.memusage_ = float(d["memusage"])
        self.max_niter_ = int(d["max_niter"])
        self.desc_[GrB_MXVMODE] = {0: GrB_PUSHPULL, 1: GrB_PUSHONLY, 2: GrB_PULLONLY}[int(d["mxvmode"])]
        self.desc_[GrB_NT] = int(d["nthread"])
        self._loaded = True
        return GrB_SUCCESS

    def set(self, field, value):
        self.desc_[field] = value
        return GrB_SUCCESS

    def get(self, field):
        return self.desc_[field]

    def toggle(self, field):
        # descriptor.hpp:141-154
        if field < 4:
            if self.desc_[field] != GrB_DEFAULT:
                self.desc_[field] = GrB_DEFAULT
            elif field > 2:
                self.desc_[field] = GrB_TRAN
            else:
                self.desc_[field] = field
        return GrB_SUCCESS

    def struconly(self): return self.struconly_
    def opreuse(self): return self.opreuse_
    def earlyexit(self): return self.earlyexit_
    def fusedmask(self): return self.fusedmask_
    def sort(self): return self.sort_
    def switchpoint(self): return self.switchpoint_


class Vector:
    """backend::Vector<T>: both representations exist at full size n."""

    def __init__(self, nsize, dtype=np.float32):
        self.dtype = np.dtype(dtype).type
        self.nsize_ = int(nsize)
        self.vec_type_ = GrB_UNKNOWN
        self.ratio_ = np.float32(0)
        self.s_ind = np.zeros(nsize, dtype=np.int32)
        self.s_val = np.zeros(nsize + 1, dtype=self.dtype)
        self.s_nvals = 0
        self.d_val = np.zeros(nsize, dtype=self.dtype)
        self.d_nnz = 0
        self.nvals_ = 0

    # --- container API -----------------------------------------------------
    def build_sparse(self, indices, values):
        n = len(indices)
        self.vec_type_ = GrB_SPARSE          # vector.hpp:154, before sparse_.build can fail
        if n > self.nsize_:
            return GrB_PANIC
        if self.s_nvals > 0:
            return GrB_OUTPUT_NOT_EMPTY
        self.s_ind[:n] = np.asarray(indices, dtype=np.int32)
        self.s_val[:n] = np.asarray(values, dtype=self.dtype)
        self.s_nvals = n
        return GrB_SUCCESS

    def build_dense(self, values):
        n = len(values)
        if n > self.nsize_:
            return GrB_INDEX_OUT_OF_BOUNDS
        self.vec_type_ = GrB_DENSE
        self.d_val[:n] = np.asarray(values, dtype=self.dtype)
        return GrB_SUCCESS

    def fill(self, val):
        self.vec_type_ = GrB_DENSE
        self.d_val[:] = self.dtype(val)
        return GrB_SUCCESS

    def fillAscending(self):
        self.vec_type_ = GrB_DENSE
        self.d_val[:] = np.arange(self.nsize_).astype(self.dtype)
        return GrB_SUCCESS

    def setElement(self, val, index):
        if self.vec_type_ == GrB_DENSE:
            self.d_val[index] = self.dtype(val)
        elif self.vec_type_ == GrB_SPARSE:
            self.s_ind[self.s_nvals] = index
            self.s_val[self.s_nvals] = self.dtype(val)
            self.s_nvals += 1
        else:
            return GrB_UNINITIALIZED_OBJECT
        return GrB_SUCCESS

    def nvals(self):
        """Vector::nvals (vector.hpp:133-146): the active representation's count, cached in nvals_;
        with no storage type the cached value of the last call is returned (0 after clear())."""
        if self.vec_type_ == GrB_SPARSE:
            self.nvals_ = self.s_nvals
        elif self.vec_type_ == GrB_DENSE:
            self.nvals_ = self.nsize_   # DenseVector::nvals == size (dense_vector.hpp:122)
        return self.nvals_

    def size(self):
        return self.nsize_

    def getStorage(self):
        return self.vec_type_

    def setStorage(self, t):
        self.vec_type_ = t

    def dup(self, rhs):
        self.vec_type_ = rhs.vec_type_
        if rhs.vec_type_ == GrB_SPARSE:
            self.s_ind[:] = rhs.s_ind
            self.s_val[:] = rhs.s_val
            self.s_nvals = rhs.s_nvals
        elif rhs.vec_type_ == GrB_DENSE:
            self.d_val[:] = rhs.d_val
        else:
            return GrB_UNINITIALIZED_OBJECT
        return GrB_SUCCESS

    def resize(self, nsize):
        """Vector::resize (vector.hpp:230-237 -> dense_vector.hpp:286-309 / sparse_vector.hpp:242-277): the
        ACTIVE representation is reallocated for nsize elements keeping its first min(nsize, nvals)
        entries (new dense elements are uninitialised there; zero here)."""
        nsize = int(nsize)
        if self.vec_type_ == GrB_DENSE:
            keep = min(nsize, self.nsize_)
            d = np.zeros(nsize, dtype=self.dtype)
            d[:keep] = self.d_val[:keep]
            self.d_val = d
            self.s_ind = np.zeros(nsize, dtype=np.int32)
            self.s_val = np.zeros(nsize + 1, dtype=self.dtype)
            self.s_nvals = 0
        elif self.vec_type_ == GrB_SPARSE:
            keep = min(nsize, self.s_nvals)
            si = np.zeros(nsize, dtype=np.int32)
            sv = np.zeros(nsize + 1, dtype=self.dtype)
            si[:keep] = self.s_ind[:keep]
            sv[:keep] = self.s_val[:keep]
            self.s_ind, self.s_val, self.s_nvals = si, sv, keep
            self.d_val = np.zeros(nsize, dtype=self.dtype)
        else:
            return GrB_UNINITIALIZED_OBJECT
        self.nsize_ = nsize
        return GrB_SUCCESS

    def clear(self):
        self.vec_type_ = GrB_UNKNOWN
        self.nvals_ = 0
        self.s_nvals = 0
        self.d_val[:] = 0
        return GrB_SUCCESS

    def swap(self, rhs):
        # vector.hpp:428-450: same storage required; ratio_ travels with the contents
        if self.vec_type_ != rhs.vec_type_ or self.vec_type_ == GrB_UNKNOWN:
            return GrB_INVALID_OBJECT
        if self.vec_type_ == GrB_SPARSE:
            self.s_ind, rhs.s_ind = rhs.s_ind, self.s_ind
            self.s_val, rhs.s_val = rhs.s_val, self.s_val
            self.s_nvals, rhs.s_nvals = rhs.s_nvals, self.s_nvals
        else:
            self.d_val, rhs.d_val = rhs.d_val, self.d_val
        self.ratio_, rhs.ratio_ = rhs.ratio_, self.ratio_
        return GrB_SUCCESS

    def extractTuples_dense(self):
        """extractTuples(values, n): a sparse vector is densified with fill 0
        (vector.hpp:208-217)."""
        if self.vec_type_ == GrB_SPARSE:
            self.sparse2dense(self.dtype(0))
        return self.d_val.copy()

    def extractTuples_sparse(self):
        return self.s_ind[:self.s_nvals].copy(), self.s_val[:self.s_nvals].copy()

    # --- representation switches (vector.hpp:291-425) ----------------------
    def computeNnz(self, identity):
        self.d_nnz = int(np.count_nonzero(self.d_val != identity))
        return self.d_nnz

    def convert(self, identity, switchpoint, desc):
        if self.vec_type_ == GrB_SPARSE:
            nvals_t, nsize_t = self.s_nvals, self.nsize_
        elif self.vec_type_ == GrB_DENSE:
            nvals_t, nsize_t = self.computeNnz(identity), self.nsize_
        else:
            return GrB_UNINITIALIZED_OBJECT
        ratio = np.float32(nvals_t) / np.float32(nsize_t)
        sp = np.float32(switchpoint)
        if self.vec_type_ == GrB_SPARSE:
            if ratio > sp and ratio > self.ratio_:
                self.sparse2dense(identity, desc)
            else:
                self.ratio_ = ratio
        else:
            if ratio <= sp and ratio < self.ratio_:
                self.dense2sparse(identity, desc)
            else:
                self.ratio_ = ratio
        return GrB_SUCCESS

    def sparse2dense(self, identity, desc=None):
        if self.vec_type_ == GrB_DENSE:
            return GrB_SUCCESS
        if self.vec_type_ == GrB_UNKNOWN:
            self.vec_type_ = GrB_DENSE
            return GrB_SUCCESS
        n = self.s_nvals
        if desc is None or not desc.opreuse():
            self.d_val[:] = self.dtype(identity)
            if desc is not None and desc.struconly():
                self.d_val[self.s_ind[:n]] = self.dtype(1)
            else:
                self.d_val[self.s_ind[:n]] = self.s_val[:n]
        self.vec_type_ = GrB_DENSE
        self.d_nnz = n
        return GrB_SUCCESS

    def dense2sparse(self, identity, desc):
        if self.vec_type_ == GrB_SPARSE:
            return GrB_INVALID_OBJECT
        keep = np.nonzero(self.d_val != identity)[0].astype(np.int32)
        k = keep.size
        self.s_ind[:k] = keep
        if not desc.struconly():
            self.s_val[:k] = self.d_val[keep]
        self.s_nvals = k
        self.vec_type_ = GrB_SPARSE
        return GrB_SUCCESS


class Matrix:
    """backend::SparseMatrix<T>: host CSR and CSC built from sorted COO."""

    def __init__(self, nrows, ncols, dtype=np.float32):
        self.nrows_, self.ncols_ = int(nrows), int(ncols)
        self.dtype = np.dtype(dtype).type
        self.nvals_ = 0

    def build(self, rows, cols, vals):
        vals = np.asarray(vals).astype(self.dtype)
        self.csrRowPtr, self.csrColInd, self.csrVal = loader.coo2csr(rows, cols, vals, self.nrows_, self.ncols_)
        self.cscColPtr, self.cscRowInd, self.cscVal = loader.coo2csc(rows, cols, vals, self.nrows_, self.ncols_)
        self.nvals_ = int(len(rows))
        return GrB_SUCCESS

    def build_csr(self, ptr, ind, val):
        self.csrRowPtr = np.asarray(ptr, dtype=np.int32)
        self.csrColInd = np.asarray(ind, dtype=np.int32)
        self.csrVal = np.asarray(val).astype(self.dtype)
        self.cscColPtr, self.cscRowInd, self.cscVal = loader.csr2csc(
            self.csrRowPtr, self.csrColInd, self.csrVal, self.nrows_, self.ncols_)
        self.nvals_ = int(self.csrColInd.size)
        return GrB_SUCCESS

    def arrays(self, use_csc):
        if use_csc:
            return self.cscColPtr, self.cscRowInd, self.cscVal, self.ncols_
        return self.csrRowPtr, self.csrColInd, self.csrVal, self.nrows_


def _fold_rows(sr, ptr, prod):
    """w[i] = identity (+) prod[ptr[i]] (+) ... in stored order."""
    n = ptr.size - 1
    out = np.full(n, sr.identity(), dtype=sr.dtype)
    name = sr.monoid.opname
    lens = np.diff(ptr)
    nz = lens > 0
    starts = ptr[:-1][nz]
    if prod.size and name in ("minimum", "maximum", "logical_or", "plus", "multiplies") \
            and not (name in ("plus", "multiplies") and np.issubdtype(sr.dtype, np.floating)):
        uf = {"minimum": np.fmin, "maximum": np.fmax, "plus": np.add,      # fminf / fmaxf: NaN operands ignored
              "multiplies": np.multiply}.get(name)
        if name == "logical_or":
            red = np.add.reduceat((prod != 0).astype(np.int64), starts) > 0
            red = red.astype(sr.dtype)
        else:
            red = uf.reduceat(prod, starts)
        red = sr.add_op(np.full(red.size, sr.identity(), dtype=sr.dtype), red)
        out[nz] = red
        return out
    for i in np.nonzero(nz)[0]:
        acc = sr.identity()
        for x in prod[ptr[i]:ptr[i + 1]]:
            acc = sr.add_op(acc, x)[()]
        out[i] = acc
    return out


def _mask_pass(mask_val, scmp):
    """kernels/assign_dense.hpp:27: `(UseScmp && m == 0) || (!UseScmp && m != 0)`."""
    return (mask_val == 0) if scmp else (mask_val != 0)


# ---------------------------------------------------------------------------
def _spmv(w, mask, accum, sr, A, u, desc):
    scmp = desc.get(GrB_MASK) == GrB_SCMP
    tran = desc.get(GrB_INP0) == GrB_TRAN or desc.get(GrB_INP1) == GrB_TRAN
    ptr, ind, val, nrows = A.arrays(use_csc=tran)
    use_mask = mask is not None
    if use_mask and desc.fusedmask() and sr.add_is_logical_or():
        if mask.getStorage() != GrB_DENSE:
            return GrB_SUCCESS if mask.getStorage() == GrB_SPARSE else GrB_UNINITIALIZED_OBJECT
        m = mask.d_val
        # row skipped when UseScmp ^ (!bool(mask)) is true
        skip = np.logical_xor(scmp, m == 0)
        src = m if desc.opreuse() else u.d_val
        ident = 0 if desc.opreuse() else sr.identity()
        out = np.zeros(nrows, dtype=w.dtype)
        hit = (src[ind] != ident)
        rows_any = np.zeros(nrows, dtype=bool)
        lens = np.diff(ptr)
        nz = lens > 0
        if hit.size:
            rows_any[nz] = np.add.reduceat(hit.astype(np.int64), ptr[:-1][nz]) > 0
        out[rows_any & ~skip] = 1
        w.d_val[:nrows] = out
        return GrB_SUCCESS
    prod = sr.mul_op(val, u.d_val[ind]) if ind.size else np.zeros(0, dtype=sr.dtype)
    tmp = _fold_rows(sr, ptr, prod).astype(w.dtype)
    if use_mask:
        # spmv.hpp:203-212: entries whose mask FAILS become identity
        fail = ~_mask_pass(mask.d_val, scmp)
        tmp[fail] = sr.identity()
    if accum:
        # spmv.hpp:213-220: combined with the SEMIRING's add, not with `accum`
        w.d_val[:nrows] = sr.add_op(w.d_val[:nrows], tmp)
    else:
        w.d_val[:nrows] = tmp
    return GrB_SUCCESS


def _spmspv_merge(w, mask, accum, sr, A, u, desc):
    scmp_desc = desc.get(GrB_MASK) == GrB_SCMP
    tran = desc.get(GrB_INP0) == GrB_TRAN or desc.get(GrB_INP1) == GrB_TRAN
    ptr, ind, val, nrows = A.arrays(use_csc=not tran)   # spmspv.hpp:52-55
    nf = u.s_nvals
    uind = u.s_ind[:nf]
    starts = ptr[uind]
    lens = ptr[uind + 1] - starts
    total = int(lens.sum())
    if total == 0:
        w.s_nvals = 0
        return GrB_SUCCESS
    owner = np.repeat(np.arange(nf), lens)
    off = np.arange(total) - np.repeat(np.cumsum(lens) - lens, lens)
    epos = starts[owner] + off
    dest = ind[epos]
    # mask storage (spmspv.hpp:147-164, :199-216): dense -> applied; sparse -> "not implemented" is
    # printed, nothing is filtered, the masked epilogue still runs; else GrB_UNINITIALIZED_OBJECT
    apply_mask = False
    if mask is not None:
        if mask.getStorage() == GrB_DENSE:
            apply_mask = True
        elif mask.getStorage() != GrB_SPARSE:
            return GrB_UNINITIALIZED_OBJECT
    if desc.struconly():
        keys = np.unique(dest).astype(np.int32)
        if apply_mask:
            keep = _mask_pass(mask.d_val[keys], scmp_desc)
            keys = keys[keep]
        w.s_ind[:keys.size] = keys
        w.s_nvals = int(keys.size)
        return GrB_SUCCESS
    aval = val[epos]
    uval = u.s_val[:nf][owner]
    ident = sr.identity()
    # kernels/ewisemult.hpp:11-30: identity short-circuit, mul_op(A_val, u_val)
    prod = np.where((aval == ident) | (uval == ident), ident, sr.mul_op(aval, uval)).astype(sr.dtype)
    order = np.argsort(dest, kind="stable")
    dsort = dest[order]
    psort = prod[order]
    keys, first = np.unique(dsort, return_index=True)
    seg_ptr = np.concatenate([first, [dsort.size]]).astype(np.int64)
    # ReduceByKey(identity, add): fold each key's values
    vals = _fold_rows(sr, seg_ptr, psort)
    keys = keys.astype(np.int32)
    if mask is not None:
        # spmspv.hpp:201-243: failing entries are set to 0, then every 0 is pruned
        vals = vals.copy()
        if apply_mask:
            fail = ~_mask_pass(mask.d_val[keys], scmp_desc)
            vals[fail] = 0
        keep = vals != 0
        keys, vals = keys[keep], vals[keep]
    w.s_ind[:keys.size] = keys
    w.s_val[:keys.size] = vals.astype(w.dtype)
    w.s_nvals = int(keys.size)
    return GrB_SUCCESS


def _mxv_common(w, mask, accum, sr, A, u, desc, is_vxm):
    if is_vxm:
        if desc.get(GrB_INP0) != GrB_DEFAULT:
            return GrB_INVALID_VALUE
        desc.toggle(GrB_INP1)
    else:
        if desc.get(GrB_INP1) != GrB_DEFAULT:
            return GrB_INVALID_VALUE
    mode = desc.get(GrB_MXVMODE)
    ident = sr.identity()
    if mode == GrB_PUSHPULL:
        u.convert(ident, desc.switchpoint(), desc)
    elif mode == GrB_PUSHONLY and u.getStorage() == GrB_DENSE:
        u.dense2sparse(ident, desc)
    elif mode == GrB_PULLONLY and u.getStorage() == GrB_SPARSE:
        u.sparse2dense(ident, desc)
    if u.getStorage() == GrB_SPARSE:
        w.setStorage(GrB_SPARSE)
        info = _spmspv_merge(w, mask, accum, sr, A, u, desc)
        desc.lastmxv_ = GrB_PUSHONLY
    else:
        if is_vxm:
            w.setStorage(GrB_DENSE)
        else:
            w.sparse2dense(ident, desc)
        info = _spmv(w, mask, accum, sr, A, u, desc)
        desc.lastmxv_ = GrB_PULLONLY
    if is_vxm:
        desc.toggle(GrB_INP1)
    return info


def vxm(w, mask, accum, sr, u, A, desc):
    if w is None or u is None or A is None or desc is None:
        return GrB_UNINITIALIZED_OBJECT
    if u.nvals() == 0:
        return GrB_UNINITIALIZED_OBJECT          # operations.hpp:71-74
    if A.nrows_ != u.size() or A.ncols_ != w.size() or (mask is not None and mask.size() != w.size()):
        return GrB_DIMENSION_MISMATCH
    return _mxv_common(w, mask, accum, sr, A, u, desc, True)


def mxv(w, mask, accum, sr, A, u, desc):
    if w is None or u is None or A is None or desc is None:
        return GrB_UNINITIALIZED_OBJECT
    if u.nvals() == 0:
        return GrB_UNINITIALIZED_OBJECT          # operations.hpp:111-114
    if A.ncols_ != u.size() or A.nrows_ != w.size() or (mask is not None and mask.size() != w.size()):
        return GrB_DIMENSION_MISMATCH
    return _mxv_common(w, mask, accum, sr, A, u, desc, False)


# ---------------------------------------------------------------------------
def assign(w, mask, accum, val, desc):
    """assign(w, mask, accum, val, GrB_ALL, n, desc): constant assign under a mask."""
    scmp = desc.get(GrB_MASK) == GrB_SCMP
    t = w.getStorage()
    if t == GrB_DENSE:
        if mask is None:
            return GrB_SUCCESS                      # prints "not implemented", no-op
        mt = mask.getStorage()
        if mt == GrB_DENSE:
            w.d_val[_mask_pass(mask.d_val, scmp)] = w.dtype(val)
        elif mt == GrB_SPARSE:
            if not scmp:                            # SCMP variant prints, no-op
                w.d_val[mask.s_ind[:mask.s_nvals]] = w.dtype(val)
        else:
            return GrB_UNINITIALIZED_OBJECT
        return GrB_SUCCESS
    if t == GrB_SPARSE:
        mt = mask.getStorage()
        if mt == GrB_SPARSE:
            mask.convert(mask.dtype(0), 0.3, desc)
            mt = mask.getStorage()
        n = w.s_nvals
        if mt == GrB_DENSE:
            hit = _mask_pass(mask.d_val[w.s_ind[:n]], scmp)
            w.s_val[:n][hit] = w.dtype(val)
        elif mt != GrB_SPARSE:
            return GrB_UNINITIALIZED_OBJECT
        keep = w.s_val[:n] != w.dtype(val)          # assign.hpp:213-233: prune == val
        k = int(keep.sum())
        w.s_ind[:k] = w.s_ind[:n][keep]
        w.s_val[:k] = w.s_val[:n][keep]
        w.s_nvals = k
        return GrB_SUCCESS
    return GrB_SUCCESS


def reduce_vector(monoid, u, desc):
    """reduce(T* val, accum, MonoidT, Vector u, desc) -> host scalar."""
    t = u.getStorage()
    if t == GrB_SPARSE:
        if desc.struconly():
            return u.dtype(u.s_nvals)               # reduce.hpp:71-72
        arr = u.s_val[:u.s_nvals]
    elif t == GrB_DENSE:
        arr = u.d_val
    else:
        raise ValueError("GrB_UNINITIALIZED_OBJECT")
    if arr.size == 0:
        return monoid.identity()
    return monoid.reduce(arr)


def reduce_matrix_rows(w, monoid, A, desc):
    """reduce(Vector w, mask=NULL, accum, MonoidT, Matrix A): row reduction of csrVal."""
    w.setStorage(GrB_DENSE)
    if desc.struconly():
        return GrB_SUCCESS
    out = np.full(A.nrows_, monoid.identity(), dtype=monoid.dtype)
    for i in range(A.nrows_):
        acc = monoid.identity()
        for x in A.csrVal[A.csrRowPtr[i]:A.csrRowPtr[i + 1]]:
            acc = monoid.op(acc, x)[()]
        out[i] = acc
    w.d_val[:] = out.astype(w.dtype)
    w.d_nnz = A.nrows_
    return GrB_SUCCESS


# ---------------------------------------------------------------------------
def eWiseAdd(w, mask, accum, sr, u, v, desc):
    """Vector (+) Vector with the semiring's ADD; output always dense, masks and accum
    are ignored (masked variants print an error and leave w untouched)."""
    ut, vt = u.getStorage(), v.getStorage()
    ident = sr.identity()
    if (u is w and ut == GrB_SPARSE) or (v is w and vt == GrB_SPARSE):
        if u is w:
            u.sparse2dense(ident, desc)
            ut = GrB_DENSE
        elif v is w:
            v.sparse2dense(ident, desc)
            vt = GrB_DENSE
    w.setStorage(GrB_DENSE)
    if mask is not None and not (ut == GrB_SPARSE and vt == GrB_SPARSE):
        return GrB_SUCCESS
    if ut == GrB_SPARSE and vt == GrB_SPARSE:
        return GrB_SUCCESS                          # "not implemented", w untouched
    if ut == GrB_DENSE and vt == GrB_DENSE:
        w.d_val[:] = sr.add_op(u.d_val, v.d_val).astype(w.dtype)
        return GrB_SUCCESS
    if ut == GrB_SPARSE and vt == GrB_DENSE:
        sp, de, reverse = u, v, False
    elif ut == GrB_DENSE and vt == GrB_SPARSE:
        sp, de, reverse = v, u, True
    else:
        return GrB_INVALID_OBJECT
    # ewiseadd.hpp:93-156
    if de is not w:
        w.d_val[:] = de.d_val.astype(w.dtype)
    idv = np.full(w.nsize_, ident, dtype=sr.dtype)
    w.d_val[:] = (sr.add_op(idv, w.d_val) if reverse else sr.add_op(w.d_val, idv)).astype(w.dtype)
    k = sp.s_nvals
    idx = sp.s_ind[:k]
    # the sparse kernel reads de's storage; when de IS w it sees the constant pass
    src = w.d_val if de is w else de.d_val
    w.d_val[idx] = sr.add_op(sp.s_val[:k], src[idx]).astype(w.dtype)
    return GrB_SUCCESS


def eWiseAdd_scalar(w, mask, accum, sr, u, val, desc):
    ut = u.getStorage()
    if mask is not None:
        return GrB_NOT_IMPLEMENTED
    if ut == GrB_DENSE:
        w.setStorage(GrB_DENSE)
        if u is not w:
            w.d_val[:] = u.d_val.astype(w.dtype)
        w.d_val[:] = sr.add_op(w.d_val, np.full(w.nsize_, val, dtype=sr.dtype)).astype(w.dtype)
        return GrB_SUCCESS
    if ut == GrB_SPARSE:
        w.setStorage(GrB_DENSE)
        w.d_val[:] = sr.add_op(sr.identity(), sr.dtype(val))[()]
        k = u.s_nvals
        idx = u.s_ind[:k]
        w.d_val[idx] = sr.add_op(u.s_val[:k], w.d_val[idx]).astype(w.dtype)
        return GrB_SUCCESS
    return GrB_INVALID_OBJECT


def _binsearch(arr, n, target):
    lo = int(np.searchsorted(arr[:n], target))
    return lo if lo < n and arr[lo] == target else -1


def eWiseMult(w, mask, accum, sr, u, v, desc):
    """Vector (x) Vector with the semiring's MUL; intersection semantics with the
    identity short-circuit of kernels/ewisemult.hpp."""
    ut, vt = u.getStorage(), v.getStorage()
    ident = sr.identity()
    if ut == GrB_SPARSE and vt == GrB_SPARSE:
        v.setStorage(GrB_DENSE)                     # operations.hpp:361-367 (storage flag only)
        ut, vt = u.getStorage(), v.getStorage()     # both re-read (:369-370): u IS v when aliased
    if ut == GrB_DENSE and vt == GrB_DENSE:
        a, b = u.d_val, v.d_val
        if mask is not None and mask.getStorage() == GrB_SPARSE:
            w.setStorage(GrB_SPARSE)
            k = mask.s_nvals
            idx = mask.s_ind[:k]
            mv = mask.s_val[:k]
            out = np.where(mv != 0, sr.mul_op(a[idx], b[idx]), 0).astype(w.dtype)
            w.s_ind[:k] = idx
            w.s_val[:k] = out
            w.s_nvals = k
            return GrB_SUCCESS
        if mask is not None and mask.getStorage() not in (GrB_DENSE,):
            return GrB_INVALID_OBJECT
        w.setStorage(GrB_DENSE)
        dead = (a == ident) | (b == ident)
        if mask is not None:
            dead |= mask.d_val == 0
        w.d_val[:] = np.where(dead, ident, sr.mul_op(a, b)).astype(w.dtype)
        return GrB_SUCCESS
    if ut == GrB_SPARSE and vt == GrB_DENSE:
        sp, de, reverse = u, v, False
    elif ut == GrB_DENSE and vt == GrB_SPARSE:
        sp, de, reverse = v, u, True
    else:
        return GrB_INVALID_OBJECT
    w.setStorage(GrB_SPARSE)
    k = sp.s_nvals
    sidx, sval = sp.s_ind[:k].copy(), sp.s_val[:k].copy()
    if mask is not None and mask.getStorage() == GrB_SPARSE:
        mk = mask.s_nvals
        out_i = mask.s_ind[:mk].copy()
        out_v = np.zeros(mk, dtype=w.dtype)
        for r in range(mk):
            i = out_i[r]
            if mask.s_val[r] != 0:
                dv = de.d_val[i]
                if dv != ident:
                    f = _binsearch(sidx, k, i)
                    if f != -1:
                        out_v[r] = (sr.mul_op(dv, sval[f]) if reverse else sr.mul_op(sval[f], dv))[()]
        w.s_ind[:mk] = out_i
        w.s_val[:mk] = out_v
        w.s_nvals = mk
        return GrB_SUCCESS
    dv = de.d_val[sidx]
    prod = sr.mul_op(dv, sval) if reverse else sr.mul_op(sval, dv)
    out = np.where(sval != ident, prod, 0).astype(w.dtype)
    if mask is not None and mask.getStorage() == GrB_DENSE:
        out = np.where(mask.d_val[sidx] == 0, ident, out).astype(w.dtype)   # zeroDenseIdentityKernel
    w.s_ind[:k] = sidx
    w.s_val[:k] = out
    w.s_nvals = k
    return GrB_SUCCESS


# ---------------------------------------------------------------------------
def _scatter_gather(w, mask, u, indices, gather):
    """assignScatter / extractGather: backend/cuda/operations.hpp:1170-1253,
    scatter.hpp:85-123, gather.hpp:11-50, kernels/scatter.hpp:24-39, kernels/gather.hpp:9-23."""
    nindices = indices.nvals()
    ut = u.getStorage()
    if indices.getStorage() != ut:
        indices.setStorage(ut)
    if w.getStorage() != ut:
        w.setStorage(ut)
    if mask is not None:
        return GrB_SUCCESS
    if ut == GrB_DENSE:
        uv, iv = u.d_val, indices.d_val
    elif ut == GrB_SPARSE:
        uv, iv = u.s_val, indices.s_val          # the dense buffer of w is written (reference quirk)
    else:
        return GrB_SUCCESS
    idx = iv[:nindices].astype(np.int64)
    ok = (idx >= 0) & (idx < w.nsize_)
    if gather:
        k = np.nonzero(ok)[0]
        w.d_val[k] = uv[idx[k]].astype(w.dtype)
    else:
        w.d_val[idx[ok]] = uv[:nindices][ok].astype(w.dtype)    # duplicates: last writer wins here
    return GrB_SUCCESS


def assignScatter(w, mask, accum, u, indices, desc):
    return _scatter_gather(w, mask, u, indices, False)


def extractGather(w, mask, accum, u, indices, desc):
    return _scatter_gather(w, mask, u, indices, True)


def scatter(w, mask, u, val, desc):
    """Extension op `scatter` (operations.hpp:748-761 -> backend/cuda/operations.hpp:1110-1142,
    scatter.hpp:10-82, kernels/scatter.hpp:7-21): w[(Index)u[k]] = val for every stored u[k]
    with 0 < u[k] < bound (index 0 is skipped by the kernel's `ind > 0`).  bound: the dense
    variant passes u's length where w's belongs (scatter.hpp:38) -- writes past w's end would
    be out of bounds there, so the restatement also bounds by w's size; the sparse variant
    passes w's length.  Masked variants print "not implemented" and do nothing."""
    ut = u.getStorage()
    w.setStorage(GrB_DENSE)
    if ut not in (GrB_SPARSE, GrB_DENSE):
        return GrB_UNINITIALIZED_OBJECT
    if mask is not None:
        return GrB_SUCCESS
    if ut == GrB_DENSE:
        src, bound = u.d_val, min(u.nsize_, w.nsize_)
    else:
        src, bound = u.s_val[:u.s_nvals], w.nsize_
    idx = src.astype(np.int64)
    ok = (idx > 0) & (idx < bound)
    w.d_val[idx[ok]] = w.dtype(val)
    return GrB_SUCCESS


# ---------------------------------------------------------------------------
def tril(A):
    """tril on the host: keep row >= col (backend/cuda/tri.hpp:21-48). Returns a new Matrix."""
    rows = np.repeat(np.arange(A.nrows_), np.diff(A.csrRowPtr))
    keep = A.csrColInd <= rows
    L = Matrix(A.nrows_, A.ncols_, A.dtype)
    L.build(rows[keep], A.csrColInd[keep], A.csrVal[keep])
    return L


def mxm_masked(mask, sr, A, B, desc):
    """C<mask> = A (+.x) B on the mask's nonzeros (spgemm.hpp:22-110, kernels/spgemm.hpp:17-79).
    GrB_INP1 = GrB_TRAN: B's CSR rows play the role of its columns. Returns C's csrVal on the
    mask's structure."""
    tran_a = desc.get(GrB_INP0) == GrB_TRAN
    tran_b = desc.get(GrB_INP1) == GrB_TRAN
    ap, ai, av, _ = A.arrays(use_csc=tran_a)
    bp, bi, bv, _ = B.arrays(use_csc=not tran_b)
    out = np.full(mask.nvals_, sr.identity(), dtype=sr.dtype)
    mp, mi, mv = mask.csrRowPtr, mask.csrColInd, mask.csrVal
    for i in range(mask.nrows_):
        arow = ai[ap[i]:ap[i + 1]]
        aval = av[ap[i]:ap[i + 1]]
        for e in range(mp[i], mp[i + 1]):
            if mv[e] == 0:
                continue
            j = mi[e]
            bcol = bi[bp[j]:bp[j + 1]]
            bval = bv[bp[j]:bp[j + 1]]
            common, ia, ib = np.intersect1d(arow, bcol, return_indices=True)
            acc = sr.identity()
            for x, y in zip(aval[ia], bval[ib]):
                acc = sr.add_op(sr.mul_op(x, y), acc)[()]
            out[e] = acc
    return out


def reduce_matrix(monoid, A_vals, nvals, desc):
    if desc.struconly():
        return monoid.dtype(nvals)
    if nvals == 0:
        return monoid.identity()
    return monoid.reduce(A_vals)


def trace_mxm_transpose(sr, A, B):
    """traceMxmTranspose (operations.hpp:698-711 -> backend/cuda/trace.hpp:10-52,
    kernels/trace.hpp:7-67): sum over rows i of (+)_k mul(A(i,k), B(i,k)); an A entry without a
    partner in B contributes mul(a, identity); B's value goes through an Index-typed temporary
    (truncation towards zero, kernels/trace.hpp:44-46); rows are folded with the semiring's add and
    the row results are then summed with + whatever the semiring (atomicAdd, :60-61)."""
    total = sr.dtype(0)
    ident = sr.identity()
    for i in range(A.nrows_):
        a0, a1 = A.csrRowPtr[i], A.csrRowPtr[i + 1]
        b0, b1 = B.csrRowPtr[i], B.csrRowPtr[i + 1]
        bcols = B.csrColInd[b0:b1]
        acc = ident
        for p in range(a0, a1):
            k = _binsearch(bcols, b1 - b0, A.csrColInd[p])
            bv = np.int32(ident) if k == -1 else np.int32(B.csrVal[b0 + k])
            acc = sr.add_op(acc, sr.mul_op(A.csrVal[p], sr.dtype(bv)))[()]
        total = sr.dtype(total + acc)
    return total
