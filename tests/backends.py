"""Uniform adapters so the reference's test cases (tests/ref_cases.py) can be run against
the oracle (CPU; pins the oracle) and against the HIP path through the C ABI (GPU)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import ops, loader
        from oracle.semiring import Semiring, Monoid
        self.ops, self.loader, self.Semiring, self.Monoid = ops, loader, Semiring, Monoid

    def descriptor(self, load=True, **args):
        d = self.ops.Descriptor()
        if load:
            d.loadArgs(**args)
        return d

    def vector(self, n, dtype=np.float32):
        return self.ops.Vector(n, dtype)

    def build_sparse(self, v, idx, vals):
        return v.build_sparse(idx, vals)

    def build_dense(self, v, vals):
        return v.build_dense(vals)

    def fill(self, v, val):
        return v.fill(val)

    def matrix_from_mtx(self, name, directed=0, dtype=np.float32):
        r, c, v, nr, nc, nv = self.loader.read_mtx(os.path.join(GOLDEN, "data", name), directed, dtype)
        A = self.ops.Matrix(nr, nc, dtype)
        A.build(r, c, v)
        return A

    def matrix_from_csr(self, n, ptr, ind, val, dtype=np.float32):
        A = self.ops.Matrix(n, n, dtype)
        A.build_csr(ptr, ind, val)
        return A

    def host_csr(self, A):
        return A.csrRowPtr, A.csrColInd, A.csrVal

    def storage(self, v):
        return v.getStorage()

    def nvals(self, v):
        return v.nvals()

    def dense_values(self, v):
        return v.extractTuples_dense()

    def sparse_tuples(self, v):
        return v.extractTuples_sparse()

    def sparse2dense(self, v, identity, desc):
        return v.sparse2dense(identity, desc)

    def toggle(self, desc, field):
        return desc.toggle(field)

    def set(self, desc, field, value):
        return desc.set(field, value)

    def lastmxv(self, desc):
        return desc.lastmxv_

    def _sr(self, name, dtype):
        return self.Semiring(name, dtype)

    def vxm(self, w, mask, accum, sr, u, A, desc):
        return self.ops.vxm(w, mask, accum, self._sr(sr, w.dtype), u, A, desc)

    def mxv(self, w, mask, accum, sr, A, u, desc):
        return self.ops.mxv(w, mask, accum, self._sr(sr, w.dtype), A, u, desc)

    def eWiseAdd(self, w, mask, accum, sr, u, v, desc):
        if isinstance(v, self.ops.Vector):
            return self.ops.eWiseAdd(w, mask, accum, self._sr(sr, u.dtype), u, v, desc)
        return self.ops.eWiseAdd_scalar(w, mask, accum, self._sr(sr, u.dtype), u, v, desc)

    def eWiseMult(self, w, mask, accum, sr, u, v, desc):
        return self.ops.eWiseMult(w, mask, accum, self._sr(sr, u.dtype), u, v, desc)

    def reduce(self, monoid, u, desc):
        return 0, self.ops.reduce_vector(self.Monoid(monoid, u.dtype), u, desc)

    def reduce_rows(self, w, monoid, A, desc):
        return self.ops.reduce_matrix_rows(w, self.Monoid(monoid, A.dtype), A, desc)

    def assign(self, w, mask, val, desc):
        return self.ops.assign(w, mask, None, val, desc)


class HipBackend:
    name = "hip"

    def __init__(self):
        import graphblast_amd as g
        from oracle import loader         # host-side ingest only (inputs), never compute
        self.g, self.loader = g, loader

    def descriptor(self, load=True, **args):
        d = self.g.Descriptor()
        if load:
            assert d.loadArgs(**args) == 0
        return d

    def vector(self, n, dtype=np.float32):
        return self.g.Vector(n, dtype)

    def build_sparse(self, v, idx, vals):
        return v.build(idx, vals, len(idx), None)

    def build_dense(self, v, vals):
        return v.build(vals, len(vals))

    def fill(self, v, val):
        return v.fill(val)

    def matrix_from_mtx(self, name, directed=0, dtype=np.float32):
        r, c, v, nr, nc, nv = self.loader.read_mtx(os.path.join(GOLDEN, "data", name), directed, dtype)
        A = self.g.Matrix(nr, nc, dtype)
        assert A.build(r, c, v, nv, None) == 0
        return A

    def matrix_from_csr(self, n, ptr, ind, val, dtype=np.float32):
        A = self.g.Matrix(n, n, dtype)
        assert A.build_csr(ptr, ind, val) == 0
        return A

    def host_csr(self, A):
        return A.host_csr()

    def storage(self, v):
        return v.getStorage()

    def nvals(self, v):
        return v.nvals()

    def dense_values(self, v):
        info, vals = v.extractTuples()
        assert info == 0, info
        return vals

    def sparse_tuples(self, v):
        info, idx, vals = v.extractTuples(sparse=True)
        assert info == 0, info
        return idx, vals

    def sparse2dense(self, v, identity, desc):
        return v.sparse2dense(identity, desc)

    def toggle(self, desc, field):
        return desc.toggle(field)

    def set(self, desc, field, value):
        return desc.set(field, value)

    def lastmxv(self, desc):
        return desc.lastmxv_

    def vxm(self, w, mask, accum, sr, u, A, desc):
        return self.g.vxm(w, mask, accum, sr, u, A, desc)

    def mxv(self, w, mask, accum, sr, A, u, desc):
        return self.g.mxv(w, mask, accum, sr, A, u, desc)

    def eWiseAdd(self, w, mask, accum, sr, u, v, desc):
        return self.g.eWiseAdd(w, mask, accum, sr, u, v, desc)

    def eWiseMult(self, w, mask, accum, sr, u, v, desc):
        return self.g.eWiseMult(w, mask, accum, sr, u, v, desc)

    def reduce(self, monoid, u, desc):
        return self.g.reduce(None, monoid, u, desc)

    def reduce_rows(self, w, monoid, A, desc):
        return self.g.reduce(None, monoid, A, desc, w=w)

    def assign(self, w, mask, val, desc):
        return self.g.assign(w, mask, None, val, None, 0, desc)
