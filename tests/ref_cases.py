"""The reference's unit tests on the hot path, restated over a backend adapter
(tests/backends.py).  Each driver follows its original one to one -- same inputs (the
literals live in tests/golden/ref_tests.json), same operation call, same inline scalar
loop for the expected value -- and returns a list of (got, correct) arrays to compare
exactly (BOOST_ASSERT_LIST, test/test.hpp:31-57).

  test/gvxm.cu:20-231        testVxm{DenseSparse,SparseSparse,SparseSparseDenseMask}
  test/greduce.cu:23-52      testReduce
  test/gewiseadd.cu:23-358   testeWiseAdd*, testeWiseMultVectorSparsemask*
  test/gewisemult.cu:23-350  testeWiseMult*
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F = np.float32
SCMP = 0
MASK = 0


def load_cases():
    return json.load(open(os.path.join(GOLDEN, "ref_tests.json")))


def _vxm_correct_dense(be, A, vec):
    ptr, ind, val = be.host_csr(A)
    n = ptr.size - 1
    correct = np.zeros(n, dtype=F)
    for row in range(n):
        for p in range(ptr[row], ptr[row + 1]):
            correct[ind[p]] += val[p] * F(vec[row])
    return correct


def _vxm_correct_sparse(be, A, vind, vval):
    ptr, ind, val = be.host_csr(A)
    n = ptr.size - 1
    correct = np.zeros(n, dtype=F)
    for i, row in enumerate(vind):
        v = int(vval[i])                       # `graphblas::Index val = vec_val[i]`
        for p in range(ptr[row], ptr[row + 1]):
            correct[ind[p]] += val[p] * F(v)
    return correct


def vxm_dense_sparse(be, mtx, vec):
    A = be.matrix_from_mtx(mtx)
    n = len(vec)
    correct = _vxm_correct_dense(be, A, vec)
    x = be.vector(n)
    be.build_dense(x, vec)
    y = be.vector(n)
    desc = be.descriptor()
    assert be.vxm(y, None, None, "PlusMultiplies", x, A, desc) == 0
    be.sparse2dense(y, 0.0, desc)
    return [(be.dense_values(y), correct)]


def vxm_sparse_sparse(be, mtx, vind, vval, n):
    A = be.matrix_from_mtx(mtx)
    correct = _vxm_correct_sparse(be, A, vind, vval)
    x = be.vector(n)
    be.build_sparse(x, vind, vval)
    y = be.vector(n)
    desc = be.descriptor()
    assert be.vxm(y, None, None, "PlusMultiplies", x, A, desc) == 0
    be.sparse2dense(y, 0.0, desc)
    return [(be.dense_values(y), correct)]


def vxm_sparse_sparse_dense_mask(be, mtx, vind, vval, mask_val, use_mask, n):
    A = be.matrix_from_mtx(mtx)
    correct = _vxm_correct_sparse(be, A, vind, vval)
    mask_val = list(mask_val) + [1.0] * (n - len(mask_val))   # gvxm.cu dup5 lists 10 of 11 entries
    for i in range(n):
        if use_mask == 0 and mask_val[i] == 0:
            correct[i] = 0
        elif use_mask == 1 and mask_val[i] != 0:
            correct[i] = 0
    x = be.vector(n)
    be.build_sparse(x, vind, vval)
    mask = be.vector(n)
    be.build_dense(mask, mask_val)
    y = be.vector(n)
    desc = be.descriptor()
    if use_mask == 1:
        be.set(desc, MASK, SCMP)
    assert be.vxm(y, mask, None, "PlusMultiplies", x, A, desc) == 0
    be.sparse2dense(y, 0.0, desc)
    return [(be.dense_values(y), correct)]


def reduce_rows(be, mtx, correct):
    A = be.matrix_from_mtx(mtx)
    n = len(correct)
    desc = be.descriptor(load=False)           # greduce.cu uses a default-constructed Descriptor
    w = be.vector(n)
    assert be.reduce_rows(w, "Plus", A, desc) == 0
    return [(be.dense_values(w), np.asarray(correct, dtype=F))]


def ewiseadd_dense_dense(be, u_val, v_val):
    n = len(u_val)
    correct = np.asarray(u_val, dtype=F) + np.asarray(v_val, dtype=F)
    u, v, w = be.vector(n), be.vector(n), be.vector(n)
    be.build_dense(u, u_val)
    be.build_dense(v, v_val)
    desc = be.descriptor()
    assert be.eWiseAdd(w, None, None, "PlusMultiplies", u, v, desc) == 0
    return [(be.dense_values(w), correct)]


def ewisemult_sparsemask_dense_dense(be, mask_ind, mask_val, u_val, v_val):
    n = len(v_val)
    cv = np.array([F(u_val[i]) * F(v_val[i]) if mv != 0 else F(0) for i, mv in zip(mask_ind, mask_val)], dtype=F)
    u, v, m, w = be.vector(n), be.vector(n), be.vector(n), be.vector(n)
    be.build_dense(u, u_val)
    be.build_dense(v, v_val)
    be.build_sparse(m, mask_ind, mask_val)
    desc = be.descriptor()
    assert be.eWiseMult(w, m, None, "PlusMultiplies", u, v, desc) == 0
    idx, val = be.sparse_tuples(w)
    return [(idx, np.asarray(mask_ind, dtype=np.int32)), (val, cv)]


def ewiseadd_sparse_dense(be, u_ind, u_val, v_val, swap=None):
    n = len(v_val)
    correct = np.asarray(v_val, dtype=F).copy()
    for i, ind in enumerate(u_ind):
        correct[ind] = F(u_val[i]) + F(v_val[ind])
    u, v = be.vector(n), be.vector(n)
    be.build_sparse(u, u_ind, u_val)
    be.build_dense(v, v_val)
    desc = be.descriptor()
    if swap is None:
        w = be.vector(n)
        assert be.eWiseAdd(w, None, None, "PlusMultiplies", u, v, desc) == 0
        return [(be.dense_values(w), correct)]
    w, a, b = {0: (v, u, v), 1: (v, v, u), 2: (u, u, v), 3: (u, v, u)}[swap]
    assert be.eWiseAdd(w, None, None, "PlusMultiplies", a, b, desc) == 0
    k = len(u_ind)                            # the original compares the first u_nvals entries
    return [(be.dense_values(w)[:k], correct[:k]), (be.dense_values(w), correct)]


def ewiseadd_sparse_sparse_inplace(be, u_ind, u_val, v_ind, v_val, nrows, swap):
    correct = np.zeros(nrows, dtype=F)
    for i, ind in enumerate(u_ind):
        correct[ind] = F(u_val[i])
    for i, ind in enumerate(v_ind):
        correct[ind] += F(v_val[i])
    u, v = be.vector(nrows), be.vector(nrows)
    be.build_sparse(u, u_ind, u_val)
    be.build_sparse(v, v_ind, v_val)
    desc = be.descriptor()
    w, a, b = {0: (v, u, v), 1: (v, v, u), 2: (u, u, v), 3: (u, v, u)}[swap]
    assert be.eWiseAdd(w, None, None, "PlusMultiplies", a, b, desc) == 0
    return [(be.dense_values(w), correct)]


def ewisemult_sparsemask_sparse_dense(be, mask_ind, mask_val, u_ind, u_val, v_val):
    n = len(v_val)
    cv = np.zeros(len(mask_ind), dtype=F)
    for i, ind in enumerate(mask_ind):
        if mask_val[i] != 0 and v_val[ind] != 0:
            for j, ui in enumerate(u_ind):
                if ui == ind:
                    cv[i] = F(u_val[j]) * F(v_val[ind])
                    break
                if ind < ui:
                    break
    u, v, m, w = be.vector(n), be.vector(n), be.vector(n), be.vector(n)
    be.build_sparse(u, u_ind, u_val)
    be.build_dense(v, v_val)
    be.build_sparse(m, mask_ind, mask_val)
    desc = be.descriptor()
    assert be.eWiseMult(w, m, None, "PlusMultiplies", u, v, desc) == 0
    idx, val = be.sparse_tuples(w)
    return [(idx, np.asarray(mask_ind, dtype=np.int32)), (val, cv)]


def ewisemult_dense_dense(be, u_val, v_val, greater=False):
    n = len(u_val)
    a, b = np.asarray(u_val, dtype=F), np.asarray(v_val, dtype=F)
    if greater:
        correct = np.where((a == 0) | (b == 0), F(0), (a > b).astype(F))
    else:
        correct = a * b
    u, v, w = be.vector(n), be.vector(n), be.vector(n)
    be.build_dense(u, u_val)
    be.build_dense(v, v_val)
    desc = be.descriptor()
    assert be.eWiseMult(w, None, None, "PlusGreater" if greater else "PlusMultiplies", u, v, desc) == 0
    return [(be.dense_values(w), correct.astype(F))]


def ewisemult_sparse_dense(be, u_ind, u_val, v_val, greater=False, mask_val=None):
    n = len(v_val)
    cv = np.zeros(len(u_ind), dtype=F)
    for i, ind in enumerate(u_ind):
        if mask_val is not None:
            if mask_val[ind] != 0 and v_val[ind] != 0:
                cv[i] = F(u_val[i]) * F(v_val[ind])
        elif greater:
            cv[i] = F(F(u_val[i]) > F(v_val[ind]))
        else:
            cv[i] = F(u_val[i]) * F(v_val[ind])
    u, v, w = be.vector(n), be.vector(n), be.vector(n)
    be.build_sparse(u, u_ind, u_val)
    be.build_dense(v, v_val)
    m = None
    if mask_val is not None:
        m = be.vector(n)
        be.build_dense(m, mask_val)
    desc = be.descriptor()
    assert be.eWiseMult(w, m, None, "PlusGreater" if greater else "PlusMultiplies", u, v, desc) == 0
    idx, val = be.sparse_tuples(w)
    return [(idx, np.asarray(u_ind, dtype=np.int32)), (val, cv)]


def run_all(be):
    """Yields (label, got, correct) for every literal case of the four test files."""
    C = load_cases()
    g = C["gvxm.cu"]
    out = []

    def add(label, pairs):
        for k, (got, cor) in enumerate(pairs):
            out.append(("%s[%d]" % (label, k), np.asarray(got), np.asarray(cor)))

    add("gvxm.dup1", vxm_dense_sparse(be, "test_cc.mtx", g["dup1"]["vectors"]["vec"]))
    add("gvxm.dup2", vxm_dense_sparse(be, "test_sgm.mtx", g["dup2"]["vectors"]["vec"]))
    add("gvxm.dup3", vxm_sparse_sparse(be, "test_cc.mtx", g["dup3"]["vectors"]["vec_ind"], g["dup3"]["vectors"]["vec_val"], 11))
    add("gvxm.dup4", vxm_sparse_sparse(be, "test_sgm.mtx", g["dup4"]["vectors"]["vec_ind"], g["dup4"]["vectors"]["vec_val"], 20))
    for um in (0, 1):
        v = g["dup5"]["vectors"]
        add("gvxm.dup5.%d" % um, vxm_sparse_sparse_dense_mask(be, "test_cc.mtx", v["vec_ind"], v["vec_val"], v["mask_val"], um, 11))
    v = g["dup6"]["vectors"]
    add("gvxm.dup6", vxm_sparse_sparse_dense_mask(be, "test_sgm.mtx", v["vec_ind"], v["vec_val"], v["mask_val"], 0, 20))

    r = C["greduce.cu"]
    add("greduce.dup1", reduce_rows(be, "test_cc.mtx", r["dup1"]["vectors"]["correct"]))
    add("greduce.dup2", reduce_rows(be, "test_bc.mtx", r["dup2"]["vectors"]["correct"]))

    a = C["gewiseadd.cu"]
    v = a["dup1"]["vectors"]
    add("gewiseadd.dup1", ewiseadd_dense_dense(be, v["u_val"], v["v_val"]))
    rng = np.random.default_rng(7)                       # dup2: n = 10 000, generated in a loop
    add("gewiseadd.dup2", ewiseadd_dense_dense(be, rng.integers(0, 50, 10000).astype(F), rng.integers(0, 50, 10000).astype(F)))
    v = a["dup3"]["vectors"]
    add("gewiseadd.dup3", ewisemult_sparsemask_dense_dense(be, v["mask_ind"], v["mask_val"], v["u_val"], v["v_val"]))
    v = a["dup5"]["vectors"]
    add("gewiseadd.dup5", ewiseadd_sparse_dense(be, v["u_ind"], v["u_val"], v["v_val"]))
    v = a["dup7"]["vectors"]
    for sw in range(4):
        add("gewiseadd.dup7.%d" % sw, ewiseadd_sparse_dense(be, v["u_ind"], v["u_val"], v["v_val"], swap=sw))
    v = a["dup9"]["vectors"]
    add("gewiseadd.dup9", ewisemult_sparsemask_sparse_dense(be, v["mask_ind"], v["mask_val"], v["u_ind"], v["u_val"], v["v_val"]))
    v = a["dup11"]["vectors"]
    for sw in range(4):
        add("gewiseadd.dup11.%d" % sw, ewiseadd_sparse_sparse_inplace(be, v["u_ind"], v["u_val"], v["v_ind"], v["v_val"], 10, sw))

    m = C["gewisemult.cu"]
    v = m["dup1"]["vectors"]
    add("gewisemult.dup1", ewisemult_dense_dense(be, v["u_val"], v["v_val"]))
    add("gewisemult.dup2", ewisemult_dense_dense(be, rng.integers(0, 9, 10000).astype(F), rng.integers(0, 9, 10000).astype(F)))
    v = m["dup3"]["vectors"]
    add("gewisemult.dup3", ewisemult_sparsemask_dense_dense(be, v["mask_ind"], v["mask_val"], v["u_val"], v["v_val"]))
    v = m["dup5"]["vectors"]
    add("gewisemult.dup5", ewisemult_sparse_dense(be, v["u_ind"], v["u_val"], v["v_val"]))
    v = m["dup7"]["vectors"]
    add("gewisemult.dup7", ewisemult_sparsemask_sparse_dense(be, v["mask_ind"], v["mask_val"], v["u_ind"], v["u_val"], v["v_val"]))
    v = m["dup9"]["vectors"]
    add("gewisemult.dup9", ewisemult_sparse_dense(be, v["u_ind"], v["u_val"], v["v_val"], mask_val=v["mask_val"]))
    v = m["dup11"]["vectors"]
    add("gewisemult.dup11", ewisemult_dense_dense(be, v["u_val"], v["v_val"], greater=True))
    add("gewisemult.dup12", ewisemult_dense_dense(be, rng.integers(0, 9, 10000).astype(F), rng.integers(0, 9, 10000).astype(F), greater=True))
    v = m["dup13"]["vectors"]
    add("gewisemult.dup13", ewisemult_sparse_dense(be, v["u_ind"], v["u_val"], v["v_val"], greater=True))
    return out
