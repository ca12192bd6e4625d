"""GPU suite (-m gpu): the HIP path through the C ABI against tests/golden/algo_ref.npz -- outputs of the
reference's OWN loader and CPU oracles (readMtx / coo2csr / SimpleReference{Bfs,Sssp,Pr,Cc,Tc} compiled from
/root/reference, tests/golden/make_golden.py).  BFS labels, CC labels, TC counts, SSSP distances on integer
weights and the loaded CSR / CSC bit-exact; PageRank within 1e-5 relative (north_star's bar)."""
import os

import numpy as np
import pytest

from backends import HipBackend, GOLDEN
from test_oracle_pinned import cases_of, mtx_path

pytestmark = pytest.mark.gpu
F = np.float32
FLT_MAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "algo_ref.npz"))


def square_cases(fx):
    return [c for c in cases_of(fx) if c + "/sources" in fx.files and int(fx[c + "/nvals"]) > 0]


def matrix(hb, fx, case, vals=None, dtype=F):
    ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
    n = ptr.size - 1
    A = hb.g.Matrix(n, n, dtype)
    if vals is None:
        vals = np.ones(ind.size, dtype=dtype)
    assert A.build_csr(ptr, ind, np.asarray(vals, dtype)) == 0
    return A, n


def test_device_mtx_loader_equals_the_references_loader(hb, fx, tmp_path):
    """grb_matrix_load_mtx (text parsed, symmetrised, sorted, de-duplicated on the device) against readMtx +
    coo2csr + coo2csc of the reference: every data/small file x --directed 0/1/2 and the synthetic pattern /
    integer / real inputs.  Index arrays bit-exact everywhere; values bit-exact on pattern inputs and on
    weighted inputs where nothing was removed (where entries were removed the reference leaves the value array
    uncompacted, util.hpp:311-323 -- the device loader keeps values with their entries; covered below)."""
    g = hb.g
    n_checked = 0
    for case in cases_of(fx):
        if int(fx[case + "/nvals"]) == 0:
            continue
        path, directed = mtx_path(fx, case, tmp_path)
        A = g.Matrix.from_mtx(path, directed=directed)
        assert (A.nrows(), A.ncols(), A.nvals()) == (int(fx[case + "/nrows"]), int(fx[case + "/ncols"]),
                                                    int(fx[case + "/nvals"])), case
        ptr, ind, val = A.host_csr()
        assert np.array_equal(ptr, fx[case + "/csr_ptr"]) and np.array_equal(ind, fx[case + "/csr_ind"]), case
        cp, ci, cv = A.host_csc()
        assert np.array_equal(cp, fx[case + "/csc_ptr"]) and np.array_equal(ci, fx[case + "/csc_ind"]), case
        if np.all(fx[case + "/csr_val"] == 1):
            assert np.all(val == 1) and np.all(cv == 1), case
        n_checked += 1
    assert n_checked >= 40


def test_bfs_labels_equal_the_references(hb, fx):
    """algorithm::bfs (op by op, the host-driven fused loop and the one-launch traversal; every mxvmode, with and
    without struconly/opreuse) == SimpleReferenceBfs depth labels, bit-exact, from every stored source."""
    g = hb.g
    n_checked = 0
    for case in square_cases(fx):
        A, n = matrix(hb, fx, case)
        for k, src in enumerate(fx[case + "/sources"]):
            want = fx["%s/bfs_%d" % (case, k)]
            for mode in (0, 1, 2):
                for struc in (0, 1):
                    d = hb.descriptor(mxvmode=mode, struconly=struc, opreuse=struc)
                    for fused in (False, True):
                        if not fused and n > 5000 and (mode, struc) != (0, 1):
                            continue                      # the op-by-op loop on the larger cases: one setting
                        v = g.Vector(n)
                        info, res = g.bfs(v, A, int(src), d, fused=fused)
                        assert info == 0
                        assert np.array_equal(hb.dense_values(v), want), (case, k, mode, struc, fused)
                        n_checked += 1
    assert n_checked > 400


def test_bfs_host_driven_loop_equals_the_references(hb, fx, monkeypatch):
    """GRB_BFS_PERSISTENT=0: the level loop driven from the host (bfs_fused.hip), same labels."""
    monkeypatch.setenv("GRB_BFS_PERSISTENT", "0")
    g = hb.g
    for case in ("chesapeake.d0", "rmat14.d2", "rmat14.d0", "grid48.d0", "test_cc.d0"):
        A, n = matrix(hb, fx, case)
        for k, src in enumerate(fx[case + "/sources"]):
            d = hb.descriptor(mxvmode=0, struconly=1, opreuse=1)
            v = g.Vector(n)
            assert g.bfs(v, A, int(src), d, fused=True)[0] == 0
            assert np.array_equal(hb.dense_values(v), fx["%s/bfs_%d" % (case, k)]), (case, k)


def test_sssp_distances_equal_the_references(hb, fx):
    """algorithm::sssp == SimpleReferenceSssp on the stored weights (integers 1..64: every path sum is exact
    in f32, so bit-exact; unreached = FLT_MAX in both)."""
    g = hb.g
    n_checked = 0
    for case in square_cases(fx):
        w = fx[case + "/weights"]
        A, n = matrix(hb, fx, case, vals=w)
        for k, src in enumerate(fx[case + "/sources"]):
            want = fx["%s/sssp_%d" % (case, k)]
            for mode in (0, 1, 2):
                d = hb.descriptor(mxvmode=mode)
                v = g.Vector(n)
                info, res = g.sssp(v, A, int(src), d)
                assert info == 0
                got = hb.dense_values(v)
                assert np.array_equal(got, want), (case, k, mode)
                n_checked += 1
    assert n_checked > 200


def test_pagerank_equals_the_references(hb, fx):
    """algorithm::pr vs SimpleReferencePr after the same number of updates, <= 1e-5 relative.  The CPU oracle
    stops on the squared residual, the driver on its root, so the driver runs with eps = 0 for exactly the
    number of updates the stored run made (min(iter + 1, max_niter))."""
    g = hb.g
    n_checked = 0
    for case in square_cases(fx):
        ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
        n = ptr.size - 1
        deg = np.diff(ptr).astype(F)
        if np.any(deg == 0):
            continue                                   # rank / 0 out-degree: inf / nan in the oracle itself
        rows = np.repeat(np.arange(n), np.diff(ptr))
        vals = (F(1.0) * F(0.85)) / deg[rows]            # gpr.cu:82-90
        A, _ = matrix(hb, fx, case, vals=vals.astype(F))
        for cap in (10, 100):
            want = fx["%s/pr%d" % (case, cap)]
            updates = min(int(fx["%s/pr%d_iter" % (case, cap)]) + 1, cap)
            for mode in (0, 2):
                d = hb.descriptor(mxvmode=mode, max_niter=updates)
                p = g.Vector(n)
                info, res = g.pr(p, A, 0.85, 0.0, d)
                assert info == 0 and res["iterations"] == updates
                got = hb.dense_values(p)
                rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
                assert rel.max() <= 1e-5, (case, cap, mode, rel.max())
                n_checked += 1
    assert n_checked >= 40


def test_cc_and_tc_equal_the_references(hb, fx):
    """algorithm::cc: the same partition as SimpleReferenceCc (labels canonicalised to the smallest member,
    which is what FastSV's parent vector converges to); algorithm::tc on tril(A) == SimpleReferenceTc."""
    from oracle import simple_reference as sr          # cc_canonical: a relabelling helper, no algorithm
    g = hb.g
    n_checked = 0
    for case in square_cases(fx):
        if case + "/cc" not in fx.files:
            continue
        ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
        A, n = matrix(hb, fx, case, dtype=np.int32)
        want = sr.cc_canonical(fx[case + "/cc"])
        for mode in (0, 1, 2):
            d = hb.descriptor(mxvmode=mode)
            v = g.Vector(n, np.int32)
            assert g.cc(v, A, 0, d)[0] == 0
            assert np.array_equal(v.extractTuples()[1], want), (case, mode)
        d = hb.descriptor()
        L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        assert g.tril(L, A, d) == 0
        info, ntris, _ = g.tc(L, B, d)
        assert info == 0 and ntris == int(fx[case + "/tc_tril"]), case
        n_checked += 1
    assert n_checked >= 15


def test_reference_library_at_full_size_if_present(hb):
    """RMAT-20 symmetrised (n = 1 Mi, ~31 M edges): labels of the one-launch traversal against the
    reference-compiled SimpleReferenceBfs itself (oracle/_ref travels with the snapshot), 4 sources."""
    from oracle import ref_simple as rs
    if not rs.available():
        pytest.skip("oracle/_ref/libsimple_ref*.so not in this snapshot")
    from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
    g = hb.g
    s, d, n = rmat_edges(20, 16, seed=31)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, np.ones(ind.size, F)) == 0
    desc = hb.descriptor(mxvmode=0, struconly=1, opreuse=1)
    for src in random_sources(ptr, 4, seed=9):
        v = g.Vector(n)
        assert g.bfs(v, A, src, desc, fused=True)[0] == 0
        assert np.array_equal(hb.dense_values(v), rs.bfs(ptr, ind, src)[0]), src
