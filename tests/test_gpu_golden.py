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


def test_binary_cache_round_trip(hb, fx, tmp_path):
    """The reference's .bin interchange format (sparse_matrix.hpp:328-407): what grb_matrix_write_cache writes is
    byte-identical to the restated writer's file (oracle/loader.py, same layout as the reference's ofs.write
    sequence), and grb_matrix_build_cache gives back the same CSR / CSC with every value 1."""
    from oracle import loader
    g = hb.g
    for case in ("chesapeake.d0", "rmat10.d1", "rmat12.d2", "grid48.d0", "test_cc.d1"):
        A, n = matrix(hb, fx, case)
        path = str(tmp_path / (case + ".bin"))
        assert A.write_cache(path) == 0
        ref_path = str(tmp_path / (case + ".ref.bin"))
        loader.write_cache(ref_path, fx[case + "/csr_ptr"], fx[case + "/csr_ind"])
        assert open(path, "rb").read() == open(ref_path, "rb").read(), case
        B = g.Matrix(n, n)
        assert B.build_cache(path) == 0
        assert (B.nrows(), B.nvals()) == (n, int(fx[case + "/nvals"]))
        ptr, ind, val = B.host_csr()
        assert np.array_equal(ptr, fx[case + "/csr_ptr"]) and np.array_equal(ind, fx[case + "/csr_ind"]), case
        assert np.all(val == 1)
        cp, ci, cv = B.host_csc()
        assert np.array_equal(cp, fx[case + "/csc_ptr"]) and np.array_equal(ci, fx[case + "/csc_ind"]), case
        d = hb.descriptor(mxvmode=0, struconly=1, opreuse=1)
        v = g.Vector(n)
        assert g.bfs(v, B, int(fx[case + "/sources"][1]), d, fused=True)[0] == 0
        assert np.array_equal(hb.dense_values(v), fx[case + "/bfs_1"]), case
    C2 = g.Matrix(4, 4)
    assert C2.build_cache(str(tmp_path / "absent.bin")) == g.GrB_NO_VALUE


def test_sparse_matrix_format_csr_only(hb, fx, monkeypatch):
    """GRB_SPARSE_MATRIX_FORMAT=1 (backend::GrB_SPARSE_MATRIX_CSRONLY, read when the matrix is created,
    sparse_matrix.hpp:34): the CSC arrays alias the CSR arrays and -- getSymmetry() being always false (quirk 5)
    -- vxm converts a dense input to sparse (push) and mxv a sparse one to dense (pull) whatever the mxvmode says
    (operations.hpp:131-133, 258-260).  Results equal the default format's; lastmxv shows the forced direction."""
    g = hb.g
    case = "rmat10.d1"                                   # directed: a pull over the aliased arrays would be wrong
    ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
    n = ptr.size - 1
    A0, _ = matrix(hb, fx, case)
    monkeypatch.setenv("GRB_SPARSE_MATRIX_FORMAT", "1")
    A1, _ = matrix(hb, fx, case)
    monkeypatch.delenv("GRB_SPARSE_MATRIX_FORMAT")
    p1, i1, _ = A1.host_csc()
    assert np.array_equal(p1, ptr) and np.array_equal(i1, ind)          # h_csc* == h_csr*
    rng = np.random.default_rng(4)
    x = (rng.random(n) < 0.3).astype(F) * rng.integers(1, 5, n).astype(F)
    for mode in (0, 1, 2):
        outs = []
        for A in (A0, A1):
            d = hb.descriptor(mxvmode=mode)
            u, w = g.Vector(n), g.Vector(n)
            assert u.build(x, n) == 0
            assert g.vxm(w, None, None, "PlusMultiplies", u, A, d) == 0
            wd = g.Vector(n)
            wd.dup(w)
            wd.sparse2dense(0.0, d)
            outs.append((hb.dense_values(wd), d.lastmxv_))
            u2, w2 = g.Vector(n), g.Vector(n)
            idx = np.nonzero(x)[0].astype(np.int32)
            assert u2.build(idx, x[idx], idx.size, None) == 0
            assert g.mxv(w2, None, None, "PlusMultiplies", A, u2, d) == 0
            w2.sparse2dense(0.0, d)
            outs.append((hb.dense_values(w2), d.lastmxv_))
        assert np.allclose(outs[0][0], outs[2][0], rtol=1e-6) and np.allclose(outs[1][0], outs[3][0], rtol=1e-6), mode
        assert outs[2][1] == g.GrB_PUSHONLY and outs[3][1] == g.GrB_PULLONLY, (mode, outs[2][1], outs[3][1])
    # whole drivers on the CSR-only matrix: same labels / distances as the reference's
    w = fx[case + "/weights"]
    monkeypatch.setenv("GRB_SPARSE_MATRIX_FORMAT", "1")
    Aw, _ = matrix(hb, fx, case, vals=w)
    monkeypatch.delenv("GRB_SPARSE_MATRIX_FORMAT")
    for k, src in enumerate(fx[case + "/sources"]):
        for mode in (0, 1, 2):
            d = hb.descriptor(mxvmode=mode, struconly=1, opreuse=1)
            v = g.Vector(n)
            info, res = g.bfs(v, A1, int(src), d, fused=True)
            assert info == 0 and np.array_equal(hb.dense_values(v), fx["%s/bfs_%d" % (case, k)]), (k, mode)
            assert res["reached"] == int(np.count_nonzero(fx["%s/bfs_%d" % (case, k)]))
            d2 = hb.descriptor(mxvmode=mode)
            v2 = g.Vector(n)
            assert g.sssp(v2, Aw, int(src), d2)[0] == 0
            assert np.array_equal(hb.dense_values(v2), fx["%s/sssp_%d" % (case, k)]), (k, mode)


def test_load_balance_mode_switch(hb, fx, monkeypatch):
    """GRB_LOAD_BALANCE_MODE is read on every vxm / mxv (operations.hpp:110, 242): 2 (merge) is the implemented
    push path; 0 returns GrB_NOT_IMPLEMENTED and leaves the descriptor's INP1 toggled (vxm, :161-163); 1 prints
    and reports success without computing (:167-177).  A dense input (pull) is not affected."""
    g = hb.g
    A, n = matrix(hb, fx, "chesapeake.d0")
    idx = np.array([0, 5], np.int32)
    val = np.ones(2, F)

    def sparse_u():
        u = g.Vector(n)
        assert u.build(idx, val, 2, None) == 0
        return u
    d = hb.descriptor(mxvmode=1)
    w = g.Vector(n)
    assert g.vxm(w, None, None, "LogicalOrAnd", sparse_u(), A, d) == 0
    want = hb.sparse_tuples(w)[0]
    monkeypatch.setenv("GRB_LOAD_BALANCE_MODE", "0")
    d0 = hb.descriptor(mxvmode=1)
    assert g.vxm(g.Vector(n), None, None, "LogicalOrAnd", sparse_u(), A, d0) == g.GrB_NOT_IMPLEMENTED
    assert d0.get(g.GrB_INP1) == g.GrB_TRAN
    monkeypatch.setenv("GRB_LOAD_BALANCE_MODE", "1")
    d1 = hb.descriptor(mxvmode=1)
    w1 = g.Vector(n)
    w1.fill(0.0)
    assert g.vxm(w1, None, None, "LogicalOrAnd", sparse_u(), A, d1) == 0
    assert w1.getStorage() == g.GrB_SPARSE and w1.nvals() == 0 and d1.lastmxv_ == g.GrB_PUSHONLY
    dd = hb.descriptor(mxvmode=2)
    ud, wd = g.Vector(n), g.Vector(n)
    x = np.zeros(n, F); x[idx] = 1
    assert ud.build(x, n) == 0
    assert g.vxm(wd, None, None, "LogicalOrAnd", ud, A, dd) == 0
    assert np.array_equal(np.nonzero(hb.dense_values(wd))[0], np.sort(want))
    monkeypatch.setenv("GRB_LOAD_BALANCE_MODE", "2")
    w2 = g.Vector(n)
    assert g.vxm(w2, None, None, "LogicalOrAnd", sparse_u(), A, hb.descriptor(mxvmode=1)) == 0
    assert np.array_equal(hb.sparse_tuples(w2)[0], want)


def test_persistent_launch_falls_back_instead_of_panicking(fx):
    """A one-launch traversal that cannot run (launch refused / grid barrier gave up) is re-run through the
    host-driven level loop (BFS) or the op-by-op rounds (SSSP): same labels / distances and a one-line note on
    stderr, not GrB_PANIC.  Forced here with the test hooks GRB_BFS_FORCE_FALLBACK / GRB_SSSP_FORCE_FALLBACK in a
    fresh process (they are read once); GRB_BFS_COOPERATIVE=1 (hipLaunchCooperativeKernel) is exercised too."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import graphblast_amd as g
fx = np.load("tests/golden/algo_ref.npz")
for case in ("rmat14.d2", "rmat14.d0", "grid48.d0"):
    ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
    n = ptr.size - 1
    A = g.Matrix(n, n); assert A.build_csr(ptr, ind, np.ones(ind.size, np.float32)) == 0
    W = g.Matrix(n, n); assert W.build_csr(ptr, ind, fx[case + "/weights"]) == 0
    for k, src in enumerate(fx[case + "/sources"]):
        d = g.Descriptor(); assert d.loadArgs(mxvmode=0, struconly=1, opreuse=1) == 0
        v = g.Vector(n)
        info, res = g.bfs(v, A, int(src), d, fused=True)
        assert info == 0, info
        assert np.array_equal(v.extractTuples()[1], fx["%s/bfs_%d" % (case, k)]), (case, k)
        d2 = g.Descriptor(); assert d2.loadArgs(mxvmode=0) == 0
        v2 = g.Vector(n)
        assert g.sssp(v2, W, int(src), d2)[0] == 0
        assert np.array_equal(v2.extractTuples()[1], fx["%s/sssp_%d" % (case, k)]), (case, k)
    # the same through tickets, one traversal per launch and four side by side: a launch that is refused leaves its
    # tickets to the wait, which runs the traversal itself
    for co in (1, 4):
        g.bfs_set_coschedule(co)
        d = g.Descriptor(); assert d.loadArgs(mxvmode=0, struconly=1, opreuse=1) == 0
        srcs = [int(x) for x in fx[case + "/sources"]]
        vs = [g.Vector(n) for _ in srcs]
        tk = [g.bfs_enqueue(x, A, s_, d) for x, s_ in zip(vs, srcs)]
        assert all(i == 0 for i, _ in tk)
        for k, ((_, t), x) in enumerate(zip(tk, vs)):
            assert g.bfs_wait(t)[0] == 0
            assert np.array_equal(x.extractTuples()[1], fx["%s/bfs_%d" % (case, k)]), (case, k, co)
    g.bfs_set_coschedule(1)
print("OK")
'''
    for env, note in (({"GRB_BFS_FORCE_FALLBACK": "1", "GRB_SSSP_FORCE_FALLBACK": "1"}, True),
                      ({"GRB_BFS_COOPERATIVE": "1"}, False)):
        e = dict(os.environ)
        e.update(env)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=e)
        assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-1500:]
        if note:
            assert "one-launch BFS unavailable" in out.stderr and "one-launch SSSP unavailable" in out.stderr
        else:
            assert "unavailable" not in out.stderr, out.stderr[-800:]
