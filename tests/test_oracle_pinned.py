"""CPU suite: the ORACLE restatements (oracle/loader.py, oracle/simple_reference.c) against
tests/golden/algo_ref.npz -- inputs and outputs of the reference's OWN readMtx / coo2csr /
coo2csc / SimpleReference{Bfs,Sssp,Pr,Cc,Tc}, compiled from /root/reference into
oracle/_ref/libsimple_ref*.so and run by tests/golden/make_golden.py.  When those libraries
are present (build container, and the GPU box: they travel with the snapshot) the
restatements are also compared live on larger seeded graphs.  Nothing here touches the HIP path."""
import os

import numpy as np
import pytest

from backends import GOLDEN

F = np.float32


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "algo_ref.npz"))


def cases_of(fx):
    return sorted({k.split("/")[0] for k in fx.files})


def mtx_path(fx, case, tmp_path):
    """The .mtx file of a fixture case: a data/small file, or the stored synthetic input written back
    out with the writer the generator used."""
    name, d = case.rsplit(".d", 1)
    p = os.path.join(GOLDEN, "data", name + ".mtx")
    if os.path.exists(p):
        return p, int(d)
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import _write_mtx
    field, symmetry = (str(x) for x in fx[case + "/in_banner"])
    val = fx[case + "/in_val"] if case + "/in_val" in fx.files else None
    p = str(tmp_path / (case + ".mtx"))
    n = int(fx[case + "/nrows"])
    _write_mtx(p, n, n, fx[case + "/in_src"], fx[case + "/in_dst"], val, field, symmetry)
    return p, int(d)


def test_fixture_covers_the_reference_data_and_synthetic_graphs(fx):
    cs = cases_of(fx)
    assert len(cs) >= 40
    for stem in ("chesapeake", "test_cc", "test_bc", "test_pr", "test_mis", "rmat10", "rmat14", "grid48",
                 "rmat8real", "rmat8int"):
        assert any(c.startswith(stem + ".d") for c in cs), stem


def test_loader_restatement_equals_the_references_loader(fx, tmp_path):
    """read_mtx + coo2csr + coo2csc (util.hpp:363-430, 263-329, 501-572) on every fixture case and
    --directed value: same nvals, same CSR / CSC index arrays, same values -- including what
    removeSelfloop's index-only compaction leaves in the value array of weighted inputs."""
    from oracle import loader
    for case in cases_of(fx):
        path, directed = mtx_path(fx, case, tmp_path)
        r, c, v, nr, nc, nv = loader.read_mtx(path, directed)
        assert (nr, nc, nv) == (int(fx[case + "/nrows"]), int(fx[case + "/ncols"]), int(fx[case + "/nvals"])), case
        ptr, ind, val = loader.coo2csr(r, c, v, nr, nc)
        assert np.array_equal(ptr, fx[case + "/csr_ptr"]), case
        assert np.array_equal(ind, fx[case + "/csr_ind"]), case
        assert np.array_equal(np.asarray(val, F), fx[case + "/csr_val"]), case
        cp, ci, cv = loader.coo2csc(r, c, v, nr, nc)
        assert np.array_equal(cp, fx[case + "/csc_ptr"]), case
        assert np.array_equal(ci, fx[case + "/csc_ind"]), case
        assert np.array_equal(np.asarray(cv, F), fx[case + "/csc_val"]), case


def test_weighted_fixture_exercises_the_values_quirk(fx):
    """The weighted synthetic inputs really contain removed entries, so the case above pins the quirk."""
    for case in ("rmat8real.d0", "rmat8real.d2", "rmat8int.d0"):
        assert int(fx[case + "/nvals"]) < fx[case + "/in_src"].size * (2 if case.endswith("d2") else 1)
        assert np.unique(fx[case + "/csr_val"]).size > 10


def test_simple_reference_restatement_equals_the_references_oracles(fx):
    """oracle/simple_reference.c vs the reference's own test_{bfs,sssp,pr,cc,tc}.hpp outputs: BFS depth
    labels, SSSP distances (integer weights: exact), triangle counts and component labels bit-exact;
    PageRank within 1e-6 relative (same loop, same order: in practice identical)."""
    from oracle import simple_reference as sr
    checked = dict(bfs=0, sssp=0, pr=0, cc=0, tc=0)
    for case in cases_of(fx):
        if case + "/sources" not in fx.files:
            continue
        ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
        n = ptr.size - 1
        w = fx[case + "/weights"]
        for k, src in enumerate(fx[case + "/sources"]):
            depth, sd, _ = sr.bfs(ptr, ind, int(src))
            assert np.array_equal(depth, fx["%s/bfs_%d" % (case, k)]), (case, k)
            assert sd == int(fx["%s/bfs_depth_%d" % (case, k)]), (case, k)
            dist, sd, _ = sr.sssp(ptr, ind, w, int(src))
            assert np.array_equal(dist, fx["%s/sssp_%d" % (case, k)]), (case, k)
            assert sd == int(fx["%s/sssp_depth_%d" % (case, k)]), (case, k)
            checked["bfs"] += 1
            checked["sssp"] += 1
        for it in (10, 100):
            rank, niter, _, _ = sr.pr(ptr, ind, 0.85, 1e-8, it)
            want = fx["%s/pr%d" % (case, it)]
            assert niter == int(fx["%s/pr%d_iter" % (case, it)]), (case, it)
            ok = np.isfinite(want)
            assert np.array_equal(ok, np.isfinite(rank)), case
            assert np.allclose(rank[ok], want[ok], rtol=1e-6, atol=0), (case, it)
            checked["pr"] += 1
        if case + "/cc" in fx.files:
            label, ncomp, _ = sr.cc(ptr, ind)
            assert np.array_equal(label, fx[case + "/cc"]), case
            assert ncomp == int(fx[case + "/cc"].max()), case
            assert sr.tc(ptr, ind)[0] == int(fx[case + "/tc_full"]), case
            rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
            keep = ind <= rows
            lptr = np.zeros(n + 1, np.int32)
            lptr[1:] = np.cumsum(np.bincount(rows[keep], minlength=n))
            assert sr.tc(lptr, ind[keep])[0] == int(fx[case + "/tc_tril"]), case
            checked["cc"] += 1
            checked["tc"] += 1
    assert min(checked.values()) >= 15, checked


def test_known_answers_are_the_fixture(fx):
    """known_answers.json is now produced by the reference's own code (it used to quote SURVEY.md)."""
    import json
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    assert "libsimple_ref" in ka["source"] and "SURVEY.md 8(c) recorded" in ka["source"]
    assert ka["chesapeake"]["bfs_depth"] == fx["chesapeake.d0/bfs_0"].astype(int).tolist()
    assert (ka["chesapeake"]["tc_tril"], ka["chesapeake"]["tc_full"]) == (194, 1164)


def _live():
    from oracle import ref_simple
    if not ref_simple.available():
        pytest.skip("oracle/_ref/libsimple_ref*.so not built (needs /root/reference: make -C oracle ref)")
    return ref_simple


def test_restatement_equals_the_reference_live_on_larger_graphs():
    """Same comparison, run live against the reference-compiled library on RMAT-16 / a 300^2 grid
    (too large to store): every source of a seeded draw."""
    rs = _live()
    from oracle import simple_reference as sr
    from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges, random_sources
    graphs = []
    s, d, n = rmat_edges(16, 16, seed=21)
    graphs.append(finalize_edges(s, d, n, symmetrize=True))
    graphs.append(finalize_edges(s, d, n, symmetrize=False))
    s, d, n = grid_edges(300, keep=0.6, seed=22)
    graphs.append(finalize_edges(s, d, n, symmetrize=True))
    rng = np.random.default_rng(23)
    for gr in graphs:
        ptr, ind = gr["csr"]
        w = rng.integers(1, 65, ind.size).astype(F)
        for src in random_sources(ptr, 4, seed=5):
            assert np.array_equal(sr.bfs(ptr, ind, src)[0], rs.bfs(ptr, ind, src)[0])
            assert np.array_equal(sr.sssp(ptr, ind, w, src)[0], rs.sssp(ptr, ind, w, src)[0])
        a, b = sr.pr(ptr, ind, 0.85, 1e-8, 20)[0], rs.pr(ptr, ind, 0.85, 1e-8, 20)[0]
        ok = np.isfinite(b)
        assert np.allclose(a[ok], b[ok], rtol=1e-6, atol=0)
        if gr["csr"] is gr["csc"]:
            assert np.array_equal(sr.cc(ptr, ind)[0], rs.cc(ptr, ind))
            assert rs.cc_verify(ptr, ind, sr.cc(ptr, ind)[0]) == 0
            if gr["nnz"] < 3_000_000 and gr["n"] > 70000:
                assert sr.tc(ptr, ind)[0] == rs.tc(ptr, ind)


def test_reference_checkers_accept_and_reject_live():
    """SimpleVerify{Cc,Mis,Gc} through the wrapper: CORRECT on valid labellings, INCORRECT on broken ones
    (their verdict is read from what they print; this checks the reading)."""
    rs = _live()
    from oracle import loader
    r, c, v, nr, nc, nv = loader.read_mtx(os.path.join(GOLDEN, "data", "chesapeake.mtx"))
    ptr, ind, _ = loader.coo2csr(r, c, v, nr, nc)
    lab = rs.cc(ptr, ind)
    assert rs.cc_verify(ptr, ind, lab) == 0
    bad = lab.copy()
    bad[3] += 1
    assert rs.cc_verify(ptr, ind, bad) == 1
    m = rs.mis(ptr, ind, 0)
    assert rs.mis_verify(ptr, ind, m) == 0
    col = rs.gc(ptr, ind, 0, 64)
    assert rs.gc_verify(ptr, ind, col) == 0
    badc = col.copy()
    badc[ind[ptr[0]]] = badc[0]
    assert rs.gc_verify(ptr, ind, badc) == 1
