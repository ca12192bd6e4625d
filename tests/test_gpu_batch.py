"""GPU suite (-m gpu): grb_bfs_batch -- up to 64 traversals in one bit-parallel sweep (the multi-frontier
product) -- gives, per source, exactly the labels of algorithm::bfs: checked against the
fixture of the reference's own SimpleReferenceBfs and against the C oracle on larger graphs."""
import os

import numpy as np
import pytest

from backends import HipBackend, GOLDEN

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def hb():
    return HipBackend()


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "algo_ref.npz"))


def run_batch(hb, A, n, sources, **args):
    g = hb.g
    d = hb.descriptor(**args)
    vs = [g.Vector(n) for _ in sources]
    info, res = g.bfs_batch(vs, A, sources, d)
    assert info == 0, info
    return [hb.dense_values(v) for v in vs], res, d


def test_batch_labels_equal_the_references_fixture(hb, fx):
    g = hb.g
    for case in ("chesapeake.d0", "test_cc.d0", "test_cc.d1", "rmat10.d1", "rmat10.d2", "rmat14.d0", "rmat14.d2",
                 "grid48.d0", "grid30.d2", "small.d0"):
        ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
        n = ptr.size - 1
        A = g.Matrix(n, n)
        assert A.build_csr(ptr, ind, np.ones(ind.size, F)) == 0
        srcs = [int(s) for s in fx[case + "/sources"]]
        for mode in (0, 1, 2):
            got, res, d = run_batch(hb, A, n, srcs, mxvmode=mode, struconly=1, opreuse=1)
            for k, lab in enumerate(got):
                assert np.array_equal(lab, fx["%s/bfs_%d" % (case, k)]), (case, mode, k)
            want_reached = sum(int(np.count_nonzero(fx["%s/bfs_%d" % (case, k)])) for k in range(len(srcs)))
            assert res["reached"] == want_reached, (case, mode)


def test_batch_of_64_on_rmat_and_grid(hb):
    """64 sources (with repeats and isolated vertices among them) on RMAT-16 symmetric / directed and a thinned
    grid; every label vector against the C oracle's SimpleReferenceBfs; the totals against the labels."""
    from oracle import simple_reference as sr
    from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges
    g = hb.g
    graphs = []
    s, d, n = rmat_edges(16, 16, seed=41)
    graphs.append(("rmat16_sym", finalize_edges(s, d, n, symmetrize=True)))
    graphs.append(("rmat16_dir", finalize_edges(s, d, n, symmetrize=False)))
    s, d, n = grid_edges(200, keep=0.62, seed=42)
    graphs.append(("grid200", finalize_edges(s, d, n, symmetrize=True)))
    rng = np.random.default_rng(43)
    for name, gr in graphs:
        ptr, ind = gr["csr"]
        n = gr["n"]
        deg = np.diff(ptr)
        A = g.Matrix(n, n)
        cptr, cind = gr["csc"]
        assert A.build_csr(ptr, ind, np.ones(ind.size, F), csc=(cptr, cind, np.ones(cind.size, F))) == 0
        srcs = [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, n, 61)] + [int(np.argmax(deg)), 7]
        want = {s_: sr.bfs(ptr, ind, s_)[0] for s_ in set(srcs)}
        for mode in (0, 1, 2):
            if mode == 1 and name == "rmat16_sym":
                pass
            got, res, d = run_batch(hb, A, n, srcs, mxvmode=mode)
            for s_, lab in zip(srcs, got):
                assert np.array_equal(lab, want[s_]), (name, mode, s_)
            assert res["reached"] == sum(int(np.count_nonzero(want[s_])) for s_ in srcs)
            assert res["edges_traversed"] == sum(int(deg[want[s_] != 0].sum()) for s_ in srcs)


def test_batch_hub_rows_are_sliced(hb):
    """A graph whose hub rows are longer than the 4096-entry slice (a few stars joined by a path) plus pendant
    chains: exercises the slice kernels in both directions."""
    from oracle import simple_reference as sr
    from graphblast_amd.graphgen import finalize_edges
    g = hb.g
    n = 40000
    rng = np.random.default_rng(5)
    src, dst = [], []
    hubs = [0, 1, 2]
    for h, cnt in zip(hubs, (20000, 9000, 5000)):
        leaves = rng.choice(np.arange(3, n), cnt, replace=False)
        src += [h] * cnt
        dst += leaves.tolist()
    src += [0, 1] + list(range(3, 3000))
    dst += [1, 2] + list(range(4, 3001))
    gr = finalize_edges(np.array(src), np.array(dst), n, symmetrize=True)
    ptr, ind = gr["csr"]
    assert np.diff(ptr).max() >= 4096
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, np.ones(ind.size, F)) == 0
    srcs = [0, 2, 2999, 17, 39999, 5, 1] + [int(x) for x in rng.integers(0, n, 20)]
    want = {s_: sr.bfs(ptr, ind, s_)[0] for s_ in set(srcs)}
    for mode in (0, 1, 2):
        got, res, d = run_batch(hb, A, n, srcs, mxvmode=mode)
        for s_, lab in zip(srcs, got):
            assert np.array_equal(lab, want[s_]), (mode, s_)


def test_batch_max_niter_cap_and_errors(hb, fx):
    """--max_niter caps the levels exactly as in algorithm::bfs (the vertices the last allowed iteration
    discovers stay 0); argument checks."""
    from oracle import simple_reference as sr
    g = hb.g
    case = "grid48.d0"
    ptr, ind = fx[case + "/csr_ptr"], fx[case + "/csr_ind"]
    n = ptr.size - 1
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, np.ones(ind.size, F)) == 0
    srcs = [int(s) for s in fx[case + "/sources"]]
    for cap in (1, 2, 5):
        got, res, d = run_batch(hb, A, n, srcs, mxvmode=0, max_niter=cap)
        for s_, lab in zip(srcs, got):
            v1 = g.Vector(n)
            d1 = hb.descriptor(mxvmode=0, max_niter=cap)
            assert g.bfs(v1, A, s_, d1, fused=True)[0] == 0
            assert np.array_equal(lab, hb.dense_values(v1)), (cap, s_)
            full = sr.bfs(ptr, ind, s_)[0]
            assert np.array_equal(lab, np.where(full <= cap, full, 0)), (cap, s_)
    d = hb.descriptor()
    assert g.bfs_batch([g.Vector(n)], A, [n], d)[0] == g.GrB_INVALID_INDEX
    assert g.bfs_batch([g.Vector(n + 1)], A, [0], d)[0] == g.GrB_DIMENSION_MISMATCH
    assert g.bfs_batch([g.Vector(n) for _ in range(65)], A, [0] * 65, d)[0] == g.GrB_INVALID_VALUE


def test_light_levels_in_one_launch_equal_the_host_loop(hb):
    """The levels grb_bfs_batch runs inside its one co-resident launch (every live source pushed, few out-edges:
    the first level, the tail, every level of a high-diameter graph) against the same sweep with the launch
    switched off: labels, level count and totals identical -- with the default limit (the whole grid sweep is one
    launch), with limits so small that the launch hands levels back to the host loop and is entered again and again
    (kept and directly-labelled levels interleave, the rotating word arrays change hands), and under iteration caps
    that end inside the launch."""
    from oracle import simple_reference as sr
    from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges
    g = hb.g
    graphs = []
    s, d, n = grid_edges(160, keep=0.62, seed=5)
    graphs.append(("grid160", finalize_edges(s, d, n, symmetrize=True)))
    s, d, n = rmat_edges(15, 12, seed=6)
    graphs.append(("rmat15", finalize_edges(s, d, n, symmetrize=True)))
    s, d, n = rmat_edges(14, 8, seed=7)
    graphs.append(("rmat14_dir", finalize_edges(s, d, n, symmetrize=False)))
    rng = np.random.default_rng(8)
    before = g.bfs_batch_set_tail(-1)
    try:
        for name, gr in graphs:
            ptr, ind = gr["csr"]
            cptr, cind = gr["csc"]
            n = gr["n"]
            deg = np.diff(ptr)
            A = g.Matrix(n, n)
            assert A.build_csr(ptr, ind, np.ones(ind.size, F), csc=(cptr, cind, np.ones(cind.size, F))) == 0
            srcs = [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, n, 62)] + [int(np.argmax(deg))]
            want = {s_: sr.bfs(ptr, ind, s_)[0] for s_ in set(srcs)}
            depth = max(int(w.max()) for w in want.values())
            for cap in (None, 3, max(2, depth // 2), depth - 1, depth):
                args = dict(mxvmode=0, struconly=1, opreuse=1)
                if cap is not None:
                    args["max_niter"] = cap
                g.bfs_batch_set_tail(0)
                ref, ref_res, _ = run_batch(hb, A, n, srcs, **args)
                if cap is None:
                    for s_, lab in zip(srcs, ref):
                        assert np.array_equal(lab, want[s_]), (name, s_)
                for limit in (1 << 20, 4096, 96, 7):
                    g.bfs_batch_set_tail(limit)
                    for rep in range(2):                        # the second sweep starts from the arrays the first left clean
                        got, res, _ = run_batch(hb, A, n, srcs, **args)
                        for k, (x, y) in enumerate(zip(got, ref)):
                            assert np.array_equal(x, y), (name, cap, limit, rep, k)
                        assert res["levels"] == ref_res["levels"], (name, cap, limit)
                        assert res["reached"] == ref_res["reached"] and res["edges_traversed"] == ref_res["edges_traversed"]
    finally:
        g.bfs_batch_set_tail(before)
