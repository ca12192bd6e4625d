"""The work-efficient order of grb_sssp (csrc/sssp_nearfar.hip) against the reference's synchronous rounds
(csrc/sssp_persist.hip, itself pinned to the oracle in test_gpu_algorithms.py / test_gpu_golden.py): the same floats
and the same round count, and the rounds themselves whenever the caller could tell the difference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


def _grid(side, keep, seed, weights):
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import grid_edges, finalize_edges
    dev = torch.device("cuda", 0)
    es, ed, n = grid_edges(side, keep=keep, seed=seed)
    gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
    ptr, ind = gg["csr"]
    nnz = gg["nnz"]
    row = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (ptr[1:] - ptr[:-1]).long())
    lo, hi = torch.minimum(row, ind.long()), torch.maximum(row, ind.long())
    h = (((lo * 1000003) ^ hi) * 2654435761 >> 7) % 64 + 1                     # symmetric: a function of the edge
    w = h.to(torch.float32) if weights == "int" else (h.to(torch.float32) * 0.37 + 0.11)
    A = g.Matrix(n, n)
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), w.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(),
                              w.data_ptr(), keep=(ptr, ind, w)) == 0
    deg = (ptr[1:] - ptr[:-1]).cpu().numpy()
    return g, A, n, deg


def _run(g, A, n, src, mode, max_niter=None, timing=0):
    before = g.sssp_set_nearfar(-2)
    g.sssp_set_nearfar(mode)
    try:
        d = g.Descriptor()
        kw = dict(mxvmode=0, timing=timing)
        if max_niter is not None:
            kw["max_niter"] = max_niter
        assert d.loadArgs(**kw) == 0
        v = g.Vector(n)
        info, res = g.sssp(v, A, src, d)
        assert info == 0
        return v.extractTuples()[1].copy(), res["iterations"], g.sssp_last_order()
    finally:
        g.sssp_set_nearfar(before)


@pytest.mark.parametrize("side,keep,seed", [(64, 0.6, 1), (300, 0.6, 2), (700, 0.8, 3)])
def test_integer_weights_same_distances_and_round_count(side, keep, seed):
    g, A, n, deg = _grid(side, keep, seed, "int")
    for src in (int(np.nonzero(deg)[0][0]), int(np.nonzero(deg)[0][len(np.nonzero(deg)[0]) // 2])):
        d_rounds, it_rounds, o0 = _run(g, A, n, src, 0)
        d_nf, it_nf, o1 = _run(g, A, n, src, -1)
        assert o0 == 0 and o1 >= 1                      # the default really picked the near / far order here
        assert np.array_equal(d_rounds, d_nf)
        assert it_rounds == it_nf
        assert (d_nf == FMAX).sum() > 0 or keep > 0.7   # thinned grids leave unreachable vertices: FLT_MAX, as the reference


def test_float_weights_same_distances_when_forced():
    """the fixed point of d[v] = min fl(d[u] + w) does not depend on the order of the relaxations"""
    g, A, n, deg = _grid(300, 0.7, 5, "float")
    src = int(np.nonzero(deg)[0][7])
    d_rounds, it_rounds, o0 = _run(g, A, n, src, 0)
    d_auto, it_auto, oa = _run(g, A, n, src, -1)
    assert oa == 0 and np.array_equal(d_rounds, d_auto) and it_auto == it_rounds      # not small integers: the rounds run
    d_nf, it_nf, o1 = _run(g, A, n, src, 1)
    assert o1 >= 1
    assert np.array_equal(d_rounds, d_nf)


def test_a_cap_or_a_round_log_takes_the_rounds():
    g, A, n, deg = _grid(200, 0.7, 7, "int")
    src = int(np.nonzero(deg)[0][0])
    d_full, it_full, _ = _run(g, A, n, src, 0)
    cap = it_full // 2
    d_cap_rounds, it_cap_rounds, _ = _run(g, A, n, src, 0, max_niter=cap)
    d_cap_auto, it_cap_auto, o = _run(g, A, n, src, -1, max_niter=cap)
    assert o == 0                                        # the fixed point said "cut off": the rounds produced the answer
    assert np.array_equal(d_cap_rounds, d_cap_auto) and it_cap_rounds == it_cap_auto
    assert not np.array_equal(d_cap_rounds, d_full)
    # exactly enough rounds: still the near / far order
    d_eq, it_eq, o = _run(g, A, n, src, -1, max_niter=it_full)
    assert o >= 1 and it_eq == it_full and np.array_equal(d_eq, d_full)
    # per-round records requested: the rounds
    _, _, o = _run(g, A, n, src, -1, timing=1)
    assert o == 0


def test_dense_graphs_keep_the_rounds_by_default_and_agree_when_forced():
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    dev = torch.device("cuda", 0)
    s, d, n = rmat_edges(14, 16, seed=3, device=dev)
    gr = finalize_edges(s, d, n, symmetrize=True)
    ptr, ind = gr["csr"]
    nnz = gr["nnz"]
    row = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (ptr[1:] - ptr[:-1]).long())
    lo, hi = torch.minimum(row, ind.long()), torch.maximum(row, ind.long())
    w = ((((lo * 1000003) ^ hi) * 2654435761 >> 7) % 16 + 1).to(torch.float32)
    A = g.Matrix(n, n)
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), w.data_ptr(), nnz, ptr.data_ptr(), ind.data_ptr(), w.data_ptr(),
                              keep=(ptr, ind, w)) == 0
    src = int(torch.argmax(ptr[1:] - ptr[:-1]))          # a hub: the whole-wave expansion of wide vertices
    d0, it0, o0 = _run(g, A, n, src, -1)
    assert o0 == 0                                       # 30 entries per row: not road-like
    d1, it1, o1 = _run(g, A, n, src, 1)
    assert o1 >= 1 and np.array_equal(d0, d1) and it0 == it1


_FORMS = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from test_gpu_sssp_nearfar import _grid, _run
g, A, n, deg = _grid(300, 0.6, 2, "int")
src = int(np.nonzero(deg)[0][len(np.nonzero(deg)[0]) // 2])
d0, it0, o0 = _run(g, A, n, src, 0)
d1, it1, o1 = _run(g, A, n, src, -1)
print(json.dumps({"same": bool(np.array_equal(d0, d1)), "it": [it0, it1], "order": [o0, o1], "work": list(g.sssp_last_work())}))
"""


@pytest.mark.parametrize("env", [{"GRB_SSSP_QUEUE": "0"}, {"GRB_SSSP_QUEUE_CAP": "40"}, {}])
def test_bitmap_form_and_the_fallback_when_a_list_is_full(env):
    """the near set as a bitmap walk (the first form), the queues (default), and the queues with lists far too short
    for this graph: the launch reports `full` and the bitmap form produces the answer -- each in its own process (the
    settings are read once)"""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    e = dict(os.environ)
    e.update(env)
    e["PYTHONPATH"] = os.path.dirname(here) + os.pathsep + e.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", _FORMS % here], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["same"] and r["it"][0] == r["it"][1]
    assert r["order"][0] == 0 and r["order"][1] >= 1     # near / far in every case: the fallback is its bitmap form, not the rounds
    assert r["work"][0] > 0


def test_bfs_on_a_road_like_graph_takes_the_queues_and_equals_the_bitmap_kernel():
    """algorithm::bfs on a long-diameter, low-degree graph runs its levels from queues (bfs_queue_run); labels, levels,
    reached and edge totals must be the one-launch bitmap kernel's (which a request for per-level records selects),
    and a max_niter that would cut the loop short must hand the call to the exact loop"""
    import torch
    import graphblast_amd as g
    from graphblast_amd.graphgen import grid_edges, finalize_edges
    dev = torch.device("cuda", 0)
    es, ed, n = grid_edges(400, keep=0.7, seed=11)
    gg = finalize_edges(torch.as_tensor(es).to(dev), torch.as_tensor(ed).to(dev), n, symmetrize=True)
    ptr, ind = gg["csr"]
    val = torch.ones(gg["nnz"], dtype=torch.float32, device=dev)
    A = g.Matrix(n, n)
    assert A.build_device_csr(ptr.data_ptr(), ind.data_ptr(), val.data_ptr(), gg["nnz"], ptr.data_ptr(), ind.data_ptr(),
                              val.data_ptr(), keep=(ptr, ind, val)) == 0
    deg = (ptr[1:] - ptr[:-1]).cpu().numpy()
    d = g.Descriptor()
    assert d.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1) == 0
    v = g.Vector(n)
    for src in (int(np.nonzero(deg)[0][0]), int(np.nonzero(deg)[0][len(np.nonzero(deg)[0]) // 2])):
        info, rq = g.bfs(v, A, src, d, fused=True)                    # the queues (n >= 65536, < 8 entries per row)
        assert info == 0
        lq = v.extractTuples()[1].copy()
        info, rb = g.bfs(v, A, src, d, fused=True, profile=1)         # per-level records: the bitmap kernel
        assert info == 0
        lb = v.extractTuples()[1].copy()
        assert np.array_equal(lq, lb)
        assert rq["levels"] == rb["levels"] and rq["reached"] == rb["reached"] and rq["edges_traversed"] == rb["edges_traversed"]
        assert rq["levels"] > 300 and (lq == 0).sum() > 0             # long diameter; thinned grids leave unreachable vertices
        # cut short: labels and totals of the capped loop, from whichever kernel
        dc = g.Descriptor()
        assert dc.loadArgs(mxvmode=0, struconly=1, opreuse=1, earlyexit=1, max_niter=50) == 0
        info, rc = g.bfs(v, A, src, dc, fused=True)
        lc = v.extractTuples()[1].copy()
        info, rc2 = g.bfs(v, A, src, dc, fused=True, profile=1)
        assert np.array_equal(lc, v.extractTuples()[1]) and rc["levels"] == rc2["levels"] and rc["reached"] == rc2["reached"]
