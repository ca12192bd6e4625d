"""The 1-D partitioned BFS with its level loop on the device (csrc/bfs_part_run.hip): one co-resident launch per
level, the all-gather of the new-bits bitmaps, nothing read back until the end.  On the one-GPU box every rank of a
world of 1 .. 8 lives on the same device (grb_bfs_part_run_group: device copies stand in for RCCL); labels bit-exact
against the oracle's SimpleReferenceBfs restatement, direction trace against the oracle's accounting run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(seed=3, scale=15, sym=True, ef=8):
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    s, d, n = rmat_edges(scale, ef, seed=seed)
    return finalize_edges(s, d, n, symmetrize=sym)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).to(dev)


def _check(labels, res, trace, ptr, ind, src, mode, switchpoint, csc=None, check_trace=True):
    from oracle import simple_reference as sr
    want = sr.bfs(ptr, ind, src)[0]
    assert np.array_equal(labels, want), (src, mode)
    deg = np.diff(ptr)
    for r in res:
        assert r["edges_traversed"] == int(deg[want != 0].sum()) and r["reached"] == int(np.count_nonzero(want))
        assert r["levels"] == res[0]["levels"] and r["launches"] == res[0]["launches"]
    if check_trace:
        cp, ci = csc if csc is not None else (ptr, ind)
        _, stats = sr.bfs_do_stats(ptr, ind, cp, ci, src, mxvmode=mode, switchpoint=switchpoint)
        assert [t[0] for t in trace] == ["pull" if s[0] else "push" for s in stats], (src, mode)
        assert res[0]["levels"] == len(stats)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_device_loop_simulated_ranks(world):
    from graphblast_amd.dist import LoopbackGroup
    gr = _graph()
    ptr, ind = gr["csr"]
    dev = torch.device("cuda", 0)
    grp = LoopbackGroup(gr["n"], _t(ptr, dev), _t(ind, dev), world, dev)
    hub = int(np.argmax(np.diff(ptr)))
    for mode in (10, 11, 12):
        for src in (hub, 11, 4097):
            labels, res, trace = grp.bfs(src, mxvmode=mode, switchpoint=0.02)
            _check(labels, res, trace, ptr, ind, src, mode, 0.02)
            assert res[0]["launches"] == res[0]["levels"] + 2        # one launch per level, one to see the end, one spare
    # the edge-aware extension: the hub's second frontier is few vertices carrying most of the edges
    labels, res, trace = grp.bfs(hub, switchpoint=0.02, edgeswitch=0.02)
    _check(labels, res, trace, ptr, ind, hub, 10, 0.02, check_trace=False)
    assert trace[1][0] == "pull"


def test_device_loop_directed_graph():
    from graphblast_amd.dist import LoopbackGroup
    gd = _graph(seed=9, scale=13, sym=False)
    ptr, ind = gd["csr"]
    cp, ci = gd["csc"]
    dev = torch.device("cuda", 0)
    for world in (1, 3):
        grp = LoopbackGroup(gd["n"], _t(ptr, dev), _t(ind, dev), world, dev, in_edges=(_t(cp, dev), _t(ci, dev)))
        for mode in (10, 11, 12):
            for src in (int(np.argmax(np.diff(ptr))), 3):
                labels, res, trace = grp.bfs(src, mxvmode=mode, switchpoint=0.02)
                _check(labels, res, trace, ptr, ind, src, mode, 0.02, csc=(cp, ci))


def test_device_loop_long_diameter_and_cap():
    """more levels than kept level bitmaps (32): later levels label at once; and a max_niter that cuts the loop"""
    from graphblast_amd.dist import LoopbackGroup
    from oracle import simple_reference as sr
    side = 48
    idx = np.arange(side * side).reshape(side, side)
    e = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]),
                        np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])], axis=1)
    from graphblast_amd.graphgen import finalize_edges
    gr = finalize_edges(e[0].astype(np.int64), e[1].astype(np.int64), side * side, symmetrize=True)
    ptr, ind = gr["csr"]
    dev = torch.device("cuda", 0)
    for world in (1, 2, 5):
        grp = LoopbackGroup(gr["n"], _t(ptr, dev), _t(ind, dev), world, dev)
        for mode in (10, 11, 12):
            labels, res, trace = grp.bfs(0, mxvmode=mode, switchpoint=0.01)
            _check(labels, res, trace, ptr, ind, 0, mode, 0.01)
            assert res[0]["levels"] > 64
        want = sr.bfs(ptr, ind, 5)[0]
        for cap in (1, 2, 7, 40):
            labels, res, trace = grp.bfs(5, max_niter=cap)
            assert np.array_equal(labels, np.where(want <= cap, want, 0)), (world, cap)
            assert res[0]["levels"] == cap and res[0]["hit_cap"] == 1
            assert res[0]["reached"] == int(np.count_nonzero(np.where(want <= cap, want, 0)))


def test_device_loop_ragged_sizes_and_empty_ranks():
    """n not a multiple of 64, a heavy tail vertex (the advisor's n = 190 case: an interior bound inside a word),
    and more ranks than the graph can feed (empty ranks)"""
    from graphblast_amd.dist import LoopbackGroup, partition_bounds
    from graphblast_amd.graphgen import finalize_edges
    rng = np.random.default_rng(4)
    n = 190
    s = rng.integers(0, n, 600)
    d = rng.integers(0, n, 600)
    s = np.concatenate([s, np.full(150, n - 1)])                   # vertex n - 1 is a hub
    d = np.concatenate([d, rng.integers(0, n - 1, 150)])
    gr = finalize_edges(s.astype(np.int64), d.astype(np.int64), n, symmetrize=True)
    ptr, ind = gr["csr"]
    dev = torch.device("cuda", 0)
    for world in (2, 4, 8):
        b = partition_bounds(ptr, world)
        grp = LoopbackGroup(n, _t(ptr, dev), _t(ind, dev), world, dev)
        for mode in (10, 11, 12):
            for src in (0, n - 1, 97):
                labels, res, trace = grp.bfs(src, mxvmode=mode, switchpoint=0.05)
                _check(labels, res, trace, ptr, ind, src, mode, 0.05)
        assert world < 8 or any(b[i] == b[i + 1] for i in range(world))


def test_device_loop_one_rank_levels_per_launch():
    """a world of one rank: one level per launch (what N > 1 runs, minus the collective) and every level in ONE
    launch give the same labels and trace; Partition1D takes the device loop by itself"""
    from graphblast_amd.dist import Partition1D
    gr = _graph(seed=5, scale=16)
    ptr, ind = gr["csr"]
    dev = torch.device("cuda", 0)
    hub = int(np.argmax(np.diff(ptr)))
    for lpl in (1, 2, 1 << 20):
        part = Partition1D(gr["n"], _t(ptr, dev), _t(ind, dev), 0, 1, dev, switchpoint=0.02, levels_per_launch=lpl)
        assert part.device_loop
        for src in (hub, 123):
            res = part.bfs(src)
            labels = part.gather_labels().cpu().numpy()
            _check(labels, [res], res["trace"], ptr, ind, src, 10, 0.02)
            if lpl == 1:
                assert res["launches"] == res["levels"] + 2
            if lpl == 1 << 20:
                assert res["launches"] == 1


def _sssp_rounds(ptr, ind, w, src, max_niter=10000):
    """the synchronous rounds of algorithm/sssp.hpp on the host: distances and the loop counter at exit"""
    n = ptr.size - 1
    fmax = np.finfo(np.float32).max
    d = np.full(n, fmax, dtype=np.float32)
    d[src] = 0
    rows = np.repeat(np.arange(n), np.diff(ptr))
    for it in range(1, max_niter + 1):
        live = d[rows] < fmax
        cand = (d[rows][live] + w[live]).astype(np.float32)
        y = d.copy()
        np.minimum.at(y, ind[live], cand)
        if not (y < d).any():
            return d, it
        d = y
    return d, max_niter + 1


def _edge_weights(ptr, ind, sym=True, seed=0):
    rows = np.repeat(np.arange(ptr.size - 1, dtype=np.int64), np.diff(ptr))
    a, b = (np.minimum(rows, ind.astype(np.int64)), np.maximum(rows, ind.astype(np.int64))) if sym else (rows, ind.astype(np.int64))
    return ((((a * 1000003) ^ (b * 7919 + seed)) * 2654435761 >> 7) % 16 + 1).astype(np.float32)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_device_loop_sssp_simulated_ranks(world):
    """algorithm::sssp in frontier form on the partition: distances bit-exact (integer weights) and the reference's
    loop counter, for an outbox that holds every round's pairs and for one so small that most rounds need several
    launches (the round then continues from the recorded resume points); also under a max_niter that ends the loop"""
    from graphblast_amd.dist import LoopbackGroup
    from oracle import simple_reference as sr
    gr = _graph(seed=3, scale=13)
    ptr, ind = gr["csr"]
    dev = torch.device("cuda", 0)
    wh = _edge_weights(ptr, ind)
    w = torch.from_numpy(wh).to(dev)
    grp = LoopbackGroup(gr["n"], _t(ptr, dev), _t(ind, dev), world, dev)
    hub = int(np.argmax(np.diff(ptr)))
    for src in (hub, 17):
        want_d, want_it = _sssp_rounds(ptr, ind, wh, src)
        assert np.array_equal(want_d, sr.sssp(ptr, ind, wh, src)[0])
        for cap in (65536, 64):
            d, res = grp.sssp(w, src, outbox_pairs=cap)
            assert np.array_equal(d, want_d), (world, src, cap)
            assert all(r["iterations"] == want_it and r["hit_cap"] == 0 for r in res), (res, want_it)
            if cap == 64 and world > 1:
                assert res[0]["launches"] > res[0]["rounds"] + 2          # rounds that took more than one launch
        for mx in (1, 2, 3):
            want_c, want_cit = _sssp_rounds(ptr, ind, wh, src, max_niter=mx)
            d, res = grp.sssp(w, src, max_niter=mx, outbox_pairs=256)
            assert np.array_equal(d, want_c) and res[0]["iterations"] == want_cit, (world, src, mx, res)


def test_device_loop_sssp_directed_and_road_like():
    from graphblast_amd.dist import LoopbackGroup, Partition1D
    from graphblast_amd.graphgen import finalize_edges
    dev = torch.device("cuda", 0)
    gd = _graph(seed=9, scale=12, sym=False)
    ptr, ind = gd["csr"]
    wh = _edge_weights(ptr, ind, sym=False, seed=5)
    src = int(np.argmax(np.diff(ptr)))
    want_d, want_it = _sssp_rounds(ptr, ind, wh, src)
    for world in (2, 5):
        grp = LoopbackGroup(gd["n"], _t(ptr, dev), _t(ind, dev), world, dev, in_edges=(_t(gd["csc"][0], dev), _t(gd["csc"][1], dev)))
        d, res = grp.sssp(torch.from_numpy(wh).to(dev), src, outbox_pairs=512)
        assert np.array_equal(d, want_d) and res[0]["iterations"] == want_it
    # a grid: hundreds of rounds with small frontiers; one rank, one round per launch and many per launch
    side = 40
    idx = np.arange(side * side).reshape(side, side)
    e = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()]),
                        np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])], axis=1)
    gr = finalize_edges(e[0].astype(np.int64), e[1].astype(np.int64), side * side, symmetrize=True)
    ptr, ind = gr["csr"]
    wh = _edge_weights(ptr, ind)
    want_d, want_it = _sssp_rounds(ptr, ind, wh, 0)
    w = torch.from_numpy(wh).to(dev)
    for rpl in (1, 64):
        part = Partition1D(gr["n"], _t(ptr, dev), _t(ind, dev), 0, 1, dev)
        d, info = part.sssp(w, 0, rounds_per_launch=rpl)
        assert np.array_equal(d.cpu().numpy(), want_d) and info["iterations"] == want_it, (rpl, info, want_it)
        assert info["form"].startswith("frontier")
    for world in (3,):
        grp = LoopbackGroup(gr["n"], _t(ptr, dev), _t(ind, dev), world, dev)
        d, res = grp.sssp(w, 0, outbox_pairs=64)
        assert np.array_equal(d, want_d) and res[0]["iterations"] == want_it


# ---- two real PROCESSES drive the device-side loops (world 2 on the one GPU of the box) ---------------------------
def _two_process_worker(rank, world, port, q):
    import os
    import traceback
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphblast_amd.dist import Partition1D, HostStagedComm, bitmap_words
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        gr = _graph(seed=3, scale=14)
        ptr, ind = gr["csr"]
        n = gr["n"]
        comm = HostStagedComm(rank, world, bitmap_words(n), dev)
        hub = int(np.argmax(np.diff(ptr)))
        out = []
        for mode in (10, 11, 12):
            part = Partition1D(n, _t(ptr, dev), _t(ind, dev), rank, world, dev, mxvmode=mode, switchpoint=0.02, comm=comm)
            assert part.device_loop                                   # grb_bfs_part_run, not the Python level loop
            for src in (hub, 11):
                res = part.bfs(src)
                labels = part.gather_labels().cpu().numpy()
                out.append((mode, src, dict(levels=res["levels"], launches=res["launches"], reached=res["reached"],
                                            edges_traversed=res["edges_traversed"]),
                            [tuple(t) for t in res["trace"]], labels))
        wh = _edge_weights(ptr, ind)
        part = Partition1D(n, _t(ptr, dev), _t(ind, dev), rank, world, dev, comm=comm)
        sssp = []
        for cap in (65536, 64):                                      # 64: most rounds take several launches
            d, info = part.sssp(torch.from_numpy(wh).to(dev), hub, outbox_pairs=cap)
            sssp.append((cap, d.cpu().numpy(), info))
        deg = torch.from_numpy(np.maximum(np.diff(ptr), 1).astype(np.float32)).to(dev)
        pvec, pinfo = part.pagerank(deg, alpha=0.85, eps=0.0, max_niter=10)          # grb_pr_part_run
        pvec2, pinfo2 = part.pagerank(deg, alpha=0.85, eps=1e-3, max_niter=50)       # stops on the residual
        pr = (pvec.cpu().numpy(), pinfo, pvec2.cpu().numpy(), pinfo2)
        comm.close()
        if rank == 0:
            q.put(("ok", out, (sssp, pr)))
    except Exception:                                                 # noqa: BLE001 -- the parent must not wait for a dead rank
        q.put(("error", "rank %d: %s" % (rank, traceback.format_exc()), None))
        raise
    finally:
        dist.destroy_process_group()


def test_device_loop_two_processes_host_staged_collectives():
    """grb_bfs_part_run / grb_sssp_part_run with world = 2 driven by two PROCESSES: each enqueues its own launches by
    the "launch k goes out when launch k - 2 has reported" rule and meets the other in a real collective per level /
    round (the library's host-staged transport over a gloo group, csrc/comm.hip), instead of the loop-back group of the
    tests above where one host thread drives every rank.  Each process takes 112 of the GPU's CUs (GRB_NUM_CU) so
    that both co-resident grids fit at once."""
    import os
    import socket
    import torch.multiprocessing as mp
    from oracle import simple_reference as sr
    gr = _graph(seed=3, scale=14)
    ptr, ind = gr["csr"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    before = os.environ.get("GRB_NUM_CU")
    os.environ["GRB_NUM_CU"] = "112"
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_two_process_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        status, out, rest = q.get(timeout=300)
        assert status == "ok", out
        sssp, pr = rest
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        if before is None:
            os.environ.pop("GRB_NUM_CU", None)
        else:
            os.environ["GRB_NUM_CU"] = before
    for mode, src, res, trace, labels in out:
        _check(labels, [res], trace, ptr, ind, src, mode, 0.02)
        assert res["launches"] == res["levels"] + 2
    wh = _edge_weights(ptr, ind)
    hub = int(np.argmax(np.diff(ptr)))
    want_d, want_it = _sssp_rounds(ptr, ind, wh, hub)
    assert np.array_equal(want_d, sr.sssp(ptr, ind, wh, hub)[0])
    for cap, d, info in sssp:
        assert np.array_equal(d, want_d), cap
        assert info["iterations"] == want_it, (cap, info)
    # PageRank: the library's iteration loop over two chunks per rank, slices gathered between the processes
    pvec, pinfo, pvec2, pinfo2 = pr
    assert pinfo["iterations"] == 10 and pinfo["overlapped_chunks"] == 2 and len(pinfo["errors"]) == 10
    want = sr.pr(ptr, ind, 0.85, 0.0, 10)[0]
    rel = np.abs(pvec - want) / np.maximum(np.abs(want), 1e-30)
    assert rel.max() <= 1e-5, rel.max()
    # eps = 1e-3: the loop leaves after the first iteration whose residual (pr.hpp:84-90, rooted) is within eps
    stop = 1 + next(i for i, e in enumerate(pinfo["errors"]) if e <= 1e-3)
    assert pinfo2["iterations"] == stop < 10, (pinfo2["iterations"], stop)
    assert np.allclose(pinfo2["errors"], pinfo["errors"][:stop], rtol=1e-5)      # atomic sums: order may differ
    want2 = sr.pr(ptr, ind, 0.85, 0.0, stop)[0]
    rel2 = np.abs(pvec2 - want2) / np.maximum(np.abs(want2), 1e-30)
    assert rel2.max() <= 1e-5, rel2.max()


def test_bench_n_greater_than_one_path_with_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` end to end on the one-GPU box: two ranks launched as the driver launches them
    (torch.distributed.run), both on cuda:0 with 112 CUs each, a gloo group, the library communicator over its
    host-staged transport (GRB_BENCH_SHARED_GPU=1).  Everything of the N > 1 path of bench.py runs -- the partition, the
    device-side level loop with a real collective per level, max-over-ranks timing, the partitioned PageRank, the
    parity check against every rank's replica, the source-sharded leg, the one JSON line from rank 0 -- except RCCL
    itself."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GRB_BENCH_SHARED_GPU="1", GRB_NUM_CU="112")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "1", "--scale", "16", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=500, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]                       # rank 0 prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0
    assert line["config"]["parallelism"] == "1d_vertex_partition_x2"
    assert line["parity"]["mismatches"] == 0 and line["parity"]["checked_sources"] == 4
    assert line["level_loop"]["where"].startswith("device") and line["level_loop"]["launches_per_traversal"] > 2
    assert line["collectives"]["collectives_per_traversal"] >= 2
    assert line["pagerank_partitioned"]["iterations"] == 10
    assert line["source_sharded_replicas"]["value"] > 0
    assert line["source_sharded_replicas"]["coscheduled_12"]["value"] > 0      # the replica leg, several traversals per launch
    assert "STRONG" in line["scaling_note"] and "WEAK" in line["scaling_note"]
