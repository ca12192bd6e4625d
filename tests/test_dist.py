"""N > 1 path: graphblast_amd/dist.py under torch.distributed with the gloo backend,
world_size 2 and 3, on CPU (numpy level engine), plus -- on the GPU box -- two simulated
ranks driving the real HIP level steps in one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _graph(seed=1, scale=11, sym=True):
    from graphblast_amd.graphgen import rmat_edges, finalize_edges
    s, d, n = rmat_edges(scale, 8, seed=seed)
    return finalize_edges(s, d, n, symmetrize=sym)


def _weights(ptr, ind):
    """a weight per stored edge that is a function of the (unordered) endpoints: symmetric graphs stay symmetric"""
    rows = np.repeat(np.arange(ptr.size - 1, dtype=np.int64), np.diff(ptr))
    lo, hi = np.minimum(rows, ind.astype(np.int64)), np.maximum(rows, ind.astype(np.int64))
    return ((((lo * 1000003) ^ hi) * 2654435761 >> 7) % 16 + 1).astype(np.float32)


def _directed_weights(major_ptr, minor_ind, major_is_source):
    """a weight per stored entry that depends on the ORDERED pair (source, target): the same edge gets the same
    weight whether it is listed in the CSR (major = source) or in the CSC (major = target)"""
    major = np.repeat(np.arange(major_ptr.size - 1, dtype=np.int64), np.diff(major_ptr))
    minor = minor_ind.astype(np.int64)
    s_, t_ = (major, minor) if major_is_source else (minor, major)
    return ((((s_ * 1000003) ^ (t_ * 7919)) * 2654435761 >> 7) % 16 + 1).astype(np.float32)


def _sssp_rounds(ptr, ind, w, src, max_niter=10000):
    """the synchronous rounds of algorithm/sssp.hpp on the host: distances and the loop counter at exit"""
    n = ptr.size - 1
    fmax = np.finfo(np.float32).max
    d = np.full(n, fmax, dtype=np.float32)
    d[src] = 0
    rows = np.repeat(np.arange(n), np.diff(ptr))
    for it in range(1, max_niter + 1):
        cand = (d[rows] + w).astype(np.float32)                 # relax every stored edge (row -> column)
        y = d.copy()
        np.minimum.at(y, ind, cand)
        if not (y < d).any():
            return d, it
        d = y
    return d, max_niter + 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = False
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from dist_helpers import NumpyEngine
        from graphblast_amd.dist import Partition1D
        gr = _graph()
        ptr, ind = gr["csr"]
        tptr, tind = torch.from_numpy(ptr.astype(np.int64)), torch.from_numpy(ind.astype(np.int64))
        out = []
        for mode in (10, 11, 12):
            part = Partition1D(gr["n"], tptr, tind, rank, world, torch.device("cpu"), engine_cls=NumpyEngine,
                               mxvmode=mode, switchpoint=0.05)
            for src in (int(np.argmax(np.diff(ptr))), 7):
                res = part.bfs(src)
                labels = part.gather_labels().numpy()
                out.append((mode, src, res["levels"], res["edges_traversed"], res["reached"],
                            [t[0] for t in res["trace"]], labels.copy()))
        # PageRank on the partition (config 4 of BASELINE.json at test size)
        part = Partition1D(gr["n"], tptr, tind, rank, world, torch.device("cpu"), engine_cls=NumpyEngine)
        deg = torch.from_numpy(np.diff(ptr).astype(np.float32))
        pvec, info = part.pagerank(deg, alpha=0.85, eps=0.0, max_niter=10)
        # SSSP on the partition (MinimumPlus product over the in-edge shard + all-gather of the slices)
        w = torch.from_numpy(_weights(ptr, ind))
        src_s = int(np.argmax(np.diff(ptr)))
        dvec, sinfo = part.sssp(w, src_s)
        dcap, cinfo = part.sssp(w, src_s, max_niter=2)
        if rank == 0:
            q.put((out, pvec.numpy().copy(), info["iterations"],
                   (src_s, dvec.numpy().copy(), sinfo["iterations"], dcap.numpy().copy(), cinfo["iterations"])))
        ok = True
    except Exception as exc:                       # fail fast instead of letting the parent time out
        import traceback
        q.put(("error", "rank %d: %s" % (rank, traceback.format_exc()), 0, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_bfs_gloo(world):
    from oracle import simple_reference as sr
    gr = _graph()
    ptr, ind = gr["csr"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, pr_vec, pr_iters, sssp_out = q.get(timeout=180)
    assert not (isinstance(out, str) and out == "error"), pr_vec
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    deg = np.diff(ptr)
    # partitioned SSSP: the reference's distances (its own CPU Dijkstra-style oracle) and the rounds' counter,
    # also under a max_niter that cuts the loop short
    src_s, dvec, s_iters, dcap, c_iters = sssp_out
    w = _weights(ptr, ind)
    assert np.array_equal(dvec, sr.sssp(ptr, ind, w, src_s)[0])
    want_d, want_it = _sssp_rounds(ptr, ind, w, src_s)
    assert np.array_equal(dvec, want_d) and s_iters == want_it
    want_c, want_cit = _sssp_rounds(ptr, ind, w, src_s, max_niter=2)
    assert np.array_equal(dcap, want_c) and c_iters == want_cit == 3
    for mode, src, levels, edges, reached, dirs, labels in out:
        want = sr.bfs(ptr, ind, src)[0]
        assert np.array_equal(labels, want), (mode, src)
        assert edges == int(deg[want != 0].sum()) and reached == int(np.count_nonzero(want))
        _, stats = sr.bfs_do_stats(ptr, ind, ptr, ind, src, mxvmode=mode, switchpoint=0.05)
        assert dirs == ["pull" if s[0] else "push" for s in stats], (mode, src)
        assert levels == len(stats)


    want_pr = sr.pr(ptr, ind, 0.85, 0.0, 10)[0]
    assert pr_iters == 10
    rel = np.abs(pr_vec - want_pr) / np.maximum(np.abs(want_pr), 1e-30)
    assert rel.max() <= 1e-5, rel.max()


def test_partition_bounds():
    from graphblast_amd.dist import partition_bounds
    gr = _graph(scale=12)
    ptr = gr["csr"][0]
    for world in (1, 2, 4, 8):
        b = partition_bounds(ptr, world)
        assert b[0] == 0 and b[-1] == gr["n"] and len(b) == world + 1
        assert all(x % 64 == 0 for x in b[:-1]) and all(b[i] <= b[i + 1] for i in range(world))
        if world > 1:
            share = np.diff(ptr[np.array(b)])
            assert share.max() <= 1.5 * gr["nnz"] / world + 64 * np.diff(ptr).max()


def test_word_range_gather_only_when_the_bounds_are_word_bounds():
    """round-2 review: n = 190 with a heavy last vertex gives bounds [0, 190, 190]; word 5 (vertices 160..189) belongs
    to rank 0 but sits in the empty rank 1's word range -- the in-place gather of word ranges must not be taken"""
    from graphblast_amd.dist import partition_bounds, word_slices_usable
    n = 190
    deg = np.ones(n, dtype=np.int64)
    deg[-1] = 5000                                          # the nnz split lands in the last vertex
    ptr = np.concatenate([[0], np.cumsum(deg)])
    b = partition_bounds(ptr, 2)
    assert b == [0, 190, 190]
    assert not word_slices_usable(b)
    assert word_slices_usable([0, 128, 190]) and word_slices_usable([0, 64, 128, 190]) and word_slices_usable([0, 190])
    assert not word_slices_usable([0, 64, 190, 190])


@pytest.mark.gpu
@pytest.mark.parametrize("edgeswitch", [0.0, 0.02])
@pytest.mark.parametrize("world", [1, 2, 4])
def test_partitioned_bfs_hip_engine_simulated_ranks(world, edgeswitch):
    """The real HIP level steps (grb_bfs_part_*) under `world` simulated ranks sharing one
    GPU: labels bit-exact vs the oracle, direction trace == single-GPU fused loop."""
    import threading
    from dist_helpers import ThreadComm, locked_engine
    from graphblast_amd.dist import Partition1D, HipEngine
    import graphblast_amd as g
    from oracle import simple_reference as sr
    gr = _graph(seed=3, scale=15)
    ptr, ind = gr["csr"]
    n = gr["n"]
    dev = torch.device("cuda", 0)
    tptr, tind = torch.from_numpy(ptr.astype(np.int64)).to(dev), torch.from_numpy(ind.astype(np.int64)).to(dev)
    shared = ThreadComm.Shared(world)
    Eng = locked_engine(HipEngine, shared.lock)
    with shared.lock:
        parts = [Partition1D(n, tptr, tind, r, world, dev, engine_cls=Eng, comm=ThreadComm(shared, r),
                             switchpoint=0.02, edgeswitch=edgeswitch) for r in range(world)]
    for src in (int(np.argmax(np.diff(ptr))), 11, 4097):
        results, labels = [None] * world, [None] * world
        def run(r):
            results[r] = parts[r].bfs(src)
            labels[r] = parts[r].gather_labels().cpu().numpy()
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(timeout=90) for t in ts]
        want = sr.bfs(ptr, ind, src)[0]
        for r in range(world):
            assert results[r] is not None, "rank %d did not finish" % r
            assert np.array_equal(labels[r], want), (src, r)
            assert results[r]["edges_traversed"] == int(np.diff(ptr)[want != 0].sum())
            assert results[r]["trace"] == results[0]["trace"]                  # every rank took the same decisions
        if edgeswitch > 0 and src == int(np.argmax(np.diff(ptr))):
            # the hub's level-2 frontier is a few thousand vertices carrying most of the edges: the
            # vertex-count rule pushes it, the edge-aware rule pulls it
            assert [t[0] for t in results[0]["trace"]][1] == "pull"
        _, stats = sr.bfs_do_stats(ptr, ind, ptr, ind, src, mxvmode=10, switchpoint=0.02)
        assert [t[0] for t in results[0]["trace"]] == ["pull" if s[0] else "push" for s in stats]
    # PageRank over the same simulated ranks (local SpMV shard + all-gather of the slices)
    deg = torch.from_numpy(np.diff(ptr).astype(np.float32)).to(dev)
    pr_out = [None] * world
    def run_pr(r):
        pr_out[r] = parts[r].pagerank(deg, alpha=0.85, eps=0.0, max_niter=10)
    ts = [threading.Thread(target=run_pr, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=90) for t in ts]
    want_pr = sr.pr(ptr, ind, 0.85, 0.0, 10)[0]
    for r in range(world):
        assert pr_out[r] is not None
        got = pr_out[r][0].cpu().numpy()
        rel = np.abs(got - want_pr) / np.maximum(np.abs(want_pr), 1e-30)
        assert rel.max() <= 1e-5 and pr_out[r][1]["iterations"] == 10, (r, rel.max())
    # SSSP over the same simulated ranks: the HIP MinimumPlus product on every in-edge shard
    wh = _weights(ptr, ind)
    w = torch.from_numpy(wh).to(dev)
    src_s = int(np.argmax(np.diff(ptr)))
    ss_out = [None] * world
    def run_sssp(r):
        ss_out[r] = parts[r].sssp(w, src_s)
    ts = [threading.Thread(target=run_sssp, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=90) for t in ts]
    want_d, want_it = _sssp_rounds(ptr, ind, wh, src_s)
    assert np.array_equal(want_d, sr.sssp(ptr, ind, wh, src_s)[0])
    for r in range(world):
        assert ss_out[r] is not None
        assert np.array_equal(ss_out[r][0].cpu().numpy(), want_d) and ss_out[r][1]["iterations"] == want_it, r


@pytest.mark.gpu
def test_part_level_step_kernels_direct():
    """The one-launch level steps of the partitioned loop against numpy: seed, push_small, OR of
    gathered bitmaps (the N > 1 combine, which no single-GPU run reaches otherwise) and apply2's three
    totals; two owner ranges of one graph play the ranks."""
    from graphblast_amd.dist import HipEngine, partition_bounds, bitmap_words
    from dist_helpers import NumpyEngine
    gr = _graph(seed=5, scale=13)
    ptr, ind = gr["csr"]
    n = gr["n"]
    dev = torch.device("cuda", 0)
    tptr = torch.from_numpy(ptr.astype(np.int64)).to(dev)
    tind = torch.from_numpy(ind.astype(np.int64)).to(dev)
    bounds = partition_bounds(ptr, 2)
    nw = bitmap_words(n)
    deg_full = torch.from_numpy(np.diff(ptr).astype(np.int32)).to(dev)
    src = int(np.argmax(np.diff(ptr)))
    engines, states = [], []
    for r in range(2):
        lo, hi = bounds[r], bounds[r + 1]
        e0, e1 = int(ptr[lo]), int(ptr[hi])
        lptr = (tptr[lo:hi + 1] - e0).to(torch.int32).contiguous()
        lind = tind[e0:e1].to(torch.int32).contiguous()
        eng = HipEngine(n, lo, lptr, lind, dev)
        ref = NumpyEngine(n, lo, lptr.cpu(), lind.cpu(), torch.device("cpu"))
        z = lambda: torch.zeros(nw, dtype=torch.int32, device=dev)
        st = dict(vis=z(), new_local=z(), new_global=z(), label=torch.zeros(max(hi - lo, 1), dtype=torch.float32, device=dev))
        eng.seed(st["vis"], st["new_global"], st["label"], src)
        engines.append((eng, ref, lo, hi))
        states.append(st)
    # reference state on the host
    hv = np.zeros(nw, dtype=np.uint32)
    hv[src >> 5] = np.uint32(1) << np.uint32(src & 31)
    hfront = hv.copy()
    for level in (2, 3):
        parts = []
        for (eng, ref, lo, hi), st in zip(engines, states):
            assert np.array_equal(st["vis"].cpu().numpy().view(np.uint32), hv)
            eng.push_small(st["new_global"], st["vis"], st["new_local"])
            want = torch.zeros(nw, dtype=torch.int32)
            ref.push(torch.from_numpy(hfront.view(np.int32).copy()), torch.from_numpy(hv.view(np.int32).copy()), want)
            assert np.array_equal(st["new_local"].cpu().numpy(), want.numpy()), (level, lo)
            parts.append(st["new_local"])
        gathered = torch.cat(parts).contiguous()
        fresh = (parts[0].cpu().numpy().view(np.uint32) | parts[1].cpu().numpy().view(np.uint32)) & ~hv
        for (eng, ref, lo, hi), st in zip(engines, states):
            eng.or_parts(gathered, 2, nw, st["new_global"])
            assert np.array_equal(st["new_global"].cpu().numpy().view(np.uint32) & ~hv, fresh)
            found, local_edges, all_edges = eng.apply2(st["new_global"], st["vis"], st["label"], level, deg_full)
            idx = np.nonzero(np.unpackbits(fresh.view(np.uint8), bitorder="little")[:n])[0]
            assert found == idx.size
            assert all_edges == int(np.diff(ptr)[idx].sum())
            own = idx[(idx >= lo) & (idx < hi)]
            assert local_edges == int(np.diff(ptr)[own].sum())
            lab = st["label"].cpu().numpy()
            assert np.all(lab[own - lo] == level)
        hv |= fresh
        hfront = fresh.copy()


def _worker_directed(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from dist_helpers import NumpyEngine
        from graphblast_amd.dist import Partition1D
        gr = _graph(seed=7, scale=11, sym=False)
        ptr, ind = gr["csr"]
        cptr, cind = gr["csc"]
        t = lambda a: torch.from_numpy(a.astype(np.int64))
        out = []
        for mode in (10, 11, 12):
            part = Partition1D(gr["n"], t(ptr), t(ind), rank, world, torch.device("cpu"), engine_cls=NumpyEngine,
                               mxvmode=mode, switchpoint=0.05, symmetric=False, in_edges=(t(cptr), t(cind)))
            for src in (int(np.argmax(np.diff(ptr))), 9):
                res = part.bfs(src)
                out.append((mode, src, res["edges_traversed"], res["reached"], part.gather_labels().numpy().copy()))
        part = Partition1D(gr["n"], t(ptr), t(ind), rank, world, torch.device("cpu"), engine_cls=NumpyEngine,
                           symmetric=False, in_edges=(t(cptr), t(cind)))
        deg = torch.from_numpy(np.maximum(np.diff(ptr), 1).astype(np.float32))
        pvec, info = part.pagerank(deg, alpha=0.85, eps=0.0, max_niter=8)
        # SSSP along the edge directions: the in-edge shard carries the weights in CSC order
        w_in = torch.from_numpy(_directed_weights(cptr, cind, major_is_source=False))
        src_s = int(np.argmax(np.diff(ptr)))
        dvec, sinfo = part.sssp(w_in, src_s)
        if rank == 0:
            q.put((out, pvec.numpy().copy(), info["iterations"], (src_s, dvec.numpy().copy(), sinfo["iterations"])))
    except Exception:
        import traceback
        q.put(("error", "rank %d: %s" % (rank, traceback.format_exc()), 0, None))
        raise
    finally:
        dist.destroy_process_group()


def test_partitioned_directed_graph_gloo():
    """A DIRECTED graph on two ranks: every rank holds the out-edge rows (push) and the in-edge rows (pull,
    PageRank) of its vertices.  BFS labels bit-exact against the oracle on the CSR (out-edges); PageRank
    against the oracle's push formulation on the same CSR."""
    from oracle import simple_reference as sr
    gr = _graph(seed=7, scale=11, sym=False)
    ptr, ind = gr["csr"]
    assert not np.array_equal(gr["csr"][1], gr["csc"][1])
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_directed, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, pr_vec, pr_iters, sssp_out = q.get(timeout=180)
    assert not (isinstance(out, str) and out == "error"), pr_vec
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    deg = np.diff(ptr)
    src_s, dvec, s_iters = sssp_out
    w_out = _directed_weights(ptr, ind, major_is_source=True)
    want_d, want_it = _sssp_rounds(ptr, ind, w_out, src_s)
    assert np.array_equal(dvec, want_d) and s_iters == want_it
    assert np.array_equal(dvec, sr.sssp(ptr, ind, w_out, src_s)[0])
    for mode, src, edges, reached, labels in out:
        want = sr.bfs(ptr, ind, src)[0]
        assert np.array_equal(labels, want), (mode, src)
        assert edges == int(deg[want != 0].sum()) and reached == int(np.count_nonzero(want))
    # SimpleReferencePr divides by the out-degree; sinks only ever divide their own (unused) contribution
    n = gr["n"]
    p = np.full(n, np.float32(1.0 / n), np.float32)
    rows = np.repeat(np.arange(n), deg)
    for _ in range(8):
        contrib = (np.float32(0.85) * p / np.maximum(deg, 1).astype(np.float32)).astype(np.float32)
        nxt = np.full(n, np.float32((1 - 0.85) / n), np.float64)
        np.add.at(nxt, ind, contrib[rows].astype(np.float64))
        p = nxt.astype(np.float32)
    assert pr_iters == 8
    rel = np.abs(pr_vec - p) / np.maximum(np.abs(p), 1e-30)
    assert rel.max() <= 1e-4, rel.max()


@pytest.mark.gpu
def test_library_communicator_on_one_gpu():
    """csrc/comm.hip with a world of one rank on the GPU box: RCCL bound with dlopen, communicator created from
    a unique id, all-gather / in-place all-gather-v / all-reduce enqueued on the communication stream and fenced
    against the compute stream; then the partitioned BFS and the chunk-overlapped PageRank through RcclComm, and
    a DIRECTED graph through the HIP engine's two shards."""
    import ctypes as C
    from graphblast_amd import _lib
    from graphblast_amd.dist import Partition1D, RcclComm, bitmap_words
    from oracle import simple_reference as sr
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    gr = _graph(seed=3, scale=14)
    ptr, ind = gr["csr"]
    n = gr["n"]
    comm = RcclComm(0, 1, bitmap_words(n), dev)
    r, w = C.c_int(-1), C.c_int(-1)
    lib.grb_comm_info(C.byref(r), C.byref(w))
    assert (r.value, w.value) == (0, 1)
    comm.timing(True)
    a = torch.arange(1000, dtype=torch.float32, device=dev)
    b = torch.zeros(1000, dtype=torch.float32, device=dev)
    assert lib.grb_comm_allgather(a.data_ptr(), b.data_ptr(), 4000) == 0
    assert lib.grb_comm_wait() == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    acc = torch.tensor([2.5, 4.0], dtype=torch.float64, device=dev)
    assert lib.grb_comm_allreduce_sum_f64(acc.data_ptr(), 2) == 0
    assert lib.grb_comm_wait() == 0
    torch.cuda.synchronize()
    assert acc.tolist() == [2.5, 4.0]
    off, cnt = (C.c_longlong * 1)(0), (C.c_longlong * 1)(4000)
    assert lib.grb_comm_allgatherv_inplace(a.data_ptr(), off, cnt) == 0
    assert lib.grb_comm_wait() == 0
    us, calls = comm.stats()
    assert calls == 3 and us > 0
    comm.timing(False)
    tptr, tind = torch.from_numpy(ptr.astype(np.int64)).to(dev), torch.from_numpy(ind.astype(np.int64)).to(dev)
    part = Partition1D(n, tptr, tind, 0, 1, dev, comm=comm, switchpoint=0.02)
    for src in (int(np.argmax(np.diff(ptr))), 5):
        res = part.bfs(src)
        want = sr.bfs(ptr, ind, src)[0]
        assert np.array_equal(part.gather_labels().cpu().numpy(), want)
        assert res["edges_traversed"] == int(np.diff(ptr)[want != 0].sum())
    deg = torch.from_numpy(np.diff(ptr).astype(np.float32)).to(dev)
    pvec, info = part.pagerank(deg, alpha=0.85, eps=0.0, max_niter=10)
    assert info["iterations"] == 10 and info["overlapped_chunks"] == 2
    want_pr = sr.pr(ptr, ind, 0.85, 0.0, 10)[0]
    rel = np.abs(pvec.cpu().numpy() - want_pr) / np.maximum(np.abs(want_pr), 1e-30)
    assert rel.max() <= 1e-5, rel.max()
    # directed graph: out-edge and in-edge shards
    gd = _graph(seed=9, scale=13, sym=False)
    dp, di = gd["csr"]
    cp, ci = gd["csc"]
    t = lambda x: torch.from_numpy(x.astype(np.int64)).to(dev)
    pd = Partition1D(gd["n"], t(dp), t(di), 0, 1, dev, comm=RcclComm(0, 1, bitmap_words(gd["n"]), dev),
                     symmetric=False, in_edges=(t(cp), t(ci)), switchpoint=0.02)
    for mode_src in (int(np.argmax(np.diff(dp))), 3):
        pd.bfs(mode_src)
        assert np.array_equal(pd.gather_labels().cpu().numpy(), sr.bfs(dp, di, mode_src)[0])
    lib.grb_comm_destroy()
