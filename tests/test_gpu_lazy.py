"""The queue of element-wise calls (csrc/lazy.hip): eWiseAdd / eWiseMult / dup on dense, library-owned
vectors are queued and run as ONE kernel when anything else is called.  Nothing but the launch count may differ: every
chain here runs twice, queued and call by call (grb_set_lazy(0) -- the path the per-operation tests and the fuzz
campaigns pin against the oracle), and the two final states must be equal bit for bit: all 17 semirings, f32 and
i32, operands that hold the semiring's identity (eWiseMult's dead-element rule), aliased operands, chains longer
than the queue and wider than its buffer table, and the calls that must never be queued."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEMIRINGS = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusDivides", "PlusGreater",
             "GreaterPlus", "PlusMinus", "PlusLess", "CustomLessPlus", "MinimumMultiplies", "MultipliesMultiplies",
             "NotEqualToPlus", "MinimumSelectSecond", "PlusNotEqualTo", "CustomLessLess", "MinimumNotEqualTo"]


def _values(rng, n, dtype):
    x = rng.integers(-3, 4, n).astype(dtype)
    special = [0, 1, np.finfo(np.float32).max if dtype == np.float32 else np.iinfo(np.int32).max,
               np.finfo(np.float32).tiny if dtype == np.float32 else np.iinfo(np.int32).min]
    pick = rng.random(n) < 0.3                              # the identities of the 17 semirings, often
    x[pick] = np.array(special, dtype=dtype)[rng.integers(0, 4, int(pick.sum()))]
    return x


def _random_chain(rng, npool, length):
    chain = []
    for _ in range(length):
        kind = rng.choice(["add", "mult", "add_scalar", "dup", "assign", "assign_scmp"], p=[0.3, 0.3, 0.1, 0.1, 0.1, 0.1])
        w, u, v = (int(z) for z in rng.integers(0, npool, 3))
        chain.append((kind, SEMIRINGS[int(rng.integers(0, len(SEMIRINGS)))], w, u, v, float(rng.integers(-2, 3))))
    return chain


def _run(g, chain, init, dtype, lazy, probe=None):
    before = g.set_lazy(1 if lazy else 0)
    try:
        d = g.Descriptor(); d.loadArgs()
        ds = g.Descriptor(); ds.loadArgs()
        assert ds.toggle(0) == 0                             # GrB_MASK -> GrB_SCMP
        vecs = []
        for x in init:
            v = g.Vector(x.size, dtype)
            assert v.build(x) == 0
            vecs.append(v)
        most = 0
        for kind, sr, w, u, v, s in chain:
            if kind == "add":
                assert g.eWiseAdd(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
            elif kind == "mult":
                assert g.eWiseMult(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
            elif kind == "add_scalar":
                assert g.eWiseAdd(vecs[w], None, None, sr, vecs[u], s, d) == 0
            elif kind in ("assign", "assign_scmp"):          # w = s where the mask u passes (w == u: run at once)
                assert g.assign(vecs[w], vecs[u], None, s, None, None, ds if kind == "assign_scmp" else d) == 0
            else:
                assert vecs[w].dup(vecs[u]) == 0
            most = max(most, g.lazy_pending())
        if probe is not None:
            probe.append(most)
        out = [v.extractTuples()[1].copy() for v in vecs]
        assert g.lazy_pending() == 0                         # reading a vector ran everything
        return out
    finally:
        g.set_lazy(before)


def _same(a, b):
    return all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


@pytest.mark.parametrize("dtype", [np.float32, np.int32])
def test_random_chains_equal_call_by_call(dtype):
    import graphblast_amd as g
    rng = np.random.default_rng(7 if dtype == np.float32 else 8)
    for trial in range(60):
        npool = int(rng.integers(2, 12))                     # up to 11 vectors: more than the queue's 8 buffers
        n = int(rng.choice([1, 63, 64, 1000, 4097]))
        init = [_values(rng, n, dtype) for _ in range(npool)]
        chain = _random_chain(rng, npool, int(rng.integers(1, 15)))     # up to 14 steps: more than the queue's 6
        depth = []
        with np.errstate(all="ignore"):
            got = _run(g, chain, init, dtype, True, depth)
            want = _run(g, chain, init, dtype, False)
        assert _same(got, want), (trial, chain)
        if len(chain) >= 2 and dtype == np.float32:
            assert depth[0] >= 1                             # the calls really waited


def test_multiply_then_add_is_not_contracted():
    """x * y + z in two calls rounds the product before the sum; a fused multiply-add would not"""
    import graphblast_amd as g
    rng = np.random.default_rng(3)
    n = 100000
    x = (rng.random(n, dtype=np.float32) + np.float32(1)).astype(np.float32)
    y = (rng.random(n, dtype=np.float32) + np.float32(1)).astype(np.float32)
    z = (-(x * y)).astype(np.float32) + np.float32(2 ** -20)
    chain = [("mult", "PlusMultiplies", 3, 0, 1, 0.0), ("add", "PlusMultiplies", 3, 3, 2, 0.0)]
    init = [x, y, z, np.zeros(n, np.float32)]
    got = _run(g, chain, init, np.float32, True)
    want = _run(g, chain, init, np.float32, False)
    assert _same(got, want)
    assert np.array_equal(got[3], ((x * y).astype(np.float32) + z).astype(np.float32))
    assert not np.array_equal(got[3], (x.astype(np.float64) * y + z).astype(np.float32))   # the fma's answer differs


def test_what_flushes_and_what_is_never_queued():
    import torch
    import graphblast_amd as g
    before = g.set_lazy(1)
    try:
        dev = torch.device("cuda", 0)
        n = 5000
        rng = np.random.default_rng(1)
        d = g.Descriptor(); d.loadArgs()
        a, b, c = (g.Vector(n) for _ in range(3))
        xa, xb = rng.integers(0, 5, n).astype(np.float32), rng.integers(0, 5, n).astype(np.float32)
        assert a.build(xa) == 0 and b.build(xb) == 0
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", a, b, d) == 0
        assert g.lazy_pending() == 1
        info, val = g.reduce(None, "Plus", c, d)               # a reduction reads c: the queue runs first
        assert info == 0 and g.lazy_pending() == 0 and val == float((xa + xb).sum())
        # a sparse operand, a mask, adopted storage: executed at once
        idx = np.arange(0, n, 7, dtype=np.int32)
        s = g.Vector(n)
        assert s.build(idx, np.ones(idx.size, np.float32), idx.size, None) == 0
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", a, s, d) == 0 and g.lazy_pending() == 0
        assert g.eWiseMult(c, a, None, "PlusMultiplies", a, b, d) == 0 and g.lazy_pending() == 0
        t = torch.from_numpy(xa.copy()).to(dev)
        ad = g.Vector(n)
        assert ad.build_device(t.data_ptr(), n) == 0           # adopted: the caller may touch t without asking
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", ad, b, d) == 0 and g.lazy_pending() == 0
        assert g.eWiseAdd(ad, None, None, "PlusMultiplies", a, b, d) == 0 and g.lazy_pending() == 0
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), xa + xb)
        # a queued result feeds a product: the product sees it
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", a, b, d) == 0
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", c, 1.0, d) == 0
        assert g.lazy_pending() == 2
        A = g.Matrix(n, n)
        rows = np.arange(n, dtype=np.int32)
        assert A.build(rows, (rows + 1) % n, np.ones(n, np.float32), n, None) == 0
        w = g.Vector(n)
        dd = g.Descriptor(); dd.loadArgs(mxvmode=2)
        assert g.mxv(w, None, None, "PlusMultiplies", A, c, dd) == 0
        assert g.lazy_pending() == 0
        assert np.array_equal(w.extractTuples()[1], np.roll(xa + xb + 1, -1))
    finally:
        g.set_lazy(before)


def test_descriptor_toggles_do_not_cut_a_chain():
    """sssp.hpp:70-83: eWiseAdd, eWiseAdd, toggle(GrB_MASK), assign, toggle(GrB_MASK) -- the toggles touch no vector,
    so the three element-wise calls stay one queue; the assign keeps the mask sense it was CALLED with"""
    import graphblast_amd as g
    before = g.set_lazy(1)
    try:
        rng = np.random.default_rng(2)
        n = 3000
        d = g.Descriptor(); d.loadArgs()
        f2x, vx = rng.integers(0, 9, n).astype(np.float32), rng.integers(0, 9, n).astype(np.float32)
        f2, v, m = g.Vector(n), g.Vector(n), g.Vector(n)
        assert f2.build(f2x) == 0 and v.build(vx) == 0 and m.fill(0.0) == 0
        assert g.eWiseAdd(m, None, None, "CustomLessPlus", f2, v, d) == 0        # m = f2 < v
        assert g.eWiseAdd(v, None, None, "MinimumPlus", v, f2, d) == 0           # v = min(v, f2)
        assert d.toggle(0) == 0
        assert g.lazy_pending() == 2
        assert g.assign(f2, m, None, 99.0, None, None, d) == 0                   # f2 = 99 where m is ZERO (scmp)
        assert d.toggle(0) == 0
        assert g.lazy_pending() == 3
        want_m = (f2x < vx).astype(np.float32)
        assert np.array_equal(m.extractTuples()[1], want_m) and g.lazy_pending() == 0
        assert np.array_equal(v.extractTuples()[1], np.minimum(vx, f2x))
        assert np.array_equal(f2.extractTuples()[1], np.where(want_m == 0, np.float32(99), f2x))
    finally:
        g.set_lazy(before)


def test_deferred_chains_against_the_oracle():
    """Chains of length > 1, read back only at the END, against oracle/ops.py (the per-call fuzz reads every vector after
    every call, so against the oracle the queue only ever held one step there).  Integer-valued data: bit for bit."""
    import graphblast_amd as g
    from backends import OracleBackend, HipBackend
    ob, hb = OracleBackend(), HipBackend()
    before = g.set_lazy(1)
    try:
        for dtype, seed in ((np.float32, 21), (np.int32, 22)):
            rng = np.random.default_rng(seed)
            deepest = 0
            for trial in range(40):
                npool = int(rng.integers(2, 10))
                n = int(rng.choice([1, 64, 1000, 4097]))
                init = [rng.integers(-3, 4, n).astype(dtype) for _ in range(npool)]
                chain = _random_chain(rng, npool, int(rng.integers(2, 13)))
                outs = []
                for be in (ob, hb):
                    d, ds = be.descriptor(), be.descriptor()
                    assert be.toggle(ds, 0) == 0
                    vecs = [be.vector(n, dtype) for _ in range(npool)]
                    for v, x in zip(vecs, init):
                        assert be.build_dense(v, x) == 0
                    with np.errstate(all="ignore"):
                        for kind, sr, w, u, v, s in chain:
                            if kind == "add":
                                assert be.eWiseAdd(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
                            elif kind == "mult":
                                assert be.eWiseMult(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
                            elif kind == "add_scalar":
                                assert be.eWiseAdd(vecs[w], None, None, sr, vecs[u], dtype(s), d) == 0
                            elif kind in ("assign", "assign_scmp"):
                                assert be.assign(vecs[w], vecs[u], dtype(s), ds if kind == "assign_scmp" else d) == 0
                            else:
                                assert vecs[w].dup(vecs[u]) == 0
                            if be is hb:
                                deepest = max(deepest, g.lazy_pending())
                    outs.append([np.asarray(be.dense_values(v)).copy() for v in vecs])
                for k, (a, b) in enumerate(zip(*outs)):
                    same = (a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
                    assert same.all(), (dtype, trial, k, chain, a[~same][:4], b[~same][:4])
            assert deepest >= 3                               # chains really were deferred
    finally:
        g.set_lazy(before)


def test_exposed_storage_is_never_deferred():
    """grb_vector_device_ptrs hands out the raw storage (torch / RCCL interop): a caller holding the pointer can read it
    without an API call, so nothing that touches such a vector may wait in the queue (sticky)."""
    import graphblast_amd as g
    before = g.set_lazy(1)
    try:
        n = 4096
        rng = np.random.default_rng(5)
        d = g.Descriptor(); d.loadArgs()
        a, b, c = (g.Vector(n) for _ in range(3))
        xa, xb = rng.integers(0, 5, n).astype(np.float32), rng.integers(0, 5, n).astype(np.float32)
        assert a.build(xa) == 0 and b.build(xb) == 0 and c.fill(0.0) == 0
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", a, b, d) == 0 and g.lazy_pending() == 1
        ptr = c.device_ptrs()[2]                               # flushes, and marks c
        assert g.lazy_pending() == 0 and ptr
        assert g.eWiseAdd(c, None, None, "PlusMultiplies", c, b, d) == 0
        assert g.lazy_pending() == 0                           # ran at once: c is exposed
        import ctypes
        host = np.empty(n, np.float32)
        hip = ctypes.CDLL("libamdhip64.so")                    # read through the raw pointer, no library call in between
        assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(4 * n), 2) == 0
        assert np.array_equal(host, xa + xb + xb)
        assert g.eWiseAdd(a, None, None, "PlusMultiplies", c, b, d) == 0 and g.lazy_pending() == 0   # as an operand too
    finally:
        g.set_lazy(before)


def test_a_refused_program_still_runs_its_steps(monkeypatch):
    """lazy_flush: the queued calls were answered GrB_SUCCESS; when the fused program cannot be launched their effect
    must still happen (one step at a time through the eager kernels), not be dropped."""
    import graphblast_amd as g
    rng = np.random.default_rng(11)
    init = [rng.integers(-3, 4, 5000).astype(np.float32) for _ in range(5)]
    chain = _random_chain(rng, 5, 6)
    want = _run(g, chain, init, np.float32, False)
    monkeypatch.setenv("GRB_LAZY_FORCE_STEPWISE", "1")
    depth = []
    got = _run(g, chain, init, np.float32, True, depth)
    monkeypatch.delenv("GRB_LAZY_FORCE_STEPWISE")
    assert depth[0] >= 2 and _same(got, want)


@pytest.mark.parametrize("dtype", [np.float32, np.int32])
def test_a_reduction_at_the_end_of_a_chain_runs_in_its_launch(dtype):
    """pr.hpp:72-80 ends eWiseMult, eWiseAdd, reduce: the reduction of a pending chain's result is folded by the chain's
    own launch, in the reduce kernel's association order -- the value is bit for bit the separate launch's, every vector
    is left as the calls would have left it, and a reduction of something the queue does not write is an ordinary one."""
    import graphblast_amd as g
    rng = np.random.default_rng(31 if dtype == np.float32 else 32)
    monoids = ["PlusMonoid", "MultipliesMonoid", "MinimumMonoid", "MaximumMonoid", "LogicalOrMonoid", "LogicalAndMonoid"]
    for trial in range(24):
        n = int(rng.choice([1, 3, 64, 1000, 4097, 100003]))
        npool = int(rng.integers(3, 7))
        if dtype == np.float32:
            init = [(rng.random(n, dtype=np.float32) * 2 - 1).astype(np.float32) for _ in range(npool)]
        else:
            init = [rng.integers(-3, 4, n).astype(np.int32) for _ in range(npool)]
        chain = [c for c in _random_chain(rng, npool, int(rng.integers(1, 6)))]
        target = chain[-1][2]                                  # the vector the last step writes
        mono = monoids[trial % len(monoids)]
        out = {}
        for lazy in (True, False):
            before = g.set_lazy(1 if lazy else 0)
            try:
                d = g.Descriptor(); d.loadArgs()
                ds = g.Descriptor(); ds.loadArgs(); assert ds.toggle(0) == 0
                vecs = []
                for x in init:
                    v = g.Vector(x.size, dtype); assert v.build(x) == 0; vecs.append(v)
                fused0 = g.lazy_fused_reductions()
                with np.errstate(all="ignore"):
                    for kind, sr, w, u, v, s in chain:
                        if kind == "add": assert g.eWiseAdd(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
                        elif kind == "mult": assert g.eWiseMult(vecs[w], None, None, sr, vecs[u], vecs[v], d) == 0
                        elif kind == "add_scalar": assert g.eWiseAdd(vecs[w], None, None, sr, vecs[u], s, d) == 0
                        elif kind in ("assign", "assign_scmp"): assert g.assign(vecs[w], vecs[u], None, s, None, None, ds if kind == "assign_scmp" else d) == 0
                        else: assert vecs[w].dup(vecs[u]) == 0
                    pending = g.lazy_pending()
                    info, val = g.reduce(None, mono, vecs[target], d)
                    assert info == 0 and g.lazy_pending() == 0
                    other = (target + 1) % npool
                    info2, val2 = g.reduce(None, "PlusMonoid", vecs[other], d)
                    assert info2 == 0
                if lazy and pending >= 1:
                    assert g.lazy_fused_reductions() == fused0 + 1, (trial, chain)
                out[lazy] = (np.float32(val) if dtype == np.float32 else np.int32(val), np.float32(val2) if dtype == np.float32 else np.int32(val2),
                             [v.extractTuples()[1].copy() for v in vecs])
            finally:
                g.set_lazy(before)
        a, b = out[True], out[False]
        assert np.array([a[0]]).view(np.uint32)[0] == np.array([b[0]]).view(np.uint32)[0] or (np.isnan(a[0]) and np.isnan(b[0])), (trial, mono, a[0], b[0], chain)
        assert np.array([a[1]]).view(np.uint32)[0] == np.array([b[1]]).view(np.uint32)[0] or (np.isnan(a[1]) and np.isnan(b[1]))
        assert _same(a[2], b[2]), (trial, chain)
