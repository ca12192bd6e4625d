"""Differential fuzzing of the op-level semantics: random SEQUENCES of GraphBLAS calls on a small pool
of vectors, run in lock-step through the C ABI (MI355X) and through the oracle (oracle/ops.py), with
the complete observable state -- every call's Info code, every vector's storage type, nvals and
contents, the descriptor's lastmxv -- compared after each call.  Sequences matter: the reference's
containers keep both representations and convert in place, so what an op does depends on what the
previous ones left behind.

usage (GPU box): python tests/tools/ops_fuzz.py [--seqs 200] [--len 30] [--seed 0] [--n 70]
Prints the first divergence of every failing sequence with the calls that led to it; exit code 1 if
any.  Test infrastructure: imports oracle/ as the checker.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from backends import HipBackend, OracleBackend  # noqa: E402

F = np.float32
# additive monoids whose result does not depend on the reduction order (the others are compared
# only where every row / output has at most one contribution, which random sequences cannot promise)
ORDER_FREE = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusGreater", "PlusMinus",
              "PlusLess", "MinimumMultiplies", "MinimumSelectSecond", "PlusNotEqualTo", "MinimumNotEqualTo"]
EWISE_SR = ["LogicalOrAnd", "PlusMultiplies", "MinimumPlus", "MaximumMultiplies", "PlusDivides", "PlusGreater",
            "GreaterPlus", "PlusMinus", "PlusLess", "CustomLessPlus", "MinimumMultiplies", "MultipliesMultiplies",
            "NotEqualToPlus", "MinimumSelectSecond", "PlusNotEqualTo", "CustomLessLess", "MinimumNotEqualTo"]
MONOIDS = ["Plus", "Minimum", "Maximum", "LogicalOr"]
NVEC = 5
# float sums past 2^24 (a 9000-entry hub row times values in the thousands) are compared at the accuracy a
# SEQUENTIAL float32 sum of that length has, not at 1e-5: the oracle adds in stored order, the kernel in a tree
RTOL = 1e-4
STRUC_ONLY = [False]
INT_MODE = [False]
BFS_MIX = [False]


def state(be, vecs, desc):
    out = []
    for v in vecs:
        st = be.storage(v)
        if st == 1:
            idx, val = be.sparse_tuples(v)
            out.append((st, np.asarray(idx).copy(), np.asarray(val).copy()))
        elif st == 2:
            # dense_values() of the adapters densifies a sparse vector, so it is only called here
            out.append((st, None, np.asarray(be.dense_values(v)).copy()))
        else:
            out.append((st, None, None))
    return out, be.lastmxv(desc)


def differs(a, b):
    (sa, la), (sb, lb) = a, b
    for k, (x, y) in enumerate(zip(sa, sb)):
        if x[0] != y[0]:
            return "vector %d: storage %d vs %d" % (k, x[0], y[0])
        if x[0] == 1:
            if x[1].shape != y[1].shape or not np.array_equal(x[1], y[1]):
                return "vector %d: sparse indices %s vs %s" % (k, x[1][:12], y[1][:12])
        if x[0] == 1 and STRUC_ONLY[0]:
            continue                                  # struconly: a sparse result carries no values
        if x[0] in (1, 2):
            if x[2].shape != y[2].shape or not np.allclose(x[2], y[2], rtol=RTOL, atol=0, equal_nan=True):
                bad = np.nonzero(~(np.isclose(x[2], y[2], rtol=RTOL, atol=0) | (np.isnan(x[2]) & np.isnan(y[2]))))[0][:8] if x[2].shape == y[2].shape else []
                return "vector %d (%s): values differ at %s: %s vs %s" % (
                    k, "sparse" if x[0] == 1 else "dense", bad, x[2][bad] if len(bad) else x[2].shape,
                    y[2][bad] if len(bad) else y[2].shape)
    if la != lb:
        return "lastmxv %d vs %d" % (la, lb)
    return None


def gen_bfs_call(rng, n):
    """The op mix of the traversal drivers (bfs.hpp / diameter.hpp): Boolean vxm / mxv under masks,
    constant assign, reduce, swap, representation changes -- the calls struconly / opreuse are for."""
    kind = rng.choice(["vxm", "mxv", "assign", "reduce", "fill", "build_sparse", "build_dense", "swap",
                       "toggle_mask", "mxvmode"], p=[0.3, 0.1, 0.2, 0.1, 0.04, 0.08, 0.06, 0.06, 0.04, 0.02])
    v = lambda: int(rng.integers(0, NVEC))
    if kind in ("vxm", "mxv"):
        return (kind, v(), v() if rng.random() < 0.7 else None, False, "LogicalOrAnd", v())
    if kind == "assign":
        return (kind, v(), v(), float(rng.integers(0, 3)))
    if kind == "reduce":
        return (kind, "Plus", v())
    if kind == "fill":
        return (kind, v(), float(rng.integers(0, 2)))
    if kind == "build_sparse":
        k = int(rng.integers(0, max(2, n // 3)))
        idx = np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
        return (kind, v(), idx, np.ones(k, dtype=F))
    if kind == "build_dense":
        return (kind, v(), (rng.random(n) < 0.3).astype(F))
    if kind == "swap":
        return (kind, v(), v())
    if kind == "mxvmode":
        return (kind, int(rng.choice([10, 11, 12])))
    return (kind,)


def gen_call(rng, n):
    """One call as a tuple the two runners interpret identically."""
    if BFS_MIX[0]:
        return gen_bfs_call(rng, n)
    kind = rng.choice(["vxm", "mxv", "eWiseAdd", "eWiseMult", "assign", "reduce", "fill", "build_sparse",
                       "build_dense", "dup", "swap", "toggle_mask", "mxvmode", "clear", "interfere"],
                      p=[0.15, 0.11, 0.11, 0.11, 0.11, 0.06, 0.05, 0.07, 0.05, 0.04, 0.03, 0.03, 0.02, 0.01, 0.05])
    if kind == "interfere":
        # a whole algorithm driver / matrix operation of the library on ITS OWN vectors, between the
        # compared calls: whatever state it keeps (scratch slots, pooled storage, plans, tickets,
        # mailboxes) must not leak into what follows.  The oracle side does nothing.
        return (kind, str(rng.choice(["bfs", "bfs_opbyop", "sssp", "pr", "cc", "tc", "trace", "mis", "gc", "lgc",
                                       "reduce_rows"])), int(rng.integers(0, n)))
    v = lambda: int(rng.integers(0, NVEC))
    if kind in ("vxm", "mxv"):
        mask = v() if rng.random() < 0.5 else None
        return (kind, v(), mask, bool(rng.random() < 0.25), str(rng.choice(ORDER_FREE)), v())
    if kind in ("eWiseAdd", "eWiseMult"):
        mask = v() if (kind == "eWiseMult" and rng.random() < 0.4) else None
        srs = [x for x in EWISE_SR if not (INT_MODE[0] and x == "PlusDivides")]     # int x / 0 is undefined
        return (kind, v(), mask, str(rng.choice(srs)), v(), v())
    if kind == "assign":
        return (kind, v(), v(), float(rng.integers(0, 4)))
    if kind == "reduce":
        return (kind, str(rng.choice(MONOIDS)), v())
    if kind == "fill":
        return (kind, v(), float(rng.integers(0, 3)))
    if kind == "build_sparse":
        k = int(rng.integers(0, max(2, n // 3)))
        idx = np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
        return (kind, v(), idx, rng.integers(0, 4, k).astype(F))
    if kind == "build_dense":
        return (kind, v(), (rng.integers(0, 4, n) * (rng.random(n) < 0.5)).astype(F))
    if kind in ("dup", "swap"):
        return (kind, v(), v())
    if kind == "mxvmode":
        return (kind, int(rng.choice([10, 11, 12])))
    if kind == "clear":
        return (kind, v())
    return (kind,)


def run_call(be, call, vecs, A, desc):
    k = call[0]
    if k == "vxm":
        _, w, m, acc, sr, u = call
        return be.vxm(vecs[w], None if m is None else vecs[m], "accum" if acc else None, sr, vecs[u], A, desc)
    if k == "mxv":
        _, w, m, acc, sr, u = call
        return be.mxv(vecs[w], None if m is None else vecs[m], "accum" if acc else None, sr, A, vecs[u], desc)
    if k == "eWiseAdd":
        _, w, m, sr, a, b = call
        return be.eWiseAdd(vecs[w], None, None, sr, vecs[a], vecs[b], desc)
    if k == "eWiseMult":
        _, w, m, sr, a, b = call
        return be.eWiseMult(vecs[w], None if m is None else vecs[m], None, sr, vecs[a], vecs[b], desc)
    if k == "assign":
        return be.assign(vecs[call[1]], vecs[call[2]], call[3], desc)
    if k == "reduce":
        try:
            info, val = be.reduce(call[1], vecs[call[2]], desc)
        except ValueError as e:                       # the oracle raises where the ABI returns a code
            from oracle import ops
            return (getattr(ops, str(e)), float("nan"))
        return (info, float(val) if info == 0 else float("nan"))
    if k == "fill":
        return be.fill(vecs[call[1]], call[2])
    if k == "build_sparse":
        return be.build_sparse(vecs[call[1]], call[2], call[3])
    if k == "build_dense":
        return be.build_dense(vecs[call[1]], call[2])
    if k == "dup":
        return vecs[call[1]].dup(vecs[call[2]])
    if k == "swap":
        return vecs[call[1]].swap(vecs[call[2]])
    if k == "toggle_mask":
        return be.toggle(desc, 0)
    if k == "mxvmode":
        return be.set(desc, 8, call[1])
    if k == "clear":
        return vecs[call[1]].clear()
    if k == "interfere":
        if hasattr(be, "g"):
            interfere(be, call[1], call[2], A)
        return 0
    raise ValueError(k)


_INT_TWIN = {}


def _is_symmetric(ptr, ind, n):
    import scipy.sparse as sp
    M = sp.csr_matrix((np.ones(ind.size, dtype=np.int8), ind, ptr), shape=(n, n))
    return (M != M.T).nnz == 0


def interfere(hb, which, arg, A):
    g = hb.g
    n = A.nrows()
    d = hb.descriptor(mxvmode=0)
    from oracle import simple_reference as sr
    ptr, ind, val = A.host_csr()
    # the int copy of the pattern for the int drivers, cached on the pattern's CONTENT: id(A) recurs once a
    # sequence's matrix is freed, and a twin of the previous sequence's matrix would then be handed to cc / tc
    key = (n, ptr.tobytes(), ind.tobytes())
    twin = _INT_TWIN.get(key)
    if twin is None:
        twin = g.Matrix(n, n, np.int32)
        assert twin.build_csr(ptr, ind, np.ones(ind.size, dtype=np.int32)) == 0
        _INT_TWIN.clear()
        _INT_TWIN[key] = twin
    if which in ("bfs", "bfs_opbyop") and A.np_dtype == np.float32:
        v = g.Vector(n)
        info, _ = g.bfs(v, A, arg, d, fused=(which == "bfs"))
        want = sr.bfs(ptr, ind, arg)[0]
        if info != 0 or not np.array_equal(v.extractTuples()[1], want):
            raise AssertionError("interfering %s from %d: labels differ from SimpleReferenceBfs (info %d)" % (which, arg, info))
    elif which == "sssp" and A.np_dtype == np.float32:
        v = g.Vector(n)
        info, _ = g.sssp(v, A, arg, d)
        want = sr.sssp(ptr, ind, val, arg)[0]
        if info != 0 or not np.array_equal(v.extractTuples()[1], want):
            raise AssertionError("interfering sssp from %d: distances differ from SimpleReferenceSssp (info %d)" % (arg, info))
    elif which == "pr":
        g.pr(g.Vector(n), A, 0.85, 0.0, hb.descriptor(mxvmode=2, max_niter=3))
    elif which == "cc":
        v = g.Vector(n, np.int32)
        info, _ = g.cc(v, twin, 0, d)
        # FastSV labels are only defined for symmetric patterns; check those
        if info == 0 and _is_symmetric(ptr, ind, n):
            want, k, _ = sr.cc(ptr, ind)
            if not np.array_equal(v.extractTuples()[1], sr.cc_canonical(want)):
                raise AssertionError("interfering cc: labels differ from SimpleReferenceCc")
    elif which == "tc":
        L, B = g.Matrix(n, n, np.int32), g.Matrix(n, n, np.int32)
        if g.tril(L, twin, d) == 0:
            g.tc(L, B, d)
    elif which == "trace":
        g.traceMxmTranspose("PlusMultiplies", A, A, d)
    elif which == "mis":
        g.mis(g.Vector(n, np.int32), twin, arg, d)
    elif which == "gc":
        g.gc(g.Vector(n, np.int32), twin, arg, 4096, arg % 3, d)
    elif which == "lgc":
        g.lgc(g.Vector(n), A, arg, 0.1, 1e-6, hb.descriptor(mxvmode=0, max_niter=4))
    elif which == "reduce_rows":
        g.reduce(None, "Plus", A, d, w=g.Vector(n))


def short(call):
    return tuple(("<%d values>" % len(c)) if isinstance(c, np.ndarray) else c for c in call)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=200)
    ap.add_argument("--len", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--n", type=int, default=70)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--struconly", action="store_true",
                    help="the traversal drivers' op mix with struconly / opreuse drawn at random")
    ap.add_argument("--int", action="store_true", help="int32 vectors and matrix instead of float")
    ap.add_argument("--only", type=int, default=-1, help="replay one sequence and dump the state before its last call")
    ap.add_argument("--self-check", action="store_true", help="oracle against itself (harness check, no GPU)")
    args = ap.parse_args()
    INT_MODE[0] = bool(args.int)
    BFS_MIX[0] = bool(args.struconly)
    hb, ob = (OracleBackend() if args.self_check else HipBackend()), OracleBackend()
    from graphblast_amd.graphgen import finalize_edges
    failures, calls_run = 0, 0
    for s in (range(args.seqs) if args.only < 0 else [args.only]):
        rng = np.random.default_rng(args.seed * 100003 + s)
        n = int(rng.choice([args.n, 200, 1000])) if s % 3 else args.n
        m = int(n * rng.integers(2, 8))
        es, ed = rng.integers(0, n, m), rng.integers(0, n, m)
        if s % 5 == 4:
            # a row and a column far longer than an SpMV wave tile (512) and a long-row slice
            n = 3000
            es, ed = rng.integers(0, n, m), rng.integers(0, n, m)
            hub = int(rng.integers(0, n))
            hd = rng.integers(0, n, 9000)
            es, ed = np.concatenate([es, np.full(9000, hub), hd]), np.concatenate([ed, hd, np.full(9000, hub)])
        g = finalize_edges(es, ed, n, symmetrize=bool(rng.random() < 0.5))
        ptr, ind = g["csr"]
        val = rng.integers(1, 4, ind.size).astype(F)
        DT = np.int32 if args.int else F
        mats = [be.matrix_from_csr(n, ptr, ind, val.astype(DT), DT) for be in (hb, ob)]
        so = int(args.struconly and rng.random() < 0.5)
        dargs = dict(mxvmode=int(rng.choice([0, 1, 2])), struconly=so, fusedmask=int(rng.random() < 0.5),
                     opreuse=int(args.struconly and rng.random() < 0.5), earlyexit=int(rng.random() < 0.5))
        descs = [be.descriptor(**dargs) for be in (hb, ob)]
        STRUC_ONLY[0] = bool(so)
        pools = []
        for be in (hb, ob):
            vs = [be.vector(n, DT) for _ in range(NVEC)]
            for v in vs:
                be.fill(v, 0.0)
            pools.append(vs)
        history = []
        for step in range(args.len):
            call = gen_call(rng, n)
            if call[0] in ("vxm", "mxv") and call[2] is not None and ob.storage(pools[1][call[2]]) == 0:
                # a mask that was never given a storage type: the reference errors out of vxm between
                # its two descriptor toggles (operations.hpp:109 / :205) and leaves GrB_INP1 flipped;
                # neither side reproduces that, so such calls are not generated
                call = call[:2] + (None,) + call[3:]
            if call[0] == "assign" and STRUC_ONLY[0] and ob.storage(pools[1][call[1]]) == 1:
                # assignSparse prunes by VALUE (assign.hpp:204-224); a struconly sparse vector has none
                continue
            if call[0] in ("vxm", "mxv") and ob.storage(pools[1][call[5]]) == 0:
                # an input without a storage type: whether the frontend lets it through depends on a
                # cached nvals_ (vector.hpp:133-146) that inspecting the state from outside also refreshes
                dense = [k for k in range(NVEC) if ob.storage(pools[1][k]) != 0]
                if not dense:
                    continue
                call = call[:5] + (dense[int(rng.integers(0, len(dense)))],)
            if call[0] in ("vxm", "mxv") and call[1] in (call[2], call[5]):
                # output aliasing the mask or the input: the reference's kernels then read what other
                # threads are writing (no defined result to compare)
                call = (call[0], (max(call[2] or 0, call[5]) + 1 + call[1]) % NVEC) + call[2:]
                if call[1] in (call[2], call[5]):
                    call = (call[0], (call[1] + 1) % NVEC) + call[2:]
                if call[1] in (call[2], call[5]):
                    call = (call[0], (call[1] + 1) % NVEC) + call[2:]
            if call[0] == "eWiseMult" and call[2] is not None and call[1] in (call[4], call[5]):
                # masked eWiseMult in place: with a sparse mask the reference writes w's index array at
                # the mask's positions while other threads still search it (ewisemult.hpp:178-270)
                w = next(k for k in range(NVEC) if k not in (call[4], call[5]))
                call = (call[0], w) + call[2:]
            history.append(short(call))
            if args.only >= 0:
                pre = (state(hb, pools[0], descs[0]), state(ob, pools[1], descs[1]), [ (be.get(d, 0) if hasattr(be, "get") else None) for be, d in ((hb, descs[0]), (ob, descs[1]))])
            r_h = run_call(hb, call, pools[0], mats[0], descs[0])
            r_o = run_call(ob, call, pools[1], mats[1], descs[1])
            calls_run += 1
            if args.only >= 0 and not args.self_check:
                for k in range(NVEC):
                    hv, ov = pools[0][k], pools[1][k]
                    st = hb.storage(hv)
                    if st == 1:                       # peek at the dense buffer behind a sparse vector
                        hv.setStorage(2)
                        hid = np.asarray(hb.dense_values(hv)).copy()
                        hv.setStorage(1)
                        if not np.array_equal(hid, ov.d_val, equal_nan=True):
                            bad = np.nonzero(~((hid == ov.d_val) | (np.isnan(hid) & np.isnan(ov.d_val))))[0][:8]
                            print("step %d %s: hidden dense buffer of sparse v%d differs at %s: %s vs %s" % (
                                step, short(call), k, bad, hid[bad], ov.d_val[bad]))
            why = None
            if isinstance(r_h, tuple):
                if r_h[0] != r_o[0] or not (np.isclose(r_h[1], r_o[1], rtol=RTOL, atol=0) or (not np.isfinite(r_h[1]) and not np.isfinite(r_o[1]))):   # inf - inf: order-dependent
                    why = "result %s vs %s" % (r_h, r_o)
            elif r_h != r_o:
                why = "info %s vs %s" % (r_h, r_o)
            if why is None:
                why = differs(state(hb, pools[0], descs[0]), state(ob, pools[1], descs[1]))
            if why is not None:
                failures += 1
                print("SEQ %d (n=%d) step %d: %s" % (s, n, step, why))
                for h in history[-6:]:
                    print("     ", h)
                if args.only >= 0:
                    np.set_printoptions(linewidth=200, threshold=100000)
                    for k in range(NVEC):
                        for side, st in (("hip", pre[0][0][k]), ("ora", pre[1][0][k])):
                            print("pre v%d %s storage %d" % (k, side, st[0]),
                                  "" if st[1] is None else ("nvals %d ind %s" % (len(st[1]), st[1][:40])),
                                  "" if st[2] is None else ("val %s" % st[2][:40]))
                    print("desc GrB_MASK hip/ora:", descs[0].get(0), descs[1].get(0), "mxvmode", descs[0].get(8), descs[1].get(8))
                    if not args.self_check:
                        for k in range(NVEC):
                            hv, ov = pools[0][k], pools[1][k]
                            st = hb.storage(hv)
                            hv.setStorage(2)
                            hid = np.asarray(hb.dense_values(hv)).copy()
                            hv.setStorage(st)
                            print("post-hidden v%d hip d_val %s" % (k, hid[:40]))
                            print("post-hidden v%d ora d_val %s" % (k, ov.d_val[:40]))
                    post = (state(hb, pools[0], descs[0]), state(ob, pools[1], descs[1]))
                    for k in range(NVEC):
                        for side, st in (("hip", post[0][0][k]), ("ora", post[1][0][k])):
                            print("post v%d %s storage %d" % (k, side, st[0]),
                                  "" if st[1] is None else ("nvals %d ind %s" % (len(st[1]), st[1][:40])),
                                  "" if st[2] is None else ("val %s" % st[2][:40]))
                break
        if args.verbose:
            print("seq", s, "ok" if why is None else "FAIL")
    print("sequences: %d, calls: %d, diverging sequences: %d" % (args.seqs, calls_run, failures))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
