"""The argument csrc/sssp_nearfar.hip rests on, checked on the host without a GPU: for non-negative integer weights
(exact float sums) the synchronous rounds of algorithm/sssp.hpp:53-90 end with
  * the distances of the fixed point d[v] = min_u (d[u] + w(u, v)), whatever order reaches it, and
  * a loop counter of max_v h(v) + 1, h(v) = the fewest edges among the shortest paths to v,
where (d, h) is the fixed point of the LEXICOGRAPHIC relaxation -- computed here by a Dijkstra on (distance, hops)
keys, an order as different from the rounds as it gets."""
import heapq

import numpy as np
import pytest

FMAX = np.finfo(np.float32).max


def _rounds(ptr, ind, w, src, max_niter=100000):
    """the reference's loop: round r relaxes every stored edge against round r - 1's distances"""
    n = ptr.size - 1
    d = np.full(n, FMAX, dtype=np.float32)
    d[src] = 0
    rows = np.repeat(np.arange(n), np.diff(ptr))
    for it in range(1, max_niter + 1):
        cand = (d[rows] + w).astype(np.float32)
        y = d.copy()
        np.minimum.at(y, ind, cand)
        if not (y < d).any():
            return d, it
        d = y
    return d, max_niter + 1


def _lexicographic_fixed_point(ptr, ind, w, src):
    n = ptr.size - 1
    dist = np.full(n, FMAX, dtype=np.float32)
    hops = np.full(n, -1, dtype=np.int64)
    best = {src: (np.float32(0), 0)}
    heap = [(0.0, 0, src)]
    done = np.zeros(n, dtype=bool)
    while heap:
        d, h, u = heapq.heappop(heap)
        if done[u]:
            continue
        done[u] = True
        dist[u], hops[u] = d, h
        for p in range(ptr[u], ptr[u + 1]):
            v = int(ind[p])
            key = (float(np.float32(np.float32(d) + w[p])), h + 1)
            if not done[v] and (v not in best or key < (float(best[v][0]), best[v][1])):
                best[v] = (np.float32(key[0]), key[1])
                heapq.heappush(heap, (key[0], key[1], v))
    return dist, hops


def _graph(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "grid":
        side = int(rng.integers(6, 14))
        n = side * side
        e = []
        for y in range(side):
            for x in range(side):
                v = y * side + x
                if x + 1 < side and rng.random() < 0.8:
                    e.append((v, v + 1))
                if y + 1 < side and rng.random() < 0.8:
                    e.append((v, v + side))
        e = np.array(e + [(b, a) for a, b in e], dtype=np.int64)
    else:
        n = int(rng.integers(20, 120))
        m = int(n * rng.integers(2, 6))
        e = np.stack([rng.integers(0, n, m), rng.integers(0, n, m)], 1)
        e = e[e[:, 0] != e[:, 1]]
    e = np.unique(e, axis=0)
    order = np.lexsort((e[:, 1], e[:, 0]))
    e = e[order]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, e[:, 0] + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr, e[:, 1].copy(), n


@pytest.mark.parametrize("kind", ["grid", "random"])
@pytest.mark.parametrize("seed", range(6))
def test_round_count_is_the_deepest_fewest_hops_shortest_path_plus_one(kind, seed):
    ptr, ind, n = _graph(kind, seed)
    rng = np.random.default_rng(100 + seed)
    w = rng.integers(0 if seed % 3 == 0 else 1, 17, ind.size).astype(np.float32)      # zero weights included
    src = int(np.argmax(np.diff(ptr)))
    d_rounds, it = _rounds(ptr, ind, w, src)
    d_fix, hops = _lexicographic_fixed_point(ptr, ind, w, src)
    assert np.array_equal(d_rounds, d_fix)
    assert it == int(hops.max()) + 1


def test_float_weights_same_distances():
    """with general float weights the fixed point (the distances) is still order-independent; only the COUNT may
    differ where rounding lets a not-yet-final distance produce a final one, which is why such matrices keep the
    rounds by default"""
    for seed in range(6):
        ptr, ind, n = _graph("random", 50 + seed)
        rng = np.random.default_rng(seed)
        w = (rng.random(ind.size) * 9 + 0.01).astype(np.float32)
        src = int(np.argmax(np.diff(ptr)))
        d_rounds, it = _rounds(ptr, ind, w, src)
        d_fix, hops = _lexicographic_fixed_point(ptr, ind, w, src)
        assert np.array_equal(d_rounds, d_fix)
        assert it <= int(hops.max()) + 1


def _queue_form(ptr, ind, w, src, delta, unit=False):
    """A sequential model of sssp_nfq_kernel's bookkeeping (csrc/sssp_nearfar.hip): (distance, hops) keys lowered by a
    lexicographic min, ONE entry per successful lowering and no "queued" bit -- an entry whose distance is no longer
    the key's is dropped when it is taken out --, entries at or above the threshold in a far pile that is dealt out
    when a pass leaves the next queue empty (threshold = smallest far distance + delta, at least the next float).
    Within a pass the entries are taken in a scrambled order: the kernel's order is whatever the hardware makes it.
    Returns distances, hops, passes, entries expanded."""
    n = ptr.size - 1
    rng = np.random.default_rng(7)
    dist = np.full(n, FMAX, dtype=np.float32)
    hops = np.full(n, -1, dtype=np.int64)
    dist[src], hops[src] = 0, 0
    T = np.float32(delta)
    near, far, minfar = [(np.float32(0), src)], [], None
    passes = expanded = 0
    while True:
        passes += 1
        assert passes < 100 * n + 100
        if not near:
            if not far:
                break
            up = np.nextafter(np.float32(minfar), np.float32(np.inf), dtype=np.float32)
            T = max(np.float32(np.float32(minfar) + np.float32(delta)), up)
            pile, far, minfar = far, [], None
            for d, v in pile:
                if dist[v] != d:
                    continue                                   # stale: lowered since, and entered again then
                if d < T:
                    near.append((d, v))
                else:
                    far.append((d, v))
                    minfar = d if minfar is None or d < minfar else minfar
            continue
        cur, near = near, []
        for k in rng.permutation(len(cur)):
            d, u = cur[k]
            if dist[u] != d:
                continue
            expanded += 1
            hu = hops[u]
            for p in range(ptr[u], ptr[u + 1]):
                v = int(ind[p])
                dn = np.float32(d + (np.float32(1) if unit else w[p]))
                if (float(dn), hu + 1) < (float(dist[v]), hops[v] if hops[v] >= 0 else 1 << 60):
                    dist[v], hops[v] = dn, hu + 1
                    if dn < T:
                        near.append((dn, v))
                    else:
                        far.append((dn, v))
                        minfar = dn if minfar is None or dn < minfar else minfar
    return dist, hops, passes, expanded


@pytest.mark.parametrize("kind", ["grid", "random"])
@pytest.mark.parametrize("seed", range(4))
def test_queue_form_reaches_the_same_fixed_point(kind, seed):
    """distances AND hop counts of the lexicographic fixed point, for narrow and wide bands, zero weights included;
    with unit weights (the BFS use of the kernel) the hops are the distances and no entry is ever stale"""
    ptr, ind, n = _graph(kind, 20 + seed)
    rng = np.random.default_rng(300 + seed)
    w = rng.integers(0 if seed % 2 == 0 else 1, 17, ind.size).astype(np.float32)
    src = int(np.argmax(np.diff(ptr)))
    d_fix, h_fix = _lexicographic_fixed_point(ptr, ind, w, src)
    for delta in (1.0, 8.0, 64.0, float(FMAX)):
        d, h, passes, expanded = _queue_form(ptr, ind, w, src, delta)
        assert np.array_equal(d, d_fix), delta
        assert np.array_equal(h[d < FMAX], h_fix[d_fix < FMAX]), delta
    ones = np.ones(ind.size, dtype=np.float32)
    d1, h1 = _lexicographic_fixed_point(ptr, ind, ones, src)
    d, h, passes, expanded = _queue_form(ptr, ind, ones, src, float(FMAX), unit=True)
    assert np.array_equal(d, d1) and np.array_equal(h[d < FMAX], h1[d1 < FMAX])
    reached = int((d < FMAX).sum())
    assert expanded == reached                                 # a vertex is expanded exactly once: a pass is a BFS level
    assert passes == int(h1.max()) + 2                         # one pass per level, the one that finds nothing, and the exit
