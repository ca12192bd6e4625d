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
