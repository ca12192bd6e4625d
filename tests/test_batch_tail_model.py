"""CPU: a host model of the light-level launch of grb_bfs_batch (csrc/bfs_batch.hip: batch_tail_kernel)
-- the frontier as a queue, one 64-bit word per vertex (bit s = source s), a claimed target queued by whoever finds its
word zero, the words a level read cleaned through its queue WHILE THE NEXT LEVEL RUNS.  The model runs the cleaner of
level j - 1 after the claims of level j (the order a slow workgroup produces on the device) and shows

  * with THREE word arrays in rotation the depth vectors are those of one BFS per source, on paths, grids and random
    digraphs, repeated sources and isolated vertices included;
  * with TWO arrays (the first version of the kernel) a claim is lost exactly when a vertex one source reached two
    levels ago is claimed for another source now -- the word the cleaner zeroes is the word the claimer just wrote.

No GPU, no library: the device kernel itself is checked against the oracle in tests/test_gpu_batch.py."""
import numpy as np
import pytest


def bfs_depths(n, adj, src):
    d = np.zeros(n, dtype=np.int64)
    d[src] = 1
    frontier, level = [src], 1
    while frontier:
        level += 1
        nxt = []
        for u in frontier:
            for v in adj[u]:
                if d[v] == 0:
                    d[v] = level
                    nxt.append(v)
        frontier = nxt
    return d


def tail_model(n, adj, sources, narrays, rng=None):
    """-> depth[s][v] as the launch would label them.  narrays = 3: the kernel; 2: its first version."""
    k = len(sources)
    seen = [0] * n
    depth = np.zeros((k, n), dtype=np.int64)
    f0 = [0] * n                                          # the stored words of the level before (never cleaned)
    for s, v in enumerate(sources):
        seen[v] |= 1 << s
        f0[v] |= 1 << s
        depth[s][v] = 1
    X = [[0] * n for _ in range(narrays)]
    queues = {0: sorted(set(sources))}
    j = 0
    while queues[j]:
        F = f0 if j == 0 else X[(j - 1) % narrays]
        Xn = X[j % narrays]
        nxt = []
        order = list(queues[j])
        if rng is not None:
            rng.shuffle(order)
        for u in order:
            fw = F[u]
            for v in adj[u]:
                bits = fw & ~seen[v]
                if not bits:
                    continue
                seen[v] |= bits                           # the claim: exactly once per (vertex, source)
                if Xn[v] == 0:
                    nxt.append(v)                         # whoever finds the word zero queues the vertex
                Xn[v] |= bits
                for s in range(k):
                    if (bits >> s) & 1:
                        depth[s][v] = j + 2               # labelled as claimed
        # the cleaner of the words level j - 1 read, running late: after this level's claims
        if j >= 2:
            Xc = X[(j - 2) % narrays]
            for u in queues[j - 1]:
                Xc[u] = 0
        queues[j + 1] = nxt
        j += 1
    return depth


def path(n):
    return [[v for v in (u - 1, u + 1) if 0 <= v < n] for u in range(n)]


def grid(side, keep, rng):
    n = side * side
    adj = [[] for _ in range(n)]
    for r in range(side):
        for c in range(side):
            u = r * side + c
            for v in ((u + 1) if c + 1 < side else -1, (u + side) if r + 1 < side else -1):
                if v >= 0 and rng.random() < keep:
                    adj[u].append(v)
                    adj[v].append(u)
    return n, adj


def test_three_arrays_give_every_sources_bfs():
    rng = np.random.default_rng(3)
    cases = [(9, path(9), [2, 0, 8, 2])]
    n, adj = grid(14, 0.7, rng)
    cases.append((n, adj, [int(x) for x in rng.integers(0, n, 24)]))
    n = 120
    adj = [sorted(set(int(x) for x in rng.integers(0, n, int(rng.integers(0, 4))))) for _ in range(n)]   # a digraph
    cases.append((n, adj, [int(x) for x in rng.integers(0, n, 40)] + [5, 5]))
    for n, adj, sources in cases:
        want = np.stack([bfs_depths(n, adj, s) for s in sources])
        for trial in range(4):
            got = tail_model(n, adj, sources, 3, np.random.default_rng(trial))
            assert np.array_equal(got, want)


def test_two_arrays_lose_a_claim_on_a_seen_vertex():
    """path 0 - 1 - ... - 8, source a = 2, source b = 0: a reaches vertex 3 at level 1, so 3 is in the frontier level 2
    reads; b claims 3 at level 3 -- into the array whose level-2 words are being cleaned.  With two arrays the late
    cleaner wipes b's bit, b's traversal stops at 3, vertices 4 .. 8 keep depth 0 for b."""
    n, adj, sources = 9, path(9), [2, 0]
    want = np.stack([bfs_depths(n, adj, s) for s in sources])
    bad = tail_model(n, adj, sources, 2)
    assert np.array_equal(bad[0], want[0])                 # the source that runs ahead is not affected
    assert bad[1][3] == want[1][3]                         # the claim itself was labelled ...
    assert not np.array_equal(bad[1], want[1]) and np.all(bad[1][4:] == 0)   # ... and then lost to the cleaner
    assert np.array_equal(tail_model(n, adj, sources, 3), want)
