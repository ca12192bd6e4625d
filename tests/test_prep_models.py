"""CPU: the preparation passes that moved from host loops to device kernels in round 5, restated in numpy step for
step and compared with the sequential rule they replace, on random and adversarial degree sequences.

* the SpMV band format's band cutting (csrc/spmv_cband.hpp: cband_light / cband_next / cband_chase / cband_place
  against the host pass kept in csrc/spmv.hip behind GRB_CB_PREP_HOST=1);
* where a big row enters each destination range (csrc/bfs_persist.hip: oc_range_off_kernel's gallop + bisection from
  the last position against a bisection of the whole row);
* an entry's row among a wave's 65 row pointers (cband_keys_kernel / bfs_hint_kernel: "the last row whose start is <= p",
  with empty rows and rows past the end).

The device kernels themselves are compared with these host forms on the GPU
(tests/test_gpu_spmv_cband.py::test_preparation_on_the_device_equals_the_host_pass, the BFS suites)."""
import numpy as np

K_ROWS = 16384          # kCbRows


def host_bands(deg, hub_above, max_entries, max_rows=K_ROWS):
    """spmv.hip's host pass: returns (band starts of the light bands, row_band, row_loc, band_first) with band numbers
    starting at 1 when there is a hub band"""
    n = len(deg)
    hub = deg > hub_above
    nhub = int(hub.sum())
    hub_entries = int(deg[hub].sum())
    band0 = 1 if nhub > 0 else 0
    starts, first = [0], [hub_entries]
    row_band = np.zeros(n, dtype=np.int64)
    row_loc = np.zeros(n, dtype=np.int64)
    at, in_band, r0, cur, hub_i = hub_entries, 0, 0, band0, 0
    for r in range(n):
        d = 0 if hub[r] else int(deg[r])
        if r > r0 and (r - r0 == max_rows or in_band + d > max_entries):
            at += in_band
            in_band, r0 = 0, r
            cur += 1
            starts.append(r)
            first.append(at)
        in_band += d
        row_band[r] = 0 if hub[r] else cur
        if hub[r]:
            row_loc[r] = hub_i
            hub_i += 1
        else:
            row_loc[r] = r - r0
    return np.array(starts), row_band, row_loc, np.array(first)


def device_bands(deg, hub_above, max_entries, max_rows=K_ROWS):
    """the device pass: light entries and hub flags, their exclusive scans, next[] by a search in the prefix sums, the chase,
    the placement"""
    n = len(deg)
    hub = deg > hub_above
    light = np.where(hub, 0, deg).astype(np.int64)
    P = np.concatenate([[0], np.cumsum(light)])             # P[r] = light entries in front of row r, P[n] = all
    H = np.concatenate([[0], np.cumsum(hub.astype(np.int64))])
    nhub = int(H[n])
    hub_entries = int(deg.sum()) - int(P[n])
    nxt = np.empty(n, dtype=np.int64)
    for r in range(n):                                       # cband_next_kernel
        lim = P[r] + max_entries
        lo, hi = r + 1, min(r + max_rows, n)
        while lo < hi:
            mid = lo + ((hi - lo) >> 1)
            if P[mid + 1] > lim:
                hi = mid
            else:
                lo = mid + 1
        nxt[r] = lo
    starts = []
    r = 0
    while r < n:                                             # cband_chase_kernel
        starts.append(r)
        r = int(nxt[r])
    starts = np.array(starts)
    band0 = 1 if nhub > 0 else 0
    row_band = np.zeros(n, dtype=np.int64)
    row_loc = np.zeros(n, dtype=np.int64)
    for r in range(n):                                       # cband_place_kernel
        if hub[r]:
            row_band[r], row_loc[r] = 0, H[r]
        else:
            b = int(np.searchsorted(starts, r, side="right")) - 1
            row_band[r], row_loc[r] = band0 + b, r - starts[b]
    first = hub_entries + P[starts]
    return starts, row_band, row_loc, first


def test_band_cutting_on_the_device_is_the_host_rule():
    rng = np.random.default_rng(4)
    cases = []
    for n in (1, 2, 63, 64, 65, 1000, 5000):
        cases.append((rng.integers(0, 9, n), 10 ** 9, 40, 100))                      # no hubs, cut by entries and by rows
        cases.append((rng.integers(0, 200, n), 150, 300, 37))                          # hubs inside the light bands
        cases.append((np.minimum(rng.zipf(1.6, n), 5000), 400, 1000, 64))              # power law
    cases.append((np.array([0, 0, 0, 0, 0]), 10 ** 9, 5, 2))                           # empty rows only
    cases.append((np.array([50, 1, 1, 50, 50, 1]), 10 ** 9, 10, 100))                  # rows above the limit on their own
    cases.append((np.full(300, 7), 10 ** 9, 21, 1000))                                 # exact fits: 3 rows make 21
    for deg, hub_above, max_entries, max_rows in cases:
        deg = np.asarray(deg, dtype=np.int64)
        a = host_bands(deg, hub_above, max_entries, max_rows)
        b = device_bands(deg, hub_above, max_entries, max_rows)
        for x, y, what in zip(a, b, ("starts", "row_band", "row_loc", "band_first")):
            assert np.array_equal(x, y), (what, len(deg), hub_above, max_entries, max_rows)


def walk_bounds(cols, bounds):
    """oc_range_off_kernel: one thread, the bounds in order, gallop + bisection from the last position"""
    e = len(cols)
    p = 0
    out = []
    loads = 0
    for key in bounds:
        lo = hi = p
        step = 1
        while hi < e and cols[hi] < key:
            loads += 1
            lo = hi + 1
            hi += step
            step <<= 1
        hi = min(hi, e)
        while lo < hi:
            mid = lo + (hi - lo) // 2
            loads += 1
            if cols[mid] < key:
                lo = mid + 1
            else:
                hi = mid
        p = lo
        out.append(p)
    return np.array(out), loads


def test_range_table_walk_equals_whole_row_bisection():
    rng = np.random.default_rng(8)
    n = 1 << 16
    for deg in (1, 5, 256, 3000, 40000):
        cols = np.sort(rng.choice(n, size=min(deg, n), replace=False))
        for nb in (2, 7, 256):
            cuts = np.sort(rng.choice(np.arange(1, n), size=nb - 1, replace=False))
            bounds = np.concatenate([[0], cuts, [n]])
            got, loads = walk_bounds(cols, bounds)
            assert np.array_equal(got, np.searchsorted(cols, bounds, side="left")), (deg, nb)
            assert got[0] == 0 and got[-1] == len(cols)
            # about one look at an entry, or a few per bound when the bounds are denser than the entries
            assert loads <= 3 * len(cols) + 4 * len(bounds) * max(1, int(np.log2(max(2, len(cols))))), (deg, nb, loads)


def row_of(starts65, p):
    """the 6-step search of the chunk kernels: the last row j in [0, 63] with starts65[j] <= p"""
    j = 0
    step = 32
    while step > 0:
        if starts65[j + step] <= p:
            j += step
        step >>= 1
    return j


def test_row_search_in_a_chunk_with_empty_rows_and_a_ragged_end():
    rng = np.random.default_rng(2)
    for rows_here in (64, 37, 1):
        deg = rng.integers(0, 6, rows_here)
        deg[rng.integers(0, rows_here, max(1, rows_here // 3))] = 0          # plenty of empty rows
        if deg.sum() == 0:
            deg[0] = 3
        ptr = np.concatenate([[0], np.cumsum(deg)]) + 1000
        starts = np.concatenate([ptr[:-1], np.full(64 - rows_here, ptr[-1]), [ptr[-1]]])   # rows past the end start at the end
        assert len(starts) == 65
        want = np.repeat(np.arange(rows_here), deg)
        got = np.array([row_of(starts, p) for p in range(ptr[0], ptr[-1])])
        assert np.array_equal(got, want), rows_here


def test_label_pass_lane_mapping_covers_every_vertex_once():
    """bfs_persist.hip's label pass: a wave takes 32 bitmap words; lane l < 32 loads word l and forms its six label planes
    (bit b of plane k = bit k of the label of vertex 32 w + b); store q (0 .. 3) of lane l writes the four labels of
    vertices 4 (l & 7) .. + 3 of word 8 q + (l >> 3), taking the planes from the lane that loaded that word -- at float
    offset 256 q + 4 l of the wave's 1024 labels, i.e. 64 x 16 consecutive bytes per store instruction."""
    rng = np.random.default_rng(6)
    levels = 9
    nwords = 32
    # disjoint level bitmaps F[0 .. levels): every vertex in at most one
    owner = rng.integers(-1, levels, nwords * 32)                   # -1: never reached
    F = np.zeros((levels, nwords), dtype=np.uint64)
    for v, L in enumerate(owner):
        if L >= 0:
            F[L, v // 32] |= np.uint64(1) << np.uint64(v % 32)
    planes = np.zeros((nwords, 6), dtype=np.uint64)                  # what lane w (the loader of word w) holds
    for L in range(levels):
        for k in range(6):
            if ((L + 1) >> k) & 1:
                planes[:, k] |= F[L]
    out = np.full(nwords * 32, -1.0)
    written = np.zeros(nwords * 32, dtype=int)
    for q in range(4):
        for lane in range(64):
            src = q * 8 + (lane >> 3)
            b0 = (lane & 7) * 4
            base = q * 256 + lane * 4                                 # float offset inside the wave's 1024 labels
            for t in range(4):
                lab = 0
                for k in range(6):
                    lab |= int((planes[src, k] >> np.uint64(b0 + t)) & np.uint64(1)) << k
                out[base + t] = lab
                written[base + t] += 1
            assert base == src * 32 + b0                              # the address IS the vertex: word src, bit b0
    assert np.all(written == 1)
    assert np.array_equal(out, np.where(owner >= 0, owner + 1, 0).astype(float))
