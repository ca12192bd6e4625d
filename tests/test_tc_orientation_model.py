"""The triangle count on the degree-ordered orientation (csrc/tc_count.hip) restated on the host: the ranking (degree, ties by
id), the numbering that leaves one code out, the lists of higher-ranked neighbours in a narrow and a wide part, the pivot rule
(the end with the longer list, the lower-ranked end on a tie) and the rule that a pivot numbered up to the code + 1 does not
stream its partners' wide parts.  The model's count must equal the reference's SimpleReferenceTc on the caller's lower triangle
whatever the left-out code is -- 65 535 in the library, small numbers here so that graphs of a few hundred vertices cross it."""
import numpy as np
import pytest

from oracle import simple_reference as sr


def _lower(n, edges):
    a = np.zeros((n, n), dtype=bool)
    for u, v in edges:
        if u != v:
            a[u, v] = a[v, u] = True
    rows, cols = np.nonzero(np.tril(a, -1))
    lp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=lp[1:])
    return a, lp, cols.astype(np.int32)


def oriented_count(a, code):
    n = a.shape[0]
    deg = a.sum(1)
    order = sorted(range(n), key=lambda v: (-deg[v], v))            # rank 0 = the highest degree, ties by id
    number = {v: r + (1 if r >= code else 0) for r, v in enumerate(order)}   # nobody is numbered `code`
    lists = {}
    for v in range(n):
        mine = sorted(number[u] for u in np.nonzero(a[v])[0] if number[u] < number[v])
        lists[number[v]] = ([x for x in mine if x < code], [x for x in mine if x > code])     # narrow part, wide part
    total, streamed_wide = 0, 0
    for u, v in zip(*np.nonzero(np.tril(a, -1))):
        lo, hi = max(number[u], number[v]), min(number[u], number[v])     # lo: the lower-ranked end
        len_lo, len_hi = sum(map(len, lists[lo])), sum(map(len, lists[hi]))
        pivot, partner = (lo, hi) if len_hi <= len_lo else (hi, lo)
        table = set(lists[pivot][0]) | set(lists[pivot][1])
        narrow, wide = lists[partner]
        total += sum(x in table for x in narrow)
        if pivot > code + 1:                                             # the pivot's list reaches beyond the narrow numbers
            total += sum(x in table for x in wide)
            streamed_wide += len(wide)
        else:
            assert not any(x in table for x in wide)                      # what is skipped could not have hit
    return total, streamed_wide


@pytest.mark.parametrize("code", [0, 1, 7, 40, 10 ** 6])
def test_oriented_count_equals_the_reference(code):
    rng = np.random.default_rng(code + 5)
    for trial in range(12):
        n = int(rng.integers(3, 120))
        m = int(rng.integers(0, 8 * n))
        if trial % 3 == 0:                                               # hubs
            edges = [(int(rng.random() ** 3 * n), int(rng.integers(0, n))) for _ in range(m)]
        else:
            edges = [(int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(m)]
        a, lp, li = _lower(n, edges)
        want = sr.tc(lp, li)[0] if li.size else 0
        got, wide = oriented_count(a, code)
        assert got == want, (code, trial, n, m, got, want)


def test_no_list_is_longer_than_the_square_root_bound():
    rng = np.random.default_rng(1)
    n = 400
    edges = [(int(rng.random() ** 4 * n), int(rng.integers(0, n))) for _ in range(6000)]
    a, lp, li = _lower(n, edges)
    deg = a.sum(1)
    order = sorted(range(n), key=lambda v: (-deg[v], v))
    rank = {v: r for r, v in enumerate(order)}
    longest = max(sum(rank[u] < rank[v] for u in np.nonzero(a[v])[0]) for v in range(n))
    assert longest <= int(np.sqrt(2 * li.size)) + 1                       # a vertex with k higher-ranked neighbours: k (k + 1) / 2 <= edges
    assert int(np.diff(lp).max()) > longest                               # ... while the caller's numbering has longer rows
