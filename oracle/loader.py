"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the reference's MatrixMarket ingest and CSR/CSC build:

  read_banner / read_size  <- graphblas/mmio.hpp:128-199 (mm_read_banner),
                              :204-236 (mm_read_mtx_crd_size)
  read_mtx                 <- graphblas/util.hpp:363-430 (readMtx)
  _read_tuples             <- graphblas/util.hpp:197-258 (readTuples, 1-based -> 0-based,
                              pattern -> value 1.0)
  remove_selfloop          <- graphblas/util.hpp:263-329 (removeSelfloop: symmetrise,
                              sort, drop self loops + duplicates; compaction moves
                              indices but NOT values -- the quirk is reproduced)
  custom_sort              <- graphblas/util.hpp:169-195 (sort by (row, col))
  coo2csr / coo2csc        <- graphblas/util.hpp:501-572
  csr2csc                  <- graphblas/util.hpp:574-600

Pinned by tests/test_oracle_loader.py against (i) the row-degree vectors the
reference asserts in test/greduce.cu:65,72, (ii) the parse results of the
reference's own mmio.hpp compiled into oracle/_ref/libmmio_ref.so.
"""
import os
import numpy as np

Index = np.int32


def read_banner(line):
    """mm_read_banner: returns the 4-char typecode list [M, C|A, R|C|P|I, G|S|H|K]."""
    tok = line.split()
    if len(tok) != 5:
        raise ValueError("MM_PREMATURE_EOF")
    banner, mtx, crd, data_type, storage = [t.lower() if i else t for i, t in enumerate(tok)]
    if banner != "%%MatrixMarket":
        raise ValueError("MM_NO_HEADER")
    code = [" "] * 4
    if mtx != "matrix":
        raise ValueError("MM_UNSUPPORTED_TYPE")
    code[0] = "M"
    if crd == "coordinate":
        code[1] = "C"
    elif crd == "array":
        code[1] = "A"
    else:
        raise ValueError("MM_UNSUPPORTED_TYPE")
    m = {"real": "R", "complex": "C", "pattern": "P", "integer": "I"}
    if data_type not in m:
        raise ValueError("MM_UNSUPPORTED_TYPE")
    code[2] = m[data_type]
    s = {"general": "G", "symmetric": "S", "hermitian": "H", "skew-symmetric": "K"}
    if storage not in s:
        raise ValueError("MM_UNSUPPORTED_TYPE")
    code[3] = s[storage]
    return code


def custom_sort(rows, cols, vals):
    """customSort: order by (row, col). np.lexsort is stable; std::sort is not, so the
    relative order of exact duplicates is unspecified in the reference (only their
    values could differ, and duplicates are dropped right after)."""
    order = np.lexsort((cols, rows))
    return rows[order], cols[order], vals[order]


def remove_selfloop(rows, cols, vals, undirected, remove_self_loops=True):
    rows = np.asarray(rows, dtype=Index)
    cols = np.asarray(cols, dtype=Index)
    vals = np.asarray(vals)
    if undirected:
        off = rows != cols
        rows, cols, vals = (np.concatenate([rows, cols[off]]),
                            np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, vals[off]]))
    rows, cols, vals = custom_sort(rows, cols, vals)
    n = rows.size
    if n == 0:
        return rows, cols, vals
    dead = np.zeros(n, dtype=bool)
    if remove_self_loops:
        dead |= rows == cols
    dup = np.zeros(n, dtype=bool)
    dup[1:] = (rows[1:] == rows[:-1]) & (cols[1:] == cols[:-1])
    dead |= dup
    keep = ~dead
    kept = int(keep.sum())
    # indices are compacted; values are NOT moved (util.hpp:311-323), only truncated
    return rows[keep], cols[keep], vals[:kept].copy()


def read_mtx(path, directed=0, dtype=np.float32):
    """readMtx. directed: 0 = symmetric iff banner says so, 1 = force directed,
    2 = force undirected. Returns (rows, cols, vals, nrows, ncols, nvals)."""
    with open(path, "r") as f:
        code = read_banner(f.readline())
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nrows, ncols, nnz = (int(x) for x in line.split()[:3])
        body = f.read().split()
    is_sym = code[3] == "S"
    undirected = (is_sym or directed == 2) and directed != 1
    if code[2] == "P":
        arr = np.array(body[:2 * nnz], dtype=np.int64).reshape(-1, 2)
        vals = np.ones(arr.shape[0], dtype=dtype)
    else:
        arr3 = np.array(body[:3 * nnz], dtype=np.float64).reshape(-1, 3)
        arr = arr3[:, :2].astype(np.int64)
        raw = arr3[:, 2]
        if code[2] == "I":
            raw = raw.astype(np.int32)
        else:
            raw = raw.astype(np.float32)
        vals = raw.astype(dtype)
    rows = (arr[:, 0] - 1).astype(Index)
    cols = (arr[:, 1] - 1).astype(Index)
    rows, cols, vals = remove_selfloop(rows, cols, vals, undirected)
    rows, cols, vals = custom_sort(rows, cols, vals)
    return rows, cols, vals, nrows, ncols, int(rows.size)


def coo2csr(rows, cols, vals, nrows, ncols):
    rows = np.asarray(rows, dtype=Index)
    cols = np.asarray(cols, dtype=Index)
    rows, cols, vals = custom_sort(rows, cols, np.asarray(vals))
    ptr = np.zeros(nrows + 1, dtype=Index)
    if rows.size:
        counts = np.bincount(rows, minlength=nrows)
        ptr[1:] = np.cumsum(counts)
    return ptr, cols.astype(Index).copy(), vals.copy()


def coo2csc(rows, cols, vals, nrows, ncols):
    return coo2csr(cols, rows, vals, ncols, nrows)


def csr2csc(ptr, ind, val, nrows, ncols):
    rows = np.repeat(np.arange(nrows, dtype=Index), np.diff(ptr))
    return coo2csc(rows, ind, val, ncols, nrows)


def cache_name(fname, is_undirected=True, remove_self_loops=True):
    """convert(): binary cache file name, util.hpp:340-357."""
    d = os.path.dirname(fname) or "."
    b = os.path.basename(fname)
    return "%s/.%s.%s.%s.bin" % (d, b, "ud" if is_undirected else "d",
                                 "nosl" if remove_self_loops else "sl")


def write_cache(path, ptr, ind):
    """sparse_matrix.hpp:328-347: int32 nrows, nvals, rowptr[nrows+1], colind[nvals]."""
    with open(path, "wb") as f:
        np.array([ptr.size - 1, ind.size], dtype=np.int32).tofile(f)
        ptr.astype(np.int32).tofile(f)
        ind.astype(np.int32).tofile(f)


def read_cache(path):
    """sparse_matrix.hpp:353-407: values are implied 1."""
    raw = np.fromfile(path, dtype=np.int32)
    nrows, nvals = int(raw[0]), int(raw[1])
    ptr = raw[2:2 + nrows + 1].copy()
    ind = raw[2 + nrows + 1:2 + nrows + 1 + nvals].copy()
    return ptr, ind, np.ones(nvals, dtype=np.float32)
