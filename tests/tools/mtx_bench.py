"""MatrixMarket ingest: text parsed on the device (grb_matrix_load_mtx) against the host paths.  GPU box.
python tests/tools/mtx_bench.py [scale]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, ".")
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges
from oracle import loader

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
s, d, n = rmat_edges(scale, 16, seed=1)
keep = s != d
lo, hi = np.minimum(s[keep], d[keep]), np.maximum(s[keep], d[keep])
key = np.unique(hi.astype(np.int64) * n + lo)
path = os.path.join(tempfile.mkdtemp(), "rmat%d.mtx" % scale)
with open(path, "w") as f:
    f.write("%%MatrixMarket matrix coordinate pattern symmetric\n")
    f.write("%d %d %d\n" % (n, n, key.size))
    np.savetxt(f, np.stack([key // n + 1, key % n + 1], 1), fmt="%d")
print("file: %.1f MB, %d entries (symmetric: %d stored after the loader)" % (os.path.getsize(path) / 1e6, key.size, 2 * key.size))
g.Matrix.from_mtx(path)                      # warm-up (context, first allocations)
t0 = time.perf_counter()
A = g.Matrix.from_mtx(path)
t1 = time.perf_counter()
print("device parse + sort + CSR/CSC + plans: %.1f ms (%.2f GB/s of text)" % ((t1 - t0) * 1e3, os.path.getsize(path) / (t1 - t0) / 1e9))
t0 = time.perf_counter()
r, c, v, nr, nc, nv = loader.read_mtx(path, 0, np.float32)
t1 = time.perf_counter()
print("host numpy restatement of readMtx (parse + removeSelfloop + customSort): %.1f ms" % ((t1 - t0) * 1e3))
assert nv == A.nvals()
