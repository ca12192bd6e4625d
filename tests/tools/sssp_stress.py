"""Differential stress of the one-launch SSSP against the oracle's Dijkstra (integer weights, so
float sums are exact): many sources, symmetric and directed RMAT plus a grid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, grid_edges, finalize_edges, random_sources
from oracle import simple_reference as sr
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
s, d, n = rmat_edges(scale, 16, seed=5)
cases = [("rmat_sym", finalize_edges(s, d, n, symmetrize=True)), ("rmat_dir", finalize_edges(s, d, n, symmetrize=False))]
gs, gd, gn = grid_edges(300, keep=0.75)
cases.append(("grid300", finalize_edges(gs, gd, gn, symmetrize=True)))
rng = np.random.default_rng(9)
for name, gr in cases:
    ptr, ind = gr["csr"]; nn = gr["n"]
    w = rng.integers(1, 65, ind.size).astype(np.float32)
    A = g.Matrix(nn, nn)
    assert A.build_csr(ptr, ind, w) == 0
    t0 = time.time()
    for src in random_sources(ptr, nsrc, seed=3):
        want = sr.sssp(ptr, ind, w, int(src))[0]
        for mode in (0, 2):
            dsc = g.Descriptor(); dsc.loadArgs(mxvmode=mode)
            v = g.Vector(nn)
            info, r = g.sssp(v, A, int(src), dsc)
            i2, got = v.extractTuples()
            if info != 0 or not np.array_equal(np.asarray(got, np.float32), np.asarray(want, np.float32)):
                bad += 1
                print("MISMATCH", name, src, mode, info)
    print("%s: %d sources checked in %.1f s" % (name, nsrc, time.time() - t0))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
