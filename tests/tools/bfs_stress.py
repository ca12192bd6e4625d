"""Differential stress of the one-launch BFS against the oracle's sequential BFS: many sources,
symmetric and directed RMAT, all three mxvmodes, with and without the edge-aware switch.
usage: python tests/tools/bfs_stress.py [scale] [sources]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import graphblast_amd as g
from graphblast_amd.graphgen import rmat_edges, finalize_edges, random_sources
from oracle import simple_reference as sr
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
s, d, n = rmat_edges(scale, 16, seed=3)
bad = 0
for sym in (True, False):
    gr = finalize_edges(s, d, n, symmetrize=sym)
    ptr, ind = gr["csr"]; cptr, cind = gr["csc"]
    A = g.Matrix(n, n)
    assert A.build_csr(ptr, ind, np.ones(ind.size, np.float32), csc=(cptr, cind, np.ones(cind.size, np.float32))) == 0
    srcs = random_sources(ptr, nsrc, seed=7)
    t0 = time.time()
    for i, src in enumerate(srcs):
        want = sr.bfs(ptr, ind, int(src))[0]
        for mode in (0, 1, 2):
            for es in (0.0, 0.08):
                dsc = g.Descriptor(); dsc.loadArgs(mxvmode=mode, struconly=1, opreuse=1, earlyexit=1, edgeswitch=es)
                v = g.Vector(n)
                info, r = g.bfs(v, A, int(src), dsc, fused=True)
                i2, got = v.extractTuples()
                if info != 0 or not np.array_equal(np.asarray(got, np.float32), want):
                    bad += 1
                    print("MISMATCH sym", sym, "src", src, "mode", mode, "es", es, "info", info)
    print("sym=%s: %d sources x 6 configurations checked in %.1f s" % (sym, len(srcs), time.time() - t0))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
