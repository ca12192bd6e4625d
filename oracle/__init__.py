"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's (gunrock/graphblast) mxv/vxm hot path, used as
the parity checker.  Only tests/, __graft_entry__.smoke() and the `cpu_baseline`
leg of bench.py may import, call, link or execute anything in this package -- and
there only as the checker / CPU baseline, never as the thing measured or shipped.
The product (graphblast_amd/) never imports it and fails loudly without its HIP
library.

Contents
  loader.py            MatrixMarket ingest + COO->CSR/CSC   (graphblas/util.hpp, mmio.hpp)
  semiring.py          operator / monoid / semiring table   (graphblas/stddef.hpp)
  ops.py               backend containers + mxv/vxm/assign/reduce/eWise* dispatch
  algorithms.py        algorithm/{bfs,sssp,pr}.hpp loops over ops.py
  simple_reference.c   SimpleReference{Bfs,Sssp,Pr,Cc,Tc} in plain C (+ ctypes wrapper
                       simple_reference.py); also the timed CPU baseline of bench.py
  Makefile             builds liboracle.so and, when /root/reference is present,
                       oracle/_ref/ (the parts of the reference that compile from their
                       own sources: mmio.hpp, stddef.hpp)

Parity status (see DESIGN.md "Oracle"):
  pinned   loader (test/greduce.cu:65,72 + reference mmio.hpp via _ref),
           semirings (reference stddef.hpp via _ref -> tests/golden/semiring_ref.json),
           per-op semantics (literal cases of test/gvxm.cu, gewiseadd.cu, gewisemult.cu,
           greduce.cu -> tests/golden/ref_tests.json)
  known-answer only   SimpleReference* algorithm oracles (the reference headers need
           Boost and cannot be built here; pinned by the chesapeake / test_cc / test_bc
           answers recorded in SURVEY.md 8(c) and by scipy.sparse.csgraph cross-checks)
"""
