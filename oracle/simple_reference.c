/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the
 * product path (graphblast_amd/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker / CPU baseline.
 *
 * Plain-C restatement of the reference's sequential CPU oracles:
 *   oracle_bfs   <- graphblas/algorithm/test_bfs.hpp:11-61   (SimpleReferenceBfs)
 *   oracle_sssp  <- graphblas/algorithm/test_sssp.hpp:14-79  (SimpleReferenceSssp)
 *   oracle_pr    <- graphblas/algorithm/test_pr.hpp:14-78    (SimpleReferencePr)
 *   oracle_cc    <- graphblas/algorithm/test_cc.hpp:14-56    (SimpleReferenceCc)
 *   oracle_cc_verify <- test_cc.hpp:58-95                    (SimpleVerifyCc)
 *   oracle_tc    <- graphblas/algorithm/test_tc.hpp:14-84    (SimpleReferenceTc)
 *   oracle_mis / oracle_mis_verify <- graphblas/algorithm/test_mis.hpp:12-56 / :65-115
 *   oracle_gc  / oracle_gc_verify  <- graphblas/algorithm/test_gc.hpp:13-57 / :59-98
 *   oracle_lgc   <- graphblas/algorithm/test_lgc.hpp:13-88   (SimpleReferenceLgc)
 *   oracle_bfs_do_stats : instrumented BFS that follows the direction decisions of
 *       backend/cuda/vector.hpp:291-323 (convert) + algorithm/bfs.hpp:48-82 and
 *       counts the per-level quantities SURVEY.md 8(d) needs (nf, mf, nu, mi).
 *
 * Parity status: PINNED.  The reference's own SimpleReference* / SimpleVerify* compile from
 * their sources (oracle/Makefile, target `ref` -> oracle/_ref/libsimple_ref*.so; util.hpp
 * without its one Boost user, parseArgs) and generated tests/golden/algo_ref.npz;
 * tests/test_oracle_pinned.py holds these restatements to it bit for bit.
 *
 * Index = int32, values = float32, exactly as graphblas/types.hpp:18-19.
 * Each timed function returns the elapsed milliseconds of the same region the
 * reference brackets with CpuTimer (the traversal loop only).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int32_t Index;

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---- BFS: depth labels, source = 1, unreachable = 0 (test_bfs.hpp:18-52) ---- */
double oracle_bfs(Index nrows, const Index* row_ptr, const Index* col_ind,
                  float* depth, Index src, Index stop, Index* search_depth_out) {
  for (Index i = 0; i < nrows; ++i) depth[i] = 0.f;
  depth[src] = 1.f;
  Index search_depth = 1;
  /* FIFO of discovered vertices: every vertex enters at most once. */
  Index* fifo = (Index*)malloc(sizeof(Index) * (size_t)(nrows > 0 ? nrows : 1));
  Index head = 0, tail = 0;
  fifo[tail++] = src;
  double t0 = now_ms();
  while (head < tail) {
    Index node = fifo[head++];
    Index next_depth = (Index)depth[node] + 1;
    if (next_depth > stop) break;
    for (Index e = row_ptr[node]; e < row_ptr[node + 1]; ++e) {
      Index nb = col_ind[e];
      if (depth[nb] == 0.f) {
        depth[nb] = (float)next_depth;
        if (search_depth < next_depth) search_depth = next_depth;
        fifo[tail++] = nb;
      }
    }
  }
  double t1 = now_ms();
  free(fifo);
  if (search_depth_out) *search_depth_out = search_depth;
  return t1 - t0;
}

/* ---- SSSP: lazy Dijkstra with a binary min-heap on (distance, vertex), the
 * ordering std::greater<std::pair<T,Index>> induces (test_sssp.hpp:21-66).
 * The `processed` flag and the FLT_MAX-weight skip are reproduced. ---- */
typedef struct { float d; Index v; } HeapItem;

static int heap_less(HeapItem a, HeapItem b) {
  return (a.d < b.d) || (a.d == b.d && a.v < b.v);
}

double oracle_sssp(Index nrows, const Index* row_ptr, const Index* col_ind,
                   const float* val, float* dist, Index src,
                   Index* search_depth_out) {
  unsigned char* processed = (unsigned char*)calloc((size_t)nrows + 1, 1);
  for (Index i = 0; i < nrows; ++i) dist[i] = FLT_MAX;
  dist[src] = 0.f;
  size_t cap = 1024, sz = 0;
  HeapItem* heap = (HeapItem*)malloc(cap * sizeof(HeapItem));
  heap[sz].d = 0.f; heap[sz].v = src; sz++;
  Index search_depth = 0;
  double t0 = now_ms();
  while (sz > 0) {
    size_t round = sz;  /* the reference drains "frontier_size" pops per depth */
    for (size_t r = 0; r < round && sz > 0; ++r) {
      HeapItem top = heap[0];
      heap[0] = heap[--sz];
      size_t i = 0;
      for (;;) {  /* sift down */
        size_t l = 2 * i + 1, rr = l + 1, m = i;
        if (l < sz && heap_less(heap[l], heap[m])) m = l;
        if (rr < sz && heap_less(heap[rr], heap[m])) m = rr;
        if (m == i) break;
        HeapItem t = heap[i]; heap[i] = heap[m]; heap[m] = t; i = m;
      }
      Index node = top.v;
      float distance = top.d;
      processed[node] = 1;
      for (Index e = row_ptr[node]; e < row_ptr[node + 1]; ++e) {
        Index nb = col_ind[e];
        float w = val[e];
        if (!processed[nb] && w != FLT_MAX) {
          float nd = distance + w;
          if (nd < dist[nb]) {
            dist[nb] = nd;
            if (sz == cap) { cap *= 2; heap = (HeapItem*)realloc(heap, cap * sizeof(HeapItem)); }
            size_t j = sz++;
            heap[j].d = nd; heap[j].v = nb;
            while (j > 0) {  /* sift up */
              size_t p = (j - 1) / 2;
              if (!heap_less(heap[j], heap[p])) break;
              HeapItem t = heap[j]; heap[j] = heap[p]; heap[p] = t; j = p;
            }
          }
        }
      }
    }
    search_depth++;
  }
  double t1 = now_ms();
  free(heap);
  free(processed);
  if (search_depth_out) *search_depth_out = search_depth;
  return t1 - t0;
}

/* ---- PageRank: push-style power iteration, out-degree from row_ptr, values
 * ignored, stops when the SQUARED residual < eps (test_pr.hpp:22-66). All
 * arithmetic in float32 in the reference's order. ---- */
double oracle_pr(Index nrows, const Index* row_ptr, const Index* col_ind,
                 float* rank, float alpha, float eps, int max_niter,
                 int* iters_out, float* resultant_out) {
  float* next = (float*)malloc(sizeof(float) * (size_t)(nrows > 0 ? nrows : 1));
  float* outdeg = (float*)malloc(sizeof(float) * (size_t)(nrows > 0 ? nrows : 1));
  for (Index i = 0; i < nrows; ++i) rank[i] = 1.f / nrows;
  for (Index i = 0; i < nrows; ++i) outdeg[i] = (float)(row_ptr[i + 1] - row_ptr[i]);
  float resultant = 0.f;
  int depth = 0;
  double t0 = now_ms();
  for (int it = 0; it < max_niter; ++it) {
    for (Index v = 0; v < nrows; ++v) next[v] = (1.f - alpha) / nrows;
    for (Index v = 0; v < nrows; ++v) {
      float contrib = rank[v] / outdeg[v];
      for (Index e = row_ptr[v]; e < row_ptr[v + 1]; ++e)
        next[col_ind[e]] += alpha * contrib;
    }
    resultant = 0.f;
    for (Index v = 0; v < nrows; ++v) {
      float diff = rank[v] - next[v];
      resultant += diff * diff;
      rank[v] = next[v];
    }
    if (fabsf(resultant) < eps) break;
    depth++;
  }
  double t1 = now_ms();
  free(next);
  free(outdeg);
  if (iters_out) *iters_out = depth;
  if (resultant_out) *resultant_out = resultant;
  return t1 - t0;
}

/* ---- CC: iterative DFS labelling 1..k in vertex order (test_cc.hpp:20-50) ---- */
double oracle_cc(Index nrows, const Index* row_ptr, const Index* col_ind,
                 int* label, int* ncomp_out) {
  for (Index i = 0; i < nrows; ++i) label[i] = 0;
  size_t cap = 1024, sp = 0;
  Index* stack = (Index*)malloc(cap * sizeof(Index));
  int current = 0;
  double t0 = now_ms();
  for (Index i = 0; i < nrows; ++i) {
    if (label[i] == 0) current++;
    stack[sp++] = i;
    while (sp > 0) {
      Index cur = stack[--sp];
      if (label[cur] == 0) {
        label[cur] = current;
        for (Index e = row_ptr[cur]; e < row_ptr[cur + 1]; ++e) {
          Index c = col_ind[e];
          if (label[c] == 0) {
            if (sp == cap) { cap *= 2; stack = (Index*)realloc(stack, cap * sizeof(Index)); }
            stack[sp++] = c;
          }
        }
      }
    }
  }
  double t1 = now_ms();
  free(stack);
  if (ncomp_out) *ncomp_out = current;
  return t1 - t0;
}

/* SimpleVerifyCc: number of edges joining different labels; distinct label count. */
int oracle_cc_verify(Index nrows, const Index* row_ptr, const Index* col_ind,
                     const int* label, int* ndistinct_out) {
  int errors = 0;
  for (Index r = 0; r < nrows; ++r)
    for (Index e = row_ptr[r]; e < row_ptr[r + 1]; ++e)
      if (label[col_ind[e]] != label[r]) errors++;
  if (ndistinct_out) {
    /* insertion of labels into an open-addressed set */
    size_t m = 1; while (m < (size_t)nrows * 2 + 8) m <<= 1;
    int* set = (int*)malloc(m * sizeof(int));
    unsigned char* used = (unsigned char*)calloc(m, 1);
    int distinct = 0;
    for (Index r = 0; r < nrows; ++r) {
      uint32_t h = (uint32_t)label[r] * 2654435761u;
      size_t p = h & (m - 1);
      while (used[p] && set[p] != label[r]) p = (p + 1) & (m - 1);
      if (!used[p]) { used[p] = 1; set[p] = label[r]; distinct++; }
    }
    free(set); free(used);
    *ndistinct_out = distinct;
  }
  return errors;
}

/* ---- TC: sorted-list intersection over every stored edge (test_tc.hpp:41-70).
 * Called on L = tril(A) by the driver; counts each triangle once there. ---- */
double oracle_tc(Index nrows, const Index* row_ptr, const Index* col_ind,
                 long long* ntris_out) {
  long long ntris = 0;
  double t0 = now_ms();
  for (Index v = 0; v < nrows; ++v) {
    Index b1 = row_ptr[v], e1 = row_ptr[v + 1];
    for (Index e = b1; e < e1; ++e) {
      Index nb = col_ind[e];
      Index i = b1, j = row_ptr[nb], je = row_ptr[nb + 1];
      while (i < e1 && j < je) {
        Index a = col_ind[i], b = col_ind[j];
        if (a < b) ++i; else if (a > b) ++j; else { ++ntris; ++i; ++j; }
      }
    }
  }
  double t1 = now_ms();
  if (ntris_out) *ntris_out = ntris;
  return t1 - t0;
}

/* ---- MIS: greedy over a vertex order (test_mis.hpp:35-52).  The reference draws the order
 * with std::shuffle(std::mt19937(seed)) -- a libstdc++-specific stream -- so the order is an
 * argument here; its own check of a result is the property test below, not this sequence. ---- */
double oracle_mis(Index nrows, const Index* row_ptr, const Index* col_ind, const Index* order, int32_t* mis) {
  unsigned char* candidate = (unsigned char*)malloc((size_t)nrows + 1);
  memset(candidate, 1, (size_t)nrows + 1);
  for (Index i = 0; i < nrows; ++i) mis[i] = 0;
  double t0 = now_ms();
  for (Index i = 0; i < nrows; ++i) {
    Index row = order[i];
    if (!candidate[row]) continue;
    mis[row] = 1;
    candidate[row] = 0;
    for (Index e = row_ptr[row]; e < row_ptr[row + 1]; ++e) candidate[col_ind[e]] = 0;
  }
  double t1 = now_ms();
  free(candidate);
  return t1 - t0;
}

/* SimpleVerifyMis (test_mis.hpp:65-115): errors = adjacent pairs both in the set (counted per
 * stored edge) + vertices neither in the set nor adjacent to it.  0 <=> "CORRECT". */
int oracle_mis_verify(Index nrows, const Index* row_ptr, const Index* col_ind, const int32_t* mis,
                      int* set_size_out) {
  int flag = 0, set_size = 0;
  unsigned char* discovered = (unsigned char*)calloc((size_t)nrows + 1, 1);
  for (Index row = 0; row < nrows; ++row) {
    if (mis[row] != 1) continue;
    ++set_size;
    discovered[row] = 1;
    for (Index e = row_ptr[row]; e < row_ptr[row + 1]; ++e) {
      Index col = col_ind[e];
      if (mis[col] == 1) ++flag;
      discovered[col] = 1;
    }
  }
  for (Index row = 0; row < nrows; ++row)
    if (!discovered[row]) ++flag;
  free(discovered);
  if (set_size_out) *set_size_out = set_size;
  return flag;
}

/* ---- GC: greedy first-fit over a vertex order, colours from 1 (test_gc.hpp:33-51). ---- */
double oracle_gc(Index nrows, const Index* row_ptr, const Index* col_ind, const Index* order, int max_colors,
                 int32_t* color) {
  unsigned char* used = (unsigned char*)malloc((size_t)max_colors + 1);
  for (Index i = 0; i < nrows; ++i) color[i] = 0;
  double t0 = now_ms();
  for (Index i = 0; i < nrows; ++i) {
    Index row = order[i];
    memset(used, 0, (size_t)max_colors + 1);
    for (Index e = row_ptr[row]; e < row_ptr[row + 1]; ++e) used[color[col_ind[e]]] = 1;
    for (int c = 1; c < max_colors; ++c)
      if (!used[c]) { color[row] = c; break; }
  }
  double t1 = now_ms();
  free(used);
  return t1 - t0;
}

/* SimpleVerifyGc (test_gc.hpp:59-98): errors = stored edges whose endpoints share a colour
 * (uncoloured vertices, colour 0, are reported but -- as there -- not counted unless two of
 * them are adjacent); *uncolored_out counts them so callers can require 0. */
int oracle_gc_verify(Index nrows, const Index* row_ptr, const Index* col_ind, const int32_t* color,
                     int* max_color_out, int* uncolored_out) {
  int num_error = 0, max_color = 0, uncolored = 0;
  for (Index row = 0; row < nrows; ++row) {
    int rc = color[row];
    if (rc > max_color) max_color = rc;
    if (rc == 0) ++uncolored;
    for (Index e = row_ptr[row]; e < row_ptr[row + 1]; ++e)
      if (color[col_ind[e]] == rc) ++num_error;
  }
  if (max_color_out) *max_color_out = max_color;
  if (uncolored_out) *uncolored_out = uncolored;
  return num_error;
}

/* ---- LGC: approximate personalised PageRank by queue pushes (test_lgc.hpp:13-88), T = float,
 * alpha / eps double: every mixed expression is evaluated in double and rounded on the store,
 * as the C++ there does. ---- */
double oracle_lgc(Index nrows, const Index* row_ptr, const Index* col_ind, float* pagerank, Index src,
                  double alpha, double eps, int max_niter) {
  float* residual = (float*)calloc((size_t)nrows + 1, sizeof(float));
  float* residual2 = (float*)calloc((size_t)nrows + 1, sizeof(float));
  float* degrees = (float*)calloc((size_t)nrows + 1, sizeof(float));
  /* the frontier deque: each iteration appends at most nrows vertices after draining */
  Index* frontier = (Index*)malloc(sizeof(Index) * ((size_t)nrows + 1));
  Index* next = (Index*)malloc(sizeof(Index) * ((size_t)nrows + 1));
  Index nf = 0;
  for (Index i = 0; i < nrows; ++i) {
    pagerank[i] = 0.f;
    degrees[i] = (float)(row_ptr[i + 1] - row_ptr[i]);
  }
  residual[src] = 1.f;
  residual2[src] = 1.f;
  frontier[nf++] = src;
  double t0 = now_ms();
  for (int it = 0; it < max_niter; ++it) {
    for (Index k = 0; k < nf; ++k) {
      Index v = frontier[k];
      pagerank[v] = (float)(pagerank[v] + alpha * residual[v]);
      residual2[v] = (float)((1 - alpha) * residual[v] / 2);
    }
    for (Index k = 0; k < nf; ++k) {
      Index v = frontier[k];
      residual[v] = (float)((1 - alpha) * residual[v] / 2);
      for (Index e = row_ptr[v]; e < row_ptr[v + 1]; ++e) {
        Index w = col_ind[e];
        residual2[w] += residual[v] / degrees[v];
      }
    }
    memcpy(residual, residual2, sizeof(float) * (size_t)nrows);
    Index nn = 0;
    for (Index v = 0; v < nrows; ++v)
      if (residual[v] >= degrees[v] * eps) next[nn++] = v;
    memcpy(frontier, next, sizeof(Index) * (size_t)nn);
    nf = nn;
  }
  double t1 = now_ms();
  free(residual); free(residual2); free(degrees); free(frontier); free(next);
  return t1 - t0;
}

/* ---- Direction-optimised BFS accounting oracle -------------------------------
 * Same depth labels as oracle_bfs, but level-synchronous and following the
 * reference's push/pull decisions so that the algorithmic-byte model of
 * SURVEY.md 8(d) / BASELINE.md 3 can be evaluated:
 *   mxvmode 10 = PUSHPULL, 11 = PUSHONLY, 12 = PULLONLY (graphblas/types.hpp:66-68).
 * convert() rule (backend/cuda/vector.hpp:291-323) incl. the two ratio_ slots that
 * Vector::swap exchanges every level (vector.hpp:428-450, algorithm/bfs.hpp:72).
 * Pull inspects in-neighbours (csc) of every unvisited vertex in stored order and
 * stops at the first visited one (kernels/spmv.hpp:33-52, earlyexit).
 * stats row per level: [dir(0 push,1 pull), nf, mf, nu, mi, nf_next].
 * edgeswitch > 0 adds graphblast_amd's optional edge-aware push->pull override.
 * Returns number of levels executed. ---- */
int oracle_bfs_do_stats(Index n, const Index* csr_ptr, const Index* csr_ind,
                        const Index* csc_ptr, const Index* csc_ind, Index src,
                        int mxvmode, float switchpoint, int max_niter,
                        float* depth, long long* stats, int max_levels, float edgeswitch) {
  for (Index i = 0; i < n; ++i) depth[i] = 0.f;
  Index* cur = (Index*)malloc(sizeof(Index) * (size_t)(n > 0 ? n : 1));
  Index* nxt = (Index*)malloc(sizeof(Index) * (size_t)(n > 0 ? n : 1));
  Index nf = 1; cur[0] = src;
  int is_dense = (mxvmode == 12);
  float ratio_u = 0.f, ratio_w = 0.f;  /* ratio_ of f1 and of f2 */
  int level = 0;
  for (int iter = 1; iter <= max_niter; ++iter) {
    /* assign(v, mask=f1, iter) */
    for (Index k = 0; k < nf; ++k) depth[cur[k]] = (float)iter;
    /* vxm: direction decision on u = f1 */
    if (mxvmode == 10) {
      float ratio = (float)nf / (float)n;
      if (!is_dense) {
        if (ratio > switchpoint && ratio > ratio_u) is_dense = 1; else ratio_u = ratio;
      } else {
        if (ratio <= switchpoint && ratio < ratio_u) is_dense = 0; else ratio_u = ratio;
      }
    } else if (mxvmode == 11) is_dense = 0;
    else is_dense = 1;
    /* graphblast_amd extension (not in the reference; off when edgeswitch == 0): leave push
     * when the frontier's out-edges exceed edgeswitch * nnz (csrc/bfs_fused.hip) */
    if (mxvmode == 10 && !is_dense && edgeswitch > 0.f && nf >= 32) {
      long long e = 0;
      for (Index k = 0; k < nf; ++k) e += csr_ptr[cur[k] + 1] - csr_ptr[cur[k]];
      if ((double)e > (double)edgeswitch * (double)csr_ptr[n]) is_dense = 1;
    }
    long long mf = 0, nu = 0, mi = 0;
    Index nn = 0;
    if (!is_dense) {
      for (Index k = 0; k < nf; ++k) {
        Index v = cur[k];
        mf += csr_ptr[v + 1] - csr_ptr[v];
        for (Index e = csr_ptr[v]; e < csr_ptr[v + 1]; ++e) {
          Index nb = csr_ind[e];
          if (depth[nb] == 0.f) { depth[nb] = -1.f; nxt[nn++] = nb; }
        }
      }
      for (Index k = 0; k < nn; ++k) depth[nxt[k]] = 0.f;  /* labelled next iter */
    } else {
      for (Index v = 0; v < n; ++v) {
        if (depth[v] != 0.f) continue;
        nu++;
        for (Index e = csc_ptr[v]; e < csc_ptr[v + 1]; ++e) {
          mi++;
          if (depth[csc_ind[e]] != 0.f) { nxt[nn++] = v; break; }
        }
      }
    }
    if (level < max_levels) {
      long long* row = stats + (size_t)level * 6;
      row[0] = is_dense; row[1] = nf; row[2] = mf; row[3] = nu; row[4] = mi; row[5] = nn;
    }
    level++;
    /* f2.swap(&f1): contents and ratio_ exchange */
    { float t = ratio_u; ratio_u = ratio_w; ratio_w = t; }
    { Index* t = cur; cur = nxt; nxt = t; }
    nf = nn;
    if (nf == 0) break;
  }
  free(cur); free(nxt);
  return level;
}
