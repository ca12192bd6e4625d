/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * NOT the reference: the reference's CPU oracles are strictly sequential (OpenMP is commented out in
 * its CMakeLists.txt:8,34).  This is the same BFS labelling (SimpleReferenceBfs, test_bfs.hpp:18-52:
 * source = 1, unreachable = 0) computed level-synchronously on all host cores, so that bench.py can
 * put "every core of the host" beside "one core running the reference's loop" (SURVEY.md 8(d)).
 * Labels are identical to oracle_bfs (BFS depth is unique); tests/test_oracle.py checks that.
 */
#include <omp.h>
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

/* Top-down while the frontier is small, bottom-up (every unvisited vertex looks for a parent in the
 * frontier) once it holds more than 1/32 of the edges -- the usual direction switch; a level array of
 * int32, compare-and-swap on discovery. */
double oracle_bfs_all_cores(Index n, const Index* row_ptr, const Index* col_ind, float* depth, Index src,
                            int nthreads, int* threads_used) {
  int32_t* level = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  Index* cur = (Index*)malloc(sizeof(Index) * ((size_t)n + 1));
  Index* nxt = (Index*)malloc(sizeof(Index) * ((size_t)n + 1));
  if (nthreads > 0) omp_set_num_threads(nthreads);
  int used = 1;
  level[src] = 1;
  cur[0] = src;
  Index ncur = 1;
  const long long nnz = row_ptr[n];
  double t0 = now_ms();
  for (int32_t lv = 1; ncur > 0; ++lv) {
    long long fedges = 0;
#pragma omp parallel for reduction(+ : fedges) schedule(static)
    for (Index i = 0; i < ncur; ++i) fedges += row_ptr[cur[i] + 1] - row_ptr[cur[i]];
    Index nnext = 0;
    if (fedges * 32 > nnz) {
      /* bottom-up */
#pragma omp parallel
      {
#pragma omp single
        used = omp_get_num_threads();
        Index* mine = (Index*)malloc(sizeof(Index) * 4096);
        Index cnt = 0;
#pragma omp for schedule(dynamic, 4096) nowait
        for (Index v = 0; v < n; ++v) {
          if (level[v]) continue;
          for (Index e = row_ptr[v]; e < row_ptr[v + 1]; ++e) {
            if (level[col_ind[e]] == lv) {      /* symmetric graphs: in-neighbours == out-neighbours */
              level[v] = -(lv + 1);             /* marked, made positive after the level */
              mine[cnt++] = v;
              if (cnt == 4096) {
                Index pos = __sync_fetch_and_add(&nnext, cnt);
                memcpy(nxt + pos, mine, sizeof(Index) * (size_t)cnt);
                cnt = 0;
              }
              break;
            }
          }
        }
        if (cnt) {
          Index pos = __sync_fetch_and_add(&nnext, cnt);
          memcpy(nxt + pos, mine, sizeof(Index) * (size_t)cnt);
        }
        free(mine);
      }
#pragma omp parallel for schedule(static)
      for (Index i = 0; i < nnext; ++i) level[nxt[i]] = lv + 1;
    } else {
#pragma omp parallel
      {
#pragma omp single
        used = omp_get_num_threads();
        Index* mine = (Index*)malloc(sizeof(Index) * 4096);
        Index cnt = 0;
#pragma omp for schedule(dynamic, 64) nowait
        for (Index i = 0; i < ncur; ++i) {
          const Index u = cur[i];
          for (Index e = row_ptr[u]; e < row_ptr[u + 1]; ++e) {
            const Index v = col_ind[e];
            if (level[v] == 0 && __sync_bool_compare_and_swap(&level[v], 0, lv + 1)) {
              mine[cnt++] = v;
              if (cnt == 4096) {
                Index pos = __sync_fetch_and_add(&nnext, cnt);
                memcpy(nxt + pos, mine, sizeof(Index) * (size_t)cnt);
                cnt = 0;
              }
            }
          }
        }
        if (cnt) {
          Index pos = __sync_fetch_and_add(&nnext, cnt);
          memcpy(nxt + pos, mine, sizeof(Index) * (size_t)cnt);
        }
        free(mine);
      }
    }
    Index* t = cur; cur = nxt; nxt = t;
    ncur = nnext;
  }
  double t1 = now_ms();
#pragma omp parallel for schedule(static)
  for (Index v = 0; v < n; ++v) depth[v] = (float)level[v];
  free(level); free(cur); free(nxt);
  if (threads_used) *threads_used = used;
  return t1 - t0;
}
