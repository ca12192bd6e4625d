/* The C ABI from plain C: build a graph from a coordinate list, run the direction-optimised BFS as one
 * launch (grb_bfs_fused) and the reference's op-by-op loop (grb_bfs), print the depth labels; then six traversals queued and
 * run four side by side per launch.
 *   gcc -std=c99 -Iinclude examples/bfs_c_abi.c -Lgraphblast_amd -lgrb_hip -Wl,-rpath,$PWD/graphblast_amd -o bfs_c_abi
 *   ./bfs_c_abi            # a 3 x 4 grid, source 0
 */
#include <stdio.h>
#include <stdlib.h>

#include "grb_hip.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    grb_info i_ = (call);                                                  \
    if (i_ != GRB_SUCCESS) {                                               \
      fprintf(stderr, "%s:%d: %s -> Info %d\n", __FILE__, __LINE__, #call, i_); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(void) {
  enum { W = 4, H = 3, N = W * H };
  grb_index rows[2 * (2 * W * H)], cols[2 * (2 * W * H)];
  float vals[2 * (2 * W * H)];
  grb_index m = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int v = y * W + x;
      if (x + 1 < W) { rows[m] = v; cols[m] = v + 1; vals[m++] = 1.f; rows[m] = v + 1; cols[m] = v; vals[m++] = 1.f; }
      if (y + 1 < H) { rows[m] = v; cols[m] = v + W; vals[m++] = 1.f; rows[m] = v + W; cols[m] = v; vals[m++] = 1.f; }
    }
  char dev[128];
  CHECK(grb_device_info(dev, sizeof dev));
  printf("device: %s  library: %s\n", dev, grb_version());

  grb_matrix A;
  grb_vector v;
  grb_descriptor desc;
  CHECK(grb_matrix_new(&A, GRB_F32, N, N));
  CHECK(grb_matrix_build(A, rows, cols, vals, m));
  CHECK(grb_vector_new(&v, GRB_F32, N));
  CHECK(grb_descriptor_new(&desc));
  CHECK(grb_descriptor_load_defaults(desc));
  CHECK(grb_descriptor_set_arg(desc, "mxvmode", 0));      /* push-pull, as run_bfs.sh */
  CHECK(grb_descriptor_set_arg(desc, "struconly", 1));
  CHECK(grb_descriptor_set_arg(desc, "opreuse", 1));

  float fused[N], opbyop[N];
  grb_index n = N;
  grb_bfs_result res;
  CHECK(grb_bfs_fused(v, A, 0, desc, &res, NULL, 0, 0));
  CHECK(grb_vector_extract_tuples_dense(v, fused, &n));
  printf("one launch : %d levels, %d reached, %lld edges\n", res.levels, (int)res.reached, (long long)res.edges_traversed);
  CHECK(grb_bfs(v, A, 0, desc, &res));
  n = N;
  CHECK(grb_vector_extract_tuples_dense(v, opbyop, &n));
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) printf(" %g", fused[y * W + x]);
    printf("\n");
  }
  for (int i = 0; i < N; ++i)
    if (fused[i] != opbyop[i] || fused[i] != (float)(1 + i % W + i / W)) {
      printf("MISMATCH at %d\n", i);
      return 2;
    }
  /* many traversals, one wait: queued (grb_bfs_fused_enqueue), four side by side per launch (grb_bfs_set_coschedule),
   * waited for afterwards -- every source's labels are the grid distances from it */
  enum { K = 6 };
  grb_vector vk[K];
  grb_bfs_ticket tk[K];
  grb_bfs_set_coschedule(4);
  for (int s = 0; s < K; ++s) {
    CHECK(grb_vector_new(&vk[s], GRB_F32, N));
    CHECK(grb_bfs_fused_enqueue(vk[s], A, 2 * s, desc, &tk[s]));
  }
  for (int s = 0; s < K; ++s) {
    float got[N];
    CHECK(grb_bfs_wait(tk[s], &res));
    n = N;
    CHECK(grb_vector_extract_tuples_dense(vk[s], got, &n));
    const int sx = (2 * s) % W, sy = (2 * s) / W;
    for (int i = 0; i < N; ++i)
      if (got[i] != (float)(1 + abs(i % W - sx) + abs(i / W - sy))) {
        printf("MISMATCH at %d of queued traversal %d\n", i, s);
        return 3;
      }
    grb_vector_free(vk[s]);
  }
  grb_bfs_set_coschedule(1);
  printf("queued, four per launch: %d traversals\n", K);
  printf("CORRECT\n");
  grb_descriptor_free(desc);
  grb_vector_free(v);
  grb_matrix_free(A);
  return 0;
}
