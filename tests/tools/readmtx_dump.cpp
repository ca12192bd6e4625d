// Test driver: the drop-in frontend's readMtx (include/graphblas/graphblas.hpp) on one file;
// prints nrows ncols nvals and the coordinate lists so tests/test_frontend_host.py can compare
// them with the reference's own readMtx output (tests/golden/algo_ref.npz).  Host code only.
#define GRB_USE_CUDA
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "graphblas/graphblas.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::vector<graphblas::Index> r, c;
  std::vector<float> v;
  graphblas::Index nr, nc, nv;
  char* dat = NULL;
  const bool want_name = argc > 3;
  readMtx(argv[1], &r, &c, &v, &nr, &nc, &nv, atoi(argv[2]), false, want_name ? &dat : NULL);
  printf("%d %d %d %d\n", nr, nc, nv, (int)r.size());
  if (want_name) printf("%s\n", dat ? dat : "");
  for (size_t i = 0; i < r.size(); ++i) printf("%d %d %.9g\n", r[i], c[i], v[i]);
  return 0;
}
