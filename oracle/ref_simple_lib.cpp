// ORACLE -- TEST INFRASTRUCTURE ONLY.
// C entry points over the REFERENCE's own CPU oracles and loader, compiled from the
// sources where they lie under /root/reference (never copied into this repo):
//   graphblas/mmio.hpp, graphblas/util.hpp (readMtx / removeSelfloop / customSort /
//   coo2csr / coo2csc), graphblas/algorithm/test_{bfs,sssp,pr,cc,tc,lgc,mis,gc}.hpp.
// util.hpp's only Boost user is parseArgs (util.hpp:39-132) + the include at :16; the
// Makefile writes a filtered copy (_ref/util_noargs.hpp: those lines deleted, nothing
// added, no stand-in header) next to the outputs.  Built only when /root/reference is
// present:  make -C oracle ref  ->  oracle/_ref/libsimple_ref.so  (git-ignored; it
// travels to the GPU box like any built .so and is the checker at full size there).
//
// Two translation units are made from this file: the traversal / numeric oracles at -O3,
// and PART_NORETURN (test_cc.hpp, test_mis.hpp, test_gc.hpp) at -O0 -- every function in
// those three is declared `int` and falls off its end (undefined behaviour that g++ -O2
// turns into a fall-through into the next function); their verdicts are read from what
// they print (CORRECT / INCORRECT), as the reference's mains do.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <typeinfo>
#include <unistd.h>
#include <fcntl.h>

#include "graphblas/types.hpp"         // the reference's own Index (types.hpp:18); -D__GRB_BACKEND_ROOT=cuda names its backend
#include "graphblas/mmio.hpp"
#include "util_noargs.hpp"

#ifndef PART_NORETURN
#include "graphblas/algorithm/test_bfs.hpp"
#include "graphblas/algorithm/test_sssp.hpp"
#include "graphblas/algorithm/test_pr.hpp"
#include "graphblas/algorithm/test_tc.hpp"
#include "graphblas/algorithm/test_lgc.hpp"
#else
#include "graphblas/algorithm/test_cc.hpp"
#include "graphblas/algorithm/test_mis.hpp"
#include "graphblas/algorithm/test_gc.hpp"
#endif

namespace {
// the reference's oracles print timings and arrays; keep the caller's stdout clean
struct Quiet {
  int saved;
  Quiet() {
    fflush(stdout);
    std::cout.flush();
    saved = dup(1);
    int nul = open("/dev/null", O_WRONLY);
    dup2(nul, 1);
    close(nul);
  }
  ~Quiet() {
    fflush(stdout);
    std::cout.flush();
    dup2(saved, 1);
    close(saved);
  }
};
}  // namespace

#ifdef PART_NORETURN
namespace {
// runs f with stdout captured; returns 0 when the text holds a line "CORRECT", else 1
template <typename F>
int verdict(F f) {
  fflush(stdout);
  std::cout.flush();
  char name[] = "/tmp/ref_verdict_XXXXXX";
  int fd = mkstemp(name);
  int saved = dup(1);
  dup2(fd, 1);
  f();
  fflush(stdout);
  std::cout.flush();
  dup2(saved, 1);
  close(saved);
  off_t len = lseek(fd, 0, SEEK_END);
  std::string text(len > 0 ? len : 0, ' ');
  lseek(fd, 0, SEEK_SET);
  if (len > 0 && read(fd, &text[0], len) != len) text.clear();
  close(fd);
  unlink(name);
  if (text.find("INCORRECT") != std::string::npos) return 1;
  return text.find("CORRECT") != std::string::npos ? 0 : 1;
}
}  // namespace
#endif

extern "C" {

#ifndef PART_NORETURN
// test_bfs.hpp:11-61
int ref_bfs(int n, const int* rp, const int* ci, float* depth, int src, int stop) {
  Quiet q;
  return graphblas::algorithm::SimpleReferenceBfs<float>(n, rp, ci, depth, NULL, src, stop);
}

// test_sssp.hpp:15-79
int ref_sssp(int n, const int* rp, const int* ci, float* val, float* dist, int src, int stop) {
  Quiet q;
  return graphblas::algorithm::SimpleReferenceSssp<float>(n, rp, ci, val, dist, src, stop);
}

// test_pr.hpp:15-80
int ref_pr(int n, const int* rp, const int* ci, float* val, float* rank, float alpha, float eps, int max_niter) {
  Quiet q;
  return graphblas::algorithm::SimpleReferencePr<float>(n, rp, ci, val, rank, alpha, eps, max_niter);
}

// test_tc.hpp:41-87
int ref_tc(int n, const int* rp, const int* ci) {
  Quiet q;
  int ntris = 0;
  graphblas::algorithm::SimpleReferenceTc<int>(n, rp, ci, &ntris);
  return ntris;
}

// test_lgc.hpp:14-87 / :89-
void ref_lgc(int n, const int* rp, const int* ci, const float* val, float* pagerank, int src, double alpha,
             double eps, int max_niter, int dense) {
  Quiet q;
  if (dense)
    graphblas::algorithm::SimpleReferenceLgcDense<float>(n, rp, ci, val, pagerank, src, alpha, eps, max_niter);
  else
    graphblas::algorithm::SimpleReferenceLgc<float>(n, rp, ci, val, pagerank, src, alpha, eps, max_niter);
}

// util.hpp:363-430 readMtx<float> -> coordinate lists (sorted, self loops / duplicates
// removed, values as the loader leaves them).  Two-call protocol: *nvals out first with
// row == NULL, then the copy.  Returns 0, or -1 when the file cannot be opened (readMtx
// itself would exit(1)).
static std::vector<int> g_row, g_col;
static std::vector<float> g_val;
static int g_nrows, g_ncols, g_nvals;

int ref_read_mtx(const char* path, int directed, int* nrows, int* ncols, int* nvals) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  fclose(f);
  Quiet q;
  g_row.clear();
  g_col.clear();
  g_val.clear();
  readMtx<float>(path, &g_row, &g_col, &g_val, &g_nrows, &g_ncols, &g_nvals, directed, false);
  *nrows = g_nrows;
  *ncols = g_ncols;
  *nvals = g_nvals;
  return 0;
}
int ref_read_mtx_size(void) { return (int)g_row.size(); }
void ref_read_mtx_copy(int* row, int* col, float* val) {
  memcpy(row, g_row.data(), g_row.size() * sizeof(int));
  memcpy(col, g_col.data(), g_col.size() * sizeof(int));
  memcpy(val, g_val.data(), g_val.size() * sizeof(float));
}

// util.hpp:501-559 coo2csr / :561-572 coo2csc on the lists of the last ref_read_mtx
void ref_coo2csr(int* ptr, int* ind, float* val) {
  Quiet q;
  coo2csr<float>(ptr, ind, val, g_row, g_col, g_val, g_nrows, g_ncols);
}
void ref_coo2csc(int* ptr, int* ind, float* val) {
  Quiet q;
  coo2csc<float>(ptr, ind, val, g_row, g_col, g_val, g_nrows, g_ncols);
}

// util.hpp:340-357: the binary cache's file name for a given .mtx path
void ref_cache_name(const char* path, int is_undirected, char* out, int cap) {
  Quiet q;
  char* s = convert(path, is_undirected != 0);
  snprintf(out, cap, "%s", s);
  free(s);
}

// the weights example/gsssp.cu draws (libstdc++ stream): for reference only, fixtures store them
void ref_sssp_weights(float* val, int nvals, int seed) {
  std::default_random_engine gen(seed);
  std::uniform_int_distribution<int> dist(1, 64);
  for (int i = 0; i < nvals; ++i) val[i] = (float)dist(gen);
}

#else  // PART_NORETURN


// test_cc.hpp:14-56 (labels from 1 in discovery order) / :58-98 SimpleVerifyCc
void ref_cc(int n, const int* rp, const int* ci, int* label) {
  Quiet q;
  std::vector<int> v(n, 0);
  graphblas::algorithm::SimpleReferenceCc(n, rp, ci, &v, 0);
  for (int i = 0; i < n; ++i) label[i] = v[i];
}
int ref_cc_verify(int n, const int* rp, const int* ci, const int* label, int suppress_zero) {
  std::vector<int> v(label, label + n);
  return verdict([&] { graphblas::algorithm::SimpleVerifyCc(n, rp, ci, v, suppress_zero != 0); });
}

// test_mis.hpp:12-62 / :64-
void ref_mis(int n, const int* rp, const int* ci, int* out, int seed) {
  Quiet q;
  std::vector<int> v(n, 0);
  graphblas::algorithm::SimpleReferenceMis(n, rp, ci, &v, seed);
  for (int i = 0; i < n; ++i) out[i] = v[i];
}
int ref_mis_verify(int n, const int* rp, const int* ci, const int* mis) {
  std::vector<int> v(mis, mis + n);
  return verdict([&] { graphblas::algorithm::SimpleVerifyMis(n, rp, ci, v); });
}

// test_gc.hpp:13-57 / :59-
void ref_gc(int n, const int* rp, const int* ci, int* out, int seed, int max_colors) {
  Quiet q;
  std::vector<int> v(n, 0);
  graphblas::algorithm::SimpleReferenceGc(n, rp, ci, &v, seed, max_colors);
  for (int i = 0; i < n; ++i) out[i] = v[i];
}
int ref_gc_verify(int n, const int* rp, const int* ci, const int* colour, int suppress_zero) {
  std::vector<int> v(colour, colour + n);
  return verdict([&] { graphblas::algorithm::SimpleVerifyGc(n, rp, ci, v, suppress_zero != 0); });
}

#endif

}  // extern "C"
