// algorithms.hip -- the remaining drivers of graphblas/algorithm/ on the path, written
// against the C ABI's own operations exactly as the reference writes them against the
// frontend:  sssp (algorithm/sssp.hpp:15-103)  and  pr (algorithm/pr.hpp:15-94).
// (bfs lives in ops.hip / bfs_fused.hip.)
#include <cmath>

#include "persist_common.hpp"
#include <chrono>

using namespace grb;

namespace {
struct VecGuard {               // frees temporaries on every exit path
  std::vector<grb_vector> v;
  ~VecGuard() { for (grb_vector x : v) grb_vector_free(x); }
  grb_info make(grb_vector* out, grb_dtype dt, Index n) {
    grb_info i = grb_vector_new(out, dt, n);
    if (i == GRB_SUCCESS) v.push_back(*out);
    return i;
  }
};
}  // namespace

static inline double host_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

grb_info sssp_persistent_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* iterations,
                             double* succ, float* tight_ms, grb_vector f1_dense, bool* handed_over);

extern "C" {

// Bellman-Ford with frontier filtering; MinimumPlus vxm + CustomLessPlus / MinimumPlus
// eWiseAdd + masked assign + reduce, as algorithm/sssp.hpp:62-91.
grb_info grb_sssp(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, grb_algo_result* result) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (source < 0 || source >= A->nrows) return GRB_INVALID_INDEX;
  const Index n = A->nrows;
  static const bool fused_ok = [] { const char* e = getenv("GRB_SSSP_FUSED"); return !e || atoi(e) != 0; }();
  const double fmax = (double)FLT_MAX;
  VecGuard g;
  grb_vector f1, f2, m;
  GRB_TRY(g.make(&f1, GRB_F32, n));
  int first_iter = 1;
  float fused_ms = 0.f;
  bool continued = false;
  if (fused_ok && A->built && A->format == 0 && v->nsize == n) {   // CSR-only format: the op-by-op rounds (forced push)
    // the same synchronous rounds in one launch (sssp_persist.hip); not eligible -> op by op.  A
    // dense frontier is handed back: from there the pull product of the op-by-op rounds is faster.
    GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
    GRB_TRY(grb_vector_set_storage(f1, GRB_DENSE));
    int it = 0;
    double sc = 0;
    static int persistent_failures = 0;       // as in grb_bfs_fused: three give-ups in a row end the attempts
    const grb_info fi = persistent_failures < 3
                            ? sssp_persistent_run(v, A, source, desc, &it, &sc, &fused_ms, f1, &continued)
                            : GRB_NOT_IMPLEMENTED;
    if (fi == GRB_PANIC) ++persistent_failures; else if (fi == GRB_SUCCESS) persistent_failures = 0;
    if (fi == GRB_SUCCESS && !continued) {
      desc->lastmxv = GRB_PUSHONLY;
      if (result) { result->iterations = it; result->tight_ms = fused_ms; result->last_value = sc; }
      return GRB_SUCCESS;
    }
    if (fi == GRB_PANIC) {            // the persistent launch gave up (grid not co-resident): the call sequence instead
      static bool told = false;
      if (!told) fprintf(stderr, "libgrb_hip: one-launch SSSP unavailable, using the op-by-op rounds\n");
      told = true;
      GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
      continued = false;
      fused_ms = 0.f;
    } else if (fi != GRB_SUCCESS && fi != GRB_NOT_IMPLEMENTED) {
      return fi;
    }
    if (continued) first_iter = it + 1;
  }
  GRB_TRY(g.make(&f2, GRB_F32, n));
  GRB_TRY(g.make(&m, GRB_F32, n));
  if (!continued) {
    GRB_TRY(grb_vector_fill(v, fmax));
    GRB_TRY(grb_vector_set_element(v, 0.0, source));
    if (desc->desc[GRB_MXVMODE] == GRB_PULLONLY) {
      GRB_TRY(grb_vector_fill(f1, fmax));
      GRB_TRY(grb_vector_set_element(f1, 0.0, source));
    } else {
      float zero = 0.f;
      GRB_TRY(grb_vector_build_sparse(f1, &source, &zero, 1));
    }
  }
  int iter = first_iter;
  grb_index f1_nvals = 1;
  double succ = 1;
  float ms = 0.f;
  const bool log = desc->timing != 0;
  if (!continued) desc->iter_log.clear();
  GRB_TRY(grb_timer_start());
  for (; iter <= desc->max_niter; ++iter) {
    GRB_TRY(grb_vxm(f2, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_PLUS, f1, A, desc));
    GRB_TRY(grb_eWiseAdd(m, nullptr, GRB_ACCUM_NULL, GRB_CUSTOM_LESS_PLUS, f2, v, desc));
    GRB_TRY(grb_eWiseAdd(v, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_PLUS, v, f2, desc));
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_info ai = grb_assign(f2, m, GRB_ACCUM_NULL, fmax, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    GRB_TRY(ai);
    GRB_TRY(grb_vector_swap(f2, f1));
    GRB_TRY(grb_vector_nvals(f1, &f1_nvals));
    GRB_TRY(grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, m, desc));
    if (log) {                      // the reference stops and restarts its GpuTimer every iteration too (sssp.hpp:55-64)
      float it_ms = 0.f;
      GRB_TRY(grb_timer_stop(&it_ms));
      ms += it_ms;
      desc->iter_log.push_back(grb_algo_iter{iter, desc->lastmxv, (double)f1_nvals, it_ms, 0});
      GRB_TRY(grb_timer_start());
    }
    if (f1_nvals == 0 || succ == 0) break;
  }
  { float last_ms = 0.f; GRB_TRY(grb_timer_stop(&last_ms)); ms += last_ms; }
  // a run that started in the fused loop keeps its meaning of last_value, the number of vertices the
  // last round improved (f1.nvals of a sparse f1, reduce(m) of a dense one); a pure op-by-op run
  // reports the reference's reduce(m) whatever the storage
  if (result) { result->iterations = iter; result->tight_ms = ms + fused_ms; result->last_value = (continued && f1->vec_type == GRB_SPARSE) ? (double)f1_nvals : succ; }
  return GRB_SUCCESS;
}

// PageRank power iteration, algorithm/pr.hpp:60-82: vxm(PlusMultiplies) + eWiseAdd scalar +
// eWiseMult(PlusMinus) + eWiseAdd(MultipliesMultiplies) + reduce(Plus); error = sqrt(sum).
// Op by op, exactly the reference's call sequence (kept for GrB_PUSHONLY, whose behaviour in
// the reference is whatever its sparse paths do, and as the fused loop's cross-check).
static grb_info pr_op_by_op(grb_vector p, grb_matrix A, float alpha, float eps, grb_descriptor desc,
                            grb_algo_result* result) {
  const Index n = A->nrows;
  GRB_TRY(grb_vector_clear(p));
  GRB_TRY(grb_vector_fill(p, (double)(1.f / n)));
  VecGuard g;
  grb_vector p_prev, p_swap, r, r_temp;
  GRB_TRY(g.make(&p_prev, GRB_F32, n));
  GRB_TRY(g.make(&p_swap, GRB_F32, n));
  GRB_TRY(g.make(&r, GRB_F32, n));
  GRB_TRY(g.make(&r_temp, GRB_F32, n));
  GRB_TRY(grb_vector_fill(r, 1.0));
  int iter = 1;
  float error = 1.f;
  float ms = 0.f;
  desc->iter_log.clear();
  GRB_TRY(grb_timer_start());
  double t_iter = host_ms();
  for (; error > eps && iter <= desc->max_niter; ++iter) {
    GRB_TRY(grb_vector_dup(p_prev, p));
    GRB_TRY(grb_vxm(p_swap, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, p_prev, A, desc));
    GRB_TRY(grb_eWiseAdd_scalar(p, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, p_swap,
                                (double)((1.f - alpha) / n), desc));
    GRB_TRY(grb_eWiseMult(r, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MINUS, p, p_prev, desc));
    GRB_TRY(grb_eWiseAdd(r_temp, nullptr, GRB_ACCUM_NULL, GRB_MULTIPLIES_MULTIPLIES, r, r, desc));
    double sum = 0;
    GRB_TRY(grb_reduce_vector(&sum, GRB_ACCUM_NULL, GRB_PLUS_MONOID, r_temp, desc));
    error = sqrtf((float)sum);
    if (desc->timing != 0) {
      const double now = host_ms();
      desc->iter_log.push_back(grb_algo_iter{iter, desc->lastmxv, (double)error, (float)(now - t_iter), 0});
      t_iter = now;
    }
  }
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = iter - 1; result->tight_ms = ms; result->last_value = error; }
  return GRB_SUCCESS;
}

// The same iteration with the five element-wise calls and the reduction folded into one pass:
//   p_swap = p (x) A            the SpMV kernel (pull; p is dense and strictly positive)
//   p_new  = p_swap + (1-alpha)/n ;  r = p_new - p ;  sum += r*r ;  p = p_new     one kernel
// The squared residual comes back through the pinned mailbox (no stream synchronise, no
// separate reduce launches, no p_prev copy).  Per element the arithmetic is the reference's.
namespace grb {
__global__ __launch_bounds__(kBlock) void pr_update_kernel(const float* __restrict__ swp, float* __restrict__ p, float c,
                                                           Index n, float* partial, unsigned int* ticket,
                                                           unsigned long long* mail, int seq) {
  __shared__ float s_sum[kWavesPerBlock];
  __shared__ int s_last;
  float acc = 0.f;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float pn = swp[i] + c;
    const float r = pn - p[i];
    p[i] = pn;
    acc += r * r;
  }
  acc = wave_reduce(acc, [](float a, float b) { return a + b; });
  if (lane_id() == 0) s_sum[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) t += s_sum[w];
    __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_workgroup_arrives(ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  float t = 0.f;
  for (int j = threadIdx.x; j < (int)gridDim.x; j += kBlock)
    t += __hip_atomic_load(&partial[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = wave_reduce(t, [](float a, float b) { return a + b; });
  __syncthreads();
  if (lane_id() == 0) s_sum[wave_id()] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) tot += s_sum[w];
    __hip_atomic_store(&mail[0], ((unsigned long long)(unsigned int)seq << 32) | __float_as_uint(tot),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace grb

grb_info grb_pr(grb_vector p, grb_matrix A, float alpha, float eps, grb_descriptor desc, grb_algo_result* result) { GRB_API_ENTER();
  if (!p || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  const Index n = A->nrows;
  static const bool fused_ok = [] { const char* e = getenv("GRB_PR_FUSED"); return !e || atoi(e) != 0; }();
  if (!fused_ok || A->format != 0 || desc->desc[GRB_MXVMODE] == GRB_PUSHONLY || !(alpha < 1.f) || p->dtype != GRB_F32 ||
      A->dtype != GRB_F32 || A->nrows != A->ncols || p->nsize != n || !A->csc.ptr)
    return pr_op_by_op(p, A, alpha, eps, desc, result);
  Context& c = ctx();
  GRB_TRY(grb_vector_clear(p));
  GRB_TRY(grb_vector_fill(p, (double)(1.f / n)));
  VecGuard g;
  grb_vector p_swap;
  GRB_TRY(g.make(&p_swap, GRB_F32, n));
  GRB_TRY(grb_vector_set_storage(p_swap, GRB_DENSE));
  const int grid = stream_grid(n, kBlock);
  void* p_part;
  GRB_TRY(scratch(1, sizeof(float) * (size_t)grid + 16, &p_part));   // not 4 / 5: those hold the push path's state
  unsigned int* d_ticket = c.d_tickets;
  const float cst = (1.f - alpha) / n;
  int iter = 1;
  float error = 1.f;
  float ms = 0.f;
  desc->iter_log.clear();
  GRB_TRY(grb_timer_start());
  double t_iter = host_ms();
  for (; error > eps && iter <= desc->max_niter; ++iter) {
    // vxm treats A as transposed: the pull product walks the CSC orientation (operations.hpp:80-209)
    GRB_TRY(k_spmv(GRB_PLUS_MULTIPLIES, GRB_F32, A->csc, A->plan_csc, p->d_val, nullptr, 0, 0, 0, p_swap->d_val,
                   A->csc_alias ? nullptr : A->csr.ptr));
    const int seq = ++c.mail_seq;
    hipLaunchKernelGGL(pr_update_kernel, dim3(grid), dim3(kBlock), 0, c.stream, (const float*)p_swap->d_val,
                       (float*)p->d_val, cst, n, (float*)p_part, d_ticket, c.d_hgran, seq);
    GRB_HIP_TRY(hipGetLastError());
    unsigned int bits = 0;
    GRB_TRY(wait_granules(seq, 1, &bits));
    float sum;
    memcpy(&sum, &bits, 4);
    error = sqrtf(sum);
    if (desc->timing != 0) {        // the host has just seen this iteration's residual: its wall-clock share
      const double now = host_ms();
      desc->iter_log.push_back(grb_algo_iter{iter, GRB_PULLONLY, (double)error, (float)(now - t_iter), 0});
      t_iter = now;
    }
  }
  desc->lastmxv = GRB_PULLONLY;
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = iter - 1; result->tight_ms = ms; result->last_value = error; }
  return GRB_SUCCESS;
}

// ---- FastSV's element-wise tail in ONE launch ----------------------------------------------------------------
// After the MinimumSelectSecond product an iteration of algorithm::cc is eight more calls (cc.hpp:83-112: three
// eWiseAdd, assignScatter, extractGather, eWiseMult, reduce, masked assign, plus two dup): nine launches and
// 4 n x 13 bytes of traffic.  One co-resident launch does them in two passes around a grid barrier:
//
//   pass A (per u)   m = min(mnp[u], mnp_temp[u]);  mnp[u] = m                    eWiseAdd (:83-85)
//                    atomicMin(parent[parent_temp[u]], m)                         assignScatter (:87-88)
//                    atomicMin(parent[u], m)                                      eWiseAdd (:92-93); the third one,
//                                                                                 min with parent_temp (:97-98), is
//                                                                                 implied: parent starts as parent_temp
//   -- grid barrier --
//   pass B (per u)   gf = parent[parent[u]]                                       extractGather (:102-103)
//                    d = grandparent_temp[u] != gf;  succ += d                     eWiseMult + reduce (:106-109)
//                    grandparent_temp[u] = gf;  grandparent[u] = d ? gf : INT_MAX  dup + masked assign (:113-119)
//                    parent_temp[u] = parent[u]                                    the next iteration's dup (:77)
//
// The reference's scatter lets racing writers of one parent[i] overwrite each other (kernels/scatter.hpp:24-39: any
// candidate may win); atomicMin picks the smallest candidate -- one of the outcomes the reference admits, and since
// the two eWiseAdds that follow only take minima with values that are all in the atomicMin already, parent after
// pass A is exactly what the call sequence yields for that winner.  Converged labels do not depend on the winner.
namespace grb {
__global__ __launch_bounds__(kPThreads) void cc_tail_kernel(int* __restrict__ mnp, const int* __restrict__ mnp_temp,
                                                            int* parent, int* __restrict__ parent_temp,
                                                            int* __restrict__ grandparent, int* __restrict__ grandparent_temp,
                                                            Index n, GridBarrier* bar, unsigned gen0,
                                                            unsigned int* partial, unsigned int* ticket,
                                                            unsigned long long* mail, int seq) {
  __shared__ unsigned int s_cnt[kPWaves];
  __shared__ int s_last;
  const long long gtid = (long long)blockIdx.x * kPThreads + threadIdx.x;
  const long long gthreads = (long long)gridDim.x * kPThreads;
  for (long long u = gtid; u < n; u += gthreads) {
    const int a = mnp[u], b = mnp_temp[u];
    const int m = a < b ? a : b;
    mnp[u] = m;
    // parent only ever decreases during this pass, so a value read now bounds it from above: an atomic that
    // could not lower it is skipped (most vertices of a large component point at one root -- millions of
    // atomics on one address serialise at ~12 ns each)
    const int pt = parent_temp[u];
    if (pt >= 0 && pt < n && m < __hip_atomic_load(&parent[pt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&parent[pt], m);
    if (m < pt) atomicMin(&parent[u], m);
  }
  unsigned gen = gen0;
  if (!grid_sync(bar, gen, true)) {
    if (gtid == 0)
      __hip_atomic_store(&mail[0], ((unsigned long long)(unsigned int)seq << 32) | 0xffffffffull, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  unsigned int cnt = 0;
  for (long long u = gtid; u < n; u += gthreads) {
    const int f = parent[u];
    const int gf = (f >= 0 && f < n) ? parent[f] : 0;
    const int gt = grandparent_temp[u];
    const bool d = gt != gf;
    cnt += d ? 1u : 0u;
    grandparent_temp[u] = gf;
    grandparent[u] = d ? gf : INT_MAX;
    parent_temp[u] = f;
  }
  cnt = wave_reduce(cnt, [](unsigned int x, unsigned int y) { return x + y; });
  if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int w = 0; w < kPWaves; ++w) t += s_cnt[w];
    __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_workgroup_arrives(ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  unsigned int t = 0;
  for (int j = threadIdx.x; j < (int)gridDim.x; j += kPThreads)
    t += __hip_atomic_load(&partial[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = wave_reduce(t, [](unsigned int x, unsigned int y) { return x + y; });
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int tot = 0;
    for (int w = 0; w < kPWaves; ++w) tot += s_cnt[w];
    if (tot == 0xffffffffu) tot = 0xfffffffeu;            // 0xffffffff is the barrier's distress value
    __hip_atomic_store(&mail[0], ((unsigned long long)(unsigned int)seq << 32) | tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace grb

// the call sequence of algorithm/cc.hpp:77-119 for one iteration, op by op (fallback and yardstick: GRB_CC_FUSED=0)
static grb_info cc_tail_op_by_op(grb_vector diff, grb_vector parent, grb_vector parent_temp, grb_vector grandparent,
                                 grb_vector grandparent_temp, grb_vector mnp, grb_vector mnp_temp, grb_descriptor desc,
                                 double* succ) {
  GRB_TRY(grb_eWiseAdd(mnp, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_SELECT_SECOND, mnp, mnp_temp, desc));
  GRB_TRY(grb_assignScatter(parent, nullptr, GRB_ACCUM_NULL, mnp, parent_temp, desc));
  GRB_TRY(grb_eWiseAdd(parent, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_PLUS, parent, mnp, desc));
  GRB_TRY(grb_eWiseAdd(parent, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_PLUS, parent, parent_temp, desc));
  GRB_TRY(grb_extractGather(grandparent, nullptr, GRB_ACCUM_NULL, parent, parent, desc));
  GRB_TRY(grb_eWiseMult(diff, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_NOT_EQUAL_TO, grandparent_temp, grandparent, desc));
  GRB_TRY(grb_reduce_vector(succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, diff, desc));
  return GRB_SUCCESS;
}

// 1 (default; GRB_CC_FUSED=0 in the environment starts with 0): the element-wise tail of an iteration in one
// launch; 0: the reference's call sequence op by op.  on < 0 only queries.  Returns the value in force.
int grb_cc_set_fused(int on) { GRB_API_ENTER_NOINFO();
  static int v = [] { const char* e = getenv("GRB_CC_FUSED"); return (!e || atoi(e) != 0) ? 1 : 0; }();
  if (on >= 0) v = on ? 1 : 0;
  return v;
}

// FastSV connected components, algorithm/cc.hpp:17-136: v = parent vector (component label =
// smallest vertex id of the component once converged). A is an int matrix (pattern values).
static grb_info cc_run(grb_vector v, grb_matrix A, grb_descriptor desc, grb_algo_result* result, bool want_fused,
                       bool* barrier_gave_up);

// The fused tail's grid barrier needs its workgroups co-resident; on a shared device it can give up.  The call then
// starts over op by op (every vector is re-initialised by cc_run), and after two such failures in a process later
// calls do not try the fused tail again -- the same rule as the one-launch BFS / SSSP (persistent_failures).
static int g_cc_barrier_failures = 0;

grb_info grb_cc(grb_vector v, grb_matrix A, int seed, grb_descriptor desc, grb_algo_result* result) { GRB_API_ENTER();
  (void)seed;
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_I32 || A->dtype != GRB_I32) return GRB_DOMAIN_MISMATCH;
  const bool want_fused = grb_cc_set_fused(-1) != 0 && g_cc_barrier_failures < 2;
  bool gave_up = false;
  const grb_info info = cc_run(v, A, desc, result, want_fused, &gave_up);
  if (!gave_up) return info;
  ++g_cc_barrier_failures;
  return cc_run(v, A, desc, result, false, &gave_up);
}

static grb_info cc_run(grb_vector v, grb_matrix A, grb_descriptor desc, grb_algo_result* result, bool want_fused,
                       bool* barrier_gave_up) {
  *barrier_gave_up = false;
  const Index n = A->nrows;
  VecGuard g;
  grb_vector diff, parent, parent_temp, grandparent, grandparent_temp, mnp, mnp_temp;
  for (grb_vector* p : {&diff, &parent, &parent_temp, &grandparent, &grandparent_temp, &mnp, &mnp_temp})
    GRB_TRY(g.make(p, GRB_I32, n));
  GRB_TRY(grb_vector_fill_ascending(parent, n));
  GRB_TRY(grb_vector_dup(mnp, parent));
  GRB_TRY(grb_vector_dup(mnp_temp, parent));
  GRB_TRY(grb_vector_dup(grandparent, parent));
  GRB_TRY(grb_vector_dup(grandparent_temp, parent));
  const bool fused = want_fused && n > 0 && A->nrows == A->ncols;
  Context& c = ctx();
  GridBarrier* d_bar = nullptr;
  unsigned int* d_partial = nullptr;
  int launches = 0;
  if (fused) {
    // the barrier's counters live across the launches of this call and the products in between use the scratch
    // slots: a small block of its own, allocated once per process
    static void* p_bar = nullptr;
    if (!p_bar) GRB_HIP_TRY(hipMalloc(&p_bar, sizeof(GridBarrier) + 4 * 1024 + 64));
    if (c.num_cu > 1024) return GRB_PANIC;
    GRB_HIP_TRY(hipMemsetAsync(p_bar, 0, sizeof(GridBarrier) + 4 * 1024 + 64, c.stream));
    d_bar = (GridBarrier*)p_bar;
    d_partial = (unsigned int*)((char*)p_bar + sizeof(GridBarrier));
    GRB_TRY(grb_vector_dup(parent_temp, parent));          // kept equal to parent by the fused tail from here on
  }
  int iter = 1;
  double succ = 0;
  float ms = 0.f;
  desc->iter_log.clear();
  GRB_TRY(grb_timer_start());
  double t_iter = host_ms();
  for (; iter <= desc->max_niter; ++iter) {
    if (!fused) GRB_TRY(grb_vector_dup(parent_temp, parent));
    GRB_TRY(grb_mxv(mnp_temp, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_SELECT_SECOND, A, grandparent, desc));
    if (fused) {
      // the product may have left its result sparse (push): the tail reads it dense; entries it does not hold
      // are the monoid's identity, under which min() changes nothing
      if (mnp_temp->vec_type == GRB_SPARSE) GRB_TRY(grb_vector_sparse2dense(mnp_temp, (double)INT_MAX, desc));
      const int seq = ++c.mail_seq;
      hipLaunchKernelGGL(cc_tail_kernel, dim3(c.num_cu), dim3(kPThreads), 0, c.stream, (int*)mnp->d_val,
                         (const int*)mnp_temp->d_val, (int*)parent->d_val, (int*)parent_temp->d_val, (int*)grandparent->d_val,
                         (int*)grandparent_temp->d_val, n, d_bar, (unsigned)launches, d_partial, c.d_tickets, c.d_hgran, seq);
      GRB_HIP_TRY(hipGetLastError());
      ++launches;
      unsigned int bits = 0;
      GRB_TRY(wait_granules(seq, 1, &bits));
      if (bits == 0xffffffffu) {                           // the grid barrier gave up: grb_cc starts over op by op
        *barrier_gave_up = true;
        return GRB_PANIC;
      }
      succ = (double)bits;
      for (grb_vector x : {mnp, parent, parent_temp, grandparent, grandparent_temp}) GRB_TRY(grb_vector_set_storage(x, GRB_DENSE));
    } else {
      GRB_TRY(cc_tail_op_by_op(diff, parent, parent_temp, grandparent, grandparent_temp, mnp, mnp_temp, desc, &succ));
    }
    if (desc->timing != 0) {
      const double now = host_ms();
      desc->iter_log.push_back(grb_algo_iter{iter, desc->lastmxv, succ, (float)(now - t_iter), 0});
      t_iter = now;
    }
    if (succ == 0) break;
    if (fused) continue;                                   // the dup and the masked assign happened in the tail
    GRB_TRY(grb_vector_dup(grandparent_temp, grandparent));
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_info ai = grb_assign(grandparent, diff, GRB_ACCUM_NULL, (double)INT_MAX, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    GRB_TRY(ai);
  }
  GRB_TRY(grb_vector_dup(v, parent));
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = iter; result->tight_ms = ms; result->last_value = succ; }
  return GRB_SUCCESS;
}

}  // extern "C"
