// comm.hip -- the library's own RCCL communicator: one process per GPU, the
// collectives of the 1-D partitioned drivers enqueued from C++ on a SECOND HIP stream and fenced
// against the compute stream with events, so a collective runs while the next local kernel does:
//
//   compute stream   kernel(chunk 0) -- ev --> kernel(chunk 1) ------- wait(done) --> next step
//   comm stream              wait(ev) --> all-gather(chunk 0) ... all-gather(chunk 1) -- done
//
// No host synchronisation inside a step.  The reference has nothing to mirror here (its --ndevice
// flag is unused, backend/cuda/descriptor.hpp:242,283-284); this is the contract of SURVEY.md 8(e).
//
// RCCL is bound at run time (dlopen: the copy already in the process -- PyTorch ships one -- else
// ROCm's), so libgrb_hip.so has no link-time dependency on it and loads on a box without RCCL.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  Rccl api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;       // the communication stream
  hipEvent_t ev_ready = nullptr;      // compute -> comm: the data to send is complete
  hipEvent_t ev_done = nullptr;       // comm -> compute: the collective has completed
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // timing of the last collective
  bool timed = false;
  double total_us = 0;                // sum over the collectives since the last reset
  long long calls = 0;
  bool pending = false;               // the last collective's time has not been added yet
  // host-staged transport (grb_comm_set_host_transport): the collectives as callbacks over pinned host buffers
  grb_comm_host_fn host_fn = nullptr;
  void* host_user = nullptr;
  char* h_stage = nullptr;            // pinned: [send | recv]
  size_t h_cap = 0;
  bool up() const { return comm != nullptr || host_fn != nullptr; }
};

Comm g_comm;

bool load_rccl(Rccl* r) {
  if (r->so) return true;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)
    if (!r->so) r->so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already holds
  const char* paths[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* p : paths)
    if (!r->so) r->so = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (!r->so) return false;
#define GRB_SYM(field, name)                                      \
  r->field = reinterpret_cast<decltype(r->field)>(dlsym(r->so, name)); \
  if (!r->field) return false;
  GRB_SYM(GetUniqueId, "ncclGetUniqueId")
  GRB_SYM(CommInitRank, "ncclCommInitRank")
  GRB_SYM(CommDestroy, "ncclCommDestroy")
  GRB_SYM(AllGather, "ncclAllGather")
  GRB_SYM(Broadcast, "ncclBroadcast")
  GRB_SYM(AllReduce, "ncclAllReduce")
  GRB_SYM(GroupStart, "ncclGroupStart")
  GRB_SYM(GroupEnd, "ncclGroupEnd")
  GRB_SYM(GetErrorString, "ncclGetErrorString")
#undef GRB_SYM
  return true;
}

#define GRB_NCCL_TRY(call)                                                                             \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      fprintf(stderr, "RCCL error: %s at %s:%d\n", g_comm.api.GetErrorString(r_), __FILE__, __LINE__); \
      return GRB_PANIC;                                                                                \
    }                                                                                                  \
  } while (0)

// the collective may start once everything enqueued on the compute stream so far has finished
grb_info fence_in() {
  Comm& c = g_comm;
  GRB_HIP_TRY(hipEventRecord(c.ev_ready, grb::ctx().stream));
  GRB_HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_ready, 0));
  if (c.timed) GRB_HIP_TRY(hipEventRecord(c.ev_t0, c.stream));
  return GRB_SUCCESS;
}
grb_info fence_out() {
  Comm& c = g_comm;
  if (c.timed) GRB_HIP_TRY(hipEventRecord(c.ev_t1, c.stream));
  GRB_HIP_TRY(hipEventRecord(c.ev_done, c.stream));
  ++c.calls;
  c.pending = true;
  return GRB_SUCCESS;
}

}  // namespace

using namespace grb;

namespace grb {
// PageRank on a row shard: the element-wise tail of an iteration for one chunk of owned rows
//   p_next = y + c ;  r = p_next - p_old ;  *acc += sum r^2        (algorithm/pr.hpp:70-80 per element)
__global__ __launch_bounds__(kBlock) void pr_part_update_kernel(const float* __restrict__ y, const float* __restrict__ p_old,
                                                                float c, float* __restrict__ p_next, Index n,
                                                                double* __restrict__ acc) {
  __shared__ double s_sum[kWavesPerBlock];
  double a = 0.0;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float pn = y[i] + c;
    const float r = pn - p_old[i];
    p_next[i] = pn;
    a += (double)(r * r);
  }
  a = wave_reduce(a, [](double x, double z) { return x + z; });
  if (lane_id() == 0) s_sum[wave_id()] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += s_sum[w];
    if (t != 0.0) atomicAdd(acc, t);
  }
}
}  // namespace grb

extern "C" {

grb_info grb_pr_part_update(const void* d_y, const void* d_p_old, float c, void* d_p_next, grb_index n, void* d_acc) { GRB_API_ENTER();
  if (!d_acc) return GRB_NULL_POINTER;
  if (n <= 0) return GRB_SUCCESS;
  if (!d_y || !d_p_old || !d_p_next) return GRB_NULL_POINTER;
  GRB_TRY(ctx_init());
  int grid = stream_grid(n, kBlock * 4);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(pr_part_update_kernel, dim3(grid), dim3(kBlock), 0, ctx().stream, (const float*)d_y,
                     (const float*)d_p_old, c, (float*)d_p_next, n, (double*)d_acc);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

grb_info grb_comm_unique_id(void* out128) { GRB_API_ENTER();
  if (!out128) return GRB_NULL_POINTER;
  if (!load_rccl(&g_comm.api)) return GRB_NOT_IMPLEMENTED;          // no RCCL on this box
  ncclUniqueId id;
  GRB_NCCL_TRY(g_comm.api.GetUniqueId(&id));
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof(id));
  return GRB_SUCCESS;
}

grb_info grb_comm_init(const void* id128, int rank, int world) { GRB_API_ENTER();
  if (!id128 || world < 1 || rank < 0 || rank >= world) return GRB_INVALID_VALUE;
  Comm& c = g_comm;
  if (c.comm) return GRB_OUTPUT_NOT_EMPTY;
  if (!load_rccl(&c.api)) return GRB_NOT_IMPLEMENTED;
  GRB_TRY(ctx_init());
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  GRB_NCCL_TRY(c.api.CommInitRank(&c.comm, world, id, rank));
  c.rank = rank;
  c.world = world;
  GRB_HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  GRB_HIP_TRY(hipEventCreateWithFlags(&c.ev_ready, hipEventDisableTiming));
  GRB_HIP_TRY(hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming));
  GRB_HIP_TRY(hipEventCreate(&c.ev_t0));
  GRB_HIP_TRY(hipEventCreate(&c.ev_t1));
  c.total_us = 0;
  c.calls = 0;
  return GRB_SUCCESS;
}

// ---- host-staged transport ---------------------------------------------------------------------------------------
// The same entry points, the data moved by the caller's own collective over HOST memory: a collective waits for the
// compute stream, copies what it sends into pinned memory, calls fn, copies what it received back.  Synchronous and
// slow on purpose -- it exists so that the device-side level / round loops (bfs_part_run.hip, sssp_part_run.hip),
// whose host logic only ever sees these entry points, can be driven by several PROCESSES where RCCL cannot run
// (ranks that share one GPU in the tests; a gloo group).  fn(user, op, send, recv, bytes, offsets, counts) -> 0 on
// success; op 0: all-gather of `bytes` per rank (send -> recv[world * bytes]); op 1: in-place all-gather of the
// byte ranges [offsets[r], offsets[r] + counts[r]) of recv (send == recv, bytes = the buffer's extent); op 2: sum
// all-reduce of bytes / 8 doubles in place (send == recv).
static grb_info host_stage(size_t bytes) {
  Comm& c = g_comm;
  if (bytes <= c.h_cap) return GRB_SUCCESS;
  if (c.h_stage) (void)hipHostFree(c.h_stage);
  c.h_stage = nullptr;
  c.h_cap = 0;
  GRB_HIP_TRY(hipHostMalloc((void**)&c.h_stage, bytes, hipHostMallocDefault));
  c.h_cap = bytes;
  return GRB_SUCCESS;
}

grb_info grb_comm_set_host_transport(int rank, int world, grb_comm_host_fn fn, void* user) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.comm) return GRB_OUTPUT_NOT_EMPTY;
  if (!fn) {                                             // off: the pinned staging block goes with it
    if (c.h_stage) (void)hipHostFree(c.h_stage);
    c.h_stage = nullptr;
    c.h_cap = 0;
    c.host_fn = nullptr;
    c.host_user = nullptr;
    c.rank = 0;
    c.world = 1;
    return GRB_SUCCESS;
  }
  if (world < 1 || rank < 0 || rank >= world) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  c.host_fn = fn;
  c.host_user = user;
  c.rank = rank;
  c.world = world;
  c.total_us = 0;
  c.calls = 0;
  return GRB_SUCCESS;
}

grb_info grb_comm_destroy(void) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.host_fn) {
    if (c.h_stage) (void)hipHostFree(c.h_stage);
    c.h_stage = nullptr;
    c.h_cap = 0;
    c.host_fn = nullptr;
    c.host_user = nullptr;
    c.rank = 0;
    c.world = 1;
    return GRB_SUCCESS;
  }
  if (!c.comm) return GRB_SUCCESS;
  (void)hipStreamSynchronize(c.stream);
  (void)c.api.CommDestroy(c.comm);
  (void)hipStreamDestroy(c.stream);
  for (hipEvent_t e : {c.ev_ready, c.ev_done, c.ev_t0, c.ev_t1})
    if (e) (void)hipEventDestroy(e);
  Rccl api = c.api;
  c = Comm();
  c.api = api;
  return GRB_SUCCESS;
}

grb_info grb_comm_info(int* rank, int* world) { GRB_API_ENTER();
  if (rank) *rank = g_comm.rank;
  if (world) *world = g_comm.up() ? g_comm.world : 0;
  return GRB_SUCCESS;
}

// Per-collective timing (HIP events on the communication stream) on / off; grb_comm_stats reads and resets.
grb_info grb_comm_timing(int on) { GRB_API_ENTER();
  g_comm.timed = on != 0;
  return GRB_SUCCESS;
}

// Waits (host) for the last collective, then adds its duration to the running total when timing is on.
static grb_info account_last() {
  Comm& c = g_comm;
  if (!c.timed || !c.pending) return GRB_SUCCESS;
  c.pending = false;
  GRB_HIP_TRY(hipEventSynchronize(c.ev_t1));
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, c.ev_t0, c.ev_t1));
  c.total_us += (double)ms * 1e3;
  return GRB_SUCCESS;
}

grb_info grb_comm_stats(double* total_us, long long* calls, int reset) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (total_us) *total_us = c.total_us;
  if (calls) *calls = c.calls;
  if (reset) { c.total_us = 0; c.calls = 0; }
  return GRB_SUCCESS;
}

// The compute stream waits (on the device) for the last collective; no host synchronisation.
grb_info grb_comm_wait(void) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.host_fn) return GRB_SUCCESS;                      // host-staged collectives have completed when they return
  if (!c.comm) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(account_last());
  GRB_HIP_TRY(hipStreamWaitEvent(ctx().stream, c.ev_done, 0));
  return GRB_SUCCESS;
}

// Equal-sized all-gather: every rank sends `bytes` from d_send, d_recv holds world * bytes.
grb_info grb_comm_allgather(const void* d_send, void* d_recv, size_t bytes) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.host_fn) {
    hipStream_t s = grb::ctx().stream;
    GRB_TRY(host_stage(bytes * (size_t)(c.world + 1)));
    GRB_HIP_TRY(hipMemcpyAsync(c.h_stage, d_send, bytes, hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    if (c.host_fn(c.host_user, 0, c.h_stage, c.h_stage + bytes, (long long)bytes, nullptr, nullptr) != 0) return GRB_PANIC;
    GRB_HIP_TRY(hipMemcpyAsync(d_recv, c.h_stage + bytes, bytes * (size_t)c.world, hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    ++c.calls;
    return GRB_SUCCESS;
  }
  if (!c.comm) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(account_last());
  GRB_TRY(fence_in());
  GRB_NCCL_TRY(c.api.AllGather(d_send, d_recv, bytes, ncclUint8, c.comm, c.stream));
  return fence_out();
}

// In-place all-gather of unequal slices: rank r's slice is d_buf[offsets[r] .. offsets[r] + counts[r]) bytes,
// already in place on rank r; afterwards every rank holds every slice.  One broadcast per rank inside a group:
// on point-to-point xGMI every peer link carries its slice concurrently.
grb_info grb_comm_allgatherv_inplace(void* d_buf, const long long* offsets, const long long* counts) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.host_fn) {
    if (!offsets || !counts) return GRB_NULL_POINTER;
    hipStream_t s = grb::ctx().stream;
    long long extent = 0;
    for (int r = 0; r < c.world; ++r) {
      if (counts[r] <= 0) continue;
      if (offsets[r] < 0) return GRB_INVALID_VALUE;                    // a slice before the buffer
      for (int q = 0; q < r; ++q)                                      // two ranks' slices must not overlap
        if (counts[q] > 0 && offsets[r] < offsets[q] + counts[q] && offsets[q] < offsets[r] + counts[r]) return GRB_INVALID_VALUE;
      if (offsets[r] + counts[r] > extent) extent = offsets[r] + counts[r];
    }
    if (extent == 0) return GRB_SUCCESS;
    GRB_TRY(host_stage((size_t)extent));
    if (counts[c.rank] > 0)
      GRB_HIP_TRY(hipMemcpyAsync(c.h_stage + offsets[c.rank], (const char*)d_buf + offsets[c.rank], (size_t)counts[c.rank],
                                 hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    if (c.host_fn(c.host_user, 1, c.h_stage, c.h_stage, extent, offsets, counts) != 0) return GRB_PANIC;
    for (int r = 0; r < c.world; ++r)
      if (r != c.rank && counts[r] > 0)
        GRB_HIP_TRY(hipMemcpyAsync((char*)d_buf + offsets[r], c.h_stage + offsets[r], (size_t)counts[r], hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    ++c.calls;
    return GRB_SUCCESS;
  }
  if (!c.comm) return GRB_UNINITIALIZED_OBJECT;
  if (!offsets || !counts) return GRB_NULL_POINTER;
  GRB_TRY(account_last());
  GRB_TRY(fence_in());
  if (c.world > 1) {
    GRB_NCCL_TRY(c.api.GroupStart());
    for (int r = 0; r < c.world; ++r) {
      if (counts[r] <= 0) continue;
      char* p = (char*)d_buf + offsets[r];
      GRB_NCCL_TRY(c.api.Broadcast(p, p, (size_t)counts[r], ncclUint8, r, c.comm, c.stream));
    }
    GRB_NCCL_TRY(c.api.GroupEnd());
  }
  return fence_out();
}

// In-place sum all-reduce of `count` doubles (convergence scalars, totals).
grb_info grb_comm_allreduce_sum_f64(void* d_buf, size_t count) { GRB_API_ENTER();
  Comm& c = g_comm;
  if (c.host_fn) {
    hipStream_t s = grb::ctx().stream;
    GRB_TRY(host_stage(8 * count));
    GRB_HIP_TRY(hipMemcpyAsync(c.h_stage, d_buf, 8 * count, hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    if (c.host_fn(c.host_user, 2, c.h_stage, c.h_stage, (long long)(8 * count), nullptr, nullptr) != 0) return GRB_PANIC;
    GRB_HIP_TRY(hipMemcpyAsync(d_buf, c.h_stage, 8 * count, hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    ++c.calls;
    return GRB_SUCCESS;
  }
  if (!c.comm) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(account_last());
  GRB_TRY(fence_in());
  GRB_NCCL_TRY(c.api.AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, c.comm, c.stream));
  return fence_out();
}

// The whole partitioned PageRank iteration loop in the library (graphblas/algorithm/pr.hpp:52-90 on a 1-D partition):
// per iteration, for every row chunk of the in-edge shard: y = chunk . p (PlusMultiplies SpMV), p_next = y + c with
// the squared residual accumulated, the chunk's slice of p_next all-gathered on the communication stream while the
// next chunk is multiplied; then the residual is all-reduced and read once (the one host read of the iteration).
//   chunks[c]      : matrix of local rows [row_cut[c], row_cut[c+1]) over all n columns (values alpha / outdeg)
//   vertex_cut     : [world][nchunks + 1] global vertex ids of every rank's chunk boundaries (same on every rank)
//   d_p_cur/d_p_next: n floats each; *result_in_next tells which holds the result
grb_info grb_pr_part_run(int nchunks, const grb_matrix* chunks, const long long* row_cut, const long long* vertex_cut,
                         grb_index lo, float c_add, float eps, int max_niter, void* d_p_cur, void* d_p_next, void* d_y,
                         void* d_acc, int* iterations, double* errors, int* result_in_next) { GRB_API_ENTER();
  if (!chunks || !row_cut || !vertex_cut || !d_p_cur || !d_p_next || !d_y || !d_acc || !iterations) return GRB_NULL_POINTER;
  if (nchunks <= 0 || nchunks > 64) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  Comm& cm = g_comm;
  const int world = (cm.comm || cm.host_fn) ? cm.world : 1;
  const bool have_comm = cm.comm || cm.host_fn;
  std::vector<long long> off((size_t)world * nchunks), cnt((size_t)world * nchunks);
  for (int c = 0; c < nchunks; ++c)
    for (int r = 0; r < world; ++r) {
      const long long* vc = vertex_cut + (size_t)r * (nchunks + 1);
      off[(size_t)c * world + r] = 4 * vc[c];
      cnt[(size_t)c * world + r] = 4 * (vc[c + 1] - vc[c]);
    }
  hipStream_t s = ctx().stream;
  double h_acc = 0.0;
  float* p_cur = (float*)d_p_cur;
  float* p_next = (float*)d_p_next;
  float* y = (float*)d_y;
  int it = 0;
  float error = 1.0f;
  grb_info info = GRB_SUCCESS;
  auto step = [&](grb_info r) { if (info == GRB_SUCCESS && r != GRB_SUCCESS) info = r; return info == GRB_SUCCESS; };
  while (error > eps && it < max_niter && info == GRB_SUCCESS) {
    if (hipMemsetAsync(d_acc, 0, 8, s) != hipSuccess) { info = GRB_PANIC; break; }
    for (int c = 0; c < nchunks && info == GRB_SUCCESS; ++c) {
      const long long a = row_cut[c], b = row_cut[c + 1];
      if (b > a) {
        if (!step(grb_k_spmv(chunks[c], 0, GRB_PLUS_MULTIPLIES, p_cur, nullptr, 0, 0, y + a))) break;
        if (!step(grb_pr_part_update(y + a, p_cur + lo + a, c_add, p_next + lo + a, (grb_index)(b - a), d_acc))) break;
      }
      if (have_comm && world > 1)
        step(grb_comm_allgatherv_inplace(p_next, off.data() + (size_t)c * world, cnt.data() + (size_t)c * world));
    }
    if (info != GRB_SUCCESS) break;
    if (have_comm && world > 1) {
      if (!step(grb_comm_allreduce_sum_f64(d_acc, 1))) break;
      if (!step(grb_comm_wait())) break;
    }
    if (!step(fetch_ints((const int*)d_acc, 2, (int*)&h_acc))) break;      // pinned mailbox; the iteration's one wait
    error = sqrtf((float)h_acc);                      // pr.hpp:84-86: the residual is reduced and rooted in float
    if (errors) errors[it] = (double)error;
    std::swap(p_cur, p_next);
    ++it;
  }
  *iterations = it;
  if (result_in_next) *result_in_next = (p_cur != (float*)d_p_cur) ? 1 : 0;
  return info;
}

}  // extern "C"
