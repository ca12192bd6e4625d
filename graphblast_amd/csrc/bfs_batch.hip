// bfs_batch.hip -- up to 64 traversals at once: the multi-frontier form of algorithm::bfs.
//
// In GraphBLAS terms a level of k simultaneous traversals is  F' = (A^T lor.land F) .* not(Seen)
// with F, Seen : n x k Boolean -- the sparse x dense product the reference leaves as a stub
// (backend/cuda/operations.hpp:52-70, spmm.hpp:15-27).  With k <= 64 a row of F is ONE 64-bit
// word (bit s = "in the frontier of source s"), the semiring's add is a word-wide OR, its
// multiply the AND with the edge's presence: one 8-byte gather per edge serves 64 traversals.
// No tile of A is dense enough for a matrix core to beat that (multi-frontier): the
// MFMA path of this library is the dense-core SpMM in spmm.hip.
//
//   seen[v], F[v]   one 64-bit word per vertex; two F buffers (read / written)
//   pull level      a lane per vertex: need = ~seen & active; the hinted in-neighbour first, four
//                   serial probes, then the wave finishes the row together (256 entries per step,
//                   OR-reduced across the wave), stopping as soon as every needed bit is found.
//                   Rows of >= 4096 entries are cut into 4096-entry slices taken by separate waves
//                   (a hub row that finds nothing must not be one wave's 2 MB)
//   push level      frontier words != 0 expand along out-edges with atomicOr into a zeroed F';
//                   an apply pass turns F' into new bits, labels and totals
//   labels          label[s][v] = level of discovery (source = 1, unreached = 0): the k depth
//                   vectors k calls of algorithm::bfs would return -- bit-identical, because BFS
//                   depth is unique
#include "bfs_kernels.hpp"

namespace grb {

constexpr int kBatchBig = 4096;       // row length from which a row is cut into slices
constexpr int kBatchSlice = 4096;
constexpr int kBatchSerial = 4;       // serial probes per lane before the wave takes over
constexpr int kBatchSlots = 64;       // counter slots (spreads same-address atomics)

typedef unsigned long long u64;

struct BatchArgs {
  const Index *optr, *oind, *iptr, *iind;
  const Index* hint;
  Index n;
  u64 amask;
  u64 *seen, *fcur, *fnext;
  u64* bigacc;                        // one word per big row (pull slices OR into it)
  const int4* slices;                 // {vertex, first entry, end entry, big index}
  int nslices;
  const Index* bigrows;               // the big rows' vertex ids
  int nbig;
  u64* counters;                      // [kBatchSlots][4]: vertices, their out-degree sum, pairs, pair edges
  float new_label;
  int k;
  float* label[64];
};

__device__ inline u64 wave_or(u64 x) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) x |= __shfl_xor(x, o, kWave);
  return x;
}

// labels + accounting of a lane's new bits; every lane of the wave must call it
__device__ inline void batch_commit(const BatchArgs& a, Index v, u64 newb, u64 (&tot)[4]) {
  const u64 any = wave_or(newb);
  for (u64 t = any; t; t &= t - 1) {
    const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)t) - 1);   // wave-uniform: a scalar index
    if ((newb >> s) & 1ull) a.label[s][v] = a.new_label;
  }
  if (newb) {
    const u64 d = (u64)(a.optr[v + 1] - a.optr[v]);
    const u64 pc = (u64)__popcll(newb);
    tot[0] += 1; tot[1] += d; tot[2] += pc; tot[3] += pc * d;
  }
}

__device__ inline void batch_flush(const BatchArgs& a, u64 (&tot)[4]) {
  __shared__ u64 s_tot[4];
  if (threadIdx.x < 4) s_tot[threadIdx.x] = 0;
  __syncthreads();
  auto add = [](u64 x, u64 y) { return x + y; };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u64 r = wave_reduce(tot[j], add);
    if (lane_id() == 0 && r) atomicAdd(&s_tot[j], r);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_tot[threadIdx.x])
    atomicAdd(&a.counters[(blockIdx.x & (kBatchSlots - 1)) * 4 + threadIdx.x], s_tot[threadIdx.x]);
}

// the wave scans entries [rs, re) of `ind`, ORs word[ind[q]] and stops once `nd` is covered
__device__ inline u64 wave_scan_or(const Index* __restrict__ ind, const u64* __restrict__ word, Index rs, Index re,
                                   u64 nd, int lane) {
  u64 got = 0;
  for (Index q = rs; q < re; q += 4 * kWave) {
    u64 w = 0;
    Index c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const Index at = q + j * kWave + lane;
      c[j] = at < re ? ind[at] : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w |= c[j] >= 0 ? word[c[j]] : 0ull;
    got |= wave_or(w);
    if ((got & nd) == nd) break;
  }
  return got;
}

__global__ __launch_bounds__(kBlock) void batch_seed_kernel(BatchArgs a, const Index* __restrict__ sources) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < a.k) {
    const Index v = sources[s];
    atomicOr(&a.seen[v], 1ull << s);
    atomicOr(&a.fcur[v], 1ull << s);
    a.label[s][v] = 1.f;
  }
}

__global__ __launch_bounds__(kBlock) void batch_pull_kernel(BatchArgs a) {
  const int lane = lane_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  u64 tot[4] = {0, 0, 0, 0};
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const bool valid = v < a.n;
    const u64 seen = valid ? a.seen[v] : ~0ull;
    u64 need = ~seen & a.amask;
    Index p = 0, e = 0;
    if (need) { p = a.iptr[v]; e = a.iptr[v + 1]; }
    const bool big = e - p >= kBatchBig;                   // the slice kernels own this row
    if (big || p == e) need = 0;
    if (__ballot(need != 0) == 0ull) {
      if (valid && !big) a.fnext[v] = 0ull;
      continue;
    }
    u64 acc = 0;
    if (need && a.hint) acc = a.fcur[a.hint[v]];
#pragma unroll
    for (int t = 0; t < kBatchSerial; ++t) {
      const bool go = need && (acc & need) != need && p < e;
      const Index c = a.iind[go ? p : 0];
      const u64 w = a.fcur[go ? c : 0];
      if (go) { acc |= w; ++p; }
    }
    u64 todo = __ballot(need && (acc & need) != need && p < e);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index rs = __shfl(p, src, kWave), re = __shfl(e, src, kWave);
      const u64 nd = __shfl(need & ~acc, src, kWave);
      const u64 got = wave_scan_or(a.iind, a.fcur, rs, re, nd, lane);
      if (lane == src) acc |= got;
    }
    const u64 newb = acc & need;
    if (valid && !big) {
      a.fnext[v] = newb;
      if (newb) a.seen[v] = seen | newb;
    }
    batch_commit(a, valid ? v : 0, newb, tot);
  }
  batch_flush(a, tot);
}

// pull, big rows: a wave per 4096-entry slice
__global__ __launch_bounds__(kBlock) void batch_pull_slices_kernel(BatchArgs a) {
  const int lane = lane_id();
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int sl = blockIdx.x * kWavesPerBlock + wave_id(); sl < a.nslices; sl += nwaves) {
    const int4 S = a.slices[sl];
    const u64 need = ~a.seen[S.x] & a.amask;
    if (!need) continue;
    const u64 got = wave_scan_or(a.iind, a.fcur, S.y, S.z, need, lane) & need;
    if (lane == 0 && got) atomicOr(&a.bigacc[S.w], got);
  }
}

__global__ __launch_bounds__(kBlock) void batch_big_apply_kernel(BatchArgs a) {
  u64 tot[4] = {0, 0, 0, 0};
  const int nthreads = gridDim.x * blockDim.x;
  for (int base = 0; base < a.nbig; base += nthreads) {
    const int b = base + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < a.nbig;
    const Index v = valid ? a.bigrows[b] : 0;
    u64 newb = 0;
    if (valid) {
      const u64 seen = a.seen[v];
      newb = a.bigacc[b] & ~seen & a.amask;
      a.bigacc[b] = 0ull;
      a.fnext[v] = newb;
      if (newb) a.seen[v] = seen | newb;
    }
    batch_commit(a, v, newb, tot);
  }
  batch_flush(a, tot);
}

__device__ inline void batch_push_edge(const BatchArgs& a, Index dst, u64 fw) {
  const u64 bits = fw & ~a.seen[dst];
  if (bits && (bits & ~a.fnext[dst])) atomicOr(&a.fnext[dst], bits);
}

__global__ __launch_bounds__(kBlock) void batch_push_kernel(BatchArgs a) {
  const int lane = lane_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index u = chunk * kWave + lane;
    const u64 fw = u < a.n ? a.fcur[u] : 0ull;
    if (__ballot(fw != 0) == 0ull) continue;
    Index p = 0, e = 0;
    if (fw) { p = a.optr[u]; e = a.optr[u + 1]; }
    if (e - p >= kBatchBig) p = e;                         // the slice kernel expands it
    if (e - p <= 8) {
      for (; p < e; ++p) batch_push_edge(a, a.oind[p], fw);
    }
    u64 todo = __ballot(p < e);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index rs = __shfl(p, src, kWave), re = __shfl(e, src, kWave);
      const u64 w = __shfl(fw, src, kWave);
      for (Index q = rs + lane; q < re; q += kWave) batch_push_edge(a, a.oind[q], w);
    }
  }
}

__global__ __launch_bounds__(kBlock) void batch_push_slices_kernel(BatchArgs a) {
  const int lane = lane_id();
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int sl = blockIdx.x * kWavesPerBlock + wave_id(); sl < a.nslices; sl += nwaves) {
    const int4 S = a.slices[sl];
    const u64 fw = a.fcur[S.x];
    if (!fw) continue;
    for (Index q = S.y + lane; q < S.z; q += kWave) batch_push_edge(a, a.oind[q], fw);
  }
}

// after a push level: F' holds ORed candidate bits; make them the new frontier
__global__ __launch_bounds__(kBlock) void batch_apply_kernel(BatchArgs a) {
  const int lane = lane_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  u64 tot[4] = {0, 0, 0, 0};
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const bool valid = v < a.n;
    const u64 raw = valid ? a.fnext[v] : 0ull;
    if (__ballot(raw != 0) == 0ull) continue;
    u64 newb = 0;
    if (raw) {
      const u64 seen = a.seen[v];
      newb = raw & ~seen & a.amask;
      if (newb != raw) a.fnext[v] = newb;
      if (newb) a.seen[v] = seen | newb;
    }
    batch_commit(a, valid ? v : 0, newb, tot);
  }
  batch_flush(a, tot);
}

__global__ void batch_unlabel_kernel(BatchArgs a, float bad) {
  for (int s = 0; s < a.k; ++s)
    for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
      if (a.label[s][i] == bad) a.label[s][i] = 0.f;
}

}  // namespace grb

using namespace grb;

// rows of >= kBatchBig entries cut into slices; cached per matrix and orientation
static grb_info ensure_slices(grb_matrix A, bool in_edges) {
  BatchSlices& B = in_edges ? A->batch_in : A->batch_out;
  if (B.ready) return GRB_SUCCESS;
  const std::vector<Index>& ptr = in_edges ? A->h_csc_ptr : A->h_csr_ptr;
  const Index n = in_edges ? A->ncols : A->nrows;
  if ((Index)ptr.size() != n + 1) return GRB_INVALID_OBJECT;
  std::vector<int4> sl;
  std::vector<Index> rows;
  for (Index v = 0; v < n; ++v) {
    const Index d = ptr[(size_t)v + 1] - ptr[v];
    if (d < kBatchBig) continue;
    for (Index s = ptr[v]; s < ptr[(size_t)v + 1]; s += kBatchSlice)
      sl.push_back(make_int4(v, s, std::min<Index>(s + kBatchSlice, ptr[(size_t)v + 1]), (int)rows.size()));
    rows.push_back(v);
  }
  B.nslices = (int)sl.size();
  B.nbig = (int)rows.size();
  if (B.nslices > 0) {
    GRB_HIP_TRY(hipMalloc((void**)&B.d_slices, sizeof(int4) * sl.size()));
    GRB_HIP_TRY(hipMalloc((void**)&B.d_rows, sizeof(Index) * rows.size()));
    GRB_HIP_TRY(hipMalloc((void**)&B.d_acc, sizeof(u64) * rows.size()));
    GRB_HIP_TRY(hipMemcpy(B.d_slices, sl.data(), sizeof(int4) * sl.size(), hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMemcpy(B.d_rows, rows.data(), sizeof(Index) * rows.size(), hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMemset(B.d_acc, 0, sizeof(u64) * rows.size()));
  }
  B.ready = true;
  return GRB_SUCCESS;
}

extern "C" grb_info grb_bfs_batch(grb_vector* v, int k, grb_matrix A, const grb_index* sources, grb_descriptor desc,
                                  grb_bfs_result* result) {
  if (!v || !A || !desc || !sources) return GRB_UNINITIALIZED_OBJECT;
  if (k < 1 || k > 64) return GRB_INVALID_VALUE;
  if (!A->built || !A->csr.ptr || !A->csc.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (A->nrows != A->ncols) return GRB_DIMENSION_MISMATCH;
  const Index n = A->nrows;
  for (int s = 0; s < k; ++s) {
    if (!v[s]) return GRB_UNINITIALIZED_OBJECT;
    if (v[s]->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
    if (v[s]->nsize != n) return GRB_DIMENSION_MISMATCH;
    if (sources[s] < 0 || sources[s] >= n) return GRB_INVALID_INDEX;
  }
  GRB_TRY(ctx_init());
  Context& c = ctx();
  hipStream_t st = c.stream;
  GRB_TRY(ensure_pull_hint(&A->d_pull_hint, A->csc, A->csr.ptr, st));
  GRB_TRY(ensure_slices(A, true));
  GRB_TRY(ensure_slices(A, false));

  void *p_words, *p_cnt, *p_src;
  GRB_TRY(scratch(7, 3 * sizeof(u64) * (size_t)n + 256, &p_words));
  c.bfs_prezero_ptr = nullptr;                              // slot 7 is the one-launch traversal's pre-zeroed block
  GRB_TRY(scratch(10, sizeof(u64) * kBatchSlots * 4, &p_cnt));
  GRB_TRY(scratch(9, sizeof(Index) * 64, &p_src));
  BatchArgs a;
  a.optr = A->csr.ptr; a.oind = A->csr.ind; a.iptr = A->csc.ptr; a.iind = A->csc.ind;
  a.hint = A->d_pull_hint;
  a.n = n;
  a.amask = k == 64 ? ~0ull : ((1ull << k) - 1ull);
  a.seen = (u64*)p_words;
  a.fcur = a.seen + n;
  a.fnext = a.fcur + n;
  a.counters = (u64*)p_cnt;
  a.k = k;
  for (int s = 0; s < 64; ++s) a.label[s] = nullptr;
  for (int s = 0; s < k; ++s) {
    GRB_TRY(grb_vector_set_storage(v[s], GRB_DENSE));
    a.label[s] = (float*)v[s]->d_val;
    GRB_HIP_TRY(hipMemsetAsync(a.label[s], 0, sizeof(float) * (size_t)n, st));
  }
  GRB_HIP_TRY(hipMemsetAsync(a.seen, 0, 2 * sizeof(u64) * (size_t)n, st));
  GRB_HIP_TRY(hipMemcpyAsync(p_src, sources, sizeof(Index) * (size_t)k, hipMemcpyHostToDevice, st));
  a.new_label = 1.f;
  a.bigacc = nullptr; a.slices = nullptr; a.nslices = 0; a.bigrows = nullptr; a.nbig = 0;
  hipLaunchKernelGGL(batch_seed_kernel, dim3(1), dim3(kBlock), 0, st, a, (const Index*)p_src);
  GRB_HIP_TRY(hipGetLastError());

  // frontier totals of the seed level on the host (k <= 64 sources)
  long long nf = 0, mf = 0;
  unsigned long long edges = 0, reached = 0;
  {
    std::vector<Index> uniq(sources, sources + k);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    nf = (long long)uniq.size();
    for (Index u : uniq) mf += A->h_csr_ptr[(size_t)u + 1] - A->h_csr_ptr[u];
    for (int s = 0; s < k; ++s) edges += (unsigned long long)(A->h_csr_ptr[(size_t)sources[s] + 1] - A->h_csr_ptr[sources[s]]);
    reached = (unsigned long long)k;
  }
  const int mode = desc->desc[GRB_MXVMODE];
  const int grid = stream_grid((long long)ceil_div(n, kWave) * kWave, kBlock);
  int iter = 1, levels = 0, last_dir = 0;
  bool hit_cap = false;
  float ms = 0.f;
  GRB_TRY(grb_timer_start());
  for (; iter <= desc->max_niter; ++iter) {
    // direction: the reference's vertex-count rule on the union frontier, plus the edge-aware switch
    bool pull = mode == GRB_PULLONLY;
    if (mode == GRB_PUSHPULL)
      pull = (double)nf > (double)desc->switchpoint * (double)n || (double)mf > 0.02 * (double)A->nvals;
    a.new_label = (float)(iter + 1);
    GRB_HIP_TRY(hipMemsetAsync(a.counters, 0, sizeof(u64) * kBatchSlots * 4, st));
    if (pull) {
      const BatchSlices& B = A->batch_in;
      a.slices = B.d_slices; a.nslices = B.nslices; a.bigrows = B.d_rows; a.nbig = B.nbig; a.bigacc = B.d_acc;
      hipLaunchKernelGGL(batch_pull_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
      if (B.nslices > 0) {
        hipLaunchKernelGGL(batch_pull_slices_kernel, dim3(stream_grid((long long)B.nslices * kWave, kBlock)), dim3(kBlock),
                           0, st, a);
        GRB_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(batch_big_apply_kernel, dim3(stream_grid(B.nbig, kBlock)), dim3(kBlock), 0, st, a);
        GRB_HIP_TRY(hipGetLastError());
      }
    } else {
      const BatchSlices& B = A->batch_out;
      a.slices = B.d_slices; a.nslices = B.nslices; a.bigrows = B.d_rows; a.nbig = B.nbig; a.bigacc = B.d_acc;
      GRB_HIP_TRY(hipMemsetAsync(a.fnext, 0, sizeof(u64) * (size_t)n, st));
      hipLaunchKernelGGL(batch_push_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
      if (B.nslices > 0) {
        hipLaunchKernelGGL(batch_push_slices_kernel, dim3(stream_grid((long long)B.nslices * kWave, kBlock)), dim3(kBlock),
                           0, st, a);
        GRB_HIP_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(batch_apply_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
    }
    u64 h[kBatchSlots * 4];
    GRB_HIP_TRY(hipMemcpyAsync(h, a.counters, sizeof(h), hipMemcpyDeviceToHost, st));
    GRB_HIP_TRY(hipStreamSynchronize(st));
    u64 t[4] = {0, 0, 0, 0};
    for (int i = 0; i < kBatchSlots; ++i)
      for (int j = 0; j < 4; ++j) t[j] += h[i * 4 + j];
    ++levels;
    last_dir = pull ? 1 : 0;
    nf = (long long)t[0];
    mf = (long long)t[1];
    reached += t[2];
    edges += t[3];
    std::swap(a.fcur, a.fnext);
    if (nf == 0) break;
  }
  if (iter > desc->max_niter && nf > 0) {
    // vertices discovered by the last allowed iteration are never assigned by the reference loop (bfs.hpp:48-66)
    hit_cap = true;
    hipLaunchKernelGGL(batch_unlabel_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, a,
                       (float)(desc->max_niter + 1));
    GRB_HIP_TRY(hipGetLastError());
  }
  GRB_TRY(grb_timer_stop(&ms));
  desc->lastmxv = last_dir ? GRB_PULLONLY : GRB_PUSHONLY;
  if (result) {
    result->levels = levels;
    result->tight_ms = ms;
    result->edges_traversed = hit_cap ? -1 : (int64_t)edges;   // under a cap the tally would count unassigned vertices
    result->reached = hit_cap ? -1 : (int32_t)(reached > 0x7fffffffull ? 0x7fffffff : reached);
  }
  return GRB_SUCCESS;
}
