// bfs_kernels.hpp -- BFS device code shared by the single-GPU fused loop (bfs_fused.hip)
// and the 1-D partitioned level steps (bfs_part.hip).
#pragma once
#include "push_common.hpp"

namespace grb {

constexpr int kPullProbe = 4;

struct BfsPushVisitor {
  unsigned int* visited;
  float* label;
  float new_label;
  __device__ bool peek(Index dst) const { return !((visited[dst >> 5] >> (dst & 31)) & 1u); }
  __device__ void visit(Index, Index, Index dst) const {
    const unsigned int bit = 1u << (dst & 31);
    const unsigned int old = atomicOr(&visited[dst >> 5], bit);
    if (label && !(old & bit)) label[dst] = new_label;
  }
};

__device__ inline bool bit_set(const unsigned int* __restrict__ bm, Index v) {
  return (bm[v >> 5] >> (v & 31)) & 1u;
}

template <bool kCountInspected>
static __global__ __launch_bounds__(kBlock) void bfs_pull_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n,
    const unsigned int* __restrict__ vin /* visited bitmap indexed by NEIGHBOUR id */,
    const unsigned int* __restrict__ vin_own /* same bitmap at this row range's own words */,
    const unsigned int* __restrict__ skip, const Index* __restrict__ hint /* may be null */,
    unsigned int* __restrict__ vout, int only_new, float* __restrict__ label, float new_label,
    unsigned long long* __restrict__ inspected_out /* profile only */) {
  __shared__ unsigned long long blk_inspected;
  const int lane = lane_id();
  const Index nchunks = (n + kWave - 1) / kWave;
  const Index wave_global = (Index)blockIdx.x * kWavesPerBlock + wave_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  unsigned long long inspected = 0;
  if (kCountInspected) {
    if (threadIdx.x == 0) blk_inspected = 0ull;
    __syncthreads();
  }
  for (Index chunk = wave_global; chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const unsigned int word = vin_own[(chunk << 1) + (lane >> 5)];
    // `skip` marks vertices without in-edges: never discoverable, but NOT visited (they may
    // still be somebody's in-neighbour, so they must not look visited to the hit test)
    const bool was = ((word | skip[(chunk << 1) + (lane >> 5)]) >> (lane & 31)) & 1u;
    bool active = (v < n) && !was;
    unsigned long long act_mask = __ballot(active);
    if (act_mask == 0ull) {                       // whole chunk already visited / unreachable
      if (lane == 0) vout[chunk << 1] = only_new ? 0u : word;
      if (lane == 32) vout[(chunk << 1) + 1] = only_new ? 0u : word;
      continue;
    }
    Index p = 0, e = 0;
    bool found = false;
    if (active && hint) {
      // the in-neighbour most likely to be in an early frontier, kept in a dense side array:
      // a coalesced 4 B read instead of one adjacency-list cache line per vertex
      if (kCountInspected) ++inspected;
      found = bit_set(vin, hint[v]);
    }
    if (active && !found) {
      p = ptr[v];
      e = ptr[v + 1];
      const Index stop = (e - p > kPullProbe) ? p + kPullProbe : e;
      for (; p < stop; ++p) {
        if (kCountInspected) ++inspected;
        if (bit_set(vin, ind[p])) { found = true; break; }
      }
      if (found) p = e;
    }
    unsigned long long todo = __ballot(active && p < e);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index rs = __shfl(p, src, kWave), re = __shfl(e, src, kWave);
      bool any = false;
      for (Index q = rs; q < re; q += kWave) {
        bool h = false;
        if (q + lane < re) h = bit_set(vin, ind[q + lane]);
        const unsigned long long hb = __ballot(h);
        if (kCountInspected && lane == 0) {
          // early-exit accounting: edges up to and including the first hit
          Index span = (re - q < kWave) ? re - q : kWave;
          inspected += hb ? (unsigned long long)__ffsll((long long)hb) : (unsigned long long)span;
        }
        if (hb) { any = true; break; }
      }
      if (lane == src && any) found = true;
    }
    const unsigned long long fb = __ballot(found);
    const unsigned int keep = only_new ? 0u : word;
    if (lane == 0) vout[chunk << 1] = keep | (unsigned int)(fb & 0xffffffffull);
    if (lane == 32) vout[(chunk << 1) + 1] = keep | (unsigned int)(fb >> 32);
    if (found) label[v] = new_label;
  }
  if (kCountInspected) {
    inspected = wave_reduce(inspected, [](unsigned long long a, unsigned long long b) { return a + b; });
    if (lane == 0) atomicAdd(&blk_inspected, inspected);
    __syncthreads();
    if (threadIdx.x == 0 && blk_inspected) atomicAdd(inspected_out, blk_inspected);
  }
}

// bit v set  <=>  vertex v has no stored entry in this orientation (ptr[v+1] == ptr[v]);
// bits >= n of the last words are set too, so padding never looks "unvisited"
static __global__ void bfs_empty_rows_kernel(const Index* __restrict__ ptr, Index n, int nwords,
                                      unsigned int* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
    unsigned int w = 0u;
    for (int b = 0; b < 32; ++b) {
      Index v = (Index)i * 32 + b;
      if (v >= n || ptr[v + 1] == ptr[v]) w |= 1u << b;
    }
    out[i] = w;
  }
}

// TEPS numerator: sum of out-degree over labelled vertices, and their count.
// Partials per workgroup, spread over 32 slots to keep atomics off a single address.
static __global__ void bfs_tally_kernel(const float* __restrict__ label, const Index* __restrict__ ptr, Index n,
                                 unsigned long long* __restrict__ out /*[32][2]: edges, reached*/) {
  __shared__ unsigned long long se[kWavesPerBlock], sr[kWavesPerBlock];
  unsigned long long edges = 0, reached = 0;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (label[i] != 0.f) { edges += (unsigned long long)(ptr[i + 1] - ptr[i]); ++reached; }
  }
  edges = wave_reduce(edges, [](unsigned long long a, unsigned long long b) { return a + b; });
  reached = wave_reduce(reached, [](unsigned long long a, unsigned long long b) { return a + b; });
  if (lane_id() == 0) { se[wave_id()] = edges; sr[wave_id()] = reached; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long e = 0, r = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) { e += se[w]; r += sr[w]; }
    const int slot = blockIdx.x & 31;
    atomicAdd(&out[slot * 2], e);
    atomicAdd(&out[slot * 2 + 1], r);
  }
}

// One launch closes a BFS level: discovered = |now & ~before| and the out-degree sum of those
// vertices, per-tile counts kept for a later ordered queue listing, and -- by the last
// workgroup to finish -- the level's record written straight into the pinned host mailbox.
// Inter-workgroup hand-off without fences (CDNA guide, G16 "R1/R2"): partials are agent-scope
// write-through stores, drained with s_waitcnt before the ticket; the last workgroup reads
// them with agent-scope (L1-bypassing) loads; the host record is four 8-byte granules
// {value, seq} -- the data is the flag, so no ordering between them is needed.
//   granule 0 discovered   1 expanded edges (push)   2 / 3 cumulative out-degree of everything
//   discovered so far (lo / hi)   4 / 5 inspected edges (profile runs) lo / hi
static __global__ __launch_bounds__(kBlock) void bfs_level_tail_kernel(
    const unsigned int* __restrict__ now, const unsigned int* __restrict__ before, int nwords,
    const Index* __restrict__ out_ptr, Index n, int* tile_counts, unsigned long long* tile_deg,
    unsigned int* ticket, const int* __restrict__ d_state, unsigned long long* edges_acc,
    unsigned long long* mail, int seq) {
  __shared__ int s_cnt[kWavesPerBlock];
  __shared__ unsigned long long s_deg[kWavesPerBlock];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * kBlock + tid;
  unsigned int d = (i < nwords) ? (now[i] & ~before[i]) : 0u;
  int c = __popc(d);
  unsigned long long deg = 0;
  while (d) {
    const int b = __ffs((int)d) - 1;
    d &= d - 1;
    const Index v = (Index)i * 32 + b;
    if (v < n) deg += (unsigned long long)(out_ptr[v + 1] - out_ptr[v]);
  }
  c = wave_reduce(c, [](int a, int b) { return a + b; });
  deg = wave_reduce(deg, [](unsigned long long a, unsigned long long b) { return a + b; });
  if (lane_id() == 0) { s_cnt[wave_id()] = c; s_deg[wave_id()] = deg; }
  __syncthreads();
  if (tid == 0) {
    int tc = 0;
    unsigned long long td = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) { tc += s_cnt[w]; td += s_deg[w]; }
    __hip_atomic_store(&tile_counts[blockIdx.x], tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&tile_deg[blockIdx.x], td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  int tc = 0;
  unsigned long long td = 0;
  for (int j = tid; j < (int)gridDim.x; j += kBlock) {
    tc += __hip_atomic_load(&tile_counts[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    td += __hip_atomic_load(&tile_deg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  tc = wave_reduce(tc, [](int a, int b) { return a + b; });
  td = wave_reduce(td, [](unsigned long long a, unsigned long long b) { return a + b; });
  __syncthreads();
  if (lane_id() == 0) { s_cnt[wave_id()] = tc; s_deg[wave_id()] = td; }
  __syncthreads();
  if (tid == 0) {
    int c2 = 0;
    unsigned long long d2 = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) { c2 += s_cnt[w]; d2 += s_deg[w]; }
    const unsigned long long acc = *edges_acc + d2;
    *edges_acc = acc;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long tag = (unsigned long long)(unsigned int)seq << 32;
    const unsigned int vals[6] = {(unsigned int)c2, (unsigned int)d_state[1], (unsigned int)(acc & 0xffffffffull),
                                  (unsigned int)(acc >> 32), (unsigned int)d_state[2], (unsigned int)d_state[3]};
#pragma unroll
    for (int k = 0; k < 6; ++k)
      __hip_atomic_store(&mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// hint[v] = the entry of row v whose own degree (deg_ptr) is largest (the smallest such entry on ties), -1 for an empty
// row.  A wave takes 64 consecutive rows: their entries are ONE contiguous piece of `ind`, streamed 64 at a time
// (coalesced; the first version walked a row per lane, every load 64 lines); an entry's row is found by a 6-step search
// of the 65 row starts in LDS and its (degree, entry) key folded into the row's slot with an LDS 64-bit max.
// Rows of kHintLong entries and more are stepped over here and taken by bfs_hint_long_kernel, a whole workgroup each:
// one wave walking the longest row of RMAT-22 alone (300 000 entries) was this kernel's whole time, 2.7 ms.
constexpr Index kHintLong = 1024;
// The degrees are looked up once per ENTRY, at random: two 4-byte reads of a 16 MB pointer array per entry were 9.1 GB
// of line fetches for the 1 GB graph of RMAT-22 (profiles/r05/pmc_traffic.json), most of the 5.3 ms a matrix's first
// traversal paid.  The hint only has to be a GOOD in-neighbour -- any in-neighbour is a correct one -- so the kernels
// compare degree CLASSES, one byte per vertex (exact below 128, then sixteen steps per doubling: monotone, so "largest
// class, smallest entry on ties" is still a maximum-degree neighbour up to 4 % of its degree): 4 MB at RMAT-22, which
// the L2s hold.
__device__ inline unsigned char hint_degree_class(unsigned int d) {
  if (d < 128u) return (unsigned char)d;
  const int e = 31 - __clz((int)d);                            // d in [2^e, 2^(e+1)), e >= 7
  const unsigned int c = 128u + (unsigned int)(e - 7) * 16u + ((d >> (e - 4)) & 15u);
  return (unsigned char)(c > 255u ? 255u : c);
}
static __global__ void bfs_degree_class_kernel(const Index* __restrict__ deg_ptr, Index n, unsigned char* __restrict__ cls) {
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (Index)gridDim.x * blockDim.x)
    cls[v] = hint_degree_class((unsigned int)(deg_ptr[v + 1] - deg_ptr[v]));
}
__device__ inline unsigned long long hint_key(Index u, const unsigned char* __restrict__ cls) {
  // (class + 1: an entry of degree 0 still beats "no entry"; ties go to the smaller entry)
  return ((unsigned long long)((unsigned int)cls[u] + 1u) << 32) | (unsigned long long)(0xffffffffu - (unsigned int)u);
}
static __global__ __launch_bounds__(kBlock) void bfs_hint_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n, const unsigned char* __restrict__ deg_ptr,
    Index* __restrict__ hint) {
  __shared__ Index s_start[kWavesPerBlock][kWave + 1];
  __shared__ unsigned long long s_best[kWavesPerBlock][kWave];
  const int lane = lane_id(), wave = wave_id();
  const Index nchunks = (n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave; chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const Index vc = v < n ? v : n;                              // rows past the end: empty
    const Index my_start = ptr[vc], my_end = ptr[vc < n ? vc + 1 : n];
    s_start[wave][lane] = my_start;
    if (lane == kWave - 1) s_start[wave][kWave] = my_end;
    s_best[wave][lane] = 0ull;
    const unsigned long long longs = __ballot(my_end - my_start >= kHintLong);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Index q1 = s_start[wave][kWave];
    Index base = s_start[wave][0];
    while (base < q1) {
      const Index q = base + lane < q1 ? base + lane : q1 - 1;
      int r = 0;                                                 // the last row whose start is <= q
#pragma unroll
      for (int step = kWave / 2; step > 0; step >>= 1)
        if (s_start[wave][r + step] <= q) r += step;
      const int r0 = __builtin_amdgcn_readfirstlane(r);          // the row this step begins in
      if ((longs >> r0) & 1ull) { base = s_start[wave][r0 + 1]; continue; }
      if (base + lane < q1 && !((longs >> r) & 1ull)) atomicMax(&s_best[wave][r], hint_key(ind[q], deg_ptr));
      base += kWave;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (v < n && !((longs >> lane) & 1ull)) {
      const unsigned long long b = s_best[wave][lane];
      hint[v] = b ? (Index)(0xffffffffu - (unsigned int)b) : -1;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// the rows of kHintLong entries and more: a workgroup looks at 1024 consecutive rows, lists the long ones, and walks
// each of them with all its threads
static __global__ __launch_bounds__(1024) void bfs_hint_long_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n, const unsigned char* __restrict__ deg_ptr,
    Index* __restrict__ hint) {
  __shared__ Index s_rows[1024];
  __shared__ int s_n;
  __shared__ unsigned long long s_max;
  for (Index v0 = (Index)blockIdx.x * 1024; v0 < n; v0 += (Index)gridDim.x * 1024) {
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const Index v = v0 + (Index)threadIdx.x;
    if (v < n && ptr[v + 1] - ptr[v] >= kHintLong) s_rows[atomicAdd(&s_n, 1)] = v;
    __syncthreads();
    const int nl = s_n;
    for (int k = 0; k < nl; ++k) {
      const Index row = s_rows[k];
      if (threadIdx.x == 0) s_max = 0ull;
      __syncthreads();
      unsigned long long best = 0ull;
      const Index e = ptr[row + 1];
      for (Index q = ptr[row] + (Index)threadIdx.x; q < e; q += 1024) {
        const unsigned long long key = hint_key(ind[q], deg_ptr);
        best = key > best ? key : best;
      }
      // wave maximum (a handful of LDS atomics per wave would do as well; sixteen of them per row is nothing)
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, kWave);
        best = o > best ? o : best;
      }
      if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(&s_max, best);
      __syncthreads();
      if (threadIdx.x == 0) hint[row] = (Index)(0xffffffffu - (unsigned int)s_max);      // (a long row is not empty)
      __syncthreads();
    }
  }
}

// Lazily built, cached per matrix: for every row of `M` its entry of largest degree.
static inline grb_info ensure_pull_hint(Index** cache, const CsrArrays& M, const Index* deg_ptr, hipStream_t s) {
  if (*cache) return GRB_SUCCESS;
  GRB_HIP_TRY(hipMalloc((void**)cache, 4 * (size_t)(M.n > 0 ? M.n : 1)));
  if (M.n > 0) {
    unsigned char* d_cls = nullptr;                              // (stream-ordered: freed when the kernels below have run)
    GRB_HIP_TRY(hipMallocAsync((void**)&d_cls, (size_t)M.n, s));
    hipLaunchKernelGGL(bfs_degree_class_kernel, dim3(stream_grid(M.n)), dim3(kBlock), 0, s, deg_ptr, M.n, d_cls);
    hipLaunchKernelGGL(bfs_hint_kernel, dim3(stream_grid((long long)ceil_div(M.n, kWave) * kWave, kBlock)),
                       dim3(kBlock), 0, s, M.ptr, M.ind, M.n, (const unsigned char*)d_cls, *cache);
    hipLaunchKernelGGL(bfs_hint_long_kernel, dim3(stream_grid((long long)M.n, 1024)), dim3(1024), 0, s, M.ptr, M.ind, M.n,
                       (const unsigned char*)d_cls, *cache);
    const hipError_t le = hipGetLastError();
    GRB_HIP_TRY(hipFreeAsync(d_cls, s));
    GRB_HIP_TRY(le);
  }
  return GRB_SUCCESS;
}

// Lazily built, cached per matrix: bitmap of rows of `M` without entries.
static inline grb_info ensure_empty_rows(unsigned int** cache, const CsrArrays& M, hipStream_t s) {
  if (*cache) return GRB_SUCCESS;
  const int nwords = 2 * ceil_div(M.n, 64);
  GRB_HIP_TRY(hipMalloc((void**)cache, 4 * (size_t)(nwords > 0 ? nwords : 1)));
  if (nwords > 0) {
    hipLaunchKernelGGL(bfs_empty_rows_kernel, dim3(stream_grid(nwords)), dim3(kBlock), 0, s, M.ptr, M.n, nwords, *cache);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}

}  // namespace grb
