// build.hip -- graph ingest on the GPU.
//
// The reference builds its matrices on the host: readMtx + removeSelfloop + customSort
// (std::sort of tuples, three passes) + coo2csr / coo2csc (graphblas/util.hpp:169-329,501-600,
// backend/cuda/sparse_matrix.hpp:289-351) -- minutes for 10^8 edges.  Here a coordinate list
// becomes CSR + CSC on the device:
//
//   optional loader semantics (util.hpp:197-329): add the reverse of every off-diagonal entry,
//   drop self loops, drop duplicates (first occurrence wins)
//   stable LSD radix sort of (row << 32 | col) keys with their values, 8-bit digits, only over
//   the digit positions the dimensions need
//   row pointers from the sorted keys; CSC by one more stable sort on the column digits of the
//   CSR-ordered list (rows stay ascending inside every column)
//
// No rocPRIM / hipCUB: the sort is the classic three-step pass (per-block digit histogram,
// exclusive scan of the digit-major count table, stable scatter).  Stability inside a block
// comes from order-preserving ranks: every wave owns a contiguous part of the block's tile and
// ranks 64 items at a time with a ballot-based match of equal digits.
#include "common.hpp"

namespace grb {

constexpr int kSortItems = 16;                         // items per thread
constexpr int kSortTile = kBlock * kSortItems;         // items per workgroup
constexpr int kWaveSpan = kSortTile / kWavesPerBlock;  // contiguous items per wave
constexpr int kScanTile = 2048;

__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const unsigned long long* __restrict__ keys, long long n,
                                                            int shift, int nblocks, unsigned int* __restrict__ cnt) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * kSortTile;
#pragma unroll
  for (int k = 0; k < kSortItems; ++k) {
    const long long i = base + threadIdx.x + (long long)k * kBlock;
    if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  cnt[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];   // digit-major
}

// ---- exclusive scan of an arbitrary-length unsigned array, three small kernels
__global__ __launch_bounds__(kBlock) void scan_local_kernel(unsigned int* __restrict__ a, long long n,
                                                            unsigned int* __restrict__ totals) {
  __shared__ int smem[kWavesPerBlock];
  const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * (kScanTile / kBlock);
  unsigned int v[kScanTile / kBlock];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kScanTile / kBlock; ++k) {
    v[k] = base + k < n ? a[base + k] : 0u;
    sum += (int)v[k];
  }
  int total = 0;
  int off = block_exclusive_scan(sum, smem, total);
#pragma unroll
  for (int k = 0; k < kScanTile / kBlock; ++k) {
    if (base + k < n) a[base + k] = (unsigned int)off;
    off += (int)v[k];
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = (unsigned int)total;
}

__global__ __launch_bounds__(kBlock) void scan_totals_kernel(unsigned int* __restrict__ totals, int nt) {
  __shared__ int smem[kWavesPerBlock];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nt; base += kBlock) {
    const int i = base + threadIdx.x;
    const int v = i < nt ? (int)totals[i] : 0;
    int total = 0;
    const int off = block_exclusive_scan(v, smem, total);
    if (i < nt) totals[i] = (unsigned int)(carry + off);
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(unsigned int* __restrict__ a, long long n,
                                                          const unsigned int* __restrict__ totals) {
  const long long base = (long long)blockIdx.x * kScanTile;
  const unsigned int add = totals[blockIdx.x];
  for (int k = threadIdx.x; k < kScanTile; k += kBlock)
    if (base + k < n) a[base + k] += add;
}

static grb_info exclusive_scan_u32(unsigned int* d, long long n, unsigned int* d_totals, hipStream_t s) {
  const int nt = (int)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_local_kernel, dim3(nt), dim3(kBlock), 0, s, d, n, d_totals);
  hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kBlock), 0, s, d_totals, nt);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nt), dim3(kBlock), 0, s, d, n, (const unsigned int*)d_totals);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

// Stable scatter of one digit pass.  offs[digit * nblocks + block] = first output position of
// this block's items with that digit.
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(
    const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ pay, long long n, int shift,
    int nblocks, const unsigned int* __restrict__ offs, unsigned long long* __restrict__ keys_out,
    unsigned int* __restrict__ pay_out) {
  __shared__ unsigned int wh[kWavesPerBlock][256];     // per-wave digit counts, then running positions
  const int lane = lane_id(), wave = wave_id();
  for (int k = threadIdx.x; k < kWavesPerBlock * 256; k += kBlock) (&wh[0][0])[k] = 0;
  __syncthreads();
  const long long wbase = (long long)blockIdx.x * kSortTile + (long long)wave * kWaveSpan;
  // a lane's sixteen items, read ONCE and all at once (the ranking loop below is a chain of LDS updates behind a fence:
  // a load inside it waits out a full memory latency per step, sixteen times per wave)
  unsigned long long kreg[kWaveSpan / kWave];
  unsigned int preg[kWaveSpan / kWave];
#pragma unroll
  for (int c = 0; c < kWaveSpan / kWave; ++c) {
    const long long i = wbase + (long long)c * kWave + lane;
    kreg[c] = i < n ? keys[i] : 0ull;
    preg[c] = i < n ? pay[i] : 0u;
  }
#pragma unroll
  for (int c = 0; c < kWaveSpan / kWave; ++c) {
    const long long i = wbase + (long long)c * kWave + lane;
    if (i < n) atomicAdd(&wh[wave][(unsigned)(kreg[c] >> shift) & 255u], 1u);
  }
  __syncthreads();
  {                                                    // thread d: start position of every wave for digit d
    const int d = threadIdx.x;
    unsigned int run = offs[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
      const unsigned int cnt = wh[w][d];
      wh[w][d] = run;
      run += cnt;
    }
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int cc = 0; cc < kWaveSpan / kWave; ++cc) {
    const long long i = wbase + (long long)cc * kWave + lane;
    const bool valid = i < n;
    const unsigned long long key = kreg[cc];
    const unsigned int p = preg[cc];
    const unsigned int d = valid ? (unsigned)(key >> shift) & 255u : 0u;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? m : ~m;
    }
    if (valid) {
      const unsigned int rank = (unsigned int)__popcll(same & lt);
      const unsigned int pos = wh[wave][d] + rank;
      keys_out[pos] = key;
      pay_out[pos] = p;
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (same & lt) == 0ull) wh[wave][d] += (unsigned int)__popcll(same);   // first lane of each group
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

struct SortBuffers {
  unsigned long long* keys[2];
  unsigned int* pay[2];
  unsigned int* cnt;
  unsigned int* totals;
};

// Sorts (keys, pay) in place of buffer index *cur; digit positions [lo_bits) of the low word
// and [32, 32 + hi_bits) of the high word.  Returns the buffer index holding the result.
static grb_info radix_sort_pairs(SortBuffers& b, int* cur, long long n, int lo_bits, int hi_bits, hipStream_t s) {
  if (n <= 1) return GRB_SUCCESS;
  const int nblocks = (int)((n + kSortTile - 1) / kSortTile);
  auto pass = [&](int shift) -> grb_info {
    const int src = *cur, dst = 1 - *cur;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(kBlock), 0, s, (const unsigned long long*)b.keys[src], n,
                       shift, nblocks, b.cnt);
    GRB_TRY(exclusive_scan_u32(b.cnt, 256ll * nblocks, b.totals, s));
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblocks), dim3(kBlock), 0, s, (const unsigned long long*)b.keys[src],
                       (const unsigned int*)b.pay[src], n, shift, nblocks, (const unsigned int*)b.cnt, b.keys[dst],
                       b.pay[dst]);
    GRB_HIP_TRY(hipGetLastError());
    *cur = dst;
    return GRB_SUCCESS;
  };
  for (int sft = 0; sft < lo_bits; sft += 8) GRB_TRY(pass(sft));
  for (int sft = 0; sft < hi_bits; sft += 8) GRB_TRY(pass(32 + sft));
  return GRB_SUCCESS;
}

static int bits_for(Index dim) {
  int b = 1;
  while (b < 32 && ((long long)1 << b) < (long long)dim) ++b;
  return b;
}

// ---- coordinate list -> keys, loader options
// A coordinate outside [0, nrows) x [0, ncols) raises *bad (the host then returns
// GrB_INDEX_OUT_OF_BOUNDS before anything is compressed: such a key would index past ptr[]).
__global__ void make_keys_kernel(const Index* __restrict__ major, const Index* __restrict__ minor,
                                 const unsigned int* __restrict__ vals, unsigned int one, long long n, int symmetrize,
                                 Index nrows, Index ncols, unsigned long long* __restrict__ keys,
                                 unsigned int* __restrict__ pay, unsigned int* __restrict__ bad) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long r = (unsigned int)major[i], c = (unsigned int)minor[i];
    if (r >= (unsigned long long)(unsigned int)nrows || c >= (unsigned long long)(unsigned int)ncols) *bad = 1u;
    const unsigned int v = vals ? vals[i] : one;
    keys[i] = (r << 32) | c;
    pay[i] = v;
    if (symmetrize) {                                   // the reverse of every off-diagonal entry; the slot of
      keys[n + i] = r == c ? ~0ull : (c << 32) | r;     // a diagonal one holds a tombstone the compaction drops
      pay[n + i] = v;
    }
  }
}

// keep[i] = 1 for entries that survive: not a self loop (if asked), not equal to the previous
// key (if asked; input is sorted, the first of a run wins)
__global__ void keep_flags_kernel(const unsigned long long* __restrict__ keys, long long n, int drop_loops,
                                  int drop_dups, unsigned int* __restrict__ keep) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = keys[i];
    bool ok = k != ~0ull;
    if (drop_loops && (unsigned int)(k >> 32) == (unsigned int)k) ok = false;
    if (drop_dups && i > 0 && keys[i - 1] == k) ok = false;
    keep[i] = ok ? 1u : 0u;
  }
}

__global__ void compact_pairs_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ pay,
                                     long long n, const unsigned int* __restrict__ pos, const unsigned int* __restrict__ keep_next,
                                     unsigned long long* __restrict__ keys_out, unsigned int* __restrict__ pay_out) {
  // pos = exclusive scan of the keep flags; an entry is kept iff pos[i + 1] != pos[i]
  (void)keep_next;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned int p0 = pos[i], p1 = pos[i + 1];
    if (p1 != p0) { keys_out[p0] = keys[i]; pay_out[p0] = pay[i]; }
  }
}

// sorted keys -> compressed arrays
__global__ void split_sorted_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ pay,
                                    long long n, Index nmajor, Index* __restrict__ ptr, Index* __restrict__ ind,
                                    unsigned int* __restrict__ val) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    long long cur = i < n ? (long long)(keys[i] >> 32) : (long long)nmajor;
    long long prev = i > 0 ? (long long)(keys[i - 1] >> 32) : -1;
    if (cur > nmajor) cur = nmajor;                       // defence in depth: inputs are range-checked upstream
    if (prev > nmajor) prev = nmajor;
    for (long long r = prev + 1; r <= cur; ++r) ptr[r] = (Index)i;
    if (i < n) { ind[i] = (Index)(unsigned int)keys[i]; val[i] = pay[i]; }
  }
}

__global__ void swap_halves_kernel(unsigned long long* __restrict__ keys, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = keys[i];
    keys[i] = (k << 32) | (k >> 32);
  }
}

// ---- column ranking for the SpMV hub packing (spmv.hip): counts, stable sort by descending count
__global__ void rank_counts_from_ptr_kernel(const Index* __restrict__ ptr, Index m, unsigned int* __restrict__ cnt) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index c = (Index)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += stride) cnt[c] = (unsigned int)(ptr[c + 1] - ptr[c]);
}
// no transposed pointer array at hand (a row shard): a plain histogram.  The hub columns' increments are
// same-address atomics; this is the one-off preparation of a shard, not a path any iteration takes
__global__ void rank_counts_hist_kernel(const Index* __restrict__ ind, Index nvals, unsigned int* __restrict__ cnt) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index p = (Index)blockIdx.x * blockDim.x + threadIdx.x; p < nvals; p += stride) atomicAdd(&cnt[ind[p]], 1u);
}
__global__ void rank_keys_kernel(const unsigned int* __restrict__ cnt, Index m, unsigned long long* __restrict__ keys,
                                 unsigned int* __restrict__ pay) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index c = (Index)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += stride) {
    keys[c] = ((unsigned long long)(0xffffffffu - cnt[c]) << 32) | (unsigned int)c;   // descending count, ties by column
    pay[c] = cnt[c];
  }
}
__global__ void rank_scatter_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ pay, Index m,
                                    Index hot, Index* __restrict__ order, Index* __restrict__ rank,
                                    unsigned long long* __restrict__ out /* [0] hot refs, [1] referenced columns */) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  unsigned long long refs = 0, used = 0;
  for (Index r = (Index)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += stride) {
    const Index c = (Index)(unsigned int)keys[r];
    order[r] = c;
    rank[c] = r;
    if (r < hot) refs += pay[r];
    if (pay[r]) ++used;
  }
  auto add = [](unsigned long long x, unsigned long long y) { return x + y; };
  refs = wave_reduce(refs, add);
  used = wave_reduce(used, add);
  if (lane_id() == 0) {
    if (refs) atomicAdd(&out[0], refs);
    if (used) atomicAdd(&out[1], used);
  }
}

}  // namespace grb

using namespace grb;

// In-place exclusive scan of n unsigned ints on the context stream (scratch for the tile totals allocated here).
grb_info grb::device_exclusive_scan_u32(unsigned int* d, long long n) {
  if (n <= 0) return GRB_SUCCESS;
  unsigned int* totals = nullptr;
  GRB_HIP_TRY(hipMalloc((void**)&totals, 4 * (size_t)(n / kScanTile + 2)));
  const grb_info info = exclusive_scan_u32(d, n, totals, ctx().stream);
  const hipError_t e = hipStreamSynchronize(ctx().stream);
  (void)hipFree(totals);
  if (info != GRB_SUCCESS) return info;
  GRB_HIP_TRY(e);
  return GRB_SUCCESS;
}

// Stable sort of n (64-bit key, 32-bit payload) pairs on the context stream, in place: digit positions [0, lo_bits)
// of the low word and [32, 32 + hi_bits) of the high word (spmv_cband.hpp: entries by (band, column rank)).
grb_info grb::device_sort_pairs(unsigned long long* d_keys, unsigned int* d_pay, long long n, int lo_bits, int hi_bits) {
  if (n <= 1) return GRB_SUCCESS;
  hipStream_t s = ctx().stream;
  const int nblocks = (int)((n + kSortTile - 1) / kSortTile) + 1;
  const size_t cnt_elems = 256 * (size_t)nblocks;
  void* raw = nullptr;
  GRB_HIP_TRY(hipMalloc(&raw, 12 * (size_t)n + 4 * cnt_elems + 4 * (cnt_elems / kScanTile + 2) + 64));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{raw};
  SortBuffers b;
  char* q = (char*)raw;
  b.keys[0] = d_keys;
  b.keys[1] = (unsigned long long*)q; q += 8 * (size_t)n;
  b.pay[0] = d_pay;
  b.pay[1] = (unsigned int*)q; q += 4 * (size_t)n;
  b.cnt = (unsigned int*)q; q += 4 * cnt_elems;
  b.totals = (unsigned int*)q;
  int cur = 0;
  GRB_TRY(radix_sort_pairs(b, &cur, n, lo_bits, hi_bits, s));
  if (cur != 0) {
    GRB_HIP_TRY(hipMemcpyAsync(d_keys, b.keys[1], 8 * (size_t)n, hipMemcpyDeviceToDevice, s));
    GRB_HIP_TRY(hipMemcpyAsync(d_pay, b.pay[1], 4 * (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  GRB_HIP_TRY(hipStreamSynchronize(s));
  return GRB_SUCCESS;
}

// The same over the digit positions [first_bit, first_bit + nbits) only (whatever sits below first_bit rides along
// unsorted: spmv_cband.hpp keeps an entry's row-in-band there).
grb_info grb::device_sort_pairs_range(unsigned long long* d_keys, unsigned int* d_pay, long long n, int first_bit, int nbits) {
  if (n <= 1 || nbits <= 0) return GRB_SUCCESS;
  hipStream_t s = ctx().stream;
  const int nblocks = (int)((n + kSortTile - 1) / kSortTile) + 1;
  const size_t cnt_elems = 256 * (size_t)nblocks;
  void* raw = nullptr;
  GRB_HIP_TRY(hipMalloc(&raw, 12 * (size_t)n + 4 * cnt_elems + 4 * (cnt_elems / kScanTile + 2) + 64));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{raw};
  SortBuffers b;
  char* q = (char*)raw;
  b.keys[0] = d_keys;
  b.keys[1] = (unsigned long long*)q; q += 8 * (size_t)n;
  b.pay[0] = d_pay;
  b.pay[1] = (unsigned int*)q; q += 4 * (size_t)n;
  b.cnt = (unsigned int*)q; q += 4 * cnt_elems;
  b.totals = (unsigned int*)q;
  int cur = 0;
  const int nb = (int)((n + kSortTile - 1) / kSortTile);
  for (int sft = first_bit; sft < first_bit + nbits; sft += 8) {
    const int src = cur, dst = 1 - cur;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(kBlock), 0, s, (const unsigned long long*)b.keys[src], n, sft, nb, b.cnt);
    GRB_TRY(exclusive_scan_u32(b.cnt, 256ll * nb, b.totals, s));
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(kBlock), 0, s, (const unsigned long long*)b.keys[src],
                       (const unsigned int*)b.pay[src], n, sft, nb, (const unsigned int*)b.cnt, b.keys[dst], b.pay[dst]);
    GRB_HIP_TRY(hipGetLastError());
    cur = dst;
  }
  if (cur != 0) {
    GRB_HIP_TRY(hipMemcpyAsync(d_keys, b.keys[1], 8 * (size_t)n, hipMemcpyDeviceToDevice, s));
    GRB_HIP_TRY(hipMemcpyAsync(d_pay, b.pay[1], 4 * (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  GRB_HIP_TRY(hipStreamSynchronize(s));
  return GRB_SUCCESS;
}

// Columns ranked by reference count, entirely on the device: counts from the transposed orientation's pointer
// array when there is one (no histogram at all), stable LSD radix sort of (0xffffffff - count, column).
grb_info grb::device_rank_columns(const Index* d_ind, Index nvals, const Index* d_other_ptr, Index m, Index hot,
                                  Index* d_order, Index* d_rank, long long* hot_refs, Index* nreferenced) {
  hipStream_t s = ctx().stream;
  const size_t cap = (size_t)(m > 0 ? m : 1);
  const int nblocks = (int)((m + kSortTile - 1) / kSortTile) + 1;
  const size_t cnt_elems = 256 * (size_t)nblocks > cap + 1 ? 256 * (size_t)nblocks : cap + 1;
  void* raw = nullptr;
  GRB_HIP_TRY(hipMalloc(&raw, 2 * 8 * cap + 2 * 4 * cap + 4 * cap + 4 * cnt_elems + 4 * (cnt_elems / kScanTile + 2) + 64));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{raw};
  SortBuffers b;
  char* q = (char*)raw;
  b.keys[0] = (unsigned long long*)q; q += 8 * cap;
  b.keys[1] = (unsigned long long*)q; q += 8 * cap;
  b.pay[0] = (unsigned int*)q; q += 4 * cap;
  b.pay[1] = (unsigned int*)q; q += 4 * cap;
  unsigned int* d_cnt = (unsigned int*)q; q += 4 * cap;
  b.cnt = (unsigned int*)q; q += 4 * cnt_elems;
  b.totals = (unsigned int*)q; q += 4 * (cnt_elems / kScanTile + 2);
  q = (char*)(((uintptr_t)q + 7) & ~(uintptr_t)7);
  unsigned long long* d_out = (unsigned long long*)q;
  if (d_other_ptr) {
    hipLaunchKernelGGL(rank_counts_from_ptr_kernel, dim3(stream_grid(m, kBlock)), dim3(kBlock), 0, s, d_other_ptr, m, d_cnt);
  } else {
    GRB_HIP_TRY(hipMemsetAsync(d_cnt, 0, 4 * cap, s));
    hipLaunchKernelGGL(rank_counts_hist_kernel, dim3(stream_grid(nvals, kBlock)), dim3(kBlock), 0, s, d_ind, nvals, d_cnt);
  }
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(rank_keys_kernel, dim3(stream_grid(m, kBlock)), dim3(kBlock), 0, s, (const unsigned int*)d_cnt, m,
                     b.keys[0], b.pay[0]);
  GRB_HIP_TRY(hipGetLastError());
  int cur = 0;
  GRB_TRY(radix_sort_pairs(b, &cur, m, 0, 32, s));          // the count digits only: stable, so ties stay in column order
  GRB_HIP_TRY(hipMemsetAsync(d_out, 0, 16, s));
  hipLaunchKernelGGL(rank_scatter_kernel, dim3(stream_grid(m, kBlock)), dim3(kBlock), 0, s,
                     (const unsigned long long*)b.keys[cur], (const unsigned int*)b.pay[cur], m, hot, d_order, d_rank, d_out);
  GRB_HIP_TRY(hipGetLastError());
  unsigned long long out[2] = {0, 0};
  GRB_HIP_TRY(hipMemcpyAsync(out, d_out, 16, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  *hot_refs = (long long)out[0];
  *nreferenced = (Index)out[1];
  return GRB_SUCCESS;
}

// Device coordinate list -> the matrix's device CSR + CSC (owned), host row/column pointers.
// flags: bit 0 add reverse entries, bit 1 drop self loops, bit 2 drop duplicates.
grb_info device_build_from_coo(grb_matrix A, const Index* d_rows, const Index* d_cols, const void* d_vals,
                               long long nvals_in, int flags) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  const bool symmetrize = (flags & 1) != 0;
  if (symmetrize && A->nrows != A->ncols) return GRB_DIMENSION_MISMATCH;   // the reverse of (r, c) must be a valid entry
  const long long n0 = symmetrize ? 2 * nvals_in : nvals_in;
  if (n0 > 0x7fffffffll) return GRB_OUT_OF_MEMORY;      // 32-bit indices throughout, like the reference
  const int nblocks = (int)((n0 + kSortTile - 1) / kSortTile) + 1;
  SortBuffers b;
  void* raw = nullptr;
  const size_t cap = (size_t)(n0 > 0 ? n0 : 1);
  const size_t cnt_elems = 256 * (size_t)nblocks > cap + 1 ? 256 * (size_t)nblocks : cap + 1;
  const size_t bytes = 2 * 8 * cap + 2 * 4 * cap + 4 * cnt_elems + 4 * (cnt_elems / kScanTile + 2) + 8;
  GRB_HIP_TRY(hipMalloc(&raw, bytes));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } guard{raw};
  char* q = (char*)raw;
  b.keys[0] = (unsigned long long*)q; q += 8 * cap;
  b.keys[1] = (unsigned long long*)q; q += 8 * cap;
  b.pay[0] = (unsigned int*)q; q += 4 * cap;
  b.pay[1] = (unsigned int*)q; q += 4 * cap;
  b.cnt = (unsigned int*)q; q += 4 * cnt_elems;
  b.totals = (unsigned int*)q; q += 4 * (cnt_elems / kScanTile + 2);
  unsigned int* d_bad = (unsigned int*)q;

  long long n = n0;
  int cur = 0;
  const unsigned int one = A->dtype == GRB_F32 ? 0x3f800000u : 1u;
  if (nvals_in > 0) {
    GRB_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, s));
    hipLaunchKernelGGL(make_keys_kernel, dim3(stream_grid(nvals_in, kBlock)), dim3(kBlock), 0, s, d_rows, d_cols,
                       (const unsigned int*)d_vals, one, nvals_in, symmetrize ? 1 : 0, A->nrows, A->ncols, b.keys[0],
                       b.pay[0], d_bad);
    GRB_HIP_TRY(hipGetLastError());
    unsigned int bad = 0;
    GRB_HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    if (bad) return GRB_INDEX_OUT_OF_BOUNDS;
  }
  const int rbits = bits_for(A->nrows), cbits = bits_for(A->ncols);
  GRB_TRY(radix_sort_pairs(b, &cur, n, cbits, rbits, s));
  if ((flags & 7) && n > 0) {
    hipLaunchKernelGGL(keep_flags_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s,
                       (const unsigned long long*)b.keys[cur], n, (flags & 2) ? 1 : 0, (flags & 4) ? 1 : 0, b.cnt);
    GRB_HIP_TRY(hipMemsetAsync(b.cnt + n, 0, 4, s));
    GRB_TRY(exclusive_scan_u32(b.cnt, n + 1, b.totals, s));
    hipLaunchKernelGGL(compact_pairs_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s,
                       (const unsigned long long*)b.keys[cur], (const unsigned int*)b.pay[cur], n,
                       (const unsigned int*)b.cnt, (const unsigned int*)nullptr, b.keys[1 - cur], b.pay[1 - cur]);
    GRB_HIP_TRY(hipGetLastError());
    unsigned int kept = 0;
    GRB_HIP_TRY(hipMemcpyAsync(&kept, b.cnt + n, 4, hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    n = kept;
    cur = 1 - cur;
  }

  // ---- CSR
  A->nvals = (Index)n;
  A->owned = true;
  const size_t vcap = n > 0 ? (size_t)n : 1;
  for (CsrArrays* m : {&A->csr, &A->csc}) {
    const Index dim = (m == &A->csr) ? A->nrows : A->ncols;
    GRB_HIP_TRY(hipMalloc((void**)&m->ptr, 4 * ((size_t)dim + 1)));
    GRB_HIP_TRY(hipMalloc((void**)&m->ind, 4 * vcap));
    GRB_HIP_TRY(hipMalloc(&m->val, 4 * vcap));
    m->n = dim;
    m->nvals = (Index)n;
  }
  hipLaunchKernelGGL(split_sorted_kernel, dim3(stream_grid(n + 1, kBlock)), dim3(kBlock), 0, s,
                     (const unsigned long long*)b.keys[cur], (const unsigned int*)b.pay[cur], n, A->nrows, A->csr.ptr,
                     A->csr.ind, (unsigned int*)A->csr.val);
  GRB_HIP_TRY(hipGetLastError());
  // ---- CSC: (col << 32 | row), stable sort on the column digits only
  if (n > 0) {
    hipLaunchKernelGGL(swap_halves_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, b.keys[cur], n);
    GRB_HIP_TRY(hipGetLastError());
  }
  GRB_TRY(radix_sort_pairs(b, &cur, n, 0, cbits, s));
  hipLaunchKernelGGL(split_sorted_kernel, dim3(stream_grid(n + 1, kBlock)), dim3(kBlock), 0, s,
                     (const unsigned long long*)b.keys[cur], (const unsigned int*)b.pay[cur], n, A->ncols, A->csc.ptr,
                     A->csc.ind, (unsigned int*)A->csc.val);
  GRB_HIP_TRY(hipGetLastError());

  // ---- host side: the pointer arrays now (the SpMV plans are cut from them), indices / values on demand
  A->h_csr_ptr.resize((size_t)A->nrows + 1);
  A->h_csc_ptr.resize((size_t)A->ncols + 1);
  GRB_HIP_TRY(hipMemcpyAsync(A->h_csr_ptr.data(), A->csr.ptr, 4 * ((size_t)A->nrows + 1), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(A->h_csc_ptr.data(), A->csc.ptr, 4 * ((size_t)A->ncols + 1), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  A->h_csr_ind.clear(); A->h_csr_val.clear(); A->h_csc_ind.clear(); A->h_csc_val.clear();
  return GRB_SUCCESS;
}
