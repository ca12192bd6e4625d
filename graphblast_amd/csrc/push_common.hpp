// push_common.hpp -- load-balanced frontier expansion shared by the generic SpMSpV
// (spmspv.hip) and the fused BFS (bfs_fused.hip).
//
//   push_degree_kernel      per frontier entry: degree, row start, tile-local exclusive scan
//   push_scan_tiles_kernel  one workgroup scans the tile sums; total stays on the device
//   push_partition_kernel   one thread per chunk boundary: which frontier entry owns the
//                           first edge of every kEdgeChunk-edge chunk (all binary searches in
//                           flight at once instead of one per workgroup, serially)
//   lb_expand_kernel<V>     one chunk of the EXPANDED edge space per workgroup iteration:
//                           stage the chunk's frontier range (scan values + row starts) in
//                           LDS, build the edge -> owner map with a scatter + max-scan (no
//                           per-edge search), then in three passes with all loads of a pass in
//                           flight together: gather the neighbour ids (coalesced along rows),
//                           visitor.peek(dst) (read-only filter, e.g. the visited bitmap),
//                           visitor.visit(k, p, dst) for the survivors.
//                           k = frontier position, p = position in (ind, val), dst = ind[p].
//
// No kernel here funnels atomics through a single address: on MI355X same-address
// atomics serialise at ~12 ns each, which at 10^5..10^6 wave-level appends per level is
// the whole budget of a BFS.
#pragma once
#include "common.hpp"

namespace grb {

constexpr int kDegItems = 4;
constexpr int kDegTile = kBlock * kDegItems;        // frontier entries per scan tile
constexpr int kEdgeItems = 8;
constexpr int kEdgeChunk = kBlock * kEdgeItems;     // expanded edges per chunk (2048)
constexpr int kSegMax = kEdgeChunk;                 // frontier entries staged per chunk

// ---- degrees + row starts + tile-local exclusive scan
static __global__ void push_degree_kernel(const Index* __restrict__ ptr, const Index* __restrict__ u_ind, Index nf,
                                          int* __restrict__ local_scan, Index* __restrict__ row_start,
                                          int* __restrict__ tile_sums, int* __restrict__ tile_off,
                                          int* __restrict__ total_out) {
  __shared__ int smem[kWavesPerBlock];
  const Index base = (Index)blockIdx.x * kDegTile + threadIdx.x * kDegItems;
  int d[kDegItems];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kDegItems; ++k) {
    Index i = base + k;
    d[k] = 0;
    if (i < nf) {
      Index r = u_ind[i];
      Index s = ptr[r];
      d[k] = ptr[r + 1] - s;
      row_start[i] = s;
    }
    sum += d[k];
  }
  int tot;
  int off = block_exclusive_scan(sum, smem, tot);
#pragma unroll
  for (int k = 0; k < kDegItems; ++k) {
    Index i = base + k;
    if (i < nf) local_scan[i] = off;
    off += d[k];
  }
  if (threadIdx.x == 0) {
    tile_sums[blockIdx.x] = tot;
    if (gridDim.x == 1) {            // one tile: the scan of the tile sums is trivial, skip that launch
      tile_off[0] = 0;
      tile_off[1] = tot;
      *total_out = tot;
    }
  }
}

// Exclusive scan of `ntiles` counts by ONE workgroup; offsets[ntiles] = total = *total_out.
static __global__ void push_scan_tiles_kernel(const int* __restrict__ counts, int ntiles, int* __restrict__ offsets,
                                              int* __restrict__ total_out) {
  __shared__ int smem[kWavesPerBlock];
  int carry = 0;
  for (int base = 0; base < ntiles; base += kBlock) {
    int i = base + threadIdx.x;
    int v = i < ntiles ? counts[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, smem, tot);
    if (i < ntiles) offsets[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) { offsets[ntiles] = carry; *total_out = carry; }
}

// global exclusive scan value of frontier entry k
__device__ inline int gscan(const int* local_scan, const int* tile_off, Index k) {
  return tile_off[k / kDegTile] + local_scan[k];
}

// last frontier entry k in [0, nf) with gscan(k) <= e   (two-level binary search)
__device__ inline Index find_owner(const int* local_scan, const int* tile_off, int ntiles, Index nf, int e) {
  int lo = 0, hi = ntiles;              // last tile with tile_off[t] <= e
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tile_off[mid] <= e) lo = mid; else hi = mid;
  }
  const int t = lo;
  const int rel = e - tile_off[t];
  Index a = (Index)t * kDegTile;
  Index b = a + kDegTile < nf ? a + kDegTile : nf;   // entries [a, b)
  Index l = a, h = b;                                 // last k with local_scan[k] <= rel
  while (h - l > 1) {
    Index mid = (l + h) >> 1;
    if (local_scan[mid] <= rel) l = mid; else h = mid;
  }
  return l;
}

// chunk_owner[c] = owner of edge min(c * kEdgeChunk, total - 1), c = 0 .. nchunks (inclusive)
static __global__ void push_partition_kernel(const int* __restrict__ local_scan, const int* __restrict__ tile_off,
                                             int ntiles, Index nf, Index* __restrict__ chunk_owner) {
  const int total = tile_off[ntiles];
  if (total <= 0) return;
  const int nchunks = (total + kEdgeChunk - 1) / kEdgeChunk;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c <= nchunks; c += gridDim.x * blockDim.x) {
    long long e = (long long)c * kEdgeChunk;
    if (e > total - 1) e = total - 1;
    chunk_owner[c] = find_owner(local_scan, tile_off, ntiles, nf, (int)e);
  }
}

template <typename V>
static __global__ __launch_bounds__(kBlock) void lb_expand_kernel(
    const Index* __restrict__ ind, Index nf, const int* __restrict__ local_scan,
    const Index* __restrict__ row_start, const int* __restrict__ tile_off, int ntiles,
    const Index* __restrict__ chunk_owner, V visitor) {
  __shared__ int seg[kSegMax];          // scan value of the staged frontier entries
  __shared__ Index rs[kSegMax];         // their row starts
  __shared__ int own[kEdgeChunk];       // edge (relative to the chunk) -> staged entry
  __shared__ int wmax[kWavesPerBlock];
  const int tid = threadIdx.x, lane = lane_id(), wid = wave_id();
  const int total = tile_off[ntiles];
  const int nchunks = (total + kEdgeChunk - 1) / kEdgeChunk;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int e0 = c * kEdgeChunk;
    const int e1 = (e0 + kEdgeChunk < total) ? e0 + kEdgeChunk : total;
    const Index k0 = chunk_owner[c];
    Index k1 = chunk_owner[c + 1];
    // chunk_owner[c + 1] owns the first edge of the NEXT chunk; entries after the owner of
    // our last edge own nothing here, so the range [k0, k1] is a safe superset
    const int count = k1 - k0 + 1;
    const bool staged = count <= kSegMax;
    Index dst[kEdgeItems], pos[kEdgeItems], kk[kEdgeItems];
    if (staged) {
      for (int i = tid; i < count; i += kBlock) {
        seg[i] = gscan(local_scan, tile_off, k0 + i);
        rs[i] = row_start[k0 + i];
      }
#pragma unroll
      for (int j = 0; j < kEdgeItems; ++j) own[tid * kEdgeItems + j] = 0;
      __syncthreads();
      // scatter: entry i starts at relative edge seg[i] - e0 (clamped); max keeps the last
      // entry among equal starts, i.e. the one that actually owns edges
      for (int i = tid; i < count; i += kBlock) {
        int rel = seg[i] - e0;
        if (rel < 0) rel = 0;
        if (rel < kEdgeChunk) atomicMax(&own[rel], i);
      }
      __syncthreads();
      // inclusive max-scan over the 2048 slots: 8 consecutive slots per thread
      int run = 0;
      int loc[kEdgeItems];
#pragma unroll
      for (int j = 0; j < kEdgeItems; ++j) {
        int x = own[tid * kEdgeItems + j];
        run = x > run ? x : run;
        loc[j] = run;
      }
      int x = run;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        int y = __shfl_up(x, o, kWave);
        if (lane >= o) x = y > x ? y : x;
      }
      if (lane == kWave - 1) wmax[wid] = x;
      int prev = __shfl_up(x, 1, kWave);       // max of the lanes before this one
      if (lane == 0) prev = 0;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w)
        if (w < wid) prev = wmax[w] > prev ? wmax[w] : prev;
#pragma unroll
      for (int j = 0; j < kEdgeItems; ++j) own[tid * kEdgeItems + j] = loc[j] > prev ? loc[j] : prev;
      __syncthreads();
      // pass A: neighbour ids, all in flight
#pragma unroll
      for (int j = 0; j < kEdgeItems; ++j) {
        const int e = e0 + j * kBlock + tid;
        dst[j] = -1;
        if (e < e1) {
          const int l = own[e - e0];
          kk[j] = k0 + l;
          pos[j] = rs[l] + (e - seg[l]);
          dst[j] = ind[pos[j]];
        }
      }
    } else {
      // degenerate range (long runs of zero-degree frontier entries): per-edge search
#pragma unroll
      for (int j = 0; j < kEdgeItems; ++j) {
        const int e = e0 + j * kBlock + tid;
        dst[j] = -1;
        if (e < e1) {
          const Index k = find_owner(local_scan, tile_off, ntiles, nf, e);
          kk[j] = k;
          pos[j] = row_start[k] + (e - gscan(local_scan, tile_off, k));
          dst[j] = ind[pos[j]];
        }
      }
    }
    // pass B: read-only filter, all in flight
    bool need[kEdgeItems];
#pragma unroll
    for (int j = 0; j < kEdgeItems; ++j) need[j] = dst[j] >= 0 && visitor.peek(dst[j]);
    // pass C: the survivors
#pragma unroll
    for (int j = 0; j < kEdgeItems; ++j)
      if (need[j]) visitor.visit(kk[j], pos[j], dst[j]);
    __syncthreads();
  }
}

// Launch helpers. lb_prepare: degree + scan (frontier out-edge total -> *d_total).
// lb_run: partition + expand. Scratch: local_scan[nf], row_start[nf],
// tile_sums/tile_off[2*ntiles+2], chunk_owner[max_chunks+2].
static inline grb_info lb_prepare(hipStream_t s, const CsrArrays& M, const Index* u_ind, Index nf, int* local_scan,
                                  Index* row_start, int* tile_sums, int* tile_off, int* d_total) {
  const int ntiles = ceil_div(nf, kDegTile);
  hipLaunchKernelGGL(push_degree_kernel, dim3(ntiles), dim3(kBlock), 0, s, M.ptr, u_ind, nf, local_scan, row_start,
                     tile_sums, tile_off, d_total);
  GRB_HIP_TRY(hipGetLastError());
  if (ntiles > 1) {
    hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, tile_sums, ntiles, tile_off, d_total);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}

template <typename V>
static inline grb_info lb_run(hipStream_t s, const CsrArrays& M, Index nf, long long max_edges, int* local_scan,
                              Index* row_start, int* tile_off, Index* chunk_owner, V visitor) {
  const int ntiles = ceil_div(nf, kDegTile);
  const long long max_chunks = (max_edges + kEdgeChunk - 1) / kEdgeChunk + 1;
  int pgrid = (int)((max_chunks + kBlock - 1) / kBlock);
  if (pgrid < 1) pgrid = 1;
  if (pgrid > 1024) pgrid = 1024;
  hipLaunchKernelGGL(push_partition_kernel, dim3(pgrid), dim3(kBlock), 0, s, local_scan, tile_off, ntiles, nf,
                     chunk_owner);
  GRB_HIP_TRY(hipGetLastError());
  long long egrid = max_chunks < 2048 ? max_chunks : 2048;
  if (egrid < 1) egrid = 1;
  hipLaunchKernelGGL((lb_expand_kernel<V>), dim3((int)egrid), dim3(kBlock), 0, s, M.ind, nf, local_scan, row_start,
                     tile_off, ntiles, chunk_owner, visitor);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

template <typename V>
static inline grb_info launch_lb_expand(hipStream_t s, const CsrArrays& M, const Index* u_ind, Index nf,
                                        long long max_edges, int* local_scan, Index* row_start, int* tile_sums,
                                        int* tile_off, Index* chunk_owner, int* d_total, V visitor) {
  GRB_TRY(lb_prepare(s, M, u_ind, nf, local_scan, row_start, tile_sums, tile_off, d_total));
  return lb_run(s, M, nf, max_edges, local_scan, row_start, tile_off, chunk_owner, visitor);
}

// ---- ordered bitmap compaction: indices of the bits set in (a & ~b), one word per thread.
// b may be null (then: bits set in a). Three steps, no atomics:
//   bitmap_count_kernel -> push_scan_tiles_kernel -> bitmap_list_kernel
static __global__ void bitmap_count_kernel(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b,
                                           int nwords, int* __restrict__ tile_counts) {
  __shared__ int smem[kWavesPerBlock];
  int i = blockIdx.x * kBlock + threadIdx.x;
  int c = 0;
  if (i < nwords) c = __popc(b ? (a[i] & ~b[i]) : a[i]);
  c = wave_reduce(c, [](int x, int y) { return x + y; });
  if (lane_id() == 0) smem[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w];
    tile_counts[blockIdx.x] = t;
  }
}

static __global__ void bitmap_list_kernel(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b,
                                          int nwords, const int* __restrict__ tile_off, Index* __restrict__ out) {
  __shared__ int smem[kWavesPerBlock];
  int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned int wd = 0u;
  if (i < nwords) wd = b ? (a[i] & ~b[i]) : a[i];
  int tot;
  int p = tile_off[blockIdx.x] + block_exclusive_scan(__popc(wd), smem, tot);
  while (wd) {
    int bit = __ffs((int)wd) - 1;
    wd &= wd - 1;
    out[p++] = (Index)i * 32 + bit;
  }
}

}  // namespace grb
