// push_common.hpp -- load-balanced frontier expansion shared by the generic SpMSpV
// (spmspv.hip) and the fused BFS (bfs_fused.hip).
//
//   push_degree_kernel      degree of every frontier entry + tile-local exclusive scan
//   push_scan_tiles_kernel  one workgroup scans the tile sums; total stays on the device
//   lb_expand_kernel<V>     every workgroup takes equal chunks of the EXPANDED edge space,
//                           locates its frontier range by binary search on the scan (staged
//                           in LDS), and calls visitor(k, row, p, dst) per edge, where k is
//                           the frontier position, p the edge's position in (ind, val) and
//                           dst = ind[p]; consecutive lanes read consecutive p of a row.
#pragma once
#include "common.hpp"

namespace grb {

constexpr int kDegItems = 4;
constexpr int kDegTile = kBlock * kDegItems;        // frontier entries per scan tile
constexpr int kEdgeItems = 8;
constexpr int kEdgeChunk = kBlock * kEdgeItems;     // expanded edges per chunk
constexpr int kSegMax = 2048;                       // frontier entries staged per chunk

// ---- 1. degrees + tile-local exclusive scan
static __global__ void push_degree_kernel(const Index* __restrict__ ptr, const Index* __restrict__ u_ind, Index nf,
                                   int* __restrict__ local_scan, int* __restrict__ tile_sums) {
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
      d[k] = ptr[r + 1] - ptr[r];
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
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

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


template <typename V>
static __global__ __launch_bounds__(kBlock) void lb_expand_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, const Index* __restrict__ u_ind, Index nf,
    const int* __restrict__ local_scan, const int* __restrict__ tile_off, int ntiles, V visitor) {
  __shared__ int seg[kSegMax + 1];
  __shared__ Index range[2];
  const int total = tile_off[ntiles];
  for (int e0 = blockIdx.x * kEdgeChunk; e0 < total; e0 += gridDim.x * kEdgeChunk) {
    const int e1 = (e0 + kEdgeChunk < total) ? e0 + kEdgeChunk : total;
    if (threadIdx.x < 2) {
      int e = threadIdx.x == 0 ? e0 : e1 - 1;
      range[threadIdx.x] = find_owner(local_scan, tile_off, ntiles, nf, e);
    }
    __syncthreads();
    const Index k0 = range[0], k1 = range[1];
    const int count = k1 - k0 + 1;
    const bool staged = count <= kSegMax;
    if (staged)
      for (int i = threadIdx.x; i < count; i += kBlock) seg[i] = gscan(local_scan, tile_off, k0 + i);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kEdgeItems; ++j) {
      const int e = e0 + j * kBlock + threadIdx.x;
      if (e < e1) {
        Index k;
        int kstart;
        if (staged) {
          int l = 0, h = count;          // last i with seg[i] <= e
          while (h - l > 1) {
            int mid = (l + h) >> 1;
            if (seg[mid] <= e) l = mid; else h = mid;
          }
          k = k0 + l;
          kstart = seg[l];
        } else {
          k = find_owner(local_scan, tile_off, ntiles, nf, e);
          kstart = gscan(local_scan, tile_off, k);
        }
        const Index row = u_ind[k];
        const Index p = ptr[row] + (e - kstart);
        visitor(k, row, p, ind[p]);
      }
    }
    __syncthreads();
  }
}

}  // namespace grb
