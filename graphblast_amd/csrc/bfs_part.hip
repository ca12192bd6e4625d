// bfs_part.hip -- level steps of the 1-D vertex-partitioned BFS.
//
// One process drives one GPU and owns the vertex range [lo, lo + n_local), lo a multiple
// of 64.  It holds the out-edges (push) and in-edges (pull) of its owned vertices as
// n_local x n_global CSR matrices with GLOBAL column ids, a replica of the visited bitmap
// (n_global bits) and the labels of its owned vertices.  Per level:
//
//   pull   grb_bfs_part_pull   owned unvisited vertices probe their in-neighbours in the
//                              replicated bitmap -> new bits land in the owner's word range
//   push   grb_bfs_part_push   owned frontier vertices expand their out-edges into a work
//                              copy of the bitmap -> new bits anywhere in the bitmap
//   (host) one collective on the n_global/8-byte "new bits" bitmap (graphblast_amd/dist.py:
//          all-gather + OR; 512 KiB per rank at RMAT-22)
//   all    grb_bfs_part_apply  visited |= new_global, owned new vertices get their label,
//                              |new_global| = next frontier size, identical on every rank
//
// The reference has no multi-GPU code; the semantics are those of the
// single-GPU level loop (bfs_fused.hip), whose direction decisions graphblast_amd/dist.py
// repeats identically on every rank from the replicated frontier size.
#include "bfs_kernels.hpp"

namespace grb {

// out = a & ~b
__global__ void bitmap_andnot_kernel(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b,
                                     int nwords, unsigned int* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) out[i] = a[i] & ~b[i];
}

// vis |= fresh ; owned fresh bits get the label ; per-block popcount partials (no atomics)
__global__ void bfs_part_apply_kernel(const unsigned int* __restrict__ fresh, unsigned int* __restrict__ vis, int nwords,
                                      int lo_word, int local_words, Index lo, Index n_local, float* __restrict__ label,
                                      float new_label, int* __restrict__ partial) {
  __shared__ int smem[kWavesPerBlock];
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
    unsigned int f = fresh[i] & ~vis[i];
    if (!f) continue;
    vis[i] |= f;
    cnt += __popc(f);
    if (i >= lo_word && i < lo_word + local_words) {
      while (f) {
        int b = __ffs((int)f) - 1;
        f &= f - 1;
        Index v = (Index)(i - lo_word) * 32 + b;
        if (v < n_local) label[v] = new_label;
      }
    }
  }
  cnt = wave_reduce(cnt, [](int a, int b) { return a + b; });
  if (lane_id() == 0) smem[wave_id()] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w];
    partial[blockIdx.x] = t;
  }
}

__global__ void sum_partials_kernel(const int* __restrict__ partial, int n, int* __restrict__ out) {
  int c = 0;
  for (int i = threadIdx.x; i < n; i += kWave) c += partial[i];
  c = wave_reduce(c, [](int a, int b) { return a + b; });
  if (threadIdx.x == 0) *out = c;
}

// ---- leaner level steps for the host-driven partitioned loop (one launch and one host wake-up each)
// apply in one launch: vis |= fresh, owned labels, and two totals through the granule mailbox --
// |fresh| (identical on every rank) and the out-degree sum of the OWNED fresh vertices (this rank's
// share of the next push, which decides locally which push kernel to use).
__global__ __launch_bounds__(kBlock) void bfs_part_apply2_kernel(
    const unsigned int* __restrict__ fresh, unsigned int* __restrict__ vis, int nwords, int lo_word, int local_words,
    Index n_local, float* __restrict__ label, float new_label, const Index* __restrict__ out_ptr,
    const int* __restrict__ deg_full /* nullable: out-degree of every vertex */,
    unsigned int* __restrict__ partial /* 3 per workgroup */, unsigned int* __restrict__ ticket,
    unsigned long long* __restrict__ mail, int seq) {
  __shared__ unsigned int smem[3][kWavesPerBlock];
  __shared__ int s_last;
  unsigned int cnt = 0, edges = 0, all_edges = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
    unsigned int f = fresh[i] & ~vis[i];
    if (!f) continue;
    vis[i] |= f;
    cnt += __popc(f);
    const bool owned = i >= lo_word && i < lo_word + local_words;
    if (!owned && !deg_full) continue;
    while (f) {
      const int b = __ffs((int)f) - 1;
      f &= f - 1;
      if (deg_full) all_edges += (unsigned int)deg_full[(Index)i * 32 + b];
      if (owned) {
        const Index v = (Index)(i - lo_word) * 32 + b;
        if (v < n_local) {
          label[v] = new_label;
          if (out_ptr) edges += (unsigned int)(out_ptr[v + 1] - out_ptr[v]);
        }
      }
    }
  }
  unsigned int mine[3] = {cnt, edges, all_edges};
#pragma unroll
  for (int k = 0; k < 3; ++k) mine[k] = wave_reduce(mine[k], [](unsigned int a, unsigned int b) { return a + b; });
  if (lane_id() == 0)
    for (int k = 0; k < 3; ++k) smem[k][wave_id()] = mine[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) {
      unsigned int t = 0;
      for (int w = 0; w < kWavesPerBlock; ++w) t += smem[k][w];
      __hip_atomic_store(&partial[3 * blockIdx.x + k], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_workgroup_arrives(ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  unsigned int f[3] = {0, 0, 0};
  for (int j = threadIdx.x; j < (int)gridDim.x; j += kBlock)
    for (int k = 0; k < 3; ++k) f[k] += __hip_atomic_load(&partial[3 * j + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int k = 0; k < 3; ++k) f[k] = wave_reduce(f[k], [](unsigned int a, unsigned int b) { return a + b; });
  __syncthreads();
  if (lane_id() == 0)
    for (int k = 0; k < 3; ++k) smem[k][wave_id()] = f[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) {
      unsigned int t = 0;
      for (int w = 0; w < kWavesPerBlock; ++w) t += smem[k][w];
      __hip_atomic_store(&mail[k], ((unsigned long long)(unsigned int)seq << 32) | t, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// push for a frontier with few out-edges on this rank: a wave per 32 owned frontier slots, lanes over
// the edge list of each set bit, unvisited targets OR-ed into the (pre-zeroed) new-bits bitmap.  One
// launch, nothing read back; the edge-balanced path above is for frontiers with many edges.
__global__ __launch_bounds__(kBlock) void bfs_part_push_small_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n_local, const unsigned int* __restrict__ f_local,
    int local_words, const unsigned int* __restrict__ vis, unsigned int* __restrict__ fresh) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / kWave);
  const int nwaves = (int)((gridDim.x * (unsigned)blockDim.x) / kWave);
  for (int i = wave; i < local_words; i += nwaves) {
    unsigned int word = f_local[i];
    while (word) {
      const int b = __ffs((int)word) - 1;
      word &= word - 1;
      const Index v = (Index)i * 32 + b;
      if (v >= n_local) break;
      const Index pb = ptr[v], pe = ptr[v + 1];
      for (Index p = pb + lane; p < pe; p += kWave * 4) {
        Index u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = p + k * kWave < pe ? ind[p + k * kWave] : -1;
        unsigned int vw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) vw[k] = u[k] >= 0 ? vis[u[k] >> 5] : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (u[k] >= 0 && !((vw[k] >> (u[k] & 31)) & 1u)) atomicOr(&fresh[u[k] >> 5], 1u << (u[k] & 31));
      }
    }
  }
}

__global__ void bfs_part_seed_kernel(unsigned int* vis, unsigned int* fresh, float* label, Index lo, Index n_local,
                                     Index source) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned int bit = 1u << (source & 31);
  vis[source >> 5] = bit;
  fresh[source >> 5] = bit;
  if (source >= lo && source < lo + n_local) label[source - lo] = 1.f;
}

// label == value -> 0: the vertices the level that ended a cut-off loop discovered are never assigned (bfs.hpp:48-66)
__global__ void bfs_part_unlabel_kernel(float* __restrict__ label, Index n, float value) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (label[i] == value) label[i] = 0.f;
}

// out = parts[0] | parts[1] | ... (the all-gathered new-bits bitmaps of every rank)
__global__ void bitmap_or_parts_kernel(const unsigned int* __restrict__ parts, int world, int nwords,
                                       unsigned int* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
    unsigned int x = 0;
    for (int r = 0; r < world; ++r) x |= parts[(size_t)r * nwords + i];
    out[i] = x;
  }
}

}  // namespace grb

using namespace grb;

extern "C" {

grb_info grb_bfs_part_pull(grb_matrix A_in, grb_index lo, grb_index n_global, const uint32_t* d_vis, uint32_t* d_new,
                           float* d_label_local, float new_label) { GRB_API_ENTER();
  if (!A_in || !A_in->built || !A_in->csr.ptr || !d_vis || !d_new) return GRB_UNINITIALIZED_OBJECT;
  if (lo % 64 != 0 || A_in->ncols != n_global) return GRB_INVALID_VALUE;
  hipStream_t s = ctx().stream;
  const Index n_local = A_in->nrows;
  GRB_HIP_TRY(hipMemsetAsync(d_new, 0, 4 * (size_t)(2 * ceil_div(n_global, 64)), s));   // only the owned words get bits
  if (n_local == 0) return GRB_SUCCESS;
  GRB_TRY(ensure_empty_rows(&A_in->d_empty_csr_rows, A_in->csr, s));
  const int grid = stream_grid((long long)ceil_div(n_local, kWave) * kWave, kBlock);
  hipLaunchKernelGGL((bfs_pull_kernel<false>), dim3(grid), dim3(kBlock), 0, s, A_in->csr.ptr, A_in->csr.ind, n_local,
                     d_vis, d_vis + lo / 32, A_in->d_empty_csr_rows, (const Index*)nullptr, d_new + lo / 32, 1, d_label_local, new_label,
                     (unsigned long long*)nullptr);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_push(grb_matrix A_out, grb_index lo, grb_index n_global, const uint32_t* d_frontier,
                           const uint32_t* d_vis, uint32_t* d_work, uint32_t* d_new, int64_t* expanded_edges_out) { GRB_API_ENTER();
  if (!A_out || !A_out->built || !A_out->csr.ptr || !d_frontier || !d_vis || !d_work || !d_new)
    return GRB_UNINITIALIZED_OBJECT;
  if (lo % 64 != 0 || A_out->ncols != n_global) return GRB_INVALID_VALUE;
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n_local = A_out->nrows;
  const int nwords = 2 * ceil_div(n_global, 64);
  const int local_words = 2 * ceil_div(n_local, 64);
  if (expanded_edges_out) *expanded_edges_out = 0;
  if (n_local == 0) { GRB_HIP_TRY(hipMemsetAsync(d_new, 0, 4 * (size_t)nwords, s)); return GRB_SUCCESS; }
  const int btiles = ceil_div(local_words, kBlock);
  const long long max_edges = A_out->nvals;
  const long long max_chunks = max_edges / kEdgeChunk + 2;
  void *p_q, *p_scan, *p_rs, *p_tiles, *p_bt;
  GRB_TRY(scratch(9, 4 * (size_t)n_local + 4, &p_q));
  GRB_TRY(scratch(2, 4 * (size_t)n_local + 4, &p_scan));
  GRB_TRY(scratch(11, 4 * (size_t)n_local + 4, &p_rs));
  const int max_tiles = ceil_div(n_local, kDegTile);
  GRB_TRY(scratch(3, 4 * (size_t)(2 * max_tiles + 2), &p_tiles));
  GRB_TRY(scratch(6, 4 * (size_t)(2 * btiles + 4 + max_chunks + 2), &p_bt));
  Index* queue = (Index*)p_q;
  int* btile_counts = (int*)p_bt;
  int* btile_off = btile_counts + btiles;
  Index* chunk_owner = (Index*)(btile_off + btiles + 2);
  int* tile_sums = (int*)p_tiles;
  int* tile_off = tile_sums + max_tiles;
  int* d_state = c.d_mail + 24;       // [0] local frontier size, [1] expanded edges
  const unsigned int* f_local = d_frontier + lo / 32;
  // owned frontier vertices, as LOCAL row indices, ordered
  hipLaunchKernelGGL(bitmap_count_kernel, dim3(btiles), dim3(kBlock), 0, s, f_local, (const unsigned int*)nullptr,
                     local_words, btile_counts);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, btile_counts, btiles, btile_off, d_state);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(bitmap_list_kernel, dim3(btiles), dim3(kBlock), 0, s, f_local, (const unsigned int*)nullptr,
                     local_words, btile_off, queue);
  GRB_HIP_TRY(hipGetLastError());
  int h[2] = {0, 0};
  GRB_TRY(fetch_ints(d_state, 1, h));
  const Index nf = h[0];
  if (nf == 0) { GRB_HIP_TRY(hipMemsetAsync(d_new, 0, 4 * (size_t)nwords, s)); return GRB_SUCCESS; }
  GRB_HIP_TRY(hipMemcpyAsync(d_work, d_vis, 4 * (size_t)nwords, hipMemcpyDeviceToDevice, s));
  BfsPushVisitor vis_fn{d_work, nullptr, 0.f};
  GRB_TRY(launch_lb_expand(s, A_out->csr, queue, nf, max_edges, (int*)p_scan, (Index*)p_rs, tile_sums, tile_off,
                           chunk_owner, d_state + 1, vis_fn));
  hipLaunchKernelGGL(bitmap_andnot_kernel, dim3(stream_grid(nwords)), dim3(kBlock), 0, s, d_work, d_vis, nwords, d_new);
  GRB_HIP_TRY(hipGetLastError());
  if (expanded_edges_out) {
    GRB_TRY(fetch_ints(d_state + 1, 1, h));
    *expanded_edges_out = h[0];
  }
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_apply(const uint32_t* d_new_global, uint32_t* d_vis, grb_index lo, grb_index n_local,
                            grb_index n_global, float* d_label_local, float new_label, int32_t* discovered_out) { GRB_API_ENTER();
  if (!d_new_global || !d_vis || !discovered_out) return GRB_UNINITIALIZED_OBJECT;
  if (lo % 64 != 0) return GRB_INVALID_VALUE;
  Context& c = ctx();
  hipStream_t s = c.stream;
  const int nwords = 2 * ceil_div(n_global, 64);
  const int local_words = 2 * ceil_div(n_local, 64);
  const int grid = stream_grid(nwords, kBlock);
  void* p;
  GRB_TRY(scratch(0, 4 * (size_t)grid + 16, &p));
  hipLaunchKernelGGL(bfs_part_apply_kernel, dim3(grid), dim3(kBlock), 0, s, d_new_global, d_vis, nwords, lo / 32,
                     local_words, lo, n_local, d_label_local, new_label, (int*)p);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kWave), 0, s, (const int*)p, grid, c.d_mail + 26);
  GRB_HIP_TRY(hipGetLastError());
  int h = 0;
  GRB_TRY(fetch_ints(c.d_mail + 26, 1, &h));
  *discovered_out = h;
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_apply2(const uint32_t* d_new_global, uint32_t* d_vis, grb_index lo, grb_index n_local,
                             grb_index n_global, grb_matrix A_out, const int32_t* d_deg_full, float* d_label_local,
                             float new_label, int32_t* discovered_out, int64_t* local_frontier_edges_out,
                             int64_t* frontier_edges_out) { GRB_API_ENTER();
  if (!d_new_global || !d_vis || !discovered_out) return GRB_UNINITIALIZED_OBJECT;
  if (lo % 64 != 0) return GRB_INVALID_VALUE;
  Context& c = ctx();
  hipStream_t s = c.stream;
  const int nwords = 2 * ceil_div(n_global, 64);
  const int local_words = 2 * ceil_div(n_local, 64);
  const int grid = stream_grid(nwords, kBlock);
  void* p;
  GRB_TRY(scratch(0, 12 * (size_t)grid + 16, &p));
  const int seq = ++c.mail_seq;
  hipLaunchKernelGGL(bfs_part_apply2_kernel, dim3(grid), dim3(kBlock), 0, s, d_new_global, d_vis, nwords, (int)(lo / 32),
                     local_words, n_local, d_label_local, new_label,
                     (A_out && A_out->built) ? (const Index*)A_out->csr.ptr : (const Index*)nullptr, (const int*)d_deg_full,
                     (unsigned int*)p, c.d_tickets, c.d_hgran, seq);
  GRB_HIP_TRY(hipGetLastError());
  unsigned int out[3] = {0, 0, 0};
  GRB_TRY(wait_granules(seq, 3, out));
  *discovered_out = (int32_t)out[0];
  if (local_frontier_edges_out) *local_frontier_edges_out = (int64_t)out[1];
  if (frontier_edges_out) *frontier_edges_out = d_deg_full ? (int64_t)out[2] : -1;
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_push_small(grb_matrix A_out, grb_index lo, grb_index n_global, const uint32_t* d_frontier,
                                 const uint32_t* d_vis, uint32_t* d_new) { GRB_API_ENTER();
  if (!A_out || !A_out->built || !A_out->csr.ptr || !d_frontier || !d_vis || !d_new) return GRB_UNINITIALIZED_OBJECT;
  if (lo % 64 != 0 || A_out->ncols != n_global) return GRB_INVALID_VALUE;
  hipStream_t s = ctx().stream;
  const Index n_local = A_out->nrows;
  const int nwords = 2 * ceil_div(n_global, 64);
  const int local_words = 2 * ceil_div(n_local, 64);
  GRB_HIP_TRY(hipMemsetAsync(d_new, 0, 4 * (size_t)nwords, s));
  if (n_local == 0) return GRB_SUCCESS;
  const int grid = stream_grid((long long)local_words * kWave, kBlock);
  hipLaunchKernelGGL(bfs_part_push_small_kernel, dim3(grid), dim3(kBlock), 0, s, A_out->csr.ptr, A_out->csr.ind, n_local,
                     d_frontier + lo / 32, local_words, d_vis, d_new);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_seed(uint32_t* d_vis, uint32_t* d_new_global, float* d_label_local, grb_index lo,
                           grb_index n_local, grb_index n_global, grb_index source) { GRB_API_ENTER();
  if (!d_vis || !d_new_global || !d_label_local) return GRB_UNINITIALIZED_OBJECT;
  if (source < 0 || source >= n_global) return GRB_INVALID_INDEX;
  hipStream_t s = ctx().stream;
  GRB_TRY(ctx_init());
  const int nwords = 2 * ceil_div(n_global, 64);
  GRB_HIP_TRY(hipMemsetAsync(d_vis, 0, 4 * (size_t)nwords, s));
  GRB_HIP_TRY(hipMemsetAsync(d_new_global, 0, 4 * (size_t)nwords, s));
  if (n_local > 0) GRB_HIP_TRY(hipMemsetAsync(d_label_local, 0, 4 * (size_t)n_local, s));
  hipLaunchKernelGGL(bfs_part_seed_kernel, dim3(1), dim3(kWave), 0, s, d_vis, d_new_global, d_label_local, lo, n_local,
                     source);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_unlabel(float* d_label_local, grb_index n_local, float value) { GRB_API_ENTER();
  if (n_local <= 0) return GRB_SUCCESS;
  if (!d_label_local) return GRB_NULL_POINTER;
  GRB_TRY(ctx_init());
  hipLaunchKernelGGL(bfs_part_unlabel_kernel, dim3(stream_grid(n_local)), dim3(kBlock), 0, ctx().stream, d_label_local,
                     (Index)n_local, value);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

grb_info grb_bitmap_or_parts(const uint32_t* d_parts, int world, grb_index nwords, uint32_t* d_out) { GRB_API_ENTER();
  if (!d_parts || !d_out || world < 1) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(ctx_init());
  hipLaunchKernelGGL(bitmap_or_parts_kernel, dim3(stream_grid(nwords)), dim3(kBlock), 0, ctx().stream, d_parts, world,
                     (int)nwords, d_out);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

// sum of out-degree over labelled owned vertices + their count (TEPS numerator, local part)
grb_info grb_bfs_part_tally(grb_matrix A_out, const float* d_label_local, int64_t* edges_out, int32_t* reached_out) { GRB_API_ENTER();
  if (!A_out || !A_out->built || !d_label_local) return GRB_UNINITIALIZED_OBJECT;
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n_local = A_out->nrows;
  unsigned long long h_tally[64] = {0};
  if (n_local > 0) {
    void* p_tally;
    GRB_TRY(scratch(10, 64 * sizeof(unsigned long long), &p_tally));
    GRB_HIP_TRY(hipMemsetAsync(p_tally, 0, 64 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(bfs_tally_kernel, dim3(stream_grid(n_local, kBlock * 8)), dim3(kBlock), 0, s, d_label_local,
                       A_out->csr.ptr, n_local, (unsigned long long*)p_tally);
    GRB_HIP_TRY(hipGetLastError());
    GRB_HIP_TRY(hipMemcpyAsync(h_tally, p_tally, sizeof(h_tally), hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
  }
  unsigned long long e = 0, r = 0;
  for (int i = 0; i < 32; ++i) { e += h_tally[2 * i]; r += h_tally[2 * i + 1]; }
  if (edges_out) *edges_out = (int64_t)e;
  if (reached_out) *reached_out = (int32_t)r;
  return GRB_SUCCESS;
}

}  // extern "C"
