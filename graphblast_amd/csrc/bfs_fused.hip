// bfs_fused.hip -- direction-optimised BFS with a device-resident level loop.
//
// Same algorithm and same push/pull decisions as graphblas/algorithm/bfs.hpp:14-89 run
// through vxm (backend/cuda/operations.hpp:80-209 + vector.hpp:291-323 `convert`), but
// the per-level GraphBLAS call sequence  assign -> vxm(convert, spmspv|spmv) -> swap ->
// reduce  is collapsed to one expansion per level on a representation chosen for
// MI355X:
//
//   visited set     bitmap, n/8 bytes (512 KiB at RMAT-22: lives in every XCD's L2), not
//                   the reference's n x 4-byte float vector used as a 1-bit flag; a second
//                   per-matrix bitmap marks vertices without in-edges (never discoverable),
//                   so pull levels skip them without touching the matrix
//   frontier        never stored as a dense vector: the vertices a level discovered are
//                   exactly  visited_now & ~visited_before ; an ordered, atomic-free bitmap
//                   compaction (count / scan / list) turns that into the sorted index queue
//                   of the next push level, and its scan total is the frontier size the host
//                   needs for the direction decision
//   push level      degree scan + chunk partition + edge-balanced expansion
//                   (push_common.hpp); atomicOr on the bitmap deduplicates, the winner
//                   writes the label.  No queue append, hence no same-address atomics
//   pull level      one wave per 64 vertices = 2 bitmap words; lanes probe their own
//                   in-neighbour list serially (early exit), leftovers are finished by the
//                   whole wave with coalesced reads + ballot; an unvisited vertex is
//                   discoverable iff ANY in-neighbour is visited as of the start of the level;
//                   visited is double-buffered (read `in`, write `out` = in | new): no atomics
//   level control   one small D2H mailbox read per level (the reference synchronises after
//                   every runtime call and returns three scan totals per level)
//
// Labels are the reference's: v[i] = level at which i was discovered, source = 1,
// unreachable = 0, float32 (bit-exact vs SimpleReferenceBfs, test_bfs.hpp:11-61).
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
    if (!(old & bit)) label[dst] = new_label;
  }
};

__device__ inline bool bit_set(const unsigned int* __restrict__ bm, Index v) {
  return (bm[v >> 5] >> (v & 31)) & 1u;
}

template <bool kCountInspected>
__global__ __launch_bounds__(kBlock) void bfs_pull_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n,
    const unsigned int* __restrict__ vin, const unsigned int* __restrict__ skip,
    unsigned int* __restrict__ vout, float* __restrict__ label, float new_label,
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
    const unsigned int word = vin[(chunk << 1) + (lane >> 5)];
    // `skip` marks vertices without in-edges: never discoverable, but NOT visited (they may
    // still be somebody's in-neighbour, so they must not look visited to the hit test)
    const bool was = ((word | skip[(chunk << 1) + (lane >> 5)]) >> (lane & 31)) & 1u;
    bool active = (v < n) && !was;
    unsigned long long act_mask = __ballot(active);
    if (act_mask == 0ull) {                       // whole chunk already visited / unreachable
      if (lane == 0) vout[chunk << 1] = word;
      if (lane == 32) vout[(chunk << 1) + 1] = word;
      continue;
    }
    Index p = 0, e = 0;
    bool found = false;
    if (active) {
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
    if (lane == 0) vout[chunk << 1] = word | (unsigned int)(fb & 0xffffffffull);
    if (lane == 32) vout[(chunk << 1) + 1] = word | (unsigned int)(fb >> 32);
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
__global__ void bfs_empty_rows_kernel(const Index* __restrict__ ptr, Index n, int nwords,
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

__global__ void bfs_seed_kernel(unsigned int* visited, float* label, Index* queue, Index source) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    visited[source >> 5] |= 1u << (source & 31);
    label[source] = 1.f;
    queue[0] = source;
  }
}

// labels == bad -> 0 (frontier discovered by the last allowed iteration is never assigned
// by the reference loop, bfs.hpp:48-66)
__global__ void bfs_unlabel_kernel(float* __restrict__ label, Index n, float bad) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (label[i] == bad) label[i] = 0.f;
}

// TEPS numerator: sum of out-degree over labelled vertices, and their count.
// Partials per workgroup, spread over 32 slots to keep atomics off a single address.
__global__ void bfs_tally_kernel(const float* __restrict__ label, const Index* __restrict__ ptr, Index n,
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

}  // namespace grb

using namespace grb;

extern "C" grb_info grb_bfs_fused(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                                  grb_bfs_result* result, grb_bfs_level* levels_out, int max_levels, int profile) {
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built || !A->csr.ptr || !A->csc.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
  if (A->nrows != A->ncols || v->nsize != A->nrows) return GRB_DIMENSION_MISMATCH;
  if (source < 0 || source >= A->nrows) return GRB_INVALID_INDEX;
  GRB_TRY(ctx_init());
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n = A->nrows;
  const int nwords = 2 * ceil_div(n, 64);            // two words per 64-vertex chunk
  const int btiles = ceil_div(nwords, kBlock);
  const int mode = desc->desc[GRB_MXVMODE];
  const long long max_edges = A->nvals;
  const long long max_chunks = max_edges / kEdgeChunk + 2;

  // vertices without in-edges, cached per matrix
  if (!A->d_no_in_edges) {
    GRB_HIP_TRY(hipMalloc((void**)&A->d_no_in_edges, 4 * (size_t)nwords));
    hipLaunchKernelGGL(bfs_empty_rows_kernel, dim3(stream_grid(nwords)), dim3(kBlock), 0, s, A->csc.ptr, n, nwords,
                       A->d_no_in_edges);
    GRB_HIP_TRY(hipGetLastError());
  }

  void *p_va, *p_vb, *p_q, *p_scan, *p_rs, *p_tiles, *p_bt;
  GRB_TRY(scratch(7, 4 * (size_t)nwords, &p_va));
  GRB_TRY(scratch(8, 4 * (size_t)nwords, &p_vb));
  GRB_TRY(scratch(9, 4 * (size_t)n + 4, &p_q));
  GRB_TRY(scratch(2, 4 * (size_t)n + 4, &p_scan));
  GRB_TRY(scratch(11, 4 * (size_t)n + 4, &p_rs));
  const int max_tiles = ceil_div(n, kDegTile);
  GRB_TRY(scratch(3, 4 * (size_t)(2 * max_tiles + 2), &p_tiles));
  GRB_TRY(scratch(6, 4 * (size_t)(2 * btiles + 4 + max_chunks + 2), &p_bt));
  unsigned int* vis = (unsigned int*)p_va;            // current visited set
  unsigned int* vis_alt = (unsigned int*)p_vb;        // the set before the last level
  Index* queue = (Index*)p_q;
  int* local_scan = (int*)p_scan;
  Index* row_start = (Index*)p_rs;
  int* tile_sums = (int*)p_tiles;
  int* tile_off = tile_sums + max_tiles;
  int* btile_counts = (int*)p_bt;
  int* btile_off = btile_counts + btiles;             // btiles + 1 entries
  Index* chunk_owner = (Index*)(btile_off + btiles + 2);
  int* d_state = c.d_mail + 8;    // [0] discovered (diff count), [1] expanded edges, [2..3] inspected (u64)

  GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
  float* label = (float*)v->d_val;
  GRB_TRY(k_fill(GRB_F32, label, 0.0, n));
  GRB_HIP_TRY(hipMemsetAsync(vis, 0, 4 * (size_t)nwords, s));
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, s, vis, label, queue, source);
  GRB_HIP_TRY(hipGetLastError());

  // profile bit 0: HIP events around every level's expansion kernels (cheap, reusable pool)
  // profile bit 1: pull levels additionally count the edges they inspect
  static std::vector<hipEvent_t> pool;
  size_t used = 0;
  const bool count_inspected = (profile & 2) != 0;
  auto mark = [&]() -> grb_info {
    if (!profile) return GRB_SUCCESS;
    if (used == pool.size()) {
      hipEvent_t e;
      GRB_HIP_TRY(hipEventCreate(&e));
      pool.push_back(e);
    }
    GRB_HIP_TRY(hipEventRecord(pool[used++], s));
    return GRB_SUCCESS;
  };

  // state of the two frontier Vector objects of bfs.hpp (f1, f2): storage + ratio_
  bool f1_dense = (mode == GRB_PULLONLY);
  float ratio_f1 = 0.f, ratio_f2 = 0.f;
  bool have_queue = true;          // queue holds the current frontier (level 1: the source)
  Index nf = 1;
  int iter = 1, levels = 0;
  GRB_HIP_TRY(hipEventRecord(c.ev0, s));
  for (; iter <= desc->max_niter; ++iter) {
    // ---- vxm's direction decision on u = f1 (operations.hpp:131-140, vector.hpp:291-323)
    if (mode == GRB_PUSHPULL) {
      const float ratio = (float)nf / (float)n;
      if (!f1_dense) {
        if (ratio > desc->switchpoint && ratio > ratio_f1) f1_dense = true; else ratio_f1 = ratio;
      } else {
        if (ratio <= desc->switchpoint && ratio < ratio_f1) f1_dense = false; else ratio_f1 = ratio;
      }
    } else {
      f1_dense = (mode == GRB_PULLONLY);
    }
    GRB_HIP_TRY(hipMemsetAsync(d_state, 0, 4 * sizeof(int), s));
    if (!f1_dense) {
      if (!have_queue) {
        // the frontier is what the previous level discovered: list (vis & ~vis_alt), ordered,
        // with the tile offsets the count + scan of that level already produced
        hipLaunchKernelGGL(bitmap_list_kernel, dim3(btiles), dim3(kBlock), 0, s, vis, vis_alt, nwords, btile_off,
                           queue);
        GRB_HIP_TRY(hipGetLastError());
      }
      GRB_TRY(mark());
      GRB_HIP_TRY(hipMemcpyAsync(vis_alt, vis, 4 * (size_t)nwords, hipMemcpyDeviceToDevice, s));
      BfsPushVisitor vis_fn{vis, label, (float)(iter + 1)};
      GRB_TRY(launch_lb_expand(s, A->csr, queue, nf, max_edges, local_scan, row_start, tile_sums, tile_off,
                               chunk_owner, d_state + 1, vis_fn));
      GRB_TRY(mark());
      desc->lastmxv = GRB_PUSHONLY;
    } else {
      const int grid = stream_grid((long long)ceil_div(n, kWave) * kWave, kBlock);
      GRB_TRY(mark());
      if (count_inspected)
        hipLaunchKernelGGL((bfs_pull_kernel<true>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           A->d_no_in_edges, vis_alt, label, (float)(iter + 1),
                           reinterpret_cast<unsigned long long*>(d_state + 2));
      else
        hipLaunchKernelGGL((bfs_pull_kernel<false>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           A->d_no_in_edges, vis_alt, label, (float)(iter + 1), (unsigned long long*)nullptr);
      GRB_HIP_TRY(hipGetLastError());
      GRB_TRY(mark());
      std::swap(vis, vis_alt);       // vis = new set, vis_alt = set before this level
      desc->lastmxv = GRB_PULLONLY;
    }
    // discovered = |vis & ~vis_alt| ; the tile offsets are kept for a possible queue listing
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(btiles), dim3(kBlock), 0, s, vis, vis_alt, nwords, btile_counts);
    GRB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, btile_counts, btiles, btile_off, d_state);
    GRB_HIP_TRY(hipGetLastError());
    int h[4] = {0, 0, 0, 0};
    GRB_TRY(fetch_ints(d_state, 4, h));
    have_queue = false;
    if (levels_out && levels < max_levels) {
      grb_bfs_level& L = levels_out[levels];
      L.direction = f1_dense ? 1 : 0;
      L.frontier = nf;
      L.frontier_edges = f1_dense ? (count_inspected ? (int64_t)(((uint64_t)(uint32_t)h[3] << 32) | (uint32_t)h[2]) : 0)
                                  : (int64_t)h[1];
      L.discovered = h[0];
      L.ms = 0.f;
    }
    ++levels;
    // f2.swap(&f1): storage follows the output of this level, ratio_ slots exchange
    std::swap(ratio_f1, ratio_f2);
    nf = h[0];
    if (nf == 0) break;             // reduce(succ) == 0, bfs.hpp:75-76
  }
  GRB_HIP_TRY(hipEventRecord(c.ev1, s));
  const bool hit_cap = iter > desc->max_niter;
  if (hit_cap && nf > 0) {
    hipLaunchKernelGGL(bfs_unlabel_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, label, n,
                       (float)(desc->max_niter + 1));
    GRB_HIP_TRY(hipGetLastError());
  }
  // tally (outside the timed loop): 32 x {edges, reached} partial slots
  void* p_tally;
  GRB_TRY(scratch(10, 64 * sizeof(unsigned long long), &p_tally));
  unsigned long long* d_tally = (unsigned long long*)p_tally;
  GRB_HIP_TRY(hipMemsetAsync(d_tally, 0, 64 * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(bfs_tally_kernel, dim3(stream_grid(n, kBlock * 8)), dim3(kBlock), 0, s, label, A->csr.ptr, n,
                     d_tally);
  GRB_HIP_TRY(hipGetLastError());
  unsigned long long h_tally[64];
  GRB_HIP_TRY(hipMemcpyAsync(h_tally, d_tally, sizeof(h_tally), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  unsigned long long tot_e = 0, tot_r = 0;
  for (int i = 0; i < 32; ++i) { tot_e += h_tally[2 * i]; tot_r += h_tally[2 * i + 1]; }
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  if (result) {
    result->levels = levels;
    result->tight_ms = ms;
    result->edges_traversed = (int64_t)tot_e;
    result->reached = (int32_t)tot_r;
  }
  if (profile) {
    for (size_t i = 0; i + 1 < used; i += 2) {
      float lm = 0.f;
      (void)hipEventElapsedTime(&lm, pool[i], pool[i + 1]);
      size_t lv = i / 2;
      if (levels_out && (int)lv < max_levels && (int)lv < levels) levels_out[lv].ms = lm;
    }
  }
  v->d_nnz = (Index)tot_r;
  return GRB_SUCCESS;
}
