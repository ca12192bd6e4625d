// bfs_fused.hip -- direction-optimised BFS with a device-resident level loop.
//
// Same algorithm and same push/pull decisions as graphblas/algorithm/bfs.hpp:14-89 run
// through vxm (backend/cuda/operations.hpp:80-209 + vector.hpp:291-323 `convert`), but
// the per-level GraphBLAS call sequence  assign -> vxm(convert, spmspv|spmv) -> swap ->
// reduce  is collapsed to one expansion per level on a representation chosen for
// MI355X:
//
//   visited set     bitmap, n/8 bytes (512 KiB at RMAT-22: lives in every XCD's L2), not
//                   the reference's n x 4-byte float vector used as a 1-bit flag
//   frontier        index queue for push levels; for pull levels no frontier object at
//                   all: an unvisited vertex is discoverable iff ANY in-neighbour is in the
//                   visited bitmap as of the start of the level (a neighbour visited
//                   earlier than the current frontier would already have discovered it)
//   push level      degree scan + edge-balanced expansion (push_common.hpp); atomicOr on
//                   the bitmap deduplicates, the winner labels the vertex and appends it to
//                   the next queue (wave-aggregated atomic)
//   pull level      one wave per 64 vertices = 2 bitmap words; lanes probe their own
//                   in-neighbour list serially (early exit), leftovers are finished by the
//                   whole wave with coalesced reads + ballot; visited is double-buffered
//                   (read `in`, write `out` = in | new) so no atomics and no same-level races
//   level control   one 16-byte D2H mailbox read per level (the reference synchronises
//                   after every runtime call and returns three scan totals per level)
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
  Index* next_queue;
  int* next_count;
  __device__ void operator()(Index, Index, Index, Index dst) const {
    const unsigned int bit = 1u << (dst & 31);
    unsigned int* w = &visited[dst >> 5];
    if (*w & bit) return;
    const unsigned int old = atomicOr(w, bit);
    if (old & bit) return;
    label[dst] = new_label;
    next_queue[atomicAdd(next_count, 1)] = dst;
  }
};

__device__ inline bool bit_set(const unsigned int* __restrict__ bm, Index v) {
  return (bm[v >> 5] >> (v & 31)) & 1u;
}

template <bool kCountInspected>
__global__ __launch_bounds__(kBlock) void bfs_pull_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index n,
    const unsigned int* __restrict__ vin, unsigned int* __restrict__ vout, float* __restrict__ label,
    float new_label, int* __restrict__ counters /*[0] discovered, [1] unvisited, [2..3] inspected (u64)*/) {
  const int lane = lane_id();
  const Index nchunks = (n + kWave - 1) / kWave;
  const Index wave_global = (Index)blockIdx.x * kWavesPerBlock + wave_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  int found_total = 0;
  unsigned long long inspected = 0;
  int unvisited_total = 0;
  for (Index chunk = wave_global; chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const unsigned int word = vin[(chunk << 1) + (lane >> 5)];
    const bool was = (word >> (lane & 31)) & 1u;
    bool active = (v < n) && !was;
    unsigned long long act_mask = __ballot(active);
    if (act_mask == 0ull) {                       // whole chunk already visited
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
    found_total += __popcll(fb);
    unvisited_total += __popcll(act_mask);
  }
  if (lane == 0 && found_total) atomicAdd(&counters[0], found_total);
  if (kCountInspected) {
    inspected = wave_reduce(inspected, [](unsigned long long a, unsigned long long b) { return a + b; });
    if (lane == 0) {
      atomicAdd(&counters[1], unvisited_total);
      atomicAdd(reinterpret_cast<unsigned long long*>(&counters[2]), inspected);
    }
  }
}

// queue <- bits set in `now` but not in `before` (vertices discovered by the last pull)
__global__ void bfs_bitmap_diff_to_queue_kernel(const unsigned int* __restrict__ now,
                                                const unsigned int* __restrict__ before, int nwords,
                                                Index* __restrict__ queue, int* __restrict__ count) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
    unsigned int d = now[i] & ~before[i];
    if (!d) continue;
    int pos = atomicAdd(count, __popc(d));
    while (d) {
      int b = __ffs((int)d) - 1;
      d &= d - 1;
      queue[pos++] = (Index)i * 32 + b;
    }
  }
}

__global__ void bfs_seed_kernel(unsigned int* visited, float* label, Index* queue, Index source) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    visited[source >> 5] = 1u << (source & 31);
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

// TEPS numerator: sum of out-degree over labelled vertices, and their count
__global__ void bfs_tally_kernel(const float* __restrict__ label, const Index* __restrict__ ptr, Index n,
                                 unsigned long long* __restrict__ out /*[0] edges, [1] reached*/) {
  unsigned long long edges = 0, reached = 0;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (label[i] != 0.f) { edges += (unsigned long long)(ptr[i + 1] - ptr[i]); ++reached; }
  }
  edges = wave_reduce(edges, [](unsigned long long a, unsigned long long b) { return a + b; });
  reached = wave_reduce(reached, [](unsigned long long a, unsigned long long b) { return a + b; });
  if (lane_id() == 0) { atomicAdd(&out[0], edges); atomicAdd(&out[1], reached); }
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
  const int mode = desc->desc[GRB_MXVMODE];

  void *p_va, *p_vb, *p_qa, *p_qb, *p_scan, *p_tiles;
  GRB_TRY(scratch(7, 4 * (size_t)nwords, &p_va));
  GRB_TRY(scratch(8, 4 * (size_t)nwords, &p_vb));
  GRB_TRY(scratch(9, 4 * (size_t)n + 4, &p_qa));
  GRB_TRY(scratch(10, 4 * (size_t)n + 4, &p_qb));
  GRB_TRY(scratch(2, 4 * (size_t)n + 4, &p_scan));
  const int max_tiles = ceil_div(n, kDegTile);
  GRB_TRY(scratch(3, 4 * (size_t)(2 * max_tiles + 2), &p_tiles));
  unsigned int* vis = (unsigned int*)p_va;            // current visited set
  unsigned int* vis_alt = (unsigned int*)p_vb;
  Index* queue = (Index*)p_qa;
  Index* queue_next = (Index*)p_qb;
  int* local_scan = (int*)p_scan;
  int* tile_sums = (int*)p_tiles;
  int* tile_off = tile_sums + max_tiles;
  int* d_state = c.d_mail + 8;   // [0] discovered, [1] unvisited, [2..3] inspected, [4] expanded edges
  float* label = (float*)v->d_val;

  GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
  label = (float*)v->d_val;
  GRB_TRY(k_fill(GRB_F32, label, 0.0, n));
  GRB_HIP_TRY(hipMemsetAsync(vis, 0, 4 * (size_t)nwords, s));
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, s, vis, label, queue, source);
  GRB_HIP_TRY(hipGetLastError());

  // profile bit 0: HIP events around every level's expansion kernels (cheap, reusable pool)
  // profile bit 1: additionally count unvisited vertices / inspected edges in pull levels
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
  bool have_queue = true;          // queue holds the current frontier
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
    GRB_HIP_TRY(hipMemsetAsync(d_state, 0, 5 * sizeof(int), s));
    int h[5] = {0, 0, 0, 0, 0};
    if (!f1_dense) {
      if (!have_queue) {
        // previous level was a pull: materialise its discoveries (vis minus vis_alt)
        hipLaunchKernelGGL(bfs_bitmap_diff_to_queue_kernel, dim3(stream_grid(nwords)), dim3(kBlock), 0, s, vis,
                           vis_alt, nwords, queue, d_state);
        GRB_HIP_TRY(hipGetLastError());
        GRB_HIP_TRY(hipMemsetAsync(d_state, 0, sizeof(int), s));
        have_queue = true;
      }
      const int ntiles = ceil_div(nf, kDegTile);
      GRB_TRY(mark());
      hipLaunchKernelGGL(push_degree_kernel, dim3(ntiles), dim3(kBlock), 0, s, A->csr.ptr, queue, nf, local_scan,
                         tile_sums);
      GRB_HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, tile_sums, ntiles, tile_off,
                         d_state + 4);
      GRB_HIP_TRY(hipGetLastError());
      BfsPushVisitor vis_fn{vis, label, (float)(iter + 1), queue_next, d_state};
      hipLaunchKernelGGL((lb_expand_kernel<BfsPushVisitor>), dim3(2048), dim3(kBlock), 0, s, A->csr.ptr, A->csr.ind,
                         queue, nf, local_scan, tile_off, ntiles, vis_fn);
      GRB_HIP_TRY(hipGetLastError());
      GRB_TRY(mark());
      GRB_TRY(fetch_ints(d_state, 5, h));
      std::swap(queue, queue_next);
      desc->lastmxv = GRB_PUSHONLY;
    } else {
      const int grid = stream_grid((long long)ceil_div(n, kWave) * kWave, kBlock);
      GRB_TRY(mark());
      if (count_inspected)
        hipLaunchKernelGGL((bfs_pull_kernel<true>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           vis_alt, label, (float)(iter + 1), d_state);
      else
        hipLaunchKernelGGL((bfs_pull_kernel<false>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           vis_alt, label, (float)(iter + 1), d_state);
      GRB_HIP_TRY(hipGetLastError());
      GRB_TRY(mark());
      GRB_TRY(fetch_ints(d_state, 5, h));
      std::swap(vis, vis_alt);       // vis = new set, vis_alt = set before this level
      have_queue = false;
      desc->lastmxv = GRB_PULLONLY;
    }
    if (levels_out && levels < max_levels) {
      grb_bfs_level& L = levels_out[levels];
      L.direction = f1_dense ? 1 : 0;
      L.frontier = nf;
      L.frontier_edges = f1_dense ? (count_inspected ? (int64_t)(((uint64_t)(uint32_t)h[3] << 32) | (uint32_t)h[2]) : 0)
                                  : (int64_t)h[4];
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
  // tally (outside the timed loop)
  unsigned long long* d_tally = reinterpret_cast<unsigned long long*>(c.d_mail + 16);
  GRB_HIP_TRY(hipMemsetAsync(d_tally, 0, 16, s));
  hipLaunchKernelGGL(bfs_tally_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, label, A->csr.ptr, n, d_tally);
  GRB_HIP_TRY(hipGetLastError());
  int t[4];
  GRB_TRY(fetch_ints(c.d_mail + 16, 4, t));
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  if (result) {
    result->levels = levels;
    result->tight_ms = ms;
    result->edges_traversed = (int64_t)(((uint64_t)(uint32_t)t[1] << 32) | (uint32_t)t[0]);
    result->reached = t[2];
  }
  if (profile) {
    for (size_t i = 0; i + 1 < used; i += 2) {
      float lm = 0.f;
      (void)hipEventElapsedTime(&lm, pool[i], pool[i + 1]);
      size_t lv = i / 2;
      if (levels_out && (int)lv < max_levels && (int)lv < levels) levels_out[lv].ms = lm;
    }
  }
  v->d_nnz = result ? result->reached : 0;
  return GRB_SUCCESS;
}
