// spmv_bands.hpp -- column bands for the hub-packed SpMV (included by spmv.hip after its tuning constants).
//
// One LDS prefix of 32 Ki packed values serves 52 % of RMAT-22's gathers; the other 61 M go to L2 at its request
// rate (~226 G/s) and are 0.27 of the kernel's 0.59 ms.  The next prefixes still carry weight (the top 2 / 4 / 8
// x 32 Ki columns take 62 / 75 / 85 % of the nonzeros; tools/spmv_band_probe.py), so the matrix is split by
// column rank into k bands, each with its OWN LDS prefix:
//
//   band b >= 1     the entries whose column rank is in [b H, (b+1) H): band-major, row-major inside, the column
//                   stored as its position in the band's prefix.  A (row, band) run is a "piece".
//   main part       everything else: rank < H (prefix 0, LDS) and rank >= k H (cold, gathered from L2)
//
// and ONE launch walks the phases band 1, ..., band k-1, main: every workgroup (one per CU, all of LDS) refills
// its prefix between phases and owns a contiguous range of ROWS through all of them, so the band sums of a row
// are accumulated in a scratch vector t by one workgroup only -- ordered by __syncthreads, no atomics, no
// partial buffers -- and the main phase adds t[row] before the mask / accumulate epilogue.  Rows longer than a
// tile are sliced in every phase; their slices write partials that the existing finalize kernel folds (slot
// order: main slices, then band 1, ...), and are dealt round-robin over all workgroups for balance.
//
// Summation order of a row: its band sums in band order (each in column order inside the band), then the main
// part -- fixed, so results are deterministic, but no longer the plain column order of the unbanded kernel
// (equal for integer-valued data and for the idempotent monoids; within rounding for float sums).
#pragma once

namespace grb {

constexpr int kMaxBands = 8;

struct BandPhase {
  const SpmvBlock* short_tiles;   // whole rows / pieces; workgroup g owns [short_lo[g], short_lo[g + 1])
  const int* short_lo;
  const SpmvBlock* long_tiles;    // slices of long rows, dealt round-robin over the waves of the grid
  int nlong_tiles;
  const Index* ptr;               // main: row pointers of the main part; band: piece pointers
  const Index* rowmap;            // band: piece -> row; main: null
  const Index* ind;
  const void* val;
  Index hot_base;                 // first packed position staged in LDS for this phase
  int nhot;                       // how many
};

struct BandArgs {
  int k;                          // phases: bands 1 .. k-1, then the main part (index 0)
  int grid;                       // workgroups the row cuts were made for
  const Index* row_cut;           // [grid + 1] rows owned by each workgroup
  BandPhase ph[kMaxBands];
};

struct SpmvBands {
  BandArgs args;
  std::vector<void*> owned;       // every device allocation behind args
  int nlong = 0;
  int* d_long_row = nullptr;
  int* d_long_slot_ptr = nullptr;
  void* d_partials = nullptr;
  void* d_t = nullptr;            // [nrows] band sums per row
  long long band_nnz = 0, pieces = 0;
};

inline void free_spmv_bands(SpmvBands* b) {
  if (!b) return;
  for (void* p : b->owned)
    if (p) (void)hipFree(p);
  delete b;
}

// ---- preparation on the device -------------------------------------------------------------------------
__device__ inline int band_of(Index rk, int k) {          // kHot is a power of two
  const int b = (int)((unsigned)rk / (unsigned)kHot);
  return b < k ? b : 0;                                    // beyond the last prefix: the main part's cold gathers
}

// entries per (band, row): a wave per row, 64 entries per step, lane b counts band b
__global__ __launch_bounds__(kBlock) void band_count_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                            const Index* __restrict__ rank, Index n, int k,
                                                            unsigned int* __restrict__ cnt0, unsigned int* __restrict__ cntb) {
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index row = (Index)blockIdx.x * kWavesPerBlock + wave_id(); row < n; row += nwaves) {
    const Index s = ptr[row], e = ptr[row + 1];
    unsigned int mine = 0;
    for (Index q = s; q < e; q += kWave) {
      const int b = q + lane < e ? band_of(rank[ind[q + lane]], k) : -1;
      for (int bb = 0; bb < k; ++bb) {
        const unsigned long long m = __ballot(b == bb);
        if (lane == bb) mine += (unsigned int)__popcll(m);
      }
    }
    if (lane == 0) cnt0[row] = mine;
    else if (lane < k) cntb[(size_t)(lane - 1) * n + row] = mine;
  }
}

__global__ void band_flags_kernel(const unsigned int* __restrict__ cnt, long long m, unsigned int* __restrict__ flag) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) flag[i] = cnt[i] ? 1u : 0u;
}

// entries to their band (stable inside a row), pieces listed
__global__ __launch_bounds__(kBlock) void band_scatter_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, const unsigned int* __restrict__ val,
    const Index* __restrict__ rank, Index n, int k, const unsigned int* __restrict__ m0_ptr,
    const unsigned int* __restrict__ boff, const unsigned int* __restrict__ pid, const unsigned int* __restrict__ flag,
    Index* __restrict__ m0_ind, unsigned int* __restrict__ m0_val, Index* __restrict__ bs_ind,
    unsigned int* __restrict__ bs_val, Index* __restrict__ piece_ptr, Index* __restrict__ piece_row) {
  const int lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index row = (Index)blockIdx.x * kWavesPerBlock + wave_id(); row < n; row += nwaves) {
    const Index s = ptr[row], e = ptr[row + 1];
    if (s == e) continue;
    unsigned int next = 0;                                 // lane b: next free position of band b
    if (lane == 0) next = m0_ptr[row];
    else if (lane < k) {
      const size_t at = (size_t)(lane - 1) * n + row;
      next = boff[at];
      if (flag[at]) { piece_ptr[pid[at]] = (Index)next; piece_row[pid[at]] = row; }
    }
    for (Index q = s; q < e; q += kWave) {
      const bool live = q + lane < e;
      const Index rk = live ? rank[ind[q + lane]] : 0;
      const unsigned int v = live ? val[q + lane] : 0u;
      const int b = live ? band_of(rk, k) : -1;
      unsigned int pos = 0;
      for (int bb = 0; bb < k; ++bb) {
        const unsigned long long m = __ballot(b == bb);
        const unsigned int base = (unsigned int)__shfl((int)next, bb, kWave);
        if (b == bb) pos = base + (unsigned int)__popcll(m & below);
        if (lane == bb) next += (unsigned int)__popcll(m);
      }
      if (b == 0) { m0_ind[pos] = rk; m0_val[pos] = v; }
      else if (b > 0) { bs_ind[pos] = rk - (Index)b * kHot; bs_val[pos] = v; }
    }
  }
}

// ---- the kernel ------------------------------------------------------------------------------------------
// One phase of one wave: the three-stage pipeline of spmv_hub_kernel (stream / gather / reduce, see there) over
// the wave's tiles of this phase -- its share of the workgroup's whole-row tiles, then its share of the
// long-row slices.  t rides along: the row's band sum so far is loaded in the gather stage (piece -> row has
// landed with the stream by then) and, in a band phase, stored back in the reduce stage.
template <int SR, typename T, bool kMain>
__device__ inline void band_phase_run(const BandPhase& P, const T* __restrict__ u, const T* hot, T* prod, int lane,
                                      int wave, int wg, int grid, const void* __restrict__ mask, int mask_f32,
                                      int scmp, int accum, T* w, T* __restrict__ partials, T* t) {
  typedef Semiring<SR, T> S;
  constexpr int kPer = kWaveTile / kWave;
  const Index* __restrict__ ptr = P.ptr;
  const Index* __restrict__ ind = P.ind;
  const T* __restrict__ val = (const T*)P.val;
  const Index* __restrict__ rowmap = P.rowmap;
  const int s_lo = P.short_lo[wg], s_hi = P.short_lo[wg + 1];
  const int ns = s_hi - s_lo > wave ? (s_hi - s_lo - wave + kHubWaves - 1) / kHubWaves : 0;
  const int gw = wg * kHubWaves + wave, gstride = grid * kHubWaves;
  const int nl = P.nlong_tiles > gw ? (P.nlong_tiles - gw + gstride - 1) / gstride : 0;
  const int count = ns + nl;
  if (count == 0) return;
  const int nhot = P.nhot;
  const int hot_clamp = nhot > 0 ? nhot - 1 : 0;
  auto tile_at = [&](int i) {
    i = i < count ? i : count - 1;
    return i < ns ? P.short_tiles[s_lo + wave + i * kHubWaves] : P.long_tiles[gw + (i - ns) * gstride];
  };

  SpmvBlock b0, b1, b2;
  T pr[kPer];
  Index rlo0, rhi0, row0;        // lane r: pointers and output row of the tile's r-th row / piece
  T told0;                       // and t of that row
  Index c1[kPer], rlo1, rhi1, row1;
  T a1[kPer];

#define GRB_BAND_STREAM(B, C, A, RLO, RHI, ROW)                              \
  _Pragma("unroll") for (int k = 0; k < kPer; ++k) {                        \
    int p = (B).nnz_start + lane + k * kWave;                               \
    p = p < (B).nnz_end ? p : (B).nnz_end - 1;                              \
    p = p > 0 ? p : 0;                                                      \
    (C)[k] = stream_load(&ind[p]);                                          \
    (A)[k] = stream_load(&val[p]);                                          \
  }                                                                         \
  {                                                                         \
    const int r = (B).row_start + lane;                                     \
    RLO = ptr[r < (B).row_end ? r : (B).row_end];                           \
    RHI = ptr[r + 1 < (B).row_end ? r + 1 : (B).row_end];                   \
    const int rc = r < (B).row_end ? r : (B).row_end - 1;                   \
    ROW = kMain ? rc : rowmap[rc];                                          \
  }
#define GRB_BAND_COLD(C) (kMain && (unsigned)(C) >= (unsigned)nhot)
#define GRB_BAND_LDS_INDEX(C) ((C) < hot_clamp ? (C) : hot_clamp)
#define GRB_BAND_GATHER(C, XG, XL, ROW, TOLD)                                                       \
  if (kMain) { _Pragma("unroll") for (int k = 0; k < kPer; ++k) (XG)[k] = u[GRB_BAND_COLD((C)[k]) ? (C)[k] : 0]; } \
  _Pragma("unroll") for (int k = 0; k < kPer; ++k) (XL)[k] = hot[GRB_BAND_LDS_INDEX((C)[k])];         \
  TOLD = t[ROW];
#define GRB_BAND_PRODUCTS(B, C, A, XG, XL)                                               \
  _Pragma("unroll") for (int k = 0; k < kPer; ++k) {                                     \
    const T x = GRB_BAND_COLD((C)[k]) ? (XG)[k] : (XL)[k];                               \
    pr[k] = lane + k * kWave < (B).nnz_end - (B).nnz_start ? S::mul((A)[k], x) : S::identity(); \
  }

  b0 = tile_at(0);
  b1 = tile_at(1);
  b2 = tile_at(2);
  {
    Index c0[kPer];
    T a0[kPer], xg[kPer], xl[kPer];
    GRB_BAND_STREAM(b0, c0, a0, rlo0, rhi0, row0)
    GRB_BAND_GATHER(c0, xg, xl, row0, told0)
    asm volatile("" ::: "memory");
    GRB_BAND_STREAM(b1, c1, a1, rlo1, rhi1, row1)
    asm volatile("" ::: "memory");
    GRB_BAND_PRODUCTS(b0, c0, a0, xg, xl)
  }
  for (int i = 0; i < count; ++i) {
    T xg[kPer], xl[kPer], told1;
    Index c2[kPer], rlo2, rhi2, row2;
    T a2[kPer];
    GRB_BAND_GATHER(c1, xg, xl, row1, told1)
    asm volatile("" ::: "memory");
    GRB_BAND_STREAM(b2, c2, a2, rlo2, rhi2, row2)
    asm volatile("" ::: "memory");
    const SpmvBlock b3 = tile_at(i + 3);

    const int nnz = b0.nnz_end - b0.nnz_start;
    if (b0.slot >= 0) {                 // slice of a long row: one partial, folded by the finalize kernel
      T acc = pr[0];
#pragma unroll
      for (int k = 1; k < kPer; ++k) acc = S::add(acc, pr[k]);
      acc = wave_reduce(acc, [](T p, T q) { return S::add(p, q); });
      if (lane == 0) partials[b0.slot] = acc;
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) prod[lane + k * kWave] = pr[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int nrows = b0.row_end - b0.row_start;
      int L = 1;
      {
        const int avg = nrows > 0 ? nnz / nrows : 0;
        while (L < kWave && L * 4 < avg) L <<= 1;
        while (L < kWave && nrows * L * 2 <= kWave) L <<= 1;
      }
      const int groups = kWave / L;
      const int g = lane / L, l = lane % L;
      for (int base = 0; base < nrows; base += groups) {
        const int rr = base + g;
        const int s = __shfl(rlo0, rr, kWave) - b0.nnz_start;
        const int e = __shfl(rhi0, rr, kWave) - b0.nnz_start;
        const Index orow = __shfl(row0, rr, kWave);
        const T before = __shfl(told0, rr, kWave);
        T acc = S::identity();
        for (int q = s + l; q < e; q += L) acc = S::add(acc, prod[q]);
        acc = group_reduce(acc, L, [](T p, T q) { return S::add(p, q); });
        if (rr < nrows && l == 0) {
          if (kMain) spmv_store<SR, T>(w, orow, S::add(before, acc), mask, mask_f32, scmp, accum);
          else t[orow] = S::add(before, acc);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    GRB_BAND_PRODUCTS(b1, c1, a1, xg, xl)
    b0 = b1; rlo0 = rlo1; rhi0 = rhi1; row0 = row1; told0 = told1;
    b1 = b2; rlo1 = rlo2; rhi1 = rhi2; row1 = row2;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { c1[k] = c2[k]; a1[k] = a2[k]; }
    b2 = b3;
  }
#undef GRB_BAND_STREAM
#undef GRB_BAND_GATHER
#undef GRB_BAND_PRODUCTS
#undef GRB_BAND_COLD
#undef GRB_BAND_LDS_INDEX
}

template <int SR, typename T>
__global__ __launch_bounds__(kHubThreads) void spmv_band_kernel(BandArgs A, const T* __restrict__ u,
                                                                const void* __restrict__ mask, int mask_f32, int scmp,
                                                                int accum, T* w, T* __restrict__ partials, T* t) {
  typedef Semiring<SR, T> S;
  __shared__ T hot[kHot];
  __shared__ T stage[kHubWaves][kWaveTile];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  for (Index r = A.row_cut[wg] + tid; r < A.row_cut[wg + 1]; r += kHubThreads) t[r] = S::identity();
  for (int phase = 1; phase <= A.k; ++phase) {
    const int b = phase < A.k ? phase : 0;
    const BandPhase& P = A.ph[b];
    __syncthreads();                       // the previous phase's LDS reads and t stores are done
    for (int i = tid; i < P.nhot; i += kHubThreads) hot[i] = u[P.hot_base + i];
    __syncthreads();
    if (b == 0) band_phase_run<SR, T, true>(P, u, hot, stage[wave], lane, wave, wg, A.grid, mask, mask_f32, scmp, accum, w, partials, t);
    else band_phase_run<SR, T, false>(P, u, hot, stage[wave], lane, wave, wg, A.grid, mask, mask_f32, scmp, accum, w, partials, t);
  }
}

}  // namespace grb
