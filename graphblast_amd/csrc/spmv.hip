// spmv.hip -- the pull half of mxv/vxm on gfx950.
//
//  k_spmv            generic semiring SpMV  w[i] = (+)_j A[i,j] (x) u[j]  with the optional
//                    mask / accum epilogue of backend/cuda/spmv.hpp:178-220.  The reference
//                    delegates to moderngpu's merge-path SpmvCsrBinary (source not in the
//                    mount); this is a row-block streaming design instead (the CSR-Adaptive
//                    idea): rows are grouped at build time into blocks of <= kTileNnz
//                    nonzeros, one 256-thread workgroup per block streams the block's
//                    column indices and values with fully coalesced loads, gathers u,
//                    stages the products in LDS and reduces each row from LDS with a
//                    per-block lanes-per-row width; rows longer than a tile are cut into
//                    slices reduced by whole workgroups into a partial array that a tiny
//                    second kernel folds in a fixed order (deterministic, no atomics).
//  k_spmv_masked_or  Boolean pull step with fused mask (kernels/spmv.hpp:10-59, all eight
//                    <Scmp, EarlyExit, OpReuse> variants): one 64-lane wave owns 64
//                    consecutive rows, lanes first probe their own row serially (the common
//                    early-exit case) and rows still undecided after kSerialProbe neighbours
//                    are finished cooperatively by the whole wave with coalesced index reads
//                    and a ballot, which bounds the divergence a hub row can cause.
//
// Algorithmic bytes per launch of k_spmv: 8*nnz + 12*n + 4.
#include "common.hpp"

namespace grb {

#ifndef GRB_SPMV_TILE
#define GRB_SPMV_TILE 2048
#endif
constexpr int kTileNnz = GRB_SPMV_TILE;      // nonzeros staged per workgroup (4 B each in LDS)
constexpr int kMaxRowsPerBlock = 1024;
constexpr int kLongSlice = 8192;    // slice of a long row reduced by one workgroup
// hub-packed kernel: one 1024-thread workgroup per CU owning all 160 KiB of LDS
constexpr int kHubThreads = 1024;
constexpr int kHubWaves = kHubThreads / kWave;
constexpr int kWaveTile = 512;      // nonzeros per wave tile (8 per lane), 2 KiB of LDS per wave
constexpr int kWaveRows = 256;      // rows per wave tile
constexpr int kHot = 32768;         // leading values of the packed vector kept in LDS (128 KiB)

// ---- plan ---------------------------------------------------------------------------
// Two tilings of the same row pointer are kept: workgroup tiles of <= kTileNnz nonzeros for
// the row-block kernel and wave tiles of <= kWaveTile nonzeros / kWaveRows rows for the
// hub-packed kernel.  A row longer than a tile is cut into slices that each produce one
// partial, folded in slice order by the finalize kernel (deterministic, no atomics).
static void cut_tiles(const std::vector<Index>& ptr, Index n, int tile_nnz, int max_rows, int slice,
                      std::vector<SpmvBlock>* blocks, std::vector<int>* long_row,
                      std::vector<int>* long_slot_ptr, int* nslots_out) {
  int nslots = 0;
  Index r = 0;
  while (r < n) {
    Index len = ptr[r + 1] - ptr[r];
    if (len > tile_nnz) {
      long_row->push_back(r);
      long_slot_ptr->push_back(nslots);
      for (Index s = ptr[r]; s < ptr[r + 1]; s += slice) {
        Index e = s + slice < ptr[r + 1] ? s + slice : ptr[r + 1];
        blocks->push_back(SpmvBlock{r, r + 1, s, e, nslots++});
      }
      ++r;
      continue;
    }
    Index start = r, nnz = 0;
    while (r < n && r - start < max_rows) {
      Index l = ptr[r + 1] - ptr[r];
      if (l > tile_nnz || nnz + l > tile_nnz) break;
      nnz += l;
      ++r;
    }
    blocks->push_back(SpmvBlock{start, r, ptr[start], ptr[r], -1});
  }
  long_slot_ptr->push_back(nslots);
  *nslots_out = nslots;
}

template <typename V>
static grb_info to_device(const std::vector<V>& h, V** d) {
  if (h.empty()) return GRB_SUCCESS;
  GRB_HIP_TRY(hipMalloc(d, sizeof(V) * h.size()));
  GRB_HIP_TRY(hipMemcpy(*d, h.data(), sizeof(V) * h.size(), hipMemcpyHostToDevice));
  return GRB_SUCCESS;
}

grb_info build_spmv_plan(const std::vector<Index>& ptr, Index n, Index nminor, SpmvPlan* plan) {
  free_spmv_plan(plan);
  plan->nminor = nminor;
  {
    std::vector<SpmvBlock> blocks;
    std::vector<int> long_row, long_slot_ptr;
    int nslots = 0;
    cut_tiles(ptr, n, kTileNnz, kMaxRowsPerBlock, kLongSlice, &blocks, &long_row, &long_slot_ptr, &nslots);
    plan->nblocks = (int)blocks.size();
    plan->nlong = (int)long_row.size();
    plan->nslots = nslots;
    GRB_TRY(to_device(blocks, &plan->d_blocks));
    if (plan->nlong) {
      GRB_TRY(to_device(long_row, &plan->d_long_row));
      GRB_TRY(to_device(long_slot_ptr, &plan->d_long_slot_ptr));
      GRB_HIP_TRY(hipMalloc(&plan->d_partials, 4 * (size_t)nslots));
    }
  }
  {
    std::vector<SpmvBlock> tiles;
    std::vector<int> long_row, long_slot_ptr;
    int nslots = 0;
    cut_tiles(ptr, n, kWaveTile, kWaveRows, kWaveTile, &tiles, &long_row, &long_slot_ptr, &nslots);
    plan->ntiles = (int)tiles.size();
    plan->t_nlong = (int)long_row.size();
    plan->t_nslots = nslots;
    GRB_TRY(to_device(tiles, &plan->d_tiles));
    if (plan->t_nlong) {
      GRB_TRY(to_device(long_row, &plan->d_t_long_row));
      GRB_TRY(to_device(long_slot_ptr, &plan->d_t_long_slot_ptr));
      GRB_HIP_TRY(hipMalloc(&plan->d_t_partials, 4 * (size_t)nslots));
    }
  }
  return GRB_SUCCESS;
}

void free_spmv_plan(SpmvPlan* plan) {
  void* ptrs[] = {plan->d_blocks, plan->d_long_row, plan->d_long_slot_ptr, plan->d_partials,
                  plan->d_tiles, plan->d_t_long_row, plan->d_t_long_slot_ptr, plan->d_t_partials,
                  plan->d_ind2, plan->d_order, plan->d_u2};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  *plan = SpmvPlan();
}

// ---- hub packing ----------------------------------------------------------------------
// Power-law graphs send most gathers to a small set of columns (RMAT-22: the 32 Ki most
// referenced columns take 52 % of the nonzeros, the top 1 Mi take 97 %), but their entries of
// u are scattered over the whole vector, so every hot value drags a mostly-cold 128 B line
// through L2 and the gather stream runs at the fabric rate (measured: 4.7 GB fetched per
// launch against 1.08 GB algorithmic).  The first SpMV of an orientation therefore ranks the
// columns by reference count and keeps a private copy of the column ids renamed by rank.
// Each launch packs u by that order (one 4 B gather per column), which puts the hot values
// in a dense prefix: the first kHot of them are staged in LDS by every workgroup and the next
// few hundred thousand stay L2-resident.  The row sums are formed in the same order as
// before, so results are bit-identical to the unpacked kernel.
__global__ void column_count_kernel(const Index* __restrict__ ind, Index nvals, int* __restrict__ cnt) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index p = (Index)blockIdx.x * blockDim.x + threadIdx.x; p < nvals; p += stride)
    atomicAdd(&cnt[ind[p]], 1);
}

__global__ void rename_columns_kernel(const Index* __restrict__ ind, Index nvals, const Index* __restrict__ rank,
                                      Index* __restrict__ ind2) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index p = (Index)blockIdx.x * blockDim.x + threadIdx.x; p < nvals; p += stride) ind2[p] = rank[ind[p]];
}

template <typename T>
__global__ void pack_vector_kernel(const T* __restrict__ u, const Index* __restrict__ order, Index n,
                                   T* __restrict__ u2) {
  const Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) u2[i] = u[order[i]];
}

static grb_info prepare_hub_packing(const CsrArrays& M, SpmvPlan& plan) {
  plan.hub_ready = true;
  const Index m = plan.nminor;
  plan.nhot = m < kHot ? m : kHot;      // a short vector fits LDS whole: no renaming needed
  if (m <= kHot || M.nvals == 0) return GRB_SUCCESS;
  int* d_cnt = nullptr;
  GRB_HIP_TRY(hipMalloc(&d_cnt, 4 * (size_t)m));
  GRB_HIP_TRY(hipMemsetAsync(d_cnt, 0, 4 * (size_t)m, ctx().stream));
  hipLaunchKernelGGL(column_count_kernel, dim3(stream_grid(M.nvals, kBlock)), dim3(kBlock), 0, ctx().stream,
                     M.ind, M.nvals, d_cnt);
  std::vector<int> cnt((size_t)m);
  GRB_HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx().stream));
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  (void)hipFree(d_cnt);
  // counting sort by descending count, ties in column order
  int maxc = 0;
  for (int c : cnt) maxc = c > maxc ? c : maxc;
  std::vector<Index> bucket((size_t)maxc + 2, 0);
  for (int c : cnt) bucket[(size_t)(maxc - c) + 1]++;
  for (size_t b = 1; b < bucket.size(); ++b) bucket[b] += bucket[b - 1];
  std::vector<Index> order((size_t)m), rank((size_t)m);
  for (Index c = 0; c < m; ++c) {
    Index r = bucket[(size_t)(maxc - cnt[c])]++;
    order[r] = c;
    rank[c] = r;
  }
  // worth it only when the LDS prefix takes a real share of the gathers
  long long hot_refs = 0;
  for (Index r = 0; r < kHot; ++r) hot_refs += cnt[order[r]];
  if (hot_refs * 8 < (long long)M.nvals) {   // < 12.5 %: keep the natural column order
    plan.nhot = 0;
    return GRB_SUCCESS;
  }
  Index* d_rank = nullptr;
  GRB_TRY(to_device(rank, &d_rank));
  GRB_TRY(to_device(order, &plan.d_order));
  GRB_HIP_TRY(hipMalloc(&plan.d_ind2, 4 * (size_t)M.nvals));
  GRB_HIP_TRY(hipMalloc(&plan.d_u2, 4 * (size_t)m));
  hipLaunchKernelGGL(rename_columns_kernel, dim3(stream_grid(M.nvals, kBlock)), dim3(kBlock), 0, ctx().stream,
                     M.ind, M.nvals, d_rank, plan.d_ind2);
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  (void)hipFree(d_rank);
  return GRB_SUCCESS;
}

template <int SR, typename T>
__device__ inline void spmv_store(T* w, Index row, T value, const void* mask, int mask_f32, int scmp,
                                  int accum) {
  typedef Semiring<SR, T> S;
  if (mask && !mask_pass(mask, mask_f32, scmp, row)) value = S::identity();
  if (accum) value = S::add(w[row], value);
  w[row] = value;
}

template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(
    const SpmvBlock* __restrict__ blocks, const Index* __restrict__ ptr, const Index* __restrict__ ind,
    const T* __restrict__ val, const T* __restrict__ u, const void* __restrict__ mask, int mask_f32,
    int scmp, int accum, T* w, T* __restrict__ partials) {
  typedef Semiring<SR, T> S;
  __shared__ T prod[kTileNnz];
  __shared__ T wsum[kWavesPerBlock];
  const SpmvBlock b = blocks[blockIdx.x];
  const int tid = threadIdx.x;

  if (b.slot >= 0) {
    // slice of one long row: every thread folds a strided share, then a block fold
    T acc = S::identity();
    for (Index p = b.nnz_start + tid; p < b.nnz_end; p += kBlock)
      acc = S::add(acc, S::mul(val[p], u[ind[p]]));
    acc = wave_reduce(acc, [](T a, T c) { return S::add(a, c); });
    if (lane_id() == 0) wsum[wave_id()] = acc;
    __syncthreads();
    if (tid == 0) {
      T t = wsum[0];
#pragma unroll
      for (int k = 1; k < kWavesPerBlock; ++k) t = S::add(t, wsum[k]);
      partials[b.slot] = t;
    }
    return;
  }

  // ---- stream the block's nonzeros: coalesced (ind, val), gathered u, product -> LDS
  const int nnz = b.nnz_end - b.nnz_start;
  constexpr int kPer = kTileNnz / kBlock;   // 8 independent loads in flight per lane
  Index c[kPer];
  T a[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    int o = tid + k * kBlock;
    if (o < nnz) {
      c[k] = ind[b.nnz_start + o];
      a[k] = val[b.nnz_start + o];
    }
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    int o = tid + k * kBlock;
    if (o < nnz) prod[o] = S::mul(a[k], u[c[k]]);
  }
  __syncthreads();

  // ---- reduce rows out of LDS with L lanes per row (L adapted to the block's density)
  const int nrows = b.row_end - b.row_start;
  int L = 1;
  {
    int avg = nrows > 0 ? nnz / nrows : 0;
    while (L < kWave && L * 4 < avg) L <<= 1;          // ~4 elements per lane
    while (L < kWave && nrows * L * 2 <= kBlock) L <<= 1;  // few rows: use the idle lanes
  }
  const int groups = kBlock / L;
  const int g = tid / L, l = tid % L;
  for (int rr = g; rr < ((nrows + groups - 1) / groups) * groups; rr += groups) {
    T acc = S::identity();
    if (rr < nrows) {
      const Index row = b.row_start + rr;
      const int s = ptr[row] - b.nnz_start, e = ptr[row + 1] - b.nnz_start;
      for (int i = s + l; i < e; i += L) acc = S::add(acc, prod[i]);
    }
    acc = group_reduce(acc, L, [](T x, T y) { return S::add(x, y); });
    if (rr < nrows && l == 0) spmv_store<SR, T>(w, b.row_start + rr, acc, mask, mask_f32, scmp, accum);
  }
}

template <int SR, typename T>
__global__ void spmv_long_finalize_kernel(const int* __restrict__ long_row, const int* __restrict__ slot_ptr,
                                          int nlong, const T* __restrict__ partials,
                                          const void* __restrict__ mask, int mask_f32, int scmp, int accum,
                                          T* w) {
  typedef Semiring<SR, T> S;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlong) return;
  T acc = S::identity();
  for (int s = slot_ptr[i]; s < slot_ptr[i + 1]; ++s) acc = S::add(acc, partials[s]);
  spmv_store<SR, T>(w, long_row[i], acc, mask, mask_f32, scmp, accum);
}

// ---- hub-packed kernel ------------------------------------------------------------------
// One persistent 1024-thread workgroup per CU.  The workgroup stages the first `nhot` values
// of the (packed) input vector in LDS once; after that its 16 waves run independently, each
// looping over wave tiles with a private 2 KiB product buffer and no workgroup barrier:
//   stream   8 coalesced (column, value) pairs per lane, non-temporal so the matrix stream does
//            not push the vector out of L2;
//   gather   column < nhot from LDS, the rest from global memory (L2 for the warm tail);
//   reduce   products -> LDS, rows folded with L lanes per row (L adapted to the tile).
template <int SR, typename T>
__global__ __launch_bounds__(kHubThreads) void spmv_hub_kernel(
    const SpmvBlock* __restrict__ tiles, int ntiles, const Index* __restrict__ ptr,
    const Index* __restrict__ ind, const T* __restrict__ val, const T* __restrict__ u, int nhot,
    const void* __restrict__ mask, int mask_f32, int scmp, int accum, T* w, T* __restrict__ partials) {
  typedef Semiring<SR, T> S;
  __shared__ T hot[kHot];
  __shared__ T stage[kHubWaves][kWaveTile];
  const int tid = threadIdx.x;
  for (int i = tid; i < nhot; i += kHubThreads) hot[i] = u[i];
  __syncthreads();
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T* prod = stage[wave];
  constexpr int kPer = kWaveTile / kWave;
  for (int t = blockIdx.x * kHubWaves + wave; t < ntiles; t += gridDim.x * kHubWaves) {
    const SpmvBlock b = tiles[t];
    const int nnz = b.nnz_end - b.nnz_start;
    Index c[kPer];
    T a[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int o = lane + k * kWave;
      if (o < nnz) {
        c[k] = __builtin_nontemporal_load(&ind[b.nnz_start + o]);
        a[k] = __builtin_nontemporal_load(&val[b.nnz_start + o]);
      }
    }
    if (b.slot >= 0) {                 // slice of a long row: fold in registers
      T acc = S::identity();
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int o = lane + k * kWave;
        if (o < nnz) {
          const T x = (unsigned)c[k] < (unsigned)nhot ? hot[c[k]] : u[c[k]];
          acc = S::add(acc, S::mul(a[k], x));
        }
      }
      acc = wave_reduce(acc, [](T p, T q) { return S::add(p, q); });
      if (lane == 0) partials[b.slot] = acc;
      continue;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int o = lane + k * kWave;
      if (o < nnz) {
        const T x = (unsigned)c[k] < (unsigned)nhot ? hot[c[k]] : u[c[k]];
        prod[o] = S::mul(a[k], x);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nrows = b.row_end - b.row_start;
    int L = 1;
    {
      const int avg = nrows > 0 ? nnz / nrows : 0;
      while (L < kWave && L * 4 < avg) L <<= 1;
      while (L < kWave && nrows * L * 2 <= kWave) L <<= 1;
    }
    const int groups = kWave / L;
    const int g = lane / L, l = lane % L;
    for (int rr = g; rr < ((nrows + groups - 1) / groups) * groups; rr += groups) {
      T acc = S::identity();
      if (rr < nrows) {
        const Index row = b.row_start + rr;
        const int s = ptr[row] - b.nnz_start, e = ptr[row + 1] - b.nnz_start;
        for (int i = s + l; i < e; i += L) acc = S::add(acc, prod[i]);
      }
      acc = group_reduce(acc, L, [](T x, T y) { return S::add(x, y); });
      if (rr < nrows && l == 0) spmv_store<SR, T>(w, b.row_start + rr, acc, mask, mask_f32, scmp, accum);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

static bool spmv_use_legacy() {
  static const bool v = [] { const char* e = getenv("GRB_SPMV_ROWBLOCK"); return e && atoi(e) != 0; }();
  return v;
}

grb_info k_spmv(int sr, int dtype, const CsrArrays& M, SpmvPlan& plan, const void* u, const void* mask,
                int mask_f32, int scmp, int accum, void* w) {
  if (plan.nblocks == 0) return GRB_SUCCESS;
  if (spmv_use_legacy()) {
    return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
      using T = decltype(t);
      constexpr int SR = decltype(tag)::value;
      hipLaunchKernelGGL((spmv_stream_kernel<SR, T>), dim3(plan.nblocks), dim3(kBlock), 0, ctx().stream,
                         plan.d_blocks, M.ptr, M.ind, (const T*)M.val, (const T*)u, mask, mask_f32, scmp, accum,
                         (T*)w, (T*)plan.d_partials);
      GRB_HIP_TRY(hipGetLastError());
      if (plan.nlong) {
        hipLaunchKernelGGL((spmv_long_finalize_kernel<SR, T>), dim3(ceil_div(plan.nlong, kBlock)), dim3(kBlock), 0,
                           ctx().stream, plan.d_long_row, plan.d_long_slot_ptr, plan.nlong,
                           (const T*)plan.d_partials, mask, mask_f32, scmp, accum, (T*)w);
        GRB_HIP_TRY(hipGetLastError());
      }
      return GRB_SUCCESS;
    });
  }
  if (!plan.hub_ready) GRB_TRY(prepare_hub_packing(M, plan));
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    const Index* ind = M.ind;
    const T* uu = (const T*)u;
    if (plan.d_ind2) {
      hipLaunchKernelGGL((pack_vector_kernel<T>), dim3(ceil_div(plan.nminor, kBlock)), dim3(kBlock), 0, ctx().stream,
                         (const T*)u, plan.d_order, plan.nminor, (T*)plan.d_u2);
      ind = plan.d_ind2;
      uu = (const T*)plan.d_u2;
    }
    int grid = ceil_div(plan.ntiles, kHubWaves);
    if (grid > ctx().num_cu) grid = ctx().num_cu;
    hipLaunchKernelGGL((spmv_hub_kernel<SR, T>), dim3(grid), dim3(kHubThreads), 0, ctx().stream, plan.d_tiles,
                       plan.ntiles, M.ptr, ind, (const T*)M.val, uu, plan.nhot, mask, mask_f32, scmp, accum, (T*)w,
                       (T*)plan.d_t_partials);
    GRB_HIP_TRY(hipGetLastError());
    if (plan.t_nlong) {
      hipLaunchKernelGGL((spmv_long_finalize_kernel<SR, T>), dim3(ceil_div(plan.t_nlong, kBlock)), dim3(kBlock), 0,
                         ctx().stream, plan.d_t_long_row, plan.d_t_long_slot_ptr, plan.t_nlong,
                         (const T*)plan.d_t_partials, mask, mask_f32, scmp, accum, (T*)w);
      GRB_HIP_TRY(hipGetLastError());
    }
    return GRB_SUCCESS;
  });
}

// ------------------------------------------------------------------------------------
// Boolean pull step with fused mask.
//   skip row            when  scmp XOR (mask[row] == 0)
//   discoverable(row)   when  some neighbour col has  (opreuse ? mask[col] != 0
//                                                              : u[col] != identity)
//   w[row] = discoverable ? 1 : 0   for EVERY row
constexpr int kSerialProbe = 8;

template <typename T, bool kOpReuse>
__device__ inline bool pull_hit(const void* mask, int mask_f32, const T* u, T identity, Index col) {
  if constexpr (kOpReuse) return mask_nonzero(mask, mask_f32, col);
  else return u[col] != identity;
}

template <typename T, bool kEarlyExit, bool kOpReuse>
__global__ __launch_bounds__(kBlock) void spmv_masked_or_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, Index nrows, const T* __restrict__ u,
    T identity, const void* __restrict__ mask, int mask_f32, int scmp, T* __restrict__ w) {
  const int lane = lane_id();
  const Index wave_global = (Index)blockIdx.x * kWavesPerBlock + wave_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index base = wave_global * kWave; base < nrows; base += nwaves * kWave) {
    const Index row = base + lane;
    bool active = false, found = false;
    Index s = 0, e = 0;
    if (row < nrows) {
      active = !(mask_nonzero(mask, mask_f32, row) ? (scmp != 0) : (scmp == 0));
      // equivalently: skip when scmp XOR (mask == 0)
      if (active) { s = ptr[row]; e = ptr[row + 1]; }
    }
    // phase 1: each lane probes up to kSerialProbe of its own neighbours
    Index p = s;
    if (active) {
      Index stop = (e - s > kSerialProbe) ? s + kSerialProbe : e;
      for (; p < stop; ++p) {
        if (pull_hit<T, kOpReuse>(mask, mask_f32, u, identity, ind[p])) {
          found = true;
          if (kEarlyExit) break;
        }
      }
      if (found && kEarlyExit) p = e;
    }
    // phase 2: rows with neighbours left are finished by the whole wave, one at a time
    unsigned long long todo = __ballot(active && p < e);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index rs = __shfl(p, src, kWave), re = __shfl(e, src, kWave);
      bool any = false;
      for (Index q = rs; q < re; q += kWave) {
        bool h = false;
        if (q + lane < re) h = pull_hit<T, kOpReuse>(mask, mask_f32, u, identity, ind[q + lane]);
        if (__ballot(h)) { any = true; if (kEarlyExit) break; }
      }
      if (lane == src && any) found = true;
    }
    if (row < nrows) w[row] = found ? (T)1 : (T)0;
  }
}

grb_info k_spmv_masked_or(int dtype, const CsrArrays& M, const void* u, double identity, const void* mask,
                          int mask_f32, int scmp, int earlyexit, int opreuse, void* w) {
  if (M.n <= 0) return GRB_SUCCESS;
  const int grid = stream_grid(M.n, kBlock);
  auto launch = [&](auto t) -> grb_info {
    using T = decltype(t);
#define GRB_PULL(EE, OR)                                                                          \
  hipLaunchKernelGGL((spmv_masked_or_kernel<T, EE, OR>), dim3(grid), dim3(kBlock), 0, ctx().stream, \
                     M.ptr, M.ind, M.n, (const T*)u, (T)identity, mask, mask_f32, scmp, (T*)w)
    if (earlyexit && opreuse) GRB_PULL(true, true);
    else if (earlyexit) GRB_PULL(true, false);
    else if (opreuse) GRB_PULL(false, true);
    else GRB_PULL(false, false);
#undef GRB_PULL
    GRB_HIP_TRY(hipGetLastError());
    return GRB_SUCCESS;
  };
  if (dtype == GRB_F32) return launch(float{});
  return launch(int{});
}

}  // namespace grb
