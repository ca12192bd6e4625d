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

grb_info build_spmv_plan(const std::vector<Index>& ptr, Index n, SpmvPlan* plan) {
  std::vector<SpmvBlock> blocks;
  std::vector<int> long_row, long_slot_ptr;
  int nslots = 0;
  Index r = 0;
  while (r < n) {
    Index len = ptr[r + 1] - ptr[r];
    if (len > kTileNnz) {
      long_row.push_back(r);
      long_slot_ptr.push_back(nslots);
      for (Index s = ptr[r]; s < ptr[r + 1]; s += kLongSlice) {
        Index e = s + kLongSlice < ptr[r + 1] ? s + kLongSlice : ptr[r + 1];
        blocks.push_back(SpmvBlock{r, r + 1, s, e, nslots++});
      }
      ++r;
      continue;
    }
    Index start = r, nnz = 0;
    while (r < n && r - start < kMaxRowsPerBlock) {
      Index l = ptr[r + 1] - ptr[r];
      if (l > kTileNnz || nnz + l > kTileNnz) break;
      nnz += l;
      ++r;
    }
    blocks.push_back(SpmvBlock{start, r, ptr[start], ptr[r], -1});
  }
  long_slot_ptr.push_back(nslots);
  free_spmv_plan(plan);
  plan->nblocks = (int)blocks.size();
  plan->nlong = (int)long_row.size();
  plan->nslots = nslots;
  if (plan->nblocks) {
    GRB_HIP_TRY(hipMalloc(&plan->d_blocks, sizeof(SpmvBlock) * blocks.size()));
    GRB_HIP_TRY(hipMemcpy(plan->d_blocks, blocks.data(), sizeof(SpmvBlock) * blocks.size(), hipMemcpyHostToDevice));
  }
  if (plan->nlong) {
    GRB_HIP_TRY(hipMalloc(&plan->d_long_row, sizeof(int) * long_row.size()));
    GRB_HIP_TRY(hipMemcpy(plan->d_long_row, long_row.data(), sizeof(int) * long_row.size(), hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMalloc(&plan->d_long_slot_ptr, sizeof(int) * long_slot_ptr.size()));
    GRB_HIP_TRY(hipMemcpy(plan->d_long_slot_ptr, long_slot_ptr.data(), sizeof(int) * long_slot_ptr.size(), hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMalloc(&plan->d_partials, 4 * (size_t)nslots));
  }
  return GRB_SUCCESS;
}

void free_spmv_plan(SpmvPlan* plan) {
  if (plan->d_blocks) (void)hipFree(plan->d_blocks);
  if (plan->d_long_row) (void)hipFree(plan->d_long_row);
  if (plan->d_long_slot_ptr) (void)hipFree(plan->d_long_slot_ptr);
  if (plan->d_partials) (void)hipFree(plan->d_partials);
  *plan = SpmvPlan();
}

// Epilogue shared by both kernels: mask -> identity where the mask FAILS
// (spmv.hpp:203-212), then optional accumulate with the semiring's add (:213-220).
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

grb_info k_spmv(int sr, int dtype, const CsrArrays& M, const SpmvPlan& plan, const void* u, const void* mask,
                int mask_f32, int scmp, int accum, void* w) {
  if (plan.nblocks == 0) return GRB_SUCCESS;
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
