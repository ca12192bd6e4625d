// spmm.hip -- mxm with a dense right-hand side:  C = A (+).(x) B,  B and C dense row-major
// with k columns.  The reference declares it and stops ("SpMat x DeMat SpMM ... not implemented",
// backend/cuda/operations.hpp:52-70, spmm.hpp:15-27); it is the multi-frontier product of
// SURVEY.md 8(f)4 (k right-hand sides = k simultaneous SpMVs: batched PageRank / SSSP / label
// propagation).  One kernel family:
//
//   spmm_tile_kernel<SR,T>   any of the 17 semirings, f32 / i32.  A wave takes a wave tile of the
//       matrix's SpMV plan (<= 512 nonzeros, <= 64 rows; a long row is cut into slices with a
//       partial slot each): the tile's (column, value) pairs are read once, coalesced, 8 per
//       lane; then lane c owns output column c -- every nonzero costs ONE coalesced read of a
//       row of B (4k bytes) and one multiply-add per lane.  The gathers of B are the traffic:
//       4k bytes per nonzero against 8 for the matrix entry.
//
// Row sums are formed in the CSR order.  (Rounds 2-4 carried a second, opt-in path that multiplied the dense
// 16 x 16 tiles of the hub x hub core on the matrix cores, v_mfma_f32_16x16x4_f32: slower than this kernel on every
// power-law graph measured -- a few percent of the entries live in such tiles, and f32 MFMA runs at the vector rate --
// and removed in round 5; the measurements are in docs/experiments.md, the code in the history.)
#include "common.hpp"

#include <algorithm>
#include <numeric>

namespace grb {

constexpr int kSpmmTile = 512;        // must equal the SpMV plan's wave tile

typedef float f32x4 __attribute__((ext_vector_type(4)));

// KG = output columns one lane group owns (16, 32 or 64); the 64 / KG groups of a wave take different
// ROWS of the tile at the same time, so a narrow right-hand side still fills the wave and every row
// keeps its CSR summation order.
template <int SR, typename T, int KG>
__global__ __launch_bounds__(kBlock) void spmm_tile_kernel(const SpmvBlock* __restrict__ tiles, int ntiles,
                                                           const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                           const T* __restrict__ val, const T* __restrict__ B,
                                                           T* __restrict__ C, T* __restrict__ partials, Index k,
                                                           Index col0) {
  typedef Semiring<SR, T> S;
  constexpr int G = kWave / KG;
  __shared__ Index s_ci[kWavesPerBlock][kSpmmTile];
  __shared__ T s_cv[kWavesPerBlock][kSpmmTile];
  const int lane = lane_id();
  const int grp = lane / KG;
  const Index c = col0 + (lane % KG);
  const bool on = c < k;
  const Index cc = on ? c : 0;
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int t = blockIdx.x * kWavesPerBlock + wave_id(); t < ntiles; t += nwaves) {
    const SpmvBlock blk = tiles[t];
    const int cnt = blk.nnz_end - blk.nnz_start;
    // the tile's entries, 8 per lane, coalesced, parked in this wave's LDS region: every lane group then
    // reads the entries of ITS row from there
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < kSpmmTile / kWave; ++j) {
      const int at = j * kWave + lane;
      // an all-empty tile (trailing empty rows) starts AT the end of the arrays: nothing of it may be read
      const int q = cnt > 0 ? blk.nnz_start + (at < cnt ? at : 0) : 0;
      s_ci[wave_id()][at] = cnt > 0 ? ind[q] : 0;
      s_cv[wave_id()][at] = cnt > 0 ? val[q] : (T)0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto entry = [&](int at, Index* col, T* v) {
      *col = s_ci[wave_id()][at & (kSpmmTile - 1)];
      *v = s_cv[wave_id()][at & (kSpmmTile - 1)];
    };
    const bool slice = blk.slot >= 0;
    for (int r0 = blk.row_start; r0 < blk.row_end; r0 += G) {
      const int r = r0 + grp;
      const bool mine = slice ? grp == 0 : r < blk.row_end;
      int at0 = 0, len = 0;
      if (mine) {
        at0 = slice ? 0 : (int)(ptr[r] - blk.nnz_start);
        len = slice ? cnt : (int)(ptr[r + 1] - ptr[r]);
      }
      T acc = S::identity();
      for (int e = 0; __any(e < len); e += 4) {            // four rows of B in flight per lane
        Index jj[4];
        T aa[4], bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ee = e + u < len ? e + u : (len > 0 ? len - 1 : 0);
          entry(at0 + ee, &jj[u], &aa[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) bb[u] = B[(size_t)jj[u] * k + cc];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e + u < len) acc = S::add(acc, S::mul(aa[u], bb[u]));
      }
      if (mine && on) {
        if (slice) partials[(size_t)blk.slot * k + c] = acc;
        else C[(size_t)r * k + c] = acc;
      }
    }
  }
}

// long rows: fold the partial vectors of a row's slices, in slice order
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void spmm_finalize_kernel(const int* __restrict__ long_row,
                                                               const int* __restrict__ slot_ptr, int nlong,
                                                               const T* __restrict__ partials, T* __restrict__ C, Index k) {
  typedef Semiring<SR, T> S;
  const long long total = (long long)nlong * k;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / k);
    const Index c = (Index)(i % k);
    T acc = S::identity();
    for (int s = slot_ptr[row]; s < slot_ptr[row + 1]; ++s) acc = S::add(acc, partials[(size_t)s * k + c]);
    C[(size_t)long_row[row] * k + c] = acc;
  }
}

template <int SR, typename T>
__global__ void spmm_empty_kernel(T* C, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    C[i] = Semiring<SR, T>::identity();
}

}  // namespace grb

using namespace grb;

namespace {

template <int SR, typename T>
grb_info run_tiles(const CsrArrays& M, const SpmvPlan& plan, const T* B, T* C, Index k) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  if (M.nvals == 0 || plan.ntiles == 0) {
    const long long total = (long long)plan.nrows * k;
    if (total > 0) {
      hipLaunchKernelGGL((spmm_empty_kernel<SR, T>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, s, C, total);
      GRB_HIP_TRY(hipGetLastError());
    }
    return GRB_SUCCESS;
  }
  void* p_part = nullptr;
  if (plan.nslots > 0) GRB_TRY(scratch(6, sizeof(T) * (size_t)plan.nslots * (size_t)k, &p_part));
  const int grid = stream_grid((long long)plan.ntiles * kWave, kBlock);
#define GRB_SPMM_LAUNCH(KG, COL0)                                                                                     \
  hipLaunchKernelGGL((spmm_tile_kernel<SR, T, KG>), dim3(grid), dim3(kBlock), 0, s, plan.d_tiles, plan.ntiles, M.ptr, \
                     M.ind, (const T*)M.val, B, C, (T*)p_part, k, COL0)
  if (k <= 16) GRB_SPMM_LAUNCH(16, 0);
  else if (k <= 32) GRB_SPMM_LAUNCH(32, 0);
  else
    for (Index col0 = 0; col0 < k; col0 += kWave) {
      if (k - col0 <= 16) GRB_SPMM_LAUNCH(16, col0);
      else if (k - col0 <= 32) GRB_SPMM_LAUNCH(32, col0);
      else GRB_SPMM_LAUNCH(64, col0);
    }
#undef GRB_SPMM_LAUNCH
  GRB_HIP_TRY(hipGetLastError());
  if (plan.nlong > 0) {
    hipLaunchKernelGGL((spmm_finalize_kernel<SR, T>), dim3(stream_grid((long long)plan.nlong * k, kBlock)), dim3(kBlock), 0,
                       s, plan.d_long_row, plan.d_long_slot_ptr, plan.nlong, (const T*)p_part, C, k);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}

}  // namespace

extern "C" {

grb_info grb_spmm(grb_semiring op, grb_matrix A, int tran, const void* d_B, void* d_C, grb_index k, grb_descriptor desc) { GRB_API_ENTER();
  (void)desc;
  if (!A || !d_B || !d_C) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (k < 1) return GRB_INVALID_VALUE;
  const CsrArrays& M = tran ? A->csc : A->csr;
  SpmvPlan& plan = tran ? A->plan_csc : A->plan_csr;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  GRB_TRY(matrix_ensure_plan(A, tran != 0));
  GRB_TRY(ctx_init());
  GRB_TRY(dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    return run_tiles<SR, T>(M, plan, (const T*)d_B, (T*)d_C, k);
  }));
  return GRB_SUCCESS;
}

}  // extern "C"
