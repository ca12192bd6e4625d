// spmm.hip -- mxm with a dense right-hand side:  C = A (+).(x) B,  B and C dense row-major
// with k columns.  The reference declares it and stops ("SpMat x DeMat SpMM ... not implemented",
// backend/cuda/operations.hpp:52-70, spmm.hpp:15-27); it is the multi-frontier product of
// SURVEY.md 8(f)4 (k right-hand sides = k simultaneous SpMVs: batched PageRank / SSSP / label
// propagation).  Two kernels:
//
//   spmm_tile_kernel<SR,T>   any of the 17 semirings, f32 / i32.  A wave takes a wave tile of the
//       matrix's SpMV plan (<= 512 nonzeros, <= 64 rows; a long row is cut into slices with a
//       partial slot each): the tile's (column, value) pairs are read once, coalesced, 8 per
//       lane; then lane c owns output column c -- every nonzero costs ONE coalesced read of a
//       row of B (4k bytes) and one multiply-add per lane.  The gathers of B are the traffic:
//       4k bytes per nonzero against 8 for the matrix entry.
//   spmm_core_kernel         PlusMultiplies f32 only: the MFMA path.  Where the matrix has dense
//       tiles -- the hub x hub core of a power-law graph: the top-H rows by length x the top-H
//       columns by reference count, cut into 16 x 16 tiles, those with >= kCoreMinNnz entries
//       stored densely -- a tile's product with its 16 rows of B is v_mfma_f32_16x16x4_f32
//       (exact f32, an fmaf chain): the 16 rows of B are read once per TILE instead of once per
//       NONZERO.  The entries of stored tiles are taken out of the CSR the tile kernel walks.
//       Everything else is not dense enough for a matrix core to help (measurements in DESIGN.md).
//
// Row sums are formed in the CSR order by the tile kernel; a core row then adds its tile products
// (a different association than plain CSR order: results agree to rounding, exactly on integer data).
#include "common.hpp"

#include <algorithm>
#include <numeric>

namespace grb {

constexpr int kSpmmTile = 512;        // must equal the SpMV plan's wave tile
constexpr int kCoreTile = 16;
constexpr int kCoreMinNnz = 24;       // a 16 x 16 tile is stored densely from this many entries

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

// ---- MFMA dense-core path ---------------------------------------------------------------
// One wave per tile row (16 core rows): for every stored tile of that tile row and every block of 16
// output columns, four v_mfma_f32_16x16x4_f32 (K = 16 in four steps).  Operand maps (CDNA4 guide,
// section 3): A lane l -> A[i = l & 15][kk = l >> 4], B lane l -> B[kk = l >> 4][j = l & 15],
// D reg q of lane l -> D[row = 4 (l >> 4) + q][col = l & 15].  Tiles are stored transposed
// ([kk][i]) so that the A operand of a K step is one coalesced 256-byte read.
// Restriction (why the path is opt-in, GRB_SPMM_CORE): a stored tile is dense, so the matrix core multiplies the
// tile's explicit zeros with rows of B -- finite B only (0 x Inf = NaN would reach core rows that have no entry in
// that column; the CSR tile kernel never forms such products), and duplicate entries of one cell were summed when
// the tiles were built, so the result can differ from the other path by more than rounding in those two cases.
__global__ __launch_bounds__(kBlock) void spmm_core_kernel(const int* __restrict__ trow_ptr, const int* __restrict__ tcol,
                                                           const float* __restrict__ tvals,
                                                           const Index* __restrict__ core_rows,
                                                           const Index* __restrict__ core_cols, int ntrows,
                                                           const float* __restrict__ B, float* __restrict__ C, Index k) {
  const int lane = lane_id();
  const int nwaves = gridDim.x * kWavesPerBlock;
  const int li = lane & 15, lk = lane >> 4;
  for (int tr = blockIdx.x * kWavesPerBlock + wave_id(); tr < ntrows; tr += nwaves) {
    const int t0 = trow_ptr[tr], t1 = trow_ptr[tr + 1];
    if (t0 == t1) continue;
    for (Index cb = 0; cb < k; cb += 64) {
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int t = t0; t < t1; ++t) {
        const float* tile = tvals + (size_t)t * (kCoreTile * kCoreTile);
        const Index* cols = core_cols + (size_t)tcol[t] * kCoreTile;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float a = tile[(ks * 4 + lk) * kCoreTile + li];
          const Index brow = cols[ks * 4 + lk];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const Index c = cb + q * 16 + li;
            const float b = (c < k && brow >= 0) ? B[(size_t)brow * k + c] : 0.f;
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
          }
        }
      }
      // C[core row 4 lk + r][cb + 16 q + li] += acc[q][r]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const Index c = cb + q * 16 + li;
        if (c >= k) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const Index row = core_rows[(size_t)tr * kCoreTile + lk * 4 + r];
          if (row >= 0) C[(size_t)row * k + c] += acc[q][r];
        }
      }
    }
  }
}

}  // namespace grb

using namespace grb;

namespace {

void free_core(SpmmCore* core) {
  for (void* p : {(void*)core->d_trow_ptr, (void*)core->d_tcol, (void*)core->d_tvals, (void*)core->d_rows,
                  (void*)core->d_cols, (void*)core->rest.ptr, (void*)core->rest.ind, core->rest.val})
    if (p) (void)hipFree(p);
  free_spmv_plan(&core->rest_plan);
  *core = SpmmCore();
}

template <typename V>
grb_info upload_vec(const std::vector<V>& h, V** d) {
  *d = nullptr;
  const size_t n = h.empty() ? 1 : h.size();
  GRB_HIP_TRY(hipMalloc((void**)d, sizeof(V) * n));
  if (!h.empty()) GRB_HIP_TRY(hipMemcpy(*d, h.data(), sizeof(V) * h.size(), hipMemcpyHostToDevice));
  return GRB_SUCCESS;
}

// Splits one orientation of the matrix into dense core tiles + the remaining CSR (host side, once).
grb_info build_core(const std::vector<Index>& ptr, const std::vector<Index>& ind, const std::vector<uint32_t>& val,
                    Index nrows, Index ncols, int H, SpmmCore* core) {
  free_core(core);
  core->built = true;
  core->H = H;
  const size_t nnz = ind.size();
  if (nnz == 0 || H < kCoreTile) return GRB_SUCCESS;
  // top-H rows by length, top-H columns by reference count (ties: smaller id first)
  std::vector<Index> colcnt((size_t)ncols, 0);
  for (size_t i = 0; i < nnz; ++i) colcnt[ind[i]]++;
  auto top = [&](Index n, auto key, std::vector<Index>* out) {
    std::vector<Index> id((size_t)n);
    std::iota(id.begin(), id.end(), 0);
    const size_t h = std::min<size_t>((size_t)H, (size_t)n);
    std::partial_sort(id.begin(), id.begin() + h, id.end(), [&](Index x, Index y) {
      const Index kx = key(x), ky = key(y);
      return kx != ky ? kx > ky : x < y;
    });
    id.resize(h);
    while (!id.empty() && key(id.back()) == 0) id.pop_back();
    *out = id;
  };
  std::vector<Index> rows, cols;
  top(nrows, [&](Index r) { return ptr[(size_t)r + 1] - ptr[r]; }, &rows);
  top(ncols, [&](Index c) { return colcnt[c]; }, &cols);
  const int ntr = (int)((rows.size() + kCoreTile - 1) / kCoreTile), ntc = (int)((cols.size() + kCoreTile - 1) / kCoreTile);
  if (ntr == 0 || ntc == 0) return GRB_SUCCESS;
  std::vector<int> crank((size_t)ncols, -1);
  for (size_t i = 0; i < cols.size(); ++i) crank[cols[i]] = (int)i;
  // count entries per tile, decide which tiles are stored
  std::vector<int> tcount((size_t)ntr * ntc, 0);
  for (size_t ri = 0; ri < rows.size(); ++ri) {
    const Index r = rows[ri];
    for (Index p = ptr[r]; p < ptr[(size_t)r + 1]; ++p) {
      const int cr = crank[ind[p]];
      if (cr >= 0) tcount[(ri / kCoreTile) * ntc + cr / kCoreTile]++;
    }
  }
  std::vector<int> tindex((size_t)ntr * ntc, -1), trow_ptr((size_t)ntr + 1, 0), tcol;
  for (int a = 0; a < ntr; ++a) {
    trow_ptr[a] = (int)tcol.size();
    for (int b = 0; b < ntc; ++b)
      if (tcount[(size_t)a * ntc + b] >= kCoreMinNnz) { tindex[(size_t)a * ntc + b] = (int)tcol.size(); tcol.push_back(b); }
  }
  trow_ptr[ntr] = (int)tcol.size();
  core->ntiles = (int)tcol.size();
  if (core->ntiles == 0) return GRB_SUCCESS;
  // fill the tiles (transposed: [kk][i]) and the remaining CSR
  std::vector<float> tvals((size_t)core->ntiles * kCoreTile * kCoreTile, 0.f);
  std::vector<int> rrank((size_t)nrows, -1);
  for (size_t i = 0; i < rows.size(); ++i) rrank[rows[i]] = (int)i;
  std::vector<Index> rptr((size_t)nrows + 1, 0), rind;
  std::vector<uint32_t> rval;
  rind.reserve(nnz);
  rval.reserve(nnz);
  long long moved = 0;
  for (Index r = 0; r < nrows; ++r) {
    const int rr = rrank[r];
    for (Index p = ptr[r]; p < ptr[(size_t)r + 1]; ++p) {
      int ti = -1;
      const int cr = rr >= 0 ? crank[ind[p]] : -1;
      if (cr >= 0) ti = tindex[(size_t)(rr / kCoreTile) * ntc + cr / kCoreTile];
      if (ti >= 0) {
        float f;
        memcpy(&f, &val[p], 4);
        tvals[(size_t)ti * 256 + (size_t)(cr % kCoreTile) * kCoreTile + (rr % kCoreTile)] += f;   // duplicates add up
        ++moved;
      } else {
        rind.push_back(ind[p]);
        rval.push_back(val[p]);
      }
    }
    rptr[(size_t)r + 1] = (Index)rind.size();
  }
  core->nnz_core = moved;
  core->ntrows = ntr;
  std::vector<Index> prow((size_t)ntr * kCoreTile, -1), pcol((size_t)ntc * kCoreTile, -1);   // -1 = padding
  std::copy(rows.begin(), rows.end(), prow.begin());
  std::copy(cols.begin(), cols.end(), pcol.begin());
  GRB_TRY(upload_vec(trow_ptr, &core->d_trow_ptr));
  GRB_TRY(upload_vec(tcol, &core->d_tcol));
  GRB_TRY(upload_vec(tvals, &core->d_tvals));
  GRB_TRY(upload_vec(prow, &core->d_rows));
  GRB_TRY(upload_vec(pcol, &core->d_cols));
  GRB_TRY(upload_vec(rptr, &core->rest.ptr));
  GRB_TRY(upload_vec(rind, &core->rest.ind));
  uint32_t* dv = nullptr;
  GRB_TRY(upload_vec(rval, &dv));
  core->rest.val = dv;
  core->rest.n = nrows;
  core->rest.nvals = (Index)rind.size();
  GRB_TRY(build_spmv_plan(rptr, nrows, ncols, &core->rest_plan));
  return GRB_SUCCESS;
}

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

void grb::free_spmm_core(SpmmCore* core) { free_core(core); }

extern "C" {

grb_info grb_spmm(grb_semiring op, grb_matrix A, int tran, const void* d_B, void* d_C, grb_index k, grb_descriptor desc) { GRB_API_ENTER();
  (void)desc;
  if (!A || !d_B || !d_C) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (k < 1) return GRB_INVALID_VALUE;
  const CsrArrays& M = tran ? A->csc : A->csr;
  SpmvPlan& plan = tran ? A->plan_csc : A->plan_csr;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  GRB_TRY(ctx_init());
  // the dense-core path: PlusMultiplies f32, switched on by GRB_SPMM_CORE=<H> (rows / columns considered)
  const char* core_env = getenv("GRB_SPMM_CORE");
  const int core_h = core_env ? atoi(core_env) : 0;
  SpmmCore* core = nullptr;
  if (core_h >= kCoreTile && op == GRB_PLUS_MULTIPLIES && A->dtype == GRB_F32) {
    core = tran ? &A->spmm_core_csc : &A->spmm_core_csr;
    if (!core->built || core->H != core_h) {
      const grb_index *hp, *hi;
      const void* hv;
      GRB_TRY(tran ? grb_matrix_host_csc(A, &hp, &hi, &hv) : grb_matrix_host_csr(A, &hp, &hi, &hv));
      const std::vector<Index>& ptr = tran ? A->h_csc_ptr : A->h_csr_ptr;
      const std::vector<Index>& ind = tran ? A->h_csc_ind : A->h_csr_ind;
      const std::vector<uint32_t>& val = tran ? A->h_csc_val : A->h_csr_val;
      GRB_TRY(build_core(ptr, ind, val, tran ? A->ncols : A->nrows, tran ? A->nrows : A->ncols, core_h, core));
    }
    if (core->ntiles == 0) core = nullptr;
  }
  const CsrArrays& W = core ? core->rest : M;
  const SpmvPlan& wplan = core ? core->rest_plan : plan;
  GRB_TRY(dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    return run_tiles<SR, T>(W, wplan, (const T*)d_B, (T*)d_C, k);
  }));
  if (core) {
    hipLaunchKernelGGL(spmm_core_kernel, dim3(stream_grid((long long)core->ntrows * kWave, kBlock)), dim3(kBlock), 0,
                       ctx().stream, core->d_trow_ptr, core->d_tcol, core->d_tvals, core->d_rows, core->d_cols, core->ntrows,
                       (const float*)d_B, (float*)d_C, k);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}

// what the dense-core split of the last grb_spmm on this orientation looks like (0s when it is off)
grb_info grb_spmm_core_info(grb_matrix A, int tran, int* ntiles, int64_t* nnz_in_tiles) { GRB_API_ENTER();
  if (!A) return GRB_UNINITIALIZED_OBJECT;
  const SpmmCore& core = tran ? A->spmm_core_csc : A->spmm_core_csr;
  if (ntiles) *ntiles = core.ntiles;
  if (nnz_in_tiles) *nnz_in_tiles = core.nnz_core;
  return GRB_SUCCESS;
}

}  // extern "C"
