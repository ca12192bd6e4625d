// mxm.hip -- masked SpGEMM (the triangle-counting path) and the two
// matrix helpers its driver needs.
//
//   grb_mxm        C<mask> = A (+.x) B for the mask's nonzeros only: backend/cuda/spgemm.hpp:22-110
//                  + kernels/spgemm.hpp:17-79 (spgemmMaskedKernel).  With GrB_INP1 = GrB_TRAN the
//                  "columns of B" are B's CSR rows, so C[i,j] = (+)_k A[i,k] (x) B[j,k] -- for
//                  B = A = L this is |N(i) n N(j)| on every edge of L, whose sum is the triangle
//                  count.  The reference gives a 32-lane warp to every row and binary-searches
//                  every A entry in the B column; here the mask entries are one flat range, a
//                  64-lane wave takes 64 consecutive entries, short dot products are done by
//                  a lane each and long ones by the whole wave (walk the shorter list, binary
//                  search the longer: min(d_i, d_j) * log max(d_i, d_j) per dot product).
//                  A's own row pointers are used (the
//                  reference walks A with the MASK's row pointers, kernels/spgemm.hpp:35-36,51-56,
//                  which is only meaningful when the two share structure -- as they do in tc()).
//   grb_matrix_tril  lower triangle on the host, as the reference (tri.hpp:21-48, sequential only)
//   grb_reduce_matrix_scalar  reduce.hpp:81-91
#include "common.hpp"

namespace grb {

constexpr int kLaneDotMax = 16;     // dot products whose shorter list is longer go to the whole wave

// row of every mask entry (the entries are processed as one flat, evenly split range)
__global__ void entry_rows_kernel(const Index* __restrict__ ptr, Index nrows, Index nvals, Index* __restrict__ row_of) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index e = (Index)blockIdx.x * blockDim.x + threadIdx.x; e < nvals; e += stride) {
    Index lo = 0, hi = nrows;               // largest r with ptr[r] <= e
    while (hi - lo > 1) {
      const Index mid = lo + ((hi - lo) >> 1);
      if (ptr[mid] <= e) lo = mid; else hi = mid;
    }
    row_of[e] = lo;
  }
}

// sum of int32 entries in 64 bits (the triangle count of a com-Orkut-sized graph does not fit an int)
__global__ __launch_bounds__(kBlock) void sum_i32_wide_kernel(const int* __restrict__ d, Index n, long long* __restrict__ out) {
  long long acc = 0;
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += d[i];
  acc = wave_reduce(acc, [](long long x, long long y) { return x + y; });
  if (lane_id() == 0 && acc) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)acc);
}

__device__ inline Index lower_bound_dev(const Index* __restrict__ a, Index lo, Index hi, Index key) {
  while (lo < hi) {
    const Index mid = lo + ((hi - lo) >> 1);
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// One 64-lane wave takes 64 consecutive mask entries.  A lane computes its own dot product
// when the shorter of the two lists has at most kLaneDotMax entries (walk it, binary-search
// the longer one, searches only move right); the others are done one after the other by the
// whole wave: the shorter list spread over the lanes, each lane binary-searching the longer,
// partial sums folded with the semiring's add.  Work per dot product is
// min(d_i, d_j) * log max(d_i, d_j) either way, but no lane of a wave is left walking a hub
// row alone (RMAT-19 triangle count: 1074 ms with a lane per entry -> see DESIGN.md).
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void spgemm_masked_kernel(
    T* __restrict__ c_val, const Index* __restrict__ m_row, const Index* __restrict__ m_ind,
    const void* __restrict__ m_val, int mask_f32, const Index* __restrict__ a_ptr, const Index* __restrict__ a_ind,
    const T* __restrict__ a_val, const Index* __restrict__ b_ptr, const Index* __restrict__ b_ind,
    const T* __restrict__ b_val, Index nvals) {
  typedef Semiring<SR, T> S;
  const int lane = lane_id();
  const Index wave_global = (Index)blockIdx.x * kWavesPerBlock + wave_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index base = wave_global * kWave; base < nvals; base += nwaves * kWave) {
    const Index e = base + lane;
    const bool valid = e < nvals && mask_nonzero(m_val, mask_f32, e);
    Index ss = 0, se = 0, ls = 0, le = 0;
    bool a_short = true;
    if (valid) {
      const Index row = m_row[e], col = m_ind[e];
      const Index as = a_ptr[row], ae = a_ptr[row + 1];
      const Index bs = b_ptr[col], be = b_ptr[col + 1];
      a_short = (ae - as) <= (be - bs);
      ss = a_short ? as : bs; se = a_short ? ae : be;
      ls = a_short ? bs : as; le = a_short ? be : ae;
    }
    T acc = S::identity();
    const bool heavy = valid && (se - ss) > kLaneDotMax;
    if (valid && !heavy) {
      const Index* s_ind = a_short ? a_ind : b_ind;
      const Index* l_ind = a_short ? b_ind : a_ind;
      Index hint = ls;
      for (Index p = ss; p < se && hint < le; ++p) {
        const Index key = s_ind[p];
        const Index lo = lower_bound_dev(l_ind, hint, le, key);
        hint = lo;
        if (lo < le && l_ind[lo] == key) {
          const T av = a_short ? a_val[p] : a_val[lo];
          const T bv = a_short ? b_val[lo] : b_val[p];
          acc = S::add(S::mul(av, bv), acc);
        }
      }
    }
    unsigned long long todo = __ballot(heavy);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index css = __shfl(ss, src, kWave), cse = __shfl(se, src, kWave);
      const Index cls = __shfl(ls, src, kWave), cle = __shfl(le, src, kWave);
      const bool c_a_short = __shfl((int)a_short, src, kWave) != 0;
      const Index* s_ind = c_a_short ? a_ind : b_ind;
      const Index* l_ind = c_a_short ? b_ind : a_ind;
      T part = S::identity();
      for (Index p = css + lane; p < cse; p += kWave) {
        const Index key = s_ind[p];
        const Index lo = lower_bound_dev(l_ind, cls, cle, key);
        if (lo < cle && l_ind[lo] == key) {
          const T av = c_a_short ? a_val[p] : a_val[lo];
          const T bv = c_a_short ? b_val[lo] : b_val[p];
          part = S::add(S::mul(av, bv), part);
        }
      }
      part = wave_reduce(part, [](T x, T y) { return S::add(x, y); });
      if (lane == src) acc = part;
    }
    if (e < nvals) c_val[e] = acc;
  }
}

// Stored values (x) a scalar, or (x) a vector entry picked by the row (by_major) or by the stored
// index (kernels/ewisemult.hpp:160-237: eWiseMultKernel scalar overload, eWiseMultCSRKernel,
// eWiseMultCSCKernel -- 32-lane warp per row there, a 64-lane wave per row here).
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void matrix_scale_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                              T* __restrict__ val, Index nmajor, const T* __restrict__ vec,
                                                              T scalar, int mode /*0 scalar, 1 by major, 2 by minor*/) {
  typedef Semiring<SR, T> S;
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index r = (Index)blockIdx.x * kWavesPerBlock + wave_id(); r < nmajor; r += nwaves) {
    const Index e = ptr[r + 1];
    const T bm = mode == 1 ? vec[r] : scalar;
    for (Index p = ptr[r] + lane; p < e; p += kWave) {
      const T b = mode == 2 ? vec[ind[p]] : bm;
      val[p] = S::mul(val[p], b);
    }
  }
}

// trace(A (+).(x) B^T) = sum over rows i of (+)_k A(i,k) (x) B(i,k)   (traceKernel, kernels/trace.hpp:7-67).
// One wave per row: lanes stride over A's row and binary-search B's; as there, a missing B entry
// contributes mul(a, identity), B's value passes through an Index-typed temporary (truncated towards
// zero, trace.hpp:44-46), rows are folded with the semiring's add and the row sums are then ADDED
// (atomicAdd whatever the semiring, trace.hpp:60-61).
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void trace_kernel(T* __restrict__ out, Index nrows, const Index* __restrict__ a_ptr,
                                                       const Index* __restrict__ a_ind, const T* __restrict__ a_val,
                                                       const Index* __restrict__ b_ptr, const Index* __restrict__ b_ind,
                                                       const T* __restrict__ b_val) {
  using S = Semiring<SR, T>;
  const int lane = threadIdx.x & (kWave - 1);
  const Index wave = (Index)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / kWave);
  const Index nwaves = (Index)((gridDim.x * (unsigned)blockDim.x) / kWave);
  T total = 0;
  for (Index row = wave; row < nrows; row += nwaves) {
    const Index ab = a_ptr[row], ae = a_ptr[row + 1], bb = b_ptr[row], be = b_ptr[row + 1];
    T sum = S::identity();
    for (Index p = ab + lane; p < ae; p += kWave) {
      const Index lo = lower_bound_dev(b_ind, bb, be, a_ind[p]);
      Index bv = (Index)S::identity();
      if (lo < be && b_ind[lo] == a_ind[p]) bv = (Index)b_val[lo];
      sum = S::add(sum, S::mul(a_val[p], (T)bv));
    }
    sum = wave_reduce(sum, [](T x, T y) { return S::add(x, y); });
    total += sum;
  }
  if (lane == 0 && total != 0) atomicAdd(out, total);
}

}  // namespace grb

using namespace grb;

static grb_info matrix_scale(grb_matrix A, int op, grb_vector B, double scalar, bool by_row) {
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (!A->owned) return GRB_INVALID_OBJECT;             // adopted storage belongs to the caller
  hipStream_t s = ctx().stream;
  const void* vec = nullptr;
  if (B) {
    if (B->dtype != A->dtype) return GRB_DOMAIN_MISMATCH;
    if (B->nsize != (by_row ? A->nrows : A->ncols)) return GRB_DIMENSION_MISMATCH;
    if (B->vec_type != GRB_DENSE) return GRB_NOT_IMPLEMENTED;   // callers densify (extractTuples semantics)
    vec = B->d_val;
  }
  grb_info info = dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    for (int o = 0; o < 2; ++o) {
      const CsrArrays& M = o == 0 ? A->csr : A->csc;
      if (!M.ptr || M.nvals == 0) continue;
      // CSR rows are matrix rows, CSC "rows" are matrix columns
      const int mode = !B ? 0 : ((o == 0) == by_row ? 1 : 2);
      hipLaunchKernelGGL((matrix_scale_kernel<SR, T>), dim3(stream_grid((long long)M.n * kWave, kBlock)), dim3(kBlock), 0,
                         s, M.ptr, M.ind, (T*)M.val, M.n, (const T*)vec, (T)scalar, mode);
      GRB_HIP_TRY(hipGetLastError());
    }
    return GRB_SUCCESS;
  });
  A->h_csr_val.clear(); A->h_csr_ind.clear();           // host mirrors are re-read on demand
  A->h_csc_val.clear(); A->h_csc_ind.clear();
  A->nonneg_values = -1; A->mean_value = -1.0; A->small_int_values = -1;
  return info;
}

extern "C" {

grb_info grb_mxm(grb_matrix C, grb_matrix mask, grb_accum accum, grb_semiring op, grb_matrix A, grb_matrix B,
                 grb_descriptor desc) {
  (void)accum;
  if (!C || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built || !B->built) return GRB_UNINITIALIZED_OBJECT;
  if (!mask) return GRB_NOT_IMPLEMENTED;                 // unmasked SpGEMM is a cuSPARSE call in the reference
  if (!mask->built) return GRB_UNINITIALIZED_OBJECT;
  if (C == A || C == B || C == mask) return GRB_NOT_IMPLEMENTED;
  if (A->dtype != B->dtype || C->dtype != A->dtype) return GRB_DOMAIN_MISMATCH;
  const bool tran_a = desc->desc[GRB_INP0] == GRB_TRAN;
  const bool tran_b = desc->desc[GRB_INP1] == GRB_TRAN;
  const CsrArrays& Aa = tran_a ? A->csc : A->csr;
  const CsrArrays& Bb = tran_b ? B->csr : B->csc;        // "columns of B"
  if (!Aa.ptr || !Bb.ptr || !mask->csr.ptr) return GRB_INVALID_OBJECT;
  if (Aa.n != mask->nrows || C->nrows != mask->nrows || C->ncols != mask->ncols) return GRB_DIMENSION_MISMATCH;
  hipStream_t s = ctx().stream;
  // C takes the mask's structure (C->dup(&mask->sparse_), spgemm.hpp:78-79)
  // whatever C held before goes, with the per-graph side arrays that described it (skip bitmaps,
  // pull hint, plans): a later traversal of C must not see hints of another graph
  matrix_release_device(C);
  C->owned = true;
  C->nvals = mask->nvals;
  const size_t cap = mask->nvals > 0 ? (size_t)mask->nvals : 1;
  GRB_HIP_TRY(hipMalloc((void**)&C->csr.ptr, 4 * ((size_t)mask->nrows + 1)));
  GRB_HIP_TRY(hipMalloc((void**)&C->csr.ind, 4 * cap));
  GRB_HIP_TRY(hipMalloc(&C->csr.val, 4 * cap));
  GRB_HIP_TRY(hipMemcpyAsync(C->csr.ptr, mask->csr.ptr, 4 * ((size_t)mask->nrows + 1), hipMemcpyDeviceToDevice, s));
  if (mask->nvals > 0)
    GRB_HIP_TRY(hipMemcpyAsync(C->csr.ind, mask->csr.ind, 4 * (size_t)mask->nvals, hipMemcpyDeviceToDevice, s));
  C->csr.n = mask->nrows;
  C->csr.nvals = mask->nvals;
  C->h_csr_ptr = mask->h_csr_ptr;
  C->h_csr_ind.clear(); C->h_csr_val.clear();
  C->h_csc_ptr.clear(); C->h_csc_ind.clear(); C->h_csc_val.clear();
  GRB_TRY(build_spmv_plan(C->h_csr_ptr, C->nrows, C->ncols, &C->plan_csr));   // mxv on the result works;
  C->built = true;                               // no CSC is made (as little as the reference's C->dup has one):
  if (mask->nvals == 0) return GRB_SUCCESS;      // products on the transpose return GrB_INVALID_OBJECT
  void* p_rows;
  GRB_TRY(scratch(9, 4 * (size_t)mask->nvals, &p_rows));            // not 4 / 5: those hold the push path's state
  hipLaunchKernelGGL(entry_rows_kernel, dim3(stream_grid(mask->nvals, kBlock)), dim3(kBlock), 0, s, mask->csr.ptr,
                     mask->nrows, mask->nvals, (Index*)p_rows);
  GRB_HIP_TRY(hipGetLastError());
  const int grid = stream_grid(mask->nvals, kBlock);
  return dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((spgemm_masked_kernel<SR, T>), dim3(grid), dim3(kBlock), 0, s, (T*)C->csr.val,
                       (const Index*)p_rows, mask->csr.ind, mask->csr.val, mask->dtype == GRB_F32, Aa.ptr, Aa.ind,
                       (const T*)Aa.val, Bb.ptr, Bb.ind, (const T*)Bb.val, mask->nvals);
    GRB_HIP_TRY(hipGetLastError());
    return GRB_SUCCESS;
  });
}

// eWiseMult, matrix (x) broadcast scalar (operations.hpp:206-228), in place (C == A)
grb_info grb_matrix_eWiseMult_scalar(grb_matrix C, grb_semiring op, grb_matrix A, double val) {
  if (!C || !A) return GRB_UNINITIALIZED_OBJECT;
  if (C != A) return GRB_NOT_IMPLEMENTED;
  return matrix_scale(A, op, nullptr, val, true);
}

// eWiseMult, matrix (x) broadcast vector (operations.hpp:240-267): C(i,j) = A(i,j) (x) B(i), or
// B(j) with GrB_INP1 = GrB_TRAN; in place (C == A)
grb_info grb_matrix_eWiseMult_vector(grb_matrix C, grb_semiring op, grb_matrix A, grb_vector B, grb_descriptor desc) {
  if (!C || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (C != A) return GRB_NOT_IMPLEMENTED;
  if (desc->desc[GRB_INP0] != GRB_DEFAULT) return GRB_INVALID_VALUE;
  return matrix_scale(A, op, B, 0.0, desc->desc[GRB_INP1] != GRB_TRAN);
}

grb_info grb_reduce_matrix_scalar(double* val, grb_accum accum, grb_monoid op, grb_matrix A, grb_descriptor desc) {
  (void)accum;
  if (!val || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (desc->struconly) { *val = (double)A->nvals; return GRB_SUCCESS; }    // reduce.hpp:86-87
  return k_reduce(op, A->dtype, A->csr.val, A->nvals, val);
}

grb_info grb_matrix_tril(grb_matrix C, grb_matrix A, grb_descriptor desc) {
  if (!C || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (C->nrows != A->nrows || C->ncols != A->ncols) return GRB_DIMENSION_MISMATCH;
  const grb_index *ptr, *ind;
  const void* val;
  GRB_TRY(grb_matrix_host_csr(A, &ptr, &ind, &val));
  const uint32_t* v = (const uint32_t*)val;
  std::vector<Index> nptr((size_t)A->nrows + 1, 0), nind;
  std::vector<uint32_t> nval;
  for (Index r = 0; r < A->nrows; ++r) {
    for (Index p = ptr[r]; p < ptr[r + 1]; ++p)
      if (ind[p] <= r) { nind.push_back(ind[p]); nval.push_back(v[p]); }    // keep row >= col (tri.hpp:33-40)
    nptr[(size_t)r + 1] = (Index)nind.size();
  }
  static const Index kZero = 0;
  static const uint32_t kZeroV = 0;
  return grb_matrix_build_csr(C, nptr.data(), nind.empty() ? &kZero : nind.data(), nval.empty() ? &kZeroV : nval.data(),
                              (Index)nind.size(), nullptr, nullptr, nullptr);
}

// traceMxmTranspose (extension, operations.hpp:698-711 -> backend :1076-1108, trace.hpp:10-52)
grb_info grb_trace_mxm_transpose(double* val, grb_semiring op, grb_matrix A, grb_matrix B, grb_descriptor desc) {
  if (!val || !A || !B) return GRB_UNINITIALIZED_OBJECT;
  (void)desc;
  if (!A->built || !B->built) return GRB_UNINITIALIZED_OBJECT;
  if (A->dtype != B->dtype) return GRB_DOMAIN_MISMATCH;
  if (A->nrows != B->nrows) return GRB_DIMENSION_MISMATCH;
  Context& c = ctx();
  void* d_out;
  GRB_TRY(scratch(10, 256, &d_out));
  GRB_HIP_TRY(hipMemsetAsync(d_out, 0, 8, c.stream));
  const Index n = A->nrows;
  if (n > 0) {
    const int grid = stream_grid((long long)n * kWave, kBlock);
    GRB_TRY(dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
      using T = decltype(t);
      constexpr int SR = decltype(tag)::value;
      hipLaunchKernelGGL((trace_kernel<SR, T>), dim3(grid), dim3(kBlock), 0, c.stream, (T*)d_out, n, A->csr.ptr,
                         A->csr.ind, (const T*)A->csr.val, B->csr.ptr, B->csr.ind, (const T*)B->csr.val);
      GRB_HIP_TRY(hipGetLastError());
      return GRB_SUCCESS;
    }));
  }
  int bits = 0;
  GRB_TRY(fetch_ints((const int*)d_out, 1, &bits));
  if (A->dtype == GRB_F32) { float f; memcpy(&f, &bits, 4); *val = (double)f; }
  else *val = (double)bits;
  return GRB_SUCCESS;
}

// algorithm::tc (algorithm/tc.hpp:15-54): B = (A x A^T) .* A on the lower triangle, ntris = sum(B)
grb_info grb_tc(int64_t* ntris, grb_matrix A, grb_matrix B, grb_descriptor desc, grb_algo_result* result) {
  if (!ntris || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  float ms = 0.f;
  grb_descriptor_toggle(desc, GRB_INP1);
  grb_info info = grb_timer_start();
  if (info == GRB_SUCCESS) info = grb_mxm(B, A, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, A, A, desc);
  double sum = 0;
  long long wide = 0;
  if (info == GRB_SUCCESS && B->dtype == GRB_I32 && !desc->struconly) {
    // reduce<int, int>(ntris, ...) in the reference (tc.hpp:41-42) wraps beyond 2^31 triangles; this entry
    // point returns the count in 64 bits (the frontend's int* overload keeps the reference's int)
    void* p_sum;
    info = scratch(10, 8, &p_sum);
    if (info == GRB_SUCCESS && hipMemsetAsync(p_sum, 0, 8, ctx().stream) != hipSuccess) info = GRB_PANIC;
    if (info == GRB_SUCCESS && B->nvals > 0) {
      hipLaunchKernelGGL(sum_i32_wide_kernel, dim3(stream_grid(B->nvals, kBlock * 8)), dim3(kBlock), 0, ctx().stream,
                         (const int*)B->csr.val, B->nvals, (long long*)p_sum);
      if (hipGetLastError() != hipSuccess) info = GRB_PANIC;
    }
    if (info == GRB_SUCCESS && (hipMemcpyAsync(&wide, p_sum, 8, hipMemcpyDeviceToHost, ctx().stream) != hipSuccess ||
                                hipStreamSynchronize(ctx().stream) != hipSuccess))
      info = GRB_PANIC;
    sum = (double)wide;
  } else if (info == GRB_SUCCESS) {
    info = grb_reduce_matrix_scalar(&sum, GRB_ACCUM_NULL, GRB_PLUS_MONOID, B, desc);
    wide = (long long)sum;
  }
  if (info == GRB_SUCCESS) info = grb_timer_stop(&ms);
  // (the reference leaves GrB_INP1 toggled: tc.hpp:23 has no matching toggle back)
  *ntris = (int64_t)wide;
  if (result) { result->iterations = 1; result->tight_ms = ms; result->last_value = sum; }
  return info;
}

}  // extern "C"
