// spmspv.hip -- the push half of mxv/vxm on gfx950 (SpMSpV).
//
// Replaces backend/cuda/spmspv.hpp:15-257 + spmspv_inner.hpp:62-320, i.e. the
// reference's  degree gather -> scan -> IntervalGather/Expand -> eWiseMult ->
// radix sort -> ReduceByKey -> mask -> flag -> scan -> compact  chain (>= 8 launches,
// three host-returning scans, a multi-pass sort over every expanded edge) with
//
//   1. push_degree_kernel   frontier degrees + per-tile exclusive scan          (nf)
//   2. scan_tiles (shared)  scan of tile sums, total stays on the device        (nf/1024)
//   3. push_expand_kernel   edge-balanced expansion: every workgroup takes equal
//                           chunks of the EXPANDED edge space, finds its frontier
//                           range by binary search on the scan (LDS-staged), reads the
//                           neighbour ids coalesced along rows, applies the mask and
//                           combines into an n-sized accumulator + touched bitmap
//                           (atomicOr for structure, atomic min/max/add or CAS for values)
//   4. bitmap compaction    ordered scan of the n/32-word bitmap -> sorted unique
//                           indices (+ gathered values), accumulator and bitmap reset
//
// so duplicates are merged by the accumulator instead of a sort, and the output is
// ordered by index exactly as the reference's sorted+reduced list is.
//
// Semantics kept: mul_op(A_val, u_val) with the identity short-circuit
// of kernels/ewisemult.hpp:22-25; masked entries are dropped; in key-value mode with a
// mask every entry whose reduced value == 0 is dropped too (spmspv.hpp:201-243);
// structure-only output carries indices only.
#include "push_common.hpp"

namespace grb {

// Monoid-specific atomic combine into the accumulator.
template <int M, typename T>
__device__ inline void atomic_combine(T* addr, T v) {
  if constexpr (M < 0) {             // a semiring registered at run time: CAS loop with its add
    unsigned int* a = reinterpret_cast<unsigned int*>(addr);
    unsigned int old = *a, assumed;
    do {
      assumed = old;
      T cur;
      memcpy(&cur, &assumed, 4);
      T nv = Semiring<GRB_RUNTIME_SR, T>::add(cur, v);
      unsigned int nb;
      memcpy(&nb, &nv, 4);
      old = atomicCAS(a, assumed, nb);
    } while (old != assumed);
    return;
  } else {
  constexpr int op = MonoidTraits<M>::op;
  if constexpr (op == OP_PLUS) {
    atomicAdd(addr, v);
  } else if constexpr (op == OP_MIN) {
    atomicMin(addr, v);
  } else if constexpr (op == OP_MAX) {
    atomicMax(addr, v);
  } else if constexpr (op == OP_LOR) {
    if (v != (T)0) *addr = (T)1;       // idempotent store: benign race
  } else {
    // generic CAS loop on the 4-byte word
    unsigned int* a = reinterpret_cast<unsigned int*>(addr);
    unsigned int old = *a, assumed;
    do {
      assumed = old;
      T cur;
      memcpy(&cur, &assumed, 4);
      T nv = Monoid<M, T>::add(cur, v);
      unsigned int nb;
      memcpy(&nb, &nv, 4);
      old = atomicCAS(a, assumed, nb);
    } while (old != assumed);
  }
  }
}

// ---- 3. per-edge visitor of lb_expand_kernel: mask (peek), multiply + combine (visit)
template <int SR, typename T, bool kStruc>
struct SpmspvVisitor {
  const T* val; const T* u_val; const void* mask; int mask_f32, use_mask, keep_when_zero;
  unsigned int* touched; T* acc;
  __device__ bool peek(Index dst) const {
    if (use_mask && ((mask_nonzero(mask, mask_f32, dst) ? 1 : 0) == keep_when_zero)) return false;
    if constexpr (kStruc) return !((touched[dst >> 5] >> (dst & 31)) & 1u);   // already recorded
    return true;
  }
  __device__ void visit(Index k, Index p, Index dst) const {
    typedef Semiring<SR, T> S;
    const unsigned int bit = 1u << (dst & 31);
    if constexpr (!kStruc) {
      const T a = val[p], x = u_val[k];
      const T ident = S::identity();
      const T prod = (a == ident || x == ident) ? ident : S::mul(a, x);   // kernels/ewisemult.hpp:22-25
      atomic_combine<S::monoid, T>(&acc[dst], prod);
      if ((touched[dst >> 5] & bit)) return;
    }
    atomicOr(&touched[dst >> 5], bit);
  }
};

// ---- 4. ordered bitmap compaction (one word per thread), resets bitmap + accumulator
template <typename T, bool kStruc>
__global__ void bitmap_write_kernel(unsigned int* __restrict__ words, int nwords,
                                    const int* __restrict__ tile_off, T* __restrict__ acc, T identity,
                                    int drop_zero, Index* __restrict__ out_ind, T* __restrict__ out_val,
                                    int* __restrict__ dropped) {
  __shared__ int smem[kWavesPerBlock];
  int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned int wd = i < nwords ? words[i] : 0u;
  int tot;
  int pos = tile_off[blockIdx.x] + block_exclusive_scan(__popc(wd), smem, tot);
  if (wd) words[i] = 0u;
  while (wd) {
    int b = __ffs((int)wd) - 1;
    wd &= wd - 1;
    Index v = (Index)i * 32 + b;
    if constexpr (kStruc) {
      out_ind[pos++] = v;
    } else {
      T x = acc[v];
      acc[v] = identity;
      out_ind[pos] = v;
      out_val[pos] = x;
      ++pos;
      if (drop_zero && x == (T)0) atomicAdd(dropped, 1);
    }
  }
}

// use_mask: 0 none, 1 dense mask applied, 2 masked epilogue only (the reference's sparse-mask case:
// nothing is filtered, key-value mode still prunes zeros)
grb_info k_spmspv(int sr, int dtype, const CsrArrays& M, Index out_size, int struconly, const Index* u_ind,
                  const void* u_val, Index nf, const void* mask, int mask_f32, int use_mask,
                  int keep_when_mask_zero, Index* w_ind, void* w_val, Index* w_nvals) {
  *w_nvals = 0;
  if (nf <= 0) return GRB_SUCCESS;
  hipStream_t s = ctx().stream;
  const int ntiles = ceil_div(nf, kDegTile);
  const int nwords = ceil_div(out_size, 32);
  const int wtiles = ceil_div(nwords, kBlock);
  void *p_scan, *p_tiles, *p_bitmap, *p_acc, *p_btiles, *p_rs;
  const long long max_edges = (long long)M.nvals;
  GRB_TRY(scratch(2, sizeof(int) * (size_t)nf, &p_scan));
  GRB_TRY(scratch(11, sizeof(int) * (size_t)nf, &p_rs));
  GRB_TRY(scratch(3, sizeof(int) * (size_t)(2 * ntiles + 2), &p_tiles));
  GRB_TRY(scratch(6, sizeof(int) * (size_t)(2 * wtiles + 2 + max_edges / kEdgeChunk + 4), &p_btiles));
  // slots 4 (bitmap) and 5 (accumulator) are persistent: kept zero / identity between calls
  size_t old_bitmap_cap = ctx().slot_cap[4];
  GRB_TRY(scratch(4, sizeof(unsigned int) * (size_t)nwords, &p_bitmap));
  if (ctx().slot_cap[4] != old_bitmap_cap)
    GRB_HIP_TRY(hipMemsetAsync(p_bitmap, 0, ctx().slot_cap[4], s));
  int* local_scan = (int*)p_scan;
  Index* row_start = (Index*)p_rs;
  int* tile_sums = (int*)p_tiles;
  int* tile_off = tile_sums + ntiles;            // ntiles + 1 entries
  int* btile_counts = (int*)p_btiles;
  int* btile_off = btile_counts + wtiles;        // wtiles + 1 entries
  Index* chunk_owner = (Index*)(btile_off + wtiles + 2);
  int* d_mail = ctx().d_mail;                    // [0] expanded edges, [1] output count, [2] dropped
  GRB_HIP_TRY(hipMemsetAsync(d_mail + 2, 0, sizeof(int), s));

  GRB_TRY(dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    typedef Semiring<SR, T> S;
    T* acc = nullptr;
    if (!struconly) {
      // accumulator slot: identity-filled on (re)allocation or when the identity changes
      GRB_TRY(scratch(5, sizeof(T) * (size_t)out_size, &p_acc));
      acc = (T*)p_acc;
      double ident = (double)S::identity();
      Context& c = ctx();
      if (c.slot_cap[5] != c.acc_cap || c.acc_identity != ident || c.acc_dtype != dtype) {
        GRB_TRY(k_fill(dtype, acc, ident, (Index)(c.slot_cap[5] / sizeof(T))));
        c.acc_cap = c.slot_cap[5];
        c.acc_identity = ident;
        c.acc_dtype = dtype;
      }
    }
    if (struconly) {
      SpmspvVisitor<SR, T, true> vis{(const T*)M.val, (const T*)u_val, mask, mask_f32, use_mask == 1 ? 1 : 0,
                                     keep_when_mask_zero, (unsigned int*)p_bitmap, acc};
      GRB_TRY(launch_lb_expand(s, M, u_ind, nf, max_edges, local_scan, row_start, tile_sums, tile_off, chunk_owner,
                               d_mail, vis));
    } else {
      SpmspvVisitor<SR, T, false> vis{(const T*)M.val, (const T*)u_val, mask, mask_f32, use_mask == 1 ? 1 : 0,
                                      keep_when_mask_zero, (unsigned int*)p_bitmap, acc};
      GRB_TRY(launch_lb_expand(s, M, u_ind, nf, max_edges, local_scan, row_start, tile_sums, tile_off, chunk_owner,
                               d_mail, vis));
    }
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(wtiles), dim3(kBlock), 0, s, (const unsigned int*)p_bitmap,
                       (const unsigned int*)nullptr, nwords, btile_counts);
    GRB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, btile_counts, wtiles, btile_off,
                       d_mail + 1);
    GRB_HIP_TRY(hipGetLastError());
    const int drop_zero = (!struconly && use_mask) ? 1 : 0;
    if (struconly)
      hipLaunchKernelGGL((bitmap_write_kernel<T, true>), dim3(wtiles), dim3(kBlock), 0, s,
                         (unsigned int*)p_bitmap, nwords, btile_off, acc, S::identity(), 0, w_ind, (T*)w_val,
                         d_mail + 2);
    else
      hipLaunchKernelGGL((bitmap_write_kernel<T, false>), dim3(wtiles), dim3(kBlock), 0, s,
                         (unsigned int*)p_bitmap, nwords, btile_off, acc, S::identity(), drop_zero, w_ind,
                         (T*)w_val, d_mail + 2);
    GRB_HIP_TRY(hipGetLastError());
    return GRB_SUCCESS;
  }));
  int h[3] = {0, 0, 0};
  GRB_TRY(fetch_ints(d_mail, 3, h));
  Index nv = h[1];
  if (!struconly && use_mask && h[2] > 0) {
    // masked key-value mode also prunes reduced values == 0 (spmspv.hpp:229-243)
    GRB_TRY(k_sparse_prune(dtype, w_ind, w_val, nv, 0.0, &nv));
  }
  *w_nvals = nv;
  return GRB_SUCCESS;
}

}  // namespace grb
