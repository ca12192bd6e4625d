// elementwise.hip -- streaming primitives of the hot path: fill / scatter / assign /
// eWiseAdd / eWiseMult / reduce / stream compaction.  All HBM-bound: one coalesced
// pass per operand, grid-stride over <= 2048 workgroups of 256 threads (4 waves).
//
// Reference kernels replaced (semantics only; none of this is a translation):
//   kernels/util.hpp (zeroKernel, updateFlagKernel, streamCompact*, scatter, countZero)
//   kernels/assign_dense.hpp, kernels/assign_sparse.hpp
//   kernels/ewiseadd.hpp, kernels/ewisemult.hpp
//   reduce.hpp (cub::DeviceReduce / DeviceSegmentedReduce call sites)
// The reference's flag + scan + compact triple (3 kernels + host-returning scan) is a
// count / scan-of-tiles / write triple here whose output is ordered by index, with the
// total staying on the device until the caller asks for it.
#include "common.hpp"

namespace grb {

template <typename F>
static inline grb_info dispatch_dtype(int dtype, F&& f) {
  if (dtype == GRB_F32) return f(float{});
  if (dtype == GRB_I32) return f(int{});
  return GRB_DOMAIN_MISMATCH;
}

#define GRB_LAUNCH_CHECK() GRB_HIP_TRY(hipGetLastError())

// ---------------------------------------------------------------- fill / scatter
template <typename T>
__global__ void fill_kernel(T* __restrict__ d, T val, Index n) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = val;
}
template <typename T>
__global__ void fill_ascending_kernel(T* __restrict__ d, Index n) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = (T)i;
}
template <typename T>
__global__ void scatter_const_kernel(T* __restrict__ d, const Index* __restrict__ ind, T val, Index n) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[ind[i]] = val;
}
template <typename T>
__global__ void scatter_vals_kernel(T* __restrict__ d, const Index* __restrict__ ind,
                                    const T* __restrict__ vals, Index n) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[ind[i]] = vals[i];
}

grb_info k_fill(int dtype, void* d, double val, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(fill_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)d, (T)val, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}
grb_info k_fill_ascending(int dtype, void* d, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(fill_ascending_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)d, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}
grb_info k_scatter_const(int dtype, void* d, const Index* ind, double val, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(scatter_const_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)d, ind, (T)val, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}
grb_info k_scatter_vals(int dtype, void* d, const Index* ind, const void* vals, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(scatter_vals_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)d, ind, (const T*)vals, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---------------------------------------------------------------- compaction
// Tile = kBlock threads x kItems consecutive elements per thread.
constexpr int kItems = 4;
constexpr int kTile = kBlock * kItems;

// flag(i): element i survives.
template <typename T> struct DenseNeq {       // dense vector, keep val != identity
  const T* val; T identity;
  __device__ bool flag(Index i) const { return val[i] != identity; }
};
template <typename T> struct SparseNeq {      // sparse list, keep val != prune
  const T* val; T prune;
  __device__ bool flag(Index i) const { return val[i] != prune; }
};

template <typename P>
__global__ void compact_count_kernel(P pred, Index n, int* __restrict__ tile_counts) {
  __shared__ int smem[kWavesPerBlock];
  const Index base = (Index)blockIdx.x * kTile + threadIdx.x * kItems;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    Index i = base + k;
    if (i < n && pred.flag(i)) ++c;
  }
  c = wave_reduce(c, [](int a, int b) { return a + b; });
  if (lane_id() == 0) smem[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w];
    tile_counts[blockIdx.x] = t;
  }
}

// Exclusive scan of `ntiles` counts by ONE workgroup; total -> *total_out.
__global__ void scan_tiles_kernel(const int* __restrict__ counts, int ntiles, int* __restrict__ offsets,
                                  int* __restrict__ total_out) {
  __shared__ int smem[kWavesPerBlock];
  int carry = 0;
  for (int base = 0; base < ntiles; base += kBlock) {
    int i = base + threadIdx.x;
    int v = i < ntiles ? counts[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, smem, tot);
    if (i < ntiles) offsets[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

template <typename T, typename P>
__global__ void compact_write_kernel(P pred, Index n, const int* __restrict__ tile_offsets,
                                     const Index* __restrict__ src_ind /*nullable: use i*/,
                                     const T* __restrict__ src_val /*nullable*/,
                                     Index* __restrict__ out_ind, T* __restrict__ out_val /*nullable*/) {
  __shared__ int smem[kWavesPerBlock];
  const Index base = (Index)blockIdx.x * kTile + threadIdx.x * kItems;
  bool f[kItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    Index i = base + k;
    f[k] = (i < n) && pred.flag(i);
    c += f[k] ? 1 : 0;
  }
  int tot;
  int pos = tile_offsets[blockIdx.x] + block_exclusive_scan(c, smem, tot);
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    if (f[k]) {
      Index i = base + k;
      out_ind[pos] = src_ind ? src_ind[i] : i;
      if (out_val) out_val[pos] = src_val[i];
      ++pos;
    }
  }
}

template <typename T, typename P>
static grb_info run_compaction(P pred, Index n, const Index* src_ind, const T* src_val, Index* out_ind,
                               T* out_val, int* d_total) {
  int ntiles = ceil_div(n, kTile);
  void* p = nullptr;
  GRB_TRY(scratch(0, sizeof(int) * (size_t)(2 * ntiles + 2), &p));
  int* counts = (int*)p;
  int* offsets = counts + ntiles;
  hipStream_t s = ctx().stream;
  hipLaunchKernelGGL(compact_count_kernel<P>, dim3(ntiles), dim3(kBlock), 0, s, pred, n, counts);
  GRB_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, counts, ntiles, offsets, d_total);
  GRB_LAUNCH_CHECK();
  hipLaunchKernelGGL((compact_write_kernel<T, P>), dim3(ntiles), dim3(kBlock), 0, s, pred, n, offsets,
                     src_ind, src_val, out_ind, out_val);
  GRB_LAUNCH_CHECK();
  return GRB_SUCCESS;
}

grb_info k_dense2sparse(int dtype, const void* d_dense, double identity, Index n, Index* out_ind,
                        void* out_val, Index* nvals_out) {
  if (n <= 0) { *nvals_out = 0; return GRB_SUCCESS; }
  int* d_total = ctx().d_mail;
  GRB_TRY(dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    DenseNeq<T> pred{(const T*)d_dense, (T)identity};
    return run_compaction<T>(pred, n, (const Index*)nullptr, (const T*)d_dense, out_ind, (T*)out_val, d_total);
  }));
  int tot = 0;
  GRB_TRY(fetch_ints(d_total, 1, &tot));
  *nvals_out = tot;
  return GRB_SUCCESS;
}

grb_info k_sparse_prune(int dtype, Index* ind, void* val, Index n, double prune_val, Index* nvals_out) {
  if (n <= 0) { *nvals_out = 0; return GRB_SUCCESS; }
  int* d_total = ctx().d_mail;
  void* tmp = nullptr;
  GRB_TRY(scratch(1, (size_t)n * 8, &tmp));
  Index* t_ind = (Index*)tmp;
  void* t_val = (char*)tmp + (size_t)n * 4;
  GRB_TRY(dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    SparseNeq<T> pred{(const T*)val, (T)prune_val};
    return run_compaction<T>(pred, n, ind, (const T*)val, t_ind, (T*)t_val, d_total);
  }));
  int tot = 0;
  GRB_TRY(fetch_ints(d_total, 1, &tot));
  if (tot > 0) {
    GRB_TRY(k_copy(ind, t_ind, (size_t)tot * 4));
    GRB_TRY(k_copy(val, t_val, (size_t)tot * 4));
  }
  *nvals_out = tot;
  return GRB_SUCCESS;
}

// ---------------------------------------------------------------- device-to-device copy
// hipMemcpyAsync D2D goes through a runtime copy kernel that reaches about 1 TB/s here (33 us for
// a 16 MB vector); a plain 16-byte grid-stride copy runs at the streaming rate.
__global__ __launch_bounds__(kBlock) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

grb_info k_copy(void* dst, const void* src, size_t bytes) {
  if (bytes == 0 || dst == src) return GRB_SUCCESS;
  hipStream_t s = ctx().stream;
  const bool aligned = (((uintptr_t)dst | (uintptr_t)src) & 15u) == 0;
  const size_t n16 = aligned ? bytes / 16 : 0;
  if (n16) {
    hipLaunchKernelGGL(copy16_kernel, dim3(stream_grid((long long)n16, kBlock)), dim3(kBlock), 0, s, (uint4*)dst,
                       (const uint4*)src, n16);
    GRB_LAUNCH_CHECK();
  }
  if (bytes > n16 * 16)
    GRB_HIP_TRY(hipMemcpyAsync((char*)dst + n16 * 16, (const char*)src + n16 * 16, bytes - n16 * 16,
                               hipMemcpyDeviceToDevice, s));
  return GRB_SUCCESS;
}

// ---------------------------------------------------------------- count / reduce
// One launch: per-workgroup partials, an arrival ticket, and the last workgroup to arrive
// folds the partials in a fixed order (deterministic for a given grid) and writes the result
// straight into the pinned host mailbox as a {value, seq} granule -- no second launch, no
// publish kernel, no stream synchronise (the hand-off is fence-free: write-through partials,
// s_waitcnt, ticket; CDNA guide G16 R1).
template <int M, typename T, bool kCountNeq>
__global__ __launch_bounds__(kBlock) void reduce_kernel(const T* __restrict__ d, Index n, T cmp, unsigned int* partial,
                                                         unsigned int* ticket, unsigned long long* mail, int seq) {
  __shared__ unsigned int smem[kWavesPerBlock];
  __shared__ int s_last;
  typedef Monoid<M, T> Mo;
  auto bits = [](T x) { unsigned int u; memcpy(&u, &x, 4); return u; };
  auto from = [](unsigned int u) { T x; memcpy(&x, &u, 4); return x; };
  T acc = kCountNeq ? (T)0 : Mo::identity();
  int cnt = 0;
  // 16-byte loads, two in flight per lane, where the fold may be re-associated (counting, and the monoids that
  // are associative and commutative: plus, multiplies, minimum, maximum, logical or / and); the order-sensitive
  // "monoids" of stddef.hpp (greater, less, not_equal_to) keep the element-per-lane stride
  constexpr bool kWide = kCountNeq || M <= GRB_LOGICAL_AND_MONOID;
  Index done = 0;
  if constexpr (kWide) {
    if ((reinterpret_cast<uintptr_t>(d) & 15u) == 0) {
      struct alignas(16) Vec4 { T x, y, z, w; };
      const Vec4* __restrict__ d4 = reinterpret_cast<const Vec4*>(d);
      const Index n4 = n / 4;
      const Index stride = (Index)gridDim.x * blockDim.x;
      Index i = blockIdx.x * blockDim.x + threadIdx.x;
      auto fold = [&](const Vec4& q) {
        if constexpr (kCountNeq) cnt += (q.x != cmp) + (q.y != cmp) + (q.z != cmp) + (q.w != cmp);
        else acc = Mo::add(Mo::add(acc, Mo::add(q.x, q.y)), Mo::add(q.z, q.w));
      };
      for (; i + stride < n4; i += 2 * stride) {
        const Vec4 a = d4[i], b = d4[i + stride];
        fold(a);
        fold(b);
      }
      if (i < n4) fold(d4[i]);
      done = n4 * 4;
    }
  }
  for (Index i = done + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const T x = d[i];
    if constexpr (kCountNeq) cnt += (x != cmp) ? 1 : 0;
    else acc = Mo::add(acc, x);
  }
  if constexpr (!kCountNeq) {
    // (the same tail as a chain of element-wise calls that ends in a reduction runs: common.hpp)
    reduce_finish<T>(acc, [](T a, T b) { return Mo::add(a, b); }, Mo::identity(), partial, ticket, mail, seq, smem, &s_last);
    return;
  }
  unsigned int mine;
  if constexpr (kCountNeq) {
    cnt = wave_reduce(cnt, [](int a, int b) { return a + b; });
    mine = (unsigned int)cnt;
  } else {
    acc = wave_reduce(acc, [](T a, T b) { return Mo::add(a, b); });
    mine = bits(acc);
  }
  if (lane_id() == 0) smem[wave_id()] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = smem[0];
    for (int w = 1; w < kWavesPerBlock; ++w) {
      if constexpr (kCountNeq) t += smem[w];
      else t = bits(Mo::add(from(t), from(smem[w])));
    }
    __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = last_workgroup_arrives(ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: thread t folds partials t, t + 256, ... ; then waves, then the block
  unsigned int f = kCountNeq ? 0u : bits(Mo::identity());
  for (int j = threadIdx.x; j < (int)gridDim.x; j += kBlock) {
    const unsigned int pj = __hip_atomic_load(&partial[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (kCountNeq) f += pj;
    else f = bits(Mo::add(from(f), from(pj)));
  }
  if constexpr (kCountNeq) f = wave_reduce(f, [](unsigned int a, unsigned int b) { return a + b; });
  else f = bits(wave_reduce(from(f), [](T a, T b) { return Mo::add(a, b); }));
  __syncthreads();
  if (lane_id() == 0) smem[wave_id()] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = smem[0];
    for (int w = 1; w < kWavesPerBlock; ++w) {
      if constexpr (kCountNeq) t += smem[w];
      else t = bits(Mo::add(from(t), from(smem[w])));
    }
    __hip_atomic_store(&mail[0], ((unsigned long long)(unsigned int)seq << 32) | t, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

grb_info reduce_launch_prep(Index n, int* grid, unsigned int** d_partial, unsigned int** d_ticket) {
  *grid = stream_grid(n, kBlock * 4);
  void* p = nullptr;
  GRB_TRY(scratch(0, sizeof(int) * (size_t)(*grid), &p));
  *d_partial = (unsigned int*)p;
  *d_ticket = ctx().d_tickets;       // zero at start, reset by the kernel
  return GRB_SUCCESS;
}

grb_info k_count_nonidentity(int dtype, const void* d, double identity, Index n, Index* count_out) {
  if (n <= 0) { *count_out = 0; return GRB_SUCCESS; }
  int grid;
  unsigned int *d_partial, *d_ticket;
  GRB_TRY(reduce_launch_prep(n, &grid, &d_partial, &d_ticket));
  Context& c = ctx();
  const int seq = ++c.mail_seq;
  GRB_TRY(dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL((reduce_kernel<GRB_PLUS_MONOID, T, true>), dim3(grid), dim3(kBlock), 0, c.stream,
                       (const T*)d, n, (T)identity, d_partial, d_ticket, c.d_hgran, seq);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  }));
  unsigned int v = 0;
  GRB_TRY(wait_granules(seq, 1, &v));
  *count_out = (Index)v;
  return GRB_SUCCESS;
}

grb_info k_reduce(int monoid, int dtype, const void* d, Index n, double* out) {
  if (n <= 0) { *out = monoid_identity(monoid, dtype); return GRB_SUCCESS; }   // reduce.hpp:25-28
  int grid;
  unsigned int *d_partial, *d_ticket;
  GRB_TRY(reduce_launch_prep(n, &grid, &d_partial, &d_ticket));
  Context& c = ctx();
  const int seq = ++c.mail_seq;
  GRB_TRY(dispatch_monoid(monoid, dtype, [&](auto mtag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int M = decltype(mtag)::value;
    hipLaunchKernelGGL((reduce_kernel<M, T, false>), dim3(grid), dim3(kBlock), 0, c.stream, (const T*)d, n, (T)0,
                       d_partial, d_ticket, c.d_hgran, seq);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  }));
  unsigned int raw = 0;
  GRB_TRY(wait_granules(seq, 1, &raw));
  if (dtype == GRB_F32) { float f; memcpy(&f, &raw, 4); *out = (double)f; }
  else *out = (double)(int)raw;
  return GRB_SUCCESS;
}

// Row-wise reduce of CSR values: one 16-lane group per row (mean degree ~16), rows with
// more entries loop; w[row] = identity for empty rows (cub::DeviceSegmentedReduce semantics).
template <int M, typename T>
__global__ void reduce_rows_kernel(const Index* __restrict__ ptr, const T* __restrict__ val, Index nrows,
                                   T* __restrict__ w) {
  constexpr int L = 16;
  const int groups_per_block = kBlock / L;
  const int g = threadIdx.x / L, l = threadIdx.x % L;
  for (Index row = blockIdx.x * groups_per_block + g; row < nrows; row += gridDim.x * groups_per_block) {
    Index s = ptr[row], e = ptr[row + 1];
    T acc = Monoid<M, T>::identity();
    for (Index i = s + l; i < e; i += L) acc = Monoid<M, T>::add(acc, val[i]);
    acc = group_reduce(acc, L, [](T a, T b) { return Monoid<M, T>::add(a, b); });
    if (l == 0) w[row] = acc;
  }
}

grb_info k_reduce_rows(int monoid, int dtype, const Index* ptr, const void* val, Index nrows, void* w) {
  if (nrows <= 0) return GRB_INVALID_OBJECT;
  return dispatch_monoid(monoid, dtype, [&](auto mtag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int M = decltype(mtag)::value;
    hipLaunchKernelGGL((reduce_rows_kernel<M, T>), dim3(stream_grid(nrows, kBlock / 16)), dim3(kBlock), 0,
                       ctx().stream, ptr, (const T*)val, nrows, (T*)w);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---------------------------------------------------------------- assign
template <typename T>
__global__ void assign_dense_mask_dense_kernel(T* __restrict__ w, Index n, const void* __restrict__ mask,
                                               int mask_f32, int scmp, T val) {
  // four elements per lane with 16-byte accesses where w and the mask are aligned: one wide store when all four
  // pass (the usual case in the drivers: masks are frontiers or their complements), single stores otherwise
  Index done = 0;
  if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mask)) & 15u) == 0) {
    struct alignas(16) Vec4 { T x, y, z, w; };
    struct alignas(16) Raw4 { unsigned int x, y, z, w; };
    const Raw4* __restrict__ m4 = reinterpret_cast<const Raw4*>(mask);
    Vec4* __restrict__ w4 = reinterpret_cast<Vec4*>(w);
    const Index n4 = n / 4;
    const bool want = scmp == 0;
    for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
      const Raw4 m = m4[i];
      // "castable to true": value != 0; for floats -0.0f (0x80000000) is zero too
      auto nz = [&](unsigned int u) { return mask_f32 ? ((u << 1) != 0u) : (u != 0u); };
      const bool p0 = nz(m.x) == want, p1 = nz(m.y) == want, p2 = nz(m.z) == want, p3 = nz(m.w) == want;
      if (p0 && p1 && p2 && p3) {
        w4[i] = Vec4{val, val, val, val};
      } else {
        if (p0) w[4 * i] = val;
        if (p1) w[4 * i + 1] = val;
        if (p2) w[4 * i + 2] = val;
        if (p3) w[4 * i + 3] = val;
      }
    }
    done = n4 * 4;
  }
  for (Index i = done + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (mask_pass(mask, mask_f32, scmp, i)) w[i] = val;
}
template <typename T>
__global__ void assign_sparse_mask_dense_kernel(const Index* __restrict__ w_ind, T* __restrict__ w_val,
                                                Index n, const void* __restrict__ mask, int mask_f32,
                                                int scmp, T val) {
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
    if (mask_pass(mask, mask_f32, scmp, w_ind[k])) w_val[k] = val;
}

grb_info k_assign_dense_mask_dense(int dtype, void* w, Index n, const void* mask, int mask_f32, int scmp,
                                   double val) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(assign_dense_mask_dense_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream,
                       (T*)w, n, mask, mask_f32, scmp, (T)val);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}
grb_info k_assign_dense_mask_sparse(int dtype, void* w, const Index* mask_ind, Index mask_nvals, double val) {
  return k_scatter_const(dtype, w, mask_ind, val, mask_nvals);
}
grb_info k_assign_sparse_mask_dense(int dtype, const Index* w_ind, void* w_val, Index w_nvals,
                                    const void* mask, int mask_f32, int scmp, double val) {
  if (w_nvals <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(assign_sparse_mask_dense_kernel<T>, dim3(stream_grid(w_nvals)), dim3(kBlock), 0,
                       ctx().stream, w_ind, (T*)w_val, w_nvals, mask, mask_f32, scmp, (T)val);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---------------------------------------------------------------- eWiseAdd
// 128-bit accesses when every operand is 16-byte aligned (they are for library-owned
// vectors); the scalar tail / unaligned adopt case falls back to dword accesses.
template <typename T> struct Vec4 { T x, y, z, w; };

template <int SR, typename T, bool kVec>
__global__ void ewise_add_dd_kernel(T* w, const T* u, const T* v, Index n) {
  typedef Semiring<SR, T> S;
  if constexpr (kVec) {
    const Index n4 = n >> 2;
    auto* w4 = reinterpret_cast<Vec4<T>*>(w);
    auto* u4 = reinterpret_cast<const Vec4<T>*>(u);
    auto* v4 = reinterpret_cast<const Vec4<T>*>(v);
    // w may BE u or v (the reference's in-place calls), never overlap them otherwise: element i of the output
    // depends on element i of the inputs only, so the loads of four strides are issued before the first store
    // (without this the compiler keeps every load behind the previous iteration's store: 5.2 against 5.9 TB/s)
    const Index stride = gridDim.x * blockDim.x;
    Index i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      Vec4<T> a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] = u4[i + k * stride]; b[k] = v4[i + k * stride]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Vec4<T> r;
        r.x = S::add(a[k].x, b[k].x); r.y = S::add(a[k].y, b[k].y); r.z = S::add(a[k].z, b[k].z); r.w = S::add(a[k].w, b[k].w);
        w4[i + k * stride] = r;
      }
    }
    for (; i < n4; i += stride) {
      Vec4<T> a = u4[i], b = v4[i], r;
      r.x = S::add(a.x, b.x); r.y = S::add(a.y, b.y); r.z = S::add(a.z, b.z); r.w = S::add(a.w, b.w);
      w4[i] = r;
    }
    for (Index j = (n4 << 2) + blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
      w[j] = S::add(u[j], v[j]);
  } else {
    for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      w[i] = S::add(u[i], v[i]);
  }
}

static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr,
                             const void* d = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

grb_info k_ewise_add_dense_dense(int sr, int dtype, void* w, const void* u, const void* v, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    // one element per lane and grid-stride, as eWiseMult: measured on 64 Mi floats, the 16-byte form of this kernel
    // ran at 5.0-5.4 TB/s (its loads wait behind the previous iteration's store: w may alias u or v) against 5.8-5.9
    // for the 4-byte form, whose many short iterations keep more loads in flight
    hipLaunchKernelGGL((ewise_add_dd_kernel<SR, T, false>), dim3(stream_grid(n)), dim3(kBlock), 0,
                       ctx().stream, (T*)w, (const T*)u, (const T*)v, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// w = reverse ? add(identity, w) : add(w, identity)   (eWiseAddDenseConstantKernel)
template <int SR, typename T>
__global__ void ewise_add_const_kernel(T* __restrict__ w, T identity, int reverse, Index n) {
  typedef Semiring<SR, T> S;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    T x = w[i];
    w[i] = reverse ? S::add(identity, x) : S::add(x, identity);
  }
}
grb_info k_ewise_add_const(int sr, int dtype, void* w, double identity, int reverse, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_add_const_kernel<SR, T>), dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream,
                       (T*)w, (T)identity, reverse, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// w[ind[k]] = add(u_val[k], v[ind[k]])   (eWiseAddSparseDenseKernel; v may alias w)
template <int SR, typename T>
__global__ void ewise_add_sd_kernel(T* w, const Index* __restrict__ u_ind, const T* __restrict__ u_val,
                                    const T* v, Index n) {
  typedef Semiring<SR, T> S;
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    Index i = u_ind[k];
    w[i] = S::add(u_val[k], v[i]);
  }
}
grb_info k_ewise_add_sparse_dense(int sr, int dtype, void* w, const Index* u_ind, const void* u_val,
                                  const void* v, Index u_nvals) {
  if (u_nvals <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_add_sd_kernel<SR, T>), dim3(stream_grid(u_nvals)), dim3(kBlock), 0, ctx().stream,
                       (T*)w, u_ind, (const T*)u_val, (const T*)v, u_nvals);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// w = op(w, val) with op = semiring add (use_add) or mul   (eWiseMultKernel scalar overload)
template <int SR, typename T, bool kAdd>
__global__ void ewise_scalar_kernel(T* __restrict__ w, T val, Index n) {
  typedef Semiring<SR, T> S;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    T x = w[i];
    w[i] = kAdd ? S::add(x, val) : S::mul(x, val);
  }
}
grb_info k_ewise_scalar(int sr, int dtype, int use_add, void* w, double val, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    if (use_add)
      hipLaunchKernelGGL((ewise_scalar_kernel<SR, T, true>), dim3(stream_grid(n)), dim3(kBlock), 0,
                         ctx().stream, (T*)w, (T)val, n);
    else
      hipLaunchKernelGGL((ewise_scalar_kernel<SR, T, false>), dim3(stream_grid(n)), dim3(kBlock), 0,
                         ctx().stream, (T*)w, (T)val, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---------------------------------------------------------------- eWiseMult
// dense x dense (optional dense mask): identity where either operand IS identity or
// the mask is zero (kernels/ewisemult.hpp:11-30, :66-90).
template <int SR, typename T>
__global__ void ewise_mult_dd_kernel(T* w, const void* __restrict__ mask, int mask_f32, const T* u,
                                     const T* v, Index n) {
  typedef Semiring<SR, T> S;
  const T ident = S::identity();
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    T a = u[i], b = v[i];
    bool dead = (a == ident) || (b == ident);
    if (mask) dead = dead || !mask_nonzero(mask, mask_f32, i);
    w[i] = dead ? ident : S::mul(a, b);
  }
}
grb_info k_ewise_mult_dense_dense(int sr, int dtype, void* w, const void* mask, int mask_f32, const void* u,
                                  const void* v, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_mult_dd_kernel<SR, T>), dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream,
                       (T*)w, mask, mask_f32, (const T*)u, (const T*)v, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// dense x dense under a SPARSE mask -> sparse output on the mask's pattern
// (kernels/ewisemult.hpp:35-61): value 0 where the mask value is 0.
template <int SR, typename T>
__global__ void ewise_mult_dd_spmask_kernel(Index* __restrict__ w_ind, T* __restrict__ w_val,
                                            const Index* __restrict__ m_ind, const void* __restrict__ m_val,
                                            int mask_f32, Index m_nvals, const T* __restrict__ u,
                                            const T* __restrict__ v) {
  typedef Semiring<SR, T> S;
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < m_nvals; k += gridDim.x * blockDim.x) {
    Index i = m_ind[k];
    T r = (T)0;
    if (mask_nonzero(m_val, mask_f32, k)) r = S::mul(u[i], v[i]);
    w_ind[k] = i;
    w_val[k] = r;
  }
}
grb_info k_ewise_mult_dense_dense_spmask(int sr, int dtype, Index* w_ind, void* w_val, const Index* m_ind,
                                         const void* m_val, int mask_f32, Index m_nvals, const void* u,
                                         const void* v) {
  if (m_nvals <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_mult_dd_spmask_kernel<SR, T>), dim3(stream_grid(m_nvals)), dim3(kBlock), 0,
                       ctx().stream, w_ind, (T*)w_val, m_ind, m_val, mask_f32, m_nvals, (const T*)u,
                       (const T*)v);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// sparse x dense (kernels/ewisemult.hpp:93-119): value 0 where u_val IS identity.
template <int SR, typename T>
__global__ void ewise_mult_sd_kernel(Index* w_ind, T* w_val, const Index* u_ind, const T* u_val,
                                     Index u_nvals, const T* __restrict__ v, int reverse) {
  typedef Semiring<SR, T> S;
  const T ident = S::identity();
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < u_nvals; k += gridDim.x * blockDim.x) {
    Index i = u_ind[k];
    T a = u_val[k];
    T r = (T)0;
    if (a != ident) {
      T b = v[i];
      r = reverse ? S::mul(b, a) : S::mul(a, b);
    }
    w_val[k] = r;
    w_ind[k] = i;
  }
}
grb_info k_ewise_mult_sparse_dense(int sr, int dtype, Index* w_ind, void* w_val, const Index* u_ind,
                                   const void* u_val, Index u_nvals, const void* v, int reverse) {
  if (u_nvals <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_mult_sd_kernel<SR, T>), dim3(stream_grid(u_nvals)), dim3(kBlock), 0, ctx().stream,
                       w_ind, (T*)w_val, u_ind, (const T*)u_val, u_nvals, (const T*)v, reverse);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// sparse x dense under a SPARSE mask (kernels/ewisemult.hpp:124-160): output on the
// mask's pattern, binary search of the (sorted) u indices.
template <int SR, typename T>
__global__ void ewise_mult_sd_spmask_kernel(Index* __restrict__ w_ind, T* __restrict__ w_val,
                                            const Index* __restrict__ m_ind, const void* __restrict__ m_val,
                                            int mask_f32, Index m_nvals, const Index* __restrict__ u_ind,
                                            const T* __restrict__ u_val, Index u_nvals,
                                            const T* __restrict__ v, int reverse) {
  typedef Semiring<SR, T> S;
  const T ident = S::identity();
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < m_nvals; k += gridDim.x * blockDim.x) {
    Index i = m_ind[k];
    T r = (T)0;
    if (mask_nonzero(m_val, mask_f32, k)) {
      T b = v[i];
      if (b != ident) {
        Index lo = 0, hi = u_nvals, found = -1;
        while (lo < hi) {
          Index mid = lo + ((hi - lo) >> 1);
          Index x = u_ind[mid];
          if (x == i) { found = mid; break; }
          if (x > i) hi = mid; else lo = mid + 1;
        }
        if (found >= 0) {
          T a = u_val[found];
          r = reverse ? S::mul(b, a) : S::mul(a, b);
        }
      }
    }
    w_ind[k] = i;
    w_val[k] = r;
  }
}
grb_info k_ewise_mult_sparse_dense_spmask(int sr, int dtype, Index* w_ind, void* w_val, const Index* m_ind,
                                          const void* m_val, int mask_f32, Index m_nvals, const Index* u_ind,
                                          const void* u_val, Index u_nvals, const void* v, int reverse) {
  if (m_nvals <= 0) return GRB_SUCCESS;
  return dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    hipLaunchKernelGGL((ewise_mult_sd_spmask_kernel<SR, T>), dim3(stream_grid(m_nvals)), dim3(kBlock), 0,
                       ctx().stream, w_ind, (T*)w_val, m_ind, m_val, mask_f32, m_nvals, u_ind, (const T*)u_val,
                       u_nvals, (const T*)v, reverse);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// u_val[k] = identity where mask[u_ind[k]] == 0   (zeroDenseIdentityKernel)
template <typename T>
__global__ void zero_dense_identity_kernel(const void* __restrict__ mask, int mask_f32, T identity,
                                           const Index* __restrict__ u_ind, T* __restrict__ u_val, Index n) {
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
    if (!mask_nonzero(mask, mask_f32, u_ind[k])) u_val[k] = identity;
}
grb_info k_zero_dense_identity(int dtype, const void* mask, int mask_f32, double identity, const Index* u_ind,
                               void* u_val, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(zero_dense_identity_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, mask,
                       mask_f32, (T)identity, u_ind, (T*)u_val, n);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---------------------------------------------------------------- scatter / gather by index vector
// w[idx[k]] = u[k]   (scatterIndexedKernel, kernels/scatter.hpp:24-39; racing duplicates: any winner)
template <typename T>
__global__ void scatter_indexed_kernel(T* __restrict__ w, Index w_n, const int* __restrict__ idx, Index n,
                                       const T* __restrict__ u) {
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const Index i = idx[k];
    if (i >= 0 && i < w_n) w[i] = u[k];
  }
}
// w[k] = u[idx[k]]   (gatherIndexedKernel, kernels/gather.hpp:9-23)
template <typename T>
__global__ void gather_indexed_kernel(T* w, Index w_n, const int* idx, Index n, const T* u) {
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const Index i = idx[k];
    if (i >= 0 && i < w_n) w[k] = u[i];
  }
}
grb_info k_scatter_indexed(int dtype, void* w, Index w_n, const int* idx, Index n, const void* u) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(scatter_indexed_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)w, w_n, idx,
                       n, (const T*)u);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}
grb_info k_gather_indexed(int dtype, void* w, Index w_n, const int* idx, Index n, const void* u) {
  if (n <= 0) return GRB_SUCCESS;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(gather_indexed_kernel<T>, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (T*)w, w_n, idx,
                       n, (const T*)u);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

// ---- apply: w[i] = f(u[i]) ----------------------------------------------------------------------------------------
// The unary operators of include/grb_hip.h (grb_unary_op); the two BIND kinds go through the run-time binary operator
// switch the registered semirings use (binop_rt: wave-uniform).  in == out is allowed.
template <typename T>
__global__ __launch_bounds__(kBlock) void apply_unary_kernel(const T* in, T* out, Index n, int unary, int op, T scalar) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const T x = in[i];
    T y;
    switch (unary) {
      case GRB_UNARY_IDENTITY: y = x; break;
      case GRB_UNARY_AINV: y = (T)0 - x; break;
      case GRB_UNARY_MINV: y = binop<OP_DIV, T>((T)1, x); break;
      case GRB_UNARY_ABS: y = x < (T)0 ? (T)0 - x : x; break;
      case GRB_UNARY_LNOT: y = (T)(x == (T)0); break;
      case GRB_UNARY_BIND_FIRST: y = binop_rt<T>(op, scalar, x); break;
      default: y = binop_rt<T>(op, x, scalar); break;
    }
    out[i] = y;
  }
}
grb_info k_apply_unary(int dtype, int unary, int op, double scalar, const void* in, void* out, Index n) {
  if (n <= 0) return GRB_SUCCESS;
  if (unary < 0 || unary >= GRB_N_UNARY_OPS) return GRB_INVALID_VALUE;
  if ((unary == GRB_UNARY_BIND_FIRST || unary == GRB_UNARY_BIND_SECOND) && (op < 0 || op >= GRB_N_BINARY_OPS)) return GRB_INVALID_VALUE;
  return dispatch_dtype(dtype, [&](auto t) -> grb_info {
    using T = decltype(t);
    hipLaunchKernelGGL(apply_unary_kernel<T>, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, ctx().stream, (const T*)in,
                       (T*)out, n, unary, op, (T)scalar);
    GRB_LAUNCH_CHECK();
    return GRB_SUCCESS;
  });
}

}  // namespace grb
