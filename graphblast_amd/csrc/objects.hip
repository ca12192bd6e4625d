// objects.hip -- library context and the three containers behind the C ABI:
// Descriptor (backend/cuda/descriptor.hpp), Vector (backend/cuda/vector.hpp +
// sparse_vector.hpp + dense_vector.hpp) and Matrix (backend/cuda/sparse_matrix.hpp).
//
// MI355X-first differences from the reference containers:
//  * no per-call cudaThreadSynchronize: everything is stream-ordered; only the calls
//    whose result is a host value synchronise (nvals of a fresh result, extractTuples,
//    reduce), through one pinned mailbox;
//  * fill / fillAscending / sparse2dense run on the device (the reference loops on the
//    host and copies n*4 bytes over PCIe every time, dense_vector.hpp:311-327);
//  * scratch is a set of grow-only slots owned by the context (the reference's
//    Descriptor::resize is malloc-new + D2D copy + free-old, descriptor.hpp:156-192);
//  * Matrix::build also prepares the row-block plans of the SpMV kernel for both
//    orientations (CSR for mxv, CSC for vxm-pull).
#include <algorithm>
#include <chrono>
#include <numeric>

#include "common.hpp"

namespace grb {

Context& ctx() {
  static Context c;
  return c;
}

grb_info ctx_init() {
  Context& c = ctx();
  if (c.inited) return GRB_SUCCESS;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "libgrb_hip: no HIP device visible -- this library has no CPU fallback\n");
    return GRB_PANIC;
  }
  GRB_HIP_TRY(hipHostMalloc((void**)&c.h_mail, 64 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  memset(c.h_mail, 0, 64 * sizeof(int));
  GRB_HIP_TRY(hipHostGetDevicePointer((void**)&c.d_hmail, c.h_mail, 0));
  GRB_HIP_TRY(hipHostMalloc((void**)&c.h_gran, 8 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
  memset(c.h_gran, 0, 8 * sizeof(unsigned long long));
  GRB_HIP_TRY(hipHostGetDevicePointer((void**)&c.d_hgran, c.h_gran, 0));
  GRB_HIP_TRY(hipMalloc((void**)&c.d_mail, 64 * sizeof(int)));
  GRB_HIP_TRY(hipMemset(c.d_mail, 0, 64 * sizeof(int)));
  GRB_HIP_TRY(hipMalloc((void**)&c.d_tickets, 9 * 32 * sizeof(unsigned int)));
  GRB_HIP_TRY(hipMemset(c.d_tickets, 0, 9 * 32 * sizeof(unsigned int)));
  GRB_HIP_TRY(hipEventCreate(&c.ev0));
  GRB_HIP_TRY(hipEventCreate(&c.ev1));
  int dev = 0;
  GRB_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  GRB_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  c.num_cu = prop.multiProcessorCount;
  // GRB_NUM_CU: workgroups of the co-resident ("persistent") launches, at most the device's CUs.  Processes that share
  // one GPU (tests/test_gpu_part_run.py: two ranks of the partitioned loops on one device) each take a part of it, so
  // that all their grids can be resident at once -- a grid barrier needs that.
  if (const char* e = getenv("GRB_NUM_CU")) {
    const int want = atoi(e);
    if (want >= 1 && want < c.num_cu) c.num_cu = want;
  }
  c.inited = true;
  return GRB_SUCCESS;
}

grb_info scratch(int i, size_t bytes, void** out) {
  Context& c = ctx();
  GRB_TRY(ctx_init());
  if (bytes < 256) bytes = 256;
  ++c.slot_epoch[i];
  if (c.slot_cap[i] < bytes) {
    if (c.slot[i]) {
      // stream-ordered users of the old buffer must finish before it is released
      GRB_HIP_TRY(hipStreamSynchronize(c.stream));
      GRB_HIP_TRY(hipFree(c.slot[i]));
      c.slot[i] = nullptr;
      c.slot_cap[i] = 0;
    }
    size_t cap = bytes + bytes / 4;
    cap = (cap + 255) & ~(size_t)255;
    GRB_HIP_TRY(hipMalloc(&c.slot[i], cap));
    c.slot_cap[i] = cap;
  }
  *out = c.slot[i];
  return GRB_SUCCESS;
}

// One wave copies `count` ints into the pinned, host-coherent mailbox and then publishes a
// sequence number with a system-scope release; the host spins on that word.  This replaces
// hipMemcpyAsync(D2H) + hipStreamSynchronize (~25-30 us of wake-up latency per call on this
// box) on every host-visible scalar: frontier sizes, nvals, reduce results.
__global__ void publish_kernel(const int* __restrict__ src, int count, int* __restrict__ mail, int seq) {
  if (threadIdx.x < count) mail[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&mail[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Spin until the mailbox shows sequence number `seq` (falls back to a blocking stream wait
// after 5 ms: long-running predecessor, or a fault that the wait then reports).
grb_info wait_mail(int seq) {
  Context& c = ctx();
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n(&c.h_mail[63], __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 1023u) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
      GRB_HIP_TRY(hipStreamSynchronize(c.stream));
      if (__atomic_load_n(&c.h_mail[63], __ATOMIC_ACQUIRE) != seq) {
        fprintf(stderr, "libgrb_hip: mailbox not published after stream sync\n");
        return GRB_PANIC;
      }
      break;
    }
  }
  return GRB_SUCCESS;
}

grb_info wait_granules(int seq, int count, unsigned int* out) {
  Context& c = ctx();
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  bool synced = false;
  for (;;) {
    bool ok = true;
    for (int k = 0; k < count; ++k) {
      const unsigned long long g = __atomic_load_n(&c.h_gran[k], __ATOMIC_ACQUIRE);
      if ((int)(g >> 32) != seq) { ok = false; break; }
      out[k] = (unsigned int)(g & 0xffffffffull);
    }
    if (ok) return GRB_SUCCESS;
    if (synced) {
      fprintf(stderr, "libgrb_hip: level record not published after stream sync\n");
      return GRB_PANIC;
    }
    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
      GRB_HIP_TRY(hipStreamSynchronize(c.stream));
      synced = true;
    }
  }
}

grb_info fetch_ints(const int* d_src, int count, int* h_dst) {
  Context& c = ctx();
  if (count > 62) return GRB_INVALID_VALUE;
  const int seq = ++c.mail_seq;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, c.stream, d_src, count, c.d_hmail, seq);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(wait_mail(seq));
  memcpy(h_dst, c.h_mail, sizeof(int) * count);
  return GRB_SUCCESS;
}

// ---- host-side semiring probes ------------------------------------------------
double semiring_identity(int sr, int dtype) {
  double out = 0;
  dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    out = (double)Semiring<decltype(tag)::value, T>::identity();
    return GRB_SUCCESS;
  });
  return out;
}
double semiring_add(int sr, int dtype, double a, double b) {
  double out = 0;
  dispatch_semiring(sr, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    out = (double)Semiring<decltype(tag)::value, T>::add((T)a, (T)b);
    return GRB_SUCCESS;
  });
  return out;
}
double monoid_identity(int m, int dtype) {
  double out = 0;
  dispatch_monoid(m, dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    out = (double)Monoid<decltype(tag)::value, T>::identity();
    return GRB_SUCCESS;
  });
  return out;
}
int semiring_monoid(int sr) {
  int out = -1;
  dispatch_semiring(sr, GRB_F32, [&](auto tag, auto t) -> grb_info {
    constexpr int SR = decltype(tag)::value;
    if constexpr (SR != GRB_RUNTIME_SR) out = SemiringTraits<SR>::monoid;
    return GRB_SUCCESS;
  });
  return out;
}

// ---- semirings registered at run time --------------------------------------------------
static std::vector<UserSemiring>& user_semirings() {
  static std::vector<UserSemiring> v;
  return v;
}
bool user_semiring_lookup(int id, UserSemiring* out) {
  const int k = id - GRB_USER_SEMIRING_BASE;
  if (k < 0 || k >= (int)user_semirings().size()) return false;
  *out = user_semirings()[(size_t)k];
  return true;
}
hipStream_t current_stream() { return ctx().stream; }

template <int SR>
static void match_builtin(int add_op, double identity, int mul_op, int* out) {
  typedef SemiringTraits<SR> Tr;
  if (MonoidTraits<Tr::monoid>::op == add_op && Tr::mul == mul_op &&
      ((double)Monoid<Tr::monoid, float>::identity() == identity || (double)Monoid<Tr::monoid, int>::identity() == identity))
    *out = SR;
}

static inline void store_scalar(int dtype, void* dst, double v) {
  if (dtype == GRB_F32) { float f = (float)v; memcpy(dst, &f, 4); }
  else { int i = (int)v; memcpy(dst, &i, 4); }
}
static inline double load_scalar(int dtype, const void* src) {
  if (dtype == GRB_F32) { float f; memcpy(&f, src, 4); return (double)f; }
  int i; memcpy(&i, src, 4); return (double)i;
}

}  // namespace grb

using namespace grb;

grb_info device_build_from_coo(grb_matrix A, const Index* d_rows, const Index* d_cols, const void* d_vals,
                               long long nvals_in, int flags);

extern "C" {

// =============================================================================== library
grb_info grb_set_stream(void* hip_stream) { GRB_API_ENTER();
  GRB_TRY(ctx_init());
  ctx().stream = (hipStream_t)hip_stream;
  return GRB_SUCCESS;
}

grb_info grb_device_info(char* buf, size_t buflen) { GRB_API_ENTER();
  GRB_TRY(ctx_init());
  int dev = 0;
  GRB_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  GRB_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buflen, "%s:%d:%s", prop.gcnArchName, prop.multiProcessorCount, prop.name);
  return GRB_SUCCESS;
}

const char* grb_version(void) { GRB_API_ENTER_NOINFO(); return "graphblast_amd 0.1 (gfx950)"; }

grb_info grb_timer_start(void) { GRB_API_ENTER();
  GRB_TRY(ctx_init());
  GRB_HIP_TRY(hipEventRecord(ctx().ev0, ctx().stream));
  return GRB_SUCCESS;
}
grb_info grb_timer_stop(float* elapsed_ms) { GRB_API_ENTER();
  GRB_HIP_TRY(hipEventRecord(ctx().ev1, ctx().stream));
  GRB_HIP_TRY(hipEventSynchronize(ctx().ev1));
  GRB_HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx().ev0, ctx().ev1));
  return GRB_SUCCESS;
}

// =============================================================================== Descriptor
grb_info grb_descriptor_new(grb_descriptor* desc) { GRB_API_ENTER();
  if (!desc) return GRB_NULL_POINTER;
  grb_descriptor d = new grb_descriptor_s();
  // backend/cuda/descriptor.hpp:17-18
  const int defaults[GRB_NDESCFIELD] = {GRB_DEFAULT, GRB_DEFAULT, GRB_DEFAULT, GRB_DEFAULT, GRB_FIXEDROW,
                                        32, 32, 128, GRB_PUSHPULL, 16, GRB_HIP};
  memcpy(d->desc, defaults, sizeof(defaults));
  *desc = d;
  return GRB_SUCCESS;
}
grb_info grb_descriptor_free(grb_descriptor desc) { GRB_API_ENTER(); delete desc; return GRB_SUCCESS; }

grb_info grb_descriptor_set(grb_descriptor desc, int field, int value) { GRB_API_ENTER_HOST();
  if (!desc) return GRB_UNINITIALIZED_OBJECT;
  if (field < 0 || field >= GRB_NDESCFIELD) return GRB_INVALID_VALUE;
  desc->desc[field] = value;
  return GRB_SUCCESS;
}
grb_info grb_descriptor_get(grb_descriptor desc, int field, int* value) { GRB_API_ENTER_HOST();
  if (!desc) return GRB_UNINITIALIZED_OBJECT;
  if (field < 0 || field >= GRB_NDESCFIELD) return GRB_INVALID_VALUE;
  *value = desc->desc[field];
  return GRB_SUCCESS;
}
// backend/cuda/descriptor.hpp:141-154
grb_info grb_descriptor_toggle(grb_descriptor desc, int field) { GRB_API_ENTER_HOST();
  if (!desc) return GRB_UNINITIALIZED_OBJECT;
  if (field >= 0 && field < 4) {
    if (desc->desc[field] != GRB_DEFAULT) desc->desc[field] = GRB_DEFAULT;
    else if (field > 2) desc->desc[field] = GRB_TRAN;
    else desc->desc[field] = field;      // MASK -> SCMP(0), OUTP -> REPLACE(1), INP0 -> TRAN(2)
  }
  return GRB_SUCCESS;
}

static grb_info desc_apply_modes(grb_descriptor d) {
  switch (d->mxvmode) {                   // descriptor.hpp:244-256
    case 0: d->desc[GRB_MXVMODE] = GRB_PUSHPULL; break;
    case 1: d->desc[GRB_MXVMODE] = GRB_PUSHONLY; break;
    case 2: d->desc[GRB_MXVMODE] = GRB_PULLONLY; break;
    default: return GRB_INVALID_VALUE;
  }
  switch (d->nthread) {                   // descriptor.hpp:258-280
    case 32: case 64: case 128: case 256: case 512: case 1024: d->desc[GRB_NT] = d->nthread; break;
    default: return GRB_INVALID_VALUE;
  }
  return GRB_SUCCESS;
}

// parseArgs defaults, graphblas/util.hpp:39-132
grb_info grb_descriptor_load_defaults(grb_descriptor d) { GRB_API_ENTER();
  if (!d) return GRB_UNINITIALIZED_OBJECT;
  d->niter = 10; d->max_niter = 10000; d->directed = 0; d->timing = 1; d->transpose = 0;
  d->mxvmode = 1; d->switchpoint = 0.01f; d->dirinfo = 0; d->struconly = 0; d->opreuse = 0;
  d->memusage = 1.0f; d->endbit = 1; d->sort = 1; d->atomic = 0; d->earlyexit = 1; d->fusedmask = 1;
  d->nthread = 128; d->debug = 0; d->edgeswitch = 0.f;
  return desc_apply_modes(d);
}

#define GRB_DESC_ARGS(X)                                                                          \
  X(mxvmode) X(switchpoint) X(struconly) X(opreuse) X(earlyexit) X(fusedmask) X(sort) X(endbit)   \
  X(memusage) X(atomic) X(dirinfo) X(nthread) X(max_niter) X(niter) X(timing) X(debug) X(directed) \
  X(transpose) X(edgeswitch)

grb_info grb_descriptor_set_arg(grb_descriptor d, const char* name, double value) { GRB_API_ENTER_HOST();
  if (!d || !name) return GRB_UNINITIALIZED_OBJECT;
#define X(f) if (strcmp(name, #f) == 0) { d->f = (decltype(d->f))value; return desc_apply_modes(d); }
  GRB_DESC_ARGS(X)
#undef X
  return GRB_INVALID_VALUE;
}
grb_info grb_descriptor_get_arg(grb_descriptor d, const char* name, double* value) { GRB_API_ENTER_HOST();
  if (!d || !name || !value) return GRB_UNINITIALIZED_OBJECT;
#define X(f) if (strcmp(name, #f) == 0) { *value = (double)d->f; return GRB_SUCCESS; }
  GRB_DESC_ARGS(X)
#undef X
  return GRB_INVALID_VALUE;
}
// REGISTER_SEMIRING(SR, ADD_MONOID, MULT_BINARYOP) / REGISTER_MONOID(M, BINARYOP, IDENTITY) at run time
// (graphblas/stddef.hpp:140-191).  Operators are numbered as the functor templates of stddef.hpp:14-138 appear there
// (grb_binary_op); the same composition registered twice gets the same id.
grb_info grb_semiring_register(int add_op, double add_identity, int mul_op, int* id) { GRB_API_ENTER();
  if (!id) return GRB_NULL_POINTER;
  if (add_op < 0 || add_op > OP_DIV || mul_op < 0 || mul_op > OP_DIV) return GRB_INVALID_VALUE;
  std::vector<UserSemiring>& reg = user_semirings();
  for (size_t k = 0; k < reg.size(); ++k)
    if (reg[k].add_op == add_op && reg[k].mul_op == mul_op && reg[k].identity == add_identity) {
      *id = GRB_USER_SEMIRING_BASE + (int)k;
      return GRB_SUCCESS;
    }
  UserSemiring u = {add_op, mul_op, add_identity, -1};
#define GRB_M(SR) match_builtin<SR>(add_op, add_identity, mul_op, &u.builtin);
  GRB_M(GRB_LOGICAL_OR_AND) GRB_M(GRB_PLUS_MULTIPLIES) GRB_M(GRB_MINIMUM_PLUS) GRB_M(GRB_MAXIMUM_MULTIPLIES)
  GRB_M(GRB_PLUS_DIVIDES) GRB_M(GRB_PLUS_GREATER) GRB_M(GRB_GREATER_PLUS) GRB_M(GRB_PLUS_MINUS) GRB_M(GRB_PLUS_LESS)
  GRB_M(GRB_CUSTOM_LESS_PLUS) GRB_M(GRB_MINIMUM_MULTIPLIES) GRB_M(GRB_MULTIPLIES_MULTIPLIES) GRB_M(GRB_NOT_EQUAL_TO_PLUS)
  GRB_M(GRB_MINIMUM_SELECT_SECOND) GRB_M(GRB_PLUS_NOT_EQUAL_TO) GRB_M(GRB_CUSTOM_LESS_LESS) GRB_M(GRB_MINIMUM_NOT_EQUAL_TO)
#undef GRB_M
  reg.push_back(u);
  *id = GRB_USER_SEMIRING_BASE + (int)reg.size() - 1;
  return GRB_SUCCESS;
}

grb_info grb_descriptor_iter_log(grb_descriptor d, grb_algo_iter* out, int cap, int* count) { GRB_API_ENTER();
  if (!d) return GRB_UNINITIALIZED_OBJECT;
  const int k = (int)d->iter_log.size();
  if (count) *count = k;
  if (out && cap > 0 && k > 0) memcpy(out, d->iter_log.data(), sizeof(grb_algo_iter) * (size_t)(k < cap ? k : cap));
  return GRB_SUCCESS;
}
grb_info grb_descriptor_lastmxv(grb_descriptor d, int* value) { GRB_API_ENTER_HOST();
  if (!d) return GRB_UNINITIALIZED_OBJECT;
  *value = d->lastmxv;
  return GRB_SUCCESS;
}

// =============================================================================== Vector
constexpr size_t kVecPoolCap = 4ull << 30;    // bytes of freed vector storage kept for reuse

static grb_info pool_get(void** out, size_t bytes) {
  Context& c = ctx();
  auto it = c.vec_pool.find(bytes);
  if (it != c.vec_pool.end() && !it->second.empty()) {
    *out = it->second.back();
    it->second.pop_back();
    c.vec_pool_bytes -= bytes;
    return GRB_SUCCESS;
  }
  GRB_HIP_TRY(hipMalloc(out, bytes));
  return GRB_SUCCESS;
}

// Work queued on the library's stream may still use the block: a block taken from the pool is
// only touched by later work on the same stream, so no synchronisation is needed to recycle it.
static void pool_put(void* p, size_t bytes) {
  if (!p) return;
  Context& c = ctx();
  if (c.vec_pool_bytes + bytes <= kVecPoolCap) {
    c.vec_pool[bytes].push_back(p);
    c.vec_pool_bytes += bytes;
    return;
  }
  (void)hipStreamSynchronize(c.stream);
  (void)hipFree(p);
}

static void vec_release_sparse(grb_vector v) {
  if (v->s_owned) {
    pool_put(v->s_ind, sizeof(Index) * (size_t)v->s_alloc_n);
    pool_put(v->s_val, 4 * ((size_t)v->s_alloc_n + 1));
  }
  v->s_ind = nullptr; v->s_val = nullptr; v->s_owned = false;
}
static void vec_release_dense(grb_vector v) {
  if (v->d_owned) pool_put(v->d_val, 4 * (size_t)v->d_alloc_n);
  v->d_val = nullptr; v->d_owned = false;
}

static grb_info vec_alloc_sparse(grb_vector v) {
  if (v->nsize > 0 && !v->s_ind) {
    GRB_TRY(pool_get((void**)&v->s_ind, sizeof(Index) * (size_t)v->nsize));
    GRB_TRY(pool_get(&v->s_val, 4 * ((size_t)v->nsize + 1)));
    v->s_owned = true;
    v->s_alloc_n = v->nsize;
  }
  return GRB_SUCCESS;
}
static grb_info vec_alloc_dense(grb_vector v) {
  if (v->nsize > 0 && !v->d_val) {
    GRB_TRY(pool_get(&v->d_val, 4 * (size_t)v->nsize));
    v->d_owned = true;
    v->d_alloc_n = v->nsize;
  }
  return GRB_SUCCESS;
}

grb_info grb_vector_new(grb_vector* out, grb_dtype dtype, grb_index nsize) { GRB_API_ENTER();
  if (!out) return GRB_NULL_POINTER;
  if (nsize < 0) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  grb_vector v = new grb_vector_s();
  v->dtype = dtype;
  v->nsize = nsize;
  // both representations exist at full size from construction (vector.hpp:30-32)
  grb_info i = vec_alloc_sparse(v);
  if (i == GRB_SUCCESS) i = vec_alloc_dense(v);
  if (i != GRB_SUCCESS) { delete v; return i; }
  *out = v;
  return GRB_SUCCESS;
}

// Vector::resize (vector.hpp:230-237 -> dense_vector.hpp:286-309 / sparse_vector.hpp:242-277): the
// active representation is reallocated for `nsize` elements keeping its first min(nsize, nvals)
// entries (new dense elements are zero here; the reference leaves them uninitialised).  Both
// representations follow the new size.
grb_info grb_vector_resize(grb_vector v, grb_index nsize) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (nsize < 0) return GRB_INVALID_VALUE;
  if (v->vec_type != GRB_SPARSE && v->vec_type != GRB_DENSE) return GRB_UNINITIALIZED_OBJECT;
  Context& c = ctx();
  const Index old = v->nsize;
  if (nsize == old) return GRB_SUCCESS;
  grb_vector_s t;                                     // new storage, built on a scratch descriptor
  t.dtype = v->dtype;
  t.nsize = nsize;
  GRB_TRY(vec_alloc_sparse(&t));
  GRB_TRY(vec_alloc_dense(&t));
  if (nsize > 0) GRB_HIP_TRY(hipMemsetAsync(t.d_val, 0, 4 * (size_t)nsize, c.stream));
  if (v->vec_type == GRB_DENSE) {
    const Index keep = nsize < old ? nsize : old;
    if (keep > 0 && v->d_val) GRB_TRY(k_copy(t.d_val, v->d_val, 4 * (size_t)keep));
  } else {
    const Index keep = nsize < v->s_nvals ? nsize : v->s_nvals;
    if (keep > 0) {
      GRB_TRY(k_copy(t.s_ind, v->s_ind, sizeof(Index) * (size_t)keep));
      GRB_TRY(k_copy(t.s_val, v->s_val, 4 * (size_t)keep));
    }
    v->s_nvals = keep;
  }
  GRB_HIP_TRY(hipStreamSynchronize(c.stream));        // the old blocks go back to the pool
  vec_release_sparse(v);
  vec_release_dense(v);
  v->s_ind = t.s_ind; v->s_val = t.s_val; v->s_owned = t.s_owned; v->s_alloc_n = t.s_alloc_n;
  v->d_val = t.d_val; v->d_owned = t.d_owned; v->d_alloc_n = t.d_alloc_n;
  v->nsize = nsize;
  return GRB_SUCCESS;
}

grb_info grb_vector_free(grb_vector v) { GRB_API_ENTER();
  if (!v) return GRB_SUCCESS;
  vec_release_sparse(v);
  vec_release_dense(v);
  delete v;
  return GRB_SUCCESS;
}

grb_info grb_vector_set_storage(grb_vector v, int storage) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  v->vec_type = storage;
  if (storage == GRB_SPARSE) return vec_alloc_sparse(v);
  if (storage == GRB_DENSE) return vec_alloc_dense(v);
  return GRB_SUCCESS;
}
grb_info grb_vector_get_storage(grb_vector v, int* storage) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  *storage = v->vec_type;
  return GRB_SUCCESS;
}

grb_info grb_vector_dup(grb_vector dst, grb_vector src) { GRB_API_ENTER_QUEUE();
  if (!dst || !src) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  if (dst->nsize != src->nsize || dst->dtype != src->dtype) { GRB_TRY(lazy_flush()); return GRB_DIMENSION_MISMATCH; }
  {
    // a dense copy between library-owned vectors joins the queue of element-wise calls (lazy.hip)
    grb_info fi = GRB_SUCCESS;
    if (src->vec_type == GRB_DENSE && dst != src && dst->d_val && dst->d_owned) {
      const int before = dst->vec_type;
      dst->vec_type = GRB_DENSE;
      if (lazy_try(LZ_DUP, 0, dst, src, nullptr, 0.0, &fi)) { dst->d_nnz = src->d_nnz; return GRB_SUCCESS; }
      dst->vec_type = before;
    } else {
      fi = lazy_flush();
    }
    GRB_TRY(fi);
  }
  dst->vec_type = src->vec_type;
  if (src->vec_type == GRB_SPARSE) {
    GRB_TRY(vec_alloc_sparse(dst));
    // the reference copies nsize elements (sparse_vector.hpp:110-113); nvals suffice
    if (src->s_nvals > 0) {
      GRB_TRY(k_copy(dst->s_ind, src->s_ind, 4 * (size_t)src->s_nvals));
      GRB_TRY(k_copy(dst->s_val, src->s_val, 4 * (size_t)src->s_nvals));
    }
    dst->s_nvals = src->s_nvals;
    return GRB_SUCCESS;
  }
  if (src->vec_type == GRB_DENSE) {
    GRB_TRY(vec_alloc_dense(dst));
    if (src->nsize > 0)
      GRB_TRY(k_copy(dst->d_val, src->d_val, 4 * (size_t)src->nsize));
    dst->d_nnz = src->d_nnz;
    return GRB_SUCCESS;
  }
  return GRB_UNINITIALIZED_OBJECT;
}

// apply on the device (include/grb_hip.h): w = f(u) on every stored element; w takes u's storage
grb_info grb_vector_apply(grb_vector w, grb_vector mask, grb_accum accum, int unary, int binop, double scalar, grb_vector u,
                          grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!w || !u || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (w->nsize != u->nsize) return GRB_DIMENSION_MISMATCH;
  if (w->dtype != u->dtype) return GRB_DOMAIN_MISMATCH;
  if (mask) return GRB_NOT_IMPLEMENTED;                  // "apply masked not implemented yet" (apply.hpp:36-37)
  if (u->vec_type == GRB_SPARSE) {
    if (w != u) {
      w->vec_type = GRB_SPARSE;
      GRB_TRY(vec_alloc_sparse(w));
      if (u->s_nvals > 0) GRB_TRY(k_copy(w->s_ind, u->s_ind, 4 * (size_t)u->s_nvals));
      w->s_nvals = u->s_nvals;
    }
    return k_apply_unary(u->dtype, unary, binop, scalar, u->s_val, w->s_val, u->s_nvals);
  }
  if (u->vec_type != GRB_DENSE) return GRB_UNINITIALIZED_OBJECT;
  if (w != u) {
    w->vec_type = GRB_DENSE;
    GRB_TRY(vec_alloc_dense(w));
  }
  GRB_TRY(k_apply_unary(u->dtype, unary, binop, scalar, u->d_val, w->d_val, u->nsize));
  w->d_nnz = u->nsize;                                   // recounted on demand (grb_vector_nvals: dense -> size)
  return GRB_SUCCESS;
}

grb_info grb_vector_clear(grb_vector v) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  v->vec_type = GRB_UNKNOWN;          // vector.hpp:105-112
  v->nvals = 0;
  v->s_nvals = 0;
  return k_fill(v->dtype, v->d_val, 0.0, v->nsize);
}

grb_info grb_vector_size(grb_vector v, grb_index* nsize) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  *nsize = v->nsize;
  return GRB_SUCCESS;
}

// vector.hpp:132-146: sparse -> stored count, dense -> size, unknown -> cached
grb_info grb_vector_nvals(grb_vector v, grb_index* nvals) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type == GRB_SPARSE) v->nvals = v->s_nvals;
  else if (v->vec_type == GRB_DENSE) v->nvals = v->nsize;
  *nvals = v->nvals;
  return GRB_SUCCESS;
}

grb_info grb_vector_build_sparse(grb_vector v, const grb_index* indices, const void* values, grb_index nvals) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  v->vec_type = GRB_SPARSE;                                // vector.hpp:154: set before sparse_.build can fail
  GRB_TRY(vec_alloc_sparse(v));
  if (nvals > v->nsize) return GRB_PANIC;                  // sparse_vector.hpp:138-142
  if (v->s_nvals > 0) return GRB_OUTPUT_NOT_EMPTY;
  if (nvals > 0) {
    GRB_HIP_TRY(hipMemcpyAsync(v->s_ind, indices, 4 * (size_t)nvals, hipMemcpyHostToDevice, ctx().stream));
    GRB_HIP_TRY(hipMemcpyAsync(v->s_val, values, 4 * (size_t)nvals, hipMemcpyHostToDevice, ctx().stream));
    GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));       // host buffers may be released by the caller
  }
  v->s_nvals = nvals;
  return GRB_SUCCESS;
}

grb_info grb_vector_build_dense(grb_vector v, const void* values, grb_index nvals) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (nvals > v->nsize) return GRB_INDEX_OUT_OF_BOUNDS;    // dense_vector.hpp:201-202
  v->vec_type = GRB_DENSE;
  GRB_TRY(vec_alloc_dense(v));
  if (nvals > 0) {
    GRB_HIP_TRY(hipMemcpyAsync(v->d_val, values, 4 * (size_t)nvals, hipMemcpyHostToDevice, ctx().stream));
    GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  }
  return GRB_SUCCESS;
}

grb_info grb_vector_adopt_dense(grb_vector v, void* d_values, grb_index nvals) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  vec_release_dense(v);
  v->d_val = d_values;
  v->d_owned = false;
  v->nsize = nvals;
  v->vec_type = GRB_DENSE;
  return GRB_SUCCESS;
}
grb_info grb_vector_adopt_sparse(grb_vector v, grb_index* d_indices, void* d_values, grb_index nvals) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  vec_release_sparse(v);
  v->s_ind = d_indices;
  v->s_val = d_values;
  v->s_owned = false;
  v->s_nvals = nvals;
  v->vec_type = GRB_SPARSE;
  return GRB_SUCCESS;
}

grb_info grb_vector_set_element(grb_vector v, double val, grb_index index) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (index < 0 || index >= v->nsize) return GRB_INDEX_OUT_OF_BOUNDS;
  uint32_t raw;
  store_scalar(v->dtype, &raw, val);
  hipStream_t s = ctx().stream;
  if (v->vec_type == GRB_DENSE) {
    GRB_HIP_TRY(hipMemcpyAsync((char*)v->d_val + 4 * (size_t)index, &raw, 4, hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    return GRB_SUCCESS;
  }
  if (v->vec_type == GRB_SPARSE) {           // appended, sparse_vector.hpp:177-185
    if (v->s_nvals >= v->nsize) return GRB_INDEX_OUT_OF_BOUNDS;
    GRB_HIP_TRY(hipMemcpyAsync(v->s_ind + v->s_nvals, &index, 4, hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipMemcpyAsync((char*)v->s_val + 4 * (size_t)v->s_nvals, &raw, 4, hipMemcpyHostToDevice, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    v->s_nvals++;
    return GRB_SUCCESS;
  }
  return GRB_UNINITIALIZED_OBJECT;
}

grb_info grb_vector_extract_element(grb_vector v, double* val, grb_index index) { GRB_API_ENTER();
  if (!v || !val) return GRB_UNINITIALIZED_OBJECT;
  if (index < 0 || index >= v->nsize) return GRB_INDEX_OUT_OF_BOUNDS;
  if (v->vec_type != GRB_DENSE) return GRB_NOT_IMPLEMENTED;
  uint32_t raw = 0;
  GRB_HIP_TRY(hipMemcpyAsync(&raw, (char*)v->d_val + 4 * (size_t)index, 4, hipMemcpyDeviceToHost, ctx().stream));
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  *val = load_scalar(v->dtype, &raw);
  return GRB_SUCCESS;
}

grb_info grb_vector_extract_tuples_sparse(grb_vector v, grb_index* indices, void* values, grb_index* n) { GRB_API_ENTER();
  if (!v || !n) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type != GRB_SPARSE) return v->vec_type == GRB_DENSE ? GRB_NOT_IMPLEMENTED : GRB_UNINITIALIZED_OBJECT;
  if (*n > v->s_nvals) return GRB_UNINITIALIZED_OBJECT;    // sparse_vector.hpp:203-211
  if (*n < v->s_nvals) return GRB_INSUFFICIENT_SPACE;
  if (*n > 0) {
    hipStream_t s = ctx().stream;
    GRB_HIP_TRY(hipMemcpyAsync(indices, v->s_ind, 4 * (size_t)*n, hipMemcpyDeviceToHost, s));
    if (values) GRB_HIP_TRY(hipMemcpyAsync(values, v->s_val, 4 * (size_t)*n, hipMemcpyDeviceToHost, s));
  }
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  return GRB_SUCCESS;
}

grb_info grb_vector_extract_tuples_dense(grb_vector v, void* values, grb_index* n) { GRB_API_ENTER();
  if (!v || !n) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type == GRB_SPARSE) GRB_TRY(grb_vector_sparse2dense(v, 0.0, nullptr));   // vector.hpp:208-217
  if (v->vec_type != GRB_DENSE) return GRB_UNINITIALIZED_OBJECT;
  if (*n > v->nsize) return GRB_UNINITIALIZED_OBJECT;      // dense_vector.hpp:255-264
  if (*n < v->nsize) return GRB_INSUFFICIENT_SPACE;
  if (*n > 0)
    GRB_HIP_TRY(hipMemcpyAsync(values, v->d_val, 4 * (size_t)*n, hipMemcpyDeviceToHost, ctx().stream));
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  return GRB_SUCCESS;
}

grb_info grb_vector_fill(grb_vector v, double val) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type != GRB_DENSE) GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
  return k_fill(v->dtype, v->d_val, val, v->nsize);
}
grb_info grb_vector_fill_ascending(grb_vector v, grb_index) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type != GRB_DENSE) GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
  return k_fill_ascending(v->dtype, v->d_val, v->nsize);
}

// vector.hpp:428-450: same storage required; size, nvals and ratio_ travel with the contents
grb_info grb_vector_swap(grb_vector a, grb_vector b) { GRB_API_ENTER();
  if (!a || !b) return GRB_UNINITIALIZED_OBJECT;
  if (a->vec_type != b->vec_type || a->vec_type == GRB_UNKNOWN) return GRB_INVALID_OBJECT;
  if (a->vec_type == GRB_SPARSE) {
    std::swap(a->s_ind, b->s_ind);
    std::swap(a->s_val, b->s_val);
    std::swap(a->s_nvals, b->s_nvals);
    std::swap(a->s_owned, b->s_owned);
    std::swap(a->s_alloc_n, b->s_alloc_n);      // the pool files a freed block under this size
  } else {
    std::swap(a->d_val, b->d_val);
    std::swap(a->d_nnz, b->d_nnz);
    std::swap(a->d_owned, b->d_owned);
    std::swap(a->d_alloc_n, b->d_alloc_n);
  }
  std::swap(a->nsize, b->nsize);
  std::swap(a->nvals, b->nvals);
  std::swap(a->ratio, b->ratio);
  return GRB_SUCCESS;
}

// vector.hpp:325-364
grb_info grb_vector_sparse2dense(grb_vector v, double identity, grb_descriptor desc) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type == GRB_DENSE) return GRB_SUCCESS;
  if (v->vec_type == GRB_UNKNOWN) return grb_vector_set_storage(v, GRB_DENSE);
  GRB_TRY(vec_alloc_dense(v));
  const Index nvals = v->s_nvals;
  if (!desc || !desc->opreuse) {
    GRB_TRY(k_fill(v->dtype, v->d_val, identity, v->nsize));
    if (desc && desc->struconly) GRB_TRY(k_scatter_const(v->dtype, v->d_val, v->s_ind, 1.0, nvals));
    else GRB_TRY(k_scatter_vals(v->dtype, v->d_val, v->s_ind, v->s_val, nvals));
  }
  v->vec_type = GRB_DENSE;
  v->d_nnz = nvals;
  return GRB_SUCCESS;
}

// vector.hpp:366-425
grb_info grb_vector_dense2sparse(grb_vector v, double identity, grb_descriptor desc) { GRB_API_ENTER();
  if (!v || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (v->vec_type == GRB_SPARSE) return GRB_INVALID_OBJECT;
  GRB_TRY(vec_alloc_sparse(v));
  GRB_TRY(vec_alloc_dense(v));
  Index nv = 0;
  GRB_TRY(k_dense2sparse(v->dtype, v->d_val, identity, v->nsize, v->s_ind, desc->struconly ? nullptr : v->s_val, &nv));
  v->s_nvals = nv;
  v->vec_type = GRB_SPARSE;
  return GRB_SUCCESS;
}

// vector.hpp:291-323: the direction-optimisation heuristic
grb_info grb_vector_convert(grb_vector v, double identity, float switchpoint, grb_descriptor desc) { GRB_API_ENTER();
  if (!v || !desc) return GRB_UNINITIALIZED_OBJECT;
  Index nvals_t = 0, nsize_t = v->nsize;
  if (v->vec_type == GRB_SPARSE) {
    nvals_t = v->s_nvals;
  } else if (v->vec_type == GRB_DENSE) {
    if (v->nsize == 0) return GRB_INVALID_OBJECT;
    GRB_TRY(k_count_nonidentity(v->dtype, v->d_val, identity, v->nsize, &nvals_t));
    v->d_nnz = nvals_t;
  } else {
    return GRB_UNINITIALIZED_OBJECT;
  }
  const float ratio = (float)nvals_t / (float)nsize_t;
  if (v->vec_type == GRB_SPARSE) {
    if (ratio > switchpoint && ratio > v->ratio) GRB_TRY(grb_vector_sparse2dense(v, identity, desc));
    else v->ratio = ratio;
  } else {
    if (ratio <= switchpoint && ratio < v->ratio) GRB_TRY(grb_vector_dense2sparse(v, identity, desc));
    else v->ratio = ratio;
  }
  return GRB_SUCCESS;
}

grb_info grb_vector_device_ptrs(grb_vector v, grb_index** d_sparse_ind, void** d_sparse_val, void** d_dense_val) { GRB_API_ENTER();
  if (!v) return GRB_UNINITIALIZED_OBJECT;
  v->exposed = true;              // the caller may now look at the storage without telling the library: never deferred again
  if (d_sparse_ind) *d_sparse_ind = v->s_ind;
  if (d_sparse_val) *d_sparse_val = v->s_val;
  if (d_dense_val) *d_dense_val = v->d_val;
  return GRB_SUCCESS;
}

// =============================================================================== Matrix
grb_info grb_matrix_new(grb_matrix* out, grb_dtype dtype, grb_index nrows, grb_index ncols) { GRB_API_ENTER();
  if (!out) return GRB_NULL_POINTER;
  if (nrows < 0 || ncols < 0) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  grb_matrix A = new grb_matrix_s();
  A->dtype = dtype;
  A->nrows = nrows;
  A->ncols = ncols;
  const char* fmt = getenv("GRB_SPARSE_MATRIX_FORMAT");        // sparse_matrix.hpp:34,45
  A->format = fmt ? atoi(fmt) : 0;
  *out = A;
  return GRB_SUCCESS;
}

extern "C++" {
void grb::matrix_release_device(grb_matrix A) {
  (void)hipStreamSynchronize(ctx().stream);
  if (A->owned) {
    for (CsrArrays* m : {&A->csr, &A->csc}) {
      if (m == &A->csc && A->csc_alias) continue;              // CSR-only format: the same arrays
      if (m->ptr) (void)hipFree(m->ptr);
      if (m->ind) (void)hipFree(m->ind);
      if (m->val) (void)hipFree(m->val);
    }
  }
  A->csr = CsrArrays();
  A->csc = CsrArrays();
  A->csc_alias = false;
  if (A->d_no_in_edges) { (void)hipFree(A->d_no_in_edges); A->d_no_in_edges = nullptr; }
  if (A->d_empty_csr_rows) { (void)hipFree(A->d_empty_csr_rows); A->d_empty_csr_rows = nullptr; }
  if (A->d_pull_hint) { (void)hipFree(A->d_pull_hint); A->d_pull_hint = nullptr; }
  A->bfs_n_in = -1; A->bfs_out_is_in = false;
  if (A->d_oc_bounds) { (void)hipFree(A->d_oc_bounds); A->d_oc_bounds = nullptr; }
  if (A->d_oc_off) { (void)hipFree(A->d_oc_off); A->d_oc_off = nullptr; }
  if (A->d_oc_bigidx) { (void)hipFree(A->d_oc_bigidx); A->d_oc_bigidx = nullptr; }
  A->oc_nb = A->oc_nrows = A->oc_state = 0;
  if (A->d_oc2_bounds) { (void)hipFree(A->d_oc2_bounds); A->d_oc2_bounds = nullptr; }
  if (A->d_oc2_off) { (void)hipFree(A->d_oc2_off); A->d_oc2_off = nullptr; }
  if (A->d_oc2_bigidx) { (void)hipFree(A->d_oc2_bigidx); A->d_oc2_bigidx = nullptr; }
  A->oc2_nb = A->oc2_nrows = A->oc2_state = A->oc2_grid = 0;
  for (BatchSlices* b : {&A->batch_in, &A->batch_out}) {
    if (b->d_slices) (void)hipFree(b->d_slices);
    if (b->d_rows) (void)hipFree(b->d_rows);
    if (b->d_range_off) (void)hipFree(b->d_range_off);
    if (b->d_range_bounds) (void)hipFree(b->d_range_bounds);
    if (b->d_range_ids) (void)hipFree(b->d_range_ids);
    *b = BatchSlices();
  }
  A->nonneg_values = -1; A->mean_value = -1.0; A->small_int_values = -1;
  tc_prep_free(A);
  free_spmv_plan(&A->plan_csr);
  free_spmv_plan(&A->plan_csc);
  A->plan_csr_pending = false;
  A->built = false;
}
}  // extern "C++"

grb_info grb_matrix_free(grb_matrix A) { GRB_API_ENTER();
  if (!A) return GRB_SUCCESS;
  matrix_release_device(A);
  delete A;
  return GRB_SUCCESS;
}

static grb_info upload(CsrArrays* d, Index n, Index nvals, const std::vector<Index>& ptr,
                       const std::vector<Index>& ind, const std::vector<uint32_t>& val) {
  d->n = n;
  d->nvals = nvals;
  GRB_HIP_TRY(hipMalloc((void**)&d->ptr, 4 * ((size_t)n + 1)));
  GRB_HIP_TRY(hipMemcpy(d->ptr, ptr.data(), 4 * ((size_t)n + 1), hipMemcpyHostToDevice));
  size_t cap = nvals > 0 ? (size_t)nvals : 1;
  GRB_HIP_TRY(hipMalloc((void**)&d->ind, 4 * cap));
  GRB_HIP_TRY(hipMalloc(&d->val, 4 * cap));
  if (nvals > 0) {
    GRB_HIP_TRY(hipMemcpy(d->ind, ind.data(), 4 * (size_t)nvals, hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMemcpy(d->val, val.data(), 4 * (size_t)nvals, hipMemcpyHostToDevice));
  }
  return GRB_SUCCESS;
}

// CSR -> CSC by counting sort (keeps row order inside every column).
static void transpose_compressed(Index nmajor, Index nminor, const std::vector<Index>& ptr,
                                 const std::vector<Index>& ind, const std::vector<uint32_t>& val,
                                 std::vector<Index>* tptr, std::vector<Index>* tind, std::vector<uint32_t>* tval) {
  const size_t nvals = ind.size();
  tptr->assign((size_t)nminor + 1, 0);
  tind->resize(nvals);
  tval->resize(nvals);
  for (size_t i = 0; i < nvals; ++i) (*tptr)[(size_t)ind[i] + 1]++;
  for (Index c = 0; c < nminor; ++c) (*tptr)[(size_t)c + 1] += (*tptr)[c];
  std::vector<Index> cursor(tptr->begin(), tptr->end() - 1);
  for (Index r = 0; r < nmajor; ++r)
    for (Index p = ptr[r]; p < ptr[(size_t)r + 1]; ++p) {
      Index dst = cursor[ind[p]]++;
      (*tind)[dst] = r;
      (*tval)[dst] = val[p];
    }
}

// GRB_SPARSE_MATRIX_FORMAT = 1 (CSR only): h_csc* / d_csc* alias the CSR arrays
// (sparse_matrix.hpp:311-319, 384-391); whatever CSC a build produced is dropped.
static grb_info apply_format(grb_matrix A) {
  if (A->format != 1 || A->csc_alias || !A->csr.ptr) return GRB_SUCCESS;
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  if (A->owned && A->csc.ptr != A->csr.ptr) {
    if (A->csc.ptr) (void)hipFree(A->csc.ptr);
    if (A->csc.ind) (void)hipFree(A->csc.ind);
    if (A->csc.val) (void)hipFree(A->csc.val);
  }
  A->csc = A->csr;
  A->csc_alias = true;
  A->h_csc_ptr = A->h_csr_ptr;
  A->h_csc_ind = A->h_csr_ind;
  A->h_csc_val = A->h_csr_val;
  free_spmv_plan(&A->plan_csc);
  return build_spmv_plan(A->h_csr_ptr, A->nrows, A->ncols, &A->plan_csc);
}

static grb_info matrix_finish_build(grb_matrix A) {
  A->owned = true;
  GRB_TRY(upload(&A->csr, A->nrows, A->nvals, A->h_csr_ptr, A->h_csr_ind, A->h_csr_val));
  GRB_TRY(upload(&A->csc, A->ncols, A->nvals, A->h_csc_ptr, A->h_csc_ind, A->h_csc_val));
  GRB_TRY(build_spmv_plan(A->h_csr_ptr, A->nrows, A->ncols, &A->plan_csr));
  GRB_TRY(build_spmv_plan(A->h_csc_ptr, A->ncols, A->nrows, &A->plan_csc));
  A->built = true;
  return apply_format(A);
}

static grb_info finish_device_build(grb_matrix A) {
  GRB_TRY(build_spmv_plan(A->h_csr_ptr, A->nrows, A->ncols, &A->plan_csr));
  GRB_TRY(build_spmv_plan(A->h_csc_ptr, A->ncols, A->nrows, &A->plan_csc));
  A->built = true;
  return apply_format(A);
}

// build(): the coordinate list is uploaded and sorted / compressed on the device (build.hip);
// ties keep input order, duplicates are kept (util.hpp:501-559).
grb_info grb_matrix_build(grb_matrix A, const grb_index* rows, const grb_index* cols, const void* values,
                          grb_index nvals) { GRB_API_ENTER();
  if (!A) return GRB_UNINITIALIZED_OBJECT;
  if (nvals < 0) return GRB_INVALID_VALUE;
  for (Index i = 0; i < nvals; ++i)
    if (rows[i] < 0 || rows[i] >= A->nrows || cols[i] < 0 || cols[i] >= A->ncols) return GRB_INDEX_OUT_OF_BOUNDS;
  GRB_TRY(ctx_init());
  matrix_release_device(A);
  Index *d_r = nullptr, *d_c = nullptr;
  void* d_v = nullptr;
  const size_t cap = nvals > 0 ? (size_t)nvals : 1;
  grb_info info = GRB_SUCCESS;
  if (hipMalloc((void**)&d_r, 4 * cap) != hipSuccess || hipMalloc((void**)&d_c, 4 * cap) != hipSuccess ||
      hipMalloc(&d_v, 4 * cap) != hipSuccess)
    info = GRB_OUT_OF_MEMORY;                              // whatever was allocated is released below
  if (info == GRB_SUCCESS && nvals > 0) {
    if (hipMemcpy(d_r, rows, 4 * cap, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_c, cols, 4 * cap, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_v, values, 4 * cap, hipMemcpyHostToDevice) != hipSuccess)
      info = GRB_PANIC;
  }
  if (info == GRB_SUCCESS) info = device_build_from_coo(A, d_r, d_c, d_v, nvals, 0);
  (void)hipFree(d_r); (void)hipFree(d_c); (void)hipFree(d_v);
  GRB_TRY(info);
  return finish_device_build(A);
}

// The loader's semantics (util.hpp:197-329) on a DEVICE coordinate list: flags bit 0 add the
// reverse of every off-diagonal entry, bit 1 drop self loops, bit 2 drop duplicates (first
// occurrence wins); d_values may be NULL (pattern: every value 1).
grb_info grb_matrix_ingest_device(grb_matrix A, const grb_index* d_rows, const grb_index* d_cols, const void* d_values,
                                  grb_index nvals, int flags) { GRB_API_ENTER();
  if (!A) return GRB_UNINITIALIZED_OBJECT;
  if (nvals < 0 || (nvals > 0 && (!d_rows || !d_cols))) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  matrix_release_device(A);
  GRB_TRY(device_build_from_coo(A, d_rows, d_cols, d_values, nvals, flags));
  return finish_device_build(A);
}

grb_info grb_matrix_build_csr(grb_matrix A, const grb_index* csr_ptr, const grb_index* csr_ind, const void* csr_val,
                              grb_index nvals, const grb_index* csc_ptr, const grb_index* csc_ind,
                              const void* csc_val) { GRB_API_ENTER();
  if (!A || !csr_ptr) return GRB_UNINITIALIZED_OBJECT;
  matrix_release_device(A);
  A->nvals = nvals;
  A->h_csr_ptr.assign(csr_ptr, csr_ptr + A->nrows + 1);
  A->h_csr_ind.assign(csr_ind, csr_ind + nvals);
  A->h_csr_val.assign((const uint32_t*)csr_val, (const uint32_t*)csr_val + nvals);
  if (csc_ptr) {
    A->h_csc_ptr.assign(csc_ptr, csc_ptr + A->ncols + 1);
    A->h_csc_ind.assign(csc_ind, csc_ind + nvals);
    A->h_csc_val.assign((const uint32_t*)csc_val, (const uint32_t*)csc_val + nvals);
  } else {
    transpose_compressed(A->nrows, A->ncols, A->h_csr_ptr, A->h_csr_ind, A->h_csr_val, &A->h_csc_ptr,
                         &A->h_csc_ind, &A->h_csc_val);
  }
  return matrix_finish_build(A);
}

grb_info grb_matrix_adopt_device_csr(grb_matrix A, grb_index* d_csr_ptr, grb_index* d_csr_ind, void* d_csr_val,
                                     grb_index nvals, grb_index* d_csc_ptr, grb_index* d_csc_ind, void* d_csc_val) { GRB_API_ENTER();
  if (!A || !d_csr_ptr) return GRB_UNINITIALIZED_OBJECT;
  matrix_release_device(A);
  A->owned = false;
  A->nvals = nvals;
  A->csr.ptr = d_csr_ptr; A->csr.ind = d_csr_ind; A->csr.val = d_csr_val; A->csr.n = A->nrows; A->csr.nvals = nvals;
  A->h_csr_ptr.resize((size_t)A->nrows + 1);
  GRB_HIP_TRY(hipMemcpy(A->h_csr_ptr.data(), d_csr_ptr, 4 * ((size_t)A->nrows + 1), hipMemcpyDeviceToHost));
  GRB_TRY(build_spmv_plan(A->h_csr_ptr, A->nrows, A->ncols, &A->plan_csr));
  if (d_csc_ptr) {
    A->csc.ptr = d_csc_ptr; A->csc.ind = d_csc_ind; A->csc.val = d_csc_val; A->csc.n = A->ncols; A->csc.nvals = nvals;
    A->h_csc_ptr.resize((size_t)A->ncols + 1);
    GRB_HIP_TRY(hipMemcpy(A->h_csc_ptr.data(), d_csc_ptr, 4 * ((size_t)A->ncols + 1), hipMemcpyDeviceToHost));
    GRB_TRY(build_spmv_plan(A->h_csc_ptr, A->ncols, A->nrows, &A->plan_csc));
  }
  A->h_csr_ind.clear(); A->h_csr_val.clear(); A->h_csc_ind.clear(); A->h_csc_val.clear();
  A->built = true;
  return apply_format(A);
}

grb_info grb_matrix_nrows(grb_matrix A, grb_index* n) { GRB_API_ENTER(); if (!A) return GRB_UNINITIALIZED_OBJECT; *n = A->nrows; return GRB_SUCCESS; }
grb_info grb_matrix_ncols(grb_matrix A, grb_index* n) { GRB_API_ENTER(); if (!A) return GRB_UNINITIALIZED_OBJECT; *n = A->ncols; return GRB_SUCCESS; }
grb_info grb_matrix_nvals(grb_matrix A, grb_index* n) { GRB_API_ENTER(); if (!A) return GRB_UNINITIALIZED_OBJECT; *n = A->nvals; return GRB_SUCCESS; }

static grb_info ensure_host_mirror(grb_matrix A, bool csc) {
  std::vector<Index>& ind = csc ? A->h_csc_ind : A->h_csr_ind;
  std::vector<uint32_t>& val = csc ? A->h_csc_val : A->h_csr_val;
  const CsrArrays& d = csc ? A->csc : A->csr;
  if ((Index)ind.size() != A->nvals && d.ind) {
    ind.resize((size_t)A->nvals);
    val.resize((size_t)A->nvals);
    if (A->nvals > 0) {
      GRB_HIP_TRY(hipMemcpy(ind.data(), d.ind, 4 * (size_t)A->nvals, hipMemcpyDeviceToHost));
      if (d.val) GRB_HIP_TRY(hipMemcpy(val.data(), d.val, 4 * (size_t)A->nvals, hipMemcpyDeviceToHost));
    }
  }
  return GRB_SUCCESS;
}

grb_info grb_matrix_host_csr(grb_matrix A, const grb_index** ptr, const grb_index** ind, const void** val) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(ensure_host_mirror(A, false));
  if (ptr) *ptr = A->h_csr_ptr.data();
  if (ind) *ind = A->h_csr_ind.data();
  if (val) *val = A->h_csr_val.data();
  return GRB_SUCCESS;
}
grb_info grb_matrix_host_csc(grb_matrix A, const grb_index** ptr, const grb_index** ind, const void** val) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  if (!A->csc.ptr) return GRB_NO_VALUE;
  GRB_TRY(ensure_host_mirror(A, true));
  if (ptr) *ptr = A->h_csc_ptr.data();
  if (ind) *ind = A->h_csc_ind.data();
  if (val) *val = A->h_csc_val.data();
  return GRB_SUCCESS;
}

grb_info grb_matrix_set_values(grb_matrix A, const void* csr_val) { GRB_API_ENTER();
  if (!A || !A->built || !A->owned) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(ensure_host_mirror(A, false));
  A->h_csr_val.assign((const uint32_t*)csr_val, (const uint32_t*)csr_val + A->nvals);
  std::vector<Index> tptr, tind;
  transpose_compressed(A->nrows, A->ncols, A->h_csr_ptr, A->h_csr_ind, A->h_csr_val, &tptr, &tind, &A->h_csc_val);
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  if (A->csc_alias) A->h_csc_val = A->h_csr_val;
  if (A->nvals > 0) {
    GRB_HIP_TRY(hipMemcpy(A->csr.val, A->h_csr_val.data(), 4 * (size_t)A->nvals, hipMemcpyHostToDevice));
    if (!A->csc_alias)
      GRB_HIP_TRY(hipMemcpy(A->csc.val, A->h_csc_val.data(), 4 * (size_t)A->nvals, hipMemcpyHostToDevice));
  }
  A->nonneg_values = -1; A->mean_value = -1.0; A->small_int_values = -1;
  matrix_values_changed(A);
  return GRB_SUCCESS;
}

// apply on the stored values of a matrix, in place, both orientations (an element-wise function of the value alone
// commutes with the transposition); the host mirrors are re-read on demand; every private copy of the values goes
grb_info grb_matrix_apply(grb_matrix C, grb_matrix mask, grb_accum accum, int unary, int binop, double scalar, grb_matrix A,
                          grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!C || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (mask) return GRB_NOT_IMPLEMENTED;                  // "SpMat apply masked not implemented yet" (apply.hpp:103-104)
  if (C != A) return GRB_NOT_IMPLEMENTED;                // the reference's callers apply in place (gsssp.cu:83)
  if (!A->owned) return GRB_INVALID_OBJECT;              // adopted storage belongs to the caller
  if (A->nvals > 0) {
    GRB_TRY(k_apply_unary(A->dtype, unary, binop, scalar, A->csr.val, A->csr.val, A->nvals));
    if (A->csc.val && A->csc.val != A->csr.val)
      GRB_TRY(k_apply_unary(A->dtype, unary, binop, scalar, A->csc.val, A->csc.val, A->nvals));
  }
  A->h_csr_val.clear(); A->h_csr_ind.clear();            // host mirrors are re-read on demand (keyed on the index array)
  A->h_csc_val.clear(); A->h_csc_ind.clear();
  A->nonneg_values = -1; A->mean_value = -1.0; A->small_int_values = -1;
  matrix_values_changed(A);
  return GRB_SUCCESS;
}

// ---- binary cache (sparse_matrix.hpp:328-348 write, :355-407 read; name rule util.hpp:340-357)
// file = int32 nrows, int32 nvals, int32 rowptr[nrows + 1], int32 colind[nvals]; values implied 1
grb_info grb_cache_name(const char* mtx_path, int is_undirected, char* out, size_t cap) { GRB_API_ENTER();
  if (!mtx_path || !out || cap == 0) return GRB_NULL_POINTER;
  std::string path(mtx_path);
  // dirname / basename (libgen semantics for the cases readMtx meets: a path to a regular file)
  size_t slash = path.find_last_of('/');
  std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : path.substr(0, slash));
  std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
  const char* env = getenv("GRB_UTIL_REMOVE_SELFLOOP");
  const bool remove_self_loops = !env || atoi(env) != 0;
  int k = snprintf(out, cap, "%s/.%s.%s.%s.%sbin", dir.c_str(), base.c_str(), is_undirected ? "ud" : "d",
                   remove_self_loops ? "nosl" : "sl", "");
  return (k < 0 || (size_t)k >= cap) ? GRB_INSUFFICIENT_SPACE : GRB_SUCCESS;
}

grb_info grb_matrix_write_cache(grb_matrix A, const char* path) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  if (!path) return GRB_NULL_POINTER;
  GRB_TRY(ensure_host_mirror(A, false));
  FILE* f = fopen(path, "wb");
  if (!f) return GRB_INVALID_VALUE;                      // "Error: Unable to open file for writing!"
  bool ok = fwrite(&A->nrows, 4, 1, f) == 1 && fwrite(&A->nvals, 4, 1, f) == 1 &&
            fwrite(A->h_csr_ptr.data(), 4, (size_t)A->nrows + 1, f) == (size_t)A->nrows + 1 &&
            (A->nvals == 0 || fwrite(A->h_csr_ind.data(), 4, (size_t)A->nvals, f) == (size_t)A->nvals);
  ok = (fclose(f) == 0) && ok;
  return ok ? GRB_SUCCESS : GRB_PANIC;
}

// Matrix::build(dat_name): the file's two arrays go to the device as they are (no text, no sort
// of the CSR side); the CSC side is made there by the stable column sort of build.hip.
grb_info grb_matrix_build_cache(grb_matrix A, const char* path) { GRB_API_ENTER();
  if (!A) return GRB_UNINITIALIZED_OBJECT;
  if (!path) return GRB_NULL_POINTER;
  FILE* f = fopen(path, "rb");
  if (!f) return GRB_NO_VALUE;                           // "Error: Unable to read file!"
  Index hdr[2] = {0, 0};
  if (fread(hdr, 4, 2, f) != 2 || hdr[0] < 0 || hdr[1] < 0) { fclose(f); return GRB_INVALID_VALUE; }
  const Index nrows = hdr[0], nvals = hdr[1];
  std::vector<Index> ptr((size_t)nrows + 1), ind((size_t)nvals);
  const bool ok = fread(ptr.data(), 4, ptr.size(), f) == ptr.size() &&
                  (nvals == 0 || fread(ind.data(), 4, ind.size(), f) == ind.size());
  fclose(f);
  if (!ok || ptr[0] != 0 || ptr[(size_t)nrows] != nvals) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  matrix_release_device(A);
  A->nrows = nrows;                                      // the file decides (ncols is assumed equal, :372-373)
  A->ncols = nrows;
  // rows of the entries from the pointer array, on the host (one pass), then the device build
  std::vector<Index> rows((size_t)nvals);
  for (Index r = 0; r < nrows; ++r) {
    if (ptr[(size_t)r + 1] < ptr[r] || ptr[(size_t)r + 1] > nvals) return GRB_INVALID_VALUE;
    for (Index p = ptr[r]; p < ptr[(size_t)r + 1]; ++p) rows[(size_t)p] = r;
  }
  Index *d_r = nullptr, *d_c = nullptr;
  const size_t cap = nvals > 0 ? (size_t)nvals : 1;
  grb_info info = GRB_SUCCESS;
  if (hipMalloc((void**)&d_r, 4 * cap) != hipSuccess || hipMalloc((void**)&d_c, 4 * cap) != hipSuccess)
    info = GRB_OUT_OF_MEMORY;
  if (info == GRB_SUCCESS && nvals > 0 &&
      (hipMemcpy(d_r, rows.data(), 4 * cap, hipMemcpyHostToDevice) != hipSuccess ||
       hipMemcpy(d_c, ind.data(), 4 * cap, hipMemcpyHostToDevice) != hipSuccess))
    info = GRB_PANIC;
  if (info == GRB_SUCCESS) info = device_build_from_coo(A, d_r, d_c, nullptr, nvals, 0);   // values = 1 (:378-379)
  (void)hipFree(d_r); (void)hipFree(d_c);
  GRB_TRY(info);
  return finish_device_build(A);
}

}  // extern "C"
