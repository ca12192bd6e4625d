// lazy.hip -- a short queue of element-wise calls, executed as ONE kernel.
//
// An application that calls the operations itself -- the reference's own algorithm/pr.hpp:66-80, sssp.hpp:67-88,
// cc.hpp:77-115 are such call sequences -- runs three to ten element-wise calls between two products: each is a
// launch that reads and writes whole vectors for a few flops per element.  eWiseAdd / eWiseMult (vector (x) vector,
// vector (x) scalar) and dup on DENSE, library-owned vectors are therefore not executed when they are called but
// QUEUED (at most kLazyMax steps, one size, one dtype), and the queue runs as one kernel -- every vector read once,
// every result written once, the steps applied to an element in call order -- when anything else happens: EVERY other
// entry point of the C ABI starts by flushing (GRB_API_ENTER, common.hpp), so no call can see, free, resize or hand
// out storage with work pending on it.  What is observable is what the calls would have left: the same functors
// (binop<>, the dense (x) dense identity rule of kernels/ewisemult.hpp:22-25), each step's result rounded where the
// separate kernels would have stored it (an optimisation barrier keeps a multiply and the next add from contracting
// into an fma), the Info codes and the storage bookkeeping done at call time.  A masked constant assign (dense w, dense
// mask of the same type) joins the queue too: sssp.hpp's and cc.hpp's tails are eWise calls around one.
//
// Not queued (flushed, then run as before): masks, sparse operands, adopted (caller-owned) storage -- the caller
// may look at it, or rewrite an input, without asking the library --, registered semirings, calls made from inside a
// driver (grb_pr, grb_cc, ...: depth > 1).  GRB_LAZY=0 / grb_set_lazy(0) switch the queue off.
#include "common.hpp"
#include <mutex>

namespace grb {

constexpr int kLazyMax = 6;       // steps per queue
constexpr int kLazyBufs = 8;      // distinct vectors per queue

struct LazyStep {
  int kind, sr;
  grb_vector w, u, v;
  double scalar;
};
struct LazyQueue {
  int n = 0;
  int dtype = 0;
  Index nsize = 0;
  LazyStep s[kLazyMax];
};
static LazyQueue g_lazy;
static int g_lazy_on = -1;
int ApiScope::depth = 0;
unsigned long long ApiScope::epoch = 0;
static std::recursive_mutex g_api_lock;   // held by every entry point (ApiScope): depth and the queue are only touched under it

bool lazy_enabled() {
  if (g_lazy_on < 0) { const char* e = getenv("GRB_LAZY"); g_lazy_on = (e && atoi(e) == 0) ? 0 : 1; }
  return g_lazy_on != 0;
}
int lazy_pending() { return g_lazy.n; }

// ---- the fused kernel: a small program over at most kLazyBufs buffers ------------------------------------------
struct LzOp {
  short kind, add_op, mul_op;
  signed char w, u, v;            // buffer slots (v < 0: the scalar)
  unsigned int scalar_bits, ident_bits;
};
struct LzProg {
  int n, nbuf;
  unsigned int load_mask, store_mask;
  LzOp op[kLazyMax];
  void* buf[kLazyBufs];
};

// A step's operands are picked by wave-uniform switches (the program sits in the kernel arguments), so decoding it
// costs scalar branches, not per-lane selects; a lane carries kLzVec consecutive elements per buffer (16-byte
// accesses: the buffers are library-owned, 256-byte aligned).
constexpr int kLzVec = 4;
template <typename T> struct LzVec { T e[kLzVec]; };

// The values a lane carries (kLzVec elements of each of up to kLazyBufs buffers) live in LDS, one column per thread:
// a step's operands are picked by slot numbers that are only known at run time, and a register array indexed that way
// ends up in scratch memory (measured: 144 bytes per lane of scratch, the fused kernel slower than the four it
// replaced); an LDS column is indexed for free.  No barrier anywhere: a thread only ever touches its own column.
__device__ __forceinline__ void lz_round(float& x) { asm volatile("" : "+v"(x)); }   // the value a store would have kept
__device__ __forceinline__ void lz_round(int& x) { asm volatile("" : "+v"(x)); }

// y = op(a, b) lane by lane, the operator chosen ONCE (binop_rt's switch outside the element loop)
template <typename T>
__device__ __forceinline__ LzVec<T> lz_binop(int op, const LzVec<T>& a, const LzVec<T>& b) {
  LzVec<T> y;
#define GRB_LZ_CASE(OP)                                                              \
  case OP:                                                                           \
    _Pragma("unroll") for (int c = 0; c < kLzVec; ++c) y.e[c] = binop<OP, T>(a.e[c], b.e[c]); \
    break;
  switch (op) {
    GRB_LZ_CASE(OP_LOR) GRB_LZ_CASE(OP_LAND) GRB_LZ_CASE(OP_LXOR) GRB_LZ_CASE(OP_EQ) GRB_LZ_CASE(OP_NE) GRB_LZ_CASE(OP_GT)
    GRB_LZ_CASE(OP_LT) GRB_LZ_CASE(OP_GE) GRB_LZ_CASE(OP_LE) GRB_LZ_CASE(OP_FIRST) GRB_LZ_CASE(OP_SECOND) GRB_LZ_CASE(OP_MIN)
    GRB_LZ_CASE(OP_MAX) GRB_LZ_CASE(OP_PLUS) GRB_LZ_CASE(OP_MINUS) GRB_LZ_CASE(OP_TIMES)
    default:
#pragma unroll
      for (int c = 0; c < kLzVec; ++c) y.e[c] = binop<OP_DIV, T>(a.e[c], b.e[c]);
      break;
  }
#undef GRB_LZ_CASE
  return y;
}

// the program on the kLzVec elements a lane holds of every buffer
template <typename T>
__device__ __forceinline__ void lz_steps(const LzProg& p, LzVec<T> (*r)[kBlock]) {
  const int me = threadIdx.x;
#pragma unroll 1
  for (int s = 0; s < p.n; ++s) {
    const LzOp o = p.op[s];
    const LzVec<T> a = r[o.u][me];
    LzVec<T> b;
    T ident, sc;
    memcpy(&ident, &o.ident_bits, 4);
    memcpy(&sc, &o.scalar_bits, 4);
    if (o.v >= 0) b = r[o.v][me];
    else {
#pragma unroll
      for (int c = 0; c < kLzVec; ++c) b.e[c] = sc;
    }
    LzVec<T> y;
    switch (o.kind) {
      case LZ_ADD_VV:
      case LZ_ADD_VS: y = lz_binop<T>(o.add_op, a, b); break;
      case LZ_MULT_VS: y = lz_binop<T>(o.mul_op, a, b); break;
      case LZ_MULT_VV: {                                             // ewisemult.hpp:22-25: identity where either operand is
        y = lz_binop<T>(o.mul_op, a, b);
#pragma unroll
        for (int c = 0; c < kLzVec; ++c)
          if (a.e[c] == ident || b.e[c] == ident) y.e[c] = ident;
        break;
      }
      case LZ_ASSIGN: {                                              // w = value where the mask (a) passes, else w stays
        y = r[o.w][me];
#pragma unroll
        for (int c = 0; c < kLzVec; ++c)
          if ((a.e[c] != (T)0) != (o.add_op != 0)) y.e[c] = sc;       // add_op carries scmp
        break;
      }
      default: y = a; break;                                          // LZ_DUP
    }
#pragma unroll
    for (int c = 0; c < kLzVec; ++c) lz_round(y.e[c]);
    r[o.w][me] = y;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void lazy_chain_kernel(LzProg p, Index n) {
  __shared__ LzVec<T> r[kLazyBufs][kBlock];
  const int me = threadIdx.x;
  const Index nv = n / kLzVec;
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.load_mask >> b) & 1u) r[b][me] = reinterpret_cast<const LzVec<T>*>(p.buf[b])[i];
    lz_steps<T>(p, r);
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.store_mask >> b) & 1u) reinterpret_cast<LzVec<T>*>(p.buf[b])[i] = r[b][me];
  }
  // the last n % kLzVec elements: one lane each, in element 0 of the vectors
  const Index t = nv * kLzVec + (Index)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t < n) {
    for (int b = 0; b < p.nbuf; ++b) {
      LzVec<T> x;
#pragma unroll
      for (int c = 0; c < kLzVec; ++c) x.e[c] = (T)0;
      if ((p.load_mask >> b) & 1u) x.e[0] = reinterpret_cast<const T*>(p.buf[b])[t];
      r[b][me] = x;
    }
    lz_steps<T>(p, r);
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.store_mask >> b) & 1u) reinterpret_cast<T*>(p.buf[b])[t] = r[b][me].e[0];
  }
}

// The same chain with a reduction of buffer `red` attached (grb_reduce_vector on a pending result): launched with the
// reduce kernel's grid, a thread folds the elements it computes in the order the reduce kernel reads them -- the 16-byte
// vectors i, i + stride, ... as add(add(acc, add(x, y)), add(z, w)), then its tail element -- and the launch ends with the
// reduce kernel's own tail (common.hpp: reduce_finish): the value is bit for bit the one the separate launch gives.
template <typename T>
__device__ __forceinline__ T lz_scalar_op(int op, T a, T b) {
  switch (op) {
    case OP_PLUS: return binop<OP_PLUS, T>(a, b);
    case OP_TIMES: return binop<OP_TIMES, T>(a, b);
    case OP_MIN: return binop<OP_MIN, T>(a, b);
    case OP_MAX: return binop<OP_MAX, T>(a, b);
    case OP_LOR: return binop<OP_LOR, T>(a, b);
    default: return binop<OP_LAND, T>(a, b);
  }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void lazy_chain_reduce_kernel(LzProg p, Index n, int red, int red_op, unsigned int ident_bits,
                                                                   unsigned int* partial, unsigned int* ticket,
                                                                   unsigned long long* mail, int seq) {
  __shared__ LzVec<T> r[kLazyBufs][kBlock];
  __shared__ unsigned int smem[kWavesPerBlock];
  __shared__ int s_last;
  const int me = threadIdx.x;
  const Index nv = n / kLzVec;
  T ident;
  memcpy(&ident, &ident_bits, 4);
  T acc = ident;
  auto add = [red_op](T a, T b) { return lz_scalar_op<T>(red_op, a, b); };
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.load_mask >> b) & 1u) r[b][me] = reinterpret_cast<const LzVec<T>*>(p.buf[b])[i];
    lz_steps<T>(p, r);
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.store_mask >> b) & 1u) reinterpret_cast<LzVec<T>*>(p.buf[b])[i] = r[b][me];
    const LzVec<T> q = r[red][me];
    acc = add(add(acc, add(q.e[0], q.e[1])), add(q.e[2], q.e[3]));
  }
  const Index t = nv * kLzVec + (Index)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t < n) {
    for (int b = 0; b < p.nbuf; ++b) {
      LzVec<T> x;
#pragma unroll
      for (int c = 0; c < kLzVec; ++c) x.e[c] = (T)0;
      if ((p.load_mask >> b) & 1u) x.e[0] = reinterpret_cast<const T*>(p.buf[b])[t];
      r[b][me] = x;
    }
    lz_steps<T>(p, r);
    for (int b = 0; b < p.nbuf; ++b)
      if ((p.store_mask >> b) & 1u) reinterpret_cast<T*>(p.buf[b])[t] = r[b][me].e[0];
    acc = add(acc, r[red][me].e[0]);
  }
  reduce_finish<T>(acc, add, ident, partial, ticket, mail, seq, smem, &s_last);
}

// one step by the kernels the eager path uses (a queue of one step, or a step the program cannot hold)
static grb_info lazy_run_step(const LazyStep& st, int dtype, Index n) {
  switch (st.kind) {
    case LZ_ADD_VV: return k_ewise_add_dense_dense(st.sr, dtype, st.w->d_val, st.u->d_val, st.v->d_val, n);
    case LZ_MULT_VV: return k_ewise_mult_dense_dense(st.sr, dtype, st.w->d_val, nullptr, 0, st.u->d_val, st.v->d_val, n);
    case LZ_ADD_VS:
    case LZ_MULT_VS:
      if (st.u != st.w) GRB_TRY(k_copy(st.w->d_val, st.u->d_val, 4 * (size_t)n));
      return k_ewise_scalar(st.sr, dtype, st.kind == LZ_ADD_VS ? 1 : 0, st.w->d_val, st.scalar, n);
    case LZ_ASSIGN:
      return k_assign_dense_mask_dense(dtype, st.w->d_val, n, st.u->d_val, dtype == GRB_F32 ? 1 : 0, st.sr, st.scalar);
    default:
      if (n > 0 && st.u != st.w) return k_copy(st.w->d_val, st.u->d_val, 4 * (size_t)n);
      return GRB_SUCCESS;
  }
}

static grb_info lazy_run_program(const LazyQueue& q);

grb_info lazy_flush() {
  if (g_lazy.n == 0) return GRB_SUCCESS;
  LazyQueue q = g_lazy;             // the queue is empty while its steps run (they use internal kernels only)
  g_lazy.n = 0;
  if (q.nsize <= 0) return GRB_SUCCESS;
  if (q.n == 1) return lazy_run_step(q.s[0], q.dtype, q.nsize);
  if (lazy_run_program(q) == GRB_SUCCESS) return GRB_SUCCESS;
  // the fused program could not be built or launched: the calls were answered GRB_SUCCESS when they were queued, so
  // their effect must still happen -- one step at a time through the kernels the eager path uses; only if THAT
  // fails does the caller of the flushing entry point see an error
  (void)hipGetLastError();
  for (int s = 0; s < q.n; ++s) GRB_TRY(lazy_run_step(q.s[s], q.dtype, q.nsize));
  return GRB_SUCCESS;
}

static grb_info lazy_build_program(const LazyQueue& q, LzProg& p);
static grb_info lazy_run_program(const LazyQueue& q) {
  // test hook: behave as if the fused program had been refused, so that lazy_flush's step-by-step fallback runs
  if (const char* e = getenv("GRB_LAZY_FORCE_STEPWISE")) if (atoi(e) != 0) return GRB_PANIC;
  LzProg p;
  GRB_TRY(lazy_build_program(q, p));
  if (q.dtype == GRB_F32)
    hipLaunchKernelGGL(lazy_chain_kernel<float>, dim3(stream_grid(q.nsize / kLzVec + 1, kBlock)), dim3(kBlock), 0, ctx().stream, p, q.nsize);
  else
    hipLaunchKernelGGL(lazy_chain_kernel<int>, dim3(stream_grid(q.nsize / kLzVec + 1, kBlock)), dim3(kBlock), 0, ctx().stream, p, q.nsize);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

static int g_fused_reductions = 0;
// grb_reduce_vector(u) while a chain is pending that WRITES u: one launch runs the chain and folds u.
grb_info lazy_flush_reduce(grb_vector u, int monoid, double* out, bool* done) {
  *done = false;
  if (g_lazy.n == 0 || !u || !out || u->vec_type != GRB_DENSE || !u->d_val) return GRB_SUCCESS;
  if (monoid < 0 || monoid > GRB_LOGICAL_AND_MONOID) return GRB_SUCCESS;       // the order-sensitive "monoids" keep their own kernel
  if (u->dtype != g_lazy.dtype || u->nsize != g_lazy.nsize || g_lazy.nsize <= 0) return GRB_SUCCESS;
  if (const char* e = getenv("GRB_LAZY_FORCE_STEPWISE")) if (atoi(e) != 0) return GRB_SUCCESS;
  bool written = false;
  for (int s = 0; s < g_lazy.n; ++s) written = written || g_lazy.s[s].w == u;
  if (!written) return GRB_SUCCESS;
  LazyQueue q = g_lazy;
  LzProg p;
  if (lazy_build_program(q, p) != GRB_SUCCESS) return GRB_SUCCESS;            // (the ordinary flush deals with it)
  int red = -1;
  for (int b = 0; b < p.nbuf; ++b)
    if (p.buf[b] == u->d_val) red = b;
  if (red < 0) return GRB_SUCCESS;
  // The queue is cleared only once the fused launch is out: the queued steps have been answered GRB_SUCCESS already, so if
  // the preparation or the launch fails they stay queued -- the caller's ordinary flush (which has a step-by-step
  // fallback) runs them, and the reduction takes its own kernel.
  int grid;
  unsigned int *d_partial, *d_ticket;
  if (reduce_launch_prep(q.nsize, &grid, &d_partial, &d_ticket) != GRB_SUCCESS) return GRB_SUCCESS;
  Context& c = ctx();
  const int seq = ++c.mail_seq;
  static const int ops[6] = {OP_PLUS, OP_TIMES, OP_MIN, OP_MAX, OP_LOR, OP_LAND};
  const double idv = monoid_identity(monoid, q.dtype);
  unsigned int ident_bits;
  if (q.dtype == GRB_F32) { const float f = (float)idv; memcpy(&ident_bits, &f, 4); }
  else { const int iv = (int)idv; memcpy(&ident_bits, &iv, 4); }
  if (q.dtype == GRB_F32)
    hipLaunchKernelGGL(lazy_chain_reduce_kernel<float>, dim3(grid), dim3(kBlock), 0, c.stream, p, q.nsize, red, ops[monoid], ident_bits,
                       d_partial, d_ticket, c.d_hgran, seq);
  else
    hipLaunchKernelGGL(lazy_chain_reduce_kernel<int>, dim3(grid), dim3(kBlock), 0, c.stream, p, q.nsize, red, ops[monoid], ident_bits,
                       d_partial, d_ticket, c.d_hgran, seq);
  if (hipGetLastError() != hipSuccess) return GRB_SUCCESS;   // (not launched: the steps are still queued, *done is false)
  g_lazy.n = 0;
  unsigned int raw = 0;
  GRB_TRY(wait_granules(seq, 1, &raw));
  if (q.dtype == GRB_F32) { float f; memcpy(&f, &raw, 4); *out = (double)f; }
  else *out = (double)(int)raw;
  *done = true;
  ++g_fused_reductions;
  return GRB_SUCCESS;
}

static grb_info lazy_build_program(const LazyQueue& q, LzProg& p) {
  memset(&p, 0, sizeof(p));
  p.n = q.n;
  unsigned int written = 0;
  auto slot = [&](grb_vector x) -> int {
    for (int b = 0; b < p.nbuf; ++b)
      if (p.buf[b] == x->d_val) return b;
    p.buf[p.nbuf] = x->d_val;
    return p.nbuf++;
  };
  for (int s = 0; s < q.n; ++s) {
    const LazyStep& st = q.s[s];
    LzOp& o = p.op[s];
    o.kind = (short)st.kind;
    int add_op = 0, mul_op = 0;
    unsigned int ident_bits = 0;
    const bool arithmetic = st.kind != LZ_DUP && st.kind != LZ_ASSIGN;
    const grb_info di = !arithmetic ? GRB_SUCCESS : dispatch_semiring(st.sr, q.dtype, [&](auto tag, auto t) -> grb_info {
      using T = decltype(t);
      constexpr int SR = decltype(tag)::value;
      if constexpr (SR == GRB_RUNTIME_SR) return GRB_INVALID_VALUE;
      else {
        add_op = MonoidTraits<SemiringTraits<SR>::monoid>::op;
        mul_op = SemiringTraits<SR>::mul;
        const T id = Semiring<SR, T>::identity();
        memcpy(&ident_bits, &id, 4);
        return GRB_SUCCESS;
      }
    });
    if (di != GRB_SUCCESS) return di;
    if (st.kind == LZ_ASSIGN) add_op = st.sr;                         // scmp
    o.add_op = (short)add_op;
    o.mul_op = (short)mul_op;
    o.ident_bits = ident_bits;
    const int su = slot(st.u);
    if (!((written >> su) & 1u)) p.load_mask |= 1u << su;
    int sv = -1;
    if (st.kind == LZ_ADD_VV || st.kind == LZ_MULT_VV) {
      sv = slot(st.v);
      if (!((written >> sv) & 1u)) p.load_mask |= 1u << sv;
    } else if (st.kind == LZ_ADD_VS || st.kind == LZ_MULT_VS || st.kind == LZ_ASSIGN) {
      if (q.dtype == GRB_F32) { const float f = (float)st.scalar; memcpy(&o.scalar_bits, &f, 4); }
      else { const int iv = (int)st.scalar; memcpy(&o.scalar_bits, &iv, 4); }
    }
    const int sw = slot(st.w);
    if (st.kind == LZ_ASSIGN && !((written >> sw) & 1u)) p.load_mask |= 1u << sw;    // w keeps its value where the mask fails
    written |= 1u << sw;
    p.store_mask |= 1u << sw;
    o.u = (signed char)su;
    o.v = (signed char)sv;
    o.w = (signed char)sw;
  }
  return GRB_SUCCESS;
}

// Queue the step if it can be queued (true), leaving the eager bookkeeping to the caller, which has done it already.
// Otherwise the queue is flushed -- the caller is about to touch data -- and false is returned.
bool lazy_try(int kind, int sr, grb_vector w, grb_vector u, grb_vector v, double scalar, grb_info* flush_info) {
  *flush_info = GRB_SUCCESS;
  const bool two = kind == LZ_ADD_VV || kind == LZ_MULT_VV;
  const bool arithmetic = kind != LZ_DUP && kind != LZ_ASSIGN;
  bool ok = lazy_enabled() && ApiScope::depth == 1 && (!arithmetic || (sr >= 0 && sr < GRB_N_SEMIRINGS)) && w && u && (!two || v);
  if (ok) {
    ok = u->vec_type == GRB_DENSE && u->d_owned && !u->exposed && u->d_val && w->d_owned && !w->exposed && w->d_val &&
         w->vec_type == GRB_DENSE && u->dtype == w->dtype && u->nsize == w->nsize && u->nsize > 0;
    if (ok && two) ok = v->vec_type == GRB_DENSE && v->d_owned && !v->exposed && v->d_val && v->dtype == w->dtype && v->nsize == w->nsize;
  }
  if (ok && g_lazy.n > 0 && (g_lazy.dtype != w->dtype || g_lazy.nsize != w->nsize || g_lazy.n == kLazyMax)) {
    *flush_info = lazy_flush();
    if (*flush_info != GRB_SUCCESS) return false;
  }
  if (ok) {                          // would the program still fit its buffer table?
    void* seen[kLazyBufs + 3];
    int ns = 0;
    auto add = [&](grb_vector x) {
      if (!x) return;
      for (int k = 0; k < ns; ++k)
        if (seen[k] == x->d_val) return;
      seen[ns++] = x->d_val;
    };
    for (int s = 0; s < g_lazy.n; ++s) { add(g_lazy.s[s].w); add(g_lazy.s[s].u); add(g_lazy.s[s].v); }
    const int before = ns;
    add(w); add(u); if (two) add(v);
    if (ns > kLazyBufs) {
      if (before == 0) ok = false;   // a single step never needs more than three
      else {
        *flush_info = lazy_flush();
        if (*flush_info != GRB_SUCCESS) return false;
      }
    }
  }
  if (!ok) {
    *flush_info = lazy_flush();
    return false;
  }
  if (g_lazy.n == 0) { g_lazy.dtype = w->dtype; g_lazy.nsize = w->nsize; }
  LazyStep& st = g_lazy.s[g_lazy.n++];
  st.kind = kind; st.sr = sr; st.w = w; st.u = u; st.v = two ? v : nullptr; st.scalar = scalar;
  return true;
}

grb_info ApiScope::enter(bool queue_aware, bool counts) {
  g_api_lock.lock();
  if (depth == 0 && counts) ++epoch;
  grb_info r = GRB_SUCCESS;
  if (depth == 0 && !queue_aware && g_lazy.n > 0) r = lazy_flush();
  // traversals that wait for a co-scheduled launch to fill: any entry point but the traversal queue's own launches them
  // first (it may read their labels' vector, free their matrix, or need the device for itself)
  if (depth == 0 && counts && !queue_aware && bfs_co_pending()) { const grb_info ci = bfs_co_flush(); if (r == GRB_SUCCESS) r = ci; }
  ++depth;
  entered_ = true;
  return r;
}
ApiScope::~ApiScope() {
  if (entered_) { --depth; g_api_lock.unlock(); }
}

}  // namespace grb

using namespace grb;

extern "C" {
// 1 (default; GRB_LAZY=0 in the environment starts with 0): element-wise calls on dense library-owned vectors are
// queued and fused; 0: every call runs when it is called.  on < 0 only queries.  Returns the previous value.
int grb_set_lazy(int on) {
  grb::ApiScope api_scope__;
  (void)api_scope__.enter(false);   // whatever is pending runs under the old setting
  const int before = lazy_enabled() ? 1 : 0;
  if (on >= 0) g_lazy_on = on ? 1 : 0;
  return before;
}
// steps waiting in the queue (tests: a chain really was deferred).  Does not flush.
int grb_lazy_pending(void) { return lazy_pending(); }
// reductions that ran inside a chain's launch so far (tests).  Does not flush.
int grb_lazy_fused_reductions(void) { return g_fused_reductions; }
}
