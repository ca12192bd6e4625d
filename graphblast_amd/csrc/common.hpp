// common.hpp -- shared declarations of libgrb_hip.so (gfx950 only).
//
// Objects behind the opaque handles of include/grb_hip.h, the semiring functor
// table (graphblas/stddef.hpp restated as compile-time traits so every kernel gets
// its operators inlined), wavefront/workgroup primitives for 64-lane waves, and the
// prototypes of the kernel launchers implemented in the *.hip files.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "grb_hip.h"

namespace grb {

typedef int32_t Index;

constexpr int kWave = 64;          // gfx950 wavefront width
constexpr int kBlock = 256;        // workgroup size used by every kernel (4 waves)
constexpr int kWavesPerBlock = kBlock / kWave;

// ----------------------------------------------------------------------------
// error handling: never exit(); HIP failures become GRB_PANIC with a message.
#define GRB_HIP_TRY(expr)                                                         \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      fprintf(stderr, "libgrb_hip: %s failed: %s (%s:%d)\n", #expr,               \
              hipGetErrorString(e__), __FILE__, __LINE__);                        \
      return GRB_PANIC;                                                           \
    }                                                                             \
  } while (0)

#define GRB_TRY(expr)                                                             \
  do {                                                                            \
    grb_info i__ = (expr);                                                        \
    if (i__ != GRB_SUCCESS) return i__;                                           \
  } while (0)

// ----------------------------------------------------------------------------
// Binary operators (graphblas/stddef.hpp:14-138)
enum BinOp {
  OP_LOR, OP_LAND, OP_LXOR, OP_EQ, OP_NE, OP_GT, OP_LT, OP_GE, OP_LE,
  OP_FIRST, OP_SECOND, OP_MIN, OP_MAX, OP_PLUS, OP_MINUS, OP_TIMES, OP_DIV
};

template <typename T> __host__ __device__ inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> __host__ __device__ inline T tmax(T a, T b) { return a > b ? a : b; }
template <> __host__ __device__ inline float tmin<float>(float a, float b) { return fminf(a, b); }
template <> __host__ __device__ inline float tmax<float>(float a, float b) { return fmaxf(a, b); }

template <int OP, typename T>
__host__ __device__ inline T binop(T a, T b) {
  if constexpr (OP == OP_LOR)    return (T)((a != (T)0) || (b != (T)0));
  if constexpr (OP == OP_LAND)   return (T)((a != (T)0) && (b != (T)0));
  if constexpr (OP == OP_LXOR)   return (T)((a != (T)0) != (b != (T)0));
  if constexpr (OP == OP_EQ)     return (T)(a == b);
  if constexpr (OP == OP_NE)     return (T)(a != b);
  if constexpr (OP == OP_GT)     return (T)(a > b);
  if constexpr (OP == OP_LT)     return (T)(a < b);
  if constexpr (OP == OP_GE)     return (T)(a >= b);
  if constexpr (OP == OP_LE)     return (T)(a <= b);
  if constexpr (OP == OP_FIRST)  return a;
  if constexpr (OP == OP_SECOND) return b;
  if constexpr (OP == OP_MIN)    return tmin<T>(a, b);
  if constexpr (OP == OP_MAX)    return tmax<T>(a, b);
  if constexpr (OP == OP_PLUS)   return a + b;
  if constexpr (OP == OP_MINUS)  return a - b;
  if constexpr (OP == OP_TIMES)  return a * b;
  if constexpr (OP == OP_DIV) {
    if constexpr (std::is_integral<T>::value) {   // integer: guard /0 (UB in C)
      return b == (T)0 ? (T)0 : (T)(a / b);
    } else {
      return a / b;
    }
  }
  return a;
}

template <typename T> struct Limits;
template <> struct Limits<float> {
  __host__ __device__ static float max() { return FLT_MAX; }
  __host__ __device__ static float min() { return FLT_MIN; }   // numeric_limits<float>::min()
};
template <> struct Limits<int> {
  __host__ __device__ static int max() { return INT_MAX; }
  __host__ __device__ static int min() { return INT_MIN; }
};

// Monoids (stddef.hpp:159-172). IDK: 0 zero, 1 one, 2 max(), 3 min().
template <int M> struct MonoidTraits;
#define GRB_DEF_MONOID(M, OP, IDK)                                   \
  template <> struct MonoidTraits<M> {                               \
    static constexpr int op = OP;                                    \
    static constexpr int idk = IDK;                                  \
  };
GRB_DEF_MONOID(GRB_PLUS_MONOID, OP_PLUS, 0)
GRB_DEF_MONOID(GRB_MULTIPLIES_MONOID, OP_TIMES, 1)
GRB_DEF_MONOID(GRB_MINIMUM_MONOID, OP_MIN, 2)
GRB_DEF_MONOID(GRB_MAXIMUM_MONOID, OP_MAX, 0)
GRB_DEF_MONOID(GRB_LOGICAL_OR_MONOID, OP_LOR, 0)
GRB_DEF_MONOID(GRB_LOGICAL_AND_MONOID, OP_LAND, 0)
GRB_DEF_MONOID(GRB_GREATER_MONOID, OP_GT, 3)
GRB_DEF_MONOID(GRB_CUSTOM_LESS_MONOID, OP_LT, 2)
GRB_DEF_MONOID(GRB_NOT_EQUAL_TO_MONOID, OP_NE, 2)
#undef GRB_DEF_MONOID

template <int M, typename T>
struct Monoid {
  static constexpr int op = MonoidTraits<M>::op;
  __host__ __device__ static T identity() {
    constexpr int k = MonoidTraits<M>::idk;
    if constexpr (k == 0) return (T)0;
    if constexpr (k == 1) return (T)1;
    if constexpr (k == 2) return Limits<T>::max();
    return Limits<T>::min();
  }
  __host__ __device__ static T add(T a, T b) { return binop<op, T>(a, b); }
};

// Semirings (stddef.hpp:195-213)
template <int SR> struct SemiringTraits;
#define GRB_DEF_SR(SR, MONOID, MUL)                                  \
  template <> struct SemiringTraits<SR> {                            \
    static constexpr int monoid = MONOID;                            \
    static constexpr int mul = MUL;                                  \
  };
GRB_DEF_SR(GRB_LOGICAL_OR_AND, GRB_LOGICAL_OR_MONOID, OP_LAND)
GRB_DEF_SR(GRB_PLUS_MULTIPLIES, GRB_PLUS_MONOID, OP_TIMES)
GRB_DEF_SR(GRB_MINIMUM_PLUS, GRB_MINIMUM_MONOID, OP_PLUS)
GRB_DEF_SR(GRB_MAXIMUM_MULTIPLIES, GRB_MAXIMUM_MONOID, OP_TIMES)
GRB_DEF_SR(GRB_PLUS_DIVIDES, GRB_PLUS_MONOID, OP_DIV)
GRB_DEF_SR(GRB_PLUS_GREATER, GRB_PLUS_MONOID, OP_GT)
GRB_DEF_SR(GRB_GREATER_PLUS, GRB_GREATER_MONOID, OP_PLUS)
GRB_DEF_SR(GRB_PLUS_MINUS, GRB_PLUS_MONOID, OP_MINUS)
GRB_DEF_SR(GRB_PLUS_LESS, GRB_PLUS_MONOID, OP_LT)
GRB_DEF_SR(GRB_CUSTOM_LESS_PLUS, GRB_CUSTOM_LESS_MONOID, OP_PLUS)
GRB_DEF_SR(GRB_MINIMUM_MULTIPLIES, GRB_MINIMUM_MONOID, OP_TIMES)
GRB_DEF_SR(GRB_MULTIPLIES_MULTIPLIES, GRB_MULTIPLIES_MONOID, OP_TIMES)
GRB_DEF_SR(GRB_NOT_EQUAL_TO_PLUS, GRB_NOT_EQUAL_TO_MONOID, OP_PLUS)
GRB_DEF_SR(GRB_MINIMUM_SELECT_SECOND, GRB_MINIMUM_MONOID, OP_SECOND)
GRB_DEF_SR(GRB_PLUS_NOT_EQUAL_TO, GRB_PLUS_MONOID, OP_NE)
GRB_DEF_SR(GRB_CUSTOM_LESS_LESS, GRB_CUSTOM_LESS_MONOID, OP_LT)
GRB_DEF_SR(GRB_MINIMUM_NOT_EQUAL_TO, GRB_MINIMUM_MONOID, OP_NE)
#undef GRB_DEF_SR

template <int SR, typename T>
struct Semiring {
  typedef Monoid<SemiringTraits<SR>::monoid, T> M;
  static constexpr int monoid = SemiringTraits<SR>::monoid;
  static constexpr int mulop = SemiringTraits<SR>::mul;
  __host__ __device__ static T identity() { return M::identity(); }
  __host__ __device__ static T add(T a, T b) { return M::add(a, b); }
  __host__ __device__ static T mul(T a, T b) { return binop<mulop, T>(a, b); }
};

// ---- semirings registered at run time (REGISTER_SEMIRING / REGISTER_MONOID of graphblas/stddef.hpp:140-191:
// any monoid = (binary operator, identity) with any binary operator as the multiply).  The 17 the reference
// itself registers are compiled as above; every other composition runs through ONE more instantiation of
// each kernel whose operators are selected by wave-uniform switches on a descriptor in constant memory
// (these kernels are bound by memory requests, not by a branch per element).
constexpr int GRB_RUNTIME_SR = 17;               // internal: never an id of the C ABI
constexpr int GRB_USER_SEMIRING_BASE = 64;       // ids grb_semiring_register hands out
struct RtSemiring {
  int add_op, mul_op;                            // BinOp codes
  float ident_f;
  int ident_i;
};
namespace {                                      // one copy per translation unit (no relocatable device code);
__device__ __constant__ RtSemiring d_rt_semiring;   // dispatch_semiring binds the copy of the unit it is
RtSemiring h_rt_semiring = {OP_PLUS, OP_TIMES, 0.f, 0};   // instantiated in, just before the kernels launch
}

template <typename T>
__host__ __device__ inline T binop_rt(int op, T a, T b) {
  switch (op) {
    case OP_LOR: return binop<OP_LOR, T>(a, b);
    case OP_LAND: return binop<OP_LAND, T>(a, b);
    case OP_LXOR: return binop<OP_LXOR, T>(a, b);
    case OP_EQ: return binop<OP_EQ, T>(a, b);
    case OP_NE: return binop<OP_NE, T>(a, b);
    case OP_GT: return binop<OP_GT, T>(a, b);
    case OP_LT: return binop<OP_LT, T>(a, b);
    case OP_GE: return binop<OP_GE, T>(a, b);
    case OP_LE: return binop<OP_LE, T>(a, b);
    case OP_FIRST: return a;
    case OP_SECOND: return b;
    case OP_MIN: return binop<OP_MIN, T>(a, b);
    case OP_MAX: return binop<OP_MAX, T>(a, b);
    case OP_PLUS: return binop<OP_PLUS, T>(a, b);
    case OP_MINUS: return binop<OP_MINUS, T>(a, b);
    case OP_TIMES: return binop<OP_TIMES, T>(a, b);
    default: return binop<OP_DIV, T>(a, b);
  }
}

__host__ __device__ inline const RtSemiring& rt_semiring() {
#if defined(__HIP_DEVICE_COMPILE__)
  return d_rt_semiring;
#else
  return h_rt_semiring;
#endif
}

template <typename T>
struct Semiring<GRB_RUNTIME_SR, T> {
  static constexpr int monoid = -1;              // no compile-time monoid: atomic_combine takes its CAS loop
  static constexpr int mulop = -1;
  __host__ __device__ static T identity() {
    if constexpr (std::is_same<T, float>::value) return rt_semiring().ident_f;
    else return (T)rt_semiring().ident_i;
  }
  __host__ __device__ static T add(T a, T b) { return binop_rt<T>(rt_semiring().add_op, a, b); }
  __host__ __device__ static T mul(T a, T b) { return binop_rt<T>(rt_semiring().mul_op, a, b); }
};

// objects.hip: the registry behind grb_semiring_register.  `builtin` >= 0 when the composition is one of the 17.
struct UserSemiring {
  int add_op, mul_op;
  double identity;
  int builtin;
};
bool user_semiring_lookup(int id, UserSemiring* out);
hipStream_t current_stream();                    // objects.hip: the library's stream (grb_set_stream)

template <int N> struct IntTag { static constexpr int value = N; };

// Runtime (semiring, dtype) -> compile-time dispatch: f(IntTag<SR>{}, T{}).
template <typename F>
inline grb_info dispatch_semiring(int sr, int dtype, F&& f) {
  if (sr >= GRB_USER_SEMIRING_BASE) {
    UserSemiring u;
    if (!user_semiring_lookup(sr, &u)) return GRB_INVALID_VALUE;
    if (u.builtin >= 0) return dispatch_semiring(u.builtin, dtype, f);      // the compiled kernels
    RtSemiring r;
    r.add_op = u.add_op;
    r.mul_op = u.mul_op;
    r.ident_f = (float)u.identity;
    r.ident_i = u.identity >= 2147483647.0 ? INT_MAX : (u.identity <= -2147483648.0 ? INT_MIN : (int)u.identity);
    h_rt_semiring = r;
    // constant memory is read by kernels still in flight on the stream: order the update behind them.  The copy's
    // source must outlive this call if the runtime does not stage it at once: a ring of static slots, not the stack
    static RtSemiring ring[32];
    static unsigned ring_at = 0;
    RtSemiring* src = &ring[ring_at++ & 31u];
    *src = r;
    GRB_HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(d_rt_semiring), src, sizeof(r), 0, hipMemcpyHostToDevice, current_stream()));
    if (dtype == GRB_F32) return f(IntTag<GRB_RUNTIME_SR>{}, float{});
    return f(IntTag<GRB_RUNTIME_SR>{}, int{});
  }
#define GRB_CASE(SR)                                                         \
  case SR:                                                                   \
    if (dtype == GRB_F32) return f(IntTag<SR>{}, float{});                   \
    return f(IntTag<SR>{}, int{});
  switch (sr) {
    GRB_CASE(GRB_LOGICAL_OR_AND) GRB_CASE(GRB_PLUS_MULTIPLIES) GRB_CASE(GRB_MINIMUM_PLUS)
    GRB_CASE(GRB_MAXIMUM_MULTIPLIES) GRB_CASE(GRB_PLUS_DIVIDES) GRB_CASE(GRB_PLUS_GREATER)
    GRB_CASE(GRB_GREATER_PLUS) GRB_CASE(GRB_PLUS_MINUS) GRB_CASE(GRB_PLUS_LESS)
    GRB_CASE(GRB_CUSTOM_LESS_PLUS) GRB_CASE(GRB_MINIMUM_MULTIPLIES)
    GRB_CASE(GRB_MULTIPLIES_MULTIPLIES) GRB_CASE(GRB_NOT_EQUAL_TO_PLUS)
    GRB_CASE(GRB_MINIMUM_SELECT_SECOND) GRB_CASE(GRB_PLUS_NOT_EQUAL_TO)
    GRB_CASE(GRB_CUSTOM_LESS_LESS) GRB_CASE(GRB_MINIMUM_NOT_EQUAL_TO)
    default: return GRB_INVALID_VALUE;
  }
#undef GRB_CASE
}

template <typename F>
inline grb_info dispatch_monoid(int m, int dtype, F&& f) {
#define GRB_CASE(M)                                                          \
  case M:                                                                    \
    if (dtype == GRB_F32) return f(IntTag<M>{}, float{});                    \
    return f(IntTag<M>{}, int{});
  switch (m) {
    GRB_CASE(GRB_PLUS_MONOID) GRB_CASE(GRB_MULTIPLIES_MONOID) GRB_CASE(GRB_MINIMUM_MONOID)
    GRB_CASE(GRB_MAXIMUM_MONOID) GRB_CASE(GRB_LOGICAL_OR_MONOID) GRB_CASE(GRB_LOGICAL_AND_MONOID)
    GRB_CASE(GRB_GREATER_MONOID) GRB_CASE(GRB_CUSTOM_LESS_MONOID) GRB_CASE(GRB_NOT_EQUAL_TO_MONOID)
    default: return GRB_INVALID_VALUE;
  }
#undef GRB_CASE
}

// Host-side evaluation of a semiring on doubles (identity(), add_op(3,5) probes).
double semiring_identity(int sr, int dtype);
double semiring_add(int sr, int dtype, double a, double b);
double monoid_identity(int m, int dtype);
int semiring_monoid(int sr);

// ----------------------------------------------------------------------------
// Wavefront / workgroup primitives (64-lane waves, 256-thread workgroups)
__device__ inline int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ inline int wave_id() { return threadIdx.x >> 6; }

template <typename T, typename F>
__device__ inline T wave_reduce(T v, F f) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = f(v, __shfl_xor(v, o, kWave));
  return v;
}

// Sum of a wave's 64 values, in every lane.  Seven data-parallel-primitive adds (row shifts inside the 16-lane rows, two
// row broadcasts) and one v_readlane instead of the twelve LDS-crossbar permutes of six 32-bit __shfl_xor steps: the
// one-launch kernels reduce their level totals in EVERY level, sixteen waves of a workgroup at once.
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes without a source (or masked out) read `old` = 0
  unsigned t = v + __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false)     // row_shr:1
                 + __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false)     // row_shr:2
                 + __builtin_amdgcn_update_dpp(0u, v, 0x113, 0xf, 0xf, false);    // row_shr:3: lane i holds i-3 .. i of its row
  t += __builtin_amdgcn_update_dpp(0u, t, 0x114, 0xf, 0xe, false);                // row_shr:4, banks 1-3: 8 lanes
  t += __builtin_amdgcn_update_dpp(0u, t, 0x118, 0xf, 0xc, false);                // row_shr:8, banks 2-3: lane 15 of a row = the row
  t += __builtin_amdgcn_update_dpp(0u, t, 0x142, 0xa, 0xf, false);                // row_bcast:15 into rows 1 and 3
  t += __builtin_amdgcn_update_dpp(0u, t, 0x143, 0xc, 0xf, false);                // row_bcast:31 into rows 2 and 3: lane 63 = the wave
  return (unsigned)__builtin_amdgcn_readlane((int)t, 63);
}

// Inclusive prefix sum over the wave's lanes by the same seven adds (after the row shifts every lane holds the prefix
// inside its row; the two broadcasts add the totals of the rows before).  Every lane of the wave must be active.
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  unsigned t = v + __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false)
                 + __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false)
                 + __builtin_amdgcn_update_dpp(0u, v, 0x113, 0xf, 0xf, false);
  t += __builtin_amdgcn_update_dpp(0u, t, 0x114, 0xf, 0xe, false);
  t += __builtin_amdgcn_update_dpp(0u, t, 0x118, 0xf, 0xc, false);
  t += __builtin_amdgcn_update_dpp(0u, t, 0x142, 0xa, 0xf, false);
  t += __builtin_amdgcn_update_dpp(0u, t, 0x143, 0xc, 0xf, false);
  return t;
}
// 64-bit sum of a wave: three 32-bit sums (the low word in two 16-bit halves, so that no partial sum can wrap)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  return (unsigned long long)wave_sum_u32(lo & 0xffffu) + ((unsigned long long)wave_sum_u32(lo >> 16) << 16) +
         ((unsigned long long)wave_sum_u32(hi) << 32);
}
// bitwise OR of a wave's 64 values, in every lane
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  unsigned t = v | __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false) | __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false) |
               __builtin_amdgcn_update_dpp(0u, v, 0x113, 0xf, 0xf, false);
  t |= __builtin_amdgcn_update_dpp(0u, t, 0x114, 0xf, 0xe, false);
  t |= __builtin_amdgcn_update_dpp(0u, t, 0x118, 0xf, 0xc, false);
  t |= __builtin_amdgcn_update_dpp(0u, t, 0x142, 0xa, 0xf, false);
  t |= __builtin_amdgcn_update_dpp(0u, t, 0x143, 0xc, 0xf, false);
  return (unsigned)__builtin_amdgcn_readlane((int)t, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  return ((unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32) | (unsigned long long)wave_or_u32((unsigned)v);
}
// min / max of a wave's 64 values, in every lane: the same shifts with v_min / v_max (a lane without a source reads
// the operation's identity)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
  unsigned t = mn(mn(v, __builtin_amdgcn_update_dpp(~0u, v, 0x111, 0xf, 0xf, false)),
                  mn(__builtin_amdgcn_update_dpp(~0u, v, 0x112, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(~0u, v, 0x113, 0xf, 0xf, false)));
  t = mn(t, __builtin_amdgcn_update_dpp(~0u, t, 0x114, 0xf, 0xe, false));
  t = mn(t, __builtin_amdgcn_update_dpp(~0u, t, 0x118, 0xf, 0xc, false));
  t = mn(t, __builtin_amdgcn_update_dpp(~0u, t, 0x142, 0xa, 0xf, false));
  t = mn(t, __builtin_amdgcn_update_dpp(~0u, t, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)t, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  unsigned t = mx(mx(v, __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false)),
                  mx(__builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0u, v, 0x113, 0xf, 0xf, false)));
  t = mx(t, __builtin_amdgcn_update_dpp(0u, t, 0x114, 0xf, 0xe, false));
  t = mx(t, __builtin_amdgcn_update_dpp(0u, t, 0x118, 0xf, 0xc, false));
  t = mx(t, __builtin_amdgcn_update_dpp(0u, t, 0x142, 0xa, 0xf, false));
  t = mx(t, __builtin_amdgcn_update_dpp(0u, t, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)t, 63);
}

// Reduce within aligned groups of L lanes (L power of two <= 64).
template <typename T, typename F>
__device__ inline T group_reduce(T v, int L, F f) {
  for (int o = L >> 1; o > 0; o >>= 1) v = f(v, __shfl_xor(v, o, kWave));
  return v;
}

// "Am I the last workgroup of this launch to get here?"  A single arrival counter costs
// ~12 ns per arriver because same-address atomics serialise (2048 workgroups: 25 us), so the
// arrivals go to eight counters first and only the last arriver of each goes on to the ninth.
// tickets: 9 x 32 unsigned ints (one 128 B line each), zero before the launch; the overall last
// arriver resets them.  Call from ONE thread per workgroup after its partials are stored
// write-through and drained (s_waitcnt vmcnt(0)).
__device__ inline bool last_workgroup_arrives(unsigned int* tickets) {
  const unsigned int G = gridDim.x;
  const unsigned int x = blockIdx.x & 7u;
  const unsigned int groups = G < 8u ? G : 8u;
  const unsigned int members = (G - x + 7u) / 8u;
  const unsigned int a = __hip_atomic_fetch_add(&tickets[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a + 1u != members) return false;
  const unsigned int b = __hip_atomic_fetch_add(&tickets[8 * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (b + 1u != groups) return false;
  for (unsigned int k = 0; k < 9u; ++k) __hip_atomic_store(&tickets[k * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// The part of a one-launch reduction that follows every thread's private fold (elementwise.hip: reduce_kernel;
// lazy.hip: a chain of element-wise calls whose last result is reduced in the same launch -- the SAME association
// order, so the value does not depend on which of the two ran): butterfly inside the wave, the waves in order, the
// workgroup's partial written through, an arrival ticket; the last workgroup to arrive folds the partials -- thread t
// takes t, t + 256, ... -- the same way and writes {value, seq} into the pinned mailbox.  smem: kWavesPerBlock words.
template <typename T, typename F>
__device__ inline void reduce_finish(T acc, F add, T identity, unsigned int* partial, unsigned int* ticket,
                                     unsigned long long* mail, int seq, unsigned int* smem, int* s_last) {
  auto bits = [](T x) { unsigned int u; memcpy(&u, &x, 4); return u; };
  auto from = [](unsigned int u) { T x; memcpy(&x, &u, 4); return x; };
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) acc = add(acc, __shfl_xor(acc, o, kWave));
  if (lane_id() == 0) smem[wave_id()] = bits(acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = smem[0];
    for (int w = 1; w < kWavesPerBlock; ++w) t = bits(add(from(t), from(smem[w])));
    __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *s_last = last_workgroup_arrives(ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!*s_last) return;
  unsigned int f = bits(identity);
  for (int j = threadIdx.x; j < (int)gridDim.x; j += kBlock) {
    const unsigned int pj = __hip_atomic_load(&partial[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f = bits(add(from(f), from(pj)));
  }
  {
    T fv = from(f);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) fv = add(fv, __shfl_xor(fv, o, kWave));
    f = bits(fv);
  }
  __syncthreads();
  if (lane_id() == 0) smem[wave_id()] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int t = smem[0];
    for (int w = 1; w < kWavesPerBlock; ++w) t = bits(add(from(t), from(smem[w])));
    __hip_atomic_store(&mail[0], ((unsigned long long)(unsigned int)seq << 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Exclusive prefix sum of one int per thread over a 256-thread workgroup.
// smem: at least kWavesPerBlock ints. Ends with a barrier, so smem may be reused.
__device__ inline int block_exclusive_scan(int v, int* smem, int& total) {
  const int lane = lane_id(), wid = wave_id();
  const int x = (int)wave_incl_scan_u32((unsigned)v);
  if (lane == kWave - 1) smem[wid] = x;
  __syncthreads();
  int wave_off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) {
    int s = smem[w];
    if (w < wid) wave_off += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return wave_off + x - v;
}

// Mask test of the reference ("castable to true", kernels/assign_dense.hpp:27):
// passes when (scmp && m == 0) || (!scmp && m != 0). Masks are 4-byte f32 or i32.
__device__ inline bool mask_nonzero(const void* mask, int mask_f32, Index i) {
  return mask_f32 ? (reinterpret_cast<const float*>(mask)[i] != 0.f)
                  : (reinterpret_cast<const int*>(mask)[i] != 0);
}
__device__ inline bool mask_pass(const void* mask, int mask_f32, int scmp, Index i) {
  return mask_nonzero(mask, mask_f32, i) != (scmp != 0);
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
// Grid for grid-stride kernels: enough workgroups to fill 256 CUs x 8, never zero.
inline int stream_grid(long long work_items, int per_block = kBlock) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// ----------------------------------------------------------------------------
// Library context: stream, grow-only scratch slots, pinned mailbox.
struct Context {
  hipStream_t stream = nullptr;
  static constexpr int kSlots = 12;
  void* slot[kSlots] = {nullptr};
  size_t slot_cap[kSlots] = {0};
  unsigned long long slot_epoch[kSlots] = {0};   // bumped by every scratch(i, ...): "nobody but me has had this slot since"
  int* h_mail = nullptr;            // pinned, host-coherent, 64 ints; [63] = published sequence number
  int* d_hmail = nullptr;           // device-side address of h_mail
  int mail_seq = 0;
  // the zeroed state block of the next persistent BFS launch, cleared right behind the previous one
  void* bfs_prezero_ptr = nullptr;
  size_t bfs_prezero_bytes = 0;
  unsigned long long* h_gran = nullptr;   // pinned, host-coherent: 8 x {value, seq} granules
  unsigned long long* d_hgran = nullptr;  // device-side address of h_gran
  int* d_mail = nullptr;            // device, 64 ints
  unsigned int* d_tickets = nullptr;  // device, 9 x 32 arrival counters (last_workgroup_arrives)
  // freed vector storage, reused by the next vector of the same size: the algorithm drivers
  // create and destroy their temporaries on every call, and hipMalloc / hipFree cost ~0.1 ms each
  std::unordered_map<size_t, std::vector<void*>> vec_pool;
  size_t vec_pool_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool inited = false;
  int num_cu = 256;
  // push accumulator (scratch slot 5) is kept filled with this identity between calls
  double acc_identity = 0.0;
  int acc_dtype = -1;
  size_t acc_cap = 0;
};
Context& ctx();
grb_info ctx_init();
// Device scratch of at least `bytes` in slot `i` (contents not preserved on growth).  Slots 4 and 5 are
// NOT scratch for anyone but k_spmspv: they hold its touched-bitmap and accumulator, which it leaves all-zero /
// all-identity between calls instead of clearing n words per call.
grb_info scratch(int i, size_t bytes, void** out);
// Copy `count` ints from device to host through the pinned mailbox; synchronises.
grb_info fetch_ints(const int* d_src, int count, int* h_dst);
// Host side of an in-kernel mailbox publish: wait until h_mail[63] == seq.
grb_info wait_mail(int seq);
// Wait until granules [0, count) of the granule mailbox carry tag `seq`; values -> out.
grb_info wait_granules(int seq, int count, unsigned int* out);

// ----------------------------------------------------------------------------
// Objects behind the handles
struct SpmvBlock {        // one wave tile of an SpMV
  int row_start, row_end; // rows [row_start, row_end)
  int nnz_start, nnz_end; // their nonzeros (or a slice of one long row)
  int slot;               // -1: whole rows; >=0: partial-result slot of a long row
};

struct SpmvPlan {         // built once per matrix orientation at build()
  Index nrows = 0;        // rows of this orientation
  Index nminor = 0;       // length of the input vector
  int ntiles = 0;         // wave tiles
  SpmvBlock* d_tiles = nullptr;
  int nlong = 0;          // rows longer than one tile, reduced in two steps
  int* d_long_row = nullptr;
  int* d_long_slot_ptr = nullptr;
  int nslots = 0;
  void* d_partials = nullptr;
  // hub packing, prepared on the device by the first SpMV of this orientation
  bool hub_ready = false;
  int nhot = 0;           // leading entries of the (packed) input vector staged in LDS
  Index npacked = 0;      // columns with at least one reference: the part of u that is packed
  Index* d_ind2 = nullptr;   // column ids renamed by descending reference count (nullptr: not renamed)
  Index* d_order = nullptr;  // [nminor] packed position -> original column
  void* d_u2 = nullptr;      // [nminor] packed copy of the input vector
  struct SpmvBands* bands = nullptr;   // column bands with an LDS prefix each (spmv_bands.hpp); null: one prefix
  struct SpmvCBand* cband = nullptr;   // row bands, entries sorted by column rank, 16-bit coded (spmv_cband.hpp)
  bool cband_tried = false;
  int csr_launches = 0;      // products of this orientation that went through the CSR kernel (the format's amortisation rule)
};

struct CsrArrays {
  Index* ptr = nullptr;   // [n+1]
  Index* ind = nullptr;   // [nvals]
  void* val = nullptr;    // [nvals] of dtype
  Index n = 0;            // number of rows of this orientation
  Index nvals = 0;        // stored entries
};

}  // namespace grb

struct grb_descriptor_s {
  int desc[GRB_NDESCFIELD];
  // Descriptor::loadArgs fields (backend/cuda/descriptor.hpp:84-122)
  int mxvmode = 0, niter = 0, max_niter = 0, directed = 0, timing = 0, nthread = 0;
  int transpose = 0, debug = 0;
  float switchpoint = 0.f, memusage = 0.f;
  int dirinfo = 0, struconly = 0, opreuse = 0, endbit = 0, sort = 0, atomic = 0;
  int earlyexit = 0, fusedmask = 0;
  // extension (0 = off = the reference's rule only): fused BFS also leaves push for pull when the
  // frontier's out-edges exceed edgeswitch * nnz
  float edgeswitch = 0.f;
  int lastmxv = GRB_PUSHONLY;
  std::vector<grb_algo_iter> iter_log;   // grb_descriptor_iter_log: per-iteration records of the last driver call
};

struct grb_vector_s {
  int dtype = GRB_F32;
  grb::Index nsize = 0;
  grb::Index nvals = 0;           // cached, refreshed by grb_vector_nvals
  int vec_type = GRB_UNKNOWN;
  float ratio = 0.f;              // backend::Vector::ratio_
  // SparseVector
  grb::Index* s_ind = nullptr;    // [nsize]
  void* s_val = nullptr;          // [nsize + 1]
  grb::Index s_nvals = 0;
  bool s_owned = true;
  grb::Index s_alloc_n = 0, d_alloc_n = 0;   // sizes the owned blocks were allocated for (pool keys)
  // DenseVector
  void* d_val = nullptr;          // [nsize]
  grb::Index d_nnz = 0;
  bool d_owned = true;
  // grb_vector_device_ptrs handed the raw storage out (torch / RCCL interop): from then on a caller may read or write it
  // without an API call, so nothing that touches this vector is ever deferred (lazy.hip); sticky
  bool exposed = false;
};

namespace grb {
// rows of >= 4096 entries cut into 4096-entry slices for the batched traversal (bfs_batch.hip)
struct BatchSlices {
  bool ready = false;
  int4* d_slices = nullptr;      // {vertex, first entry, end entry, big index}
  Index* d_rows = nullptr;       // the big rows' vertex ids
  int nslices = 0, nbig = 0;
  Index* d_range_off = nullptr;  // out-edges only: [nranges + 1][nbig] where a big row's entries enter each destination range
  int nranges = 0;
  Index* d_range_bounds = nullptr;   // [nranges + 1] first destination of each range: equal in-degree mass, at most 16 Ki rows
  int* d_range_ids = nullptr;        // the ranges of at most 2 Ki rows first (nsmall of them), then the wider ones
  int nsmall = 0;
};
int spmv_reuse_threshold(int set);
grb_info k_apply_unary(int dtype, int unary, int op, double scalar, const void* in, void* out, Index n);   // elementwise.hip
// ---- lazy.hip: the queue of element-wise calls.  EVERY entry point of the C ABI starts with one of these macros:
// GRB_API_ENTER flushes the queue (top-level calls only: depth 0 -> 1) before the function touches anything,
// GRB_API_ENTER_QUEUE is for the functions that may append to it.
enum { LZ_ADD_VV = 0, LZ_MULT_VV, LZ_ADD_VS, LZ_MULT_VS, LZ_DUP, LZ_ASSIGN /* u = the mask, sr = scmp, scalar = the value */ };
// The ABI is serialised: every entry point holds one process-wide recursive lock for its duration (ctypes releases the
// GIL, so two Python threads can be inside the library at once; the context's scratch slots, mailboxes and the queue
// below are shared state).  depth counts the nesting of entry points on the thread that holds the lock.
struct ApiScope {
  static int depth;
  static unsigned long long epoch;   // top-level entries so far, the traversal queue's own calls excepted (bfs_persist.hip:
                                     // a lane fences against the library's stream only when something else has been called)
  bool entered_ = false;
  grb_info enter(bool queue_aware, bool counts = true);
  ~ApiScope();
};
grb_info lazy_flush();
// grb_reduce_vector on the result of a pending chain: the chain's launch also folds it (lazy.hip).  *done = false: not
// this case -- the caller flushes and reduces as usual.
grb_info lazy_flush_reduce(grb_vector_s* u, int monoid, double* out, bool* done);
grb_info reduce_launch_prep(Index n, int* grid, unsigned int** d_partial, unsigned int** d_ticket);   // elementwise.hip
bool lazy_try(int kind, int sr, grb_vector_s* w, grb_vector_s* u, grb_vector_s* v, double scalar, grb_info* flush_info);
#define GRB_API_ENTER()                                                         \
  grb::ApiScope api_scope__;                                                    \
  do {                                                                          \
    const grb_info api_fi__ = api_scope__.enter(false);                         \
    if (api_fi__ != GRB_SUCCESS) return api_fi__;                               \
  } while (0)
#define GRB_API_ENTER_NOINFO() grb::ApiScope api_scope__; (void)api_scope__.enter(false)
// the traversal queue's own entry points (enqueue / wait / host times / lanes): flush like any other, not counted in epoch
#define GRB_API_ENTER_BFSQ()                                                    \
  grb::ApiScope api_scope__;                                                    \
  do {                                                                          \
    const grb_info api_fi__ = api_scope__.enter(false, false);                  \
    if (api_fi__ != GRB_SUCCESS) return api_fi__;                               \
  } while (0)
#define GRB_API_ENTER_QUEUE() grb::ApiScope api_scope__; (void)api_scope__.enter(true)
// host-only entry points that never look at a vector or a matrix (descriptor fields): no flush, so that the toggles an
// application puts between two element-wise calls (sssp.hpp:77-81) do not cut its chain
#define GRB_API_ENTER_HOST() grb::ApiScope api_scope__; (void)api_scope__.enter(true)
void spmv_plan_values_changed(SpmvPlan* plan);   // drops every private copy of the stored values (spmv.hip)
}  // namespace grb

struct grb_matrix_s {
  int dtype = GRB_F32;
  grb::Index nrows = 0, ncols = 0, nvals = 0;
  bool built = false;
  bool owned = true;
  // backend::SparseMatrixFormat read from GRB_SPARSE_MATRIX_FORMAT when the matrix is created
  // (sparse_matrix.hpp:34,45): 0 CSR + CSC, 1 CSR only -- then the "CSC" arrays ARE the CSR arrays
  // (sparse_matrix.hpp:311-319) and vxm / mxv ignore the mxvmode (operations.hpp:131-133, 258-260)
  int format = 0;
  bool csc_alias = false;
  // host mirrors (sparse_matrix.hpp:120-132)
  std::vector<grb::Index> h_csr_ptr, h_csr_ind, h_csc_ptr, h_csc_ind;
  std::vector<uint32_t> h_csr_val, h_csc_val;   // raw 4-byte values of dtype
  grb::CsrArrays csr, csc;                       // device
  grb::SpmvPlan plan_csr, plan_csc;
  bool plan_csr_pending = false;                 // the CSR side's plan is built at its first use (mxm results: most are never multiplied)
  unsigned int* d_no_in_edges = nullptr;         // bitmap: CSC column empty (built lazily by bfs_fused)
  unsigned int* d_empty_csr_rows = nullptr;      // bitmap: CSR row empty (built lazily by bfs_part)
  int nonneg_values = -1;                        // -1 unknown, else whether every stored value is >= 0 (sssp_persist)
  double mean_value = -1.0;                      // < 0 unknown (sssp_nearfar: the bucket width)
  int small_int_values = -1;                     // -1 unknown, else whether every stored value is an integer in [0, 2^20]
  grb::Index* d_pull_hint = nullptr;                  // per vertex: its in-neighbour of largest out-degree (bfs_fused)
  long long bfs_n_in = -1;                       // vertices with in-edges (-1 unknown) and whether csr.ptr == csc.ptr in
  bool bfs_out_is_in = false;                    // content (bfs_persist.hip: sparse pull levels, degree bookkeeping)
  // bfs_persist.hip, owner-computes push: destination ranges of equal in-edge mass, the big rows' numbers and where
  // each big row enters each range ([oc_nb + 1][oc_nrows]); oc_state: 0 not tried, 1 ready, -1 not applicable
  grb::Index* d_oc_bounds = nullptr;
  grb::Index* d_oc_off = nullptr;
  int* d_oc_bigidx = nullptr;
  int oc_nb = 0, oc_nrows = 0, oc_state = 0;
  // the same tables for the narrower grid of concurrent traversals (grb_bfs_set_lanes: the ranges are cut per workgroup)
  grb::Index* d_oc2_bounds = nullptr;
  grb::Index* d_oc2_off = nullptr;
  int* d_oc2_bigidx = nullptr;
  int oc2_nb = 0, oc2_nrows = 0, oc2_state = 0, oc2_grid = 0;
  grb::BatchSlices batch_in, batch_out;          // bfs_batch.hip, built lazily
  void* tc_prep = nullptr;                       // tc_count.hip: the degree-oriented lists of a lower triangle (built by the first count)
};

namespace grb { void tc_prep_free(grb_matrix_s* A); }   // tc_count.hip
// Every cache that holds a copy of the stored VALUES (not structure) is dropped: the SpMV band formats.  Called by whatever rewrites csr.val / csc.val in place.
inline void matrix_values_changed(grb_matrix_s* A) {
  grb::spmv_plan_values_changed(&A->plan_csr);
  grb::spmv_plan_values_changed(&A->plan_csc);
  grb::tc_prep_free(A);                          // (whether every value is 1 is part of what it found)
}

namespace grb {

// objects.hip: frees everything a matrix holds on the device (arrays it owns, plans, cached
// per-graph side arrays: skip bitmaps, pull hint, SpMV hub packing) and marks it unbuilt
void matrix_release_device(grb_matrix A);

// ---- kernel launchers (implemented in the *.hip files) -----------------------
// elementwise.hip
grb_info k_copy(void* dst, const void* src, size_t bytes);
grb_info k_fill(int dtype, void* d, double val, Index n);
grb_info k_fill_ascending(int dtype, void* d, Index n);
grb_info k_scatter_const(int dtype, void* d_dense, const Index* ind, double val, Index n);
grb_info k_scatter_vals(int dtype, void* d_dense, const Index* ind, const void* vals, Index n);
grb_info k_count_nonidentity(int dtype, const void* d, double identity, Index n, Index* count_out);
// dense -> sparse (flag = val != identity); struconly: indices only. Sorted by index.
grb_info k_dense2sparse(int dtype, const void* d_dense, double identity, Index n, Index* out_ind,
                        void* out_val /*nullable*/, Index* nvals_out);
// prune a sparse (ind, val) list: keep entries with val != prune_val.
grb_info k_sparse_prune(int dtype, Index* ind, void* val, Index n, double prune_val,
                        Index* nvals_out);
grb_info k_assign_dense_mask_dense(int dtype, void* w, Index n, const void* mask, int mask_f32,
                                   int scmp, double val);
grb_info k_assign_dense_mask_sparse(int dtype, void* w, const Index* mask_ind, Index mask_nvals,
                                    double val);
grb_info k_assign_sparse_mask_dense(int dtype, const Index* w_ind, void* w_val, Index w_nvals,
                                    const void* mask, int mask_f32, int scmp, double val);
grb_info k_reduce(int monoid, int dtype, const void* d, Index n, double* out);
grb_info k_reduce_rows(int monoid, int dtype, const Index* ptr, const void* val, Index nrows,
                       void* w);
grb_info k_ewise_add_dense_dense(int sr, int dtype, void* w, const void* u, const void* v, Index n);
grb_info k_ewise_add_const(int sr, int dtype, void* w, double identity, int reverse, Index n);
grb_info k_ewise_add_sparse_dense(int sr, int dtype, void* w, const Index* u_ind, const void* u_val,
                                  const void* v, Index u_nvals);
grb_info k_ewise_scalar(int sr, int dtype, int use_add, void* w, double val, Index n);
grb_info k_ewise_mult_dense_dense(int sr, int dtype, void* w, const void* mask, int mask_f32,
                                  const void* u, const void* v, Index n);
grb_info k_ewise_mult_dense_dense_spmask(int sr, int dtype, Index* w_ind, void* w_val,
                                         const Index* m_ind, const void* m_val, int mask_f32,
                                         Index m_nvals, const void* u, const void* v);
grb_info k_ewise_mult_sparse_dense(int sr, int dtype, Index* w_ind, void* w_val, const Index* u_ind,
                                   const void* u_val, Index u_nvals, const void* v, int reverse);
grb_info k_ewise_mult_sparse_dense_spmask(int sr, int dtype, Index* w_ind, void* w_val,
                                          const Index* m_ind, const void* m_val, int mask_f32,
                                          Index m_nvals, const Index* u_ind, const void* u_val,
                                          Index u_nvals, const void* v, int reverse);
grb_info k_zero_dense_identity(int dtype, const void* mask, int mask_f32, double identity,
                               const Index* u_ind, void* u_val, Index n);

grb_info k_scatter_indexed(int dtype, void* w, Index w_n, const int* idx, Index n, const void* u);
grb_info k_gather_indexed(int dtype, void* w, Index w_n, const int* idx, Index n, const void* u);

// spmv.hip
grb_info build_spmv_plan(const std::vector<Index>& ptr, Index n, Index nminor, SpmvPlan* plan);
}  // namespace grb
// the plan of one orientation, built now if its construction was put off (grb_matrix_s::plan_csr_pending)
inline grb_info matrix_ensure_plan(grb_matrix_s* A, bool tran) {
  if (tran || !A->plan_csr_pending) return GRB_SUCCESS;
  A->plan_csr_pending = false;
  return grb::build_spmv_plan(A->h_csr_ptr, A->nrows, A->ncols, &A->plan_csr);
}
namespace grb {
void free_spmv_plan(SpmvPlan* plan);
// build.hip: columns ranked by descending reference count on the device (d_other_ptr: the transposed
// orientation's pointer array, whose differences ARE the counts; nullptr: histogram of d_ind)
grb_info device_exclusive_scan_u32(unsigned int* d, long long n);   // build.hip
grb_info device_sort_pairs(unsigned long long* d_keys, unsigned int* d_pay, long long n, int lo_bits, int hi_bits);   // build.hip
grb_info device_sort_pairs_range(unsigned long long* d_keys, unsigned int* d_pay, long long n, int first_bit, int nbits);
grb_info device_rank_columns(const Index* d_ind, Index nvals, const Index* d_other_ptr, Index m, Index hot,
                             Index* d_order, Index* d_rank, long long* hot_refs, Index* nreferenced);
// other_ptr: pointer array of the transposed orientation (length nminor + 1) or nullptr; only read by the
// one-off hub-packing preparation
grb_info k_spmv(int sr, int dtype, const CsrArrays& M, SpmvPlan& plan, const void* u,
                const void* mask, int mask_f32, int scmp, int accum, void* w, const Index* other_ptr = nullptr);
int spmv_bands_setting(int set);   // spmv.hip: 0 = query
int k_spmv_cband_info(const SpmvPlan& plan, long long* groups, int* bands, int* items, int* hub_rows, int* iso,
                      long long* bytes_per_launch);   // spmv.hip: 0 when the column-sorted format is not prepared
int spmv_format_setting(int set);  // spmv.hip: < 0 = query; 0 CSR kernel, 1 auto, 2 column-sorted bands wherever allowed
grb_info k_spmv_plan_info(const CsrArrays& M, SpmvPlan& plan, const Index* other_ptr, int warm, int* bands,
                          long long* band_nnz, long long* pieces, int* nhot);
grb_info tc_count_try(grb_matrix_s* A, long long* count, bool* done);   // tc_count.hip
int tc_product_setting(int set);                 // tc_count.hip: < 0 queries
int sssp_nearfar_setting(int set, bool apply);   // sssp_nearfar.hip
int sssp_last_order(int set);                    // set < 0 queries
void sssp_last_work(long long* out3);            // near / far: vertices expanded, out-edges relaxed, vertices marked, all passes
grb_info sssp_nearfar_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* iterations,
                          double* succ, float* tight_ms, int* passes);
constexpr int kOcBin = 256;        // owner-computes push (bfs_persist.hip, bfs_part_run.hip): the ranges are cut on multiples of this many vertices
constexpr int kOcWords = 8192;     // ... and a range's slice of the visited bitmap is at most this many words (32 KiB of LDS)
// ---- the dense core of the masked product C<L> = L (+.x) L^T (mxm_core.hip; used by grb_tc_dense_core and by grb_mxm)
struct TcCoreDev {
  int K = 0, Wr = 0, nt = 0;
  Index n = 0, theta = 0;
  unsigned int nent = 0, nt_elems = 0;               // entries between core rows; elements of the core rows' T-lists
  int* rank = nullptr;                               // [n]: a vertex's number among the core rows, -1 outside
  Index* rows = nullptr;                             // [K]
  unsigned int* H = nullptr;                         // [K][Wr] bit rows
  unsigned short* pre = nullptr;                     // [K][Wr] bits of the row before the word
  unsigned int* rowstart = nullptr;                  // [K + 1] first entry of a row = the core mask's CSR pointers
  Index *pos = nullptr, *ccind = nullptr;            // [nent] an entry's position in the whole mask's CSR; its column (a rank)
  unsigned int *tptr = nullptr, *cscptr = nullptr;   // [K + 1]
  Index *tind = nullptr, *cscind = nullptr;
  void *tiles_popc = nullptr, *tiles_mfma = nullptr;
  int n_popc = 0, n_mfma = 0;
  std::vector<Index> h_mptr, h_tptr, h_cscptr;
  std::vector<void*> owned;
  TcCoreDev() = default;
  TcCoreDev(const TcCoreDev&) = delete;
  ~TcCoreDev();
};
grb_info tc_core_alloc(TcCoreDev* d, void** p, size_t bytes);   // device memory that lives as long as d
grb_info tc_core_rows(const Index* ptr, Index n, int k_want, TcCoreDev* d);
grb_info tc_core_bits(const Index* ptr, const Index* ind, TcCoreDev* d, bool split);
grb_info tc_core_tiles(TcCoreDev* d, int method, int dense_from, grb_tc_core_result* res);
grb_info tc_core_hproduct(const TcCoreDev* d, int* out, unsigned long long* total);
grb_info tc_core_split(const Index* ptr, const Index* ind, TcCoreDev* d, unsigned int* mval2);
grb_info tc_core_combine(int dtype, void* c_val, const TcCoreDev* d, const int* ch, const void* ct, unsigned int one_bits, const void* m_val,
                         int mask_f32);
grb_info oc_tables_build(const Index* d_ptr, const Index* d_ind, const std::vector<Index>& h_ptr, Index nrows, Index ncols, int G,
                         Index** d_bounds, Index** d_off, int** d_bigidx, int* nb, int* nbig, int max_words = kOcWords);
grb_info bfs_queue_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* levels,
                       long long* reached, unsigned long long* edges, float* tight_ms);   // sssp_nearfar.hip
bool bfs_queue_wanted(grb_matrix A, grb_descriptor desc);   // would bfs_queue_run take this traversal? (sssp_nearfar.hip)
// bfs_persist.hip: traversals queued without waiting (grb_bfs_fused_enqueue / grb_bfs_wait, bfs_fused.hip)
grb_info bfs_ticket_take(int* slot);
grb_info bfs_persistent_enqueue(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int slot, int* seq);
void bfs_ticket_store(int slot, int seq, const grb_bfs_result& res);
int bfs_ticket_state(int slot, int seq, grb_vector* v, grb_matrix* A, grb_descriptor* desc, grb_index* source, grb_bfs_result* parked,
                     int* max_niter = nullptr);
void bfs_ticket_release(int slot);
grb_info bfs_persistent_wait(int slot, int seq, int* levels, int* last_dir, long long* reached, unsigned long long* edges,
                             Index* nf_left, bool* hit_cap, float* tight_ms);
void bfs_host_times(double* enqueue_us, double* wait_us, long long* calls, bool reset);
int bfs_lanes_setting(int set);                  // traversals in flight at once (grb_bfs_set_lanes); set < 1 only queries
int bfs_co_setting(int set);                     // traversals per launch (grb_bfs_set_coschedule); set < 1 only queries
grb_info bfs_co_profile(int on, double* ms_total, int* launches, int* traversals);   // HIP events around those launches
bool bfs_co_pending();                           // traversals that have a ticket and no launch yet
grb_info bfs_co_flush();                         // ... launched now (every entry point but the queue's own does this first)
grb_info bfs_lanes_fence(hipStream_t s);         // s waits for what the lanes have in flight (before a whole-device grid)
void bfs_lanes_unfence();                        // the lanes' next launches wait for the library's stream again
grb_info k_spmv_masked_or(int dtype, const CsrArrays& M, const void* u, double identity,
                          const void* mask, int mask_f32, int scmp, int earlyexit, int opreuse,
                          const Index* hint /* per-row best neighbour, may be null */, void* w);

// spmspv.hip
grb_info k_spmspv(int sr, int dtype, const CsrArrays& M, Index out_size, int struconly,
                  const Index* u_ind, const void* u_val, Index u_nvals, const void* mask,
                  int mask_f32, int use_mask, int keep_when_mask_zero, Index* w_ind, void* w_val,
                  Index* w_nvals);

}  // namespace grb
