// persist_common.hpp -- what the one-launch ("persistent") algorithm kernels share: the
// write-through publish and the XCD-hierarchical grid barrier.  A persistent launch is one
// co-resident grid (one 1024-thread workgroup per CU); its phases are separated by
// grid_sync(), every spin of which is bounded: a barrier that cannot complete raises the
// abort flag, every workgroup leaves the kernel, and the host reports GrB_PANIC.
#pragma once
#include "common.hpp"

namespace grb {

constexpr int kPThreads = 1024;
constexpr int kPWaves = kPThreads / kWave;
constexpr unsigned kSpinLimit = 1u << 22;

struct GridBarrier {                // zeroed by the host before every launch
  unsigned xcd_count[8][32];        // one 128 B line per counter
  unsigned top_count[32];
  unsigned abort_flag[32];
};

// Everything one workgroup writes for another to read goes out as an agent-scope
// write-through store (or an atomic): the data is in memory when the store has completed,
// which __syncthreads() waits for, so the barrier needs no L2 write-back on the way in --
// only the L1/L2 invalidate on the way out (CDNA guide G16, recipe R1).  Labels are read by
// nobody but the host and stay ordinary stores.
template <typename V>
__device__ inline void publish(V* p, V v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The reading side of the same recipe: data another workgroup published during this launch
// is read with an agent-scope load (served coherently, not from this CU's L1), so that a
// barrier whose readers all use fresh() can skip the L1/L2 invalidate on the way out.
template <typename V>
__device__ inline V fresh(const V* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- grid barrier ---------------------------------------------------------------------
// Monotonic counters: generation g of a group of m arrivers completes when its counter
// reaches m * g.  Returns false when the barrier was abandoned (spin bound hit somewhere).
__device__ inline bool grid_sync(GridBarrier* st, unsigned& gen, bool invalidate = true) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = gen + 1;
    const unsigned G = gridDim.x;
    const unsigned x = blockIdx.x & 7u;
    const unsigned groups = G < 8u ? G : 8u;
    const unsigned members = (G - x + 7u) / 8u;
    const unsigned a = __hip_atomic_fetch_add(&st->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1u == members * g) {
      (void)__hip_atomic_fetch_add(&st->top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(&st->top_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * g) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit ||
          __hip_atomic_load(&st->abort_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&st->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    if (invalidate) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok = ok;
  }
  __syncthreads();
  ++gen;
  return s_ok != 0;
}

// The set bits of a wave's 64 bitmap words, one per lane per step.  A thread that walks its own word bit by bit
// is fine while the frontier is a few scattered vertices; a road network's wave front fills whole words, and
// then a handful of lanes each run 32 vertices' dependent memory chains one after the other while the rest of
// the machine idles (measured on the 4896^2 grid: 240 us per round for 0.4 M frontier vertices).  Here the
// wave's bits are numbered by a prefix sum over the lanes' popcounts and lane j of step t takes bit 64 t + j:
// its word by a 6-step search of the prefix (LDS), its position in the word by 5 popcount halvings.
// f(word_lane, bit) is called with word_lane < 0 for a lane without a bit in the last step.
struct WaveBits {
  int pre[kWave];
  unsigned int word[kWave];
};
template <typename F>
__device__ inline void wave_for_each_bit(WaveBits* sb, unsigned int w, int lane, F f) {
  const int cnt = __popc(w);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int y = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += y;
  }
  const int total = __shfl(incl, kWave - 1, kWave);
  if (total == 0) return;
  __builtin_amdgcn_wave_barrier();
  sb->pre[lane] = incl - cnt;
  sb->word[lane] = w;
  __builtin_amdgcn_wave_barrier();
  for (int j0 = 0; j0 < total; j0 += kWave) {
    const int j = j0 + lane;
    int L = -1, bit = 0;
    if (j < total) {
      L = 0;
#pragma unroll
      for (int step = kWave / 2; step > 0; step >>= 1)
        if (sb->pre[L + step] <= j) L += step;
      unsigned int x = sb->word[L];
      int k = j - sb->pre[L];                               // the k-th set bit of x (k < popc(x))
#pragma unroll
      for (int h = 16; h > 0; h >>= 1) {
        const int c = __popc(x & ((1u << h) - 1u));
        if (k >= c) { k -= c; bit += h; x >>= h; }
      }
    }
    f(L, bit);
  }
  __builtin_amdgcn_wave_barrier();
}


// The same over four words per lane (entry 4 L + k = word k of lane L): a pass of the kernel covers four times the
// words, so a sparse bitmap costs a quarter of the workgroup barriers and the slowest wave of a pass matters less.
constexpr int kBitsWords = 4;
struct WaveBits4 {
  int pre[kWave * kBitsWords];
  unsigned int word[kWave * kBitsWords];
};
template <typename F>
__device__ inline void wave_for_each_bit4(WaveBits4* sb, const unsigned int (&w)[kBitsWords], int lane, F f) {
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < kBitsWords; ++k) cnt += __popc(w[k]);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int y = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += y;
  }
  const int total = __shfl(incl, kWave - 1, kWave);
  if (total == 0) return;
  __builtin_amdgcn_wave_barrier();
  {
    int run = incl - cnt;
#pragma unroll
    for (int k = 0; k < kBitsWords; ++k) {
      sb->pre[lane * kBitsWords + k] = run;
      sb->word[lane * kBitsWords + k] = w[k];
      run += __popc(w[k]);
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int j0 = 0; j0 < total; j0 += kWave) {
    const int j = j0 + lane;
    int E = -1, bit = 0;
    if (j < total) {
      E = 0;
#pragma unroll
      for (int step = kWave * kBitsWords / 2; step > 0; step >>= 1)
        if (sb->pre[E + step] <= j) E += step;
      unsigned int x = sb->word[E];
      int k = j - sb->pre[E];
#pragma unroll
      for (int h = 16; h > 0; h >>= 1) {
        const int c = __popc(x & ((1u << h) - 1u));
        if (k >= c) { k -= c; bit += h; x >>= h; }
      }
    }
    f(E < 0 ? -1 : E / kBitsWords, E < 0 ? 0 : E % kBitsWords, bit);   // (lane, word of that lane, bit)
  }
  __builtin_amdgcn_wave_barrier();
}

}  // namespace grb
