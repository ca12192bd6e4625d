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
// (bid, G): the workgroup's number and the number of workgroups of the grid the barrier is for -- the launch's own
// (grid_sync) or one of the sub-grids that share a launch (bfs_persist.hip: co-scheduled traversals, each with its own
// GridBarrier; bid & 7 is still the XCD the workgroup runs on there)
__device__ inline bool grid_sync_at(GridBarrier* st, unsigned& gen, const unsigned bid, const unsigned G, bool invalidate = true) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = gen + 1;
    const unsigned x = bid & 7u;
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
__device__ inline bool grid_sync(GridBarrier* st, unsigned& gen, bool invalidate = true) {
  return grid_sync_at(st, gen, blockIdx.x, gridDim.x, invalidate);
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
incl = (int)wave_incl_scan_u32((unsigned)incl);
  const int total = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
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
incl = (int)wave_incl_scan_u32((unsigned)incl);
  const int total = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
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


struct __attribute__((packed, aligned(4))) Quad { Index x, y, z, w; };   // four consecutive column ids
#ifndef GRB_SPARSE_WORDS
#define GRB_SPARSE_WORDS 32
#endif
constexpr int kSparseWords = GRB_SPARSE_WORDS;  // bitmap words (of 32 vertices) a wave takes per step of a pull level with a sparse active set
constexpr int kPullProbe4 = 4;

// ---- pull levels: what a wave does with the rows its first probe (the hinted in-neighbour) did not settle --------
// They are QUEUED in the wave's LDS ({next entry, end}, id of the vertex inside the wave's block) and then taken
// dense: kPullR rows per lane and round for the first kPullProbe4 entries (one 16-byte load + four bitmap probes per
// row, all rounds' loads in flight together), survivors written back to the front of the queue; then the survivors'
// remaining entries are laid end to end and dealt to the lanes, 256 entries per step, whatever the rows' lengths
// (offsets by a wave scan, a lane's row by a 6-step search) -- a step is one index load + one probe for 256 entries.
// A row contributes at most `cap` entries per pass (64, doubling): a hub with an early hit is not read to its end.
// The first hit of a row is the smallest offset that hits (LDS atomicMin), so the inspected-edge count stays the
// sequential early-exit count of the oracle: entries up to and including the first hit, or all of them.
constexpr int kPullQueue = 8 * kWave;               // rows a wave can queue: every vertex of an 8-chunk block
constexpr int kPullR = 4;                           // rows per lane and round of the first-entries stage
template <int kQueue>
struct PullLdsT {
  int2 row[kQueue];                                 // {next entry, end}
  unsigned short id[kQueue];
  unsigned int found[32];                           // bit id: the row had a hit
  int off[kWave], nxt[kWave], hit[kWave];           // dealing a pass: first slot / first entry / smallest hitting offset
  WaveBits bits;                                    // sparse active sets: the active bits numbered (wave_for_each_bit)
  unsigned int fresh_bits[kSparseWords];            // ... and what they discovered, by word
};
typedef PullLdsT<kPullQueue> PullLds;

// T rows are queued; afterwards found[] has the bit of every queued row with an in-neighbour in vin.
// kFresh: the bitmap is probed with agent-scope loads (a level with few probes skips the L1 invalidate instead).
template <bool kFresh>
__device__ __forceinline__ unsigned int probe_word(const unsigned int* vin, Index w) {
  return kFresh ? fresh(&vin[w]) : vin[w];
}
// kR rows per lane and round in the first stage, kD entries per lane and step in the second (4 and 4 where the kernel has
// the registers; the sub-grid kernels built for six waves per SIMD take 2 and 2: 34 registers less at their fattest point)
template <bool kFresh, int kR = kPullR, int kD = 4, typename LDS = PullLds>
__device__ __forceinline__ void pull_queue_run(const Index* __restrict__ iind, long long innz, const unsigned int* vin,
                                               LDS& L, int lane, int T, unsigned long long& inspected) {
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int S = 0;                                                       // survivors, compacted to the front of the queue
  if (innz >= kPullProbe4) {
    const Index last = (Index)innz - kPullProbe4;
    for (int q0 = 0; q0 < T; q0 += kWave * kR) {
      int2 it[kR];
      int id[kR];
      Quad cq[kR];
#pragma unroll
      for (int k = 0; k < kR; ++k) {
        const int qi = q0 + k * kWave + lane;
        const bool valid = qi < T;
        it[k] = valid ? L.row[qi] : make_int2(0, 0);
        id[k] = valid ? (int)L.id[qi] : 0;
      }
#pragma unroll
      for (int k = 0; k < kR; ++k) {
        Index at = it[k].x;
        const int shift = at > last ? at - last : 0;               // only the final entries of the array
        at -= shift;
        cq[k] = *reinterpret_cast<const Quad*>(iind + at);
        for (int t = 0; t < shift; ++t) { cq[k].x = cq[k].y; cq[k].y = cq[k].z; cq[k].z = cq[k].w; }
      }
      unsigned int wq[kR][kPullProbe4];
#pragma unroll
      for (int k = 0; k < kR; ++k) {
        const Index len = it[k].y - it[k].x;
        const Index c4[kPullProbe4] = {cq[k].x, cq[k].y, cq[k].z, cq[k].w};
#pragma unroll
        for (int t = 0; t < kPullProbe4; ++t) wq[k][t] = probe_word<kFresh>(vin, t < len ? (c4[t] >> 5) : 0);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < kR; ++k) {
        const Index len = it[k].y - it[k].x;
        const Index c4[kPullProbe4] = {cq[k].x, cq[k].y, cq[k].z, cq[k].w};
        int first = -1;
#pragma unroll
        for (int t = kPullProbe4 - 1; t >= 0; --t)
          if (t < len && ((wq[k][t] >> (c4[t] & 31)) & 1u)) first = t;
        const int seen = first >= 0 ? first + 1 : (len < kPullProbe4 ? (int)len : kPullProbe4);
        inspected += (unsigned long long)seen;
        if (first >= 0) atomicOr(&L.found[id[k] >> 5], 1u << (id[k] & 31));
        const bool surv = first < 0 && len > kPullProbe4;
        const unsigned long long m = __ballot(surv);
        if (surv) {
          const int slot = S + __popcll(m & lt_mask);
          L.row[slot] = make_int2(it[k].x + kPullProbe4, it[k].y);
          L.id[slot] = (unsigned short)id[k];
        }
        S += __popcll(m);
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    S = T;                                                         // a matrix of fewer than four entries: all left over
  }
  for (int g0 = 0; g0 < S; g0 += kWave) {
    const int gi = g0 + lane;
    bool alive = gi < S;
    Index nx = 0, en = 0;
    int id = 0;
    if (alive) { const int2 r = L.row[gi]; nx = r.x; en = r.y; id = L.id[gi]; }
    alive = alive && nx < en;
    Index cap = kWave;
    while (__ballot(alive)) {
      const Index rem = alive ? (en - nx < cap ? en - nx : cap) : 0;
      Index incl = rem;
incl = (Index)wave_incl_scan_u32((unsigned)incl);
      const Index total = (Index)__builtin_amdgcn_readlane((int)incl, kWave - 1);
      __builtin_amdgcn_wave_barrier();
      L.off[lane] = incl - rem;
      L.nxt[lane] = nx;
      L.hit[lane] = 0x7fffffff;
      __builtin_amdgcn_wave_barrier();
      for (Index t0 = 0; t0 < total; t0 += kD * kWave) {
        int r[kD];
        Index o[kD], col[kD];
#pragma unroll
        for (int j = 0; j < kD; ++j) {
          const Index t = t0 + j * kWave + lane;
          r[j] = -1;
          o[j] = 0;
          col[j] = 0;
          if (t < total) {
            int x = 0;                                             // the last row whose first slot is <= t
#pragma unroll
            for (int step = kWave / 2; step > 0; step >>= 1)
              if (L.off[x + step] <= t) x += step;
            r[j] = x;
            o[j] = t - L.off[x];
            col[j] = iind[L.nxt[x] + o[j]];
          }
        }
        unsigned int w[kD];
#pragma unroll
        for (int j = 0; j < kD; ++j) w[j] = probe_word<kFresh>(vin, col[j] >> 5);
#pragma unroll
        for (int j = 0; j < kD; ++j)
          if (r[j] >= 0 && ((w[j] >> (col[j] & 31)) & 1u)) atomicMin(&L.hit[r[j]], (int)o[j]);
      }
      __builtin_amdgcn_wave_barrier();
      const int h = L.hit[lane];
      if (alive) {
        if (h != 0x7fffffff) {
          inspected += (unsigned long long)(h + 1);
          atomicOr(&L.found[id >> 5], 1u << (id & 31));
          alive = false;
        } else {
          inspected += (unsigned long long)rem;
          nx += rem;
          alive = nx < en;
        }
      }
      if (cap < 4096) cap <<= 1;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_wave_barrier();
}


}  // namespace grb
