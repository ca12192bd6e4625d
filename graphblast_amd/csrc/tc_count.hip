// Triangle COUNT without the product matrix (the reference's tc(ntris, A, B): B is its "buffer matrix",
// graphblas/algorithm/tc.hpp:14-43 -- mxm into B, reduce B, B never read again).
//
// The masked product C<L> = L (+.x) L^T intersects, per entry (i, j) of L, the lists of i and j AS THE CALLER LABELLED the
// graph: whatever vertex happens to have a small id is in everybody's list, rows of 10^5 entries meet rows of 10^5
// entries, and the pivot kernels of mxm.hip stream 49 G list elements on the bench's graph.  The SUM of the product does
// not depend on the orientation of the edges.  Here every edge points from its endpoint of lower degree to the one of
// higher degree (ties by id): list(v) = the neighbours of v that rank above it -- at most sqrt(2 nnz) of them, 1 624 on
// the bench's graph against 134 504 -- and the count is
//
//     sum over edges {p, q}, p the endpoint with the LONGER list, of | list(p) ^ list(q) |
//
// with list(p) as an LDS bitmap (the vertices are numbered by rank: list(p) holds numbers below p) or hash table of the
// workgroup that owns the pivot p, and the lists of p's partners streamed past it a wave to a list, 16 bytes per lane per
// step.  DESIGN.md 5.4a; docs/experiments.md R6.8 for how it got here.
#include "common.hpp"
#include <chrono>

namespace grb {

constexpr unsigned kTcEmpty = 0xffffffffu;
constexpr unsigned kTcPad = 0x7fffffffu;       // fills a list up to a multiple of four entries: no vertex has this number
constexpr int kTcSlots = 1024;               // partial sums (one address would serialise a million atomics)
#ifndef GRB_TC_SLOTS
#define GRB_TC_SLOTS 4
#endif

#ifndef GRB_TC_DEPTH
#define GRB_TC_DEPTH 2
#endif

// A list is kept in two parts: the numbers below 65 535 as 16-bit entries of D16, the others as 32-bit entries of D32
// (no vertex is numbered 65 535: the 16-bit code is the filling).  The numbering is by rank, so a list is mostly hubs:
// 86 % of the entries of the bench's graph are below 2^16.  And a list holds numbers BELOW its owner's: a pivot
// numbered below 65 536 has nothing to find in the 32-bit part of a partner, and does not stream it -- those pivots
// carry 93 % of the streamed elements, so nine tenths of the stream is two bytes an element, 512 elements a step.
// Every part starts at a multiple of 16 bytes and is filled up to one (0xffff / kTcPad, numbers no list holds): a
// lane's 16 bytes are entries of ONE part or nothing, and no entry needs a range check of its own -- the kernels are
// bound by their vector instructions (4 cycles of a SIMD per instruction and wave) as much as by the stream.
//
// A wave streams the lists of up to 64 partners (lane l holds partner l's {first 16-bit entry, entries, first 32-bit
// entry, entries}): 64 lanes x 16 bytes a step, kDepth steps' loads in flight -- the same part's next bytes, the
// partner's other part or the next partner's first bytes, whichever follows -- before the oldest step's elements are
// looked up.  The batch's steps are counted first: the loop is a counted loop whose control is the scalar unit's alone.
// pair(w): two 16-bit entries of a partner's list (or fillings) as they lie in a 32-bit word; look(x): one 32-bit entry.
// wide: the 32-bit parts are streamed too (wave-uniform).
template <int kDepth, typename F2, typename F>
__device__ __forceinline__ void tc_stream_batch(const unsigned short* __restrict__ D16, const int* __restrict__ D32, const int4 my,
                                                const bool wide, const int lane, F2&& pair, F&& look) {
  const unsigned mine = ((unsigned)my.y + 8 * kWave - 1) / (8u * kWave) + (wide ? ((unsigned)my.w + 4 * kWave - 1) / (4u * kWave) : 0u);
  const int total = (int)wave_sum_u32(mine);
  int gi = -1, gs = 1, gk = 0, ge1 = 0;         // the generator: partner gi, part gs (0: 16-bit), the next step starts at gk (of [.., ge1))
  int4 v[kDepth];
  int se1[kDepth], sk[kDepth], sw[kDepth];
  auto issue = [&](int d) {                     // (only called while a step is left)
    while (gk >= ge1) {                         // the next part that holds anything
      if (gs == 0 && wide) { gs = 1; } else { gs = 0; ++gi; }
      gk = __builtin_amdgcn_readlane(gs ? my.z : my.x, gi);
      ge1 = gk + __builtin_amdgcn_readlane(gs ? my.w : my.y, gi);
    }
    se1[d] = ge1; sk[d] = gk; sw[d] = gs;
    if (gs) {
      if (gk + 4 * lane < ge1) v[d] = *reinterpret_cast<const int4*>(D32 + gk + 4 * lane);
      gk += 4 * kWave;
    } else {
      if (gk + 8 * lane < ge1) v[d] = *reinterpret_cast<const int4*>(D16 + gk + 8 * lane);
      gk += 8 * kWave;
    }
  };
#pragma unroll
  for (int d = 0; d < kDepth; ++d)
    if (d < total) issue(d);
  for (int st = 0; st < total; st += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      if (st + d >= total) break;
      const unsigned w[4] = {(unsigned)v[d].x, (unsigned)v[d].y, (unsigned)v[d].z, (unsigned)v[d].w};
      if (sw[d]) {
        if (sk[d] + 4 * lane < se1[d]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) look(w[c]);
        }
      } else {
        if (sk[d] + 8 * lane < se1[d]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) pair(w[c]);
        }
      }
      if (st + d + kDepth < total) issue(d);
    }
  }
}

// the numbers of a pivot's own list, one by one (both parts, fillings skipped)
template <int kThreads, typename F>
__device__ __forceinline__ void tc_own_list(const unsigned short* __restrict__ D16, const int* __restrict__ D32, const int4 own, const int tid, F&& put) {
  for (int i = tid; i < own.y; i += kThreads) put((unsigned)D16[own.x + i]);
  for (int i = tid; i < own.w; i += kThreads) put((unsigned)D32[own.z + i]);
}

// One task = a pivot and up to a few hundred of its partners: {pivot, first partner, partners, -}.  A pivot with many
// partners is cut into tasks that each build the table again (<= 2 len LDS operations against thousands of streamed
// elements).  The waves of a workgroup draw batches of 64 partners from an LDS counter.
template <int kThreads, int kTable>
__global__ __launch_bounds__(kThreads) void tc_count_pivot_kernel(const unsigned short* __restrict__ D16, const int* __restrict__ D32,
                                                                   const int4* __restrict__ lists, const int4* __restrict__ P,
                                                                   const int4* __restrict__ tasks, int ntasks, unsigned long long* total) {
  __shared__ unsigned table[kTable];
  __shared__ int next;
  __shared__ unsigned part[kThreads / kWave];
  constexpr int kWaves = kThreads / kWave;
  const int tid = threadIdx.x, lane = lane_id();
  unsigned count = 0;
  // (GRB_TC_WAVE_GRID: the one-wave instance with fewer workgroups than tasks -- measured, no gain: 8 192 workgroups 17.9 ms
  // per count, 65 536 16.7, one per task 16.6)
  for (int ti = blockIdx.x; ti < ntasks; ti += gridDim.x) {
    const int4 task = tasks[ti];
    const int4 own = lists[task.x];
    const int len = own.y + own.w;
    int need = len * GRB_TC_SLOTS - 1 > 63 ? len * GRB_TC_SLOTS - 1 : 63;
    if (need > kTable - 1) need = kTable - 1;
    const int lg = 32 - __clz(need);           // a table of 2^lg >= 4 len slots (2 len for the longest lists), at least 64
    const unsigned mask = (1u << lg) - 1u;
    const int shift = 32 - lg;
    if (kWaves > 1) __syncthreads();
    for (unsigned i = tid; i <= mask; i += kThreads) table[i] = kTcEmpty;
    if (tid == 0) next = kWaves;
    __syncthreads();
    tc_own_list<kThreads>(D16, D32, own, tid, [&](unsigned x) {
      unsigned slot = (x * 0x9E3779B1u) >> shift;
      while (atomicCAS(&table[slot], kTcEmpty, x) != kTcEmpty) slot = (slot + 1) & mask;
    });
    __syncthreads();
    const bool wide = task.x > 65536;           // the pivot's list reaches beyond the 16-bit numbers
    const int nbatch = (task.z + kWave - 1) / kWave;
    int b = kWaves > 1 ? wave_id() : 0;
    while (b < nbatch) {
      const int nb = task.z - b * kWave < kWave ? task.z - b * kWave : kWave;
      const int4 my = lane < nb ? P[task.y + b * kWave + lane] : make_int4(0, 0, 0, 0);
      auto look = [&](unsigned x) {
        unsigned slot = (x * 0x9E3779B1u) >> shift;
        unsigned t = table[slot];
        while (t != x && t != kTcEmpty) { slot = (slot + 1) & mask; t = table[slot]; }
        count += t == x;
      };
      tc_stream_batch<GRB_TC_DEPTH>(D16, D32, my, wide, lane, [&](unsigned w) { look(w & 0xffffu); look(w >> 16); }, look);
      if (kWaves > 1) {
        if (lane == 0) b = atomicAdd(&next, 1);
        b = __builtin_amdgcn_readfirstlane(b);
      } else {
        ++b;
      }
    }
  }
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) count += __shfl_down(count, off);
  if (kWaves == 1) {
    if (lane == 0 && count) atomicAdd(&total[blockIdx.x & (kTcSlots - 1)], (unsigned long long)count);
    return;
  }
  if (lane == 0) part[wave_id()] = count;
  __syncthreads();
  if (tid == 0) {
    unsigned long long sum = 0;
    for (int w = 0; w < kWaves; ++w) sum += part[w];
    if (sum) atomicAdd(&total[blockIdx.x & (kTcSlots - 1)], sum);
  }
}

// The pivots with short lists (<= 256 entries: 851 506 of the bench's graph, 2 % of the streamed elements): their partners'
// lists are no longer than theirs -- a dozen entries, two parts -- and a wave step per part leaves sixty lanes idle.  Here a
// LANE walks a partner: 16 bytes (eight 16-bit or four 32-bit entries) per step, 64 partners of the task at a time, the
// pivot's list in the wave's 512-slot hash table.  One wave per task, no barriers but the wave's own.
__global__ __launch_bounds__(kWave) void tc_count_small_kernel(const unsigned short* __restrict__ D16, const int* __restrict__ D32,
                                                               const int4* __restrict__ lists, const int4* __restrict__ P,
                                                               const int4* __restrict__ tasks, unsigned long long* total) {
  constexpr int kTable = 512;
  __shared__ unsigned table[kTable];
  const int lane = lane_id();
  const int4 task = tasks[blockIdx.x];
  const int4 own = lists[task.x];
  const int len = own.y + own.w;
  const int need = len * GRB_TC_SLOTS - 1 > 63 ? (len * GRB_TC_SLOTS - 1 < kTable - 1 ? len * GRB_TC_SLOTS - 1 : kTable - 1) : 63;
  const int lg = 32 - __clz(need);
  const unsigned mask = (1u << lg) - 1u;
  const int shift = 32 - lg;
  for (unsigned i = lane; i <= mask; i += kWave) table[i] = kTcEmpty;
  __syncthreads();
  tc_own_list<kWave>(D16, D32, own, lane, [&](unsigned x) {
    unsigned slot = (x * 0x9E3779B1u) >> shift;
    while (atomicCAS(&table[slot], kTcEmpty, x) != kTcEmpty) slot = (slot + 1) & mask;
  });
  __syncthreads();
  const bool wide = task.x > 65536;
  unsigned count = 0;
  auto look = [&](unsigned x) {
    unsigned slot = (x * 0x9E3779B1u) >> shift;
    unsigned t = table[slot];
    while (t != x && t != kTcEmpty) { slot = (slot + 1) & mask; t = table[slot]; }
    count += t == x;
  };
  for (int q = lane; q < task.z; q += kWave) {
    const int4 my = P[task.y + q];
    for (int k = my.x; k < my.x + my.y; k += 8) {
      const int4 v = *reinterpret_cast<const int4*>(D16 + k);
      const unsigned w[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) { look(w[c] & 0xffffu); look(w[c] >> 16); }
    }
    if (wide)
      for (int k = my.z; k < my.z + my.w; k += 4) {
        const int4 v = *reinterpret_cast<const int4*>(D32 + k);
        look((unsigned)v.x); look((unsigned)v.y); look((unsigned)v.z); look((unsigned)v.w);
      }
  }
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) count += __shfl_down(count, off);
  if (lane == 0 && count) atomicAdd(&total[blockIdx.x & (kTcSlots - 1)], (unsigned long long)count);
}

// The pivots numbered up to kBits (vertices are NUMBERED BY RANK, 0 = the highest degree, so list(p) holds numbers below
// p): the pivot's list is a BITMAP of p bits -- one LDS read and a bit test per streamed element, no hashing, no walk.
template <int kThreads, int kBits>
__global__ __launch_bounds__(kThreads) void tc_count_bitmap_kernel(const unsigned short* __restrict__ D16, const int* __restrict__ D32,
                                                                    const int4* __restrict__ lists, const int4* __restrict__ P,
                                                                    const int4* __restrict__ tasks, int short_upto, unsigned long long* total) {
  __shared__ unsigned bits[kBits / 32 + 1];
  __shared__ int next;
  __shared__ unsigned part[kThreads / kWave];
  constexpr int kWaves = kThreads / kWave;
  const int tid = threadIdx.x, lane = lane_id();
  const int4 task = tasks[blockIdx.x];
  const int4 own = lists[task.x];
  // the bitmap: task.x bits (task.x <= kBits) and never fewer than 65 536 -- a 16-bit entry is looked up without a bound
  // check -- and one word more that stays zero: every 32-bit entry from 32 nwords on looks there
  const int nwords = task.x > 65536 ? (task.x + 31) / 32 : 2048;
  for (int i = tid; i <= nwords; i += kThreads) bits[i] = 0u;
  if (tid == 0) next = kWaves;
  __syncthreads();
  tc_own_list<kThreads>(D16, D32, own, tid, [&](unsigned x) { atomicOr(&bits[x >> 5], 1u << (x & 31)); });
  __syncthreads();
  const unsigned top = (unsigned)nwords;
  const bool wide = task.x > 65536;
  unsigned count = 0;
  const int nbatch = (task.z + kWave - 1) / kWave;
  int b = wave_id();
  while (b < nbatch) {
    const int nb = task.z - b * kWave < kWave ? task.z - b * kWave : kWave;
    int4 my = lane < nb ? P[task.y + b * kWave + lane] : make_int4(0, 0, 0, 0);
    // the batch's SHORT partners are walked by their lanes, 16 bytes a step, all of them at once -- a wave step for a
    // list of twenty entries would leave sixty lanes idle; the wave then streams the others
    if (my.y + my.w > 0 && my.y <= short_upto && my.w <= short_upto / 2) {
      for (int k = my.x; k < my.x + my.y; k += 8) {
        const int4 v = *reinterpret_cast<const int4*>(D16 + k);
        const unsigned w[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned lo = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(bits) + ((w[c] >> 3) & 0x1ffcu));
          const unsigned hi = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(bits) + ((w[c] >> 19) & 0x1ffcu));
          count += __builtin_amdgcn_ubfe(lo, w[c], 1u) + __builtin_amdgcn_ubfe(hi, w[c] >> 16, 1u);
        }
      }
      if (wide)
        for (int k = my.z; k < my.z + my.w; k += 4) {
          const int4 v = *reinterpret_cast<const int4*>(D32 + k);
          const unsigned w[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned i = w[c] >> 5;
            count += __builtin_amdgcn_ubfe(bits[i < top ? i : top], w[c], 1u);
          }
        }
      my = make_int4(0, 0, 0, 0);
    }
    // (the fillings: bit 65 535 is nobody's, kTcPad looks at word `top`; v_bfe_u32 takes the low five bits of its offset)
    tc_stream_batch<GRB_TC_DEPTH>(D16, D32, my, wide, lane,
      [&](unsigned w) {
        const unsigned lo = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(bits) + ((w >> 3) & 0x1ffcu));
        const unsigned hi = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(bits) + ((w >> 19) & 0x1ffcu));
        count += __builtin_amdgcn_ubfe(lo, w, 1u) + __builtin_amdgcn_ubfe(hi, w >> 16, 1u);
      },
      [&](unsigned x) {
        const unsigned w = x >> 5;
        count += __builtin_amdgcn_ubfe(bits[w < top ? w : top], x, 1u);
      });
    if (lane == 0) b = atomicAdd(&next, 1);
    b = __builtin_amdgcn_readfirstlane(b);
  }
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) count += __shfl_down(count, off);
  if (lane == 0) part[wave_id()] = count;
  __syncthreads();
  if (tid == 0) {
    unsigned long long sum = 0;
    for (int w = 0; w < kWaves; ++w) sum += part[w];
    if (sum) atomicAdd(&total[blockIdx.x & (kTcSlots - 1)], sum);
  }
}

// ---- the orientation, on the device ------------------------------------------------------------------------------------
constexpr int kTcWaveLen = 256;               // lists up to here: a wave and a hash table of 512 slots
constexpr int kTcBits = 1 << 18;              // pivots numbered up to here: the bitmap kernel
constexpr int kTcHashLen = 4096;              // longer lists than this (of a pivot beyond kTcBits): not handled, the product runs
#ifndef GRB_TC_CHUNK
#define GRB_TC_CHUNK 2048
#endif
constexpr int kTcChunk[3] = {256, GRB_TC_CHUNK, 2048};   // partners per task, by kernel

// the row of every entry (a wave per row: stores only, a hub row is a few hundred of them)
__global__ __launch_bounds__(kBlock) void tc_rows_kernel(const Index* __restrict__ ptr, Index n, int* __restrict__ erow) {
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index r = (Index)blockIdx.x * kWavesPerBlock + wave_id(); r < n; r += nwaves) {
    const Index a = ptr[r], b = ptr[r + 1];
    for (Index e = a + lane; e < b; e += kWave) erow[e] = (int)r;
  }
}

// degree in the symmetric graph = row length + column length of the triangle; the sort key puts high degrees first
__global__ __launch_bounds__(kBlock) void tc_degree_kernel(const Index* __restrict__ rptr, const Index* __restrict__ cptr, Index n,
                                                           unsigned long long* __restrict__ key, unsigned* __restrict__ pay) {
  const Index v = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (v >= n) return;
  const unsigned deg = (unsigned)(rptr[v + 1] - rptr[v]) + (unsigned)(cptr[v + 1] - cptr[v]);
  key[v] = (unsigned long long)(0xffffffffu - deg);
  pay[v] = (unsigned)v;
}

// The number of the vertex of rank r: r, but nobody is numbered 65 535 (the 16-bit parts' filling).
// (pbase: the degree of the vertex of a number -- room for its partners, whoever they turn out to be; scanned afterwards)
__global__ __launch_bounds__(kBlock) void tc_number_kernel(const unsigned* __restrict__ order, const unsigned long long* __restrict__ key,
                                                           Index n, int* __restrict__ number, unsigned* __restrict__ pbase) {
  const Index r = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (r >= n) return;
  const int num = (int)r + (r >= 65535 ? 1 : 0);
  number[order[r]] = num;
  pbase[num] = 0xffffffffu - (unsigned)key[r];
}

// pass A: every entry (i, j) as {lower-ranked end, higher-ranked end} in the new numbers; the lower end's list grows by one
// -- its 16-bit part or its 32-bit part.
// bad: bit 0 an entry on the diagonal or a value that is not 1, bits 1 / 2 entries below / above the diagonal (both: the sum
// of the product is not a count)
__global__ __launch_bounds__(kBlock) void tc_orient_kernel(const int* __restrict__ erow, const Index* __restrict__ ind,
                                                           const unsigned* __restrict__ val, unsigned one, long long nnz,
                                                           const int* __restrict__ number, int* __restrict__ elo, int* __restrict__ ehi,
                                                           unsigned* __restrict__ c16, unsigned* __restrict__ c32, int* __restrict__ bad) {
  const long long stride = (long long)gridDim.x * kBlock;
  bool wrong = false, below = false, above = false;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += stride) {
    const int i = erow[e], j = (int)ind[e];
    wrong |= j == i || val[e] != one;
    below |= j < i;
    above |= j > i;
    const int a = number[i], b = number[j];
    const int lo = a > b ? a : b, hi = a > b ? b : a;
    elo[e] = lo; ehi[e] = hi;
    atomicAdd(hi < 65535 ? &c16[lo] : &c32[lo], 1u);
  }
  // (a strictly UPPER triangle of ones sums to the same count: every edge once, either way round; entries on both sides do not)
  const int f = (__any(wrong) ? 1 : 0) | (__any(below) ? 2 : 0) | (__any(above) ? 4 : 0);
  if (lane_id() == 0) atomicOr(bad, f);
}

// the two parts' lengths of every list, and their rooms: the next multiple of 16 bytes (before the scans that make the
// rooms positions)
__global__ __launch_bounds__(kBlock) void tc_lengths_kernel(unsigned* __restrict__ c16, unsigned* __restrict__ c32, Index nn,
                                                            int* __restrict__ len16, int* __restrict__ len32, int* __restrict__ len,
                                                            int* __restrict__ longest) {
  const Index v = (Index)blockIdx.x * kBlock + threadIdx.x;
  int l = 0;
  if (v < nn) {
    const int a = (int)c16[v], b = (int)c32[v];
    len16[v] = a; len32[v] = b; l = a + b; len[v] = l;
    c16[v] = (unsigned)(a + 7) & ~7u;
    c32[v] = (unsigned)(b + 3) & ~3u;
  }
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) { const int o = __shfl_down(l, off); l = o > l ? o : l; }
  if (lane_id() == 0 && l > 0) atomicMax(longest, l);
}

__global__ __launch_bounds__(kBlock) void tc_describe_kernel(const unsigned* __restrict__ ptr16, const unsigned* __restrict__ ptr32,
                                                             const int* __restrict__ len16, const int* __restrict__ len32, Index nn,
                                                             int4* __restrict__ lists) {
  const Index v = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (v < nn) lists[v] = make_int4((int)ptr16[v], len16[v], (int)ptr32[v], len32[v]);
}

// who intersects an edge: the end with the longer list streams nothing, it is the PIVOT (the lower-ranked end on a tie);
// the other end's list is streamed past it.  A partner with an empty list is dropped.
__device__ __forceinline__ void tc_roles(int lo, int hi, const int* __restrict__ len, int* pivot, int* partner, int* plen) {
  const int llo = len[lo], lhi = len[hi];
  const bool lo_is_pivot = lhi <= llo;
  *pivot = lo_is_pivot ? lo : hi;
  *partner = lo_is_pivot ? hi : lo;
  *plen = lo_is_pivot ? lhi : llo;
}

// pass B: the lists themselves (in whatever order the atomics hand out: they are looked up, never merged) and every
// pivot's partners -- their lists' descriptors -- in the room its degree reserves (cur16 / cur32 / pcur start as copies of
// the position arrays: the atomics hand out absolute positions)
__global__ __launch_bounds__(kBlock) void tc_lists_kernel(const int* __restrict__ elo, const int* __restrict__ ehi, long long nnz,
                                                          const int4* __restrict__ lists, const int* __restrict__ len,
                                                          unsigned* __restrict__ cur16, unsigned* __restrict__ cur32,
                                                          unsigned short* __restrict__ D16, int* __restrict__ D32,
                                                          unsigned* __restrict__ pcur, int4* __restrict__ P) {
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += stride) {
    const int lo = elo[e], hi = ehi[e];
    if (hi < 65535) D16[atomicAdd(&cur16[lo], 1u)] = (unsigned short)hi;
    else D32[atomicAdd(&cur32[lo], 1u)] = hi;
    int pivot, partner, plen;
    tc_roles(lo, hi, len, &pivot, &partner, &plen);
    if (plen > 0) P[atomicAdd(&pcur[pivot], 1u)] = lists[partner];
  }
}

// which kernel a pivot goes to (-1: nothing to do), and in how many tasks
__device__ __forceinline__ int tc_class(int p, int len, unsigned np, int bitmap_upto) {
  if (np == 0u) return -1;
  if (len <= kTcWaveLen) return 0;
  return p <= bitmap_upto ? 1 : 2;
}
__global__ __launch_bounds__(kBlock) void tc_task_count_kernel(const int* __restrict__ len, const unsigned* __restrict__ pbase,
                                                               const unsigned* __restrict__ pcur, Index n, int bitmap_upto,
                                                               unsigned* __restrict__ c0, unsigned* __restrict__ c1, unsigned* __restrict__ c2) {
  const Index p = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const unsigned np = pcur[p] - pbase[p];
  const int c = tc_class((int)p, len[p], np, bitmap_upto);
  c0[p] = c == 0 ? (np + kTcChunk[0] - 1) / kTcChunk[0] : 0u;
  c1[p] = c == 1 ? (np + kTcChunk[1] - 1) / kTcChunk[1] : 0u;
  c2[p] = c == 2 ? (np + kTcChunk[2] - 1) / kTcChunk[2] : 0u;
}
__global__ __launch_bounds__(kBlock) void tc_task_fill_kernel(const int* __restrict__ len, const unsigned* __restrict__ pbase,
                                                              const unsigned* __restrict__ pcur, Index n, int bitmap_upto,
                                                              const unsigned* __restrict__ c0, const unsigned* __restrict__ c1,
                                                              const unsigned* __restrict__ c2, int4* __restrict__ t0, int4* __restrict__ t1,
                                                              int4* __restrict__ t2) {
  const Index p = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const unsigned first = pbase[p], np = pcur[p] - first;
  const int c = tc_class((int)p, len[p], np, bitmap_upto);
  if (c < 0) return;
  int4* const t = c == 0 ? t0 : c == 1 ? t1 : t2;
  const unsigned at = c == 0 ? c0[p] : c == 1 ? c1[p] : c2[p];
  const unsigned chunk = (unsigned)kTcChunk[c];
  for (unsigned k = 0; k * chunk < np; ++k)
    t[at + k] = make_int4((int)p, (int)(first + k * chunk), (int)(np - k * chunk < chunk ? np - k * chunk : chunk), 0);
}

__global__ __launch_bounds__(kTcSlots) void tc_total_kernel(unsigned long long* slots) {
  __shared__ unsigned long long part[kTcSlots / kWave];
  unsigned long long v = slots[threadIdx.x];
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) v += __shfl_down(v, off);
  if (lane_id() == 0) part[wave_id()] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
    for (int w = 0; w < kTcSlots / kWave; ++w) s += part[w];
    slots[kTcSlots] = s;
  }
}

// sum of the squared row lengths of the triangle as the caller numbered it: what the PRODUCT's intersections cost at most
__global__ __launch_bounds__(kBlock) void tc_row_squares_kernel(const Index* __restrict__ ptr, Index n, unsigned long long* __restrict__ out) {
  const Index v = (Index)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long q = 0;
  if (v < n) { const unsigned long long l = (unsigned long long)(ptr[v + 1] - ptr[v]); q = l * l; }
  q = wave_sum_u64(q);
  if (lane_id() == 0 && q) atomicAdd(out, q);
}

struct TcPrep {
  int state = 0;                              // 0 not tried, 1 ready, -1 the sum of this matrix's product is not such a count
  int calls = 0;                              // counts asked of this matrix so far
  unsigned short* D16 = nullptr;              // the lists' numbers below 65 535, part by part (+ 16 entries: a step reads whole 16-byte groups)
  int* D32 = nullptr;                         // ... and the others (+ 8)
  int4* lists = nullptr;                      // [n + 1] by number: {first entry in D16, entries, first entry in D32, entries}
  int4* P = nullptr;                          // the partners' lists (copies of their descriptors), pivot by pivot
  int4* tasks[3] = {nullptr, nullptr, nullptr};
  int ntasks[3] = {0, 0, 0};
  int longest = 0;
  float prep_ms = 0.f;
};
static void tc_prep_release(TcPrep* t) {
  for (void** q : {(void**)&t->D16, (void**)&t->D32, (void**)&t->lists, (void**)&t->P, (void**)&t->tasks[0], (void**)&t->tasks[1], (void**)&t->tasks[2]})
    if (*q) { (void)hipFree(*q); *q = nullptr; }
}

static int g_tc_product = -1;                 // grb_tc_set_product
static struct { int path; float prep_ms, count_ms; int longest; int ntasks[3]; } g_tc_last = {0, 0.f, 0.f, 0, {0, 0, 0}};

void tc_prep_free(grb_matrix_s* A) {
  TcPrep* t = (TcPrep*)A->tc_prep;
  if (!t) return;
  tc_prep_release(t);
  delete t;
  A->tc_prep = nullptr;
}

// temporaries of the preparation: freed when the scope ends, whatever the way out
struct TcTemps {
  std::vector<void*> v;
  ~TcTemps() { for (void* q : v) (void)hipFree(q); }
  template <typename T> grb_info get(T** out, size_t count) {
    void* q = nullptr;
    GRB_HIP_TRY(hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    v.push_back(q);
    *out = (T*)q;
    return GRB_SUCCESS;
  }
};

static grb_info tc_prepare(grb_matrix_s* A, TcPrep* t) {
  hipStream_t s = ctx().stream;
  const Index n = A->nrows;
  const Index nn = n + 1;                     // numbers: 0 .. n, 65 535 left out
  const long long nnz = A->nvals;
  t->state = -1;
  // (positions are 32-bit: the entries, two descriptors an edge, the parts' rooms with up to seven fillings a vertex)
  if (A->csc_alias || !A->csc.ptr || !A->csr.val || n != A->ncols || nnz < 1 || nnz > 0x7ffffff0ll || n >= 0x7ffffff0 ||
      nnz + 8ll * ((long long)n + 1) >= 0xfffffff0ll)
    return GRB_SUCCESS;
  // GRB_TC_TRACE=1: the host's clock after every stage (each with a stream wait), to stderr
  static const bool trace = [] { const char* e = getenv("GRB_TC_TRACE"); return e && atoi(e) != 0; }();
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = trace ? now_us() : 0.0;
  auto stage = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    const double t = now_us();
    fprintf(stderr, "tc_prepare: %-28s %9.1f us\n", what, t - t_last);
    t_last = t;
  };
  TcTemps tmp;
  int *erow, *number, *elo, *ehi, *len, *len16, *len32, *flags;
  unsigned long long* key;
  unsigned *order, *ptr16, *ptr32, *cur16, *cur32, *pptr, *pcur, *c0, *c1, *c2;
  GRB_TRY(tmp.get(&erow, (size_t)nnz));
  GRB_TRY(tmp.get(&elo, (size_t)nnz));
  GRB_TRY(tmp.get(&ehi, (size_t)nnz));
  GRB_TRY(tmp.get(&number, (size_t)n));
  GRB_TRY(tmp.get(&len, (size_t)nn));
  GRB_TRY(tmp.get(&len16, (size_t)nn));
  GRB_TRY(tmp.get(&len32, (size_t)nn));
  GRB_TRY(tmp.get(&key, (size_t)n));
  GRB_TRY(tmp.get(&order, (size_t)n));
  GRB_TRY(tmp.get(&ptr16, (size_t)nn + 1));
  GRB_TRY(tmp.get(&ptr32, (size_t)nn + 1));
  GRB_TRY(tmp.get(&cur16, (size_t)nn));
  GRB_TRY(tmp.get(&cur32, (size_t)nn));
  GRB_TRY(tmp.get(&pptr, (size_t)nn + 1));
  GRB_TRY(tmp.get(&pcur, (size_t)nn));
  GRB_TRY(tmp.get(&c0, (size_t)nn + 1));
  GRB_TRY(tmp.get(&c1, (size_t)nn + 1));
  GRB_TRY(tmp.get(&c2, (size_t)nn + 1));
  GRB_TRY(tmp.get(&flags, 2));                // {bad, longest list}
  GRB_HIP_TRY(hipMalloc((void**)&t->lists, 16 * (size_t)nn));
  stage("temporaries allocated");
  const int vgrid = (int)((n + kBlock - 1) / kBlock), ngrid = (int)((nn + kBlock - 1) / kBlock), egrid = stream_grid(nnz, kBlock * 4);
  // the numbering: by degree, the highest first, ties by the caller's number (the sort is stable)
  hipLaunchKernelGGL(tc_degree_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const Index*)A->csr.ptr, (const Index*)A->csc.ptr, n, key, order);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_sort_pairs(key, order, n, 32, 0));
  stage("degrees sorted");
  GRB_HIP_TRY(hipMemsetAsync(pptr, 0, 4 * ((size_t)nn + 1), s));
  hipLaunchKernelGGL(tc_number_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const unsigned*)order, (const unsigned long long*)key, n, number, pptr);
  hipLaunchKernelGGL(tc_rows_kernel, dim3(stream_grid((long long)n * 16, kBlock)), dim3(kBlock), 0, s, (const Index*)A->csr.ptr, n, erow);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipMemsetAsync(ptr16, 0, 4 * ((size_t)nn + 1), s));
  GRB_HIP_TRY(hipMemsetAsync(ptr32, 0, 4 * ((size_t)nn + 1), s));
  GRB_HIP_TRY(hipMemsetAsync(flags, 0, 8, s));
  const unsigned one = A->dtype == GRB_F32 ? 0x3f800000u : 1u;
  hipLaunchKernelGGL(tc_orient_kernel, dim3(egrid), dim3(kBlock), 0, s, (const int*)erow, (const Index*)A->csr.ind,
                     (const unsigned*)A->csr.val, one, nnz, (const int*)number, elo, ehi, ptr16, ptr32, flags);
  hipLaunchKernelGGL(tc_lengths_kernel, dim3(ngrid), dim3(kBlock), 0, s, ptr16, ptr32, nn, len16, len32, len, flags + 1);
  GRB_HIP_TRY(hipGetLastError());
  stage("rows, pass A, lengths");
  GRB_TRY(device_exclusive_scan_u32(ptr16, (long long)nn + 1));
  GRB_TRY(device_exclusive_scan_u32(ptr32, (long long)nn + 1));
  GRB_TRY(device_exclusive_scan_u32(pptr, (long long)nn + 1));
  hipLaunchKernelGGL(tc_describe_kernel, dim3(ngrid), dim3(kBlock), 0, s, (const unsigned*)ptr16, (const unsigned*)ptr32, (const int*)len16,
                     (const int*)len32, nn, t->lists);
  GRB_HIP_TRY(hipGetLastError());
  int h_flags[2] = {0, 0};
  unsigned room16 = 0, room32 = 0;
  GRB_HIP_TRY(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(&room16, ptr16 + nn, 4, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(&room32, ptr32 + nn, 4, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  stage("scans");
  if ((h_flags[0] & 1) || (h_flags[0] & 6) == 6) return GRB_SUCCESS;   // not a strict triangle of ones
  t->longest = h_flags[1];
  GRB_HIP_TRY(hipMalloc((void**)&t->D16, 2 * ((size_t)room16 + 16)));
  GRB_HIP_TRY(hipMalloc((void**)&t->D32, 4 * ((size_t)room32 + 8)));
  GRB_HIP_TRY(hipMalloc((void**)&t->P, 16 * (2 * (size_t)nnz + 1)));         // (a vertex's room: its degree)
  stage("lists and partners allocated");
  GRB_HIP_TRY(hipMemsetAsync(t->D16, 0xff, 2 * ((size_t)room16 + 16), s));
  GRB_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)t->D32, (int)kTcPad, (size_t)room32 + 8, s));
  GRB_HIP_TRY(hipMemcpyAsync(cur16, ptr16, 4 * (size_t)nn, hipMemcpyDeviceToDevice, s));
  GRB_HIP_TRY(hipMemcpyAsync(cur32, ptr32, 4 * (size_t)nn, hipMemcpyDeviceToDevice, s));
  GRB_HIP_TRY(hipMemcpyAsync(pcur, pptr, 4 * (size_t)nn, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(tc_lists_kernel, dim3(egrid), dim3(kBlock), 0, s, (const int*)elo, (const int*)ehi, nnz, (const int4*)t->lists,
                     (const int*)len, cur16, cur32, t->D16, t->D32, pcur, t->P);
  GRB_HIP_TRY(hipGetLastError());
  stage("pass B");
  // the tasks (GRB_TC_BITMAP_UPTO: tests send the pivots beyond a smaller number to the hash-table kernel)
  int bitmap_upto = kTcBits;
  if (const char* e = getenv("GRB_TC_BITMAP_UPTO")) { const int v = atoi(e); if (v >= 0 && v < kTcBits) bitmap_upto = v; }
  for (unsigned* c : {c0, c1, c2}) GRB_HIP_TRY(hipMemsetAsync(c + nn, 0, 4, s));
  hipLaunchKernelGGL(tc_task_count_kernel, dim3(ngrid), dim3(kBlock), 0, s, (const int*)len, (const unsigned*)pptr, (const unsigned*)pcur, nn, bitmap_upto, c0, c1, c2);
  GRB_HIP_TRY(hipGetLastError());
  unsigned nt[3] = {0, 0, 0};
  int k = 0;
  for (unsigned* c : {c0, c1, c2}) {
    GRB_TRY(device_exclusive_scan_u32(c, (long long)nn + 1));
    GRB_HIP_TRY(hipMemcpyAsync(&nt[k++], c + nn, 4, hipMemcpyDeviceToHost, s));
  }
  GRB_HIP_TRY(hipStreamSynchronize(s));
  if (nt[2] > 0 && t->longest > kTcHashLen) return GRB_SUCCESS;     // a list no table here holds
  for (k = 0; k < 3; ++k) {
    t->ntasks[k] = (int)nt[k];
    GRB_HIP_TRY(hipMalloc((void**)&t->tasks[k], 16 * ((size_t)nt[k] + 1)));
  }
  hipLaunchKernelGGL(tc_task_fill_kernel, dim3(ngrid), dim3(kBlock), 0, s, (const int*)len, (const unsigned*)pptr, (const unsigned*)pcur, nn, bitmap_upto, (const unsigned*)c0,
                     (const unsigned*)c1, (const unsigned*)c2, t->tasks[0], t->tasks[1], t->tasks[2]);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipStreamSynchronize(s));
  stage("tasks");
  t->state = 1;
  return GRB_SUCCESS;
}

// 0: the count where it pays (below), 1: always the product, 2: the count wherever it is a count
int tc_product_setting(int set) {
  if (g_tc_product < 0) { const char* e = getenv("GRB_TC_PRODUCT"); const int v = e ? atoi(e) : 0; g_tc_product = v >= 0 && v <= 2 ? v : 0; }
  const int prev = g_tc_product;
  if (set >= 0) g_tc_product = set <= 2 ? set : 1;
  return prev;
}

// A matrix WITHOUT long rows (a road network, a uniform random graph) is cheap for the product -- 9 ms for 40 M entries whose
// rows hold 44 entries at most -- and the orientation's preparation is not (45 ms there): the first count asked of such a
// matrix goes through the product, the second prepares the orientation (2.7 ms per count from then on).  "Without long
// rows": the squared row lengths sum to at most kTcLight per entry (RMAT-22 ef 28: 4 000; the uniform graph: 11; a grid: 3).
constexpr unsigned long long kTcLight = 256;
static grb_info tc_is_light(grb_matrix_s* A, bool* light) {
  hipStream_t s = ctx().stream;
  void* p;
  GRB_TRY(scratch(10, 8 * (kTcSlots + 1), &p));
  GRB_HIP_TRY(hipMemsetAsync(p, 0, 8, s));
  hipLaunchKernelGGL(tc_row_squares_kernel, dim3((A->nrows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, (const Index*)A->csr.ptr, A->nrows,
                     (unsigned long long*)p);
  GRB_HIP_TRY(hipGetLastError());
  unsigned long long sq = 0;
  GRB_HIP_TRY(hipMemcpyAsync(&sq, p, 8, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  *light = sq <= kTcLight * (unsigned long long)(A->nvals > 0 ? A->nvals : 1);
  return GRB_SUCCESS;
}

// *done = true: *count is the sum of L (+.x) L^T over the entries of L.  false: A is not a strictly lower triangle of ones
// (or holds a list no table here takes) -- the caller forms the product.
grb_info tc_count_try(grb_matrix_s* A, long long* count, bool* done) {
  *done = false;
  g_tc_last.path = 0;
  const int mode = tc_product_setting(-1);
  if (mode == 1) return GRB_SUCCESS;
  hipStream_t s = ctx().stream;
  hipEvent_t ev[3];
  for (auto& e : ev) GRB_HIP_TRY(hipEventCreate(&e));
  struct EvFree { hipEvent_t* e; ~EvFree() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } ev_free{ev};
  GRB_HIP_TRY(hipEventRecord(ev[0], s));
  TcPrep* t = (TcPrep*)A->tc_prep;
  float prep_ms = 0.f;
  if (!t) {
    t = new TcPrep();
    A->tc_prep = t;
  }
  if (t->state == 0) {
    if (mode == 0 && t->calls++ == 0 && A->nrows > 0 && A->csr.ptr) {
      bool light = false;
      GRB_TRY(tc_is_light(A, &light));
      if (light) return GRB_SUCCESS;          // the product this once
    }
    const grb_info info = tc_prepare(A, t);
    if (info != GRB_SUCCESS || t->state != 1) {
      tc_prep_release(t);                     // (whatever was allocated goes; the verdict stays: the next call does not try again)
      if (info != GRB_SUCCESS) { delete t; A->tc_prep = nullptr; return info; }
    }
  }
  if (t->state != 1) return GRB_SUCCESS;
  GRB_HIP_TRY(hipEventRecord(ev[1], s));
  void* p_slots;
  GRB_TRY(scratch(10, 8 * (kTcSlots + 1), &p_slots));
  unsigned long long* slots = (unsigned long long*)p_slots;
  GRB_HIP_TRY(hipMemsetAsync(slots, 0, 8 * (kTcSlots + 1), s));
  // (the short pivots' kernel on a second stream beside the long pivots' was measured: the same 23.5 ms, docs/experiments.md R6.8)
  // GRB_TC_SHORT: partners of the bitmap kernel with at most this many 16-bit entries (half as many 32-bit ones) are walked by a lane
  static const int short_upto = [] { const char* e = getenv("GRB_TC_SHORT"); return e ? atoi(e) : 128; }();
  if (t->ntasks[1] > 0)
    hipLaunchKernelGGL((tc_count_bitmap_kernel<512, kTcBits>), dim3(t->ntasks[1]), dim3(512), 0, s, (const unsigned short*)t->D16, (const int*)t->D32,
                       (const int4*)t->lists, (const int4*)t->P, (const int4*)t->tasks[1], short_upto, slots);
  if (t->ntasks[2] > 0)
    hipLaunchKernelGGL((tc_count_pivot_kernel<512, 2 * kTcHashLen>), dim3(t->ntasks[2]), dim3(512), 0, s, (const unsigned short*)t->D16, (const int*)t->D32,
                       (const int4*)t->lists, (const int4*)t->P, (const int4*)t->tasks[2], t->ntasks[2], slots);
  if (t->ntasks[0] > 0) {
    // GRB_TC_SMALL=0: the wave-per-partner kernel for the short pivots too (the A/B of docs/experiments.md R6.8)
    static const bool small = [] { const char* e = getenv("GRB_TC_SMALL"); return !e || atoi(e) != 0; }();
    if (small)
      hipLaunchKernelGGL(tc_count_small_kernel, dim3(t->ntasks[0]), dim3(kWave), 0, s, (const unsigned short*)t->D16, (const int*)t->D32,
                         (const int4*)t->lists, (const int4*)t->P, (const int4*)t->tasks[0], slots);
    else
      hipLaunchKernelGGL((tc_count_pivot_kernel<64, 2 * kTcWaveLen>), dim3(t->ntasks[0]), dim3(64), 0, s,
                         (const unsigned short*)t->D16, (const int*)t->D32, (const int4*)t->lists, (const int4*)t->P, (const int4*)t->tasks[0],
                         t->ntasks[0], slots);
  }
  hipLaunchKernelGGL(tc_total_kernel, dim3(1), dim3(kTcSlots), 0, s, slots);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipEventRecord(ev[2], s));
  unsigned long long tot = 0;
  GRB_HIP_TRY(hipMemcpyAsync(&tot, slots + kTcSlots, 8, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  GRB_HIP_TRY(hipEventElapsedTime(&prep_ms, ev[0], ev[1]));
  GRB_HIP_TRY(hipEventElapsedTime(&g_tc_last.count_ms, ev[1], ev[2]));
  if (t->prep_ms == 0.f) t->prep_ms = prep_ms;
  g_tc_last.path = 1;
  g_tc_last.prep_ms = prep_ms;
  g_tc_last.longest = t->longest;
  for (int k = 0; k < 3; ++k) g_tc_last.ntasks[k] = t->ntasks[k];
  *count = (long long)tot;
  *done = true;
  return GRB_SUCCESS;
}

}  // namespace grb

using namespace grb;


extern "C" int grb_tc_set_product(int on) { GRB_API_ENTER_HOST(); return tc_product_setting(on); }

extern "C" grb_info grb_tc_release(grb_matrix A) { GRB_API_ENTER();
  if (!A) return GRB_NULL_POINTER;
  GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
  tc_prep_free(A);
  return GRB_SUCCESS;
}

extern "C" grb_info grb_tc_last(grb_tc_info* out) { GRB_API_ENTER_HOST();
  if (!out) return GRB_NULL_POINTER;
  out->path = g_tc_last.path;
  out->prep_ms = g_tc_last.path ? g_tc_last.prep_ms : 0.f;
  out->count_ms = g_tc_last.path ? g_tc_last.count_ms : 0.f;
  out->longest_list = g_tc_last.path ? g_tc_last.longest : 0;
  for (int k = 0; k < 3; ++k) out->tasks[k] = g_tc_last.path ? g_tc_last.ntasks[k] : 0;
  return GRB_SUCCESS;
}
