// bfs_batch.hip -- up to 64 traversals at once: the multi-frontier form of algorithm::bfs.
//
// In GraphBLAS terms a level of k simultaneous traversals is  F' = (A^T lor.land F) .* not(Seen)
// with F, Seen : n x k Boolean -- the sparse x dense product the reference leaves as a stub
// (backend/cuda/operations.hpp:52-70, spmm.hpp:15-27).  With k <= 64 a row of F is ONE 64-bit
// word (bit s = "in the frontier of source s"), the semiring's add is a word-wide OR, its
// multiply the AND with the edge's presence: one 8-byte gather per edge serves 64 traversals.
// No tile of A is dense enough for a matrix core to beat that (the dense-core SpMM path that
// tried was measured slower and removed in round 5).
//
//   seen[v], W_L[v]  one 64-bit word per vertex; W_L = the bits discovered by level L, kept for
//                    the first kStore levels (the frontier of level L + 1 IS W_L)
//   direction        chosen PER SOURCE every level from that source's frontier size and out-degree
//                    sum: bits of small frontiers are pushed, bits of large ones pulled, in the same
//                    level.  (A union rule makes every vertex scan its whole list for the 63 sources
//                    that are still near their start while one hub source already floods the graph.)
//   pull (bits Q)    a lane per vertex: need = ~seen & Q; the hinted in-neighbour first, four serial
//                    probes, then the wave finishes its open rows together in rounds (each row's next 32,
//                    128, 512 ... entries laid end to end and dealt evenly to the lanes, ORed per row in
//                    LDS), a row stopping once every needed bit is found.  Rows of >= 4096 entries are
//                    cut into 4096-entry slices taken by separate waves
//   push (bits P)    frontier words with P bits expand along out-edges, the edges of a wave's 64 vertices dealt
//                    evenly to its lanes: an atomicOr into seen claims the unseen bits, a second into W_L
//                    records them (the pull pass has just written every word of W_L; zeroed when nothing is
//                    pulled); a streaming pass then counts the pushed bits of W_L per source
//   labels           not written while traversing: one final pass turns the stored level words into
//                    the k depth vectors (label[s][v] = level of discovery, source = 1, unreached =
//                    0) with full-width coalesced stores -- no memset, no scattered 4-byte stores.
//                    Levels beyond kStore (long-diameter graphs, tiny frontiers) label directly.
//                    The vectors are what k calls of algorithm::bfs return: BFS depth is unique.
#include "bfs_kernels.hpp"
#include "persist_common.hpp"
#include <chrono>

namespace grb {

constexpr int kBatchBig = 4096;       // row length from which a row is cut into slices (GRB_BATCH_BIG_IN / _OUT)
constexpr int kBatchBigPush = 512;    // the same for out-edge rows: a pushed edge costs more than a pulled one
constexpr int kBatchSlice = 4096;     // pull: entries per slice, at most
constexpr int kBatchPushSlice = 1024; // push: out-edge slices (every edge is a chain of dependent memory steps)
constexpr int kBatchSerial = 4;       // serial probes per lane before the wave takes over
constexpr int kBatchSlots = 16;       // counter slots (spreads same-address atomics)
constexpr int kBatchStoreMax = 16;    // level words kept for the final label pass
constexpr int kBatchCounters = 2 + 128;   // per slot: vertices, out-degree sum, then {nf_s, mf_s} per source

typedef unsigned long long u64;

struct BatchArgs {
  const Index *optr, *oind, *iptr, *iind;
  const Index* hint;
  Index n;
  u64 qmask, pmask;                   // bits pulled / pushed this level
  u64* seen;
  const u64* fcur;
  u64* fnext;
  int big;                            // rows of this many entries or more belong to the slice kernels
  const u64* prev;                    // heavy push levels: seen as it stood before the push kernels (else null)
  const int4* slices;                 // {vertex, first entry, end entry, big index}
  int nslices;
  const Index* bigrows;               // the big rows' vertex ids
  int nbig;
  u64* counters;                      // [kBatchSlots][kBatchCounters]
  float new_label;
  int direct_labels;                  // levels beyond the stored ones write labels as they discover
  int k;
  float* label[64];
};

struct LabelArgs {
  Index n;
  int k, nstored;
  const u64* seen;
  const u64* W[kBatchStoreMax + 1];   // the stored levels' words ...
  float lab[kBatchStoreMax + 1];      // ... and the depth each of them assigns (0: the level the iteration cap cuts off)
  float* label[64];
};

__device__ inline u64 wave_or(u64 x) { return wave_or_u64(x); }   // DPP (common.hpp): twelve LDS-crossbar permutes otherwise

struct BatchTotals {                  // per workgroup, in LDS
  u64 v[kBatchCounters];
};

#ifndef GRB_BATCH_TRANSPOSE_FROM
#define GRB_BATCH_TRANSPOSE_FROM 8
#endif
constexpr int kTransposeFrom = GRB_BATCH_TRANSPOSE_FROM;     // live sources in a wave from which the bit matrix is transposed by exchanges

// A wave's running totals, source s in lane s: nf = (vertex, source) pairs discovered, mf = their out-degrees
struct WaveTotals {
  u64 nf = 0, mf = 0;
  unsigned int verts = 0;             // wave-uniform: vertices with any new bit
};

// accounting (and, beyond the stored levels, labels) of a lane's new bits; wave-collective.  The 64 x 64 bit
// matrix (lane = vertex, bit = source) is transposed with one ballot per live source, so lane s holds the
// mask m of vertices new to source s: nf_s += popcount(m), and the degree sum comes from the degrees' bit
// planes, mf_s += sum_b 2^b popcount(m & plane_b) -- all 64 sources in parallel, no per-source reduction.
__device__ inline void batch_commit_l(const BatchArgs& a, WaveTotals& acc, Index v, u64 newb, bool direct, float label) {
  const unsigned long long mv = __ballot(newb != 0);
  if (!mv) return;
  const int lane = lane_id();
  acc.verts += (unsigned int)__popcll(mv);
  unsigned int deg = 0;
  if (newb) deg = (unsigned int)(a.optr[v + 1] - a.optr[v]);
  u64 m = 0;
  const u64 live = wave_or(newb);
  if (__popcll(live) <= kTransposeFrom) {
    for (u64 t = live; t; t &= t - 1) {
      const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)t) - 1);   // wave-uniform: a scalar index
      const unsigned long long col = __ballot((newb >> s) & 1ull);
      if (lane == s) m = col;
    }
  } else {
    // many live sources: the 64 x 64 bit matrix is transposed in six exchange steps (blocks of 32, 16, ... 1 swapped
    // with the lane `k` away) instead of one ballot per source -- ~70 instructions against ~10 per source
    m = newb;
    auto exchange = [&](int k, u64 keep) {
      const u64 y = __shfl_xor(m, k, kWave);
      m = (lane & k) ? ((m & ~keep) | ((y >> k) & keep)) : ((m & keep) | ((y << k) & ~keep));
    };
    exchange(32, 0x00000000ffffffffull);
    exchange(16, 0x0000ffff0000ffffull);
    exchange(8, 0x00ff00ff00ff00ffull);
    exchange(4, 0x0f0f0f0f0f0f0f0full);
    exchange(2, 0x3333333333333333ull);
    exchange(1, 0x5555555555555555ull);
  }
  acc.nf += (u64)__popcll(m);
  unsigned int dor = deg;
  dor = wave_or_u32(dor);
  for (unsigned int t = dor; t; t &= t - 1) {
    const int b = __builtin_amdgcn_readfirstlane(__ffs((int)t) - 1);
    const unsigned long long plane = __ballot((deg >> b) & 1u);
    acc.mf += (u64)__popcll(m & plane) << b;
  }
  if (direct)
    for (u64 t = newb; t; t &= t - 1) a.label[__ffsll((long long)t) - 1][v] = label;
}
__device__ inline void batch_commit(const BatchArgs& a, WaveTotals& acc, Index v, u64 newb) {
  batch_commit_l(a, acc, v, newb, a.direct_labels != 0, a.new_label);
}

__device__ inline void totals_init(BatchTotals* lds) {
  for (int i = threadIdx.x; i < kBatchCounters; i += blockDim.x) lds->v[i] = 0;
  __syncthreads();
}
__device__ inline void totals_flush_to(u64* slots, BatchTotals* lds, const WaveTotals& acc) {
  const int lane = lane_id();
  if (acc.nf) atomicAdd(&lds->v[2 + 2 * lane], acc.nf);
  if (acc.mf) atomicAdd(&lds->v[3 + 2 * lane], acc.mf);
  if (lane == 0 && acc.verts) atomicAdd(&lds->v[0], (u64)acc.verts);
  __syncthreads();
  u64* dst = slots + (size_t)(blockIdx.x & (kBatchSlots - 1)) * kBatchCounters;
  for (int i = threadIdx.x; i < kBatchCounters; i += blockDim.x)
    if (lds->v[i]) atomicAdd(&dst[i], lds->v[i]);
}
__device__ inline void totals_flush(const BatchArgs& a, BatchTotals* lds, const WaveTotals& acc) {
  totals_flush_to(a.counters, lds, acc);
}

// the wave scans entries [rs, re) of `ind`, ORs word[ind[q]] and stops once `nd` is covered
__device__ inline u64 wave_scan_or(const Index* __restrict__ ind, const u64* __restrict__ word, Index rs, Index re,
                                   u64 nd, int lane) {
  u64 got = 0;
  for (Index q = rs; q < re; q += 4 * kWave) {
    u64 w = 0;
    Index c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const Index at = q + j * kWave + lane;
      c[j] = at < re ? ind[at] : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w |= c[j] >= 0 ? word[c[j]] : 0ull;
    got |= wave_or(w);
    if ((got & nd) == nd) break;
  }
  return got;
}

__global__ __launch_bounds__(kBlock) void batch_seed_kernel(u64* seen, u64* w0, const Index* __restrict__ sources, int k) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < k) {
    const Index v = sources[s];
    atomicOr(&seen[v], 1ull << s);
    atomicOr(&w0[v], 1ull << s);
  }
}

__global__ __launch_bounds__(kBlock) void batch_pull_kernel(BatchArgs a) {
  __shared__ BatchTotals lds;
  __shared__ Index s_pre[kWavesPerBlock][kWave];
  __shared__ Index s_p[kWavesPerBlock][kWave];
  __shared__ u64 s_acc[kWavesPerBlock][kWave];
  totals_init(&lds);
  WaveTotals tot;
  const int lane = lane_id(), w = wave_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const bool valid = v < a.n;
    const u64 seen = valid ? a.seen[v] : ~0ull;
    u64 need = ~seen & a.qmask;
    Index p = 0, e = 0;
    if (need) { p = a.iptr[v]; e = a.iptr[v + 1]; }
    const bool big = e - p >= a.big;                       // probed here, finished by the slice kernels
    if (p == e) need = 0;
    if (__ballot(need != 0) == 0ull) {
      if (valid) a.fnext[v] = 0ull;
      continue;
    }
    u64 acc = 0;
    if (need && a.hint) acc = a.fcur[a.hint[v]];
#pragma unroll
    for (int t = 0; t < kBatchSerial; ++t) {
      const bool go = need && (acc & need) != need && p < e;
      const Index c = a.iind[go ? p : 0];
      const u64 w = a.fcur[go ? c : 0];
      if (go) { acc |= w; ++p; }
    }
    // the rows still open finish together: round by round each takes its next `quota` entries (32, 128, 512, ...),
    // the entries of all of them are laid end to end and dealt to the lanes 256 at a time, the gathered
    // words are ORed per row in LDS -- full lanes whatever the row lengths, an exit test per row per round
    Index quota = 32;
    for (;;) {
      const bool open = !big && need && (acc & need) != need && p < e;
      if (__ballot(open) == 0ull) break;
      const Index cnt = open ? (e - p < quota ? e - p : quota) : 0;
      Index inc = cnt;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
      const Index total = __shfl(inc, kWave - 1, kWave);
      __builtin_amdgcn_wave_barrier();
      s_pre[w][lane] = inc - cnt;
      s_p[w][lane] = p;
      s_acc[w][lane] = 0ull;
      __builtin_amdgcn_wave_barrier();
      for (Index base = 0; base < total; base += 4 * kWave) {
        Index c[4];
        int row[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const Index at = base + j * kWave + lane;
          c[j] = -1; row[j] = 0;
          if (at < total) {
            int r = 0;                                     // the last row whose first entry is <= at
#pragma unroll
            for (int step = kWave / 2; step > 0; step >>= 1)
              if (s_pre[w][r + step] <= at) r += step;
            row[j] = r;
            c[j] = a.iind[s_p[w][r] + (at - s_pre[w][r])];
          }
        }
        u64 wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = c[j] >= 0 ? a.fcur[c[j]] : 0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (wv[j]) atomicOr(&s_acc[w][row[j]], wv[j]);
      }
      __builtin_amdgcn_wave_barrier();
      acc |= s_acc[w][lane];
      p += cnt;
      if (quota < (1 << 20)) quota *= 4;
    }
    const u64 newb = acc & need;
    if (valid) {
      a.fnext[v] = newb;                                   // a big row's probe result: the slices add to it,
      if (newb && !big) a.seen[v] = seen | newb;           // batch_big_apply_kernel claims and counts it
    }
    batch_commit(a, tot, valid ? v : 0, big ? 0ull : newb);
  }
  totals_flush(a, &lds, tot);
}

// pull, big rows: a wave per 4096-entry slice
__global__ __launch_bounds__(kBlock) void batch_pull_slices_kernel(BatchArgs a) {
  const int lane = lane_id();
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int sl = blockIdx.x * kWavesPerBlock + wave_id(); sl < a.nslices; sl += nwaves) {
    const int4 S = a.slices[sl];
    const u64 need = ~a.seen[S.x] & a.qmask & ~a.fnext[S.x];   // what the probes (and other slices) left open
    if (!need) continue;
    const u64 got = wave_scan_or(a.iind, a.fcur, S.y, S.z, need, lane) & need;
    if (lane == 0 && got) atomicOr(&a.fnext[S.x], got);
  }
}

__global__ __launch_bounds__(kBlock) void batch_big_apply_kernel(BatchArgs a) {
  __shared__ BatchTotals lds;
  totals_init(&lds);
  WaveTotals tot;
  const int nthreads = gridDim.x * blockDim.x;
  for (int base = 0; base < a.nbig; base += nthreads) {
    const int b = base + blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < a.nbig;
    const Index v = valid ? a.bigrows[b] : 0;
    u64 newb = 0;
    if (valid) {
      const u64 seen = a.seen[v];
      newb = a.fnext[v] & ~seen & a.qmask;
      if (newb) a.seen[v] = seen | newb;
    }
    batch_commit(a, tot, v, newb);
  }
  totals_flush(a, &lds, tot);
}

// N out-edge slots of frontier vertices (entry q[j] < 0 = empty) carrying the pushed bits fw[j]: the dependent
// steps are issued stage by stage (targets, seen words, the claim in seen, a fire-and-forget OR of the claimed
// bits into fnext) -- one chain of memory latencies per N edges.  The claimed pairs are counted (and, beyond
// the stored levels, labelled) by batch_push_commit_kernel, which reads them back from fnext in one streaming
// pass instead of two random row-pointer reads and LDS atomics per pair here.
template <int N>
__device__ inline void batch_push_slots(const BatchArgs& a, const Index* __restrict__ ind, const Index (&q)[N],
                                        const u64 (&fw)[N]) {
  Index dst[N];
  u64 bits[N];
#pragma unroll
  for (int j = 0; j < N; ++j) dst[j] = q[j] >= 0 ? ind[q[j]] : -1;
#pragma unroll
  for (int j = 0; j < N; ++j) bits[j] = dst[j] >= 0 ? (fw[j] & ~a.seen[dst[j]]) : 0ull;
  if (a.prev) {
    // heavy level: the claim alone, fire and forget -- one random line per edge; the commit pass finds the
    // claimed bits as seen & ~prev
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (bits[j]) atomicOr(&a.seen[dst[j]], bits[j]);
    return;
  }
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (bits[j]) bits[j] &= ~atomicOr(&a.seen[dst[j]], bits[j]);
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (bits[j]) atomicOr(&a.fnext[dst[j]], bits[j]);
}

// push, rows below kBatchBig: the out-edges of a wave's 64 vertices are laid end to end (prefix sum of the
// degrees in LDS) and dealt to the lanes 256 at a time, so a wave pays one chain of memory latencies per 256
// edges whatever the row lengths are
__global__ __launch_bounds__(kBlock) void batch_push_kernel(BatchArgs a) {
  __shared__ Index s_pre[kWavesPerBlock][kWave];
  __shared__ Index s_p[kWavesPerBlock][kWave];
  __shared__ u64 s_fw[kWavesPerBlock][kWave];
  const int lane = lane_id(), w = wave_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + w; chunk < nchunks; chunk += nwaves) {
    const Index u = chunk * kWave + lane;
    const u64 fw = u < a.n ? (a.fcur[u] & a.pmask) : 0ull;
    if (__ballot(fw != 0) == 0ull) continue;
    Index p = 0, e = 0;
    if (fw) { p = a.optr[u]; e = a.optr[u + 1]; }
    if (e - p >= a.big) p = e;                             // the slice kernel expands it
    const Index d = e - p;
    Index inc = d;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
    const Index total = __shfl(inc, kWave - 1, kWave);
    __builtin_amdgcn_wave_barrier();
    s_pre[w][lane] = inc - d;
    s_p[w][lane] = p;
    s_fw[w][lane] = fw;
    __builtin_amdgcn_wave_barrier();
    for (Index base = 0; base < total; base += 4 * kWave) {
      Index q[4];
      u64 f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Index at = base + j * kWave + lane;
        q[j] = -1; f[j] = 0ull;
        if (at < total) {
          int r = 0;                                       // the last row whose first edge is <= at
#pragma unroll
          for (int step = kWave / 2; step > 0; step >>= 1)
            if (s_pre[w][r + step] <= at) r += step;
          q[j] = s_p[w][r] + (at - s_pre[w][r]);
          f[j] = s_fw[w][r];
        }
      }
      batch_push_slots<4>(a, a.oind, q, f);
    }
  }
}

__global__ __launch_bounds__(kBlock) void batch_push_slices_kernel(BatchArgs a) {
  const int lane = lane_id();
  const int nwaves = gridDim.x * kWavesPerBlock;
  for (int sl = blockIdx.x * kWavesPerBlock + wave_id(); sl < a.nslices; sl += nwaves) {
    const int4 S = a.slices[sl];
    const u64 fw = a.fcur[S.x] & a.pmask;
    if (!fw) continue;
    for (Index base = S.y + lane; base < S.z; base += 4 * kWave) {
      Index q[4];
      u64 f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Index at = base + j * kWave;
        q[j] = at < S.z ? at : -1;
        f[j] = fw;
      }
      batch_push_slots<4>(a, a.oind, q, f);
    }
  }
}

// ---- heavy push levels, the big rows by destination range ----------------------------------------------------------
// A heavy level pushes tens of millions of edges, nearly all of them out of a few thousand hub rows, and every edge
// was one atomicOr on a random 8-byte word of the 33 MB `seen` array: ~45 G edges/s, 290 us for the 15 M edges of
// RMAT-22's second level.  The rows are sorted, so a row's entries that fall into one range of destinations are one
// contiguous piece: a workgroup OWNS a range, keeps its words in LDS (at most 8 Ki of them, 64 KiB), ORs into them
// the pieces of every big frontier row (offsets per (row, range) precomputed once per matrix), and writes the range
// back with plain coalesced stores.  No global atomics, every edge read once: 293 -> 105 us for that level's big rows.
constexpr int kOwnRows = 8192;         // 64 KiB of LDS: two workgroups of 1024 per CU
constexpr int kOwnSmallRows = 2048;

__global__ void batch_range_off_kernel(const Index* __restrict__ optr, const Index* __restrict__ oind,
                                       const Index* __restrict__ bigrows, int nbig, int R, const Index* __restrict__ bounds,
                                       Index* __restrict__ off) {
  const long long total = (long long)nbig * (R + 1);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int bi = (int)(i / (R + 1)), b = (int)(i % (R + 1));
    const Index u = bigrows[bi];
    Index lo = optr[u], hi = optr[u + 1];
    const long long key = (long long)bounds[b];            // first entry with a destination >= key
    while (lo < hi) {
      const Index mid = lo + (hi - lo) / 2;
      if ((long long)oind[mid] < key) lo = mid + 1; else hi = mid;
    }
    off[(size_t)b * nbig + bi] = lo;                       // range-major: a range's owner reads neighbouring rows' offsets from one line
  }
}

__global__ __launch_bounds__(kBlock) void batch_big_list_kernel(BatchArgs a, int* __restrict__ list, u64* __restrict__ list_fw,
                                                                unsigned int* __restrict__ count) {
  const int lane = lane_id();
  for (int base = (blockIdx.x * kWavesPerBlock + wave_id()) * kWave; base < a.nbig; base += gridDim.x * kBlock) {
    const int bi = base + lane;
    const u64 fw = bi < a.nbig ? (a.fcur[a.bigrows[bi]] & a.pmask) : 0ull;
    const unsigned long long m = __ballot(fw != 0ull);
    if (!m) continue;
    unsigned int b0 = 0;
    if (lane == 0) b0 = atomicAdd(count, (unsigned int)__popcll(m));
    b0 = __shfl(b0, 0, kWave);
    if (fw) {
      const unsigned int at = b0 + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
      list[at] = bi;
      list_fw[at] = fw;                                    // the owners read (row, bits) in one step, not three
    }
  }
}

// A power-law graph sends a fifth of all edges to its first 16 Ki vertices: the ranges are cut at equal in-degree
// mass (at most 8 Ki rows each), so the hub region is many narrow ranges.  (Equal-width ranges with the heavy ones
// shared between several workgroups -- each with its own LDS copy, written back with atomicOr -- were measured
// first: 115-130 us.)  A lane takes a list entry: the piece of one big row inside the range; the pieces of a wave's
// entries are laid end to end and dealt to the lanes.  Two instantiations: the narrow ranges of the hub region
// (<= 2 Ki rows, 16 KiB of LDS, 512 threads: four workgroups per CU -- a workgroup's work is a short chain of
// dependent steps, and with one per CU the chip mostly waits) and the wide ones (two per CU): 27 + 77 us.
template <int kRows, int kThreads>
__global__ __launch_bounds__(kThreads) void batch_push_owner_kernel(BatchArgs a, const Index* __restrict__ range_off, int R,
                                                                    const int* __restrict__ list, const u64* __restrict__ list_fw,
                                                                    const unsigned int* __restrict__ count, const Index* __restrict__ bounds,
                                                                    const int* __restrict__ ids) {
  __shared__ u64 acc[kRows];
  __shared__ Index s_pre[kThreads / kWave][kWave];
  __shared__ Index s_p[kThreads / kWave][kWave];
  __shared__ u64 s_fw[kThreads / kWave][kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int b = ids[blockIdx.x];
  const Index base = bounds[b];
  const int rows = (int)(bounds[b + 1] - base);
  for (int i = tid; i < rows; i += kThreads) acc[i] = 0ull;
  __syncthreads();
  const long long nlist = (long long)*count;
  for (long long k0 = 0; k0 < nlist; k0 += kThreads) {
    const long long i = k0 + tid;
    Index o0 = 0, o1 = 0;
    u64 fw = 0ull;
    if (i < nlist) {
      const int bi = list[i];
      fw = list_fw[i];
      o0 = range_off[(size_t)b * a.nbig + bi];
      o1 = range_off[(size_t)(b + 1) * a.nbig + bi];
    }
    // the pieces of a wave's 64 entries laid end to end and dealt to the lanes 256 edges at a time (as batch_push_kernel
    // does with rows): a wave pays one chain of memory latencies per 256 edges whatever the piece lengths are
    const Index len = o1 - o0;
    Index inc = len;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
    const Index total = __shfl(inc, kWave - 1, kWave);
    if (total == 0) continue;
    const int w = tid >> 6;
    __builtin_amdgcn_wave_barrier();
    s_pre[w][lane] = inc - len;
    s_p[w][lane] = o0;
    s_fw[w][lane] = fw;
    __builtin_amdgcn_wave_barrier();
    constexpr int kPer = 8;                                // edges per lane and step: that many loads in flight
    for (Index at0 = 0; at0 < total; at0 += kPer * kWave) {
      Index q[kPer], d[kPer];
      u64 f[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        const Index at = at0 + j * kWave + lane;
        q[j] = -1; f[j] = 0ull;
        if (at < total) {
          int r = 0;                                       // the last entry whose first edge is <= at
#pragma unroll
          for (int step = kWave / 2; step > 0; step >>= 1)
            if (s_pre[w][r + step] <= at) r += step;
          q[j] = s_p[w][r] + (at - s_pre[w][r]);
          f[j] = s_fw[w][r];
        }
      }
#pragma unroll
      for (int j = 0; j < kPer; ++j) d[j] = q[j] >= 0 ? a.oind[q[j]] - base : -1;
#pragma unroll
      for (int j = 0; j < kPer; ++j)
        if (d[j] >= 0) atomicOr(&acc[d[j]], f[j]);
    }
  }
  __syncthreads();
  for (int i = tid; i < rows; i += kThreads) {
    const Index v = base + i;
    const u64 x = acc[i];
    if (x != 0ull) {
      const u64 sv = a.seen[v];
      if (x & ~sv) a.seen[v] = sv | x;                     // the claim; the commit pass finds it as seen & ~prev
    }
  }
}

// after the push kernels of a level: the pushed bits that arrived in fnext are this level's discoveries
__global__ __launch_bounds__(kBlock) void batch_push_commit_kernel(BatchArgs a) {
  __shared__ BatchTotals lds;
  totals_init(&lds);
  WaveTotals tot;
  const int lane = lane_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    u64 pb = 0;
    if (v < a.n) {
      if (a.prev) {
        pb = a.seen[v] & ~a.prev[v];
        if (pb) a.fnext[v] |= pb;
      } else {
        pb = a.fnext[v] & a.pmask;
      }
    }
    if (__ballot(pb != 0) == 0ull) continue;
    batch_commit(a, tot, pb ? v : 0, pb);
  }
  totals_flush(a, &lds, tot);
}

// the depth vectors from the stored level words: full 256-byte stores, every element written once
__global__ __launch_bounds__(kBlock) void batch_labels_kernel(LabelArgs a) {
  __shared__ float s_lab[32];
  if (threadIdx.x < 32) s_lab[threadIdx.x] = threadIdx.x >= 1 && (int)threadIdx.x <= a.nstored ? a.lab[threadIdx.x - 1] : 0.f;
  __syncthreads();
  const int lane = lane_id();
  const Index nchunks = (a.n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index chunk = (Index)blockIdx.x * kWavesPerBlock + wave_id(); chunk < nchunks; chunk += nwaves) {
    const Index v = chunk * kWave + lane;
    const bool valid = v < a.n;
    // which stored level found the vertex (1-based, <= kBatchStoreMax + 1 < 32) as five bit planes: plane b holds,
    // per source, bit b of that number (the level words are disjoint, so OR composes them); 0 = none of them
    u64 plane[5] = {0, 0, 0, 0, 0}, hit = 0;
    for (int L = 0; L < a.nstored; ++L) {
      const u64 w = valid ? a.W[L][v] : 0ull;
      hit |= w;
#pragma unroll
      for (int b = 0; b < 5; ++b) plane[b] |= (((L + 1) >> b) & 1) ? w : 0ull;
    }
    const u64 later = valid ? (a.seen[v] & ~hit) : ~0ull;  // seen, but by a level that labelled as it went
    for (int s = 0; s < a.k; ++s) {
      int idx = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) idx |= (int)((plane[b] >> s) & 1ull) << b;
      if (!((later >> s) & 1ull)) a.label[s][v] = s_lab[idx];
    }
  }
}

// the level's totals summed over the counter slots (left zero for the next level) and published to the host:
// box[0 .. kBatchCounters) the totals, box[kBatchCounters] the level's sequence number, written last
__global__ __launch_bounds__(kBlock) void batch_totals_kernel(u64* counters, u64* box, u64 seq) {
  const int i = threadIdx.x;
  if (i < kBatchCounters) {
    u64 t = 0;
    for (int slot = 0; slot < kBatchSlots; ++slot) {
      t += counters[slot * kBatchCounters + i];
      counters[slot * kBatchCounters + i] = 0ull;
    }
    __hip_atomic_store(&box[i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (i == 0) __hip_atomic_store(&box[kBatchCounters], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the light levels in one launch ----------------------------------------------------------------------------------
// A level that pushes a few thousand edges costs what its launches cost: four kernels over all n words, a totals
// kernel and a host round trip -- 45-100 us for microseconds of work, on the first level of a sweep and on every level
// of its tail (and on EVERY level of a high-diameter graph, where 64 small frontiers never grow).  While every live
// source is pushed and the level's out-edges stay under a limit, the levels run inside one co-resident launch
// instead: the frontier is a QUEUE of {vertex, first edge} pieces (rows cut at kTailPiece edges, so a hub that turns
// up is shared by many waves), a wave lays the edges of 64 pieces end to end and deals them to its lanes, a claimed
// target gets its bits ORed into the next level's words, and whoever finds that word zero appends the target to the
// next queue.  The words a level read are cleaned through its queue while the following level runs (three word arrays
// in rotation: a word holds bits of several sources, and a vertex one source reached two levels ago may be claimed by
// another right now -- the cleaner and the claimers must not share an array): no memset, no pass over n words after
// the first.
// Labels are written as the pairs are claimed (few of them, by definition); totals per source go through the same
// wave transposition as everywhere else.  One grid barrier per level.
constexpr int kTailPiece = 256;        // edges per queue entry: one step of a wave
constexpr int kTailList = 4096;        // a workgroup's list of newly claimed vertices (beyond it a wave queues its own)
constexpr int kTailLong = 64;          // ... of rows of more than eight pieces

struct TailState {                    // zeroed by the host before every launch
  GridBarrier bar;
  unsigned int count[4][32];          // queue lengths (one line each; [2..3]: the out-edges queued): level j reads [j & 3], appends to [(j + 1) & 3]
  u64 slots[3][kBatchSlots * kBatchCounters];
  u64 ts[64];                         // GRB_BATCH_TRACE: wall-clock stamps of workgroup 0 (100 MHz)
  u64 dec[3][8][16];                  // what the level loop decides on: {pairs, out-edges} of a level, one line per eighth of the grid
};

struct TailArgs {
  const u64* f0;                      // frontier words of the level before (a stored level: read, never cleaned)
  u64* X[3];                          // all-zero on entry: level j writes X[j % 3], reads X[(j - 1) % 3] and cleans X[(j - 2) % 3]
  u64* queue[3];                      // entries (first edge << 32 | vertex), vertex complemented on all but a row's first piece
  TailState* st;
  u64* box;                           // host-coherent: the last level's totals, then {seq, levels, edges, pairs, status}
  u64 seq;
  int iter0, max_niter;
  unsigned long long edge_limit;      // leave when the next level would push more out-edges than this
  const Index* src;                   // the sweep's first level: the sources (nsrc > 0), else the words of f0 are scanned
  int nsrc;
};

__global__ __launch_bounds__(kPThreads) void batch_tail_kernel(BatchArgs a, TailArgs t) {
  __shared__ BatchTotals lds;
  __shared__ Index s_pre[kPWaves][kWave];
  __shared__ Index s_p[kPWaves][kWave];
  __shared__ u64 s_fw[kPWaves][kWave];
  __shared__ u64 s_sum[kBatchCounters];
  __shared__ u64 s_dec[2];
  __shared__ Index s_list[kTailList];                       // vertices this workgroup was the first to claim this level
  __shared__ unsigned int s_nlist, s_wp[kPWaves], s_we[kPWaves], s_base, s_nlong;
  __shared__ unsigned long long s_before;
  __shared__ int4 s_long[kTailLong];                        // {vertex, first edge, pieces, where}: rows the whole workgroup writes
  const int lane = lane_id(), w = threadIdx.x / kWave;
  const unsigned int gw = blockIdx.x * kPWaves + w, nw = gridDim.x * kPWaves;
  const unsigned int gt = blockIdx.x * kPThreads + threadIdx.x, nt = gridDim.x * kPThreads;
  TailState* st = t.st;
  unsigned gen = 0;
  int nts = 0;
  auto stamp = [&] { if (blockIdx.x == 0 && threadIdx.x == 0 && nts < 64) st->ts[nts++] = wall_clock64(); };
  stamp();

  // wave-collective: every lane brings up to N vertices (v[i] < 0: none); one counter update per call for the whole
  // wave (a level that discovers 300 K vertices would otherwise queue behind 300 K updates of one address).  The
  // out-edges queued so far are counted as well: once they pass the limit the launch is going to hand the next level
  // back to the host, which reads the words, not the queue -- nothing more is written (the level that ends a sweep's
  // first launch discovers rows with 16 M out-edges: 60 K pieces nobody would read).  Rows of more than four pieces are
  // written by the whole wave.
  auto enqueue = [&](u64* q, unsigned int* cnt, const Index (&v)[4], int N) {
    Index p[4];
    unsigned int pieces[4], mine = 0, mine_edges = 0;
    for (int i = 0; i < N; ++i) {
      pieces[i] = 0; p[i] = 0;
      if (v[i] >= 0) {
        p[i] = a.optr[v[i]];
        const Index e = a.optr[v[i] + 1];
        pieces[i] = e > p[i] ? (unsigned int)((e - p[i] + kTailPiece - 1) / kTailPiece) : 1u;   // an empty row is queued too: its word must be cleaned
        mine += pieces[i];
        mine_edges += (unsigned int)(e - p[i]);
      }
    }
    if (__ballot(mine != 0u) == 0ull) return;
    unsigned int inc = mine, ince = mine_edges;
inc = (unsigned int)wave_incl_scan_u32((unsigned)inc);
ince = (unsigned int)wave_incl_scan_u32((unsigned)ince);
    unsigned int base = 0;
    unsigned long long before = 0;
    if (lane == kWave - 1) {
      base = atomicAdd(cnt, inc);
      before = atomicAdd((unsigned long long*)(cnt + 2), (unsigned long long)ince);
    }
    base = __shfl(base, kWave - 1, kWave);
    before = __shfl(before, kWave - 1, kWave);
    if (before > t.edge_limit) return;
    unsigned int at = base + inc - mine;
    for (int i = 0; i < N; ++i) {
      if (pieces[i] <= 4u)
        for (unsigned int c = 0; c < pieces[i]; ++c)
          publish(&q[at + c], ((u64)(unsigned int)(p[i] + (Index)c * kTailPiece) << 32) | (u64)(unsigned int)(c == 0 ? v[i] : ~v[i]));
      for (unsigned long long mk = __ballot(pieces[i] > 4u); mk; mk &= mk - 1) {
        const int l = __ffsll((long long)mk) - 1;
        const Index vv = __shfl(v[i], l, kWave), pp = __shfl(p[i], l, kWave);
        const unsigned int np = __shfl(pieces[i], l, kWave), aa = __shfl(at, l, kWave);
        for (unsigned int c = lane; c < np; c += kWave)
          publish(&q[aa + c], ((u64)(unsigned int)(pp + (Index)c * kTailPiece) << 32) | (u64)(unsigned int)(c == 0 ? vv : ~vv));
      }
      at += pieces[i];
    }
  };

  // The queue's length is one word: a wave per piece (what keeps a level's latency short) would mean a same-address
  // update per piece -- 5 ns each, 50 us for a level of 5 K pieces, measured.  So the waves of a workgroup collect
  // what they claim in LDS and the workgroup takes its room in the queue with ONE update per level; only a wave that
  // finds the LDS list full queues by itself.
  auto collect = [&](u64* q, unsigned int* cnt, const Index (&v)[4], int N) {
    unsigned int mine = 0;
    for (int i = 0; i < N; ++i) mine += v[i] >= 0 ? 1u : 0u;
    if (__ballot(mine != 0u) == 0ull) return;
    unsigned int inc = mine;
inc = (unsigned int)wave_incl_scan_u32((unsigned)inc);
    unsigned int base = 0;
    if (lane == kWave - 1) base = atomicAdd(&s_nlist, inc);
    base = __shfl(base, kWave - 1, kWave);
    const unsigned int total = __shfl(inc, kWave - 1, kWave);
    unsigned int at = base + inc - mine;
    Index rest[4] = {-1, -1, -1, -1};
    for (int i = 0; i < N; ++i)
      if (v[i] >= 0) {
        if (at < (unsigned int)kTailList) s_list[at] = v[i]; else rest[i] = v[i];   // the list stays dense up to its capacity
        ++at;
      }
    if (base + total > (unsigned int)kTailList) enqueue(q, cnt, rest, N);
  };
  auto block_flush = [&](u64* q, unsigned int* cnt) {
    __syncthreads();
    const unsigned int nl = s_nlist < (unsigned int)kTailList ? s_nlist : (unsigned int)kTailList;
    if (threadIdx.x == 0) s_nlong = 0;
    Index v[4], p[4];
    unsigned int pieces[4], mine = 0, mine_e = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int at = threadIdx.x * 4 + i;
      v[i] = -1; p[i] = 0; pieces[i] = 0;
      if (at < nl) {
        v[i] = s_list[at];
        p[i] = a.optr[v[i]];
        const Index e = a.optr[v[i] + 1];
        pieces[i] = e > p[i] ? (unsigned int)((e - p[i] + kTailPiece - 1) / kTailPiece) : 1u;
        mine += pieces[i];
        mine_e += (unsigned int)(e - p[i]);
      }
    }
    unsigned int inc = mine, ince = mine_e;
inc = (unsigned int)wave_incl_scan_u32((unsigned)inc);
ince = (unsigned int)wave_incl_scan_u32((unsigned)ince);
    if (lane == kWave - 1) { s_wp[w] = inc; s_we[w] = ince; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int tp = 0;
      unsigned long long te = 0;
      for (int i = 0; i < kPWaves; ++i) { const unsigned int x = s_wp[i]; s_wp[i] = tp; tp += x; te += s_we[i]; }
      s_base = 0; s_before = 0;
      if (tp) {
        s_base = atomicAdd(cnt, tp);
        s_before = atomicAdd((unsigned long long*)(cnt + 2), te);
      }
      s_nlist = 0;
    }
    __syncthreads();
    if (s_before <= t.edge_limit) {
      unsigned int at = s_base + s_wp[w] + inc - mine;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (pieces[i] > 8u) {
          const unsigned int li = atomicAdd(&s_nlong, 1u);
          if (li < (unsigned int)kTailLong) s_long[li] = make_int4(v[i], p[i], (int)pieces[i], (int)at);
          else
            for (unsigned int c = 0; c < pieces[i]; ++c)
              publish(&q[at + c], ((u64)(unsigned int)(p[i] + (Index)c * kTailPiece) << 32) | (u64)(unsigned int)(c == 0 ? v[i] : ~v[i]));
        } else {
          for (unsigned int c = 0; c < pieces[i]; ++c)
            publish(&q[at + c], ((u64)(unsigned int)(p[i] + (Index)c * kTailPiece) << 32) | (u64)(unsigned int)(c == 0 ? v[i] : ~v[i]));
        }
        at += pieces[i];
      }
      __syncthreads();
      const unsigned int nlong = s_nlong < (unsigned int)kTailLong ? s_nlong : (unsigned int)kTailLong;
      for (unsigned int li = 0; li < nlong; ++li) {
        const int4 L = s_long[li];
        for (unsigned int c = threadIdx.x; c < (unsigned int)L.z; c += kPThreads)
          publish(&q[(unsigned int)L.w + c], ((u64)(unsigned int)(L.y + (Index)c * kTailPiece) << 32) | (u64)(unsigned int)(c == 0 ? L.x : ~L.x));
      }
    }
    __syncthreads();
  };
  if (threadIdx.x == 0) s_nlist = 0;
  __syncthreads();

  if (t.nsrc > 0) {                                         // the first queue of a sweep's first level: the sources themselves
    if (gw == 0) {
      const Index u = lane < t.nsrc ? t.src[lane] : -1;
      // a vertex named twice is queued once: by the lane of its lowest source
      bool first = u >= 0;
      for (int o = 0; o < kWave; ++o) {
        const Index other = __shfl(u, o, kWave);
        if (o < lane && other == u) first = false;
      }
      const Index vq[4] = {first ? u : -1, -1, -1, -1};
      collect(t.queue[0], &st->count[0][0], vq, 1);
    }
  } else {                                                  // otherwise one pass over the stored frontier words, four chunks in flight
    const Index nchunks = (a.n + kWave - 1) / kWave;
    for (Index chunk = (Index)gw * 4; chunk < nchunks; chunk += (Index)nw * 4) {
      u64 fw[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const Index u = (chunk + r4) * kWave + lane;
        fw[r4] = u < a.n ? (t.f0[u] & a.pmask) : 0ull;
      }
      Index vq[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) vq[r4] = fw[r4] ? (chunk + r4) * kWave + lane : -1;
      collect(t.queue[0], &st->count[0][0], vq, 4);
    }
  }
  block_flush(t.queue[0], &st->count[0][0]);
  stamp();
  if (!grid_sync(&st->bar, gen)) return;
  stamp();

  int iter = t.iter0, j = 0, status = 0;
  unsigned long long levels = 0, cum_edges = 0, cum_pairs = 0;
  for (;;) {
    totals_init(&lds);
    WaveTotals tot;
    if (blockIdx.x == 0) {
      u64* z = st->slots[(j + 1) % 3];
      for (int i = threadIdx.x; i < kBatchSlots * kBatchCounters; i += kPThreads) publish(&z[i], 0ull);
      if (threadIdx.x < 4) publish(&st->count[(j + 2) & 3][threadIdx.x], 0u);   // length, and the out-edges behind it
      if (threadIdx.x < 16) publish(&st->dec[(j + 1) % 3][threadIdx.x >> 1][threadIdx.x & 1], 0ull);
    }
    if (j >= 2) {                                           // the words level j - 1 read: the array level j + 1 will write
      const unsigned int mq = fresh(&st->count[(j - 1) & 3][0]);
      const u64* qp = t.queue[(j - 1) % 3];
      u64* Xw = t.X[(j - 2) % 3];
      for (unsigned int i = gt; i < mq; i += nt) {
        const int x = (int)(unsigned int)qp[i];
        if (x >= 0) publish(&Xw[x], 0ull);
      }
    }
    const u64* F = j == 0 ? t.f0 : t.X[(j - 1) % 3];
    u64* Xn = t.X[j % 3];
    const u64* qc = t.queue[j % 3];
    u64* qn = t.queue[(j + 1) % 3];
    unsigned int* cn = &st->count[(j + 1) & 3][0];
    const unsigned int m = fresh(&st->count[j & 3][0]);
    const float lab = (float)(iter + 1);
    // a short queue is spread over the waves (a step is a chain of dependent memory round trips: 64 pieces in one
    // wave are 64 such chains one after the other, one piece in each of 64 waves is one)
    const unsigned int per = m >= (unsigned long long)nw * kWave ? (unsigned int)kWave : (m + nw - 1) / nw > 0 ? (m + nw - 1) / nw : 1u;
    for (unsigned int c = gw; (unsigned long long)c * per < m; c += nw) {
      const unsigned int idx = c * per + lane;
      const bool has = (unsigned int)lane < per && idx < m;
      const u64 ent = has ? qc[idx] : 0ull;
      const int x = (int)(unsigned int)ent;
      const Index u = x >= 0 ? x : ~x;
      Index p = (Index)(ent >> 32), e = p;
      u64 fw = 0;
      if (has) {
        const Index re = a.optr[u + 1];
        e = p + kTailPiece < re ? p + kTailPiece : re;
        fw = F[u] & a.pmask;
      }
      const Index d = e - p;
      Index inc = d;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
      const Index total = __shfl(inc, kWave - 1, kWave);
      __builtin_amdgcn_wave_barrier();
      s_pre[w][lane] = inc - d;
      s_p[w][lane] = p;
      s_fw[w][lane] = fw;
      __builtin_amdgcn_wave_barrier();
      for (Index base = 0; base < total; base += 4 * kWave) {
        Index dst[4];
        u64 bits[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const Index at = base + r4 * kWave + lane;
          dst[r4] = -1; bits[r4] = 0ull;
          if (at < total) {
            int r = 0;                                     // the last piece whose first edge is <= at
#pragma unroll
            for (int step = kWave / 2; step > 0; step >>= 1)
              if (s_pre[w][r + step] <= at) r += step;
            dst[r4] = a.oind[s_p[w][r] + (at - s_pre[w][r])];
            bits[r4] = s_fw[w][r];
          }
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) bits[r4] = dst[r4] >= 0 ? (bits[r4] & ~a.seen[dst[r4]]) : 0ull;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          if (bits[r4]) bits[r4] &= ~atomicOr(&a.seen[dst[r4]], bits[r4]);
        Index vq[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) vq[r4] = bits[r4] && atomicOr(&Xn[dst[r4]], bits[r4]) == 0ull ? dst[r4] : -1;
        collect(qn, cn, vq, 4);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) batch_commit_l(a, tot, bits[r4] ? dst[r4] : 0, bits[r4], true, lab);
      }
      __builtin_amdgcn_wave_barrier();
    }
    block_flush(qn, cn);
    stamp();
    totals_flush_to(st->slots[j % 3], &lds, tot);          // per source: read by workgroup 0 when the launch ends, by nobody else
    if (threadIdx.x < kWave) {                             // what everybody needs: the level's pairs and their out-edges
      u64 np = lds.v[2 + 2 * lane], ne = lds.v[3 + 2 * lane];
      np = wave_sum_u64(np); ne = wave_sum_u64(ne);
      if (lane == 0 && np) {
        atomicAdd(&st->dec[j % 3][blockIdx.x & 7][0], np);
        atomicAdd(&st->dec[j % 3][blockIdx.x & 7][1], ne);
      }
    }
    stamp();
    if (!grid_sync(&st->bar, gen)) return;
    stamp();
    if (threadIdx.x < kWave) {
      u64 x = lane < 16 ? fresh(&st->dec[j % 3][lane >> 1][lane & 1]) : 0ull;
      x += __shfl_xor(x, 2, kWave); x += __shfl_xor(x, 4, kWave); x += __shfl_xor(x, 8, kWave);   // lane 0: pairs, lane 1: edges
      if (lane < 2) s_dec[lane] = x;
    }
    __syncthreads();
    const u64 pairs = s_dec[0], edges = s_dec[1];
    ++levels;
    cum_pairs += pairs;
    cum_edges += edges;
    if (pairs == 0) { status = 0; break; }
    if (iter >= t.max_niter) { status = 1; break; }
    if (edges > t.edge_limit) { status = 2; break; }
    ++iter;
    ++j;
    __syncthreads();                                        // s_sum / s_dec are rewritten next level
  }
  if (j >= 1) {                                             // the words the last level read: every array but the new frontier's is left all-zero
    const unsigned int mq = fresh(&st->count[j & 3][0]);
    const u64* qp = t.queue[j % 3];
    u64* Xw = t.X[(j - 1) % 3];
    for (unsigned int i = gt; i < mq; i += nt) {
      const int x = (int)(unsigned int)qp[i];
      if (x >= 0) publish(&Xw[x], 0ull);
    }
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < kBatchCounters; i += kPThreads) s_sum[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < kBatchSlots * kBatchCounters; i += kPThreads) {
      const u64 x = fresh(&st->slots[j % 3][i]);
      if (x) atomicAdd(&s_sum[i % kBatchCounters], x);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBatchCounters; i += kPThreads)
      __hip_atomic_store(&t.box[i], s_sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&t.box[kBatchCounters + 1], (u64)levels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&t.box[kBatchCounters + 2], (u64)cum_edges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&t.box[kBatchCounters + 3], (u64)cum_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&t.box[kBatchCounters + 4], (u64)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&t.box[kBatchCounters], t.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void batch_unlabel_kernel(BatchArgs a, float bad) {
  for (int s = 0; s < a.k; ++s)
    for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x)
      if (a.label[s][i] == bad) a.label[s][i] = 0.f;
}

}  // namespace grb

using namespace grb;

// rows of >= kBatchBig entries cut into slices; cached per matrix and orientation
static int batch_big(bool in_edges) {
  static const int big_in = getenv("GRB_BATCH_BIG_IN") ? atoi(getenv("GRB_BATCH_BIG_IN")) : kBatchBig;
  static const int big_out = getenv("GRB_BATCH_BIG_OUT") ? atoi(getenv("GRB_BATCH_BIG_OUT")) : kBatchBigPush;
  const int b = in_edges ? big_in : big_out;
  return b < 64 ? 64 : b;
}

static grb_info ensure_slices(grb_matrix A, bool in_edges) {
  BatchSlices& B = in_edges ? A->batch_in : A->batch_out;
  if (B.ready) return GRB_SUCCESS;
  const std::vector<Index>& ptr = in_edges ? A->h_csc_ptr : A->h_csr_ptr;
  const Index n = in_edges ? A->ncols : A->nrows;
  if ((Index)ptr.size() != n + 1) return GRB_INVALID_OBJECT;
  std::vector<int4> sl;
  std::vector<Index> rows;
  for (Index v = 0; v < n; ++v) {
    const Index d = ptr[(size_t)v + 1] - ptr[v];
    if (d < batch_big(in_edges)) continue;
    const Index step = std::min<Index>(batch_big(in_edges), in_edges ? kBatchSlice : kBatchPushSlice);
    for (Index s = ptr[v]; s < ptr[(size_t)v + 1]; s += step)
      sl.push_back(make_int4(v, s, std::min<Index>(s + step, ptr[(size_t)v + 1]), (int)rows.size()));
    rows.push_back(v);
  }
  B.nslices = (int)sl.size();
  B.nbig = (int)rows.size();
  if (B.nslices > 0) {
    GRB_HIP_TRY(hipMalloc((void**)&B.d_slices, sizeof(int4) * sl.size()));
    GRB_HIP_TRY(hipMalloc((void**)&B.d_rows, sizeof(Index) * rows.size()));
    GRB_HIP_TRY(hipMemcpy(B.d_slices, sl.data(), sizeof(int4) * sl.size(), hipMemcpyHostToDevice));
    GRB_HIP_TRY(hipMemcpy(B.d_rows, rows.data(), sizeof(Index) * rows.size(), hipMemcpyHostToDevice));
    // out-edges: where each big row enters each destination range (the owner-computes push of heavy levels); ranges of
    // equal in-degree mass, about four per CU-sized share, at most kOwnRows rows
    if (!in_edges && n >= 2 * kOwnRows) {
      const std::vector<Index>& iptr = (Index)A->h_csc_ptr.size() == n + 1 ? A->h_csc_ptr : A->h_csr_ptr;
      const long long total = (long long)iptr[(size_t)n];
      const long long target = std::max<long long>(1, total / (3 * 256));
      std::vector<Index> bounds(1, 0);
      while (bounds.back() < n) {
        const Index s0 = bounds.back();
        Index e = (Index)std::min<long long>((long long)n, (long long)s0 + kOwnRows);
        // the last row whose prefix stays within the target, at least one row
        const Index* lo = iptr.data() + s0 + 1;
        const Index* hi = iptr.data() + e + 1;
        const Index* cut = std::upper_bound(lo, hi, (Index)std::min<long long>((long long)iptr[s0] + target, 0x7fffffffll));
        Index e2 = (Index)(cut - iptr.data()) - 1;
        if (e2 <= s0) e2 = s0 + 1;
        if (e2 < e) e = e2;
        bounds.push_back(e);
      }
      const long long R = (long long)bounds.size() - 1;
      if (R >= 2 && (long long)rows.size() * (R + 1) <= (128ll << 20)) {
        GRB_HIP_TRY(hipMalloc((void**)&B.d_range_bounds, sizeof(Index) * bounds.size()));
        GRB_HIP_TRY(hipMemcpy(B.d_range_bounds, bounds.data(), sizeof(Index) * bounds.size(), hipMemcpyHostToDevice));
        GRB_HIP_TRY(hipMalloc((void**)&B.d_range_off, sizeof(Index) * rows.size() * (size_t)(R + 1)));
        hipLaunchKernelGGL(batch_range_off_kernel, dim3(stream_grid((long long)rows.size() * (R + 1), kBlock)), dim3(kBlock), 0,
                           ctx().stream, A->csr.ptr, A->csr.ind, B.d_rows, (int)rows.size(), (int)R, (const Index*)B.d_range_bounds,
                           B.d_range_off);
        GRB_HIP_TRY(hipGetLastError());
        B.nranges = (int)R;
        std::vector<int> ids;
        for (int pass = 0; pass < 2; ++pass) {
          for (long long b = 0; b < R; ++b)
            if ((bounds[(size_t)b + 1] - bounds[(size_t)b] <= kOwnSmallRows) == (pass == 0)) ids.push_back((int)b);
          if (pass == 0) B.nsmall = (int)ids.size();
        }
        GRB_HIP_TRY(hipMalloc((void**)&B.d_range_ids, sizeof(int) * ids.size()));
        GRB_HIP_TRY(hipMemcpy(B.d_range_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice));
      }
    }
  }
  B.ready = true;
  return GRB_SUCCESS;
}

// out-edges a level may push inside the light-level launch (0: every level through the host loop)
static long long& batch_tail_limit() {
  static long long limit = [] {
    const char* on = getenv("GRB_BATCH_TAIL");
    if (on && atoi(on) == 0) return 0ll;
    const char* e = getenv("GRB_BATCH_TAIL_EDGES");
    return e ? atoll(e) : 1048576ll;
  }();
  return limit;
}

extern "C" long long grb_bfs_batch_set_tail(long long edges) { GRB_API_ENTER_NOINFO();
  const long long before = batch_tail_limit();
  if (edges >= 0) batch_tail_limit() = edges;
  return before;
}

extern "C" grb_info grb_bfs_batch(grb_vector* v, int k, grb_matrix A, const grb_index* sources, grb_descriptor desc,
                                  grb_bfs_result* result) { GRB_API_ENTER();
  if (!v || !A || !desc || !sources) return GRB_UNINITIALIZED_OBJECT;
  if (k < 1 || k > 64) return GRB_INVALID_VALUE;
  if (!A->built || !A->csr.ptr || !A->csc.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (A->nrows != A->ncols) return GRB_DIMENSION_MISMATCH;
  const Index n = A->nrows;
  for (int s = 0; s < k; ++s) {
    if (!v[s]) return GRB_UNINITIALIZED_OBJECT;
    if (v[s]->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
    if (v[s]->nsize != n) return GRB_DIMENSION_MISMATCH;
    if (sources[s] < 0 || sources[s] >= n) return GRB_INVALID_INDEX;
  }
  GRB_TRY(ctx_init());
  Context& c = ctx();
  hipStream_t st = c.stream;
  GRB_TRY(ensure_pull_hint(&A->d_pull_hint, A->csc, A->csr.ptr, st));
  GRB_TRY(ensure_slices(A, true));
  GRB_TRY(ensure_slices(A, false));

  // stored level words: as many as fit 1 GiB, at most kBatchStoreMax (+ the seed words) + two rotating buffers
  int nstore = (int)((1ull << 30) / (sizeof(u64) * (size_t)(n > 0 ? n : 1)));
  if (nstore > kBatchStoreMax) nstore = kBatchStoreMax;
  if (nstore < 1) nstore = 1;
  const int nbuf = 1 + (nstore + 1) + 4;                    // seen, the kept levels' words, four rotating arrays
  void *p_words, *p_cnt, *p_src;
  GRB_TRY(scratch(7, (size_t)nbuf * sizeof(u64) * (size_t)n + 256, &p_words));
  c.bfs_prezero_ptr = nullptr;                              // slot 7 is the one-launch traversal's pre-zeroed block
  GRB_TRY(scratch(10, sizeof(u64) * kBatchSlots * kBatchCounters, &p_cnt));
  GRB_TRY(scratch(9, sizeof(Index) * 64, &p_src));
  u64* seen = (u64*)p_words;
  // level words: slot 0 .. nstore are kept for the label pass (which level each holds is recorded as it is taken);
  // four more rotate -- a level beyond the kept ones (it labels as it discovers), the copy of `seen` a heavy push
  // compares against, the three word arrays of the light-level launch
  auto Slot = [&](int i) -> u64* { return seen + (size_t)(1 + i) * (size_t)n; };
  u64* const pool[4] = {Slot(nstore + 1), Slot(nstore + 2), Slot(nstore + 3), Slot(nstore + 4)};
  // known all-zero: the light-level launch leaves its arrays so, and the knowledge survives to the next sweep if nobody
  // else has had the scratch slot in between
  static struct { bool valid = false; unsigned long long epoch = 0; void* ptr = nullptr; Index n = 0; int nstore = 0; bool clean[4]; } kept_pool;
  bool pool_clean[4] = {false, false, false, false};
  if (kept_pool.valid && kept_pool.epoch + 1 == c.slot_epoch[7] && kept_pool.ptr == p_words && kept_pool.n == n && kept_pool.nstore == nstore)
    for (int i = 0; i < 4; ++i) pool_clean[i] = kept_pool.clean[i];
  kept_pool.valid = false;
  // a rotating array other than x and y: for the light-level launch a clean one if there is one (no memset), for
  // everybody else a dirty one (so that the clean ones stay clean)
  auto pick = [&](const u64* x, const u64* y, const u64* z, bool want_clean) -> int {
    int other = -1;
    for (int i = 0; i < 4; ++i) {
      if (pool[i] == x || pool[i] == y || pool[i] == z) continue;
      if (pool_clean[i] == want_clean) return i;
      if (other < 0) other = i;
    }
    return other;
  };
  const u64* kept_words[kBatchStoreMax + 1];
  int kept_level[kBatchStoreMax + 1];
  int nkept = 1;                                            // the seeds
  kept_words[0] = Slot(0);
  kept_level[0] = 0;
  bool any_direct = false;
  const u64* fcur_words = Slot(0);
  BatchArgs a;
  a.optr = A->csr.ptr; a.oind = A->csr.ind; a.iptr = A->csc.ptr; a.iind = A->csc.ind;
  a.hint = A->d_pull_hint;
  a.n = n;
  a.seen = seen;
  a.counters = (u64*)p_cnt;
  a.k = k;
  for (int s = 0; s < 64; ++s) a.label[s] = nullptr;
  for (int s = 0; s < k; ++s) {
    GRB_TRY(grb_vector_set_storage(v[s], GRB_DENSE));
    a.label[s] = (float*)v[s]->d_val;
  }
  static u64 *h_box = nullptr, *d_box = nullptr;            // {totals, sequence number}: pinned, host-coherent
  static u64 box_seq = 0;
  if (!h_box) {
    GRB_HIP_TRY(hipHostMalloc((void**)&h_box, sizeof(u64) * (kBatchCounters + 8), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h_box, 0, sizeof(u64) * (kBatchCounters + 8));
    GRB_HIP_TRY(hipHostGetDevicePointer((void**)&d_box, h_box, 0));
  }
  GRB_HIP_TRY(hipMemsetAsync(a.counters, 0, sizeof(u64) * kBatchSlots * kBatchCounters, st));
  GRB_HIP_TRY(hipMemsetAsync(seen, 0, 2 * sizeof(u64) * (size_t)n, st));       // seen and the seeds' words
  GRB_HIP_TRY(hipMemcpyAsync(p_src, sources, sizeof(Index) * (size_t)k, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(batch_seed_kernel, dim3(1), dim3(kBlock), 0, st, seen, Slot(0), (const Index*)p_src, k);
  GRB_HIP_TRY(hipGetLastError());

  // per-source frontier totals of the seed level
  unsigned long long nf_s[64], mf_s[64];
  unsigned long long edges = 0, reached = (unsigned long long)k;
  for (int s = 0; s < k; ++s) {
    nf_s[s] = 1;
    mf_s[s] = (unsigned long long)(A->h_csr_ptr[(size_t)sources[s] + 1] - A->h_csr_ptr[sources[s]]);
    edges += mf_s[s];
  }
  // GRB_SPARSE_MATRIX_FORMAT = 1: no CSC storage -- the "CSC" arrays ARE the CSR arrays, and the reference's vxm
  // is forced to push whatever the mxvmode says (operations.hpp:131-133).  A pull over them would walk out-edges
  // as if they were in-edges: every source is pushed.
  const int mode = (A->format != 0 || A->csc_alias) ? GRB_PUSHONLY : desc->desc[GRB_MXVMODE];
  const int grid = stream_grid((long long)ceil_div(n, kWave) * kWave, kBlock);
  int iter = 1, levels = 0, last_dir = 0;
  bool any_left = true;
  float ms = 0.f;
  static const bool trace = getenv("GRB_BATCH_TRACE") != nullptr;
  static const double budget = getenv("GRB_BATCH_BUDGET") ? atof(getenv("GRB_BATCH_BUDGET")) : 0.15;
  const double tail_edges = (double)batch_tail_limit();
  const bool tail_on = tail_edges > 0;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // results come back through the pinned, host-coherent box the host polls: no copy, no stream wait
  auto wait_box = [&]() -> grb_info {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&h_box[kBatchCounters], __ATOMIC_ACQUIRE) != box_seq) {
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
        GRB_HIP_TRY(hipStreamSynchronize(st));               // a long level, or a fault this wait reports
        if (__atomic_load_n(&h_box[kBatchCounters], __ATOMIC_ACQUIRE) != box_seq) return GRB_PANIC;
      }
    }
    return GRB_SUCCESS;
  };
  GRB_TRY(grb_timer_start());
  for (; iter <= desc->max_niter; ++iter) {
    const double t_lvl = trace ? now_us() : 0.0;
    // ---- direction per source: the reference's vertex-count rule (switchpoint) on that source's own
    // frontier, and a budget on the edges pushed in one level
    u64 P = 0, Q = 0;
    double pushed_edges = 0;
    {
      int order[64];
      int m = 0;
      for (int s = 0; s < k; ++s) if (nf_s[s] > 0) order[m++] = s;
      if (mode == GRB_PULLONLY) { for (int i = 0; i < m; ++i) Q |= 1ull << order[i]; }
      else if (mode == GRB_PUSHONLY) { for (int i = 0; i < m; ++i) P |= 1ull << order[i]; }
      else {
        // pull the sources whose own frontier passed the reference's switch point (their bits are found within
        // a few probes, so the early exit works); push the others -- unless their out-edges together exceed 0.15
        // of the matrix (GRB_BATCH_BUDGET: a pushed edge is a random line of seen plus an atomic on it, about
        // 50 G edges/s on RMAT-22 against 85 G/s for a pulled entry), then the heaviest of them are pulled as well
        std::sort(order, order + m, [&](int x, int y) { return mf_s[x] > mf_s[y]; });
        double pushed = 0;
        for (int i = 0; i < m; ++i)
          if ((double)nf_s[order[i]] <= (double)desc->switchpoint * (double)n) pushed += (double)mf_s[order[i]];
        for (int i = 0; i < m; ++i) {
          const int s = order[i];
          bool pull = (double)nf_s[s] > (double)desc->switchpoint * (double)n;
          if (!pull && pushed > budget * (double)A->nvals) { pull = true; pushed -= (double)mf_s[s]; }
          if (pull) Q |= 1ull << s; else P |= 1ull << s;
        }
      }
    }
    for (int s = 0; s < k; ++s) if ((P >> s) & 1ull) pushed_edges += (double)mf_s[s];
    a.qmask = Q; a.pmask = P;
    a.prev = nullptr;
    a.fcur = fcur_words;
    // ---- every live source pushed and few edges to push: this level and the light ones after it in one launch
    if (tail_on && Q == 0 && P != 0 && pushed_edges <= tail_edges) {
      int xi[3];
      xi[0] = pick(fcur_words, nullptr, nullptr, true);
      xi[1] = pick(fcur_words, pool[xi[0]], nullptr, true);
      xi[2] = pick(fcur_words, pool[xi[0]], pool[xi[1]], true);
      void* p_tail;
      const size_t qcap = (size_t)n + (size_t)(A->nvals / kTailPiece) + 64;
      const size_t st_bytes = (sizeof(TailState) + 255) & ~(size_t)255;
      GRB_TRY(scratch(8, st_bytes + 3 * sizeof(u64) * qcap, &p_tail));
      TailArgs t;
      t.f0 = fcur_words;
      for (int i = 0; i < 3; ++i) t.X[i] = pool[xi[i]];
      t.st = (TailState*)p_tail;
      for (int i = 0; i < 3; ++i) t.queue[i] = (u64*)((char*)p_tail + st_bytes) + (size_t)i * qcap;
      t.box = d_box;
      t.seq = ++box_seq;
      t.iter0 = iter;
      t.max_niter = desc->max_niter;
      t.edge_limit = (unsigned long long)tail_edges;
      t.src = (const Index*)p_src;
      t.nsrc = iter == 1 ? k : 0;
      a.direct_labels = 1;
      GRB_HIP_TRY(hipMemsetAsync(p_tail, 0, st_bytes, st));
      for (int i = 0; i < 3; ++i)
        if (!pool_clean[xi[i]]) GRB_HIP_TRY(hipMemsetAsync(t.X[i], 0, sizeof(u64) * (size_t)n, st));
      hipLaunchKernelGGL(batch_tail_kernel, dim3(c.num_cu), dim3(kPThreads), 0, st, a, t);
      GRB_HIP_TRY(hipGetLastError());
      GRB_TRY(wait_box());                                   // any number of levels; GRB_PANIC: a grid barrier gave up
      const int done = (int)h_box[kBatchCounters + 1];
      const int status = (int)h_box[kBatchCounters + 4];
      edges += h_box[kBatchCounters + 2];
      reached += h_box[kBatchCounters + 3];
      any_left = false;
      for (int s = 0; s < k; ++s) {
        nf_s[s] = h_box[2 + 2 * s];
        mf_s[s] = h_box[3 + 2 * s];
        if (nf_s[s]) any_left = true;
      }
      levels += done;
      last_dir = 0;
      any_direct = true;
      for (int i = 0; i < 3; ++i) pool_clean[xi[i]] = true;
      fcur_words = t.X[(done - 1) % 3];                     // the last level run, j = done - 1, wrote X[j % 3]
      if (status != 0) pool_clean[xi[(done - 1) % 3]] = false;   // the new frontier; every other array was left all-zero
      if (trace) {
        u64 ts[64];
        GRB_HIP_TRY(hipMemcpy(ts, &t.st->ts[0], sizeof(ts), hipMemcpyDeviceToHost));
        fprintf(stderr, "  launch stamps (us from start; prologue, barrier, then per level: work, flush, barrier):");
        for (int i = 1; i < 64 && ts[i]; ++i) fprintf(stderr, " %.1f", (double)(ts[i] - ts[0]) * 0.01);
        fprintf(stderr, "\n");
      }
      if (trace)
        fprintf(stderr, "batch levels %d..%d: one launch (light levels), status %d, pairs %llu  %.1f us\n", iter, iter + done - 1,
                status, (unsigned long long)h_box[kBatchCounters + 3], now_us() - t_lvl);
      iter += done - 1;                                     // the level the launch ended on
      if (status == 0) break;                               // nothing new: the traversals are over
      continue;                                             // capped (the loop condition ends it) or grown heavy again
    }
    const bool keep = nkept <= nstore;                       // a slot left: the label pass reads this level's words
    const int fi = keep ? -1 : pick(fcur_words, nullptr, nullptr, false);
    a.fnext = keep ? Slot(nkept) : pool[fi];
    if (fi >= 0) pool_clean[fi] = false;
    a.new_label = (float)(iter + 1);
    a.direct_labels = keep ? 0 : 1;
    if (keep) { kept_words[nkept] = a.fnext; kept_level[nkept] = iter; ++nkept; } else any_direct = true;
    if (Q) {
      const BatchSlices& B = A->batch_in;
      a.big = batch_big(true);
      a.slices = B.d_slices; a.nslices = B.nslices; a.bigrows = B.d_rows; a.nbig = B.nbig;
      hipLaunchKernelGGL(batch_pull_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
      if (B.nslices > 0) {
        hipLaunchKernelGGL(batch_pull_slices_kernel, dim3(stream_grid((long long)B.nslices * kWave, kBlock)), dim3(kBlock),
                           0, st, a);
        GRB_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(batch_big_apply_kernel, dim3(stream_grid(B.nbig, kBlock)), dim3(kBlock), 0, st, a);
        GRB_HIP_TRY(hipGetLastError());
      }
    } else {
      GRB_HIP_TRY(hipMemsetAsync(a.fnext, 0, sizeof(u64) * (size_t)n, st));
    }
    if (P) {
      const BatchSlices& B = A->batch_out;
      a.big = batch_big(false);
      if (pushed_edges > 0.5 * (double)n) {                 // heavy: claims only, the new bits read back against a copy
        const int pi = pick(fcur_words, a.fnext, nullptr, false);
        u64* prev = pool[pi];
        pool_clean[pi] = false;
        GRB_HIP_TRY(hipMemcpyAsync(prev, seen, sizeof(u64) * (size_t)n, hipMemcpyDeviceToDevice, st));
        a.prev = prev;
      }
      a.slices = B.d_slices; a.nslices = B.nslices; a.bigrows = B.d_rows; a.nbig = B.nbig;
      hipLaunchKernelGGL(batch_push_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
      static const bool owner_ok = [] { const char* e = getenv("GRB_BATCH_OWNER"); return !e || atoi(e) != 0; }();
      if (B.nslices > 0 && a.prev && owner_ok && B.d_range_off) {
        // heavy level: the big rows' edges are settled by the owners of their destination ranges, in LDS
        void* p_list;
        const size_t list_bytes = (sizeof(int) * (size_t)B.nbig + 255) & ~(size_t)255;
        GRB_TRY(scratch(11, 256 + list_bytes + sizeof(u64) * (size_t)B.nbig, &p_list));
        unsigned int* d_count = (unsigned int*)p_list;
        int* d_list = (int*)((char*)p_list + 256);
        u64* d_list_fw = (u64*)((char*)p_list + 256 + list_bytes);
        GRB_HIP_TRY(hipMemsetAsync(d_count, 0, 4, st));
        hipLaunchKernelGGL(batch_big_list_kernel, dim3(stream_grid(B.nbig, kBlock)), dim3(kBlock), 0, st, a, d_list, d_list_fw, d_count);
        GRB_HIP_TRY(hipGetLastError());
        if (B.nsmall > 0)
          hipLaunchKernelGGL((batch_push_owner_kernel<kOwnSmallRows, 512>), dim3(B.nsmall), dim3(512), 0, st, a, (const Index*)B.d_range_off,
                             B.nranges, (const int*)d_list, (const u64*)d_list_fw, (const unsigned int*)d_count,
                             (const Index*)B.d_range_bounds, (const int*)B.d_range_ids);
        if (B.nranges > B.nsmall)
          hipLaunchKernelGGL((batch_push_owner_kernel<kOwnRows, 1024>), dim3(B.nranges - B.nsmall), dim3(1024), 0, st, a,
                             (const Index*)B.d_range_off, B.nranges, (const int*)d_list, (const u64*)d_list_fw, (const unsigned int*)d_count,
                             (const Index*)B.d_range_bounds, (const int*)B.d_range_ids + B.nsmall);
        GRB_HIP_TRY(hipGetLastError());
        if (trace) {
          unsigned int hc = 0;
          GRB_HIP_TRY(hipMemcpy(&hc, d_count, 4, hipMemcpyDeviceToHost));
          fprintf(stderr, "batch level %d: owner-computes push, %u of %d big rows in the frontier, %d ranges\n", iter, hc, B.nbig, B.nranges);
        }
      } else if (B.nslices > 0) {
        hipLaunchKernelGGL(batch_push_slices_kernel, dim3(stream_grid((long long)B.nslices * kWave, kBlock)), dim3(kBlock),
                           0, st, a);
        GRB_HIP_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(batch_push_commit_kernel, dim3(grid), dim3(kBlock), 0, st, a);
      GRB_HIP_TRY(hipGetLastError());
    }
    // the totals come back through a pinned, host-coherent box the host polls: no copy, no stream wait
    hipLaunchKernelGGL(batch_totals_kernel, dim3(1), dim3(kBlock), 0, st, a.counters, d_box, ++box_seq);
    GRB_HIP_TRY(hipGetLastError());
    GRB_TRY(wait_box());
    u64 t[kBatchCounters];
    for (int j = 0; j < kBatchCounters; ++j) t[j] = __atomic_load_n(&h_box[j], __ATOMIC_RELAXED);
    ++levels;
    last_dir = Q ? 1 : 0;
    fcur_words = a.fnext;
    any_left = false;
    u64 pairs = 0;
    for (int s = 0; s < k; ++s) {
      nf_s[s] = t[2 + 2 * s];
      mf_s[s] = t[3 + 2 * s];
      pairs += nf_s[s];
      reached += nf_s[s];
      edges += mf_s[s];
      if (nf_s[s]) any_left = true;
    }
    if (trace)
      fprintf(stderr, "batch level %d: pull %d sources, push %d (%.0f edges, %s) -> vertices %llu pairs %llu  %.1f us\n",
              iter, __builtin_popcountll(Q), __builtin_popcountll(P), pushed_edges, a.prev ? "claims" : "direct", t[0],
              pairs, now_us() - t_lvl);
    if (!any_left) break;
  }
  const bool hit_cap = iter > desc->max_niter && any_left;
  // ---- the depth vectors
  {
    LabelArgs L;
    L.n = n; L.k = k;
    L.nstored = nkept;
    L.seen = seen;
    for (int i = 0; i <= kBatchStoreMax; ++i) {
      L.W[i] = i < nkept ? kept_words[i] : nullptr;
      // discovered by the last allowed iteration: never assigned (bfs.hpp:48-66)
      L.lab[i] = i < nkept && kept_level[i] + 1 <= desc->max_niter ? (float)(kept_level[i] + 1) : 0.f;
    }
    for (int s = 0; s < 64; ++s) L.label[s] = a.label[s];
    hipLaunchKernelGGL(batch_labels_kernel, dim3(grid), dim3(kBlock), 0, st, L);
    GRB_HIP_TRY(hipGetLastError());
  }
  if (hit_cap && any_direct) {
    // vertices discovered by the last allowed iteration are never assigned by the reference loop (bfs.hpp:48-66)
    hipLaunchKernelGGL(batch_unlabel_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, a,
                       (float)(desc->max_niter + 1));
    GRB_HIP_TRY(hipGetLastError());
  }
  GRB_TRY(grb_timer_stop(&ms));
  kept_pool.valid = true;
  kept_pool.epoch = c.slot_epoch[7];
  kept_pool.ptr = p_words;
  kept_pool.n = n;
  kept_pool.nstore = nstore;
  for (int i = 0; i < 4; ++i) kept_pool.clean[i] = pool_clean[i];
  desc->lastmxv = last_dir ? GRB_PULLONLY : GRB_PUSHONLY;
  if (result) {
    result->levels = levels;
    result->tight_ms = ms;
    result->edges_traversed = hit_cap ? -1 : (int64_t)edges;   // under a cap the tally would count unassigned vertices
    result->reached = hit_cap ? -1 : (int32_t)(reached > 0x7fffffffull ? 0x7fffffff : reached);
  }
  return GRB_SUCCESS;
}
