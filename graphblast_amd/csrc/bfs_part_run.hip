// bfs_part_run.hip -- the 1-D vertex-partitioned BFS with its level loop on the DEVICE.
//
// bfs_part.hip holds the level steps a host loop drives (two launches, a device-to-host wait and the
// host's own bookkeeping per level: 0.49 ms per traversal of RMAT-22 on one rank against 0.17 ms for the
// one-launch kernel of bfs_persist.hip).  Here a rank runs ONE co-resident launch per level and nothing is
// read back until the traversal has ended:
//
//   launch k (level k + 1)   apply   OR of every rank's new-bits bitmap of the previous level (the all-gather's
//                                    receive buffer) -> visited |=, the owned part becomes this level's frontier
//                                    (kept for the final label pass), level totals from the replicated
//                                    out-degree array: |frontier|, its out-degree sum, owned big vertices
//                                    (listed as 1024-edge entries in the same pass)
//                            -- grid barrier --
//                            decide  the reference's convert rule (vector.hpp:291-323) + the edge-aware
//                                    extension, evaluated redundantly by every workgroup of every rank from
//                                    the totals: identical everywhere, no collective, no host
//                            expand  push the owned frontier's out-edges (new bits anywhere) or pull the owned
//                                    unvisited vertices' in-edges (new bits in the owned words)
//   collective k             all-gather of the n/8-byte new-bits bitmaps (csrc/comm.hip: RCCL on the
//                            communication stream, event-fenced; 512 KiB per rank at RMAT-22)
//
// The scalars of the loop (level counter, ratio slots, frontier size, cumulative counts, the `done` flag)
// travel from launch to launch in device memory.  The host enqueues launch k + 1 when launch k - 1 has
// reported (one word in pinned memory, polled -- the device always has a launch queued and never waits for
// the host); that rule depends only on values every rank computes identically, so all ranks enqueue the same
// number of collectives.  A launch that finds the traversal finished exits at once; one launch and one
// collective at most are spent after the end.  With one rank and levels_per_launch > 1 the same kernel runs
// several levels per launch (there is no collective to wait for).
//
// The reference has nothing to mirror (--ndevice is parsed and ignored, backend/cuda/descriptor.hpp:242);
// results are those of algorithm::bfs (graphblas/algorithm/bfs.hpp:14-89) on the whole graph.
#include "bfs_kernels.hpp"
#include "persist_common.hpp"

#include <chrono>

namespace grb {

constexpr int kPKeep = 32;          // levels whose discovered-bitmaps are kept for the final label pass
#ifndef GRB_PART_SMALL_DEG
#define GRB_PART_SMALL_DEG 16
#endif
#ifndef GRB_PART_BIG_DEG
#define GRB_PART_BIG_DEG 512
#endif
constexpr int kPSmallDeg = GRB_PART_SMALL_DEG;      // push: below, expanded by the lane that found the vertex
constexpr int kPBigDeg = GRB_PART_BIG_DEG;          // push: from here, cut into kPBigChunk-edge entries for whole workgroups
constexpr int kPBigChunk = 1024;
constexpr int kPPullBlock = 8;      // 64-vertex chunks a wave carries through the pull stages together
constexpr int kPMedCap = 4096;

struct PartCarry {                  // the level loop's scalars: identical on every rank and in every workgroup
  int iter;                         // the next level to expand (1 = the source's)
  int f1_dense;
  float ratio_f1, ratio_f2;
  unsigned nf;                      // vertices in the frontier of level `iter`
  unsigned nbig;                    // OWNED frontier vertices with >= kPBigDeg out-edges
  unsigned long long mf;            // out-degree sum of the whole frontier
  long long reached;
  unsigned long long edges_cum;     // out-degree sum of everything reached (TEPS numerator)
  int levels, last_dir, done, hit_cap;
  int done_at;                      // the launch that ended the traversal
};

struct PartState {                  // zeroed by the host before a traversal
  GridBarrier bar[2];               // launch k uses bar[k & 1] and clears the other one
  unsigned big_count[2][32];
  unsigned long long acc[3][8][16]; // level totals, one line per (set, XCD group): found, out-degree sum, owned big
  unsigned panic[32];
  PartCarry carry[2];               // launch k reads carry[k & 1], writes carry[(k + 1) & 1]
};

struct PartArgs {
  const Index *optr, *oind;         // out-edges of the owned vertices: local rows, GLOBAL column ids
  const Index *iptr, *iind;         // their in-edges
  long long innz;                   // stored in-edges of this shard
  const unsigned int* skip;         // local bitmap: owned vertices without in-edges (and the padding bits)
  long long n_in_local;             // owned vertices WITH in-edges (-1 unknown)
  const Index* hint;                // local: the in-neighbour (global id) of largest out-degree; may be null
  const int* deg;                   // [n] out-degree of every vertex (replicated)
  Index n, lo, n_local;
  long long nnz;                    // stored edges of the whole graph (edge-aware switch)
  Index source;
  int mode;
  float switchpoint, edgeswitch;
  int max_niter;
  int world;
  const unsigned int* parts;        // `world` bitmaps of nwords words: the ranks' new bits of the previous level
  unsigned int* Fn;                 // this rank's new bits: the send buffer (== parts when world == 1)
  unsigned int* V;                  // visited, replicated
  unsigned int* F;                  // (kPKeep + 1) x local_words: the owned part of every level's frontier
  float* label;                     // owned labels
  int2* big_list;
  int big_cap;
  // owner-computes push (bfs_persist.hip: oc_tables_build) for this rank's big rows; oc_off == nullptr: the big
  // vertices' edges go out with atomics, 1024 at a time
  const Index* oc_bounds;
  const Index* oc_off;
  const int* oc_bigidx;
  int oc_nb, oc_nrows;
  PartState* st;
  grb_bfs_level* rec;               // pinned host memory: one record per level
  int rec_cap;
  unsigned long long* mail;         // pinned host word: (launch that ended the traversal + 1) << 32 | launches completed
  PartCarry* result;                // pinned host memory: the scalars as the ending launch leaves them
  int launch;                       // index of this launch within the traversal
  int nlevels;                      // levels this launch may expand (1 unless world == 1)
};

struct __attribute__((packed, aligned(4))) PQuad { Index x, y, z, w; };

__device__ inline int part_fslot(int level) { return level < kPKeep ? level : kPKeep; }

__device__ inline void part_push_visit(unsigned int* V, unsigned int* Fn, Index dst) {
  const unsigned int bit = 1u << (dst & 31);
  if (fresh(&V[dst >> 5]) & bit) return;
  const unsigned int old = atomicOr(&V[dst >> 5], bit);
  if (old & bit) return;
  atomicOr(&Fn[dst >> 5], bit);
}

__global__ __launch_bounds__(kPThreads) void bfs_part_level_kernel(PartArgs a) {
  __shared__ unsigned long long s_red[kPWaves][3];
  __shared__ unsigned long long s_tot[3];
  __shared__ Index s_med[kPMedCap];
  __shared__ int s_nmed;
  __shared__ PullLds s_pull[kPWaves];                 // one per wave: the pull levels' row queue (persist_common.hpp)
  __shared__ unsigned int s_ocw[kOcWords];
  int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int G = gridDim.x;
  long long gtid = (long long)blockIdx.x * kPThreads + tid;
  // every phase starts from a thread id the optimiser cannot see through (bfs_persist.hip: what a phase derives from
  // it lives in that phase only, instead of in registers across the whole kernel)
#define GRB_PART_PHASE()                                          \
  do {                                                            \
    asm volatile("" : "+v"(tid));                                 \
    lane = tid & (kWave - 1);                                     \
    wave = tid >> 6;                                              \
    gtid = (long long)blockIdx.x * kPThreads + tid;               \
  } while (0)
  const long long gthreads = (long long)G * kPThreads;
  const Index n = a.n;
  const int nwords = 2 * ((n + 63) / 64);
  const int local_words = 2 * ((a.n_local + 63) / 64);
  const int lo_w = a.lo >> 5;
  const int k = a.launch;
  PartState* st = a.st;
  GridBarrier* bar = &st->bar[k & 1];
  unsigned gen = 0;

  if (blockIdx.x == 0) {            // the next launch's barrier block (nobody touches it during this launch)
    unsigned* z = reinterpret_cast<unsigned*>(&st->bar[(k + 1) & 1]);
    for (int i = tid; i < (int)(sizeof(GridBarrier) / sizeof(unsigned)); i += kPThreads) publish(&z[i], 0u);
  }

  PartCarry cy;
  if (k == 0) {
    cy.iter = 1;
    cy.f1_dense = (a.mode == GRB_PULLONLY) ? 1 : 0;
    cy.ratio_f1 = cy.ratio_f2 = 0.f;
    cy.nf = 1;
    cy.nbig = 0;
    cy.mf = (unsigned long long)a.deg[a.source];
    cy.reached = 1;
    cy.edges_cum = cy.mf;
    cy.levels = cy.last_dir = cy.done = cy.hit_cap = 0;
    cy.done_at = -1;
  } else {
    cy = st->carry[k & 1];
  }
  const bool panicked = fresh(&st->panic[0]) != 0u;
  auto report = [&](const PartCarry& y, bool bad) {   // one 8-byte word: the data is the flag
    const unsigned long long hi = bad ? 0xffffffffull : (unsigned long long)(unsigned)(y.done_at + 1);
    __hip_atomic_store(a.mail, (hi << 32) | (unsigned long long)(unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (cy.done || panicked) {                           // the traversal ended in an earlier launch: hand the scalars on
    if (gtid == 0) {
      st->carry[(k + 1) & 1] = cy;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      report(cy, panicked);
    }
    return;
  }
  auto give_up = [&]() { if (tid == 0) publish(&st->panic[0], 1u); };

  for (int step = 0; step < a.nlevels; ++step) {
    const int iter = cy.iter;
    unsigned int* Fcur = a.F + (size_t)part_fslot(iter - 1) * (size_t)local_words;   // frontier of this level (owned words)

    if (iter > 1) {
      // ================= apply: what level iter - 1 discovered, on every rank =================
      GRB_PART_PHASE();
      const bool count_only = iter > a.max_niter;        // the loop has ended: the last level is only counted
      const bool direct = (iter - 1) >= kPKeep;          // its bitmap will be recycled: label now
      const float lab = (float)iter;
      unsigned* bcount = &st->big_count[iter & 1][0];
      if (gtid == 0) publish(&st->big_count[(iter + 1) & 1][0], 0u);
      if (blockIdx.x == 0 && tid < 32) publish(&st->acc[(iter + 1) % 3][tid >> 2][tid & 3], 0ull);
      unsigned long long c_found = 0, c_deg = 0, c_big = 0;
      for (long long base = 0; base < nwords; base += gthreads) {
        const long long i = base + gtid;
        unsigned int f = 0;
        bool owned = false;
        if (i < nwords) {
          unsigned int x = 0;
          for (int r = 0; r < a.world; ++r) x |= fresh(&a.parts[(size_t)r * (size_t)nwords + (size_t)i]);
          const unsigned int mine = (a.world == 1) ? x : fresh(&a.Fn[i]);
          const unsigned int v = fresh(&a.V[i]);
          // a push marks its own discoveries in V at once (deduplication), so they are taken from `mine`
          f = (x & ~v) | mine;
          owned = i >= lo_w && i < lo_w + local_words;
          if (!count_only) {
            if (mine) publish(&a.Fn[i], 0u);             // the send buffer of the next level starts clear
            if (f & ~v) publish(&a.V[i], v | f);
            if (owned) publish(&Fcur[i - lo_w], f);
          }
        }
        int ent = 0;
        // the word's vertices eight at a time: their out-degrees are eight independent random reads of the replicated
        // degree array, and a bit-by-bit loop waits for each before it asks for the next -- a word of the big level holds
        // up to 32 discoveries, i.e. 32 memory latencies on the level's critical path (the apply pass of that level was
        // ~20 us of the traversal's 0.2 ms)
        for (unsigned int t = f; t;) {
          Index vt[8];
          int dd[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            vt[j] = t ? (Index)i * 32 + (__ffs((int)t) - 1) : -1;
            t &= t - 1;                                      // (0 stays 0)
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) dd[j] = vt[j] >= 0 ? a.deg[vt[j]] : 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (vt[j] < 0) continue;
            const Index vtx = vt[j];
            const int d = dd[j];
            ++c_found;
            c_deg += (unsigned long long)d;
            if (owned && count_only) a.label[vtx - a.lo] = 0.f;   // found by the level that ended the loop: never assigned (bfs.hpp:48-66)
            if (owned && !count_only) {
              if (direct) a.label[vtx - a.lo] = lab;
              if (d >= kPBigDeg) { ++c_big; ent += a.oc_off ? 1 : (d + kPBigChunk - 1) / kPBigChunk; }
            }
          }
        }
        if (__ballot(ent > 0)) {                          // the owned big vertices as 1024-edge entries
          int incl = ent;
incl = (int)wave_incl_scan_u32((unsigned)incl);
          const int total = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
          unsigned b0 = 0;
          if (lane == 0) b0 = atomicAdd(bcount, (unsigned)total);
          b0 = __shfl(b0, 0, kWave);
          int at = (int)b0 + incl - ent;
          if (ent > 0)
            for (unsigned int t = f; t; t &= t - 1) {
              const Index vtx = (Index)i * 32 + (__ffs((int)t) - 1);
              const int d = a.deg[vtx];
              if (d >= kPBigDeg)
                for (int kk = 0; kk < (a.oc_off ? 1 : (d + kPBigChunk - 1) / kPBigChunk); ++kk, ++at)
                  if (at < a.big_cap)
                    publish(reinterpret_cast<unsigned long long*>(&a.big_list[at]),
                            ((unsigned long long)(unsigned)kk << 32) | (unsigned)(vtx - a.lo));
            }
        }
      }
      // ---- totals: one atomic per value per workgroup into this XCD group's line
      const unsigned long long r0 = wave_sum_u64(c_found), r1 = wave_sum_u64(c_deg), r2 = wave_sum_u64(c_big);
      if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; }
      __syncthreads();
      unsigned long long* acc = &st->acc[iter % 3][0][0];
      if (tid < 3) {
        unsigned long long t = 0;
        for (int w = 0; w < kPWaves; ++w) t += s_red[w][tid];
        if (t) __hip_atomic_fetch_add(&acc[(blockIdx.x & 7) * 16 + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!grid_sync(bar, gen, false)) { give_up(); return; }
      if (wave == 0) {
        unsigned long long q = 0;
        if (lane < 32) q = __hip_atomic_load(&acc[(lane >> 2) * 16 + (lane & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q += __shfl_xor(q, 4, kWave);
        q += __shfl_xor(q, 8, kWave);
        q += __shfl_xor(q, 16, kWave);
        if (lane < 3) s_tot[lane] = q;
      }
      __syncthreads();
      const unsigned long long tot_found = s_tot[0], tot_deg = s_tot[1], tot_big = s_tot[2];
      __syncthreads();
      if (gtid == 0 && cy.levels < a.rec_cap) {
        grb_bfs_level& L = a.rec[cy.levels];
        L.direction = cy.last_dir;
        L.frontier = (int32_t)cy.nf;
        L.frontier_edges = cy.last_dir ? 0 : (int64_t)cy.mf;
        L.discovered = (int32_t)tot_found;
        L.ms = 0.f;
      }
      ++cy.levels;
      const float tmp = cy.ratio_f1; cy.ratio_f1 = cy.ratio_f2; cy.ratio_f2 = tmp;
      cy.nf = (unsigned)tot_found;
      cy.mf = tot_deg;
      cy.nbig = (unsigned)tot_big;
      if (count_only) { cy.hit_cap = cy.nf > 0 ? 1 : 0; cy.done = 1; cy.done_at = k; break; }
      cy.reached += (long long)tot_found;
      cy.edges_cum += tot_deg;
      if (cy.nf == 0) { cy.done = 1; cy.done_at = k; break; }
    }

    // ================= decide (vector.hpp:291-323; identical in every workgroup of every rank) =================
    if (a.mode == GRB_PUSHPULL) {
      const float ratio = (float)cy.nf / (float)n;
      if (!cy.f1_dense) {
        if (ratio > a.switchpoint && ratio > cy.ratio_f1) cy.f1_dense = 1; else cy.ratio_f1 = ratio;
      } else {
        if (ratio <= a.switchpoint && ratio < cy.ratio_f1) cy.f1_dense = 0; else cy.ratio_f1 = ratio;
      }
      if (!cy.f1_dense && a.edgeswitch > 0.f && cy.nf >= 32 && (double)cy.mf > (double)a.edgeswitch * (double)a.nnz)
        cy.f1_dense = 1;
    } else {
      cy.f1_dense = (a.mode == GRB_PULLONLY) ? 1 : 0;
    }
    const bool pull = cy.f1_dense != 0;

    if (iter == 1) {
      // the source: visited everywhere, in the frontier bitmap of its owner (V, Fn and F[0] arrive zeroed)
      if (gtid == 0) {
        atomicOr(&a.V[a.source >> 5], 1u << (a.source & 31));
        if (a.source >= a.lo && a.source < a.lo + a.n_local) publish(&Fcur[(a.source - a.lo) >> 5], 1u << (a.source & 31));
      }
      if (pull && !grid_sync(bar, gen, false)) { give_up(); return; }
    }

    if (!pull) {
      // ================= push: the owned frontier's out-edges =================
      GRB_PART_PHASE();
      if (iter == 1) {
        if (a.source >= a.lo && a.source < a.lo + a.n_local) {
          const Index sl = a.source - a.lo;
          const Index e = a.optr[sl + 1];
          for (long long p = a.optr[sl] + gtid; p < e; p += gthreads) {
            const Index dst = a.oind[p];
            if (dst != a.source) part_push_visit(a.V, a.Fn, dst);
          }
        }
      } else {
        if (cy.nbig > 0) {
          int nent = (int)fresh(&st->big_count[iter & 1][0]);
          if (nent > a.big_cap) nent = a.big_cap;
          if (!a.oc_off) {
            for (int e = blockIdx.x; e < nent; e += G) {
              const unsigned long long eb = fresh(reinterpret_cast<const unsigned long long*>(&a.big_list[e]));
              const Index vl = (Index)(eb & 0xffffffffull);
              const Index p = a.optr[vl] + (Index)(eb >> 32) * kPBigChunk + tid;
              if (p < a.optr[vl + 1]) part_push_visit(a.V, a.Fn, a.oind[p]);
            }
          } else {
            // owner-computes (as bfs_persistent_kernel's heavy levels): this workgroup's destination ranges, the
            // pieces of every listed big vertex ORed into LDS, one atomicOr per changed word of V / Fn
            for (int b = blockIdx.x; b < a.oc_nb; b += G) {
              const Index v0 = a.oc_bounds[b];
              const int w0 = (int)(v0 >> 5);
              int nw = (int)((a.oc_bounds[b + 1] - v0 + 31) >> 5);
              if (w0 + nw > nwords) nw = nwords - w0;
              __syncthreads();
              for (int i = tid; i < nw; i += kPThreads) s_ocw[i] = 0u;
              __syncthreads();
              for (int e0 = 0; e0 < nent; e0 += kPThreads) {
                const int e = e0 + tid;
                Index o0 = 0, o1 = 0;
                if (e < nent) {
                  const Index vl = (Index)(fresh(reinterpret_cast<const unsigned long long*>(&a.big_list[e])) & 0xffffffffull);
                  const int r = a.oc_bigidx[vl];
                  o0 = a.oc_off[(size_t)b * a.oc_nrows + r];
                  o1 = a.oc_off[(size_t)(b + 1) * a.oc_nrows + r];
                }
                const Index len = o1 - o0;
                Index inc = len;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
                const Index total = (Index)__builtin_amdgcn_readlane((int)inc, kWave - 1);
                if (total == 0) continue;
                __builtin_amdgcn_wave_barrier();
                s_pull[wave].row[lane] = make_int2(inc - len, o0);
                __builtin_amdgcn_wave_barrier();
                for (Index at0 = 0; at0 < total; at0 += 4 * kWave) {
                  Index q[4], d[4];
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const Index at = at0 + j * kWave + lane;
                    q[j] = -1;
                    if (at < total) {
                      int r = 0;                           // the last entry whose first edge is <= at
#pragma unroll
                      for (int step = kWave / 2; step > 0; step >>= 1)
                        if (s_pull[wave].row[r + step].x <= at) r += step;
                      q[j] = s_pull[wave].row[r].y + (at - s_pull[wave].row[r].x);
                    }
                  }
#pragma unroll
                  for (int j = 0; j < 4; ++j) d[j] = q[j] >= 0 ? a.oind[q[j]] : -1;
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (d[j] >= 0) atomicOr(&s_ocw[(d[j] >> 5) - w0], 1u << (d[j] & 31));
                }
                __builtin_amdgcn_wave_barrier();
              }
              __syncthreads();
              for (int i = tid; i < nw; i += kPThreads) {
                const unsigned int acc = s_ocw[i];
                if (!acc) continue;
                unsigned int newb = acc & ~fresh(&a.V[w0 + i]);
                if (!newb) continue;
                newb &= ~atomicOr(&a.V[w0 + i], newb);
                if (newb) atomicOr(&a.Fn[w0 + i], newb);
              }
            }
          }
        }
        if (tid == 0) s_nmed = 0;
        __syncthreads();
        for (long long base = 0; base < local_words; base += gthreads) {
          // 64 consecutive words per wave (two cache lines per load), consecutive chunks to different workgroups
          const long long i = ((base / kWave) + (long long)wave * G + blockIdx.x) * kWave + lane;
          unsigned int w = (i < local_words) ? fresh(&Fcur[i]) : 0u;
          for (; w; w &= w - 1) {
            const Index vl = (Index)i * 32 + (__ffs((int)w) - 1);
            const Index s = a.optr[vl], e = a.optr[vl + 1];
            const Index d = e - s;
            if (d >= kPBigDeg) continue;
            if (d >= kPSmallDeg) {
              const int slot = atomicAdd(&s_nmed, 1);
              if (slot < kPMedCap) { s_med[slot] = vl; continue; }
            }
            for (Index p = s; p < e; ++p) part_push_visit(a.V, a.Fn, a.oind[p]);
          }
          __syncthreads();
          const int nm = s_nmed < kPMedCap ? s_nmed : kPMedCap;
          for (int q = wave; q < nm; q += kPWaves) {
            const Index vl = s_med[q];
            const Index e = a.optr[vl + 1];
            for (Index p = a.optr[vl] + lane; p < e; p += kWave) part_push_visit(a.V, a.Fn, a.oind[p]);
          }
          __syncthreads();
          if (tid == 0) s_nmed = 0;
          __syncthreads();
        }
      }
    } else {
      // ================= pull: the owned unvisited vertices' in-edges =================
      GRB_PART_PHASE();
      const unsigned int* vin = a.V;
      const Index* hint = a.hint;
      const Index nwaves = (Index)G * kPWaves;
      const unsigned long long lt_mask = (1ull << lane) - 1ull;
      PullLds& L = s_pull[wave];
      unsigned long long unused = 0;                       // the partitioned loop keeps no inspected-edge count
      // Few owned vertices left to discover (this rank's share of what has been reached, taken as even): the active
      // bits are numbered and taken 64 at a time, probed with agent-scope loads (no invalidate) -- as the one-launch
      // kernel does.  Ranks may decide differently: both walks find the same vertices.
      const bool sparse_act = a.n_in_local >= 0 &&
                              (a.n_in_local - (cy.reached * (long long)a.n_local) / (long long)(n > 0 ? n : 1)) * 8 < (long long)a.n_local;
      if (sparse_act) {
        const Index ngroups = (Index)((local_words + kSparseWords - 1) / kSparseWords);
        for (Index g = (Index)blockIdx.x * kPWaves + wave; g < ngroups; g += nwaves) {
          const Index wl = g * kSparseWords + lane;
          const bool has_word = lane < kSparseWords && wl < local_words;
          unsigned int act = 0u;
          if (has_word) act = ~(fresh(&vin[lo_w + wl]) | a.skip[wl]);
          if (__ballot(act != 0u) == 0ull) continue;
          if (lane < kSparseWords) L.fresh_bits[lane] = 0u;
          wave_for_each_bit(&L.bits, act, lane, [&](int wlane, int bit) {
            const bool on = wlane >= 0;
            const Index v = on ? (g * kSparseWords + wlane) * 32 + bit : 0;     // local id
            const Index hv = hint ? hint[v] : -1;
            const Index p = a.iptr[v], e = a.iptr[v + 1];
            bool found = false;
            if (hv >= 0) found = on && ((fresh(&vin[hv >> 5]) >> (hv & 31)) & 1u);
            const bool und = on && !found && p < e;
            const unsigned long long um = __ballot(und);
            if (um) {
              if (lane < 2) L.found[lane] = 0u;
              if (und) {
                const int slot = __popcll(um & lt_mask);
                L.row[slot] = make_int2(p, e);
                L.id[slot] = (unsigned short)lane;
              }
              __builtin_amdgcn_wave_barrier();
              pull_queue_run<true>(a.iind, a.innz, vin, L, lane, __popcll(um), unused);
              if (und && ((L.found[lane >> 5] >> (lane & 31)) & 1u)) found = true;
              __builtin_amdgcn_wave_barrier();
            }
            if (found) atomicOr(&L.fresh_bits[wlane], 1u << bit);
          });
          __builtin_amdgcn_wave_barrier();
          if (has_word) {
            const unsigned int nb = L.fresh_bits[lane];
            if (nb) publish(&a.Fn[lo_w + wl], nb);
          }
          __builtin_amdgcn_wave_barrier();
        }
      } else {
      // The visited bitmap was published before the barrier; the probes below are ordinary loads through L1,
      // so this workgroup drops what its L1 may still hold (bfs_persist.hip does the same).
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      const Index nchunks = (a.n_local + kWave - 1) / kWave;
      const Index nblocks = (nchunks + kPPullBlock - 1) / kPPullBlock;
      for (Index blk = (Index)blockIdx.x * kPWaves + wave; blk < nblocks; blk += nwaves) {
        // ---- stage 0: the block's 16 words; a lane's 8 vertices are vbase + 64 j (local ids)
        const Index wl = blk * (2 * kPPullBlock) + lane;
        const bool has_word = lane < 2 * kPPullBlock && wl < local_words;
        unsigned int vw = 0xffffffffu, inact = 0xffffffffu;
        if (has_word) { vw = vin[lo_w + wl]; inact = vw | a.skip[wl]; }
        unsigned int act = 0;
#pragma unroll
        for (int j = 0; j < kPPullBlock; ++j) {
          const unsigned int wj = __shfl(inact, 2 * j + (lane >> 5), kWave);
          act |= ((~wj >> (lane & 31)) & 1u) << j;
        }
        if (__ballot(act != 0u) == 0ull) continue;
        const Index vbase = blk * (kPPullBlock * kWave) + lane;
        unsigned int fnd = 0;
        // ---- stage 1: the hinted in-neighbour of every active vertex; the row pointers travel with it
        Index p[kPPullBlock], e[kPPullBlock];
        {
          Index hv[kPPullBlock];
#pragma unroll
          for (int j = 0; j < kPPullBlock; ++j) {
            const Index vj = ((act >> j) & 1u) ? vbase + kWave * j : 0;
            hv[j] = hint ? hint[vj] : 0;
            p[j] = a.iptr[vj];
            e[j] = a.iptr[vj + 1];
          }
          if (hint) {
#pragma unroll
            for (int j = 0; j < kPPullBlock; ++j) {
              const unsigned int on = ((act >> j) & 1u) && hv[j] >= 0 ? 1u : 0u;
              const unsigned int w = vin[on ? (hv[j] >> 5) : 0];
              fnd |= (on & (w >> (hv[j] & 31)) & 1u) << j;
            }
          }
        }
        unsigned int und = act & ~fnd;
#pragma unroll
        for (int j = 0; j < kPPullBlock; ++j)
          if (p[j] >= e[j]) und &= ~(1u << j);
        if (__ballot(und != 0u)) {
          // ---- the undecided rows, queued in lane order and taken dense (persist_common.hpp: pull_queue_run)
          if (lane < 2 * kPPullBlock) L.found[lane] = 0u;
          const int mine = __popc(und);
          int incl = mine;
incl = (int)wave_incl_scan_u32((unsigned)incl);
          const int T = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
          int at = incl - mine;
#pragma unroll
          for (int j = 0; j < kPPullBlock; ++j)
            if ((und >> j) & 1u) {
              L.row[at] = make_int2(p[j], e[j]);
              L.id[at] = (unsigned short)(j * kWave + lane);
              ++at;
            }
          __builtin_amdgcn_wave_barrier();
          pull_queue_run<false>(a.iind, a.innz, vin, L, lane, T, unused);
#pragma unroll
          for (int j = 0; j < kPPullBlock; ++j) fnd |= ((L.found[2 * j + (lane >> 5)] >> (lane & 31)) & 1u) << j;
          __builtin_amdgcn_wave_barrier();
        }
        // ---- output: the new bits of the owned words
        unsigned int nb = 0;
#pragma unroll
        for (int j = 0; j < kPPullBlock; ++j) {
          const unsigned long long fb = __ballot((fnd >> j) & 1u);
          if ((lane >> 1) == j) nb = (lane & 1) ? (unsigned int)(fb >> 32) : (unsigned int)(fb & 0xffffffffull);
        }
        if (has_word && nb) publish(&a.Fn[lo_w + wl], nb);
      }
      }
    }
    cy.last_dir = pull ? 1 : 0;
    cy.iter = iter + 1;
    // several levels in one launch (one rank only): this level's bits must be complete before they are applied
    if (step + 1 < a.nlevels && !grid_sync(bar, gen, false)) { give_up(); return; }
  }

  if (cy.done) {
    // ---- the depth vector of the owned vertices, written once and coalesced from the kept level bitmaps: level
    // L + 1 for the vertices of F[L]; 0 for everything never reached; levels beyond the kept ones were labelled when
    // found.  (V and every F are final: the ending launch passed the apply barrier.)
    GRB_PART_PHASE();
    int kept = cy.hit_cap ? cy.levels : cy.levels + 1;      // F[levels] of a cut-off traversal was never written
    if (kept > kPKeep) kept = kPKeep;
    // one owned bitmap word (32 vertices) per lane: the visited word and the word of every kept level in flight
    // together (agent-scope loads, no invalidate), the 32 labels as eight 16-byte stores (bfs_persist.hip)
    const unsigned int* V_own = a.V + lo_w;
    const bool label_aligned = (reinterpret_cast<unsigned long long>(a.label) & 15ull) == 0ull;
    for (long long wi = gtid; wi < local_words; wi += gthreads) {
      const unsigned int vis = fresh(&V_own[wi]);
      unsigned int f[kPKeep];
#pragma unroll
      for (int L0 = 0; L0 < kPKeep; L0 += 8) {
        if (L0 < kept) {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = L0 + u < kept ? fresh(&a.F[(size_t)(L0 + u) * local_words + wi]) : 0u;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = 0u;
        }
      }
      unsigned int pl[6] = {0u, 0u, 0u, 0u, 0u, 0u};      // bit b of plane q = bit q of the label of vertex 32 wi + b
#pragma unroll
      for (int L = 0; L < kPKeep; ++L) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (((L + 1) >> q) & 1) pl[q] |= f[L];
      }
      const long long v0 = wi * 32;
      // visited but in no kept bitmap: labelled when found (a level beyond the kept ones), or never assigned
      const unsigned int keepm = vis & ~(pl[0] | pl[1] | pl[2] | pl[3] | pl[4] | pl[5]);
      if (v0 + 32 <= (long long)a.n_local && keepm == 0u && label_aligned) {
        float4* out = reinterpret_cast<float4*>(a.label + v0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float x[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int b = q * 4 + t;
            unsigned int lab = 0u;
#pragma unroll
            for (int z = 0; z < 6; ++z) lab |= ((pl[z] >> b) & 1u) << z;
            x[t] = (float)lab;
          }
          out[q] = make_float4(x[0], x[1], x[2], x[3]);
        }
      } else {
        for (int b = 0; b < 32 && v0 + b < (long long)a.n_local; ++b) {
          unsigned int lab = 0u;
#pragma unroll
          for (int z = 0; z < 6; ++z) lab |= ((pl[z] >> b) & 1u) << z;
          if (!((keepm >> b) & 1u)) a.label[v0 + b] = (float)lab;
        }
      }
    }
  }
  if (gtid == 0) {
    st->carry[(k + 1) & 1] = cy;
    if (cy.done) *a.result = cy;                            // pinned host memory: read after the final synchronisation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    report(cy, false);
  }
}

// hint[v] = the in-neighbour of owned vertex v with the largest out-degree (global id), -1 for an empty row
__global__ __launch_bounds__(kBlock) void bfs_part_hint_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                               Index n_local, const int* __restrict__ deg,
                                                               Index* __restrict__ hint) {
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index v = (Index)blockIdx.x * kWavesPerBlock + wave_id(); v < n_local; v += nwaves) {
    Index wb = -1;
    int wd = -1;
    const Index re = ptr[v + 1];
    for (Index q = ptr[v] + lane; q < re; q += kWave) {
      const Index u = ind[q];
      const int d = deg[u];
      if (d > wd) { wd = d; wb = u; }
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const int od = __shfl_xor(wd, o, kWave);
      const Index ob = __shfl_xor(wb, o, kWave);
      if (od > wd || (od == wd && ob >= 0 && (wb < 0 || ob < wb))) { wd = od; wb = ob; }
    }
    if (lane == 0) hint[v] = wb;
  }
}

// grb_bfs_part_run_group's stand-in for the all-gather: every rank's send buffer into slot r of every rank's
// receive buffer, one launch (all ranks live on this device)
constexpr int kLoopMaxRanks = 16;
struct LoopbackPtrs {
  const unsigned int* send[kLoopMaxRanks];
  unsigned int* recv[kLoopMaxRanks];
};
__global__ __launch_bounds__(kBlock) void loopback_allgather_kernel(LoopbackPtrs p, int world, int nwords) {
  const long long total = (long long)world * nwords;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int r = (int)(i / nwords);
    const unsigned int x = p.send[r][i - (long long)r * nwords];
    for (int q = 0; q < world; ++q) p.recv[q][i] = x;
  }
}

}  // namespace grb

using namespace grb;

struct grb_part_s {
  int rank = 0, world = 1;
  Index n = 0, lo = 0, n_local = 0;
  int nwords = 0, local_words = 0;
  long long nnz = 0;
  grb_matrix A_out = nullptr, A_in = nullptr;
  const int* d_deg = nullptr;
  char* d_block = nullptr;            // [PartState | V | Fn | F[0] | F[1 .. kPKeep]]
  size_t st_bytes = 0, zero_bytes = 0;
  unsigned int* d_gathered = nullptr;   // world > 1: the all-gather's receive buffer (world x nwords)
  int2* d_big = nullptr;
  int big_cap = 0;
  Index* d_oc_bounds = nullptr;       // owner-computes push tables of this rank's out-edge shard (oc_tables_build)
  Index* d_oc_off = nullptr;
  int* d_oc_bigidx = nullptr;
  int oc_nb = 0, oc_nrows = 0;
  grb_bfs_level* h_rec = nullptr;       // pinned; d_rec is its device-side address
  grb_bfs_level* d_rec = nullptr;
  int rec_cap = 1 << 15;
  bool prezeroed = false;               // the zeroed part of d_block was cleared behind the previous traversal
  Index* d_hint = nullptr;
  long long n_in_local = -1;            // owned vertices with in-edges
  unsigned long long* h_mail = nullptr;   // pinned
  unsigned long long* d_mail = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

struct PartPtrs {
  PartState* st;
  unsigned int *V, *Fn, *F;
};
PartPtrs part_ptrs(grb_part p) {
  PartPtrs q;
  q.st = (PartState*)p->d_block;
  q.V = (unsigned int*)(p->d_block + p->st_bytes);
  q.Fn = q.V + p->nwords;
  q.F = q.Fn + p->nwords;
  return q;
}

// launches that have reported, and the launch that ended the traversal (-1: none yet; -2: a barrier gave up)
void mail_peek(grb_part p, int* launches_done, int* done_at) {
  const unsigned long long g = __atomic_load_n(p->h_mail, __ATOMIC_ACQUIRE);
  *launches_done = (int)(g & 0xffffffffull);
  const unsigned int hi = (unsigned int)(g >> 32);
  *done_at = hi == 0xffffffffu ? -2 : (int)hi - 1;
}

// One traversal over `nranks` rank contexts driven in lock-step on this device.  nranks == 1 is a real run
// (collectives through the library communicator when its world is > 1); nranks > 1 is the single-device
// stand-in the tests use: every rank of a world of `nranks` lives on this GPU and the all-gather is a set of
// device copies on the same stream.
grb_info part_bfs_run(grb_part* ps, int nranks, grb_index source, int mode, float switchpoint, float edgeswitch,
                      int max_niter, int levels_per_launch, float* const* labels, grb_part_bfs_result* res,
                      grb_bfs_level* levels_out, int max_levels) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  grb_part p0 = ps[0];
  const bool loopback = nranks > 1;
  const int world = p0->world;
  if (loopback && (world != nranks || nranks > kLoopMaxRanks)) return GRB_INVALID_VALUE;
  LoopbackPtrs loop_ptrs = {};
  if (source < 0 || source >= p0->n) return GRB_INVALID_INDEX;
  if (mode != GRB_PUSHPULL && mode != GRB_PUSHONLY && mode != GRB_PULLONLY) return GRB_INVALID_VALUE;
  if (!loopback && world > 1) {
    int r = -1, w = 0;
    grb_comm_info(&r, &w);
    if (w != world || r != p0->rank) return GRB_UNINITIALIZED_OBJECT;
  }
  if (world > 1 || levels_per_launch < 1) levels_per_launch = 1;
  GRB_TRY(bfs_lanes_fence(ctx().stream));   // a whole-device grid must not meet a BFS lane's narrower one half-way (bfs_persist.hip)
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, bfs_part_level_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_PANIC;
  }
  const int G = c.num_cu;

  std::vector<PartArgs> args(nranks);
  for (int r = 0; r < nranks; ++r) {
    grb_part p = ps[r];
    if (!labels[r] && p->n_local > 0) return GRB_NULL_POINTER;
    PartPtrs q = part_ptrs(p);
    PartArgs& a = args[r];
    a.optr = p->A_out->csr.ptr; a.oind = p->A_out->csr.ind;
    a.iptr = p->A_in->csr.ptr;  a.iind = p->A_in->csr.ind;
    a.innz = p->A_in->nvals;
    a.skip = p->A_in->d_empty_csr_rows;
    a.n_in_local = p->n_in_local;
    a.hint = p->d_hint;
    a.deg = p->d_deg;
    a.n = p->n; a.lo = p->lo; a.n_local = p->n_local;
    a.nnz = p->nnz;
    a.source = source;
    a.mode = mode;
    a.switchpoint = switchpoint;
    a.edgeswitch = edgeswitch;
    a.max_niter = max_niter;
    a.world = world;
    a.parts = world > 1 ? p->d_gathered : q.Fn;
    a.Fn = q.Fn; a.V = q.V; a.F = q.F;
    a.label = labels[r];
    a.big_list = p->d_big; a.big_cap = p->big_cap;
    a.oc_bounds = p->d_oc_bounds; a.oc_off = p->d_oc_off; a.oc_bigidx = p->d_oc_bigidx;
    a.oc_nb = p->oc_nb; a.oc_nrows = p->oc_nrows;
    a.st = q.st;
    a.rec = p->d_rec; a.rec_cap = p->rec_cap;
    a.mail = p->d_mail;
    a.result = reinterpret_cast<PartCarry*>(p->d_mail + 8);
    __atomic_store_n(p->h_mail, 0ull, __ATOMIC_RELEASE);   // nothing of this device is in flight: every run ends synchronised
    a.launch = 0;
    a.nlevels = levels_per_launch;
    // zeroed: the state, V, Fn and F[0]; the other level bitmaps are written in full before they are read
    if (!p->prezeroed) GRB_HIP_TRY(hipMemsetAsync(p->d_block, 0, p->zero_bytes, s));
    p->prezeroed = false;
    if (loopback) { loop_ptrs.send[r] = q.Fn; loop_ptrs.recv[r] = p->d_gathered; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  GRB_HIP_TRY(hipEventRecord(p0->ev0, s));
  int k = 0;
  const size_t bm_bytes = 4 * (size_t)p0->nwords;
  const bool one_shot = world == 1 && levels_per_launch > max_niter;   // the first launch runs every level
  for (;;) {
    // Launch k goes out when launch k - 2 has reported and had not ended the traversal: the device always has
    // launch k - 1 queued behind it.  The rule reads only what launch k - 2 computed -- the same on every rank --
    // so every rank enqueues the same number of launches and collectives: (the ending launch's index) + 2.
    if (k >= 2) {
      unsigned spins = 0;
      bool synced = false;
      int ld = 0, done_at = -1;
      for (;;) {
        mail_peek(p0, &ld, &done_at);
        if (ld >= k - 1) break;
        if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
          if (synced) return GRB_PANIC;
          GRB_HIP_TRY(hipStreamSynchronize(s));
          synced = true;
        }
      }
      if (done_at == -2) return GRB_PANIC;
      if (done_at >= 0 && done_at <= k - 2) break;
    }
    for (int r = 0; r < nranks; ++r) {
      args[r].launch = k;
      hipLaunchKernelGGL(bfs_part_level_kernel, dim3(G), dim3(kPThreads), 0, s, args[r]);
      GRB_HIP_TRY(hipGetLastError());
    }
    if (loopback) {
      hipLaunchKernelGGL(loopback_allgather_kernel, dim3(stream_grid((long long)nranks * p0->nwords)), dim3(kBlock), 0, s,
                         loop_ptrs, nranks, p0->nwords);
      GRB_HIP_TRY(hipGetLastError());
    } else if (world > 1) {
      GRB_TRY(grb_comm_allgather(args[0].Fn, p0->d_gathered, bm_bytes));
      GRB_TRY(grb_comm_wait());
    }
    ++k;
    if (one_shot) break;
    if (k > (1 << 28)) return GRB_PANIC;
  }
  // k launches went out; the one that ended the traversal wrote the labels and left the scalars in pinned memory
  GRB_HIP_TRY(hipEventRecord(p0->ev1, s));
  GRB_HIP_TRY(hipEventSynchronize(p0->ev1));
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, p0->ev0, p0->ev1));
  for (int r = 0; r < nranks; ++r) {
    grb_part p = ps[r];
    int ld = 0, done_at = -1;
    mail_peek(p, &ld, &done_at);
    PartCarry cy;
    memcpy(&cy, p->h_mail + 8, sizeof(cy));
    if (done_at < 0 || !cy.done) return GRB_PANIC;          // -2: a grid barrier gave up
    if (res) {
      res[r].levels = cy.levels;
      res[r].launches = k;
      res[r].hit_cap = cy.hit_cap;
      res[r].edges_traversed = (int64_t)cy.edges_cum;
      res[r].reached = (int64_t)cy.reached;
      res[r].ms = ms;
    }
    if (r == 0 && levels_out && max_levels > 0) {
      const int cnt = cy.levels < max_levels ? (cy.levels < p->rec_cap ? cy.levels : p->rec_cap) : max_levels;
      if (cnt > 0) memcpy(levels_out, p->h_rec, sizeof(grb_bfs_level) * (size_t)cnt);
    }
    // for the next traversal, queued now and off its critical path: the state, V, Fn and F[0] clear again
    GRB_HIP_TRY(hipMemsetAsync(p->d_block, 0, p->zero_bytes, s));
    p->prezeroed = true;
  }
  return GRB_SUCCESS;
}

}  // namespace

extern "C" {

grb_info grb_part_new(grb_part* out, int rank, int world, grb_index n_global, grb_index lo, grb_matrix A_out,
                      grb_matrix A_in, const int32_t* d_deg_full, int64_t nnz_global) { GRB_API_ENTER();
  if (!out) return GRB_NULL_POINTER;
  if (!A_out || !A_out->built || !d_deg_full) return GRB_UNINITIALIZED_OBJECT;
  if (!A_in) A_in = A_out;
  if (!A_in->built) return GRB_UNINITIALIZED_OBJECT;
  if (world < 1 || rank < 0 || rank >= world || lo < 0) return GRB_INVALID_VALUE;
  if (lo % 64 != 0 && A_out->nrows > 0) return GRB_INVALID_VALUE;      // an empty rank may sit at lo == n
  if (A_out->ncols != n_global || A_in->ncols != n_global || A_in->nrows != A_out->nrows) return GRB_DIMENSION_MISMATCH;
  if (lo + A_out->nrows > n_global) return GRB_INVALID_VALUE;
  GRB_TRY(ctx_init());
  Context& c = ctx();
  hipStream_t s = c.stream;
  grb_part p = new grb_part_s();
  p->rank = rank; p->world = world;
  p->n = n_global; p->lo = lo; p->n_local = A_out->nrows;
  p->nwords = 2 * ceil_div(n_global, 64);
  p->local_words = 2 * ceil_div(p->n_local, 64);
  p->nnz = nnz_global;
  p->A_out = A_out; p->A_in = A_in;
  p->d_deg = d_deg_full;
  p->st_bytes = (sizeof(PartState) + 255) & ~(size_t)255;
  p->zero_bytes = p->st_bytes + 4 * ((size_t)2 * p->nwords + (size_t)p->local_words);
  const size_t block = p->st_bytes + 4 * ((size_t)2 * p->nwords + (size_t)(kPKeep + 1) * (size_t)(p->local_words > 0 ? p->local_words : 1));
  auto fail = [&](grb_info i) { grb_part_free(p); return i; };
  if (hipMalloc((void**)&p->d_block, block) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  if (world > 1 && hipMalloc((void**)&p->d_gathered, 4 * (size_t)world * (size_t)p->nwords) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  p->big_cap = (int)(A_out->nvals / kPBigDeg) + 2;
  if (hipMalloc((void**)&p->d_big, sizeof(int2) * (size_t)p->big_cap) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  {
    // GRB_PART_OC=0: the big vertices' edges with atomics, as before round 3's owner-computes push
    static const bool oc_ok = [] { const char* e = getenv("GRB_PART_OC"); return !e || atoi(e) != 0; }();
    if (oc_ok && A_out->nvals > 0 && (Index)A_out->h_csr_ptr.size() == A_out->nrows + 1) {
      const grb_info oi = oc_tables_build(A_out->csr.ptr, A_out->csr.ind, A_out->h_csr_ptr, A_out->nrows, (Index)n_global, c.num_cu,
                                          &p->d_oc_bounds, &p->d_oc_off, &p->d_oc_bigidx, &p->oc_nb, &p->oc_nrows);
      if (oi != GRB_SUCCESS) return fail(oi);
    }
  }
  if (hipHostMalloc((void**)&p->h_rec, sizeof(grb_bfs_level) * (size_t)p->rec_cap, hipHostMallocMapped) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  if (hipHostGetDevicePointer((void**)&p->d_rec, p->h_rec, 0) != hipSuccess) return fail(GRB_PANIC);
  if (hipHostMalloc((void**)&p->h_mail, 256, hipHostMallocMapped) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  memset(p->h_mail, 0, 256);
  static_assert(sizeof(PartCarry) <= 256 - 64, "the result record follows the mail word in pinned memory");
  if (hipHostGetDevicePointer((void**)&p->d_mail, p->h_mail, 0) != hipSuccess) return fail(GRB_PANIC);
  if (hipEventCreate(&p->ev0) != hipSuccess || hipEventCreate(&p->ev1) != hipSuccess) return fail(GRB_PANIC);
  if (ensure_empty_rows(&A_in->d_empty_csr_rows, A_in->csr, s) != GRB_SUCCESS) return fail(GRB_PANIC);
  if (p->n_local > 0) {
    if (hipMalloc((void**)&p->d_hint, 4 * (size_t)p->n_local) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
    hipLaunchKernelGGL(bfs_part_hint_kernel, dim3(stream_grid((long long)p->n_local * kWave, kBlock)), dim3(kBlock), 0, s,
                       A_in->csr.ptr, A_in->csr.ind, p->n_local, d_deg_full, p->d_hint);
    if (hipGetLastError() != hipSuccess) return fail(GRB_PANIC);
  }
  if (hipStreamSynchronize(s) != hipSuccess) return fail(GRB_PANIC);
  {                                     // owned vertices with in-edges: the complement of the skip bitmap (padding bits are set)
    std::vector<unsigned int> h_skip((size_t)(p->local_words > 0 ? p->local_words : 1), 0xffffffffu);
    if (p->local_words > 0 &&
        hipMemcpy(h_skip.data(), A_in->d_empty_csr_rows, 4 * (size_t)p->local_words, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(GRB_PANIC);
    long long empty = 0;
    for (int i = 0; i < p->local_words; ++i) empty += __builtin_popcount(h_skip[(size_t)i]);
    p->n_in_local = (long long)p->local_words * 32 - empty;
  }
  *out = p;
  return GRB_SUCCESS;
}

grb_info grb_part_free(grb_part p) { GRB_API_ENTER();
  if (!p) return GRB_SUCCESS;
  (void)hipStreamSynchronize(ctx().stream);
  if (p->d_block) (void)hipFree(p->d_block);
  if (p->d_gathered) (void)hipFree(p->d_gathered);
  if (p->d_big) (void)hipFree(p->d_big);
  if (p->d_oc_bounds) (void)hipFree(p->d_oc_bounds);
  if (p->d_oc_off) (void)hipFree(p->d_oc_off);
  if (p->d_oc_bigidx) (void)hipFree(p->d_oc_bigidx);
  if (p->h_rec) (void)hipHostFree(p->h_rec);
  if (p->d_hint) (void)hipFree(p->d_hint);
  if (p->h_mail) (void)hipHostFree(p->h_mail);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  delete p;
  return GRB_SUCCESS;
}

grb_info grb_bfs_part_run(grb_part p, grb_index source, int mxvmode, float switchpoint, float edgeswitch, int max_niter,
                          int levels_per_launch, float* d_label_local, grb_part_bfs_result* result,
                          grb_bfs_level* levels_out, int max_levels) { GRB_API_ENTER();
  if (!p) return GRB_UNINITIALIZED_OBJECT;
  float* labels[1] = {d_label_local};
  return part_bfs_run(&p, 1, source, mxvmode, switchpoint, edgeswitch, max_niter, levels_per_launch, labels, result,
                      levels_out, max_levels);
}

grb_info grb_bfs_part_run_group(grb_part* parts, int nranks, grb_index source, int mxvmode, float switchpoint,
                                float edgeswitch, int max_niter, float* const* d_labels, grb_part_bfs_result* results,
                                grb_bfs_level* levels_out, int max_levels) { GRB_API_ENTER();
  if (!parts || !d_labels || nranks < 1) return GRB_NULL_POINTER;
  for (int r = 0; r < nranks; ++r)
    if (!parts[r] || parts[r]->rank != r) return GRB_INVALID_VALUE;
  return part_bfs_run(parts, nranks, source, mxvmode, switchpoint, edgeswitch, max_niter, 1, d_labels, results, levels_out,
                      max_levels);
}

}  // extern "C"
