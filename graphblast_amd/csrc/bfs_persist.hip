// bfs_persist.hip -- the whole direction-optimised BFS as ONE persistent launch.
//
// The level loop of bfs_fused.hip costs five to seven dependent kernel boundaries and one
// host round trip per level; on RMAT-22 (six levels, 0.2 ms of expansion work) that is half
// of the traversal.  Here one co-resident grid (one 1024-thread workgroup per CU) runs every
// level: the direction decision of vxm/convert (operations.hpp:131-140, vector.hpp:291-323)
// is evaluated redundantly by every workgroup from the level totals, levels are separated by
// an XCD-hierarchical grid barrier (per-XCD arrival counters -> top counter -> generation
// word; release before arriving, acquire after leaving), and the host sees one record at the
// end.  Same labels, same per-level trace as the fused loop.
//
//   visited V[2]   bitmaps; pull reads V[cur] and writes V[cur^1] (= old | found), push sets
//                  bits in V[cur] with atomicOr
//   frontier F[3]  bitmap of the vertices the previous level discovered, three buffers in
//                  rotation (read / being written / being cleared)
//   push           frontier vertices are classified by out-degree: < 16 expanded by the lane
//                  that found them, 16..511 by a wave (workgroup-local LDS list), >= 512 cut
//                  into 1024-edge entries of a global list that whole workgroups consume.  The
//                  number of such vertices is part of the previous level's totals, so the
//                  extra listing pass (and its barrier) only runs when there are any.
//   pull           as bfs_pull_kernel (hint probe, serial probes, wave-cooperative finish)
//   totals         every workgroup writes {discovered, their out-degree sum, inspected, big}
//                  into its slot; after the barrier every workgroup sums all slots
#include "bfs_kernels.hpp"
#include "persist_common.hpp"
#include <chrono>

namespace grb {

#ifndef GRB_PULL_BLOCK
#define GRB_PULL_BLOCK 8
#endif
#ifndef GRB_BFS_SPARSE_PULL
#define GRB_BFS_SPARSE_PULL 1
#endif
#ifndef GRB_BFS_LABEL_COAL
#define GRB_BFS_LABEL_COAL 1
#endif
#ifndef GRB_BFS_SPARSE_DIV
#define GRB_BFS_SPARSE_DIV 4
#endif
#ifndef GRB_BFS_SYM
#define GRB_BFS_SYM 1
#endif
#ifndef GRB_BFS_RELAUNDER
#define GRB_BFS_RELAUNDER 1
#endif
#ifndef GRB_BFS_SPARSE_FRESH
#define GRB_BFS_SPARSE_FRESH 1
#endif
// the big-vertex list: one global reservation per WORKGROUP and pass (waves reserve inside it through LDS) instead of one
// per wave -- a heavy level's 3 000 appends to one address cost 15 us at the listing barrier
#ifndef GRB_BFS_LIST_WG
#define GRB_BFS_LIST_WG 1
#endif
#ifndef GRB_BFS_DPP_TOTALS
#define GRB_BFS_DPP_TOTALS 1       // the level totals' wave sums by DPP adds on 32-bit values (0: six 64-bit shuffle steps each)
#endif
#ifndef GRB_BFS_SCAN_CHUNKS
#define GRB_BFS_SCAN_CHUNKS 1      // push levels read the frontier bitmap 64 consecutive words per wave (0: one word per lane, stride G)
#endif
#ifndef GRB_BFS_PULL_DYN
#define GRB_BFS_PULL_DYN 1         // dense pull: a workgroup's blocks handed out to its waves dynamically (0: two fixed blocks per wave)
#endif
// the 128-thread instance (launches of five to twelve traversals): waves per SIMD it is built for, and the shape of its
// pull queue (rows per lane and round / entries per lane and step) -- tools/bfs_co_bench.py, docs/experiments.md R6.1
#ifndef GRB_CO_LEAN_WPE
#define GRB_CO_LEAN_WPE 6
#endif
#ifndef GRB_CO_LEAN_FROM
#define GRB_CO_LEAN_FROM 128       // workgroup widths up to this one are built that way
#endif
#ifndef GRB_CO_LEAN_R
#define GRB_CO_LEAN_R 1
#endif
#ifndef GRB_CO_LEAN_D
#define GRB_CO_LEAN_D 1
#endif
#ifndef GRB_BFS_FINE_TRACE
#define GRB_BFS_FINE_TRACE 0       // 1: the level barrier stamped step by step (tools/bfs_trace.py; measurement builds only)
#endif
// (the barrier's own variants -- the totals as the barrier, a flat poll of the eight group counters, a generation word per
// group -- measured the same or slower: docs/experiments.md A.1)
#define GRB_BFS_GRID_SYNC(bar, gen) grid_sync(bar, gen, false)
#ifndef GRB_BFS_SMALL_DEG
#define GRB_BFS_SMALL_DEG 8
#endif
#ifndef GRB_BFS_BIG_DEG
#define GRB_BFS_BIG_DEG 256
#endif
constexpr int kSmallDeg = GRB_BFS_SMALL_DEG;  // below: expanded inline by the discovering lane
constexpr int kBigDeg = GRB_BFS_BIG_DEG;      // from here: split into kBigChunk-edge entries for workgroups
constexpr int kBigChunk = 1024;
constexpr int kPullBlock = GRB_PULL_BLOCK;    // chunks (of 64 vertices) one wave carries through the pull stages together
static_assert(kPullBlock * kWave <= kPullQueue, "a wave queues at most every vertex of its block");
typedef float LabelQuad __attribute__((ext_vector_type(4)));
constexpr int kKeep = 32;         // levels whose discovered-bitmaps are kept for the final label pass

struct PersistState {               // zeroed by the host before every launch
  GridBarrier bar;
  unsigned big_count[2][32];
  unsigned long long acc[3][8][16]; // level totals, one line per (set, XCD group): found, deg, inspected, big
  unsigned next_idx[32];            // launches of several traversals: the one this sub-grid runs next (written before the end barrier)
};

// What every traversal of a launch shares: the matrix, the rules, the tables.
struct PersistArgs {
  const Index *optr, *oind;         // out-edges (CSR), walked by push
  const Index *iptr, *iind;         // in-edges (CSC), walked by pull
  const unsigned int* skip;         // vertices without in-edges
  const Index* hint;                // best in-neighbour per vertex (nullptr in accounting runs)
  Index n;
  long long nnz;
  long long n_in;                   // vertices with at least one in-edge (the only ones a pull level can discover); < 0 unknown
  int out_is_in;                    // optr and iptr hold the same numbers (a structurally symmetric matrix): a row's
                                    // out-degree is the difference of the in-edge pointers a pull level already holds
  int mode;
  float switchpoint, edgeswitch;
  int max_niter;
  int count_inspected;
  // State blocks [state | V0 | F0 .. F(kKeep + 2)] are used in rotation by the traversals of this grid: a traversal runs on
  // a clean one and clears -- spread over the whole grid, in front of its latency-bound first level -- one that an
  // earlier traversal dirtied (rot[1 + block] = that one's level count: how much of the block it dirtied), so
  // consecutive traversals need neither a memset nor a clean-up launch between them.
  //   one traversal per launch (T = 1024): two blocks, the host alternates them (TravArgs::block / ::clean, at fixed
  //     places of the kernarg segment: the kernel re-reads them where it needs them instead of holding registers);
  //     plain stores clear, the launch boundary publishes them
  //   several per launch: three blocks; traversal number t of the sub-grid (rot[0], kept on the device: which
  //     sub-grid runs how many traversals of a launch is decided on the device) runs on block t % 3 and clears block
  //     (t + 1) % 3, which traversal t - 2 used, with write-through stores (the next traversal may be in this very
  //     launch and its readers on other XCDs).  Nobody can still be inside traversal t - 2 -- its barrier counters
  //     live in that block -- when t begins: t - 1 had barriers of its own.
  //   V0      visited bitmap, the one a traversal starts on (arrives zeroed); v1 is its pull levels' other buffer
  //   F[L]    bitmap of the vertices discovered by level L (F[0] = the source), kept for L < kKeep: labels are
  //           NOT written while traversing -- one coalesced pass at the end turns the kept bitmaps into the depth vector
  //           (a scattered 4-byte label store costs a 32-byte memory write: 131 MB per traversal of RMAT-22 against
  //           17 MB of labels).  Levels >= kKeep (long-diameter graphs, tiny frontiers) rotate through three more
  //           buffers and label directly.
  unsigned long long block_bytes, st_bytes;
  int big_cap;
  // owner-computes push for heavy sparse frontiers (oc_off == nullptr: off).  The vertices are cut into oc_nb ranges
  // [oc_bounds[b], oc_bounds[b + 1]) of about equal in-edge mass (word-aligned, at most kOcWords words); a row of
  // >= kBigDeg entries is sorted, so its entries inside a range are one piece, oc_off[b * oc_nrows + r] ..
  // oc_off[(b + 1) * oc_nrows + r) for big row number r = oc_bigidx[v] (precomputed once per matrix).  The
  // workgroup that owns a range ORs the pieces of all big frontier vertices into its slice of the visited bitmap
  // in LDS: no edge is written anywhere, no global atomic per edge.
  const Index* oc_bounds;
  const Index* oc_off;
  const int* oc_bigidx;
  int oc_nb, oc_nrows;
  unsigned long long oc_min_edges;
  int rec_cap;
  float ticks_to_ms;
  unsigned long long* trace;        // optional (GRB_BFS_TRACE): wall-clock stamps of workgroup 0
};
// ... what is a (sub-)grid's own: the buffers its traversals run on, one after the other
struct GridArgs {
  char* blocks;                     // the state blocks
  unsigned int* v1;
  unsigned int* rot;                // device words: [0] traversals this grid has run, [1 + b] level count of block b's last one
                                    // (one traversal per launch: [0] = the last traversal's level count)
  int2* big_list;
  grb_bfs_level* rec;
};
// ... and what is a traversal's own
struct TravArgs {
  float* label;
  unsigned long long* mail;         // pinned host granules {value, seq}: the traversal's record
  char* block;                      // one traversal per launch (the host keeps the rotation): the block it runs on ...
  char* clean;                      // ... and the one it clears
  Index source;
  int seq;
};


struct LevelCounters {
  unsigned long long found = 0, deg = 0, inspected = 0, big = 0;
};

__device__ inline int fbuf(int level) { return level < kKeep ? level : kKeep + (level - kKeep) % 3; }

// a vertex was discovered: account for it (and label it at once beyond the kept levels: new_label > 0)
// (A = PersistArgs wherever it lives: the one-traversal kernel's by-value parameter, or one entry of the co-scheduled
// kernel's argument table read through the kernarg segment pointer)
template <typename A>
__device__ inline void discovered(const A& a, float* label, Index v, float new_label, LevelCounters& c) {
  if (new_label > 0.f) label[v] = new_label;
  const Index d = a.optr[v + 1] - a.optr[v];
  ++c.found;
  c.deg += (unsigned long long)d;
  if (d >= kBigDeg) ++c.big;
}

template <typename A>
__device__ inline void push_visit(const A& a, float* label, unsigned int* V, unsigned int* Fn, Index dst,
                                  float new_label, LevelCounters& c) {
  const unsigned int bit = 1u << (dst & 31);
  if (fresh(&V[dst >> 5]) & bit) return;
  const unsigned int old = atomicOr(&V[dst >> 5], bit);
  if (old & bit) return;
  atomicOr(&Fn[dst >> 5], bit);
  discovered(a, label, dst, new_label, c);
}

// One traversal on a grid of G workgroups of T threads; `bid` is the workgroup's number inside that grid.  The
// one-traversal kernel is this with T = 1024 on the launch's whole grid; the co-scheduled kernel (below) runs several
// of these side by side in one launch, T = 512 or 256, each on its own sub-grid of G = CUs workgroups -- a CU then holds
// one workgroup of every traversal, and the hardware's wave scheduler fills one traversal's barriers and latency chains
// with the other's work.
//
// LDS: the pull levels' per-wave row queues and the owner-computes push's slice of the visited bitmap are never live
// at the same time (a level is one or the other), so they share their bytes; at T = 256 four workgroups fit a CU, at
// T = 128 (a narrower slice of the bitmap per range) eight.
template <int T>
struct PersistLds {
  static constexpr int W = T / kWave;
  static constexpr bool kLean = T <= GRB_CO_LEAN_FROM;                     // built for GRB_CO_LEAN_WPE waves per SIMD
  static constexpr int kPB = !kLean ? kPullBlock : (kPullBlock < 4 ? kPullBlock : 4);   // chunks a wave carries through a dense pull
  typedef PullLdsT<kPB * kWave> Pull;
  static constexpr int kOcW = T >= 256 ? kOcWords : kOcWords / 4;  // (up to twelve workgroups per CU share its LDS: tables cut narrower)
  struct OcView { int2 row[W][kWave]; unsigned int ocw[kOcW]; };
  union U { Pull pull[W]; OcView oc; };
};

// Returns the number (in the launch's table) of the traversal this grid runs next -- >= the table's size: none -- or -1
// when a barrier gave up.  trot = how many traversals the grid has run before this one (PersistArgs::rot).  chained:
// the launch carries more traversals than grids; a grid that finishes one draws the next from a counter (*ctr, which
// stood at ctr_base when the launch began; the first n_grids traversals are dealt statically), so that a launch ends
// when the work does, not when the grid with the longest traversals does.
template <int T, typename AP, typename GP, typename TP>
__device__ __forceinline__ int bfs_persistent_body(AP ap, GP gp, TP tp, const int bid, const int G, const unsigned trot, const bool chained,
                                                   unsigned int* ctr, const unsigned ctr_base, const int n_grids) {
  const auto& a = *ap;
  constexpr bool kHostRot = T == kPThreads;                        // one traversal per launch: the host keeps the rotation
  // A sub-grid's and a traversal's own arguments sit at a run-time place of the kernarg segment.  What is read from there
  // cannot be re-read for free where it is needed (the shared block's fields, at fixed places, can), so the compiler
  // would read all of it once and hold it in registers for the whole kernel -- which has none to spare.  Every phase
  // therefore reads what it needs through a pointer the optimiser cannot see through (one scalar load per phase).
  auto ph = [](auto q) { if constexpr (!kHostRot) asm volatile("" : "+s"(q)); return q; };
  constexpr int W = T / kWave;
  // The 128-thread instance is built for SIX waves per SIMD (80 registers: up to twelve workgroups = 24 waves on a CU
  // instead of 16): the pull levels' row queue is where the kernel is fattest (docs/experiments.md R6.1: without it the
  // kernel fits 74 registers), so there a wave carries 4 chunks instead of 8 and the queue takes one row per lane and round
  // and one entry per lane and step instead of 4 and 4 -- and its LDS halves with the block.
  constexpr int kPB = PersistLds<T>::kPB;
  constexpr int kQR = !PersistLds<T>::kLean ? kPullR : GRB_CO_LEAN_R, kQD = !PersistLds<T>::kLean ? 4 : GRB_CO_LEAN_D;
  constexpr int kMed = T >= 512 ? 4 * T : T >= 256 ? 512 : 256;                     // LDS list of medium vertices per workgroup pass
  __shared__ unsigned long long s_red[W][4];
  __shared__ unsigned long long s_tot[4];
#if GRB_BFS_PULL_DYN
  __shared__ int s_pull_next;                                      // dense pull: the next block of this workgroup's share
#endif
  __shared__ int s_lcnt[2];                                        // big-vertex listing: this workgroup's entries of a pass
  __shared__ unsigned s_lbase;                                     // ... and where its block starts in the global list
  __shared__ Index s_med[kMed];
  __shared__ int s_nmed;
  __shared__ typename PersistLds<T>::U s_u;                        // pull: one row queue per wave | heavy push: the range's slice
  int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  long long gtid = (long long)bid * T + tid;
  // The phases of a level are long and disjoint; left alone, the compiler computes every lane-derived index, mask and
  // LDS address of ALL of them once before the level loop and keeps them in registers for the whole kernel (the
  // kernel sits at its 128-register limit, so each such value is a spill somewhere else).  Each phase therefore
  // starts from a thread id the optimiser cannot see through: what it derives from it lives inside that phase only.
#if GRB_BFS_RELAUNDER
#define GRB_PHASE_START()                                         \
  do {                                                            \
    asm volatile("" : "+v"(tid));                                 \
    lane = tid & (kWave - 1);                                     \
    wave = tid >> 6;                                              \
    gtid = (long long)bid * T + tid;               \
  } while (0)
#else
#define GRB_PHASE_START() do { } while (0)
#endif
  const long long gthreads = (long long)G * T;
  const Index n = a.n;
  const int nwords = 2 * ((n + 63) / 64);
  const unsigned blk = kHostRot ? 0u : trot % 3u;
  // (the block's address is worked out again wherever it is needed -- a scalar load and a multiply-add per use, a few per
  // level -- instead of living in registers)
  auto pblock = [&]() -> char* {
    if constexpr (kHostRot) return tp->block;
    else return ph(gp)->blocks + (unsigned long long)blk * a.block_bytes;
  };
#define st (reinterpret_cast<PersistState*>(pblock()))
#define V0 (reinterpret_cast<unsigned int*>(pblock() + a.st_bytes))
  auto Vp = [&](int which) -> unsigned int* { return which ? ph(gp)->v1 : V0; };
  auto Fp = [&](int level) -> unsigned int* { return V0 + (size_t)(1 + fbuf(level)) * (size_t)nwords; };
  unsigned gen = 0;
  const unsigned long long t_start = wall_clock64();
  int ntrace = 0;
  auto stamp = [&]() { if (a.trace && gtid == 0 && ntrace < 255) a.trace[1 + ntrace++] = wall_clock64() - t_start; };

  // ---- the block a later traversal will run on: what an earlier one dirtied there (its state, V0, the level bitmaps it
  // wrote; every buffer when it went past the kept levels)
  if constexpr (kHostRot) {
    const unsigned int lv = *ph(gp)->rot;
    const unsigned int used = lv + 2u >= (unsigned int)kKeep ? (unsigned int)kKeep + 3u : lv + 2u;
    const long long n16 = (long long)((a.st_bytes + 4ull * (1ull + used) * (unsigned long long)nwords + 15ull) / 16ull);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    uint4* clean = reinterpret_cast<uint4*>(tp->clean);
    for (long long i = gtid; i < n16; i += gthreads) clean[i] = z;
  } else {
    const unsigned nb = (trot + 1u) % 3u;
    const unsigned int lv = fresh(&ph(gp)->rot[1 + nb]);
    const unsigned int used = lv + 2u >= (unsigned int)kKeep ? (unsigned int)kKeep + 3u : lv + 2u;
    const long long n8 = (long long)((a.st_bytes + 4ull * (1ull + used) * (unsigned long long)nwords + 7ull) / 8ull);
    unsigned long long* clean = reinterpret_cast<unsigned long long*>(ph(gp)->blocks + (unsigned long long)nb * a.block_bytes);
    for (long long i = gtid; i < n8; i += gthreads) publish(&clean[i], 0ull);
  }

  // ---- the source.  The bitmaps arrive zeroed; unreached labels are written at the very end,
  // so the first level starts without a barrier (unless it is a pull, which must see the bit).
  Index src_deg;
  {
    const Index source = ph(tp)->source;
    src_deg = a.optr[source + 1] - a.optr[source];
    if (gtid == 0) {
      atomicOr(&Vp(0)[source >> 5], 1u << (source & 31));
      atomicOr(&Fp(0)[source >> 5], 1u << (source & 31));
    }
  }
  if (a.mode == GRB_PULLONLY && !grid_sync_at(&st->bar, gen, bid, G, false)) return -1;

  // ---- level loop (all scalars below are identical in every workgroup)
  Index nf = 1;
  unsigned long long mf = (unsigned long long)src_deg;
  unsigned long long nbig = 0;
  long long reached = 1;
  unsigned long long edges_cum = mf;
  bool f1_dense = (a.mode == GRB_PULLONLY);
  float ratio_f1 = 0.f, ratio_f2 = 0.f;
  int cur = 0, levels = 0, last_dir = 0;
  int iter = 1;
  for (; iter <= a.max_niter; ++iter) {
    const unsigned long long t_level = wall_clock64();
    if (a.mode == GRB_PUSHPULL) {
      const float ratio = (float)nf / (float)n;
      if (!f1_dense) {
        if (ratio > a.switchpoint && ratio > ratio_f1) f1_dense = true; else ratio_f1 = ratio;
      } else {
        if (ratio <= a.switchpoint && ratio < ratio_f1) f1_dense = false; else ratio_f1 = ratio;
      }
      if (!f1_dense && a.edgeswitch > 0.f && nf >= 32 &&
          (double)mf > (double)a.edgeswitch * (double)a.nnz)
        f1_dense = true;
    } else {
      f1_dense = (a.mode == GRB_PULLONLY);
    }
    const unsigned int* Fc = Fp(iter - 1);
    unsigned int* Fn = Fp(iter);
    const bool direct = iter >= kKeep;                    // this level's bitmap will be recycled: label now
    const float new_label = direct ? (float)(iter + 1) : 0.f;
    LevelCounters c;
    // recycle: the frontier buffer of two levels ahead, the entry counter and the totals of
    // the next level (nobody touches them during this one)
    if (iter + 1 >= kKeep + 3)                            // a rotating buffer about to be reused
      for (long long i = gtid; i < nwords; i += gthreads) publish(&Fp(iter + 1)[i], 0u);
    if (gtid == 0) publish(&st->big_count[(iter + 1) & 1][0], 0u);
    if (bid == 0 && tid < 32) publish(&st->acc[(iter + 1) % 3][tid >> 2][tid & 3], 0ull);
    // A push level whose frontier carries many edges through few vertices runs at the rate of racing global
    // atomics (two per discovery, most attempts losers).  Such a level buckets its big vertices' edges by
    // destination range instead and lets the range's owner settle them in LDS: no global atomics at all.
    const bool heavy = !f1_dense && iter > 1 && a.oc_off != nullptr && nbig > 0 && mf >= a.oc_min_edges;

    if (!f1_dense) {
      // ================= push =================
      GRB_PHASE_START();
      unsigned int* V = Vp(cur);
      float* const label = ph(tp)->label;
      int2* const big_list = ph(gp)->big_list;
      if (iter == 1) {
        const Index source = ph(tp)->source;
        // the frontier is the source alone: its edges spread over the whole grid
        const Index e = a.optr[source + 1];
        for (long long p = a.optr[source] + gtid; p < e; p += gthreads) {
          const Index dst = a.oind[p];
          if (dst == source) continue;
          push_visit(a, label, V, Fn, dst, new_label, c);
        }
      } else {
        unsigned* bcount = &st->big_count[iter & 1][0];
        // One scan of the frontier bitmap (words interleaved over the workgroups) does everything that needs a
        // vertex's degree: rows of >= kBigDeg entries are LISTED (wave-aggregated append; they are expanded after a
        // barrier, by whole workgroups or by the owners of destination ranges), rows of < kSmallDeg entries are
        // expanded by the lane that found them, the ones in between by a wave each (workgroup-local LDS list).
        // A heavy level lists first and expands its small rows after the owners' phase (their racing atomics would
        // otherwise sit in front of the barrier every workgroup waits at); any other level does both in one scan.
        auto scan = [&](const bool do_list, const bool do_expand) {
        if (tid == 0) { s_nmed = 0; s_lcnt[0] = 0; s_lcnt[1] = 0; }
        __syncthreads();
        for (long long base = 0; base < nwords; base += gthreads) {
#if GRB_BFS_SCAN_CHUNKS
          // a wave reads 64 CONSECUTIVE words (two cache lines); consecutive 64-word chunks go to different workgroups, so
          // a frontier that is contiguous in vertex order (a grid's wave front) is still spread over the grid.  (One word
          // per lane with stride G, the first version, made every lane of a load a cache line of its own: 262 144 line
          // requests for the 4 096 lines of RMAT-22's bitmap in every push level.)
          const long long chunk = (base / kWave) + (long long)wave * G + bid;
          const long long i = chunk * kWave + lane;
#else
          const long long i = (base / G + tid) * G + bid;      // word index, stride G inside the WG
#endif
          const unsigned int w = (i < nwords) ? fresh(&Fc[i]) : 0u;
          int mine = 0;
          for (unsigned int t = w; t; t &= t - 1) {
            const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
            const Index s = a.optr[v], e = a.optr[v + 1];
            const Index d = e - s;
            if (d >= kBigDeg) { mine += heavy ? 1 : (d + kBigChunk - 1) / kBigChunk; continue; }
            if (!do_expand) continue;
            if (d >= kSmallDeg) {
              const int slot = atomicAdd(&s_nmed, 1);
              if (slot < kMed) { s_med[slot] = v; continue; }
            }
            for (Index p = s; p < e; ++p) push_visit(a, label, V, Fn, a.oind[p], new_label, c);
          }
          if (do_list) {
            int incl = mine;
incl = (int)wave_incl_scan_u32((unsigned)incl);
            const int total = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
#if GRB_BFS_LIST_WG
            // the waves reserve inside the workgroup's block (LDS), thread 0 reserves the block (one global atomic)
            unsigned b0 = 0;
            {
              int* lcnt = &s_lcnt[(int)((base / gthreads) & 1)];
              if (lane == 0 && total > 0) b0 = (unsigned)atomicAdd(lcnt, total);
              __syncthreads();
              if (tid == 0) {
                const int wg_total = *lcnt;
                s_lbase = wg_total > 0 ? atomicAdd(bcount, (unsigned)wg_total) : 0u;
                s_lcnt[(int)((base / gthreads) & 1) ^ 1] = 0;   // the next pass's counter (nobody is on it now)
              }
              __syncthreads();
              b0 = __shfl(b0, 0, kWave) + s_lbase;
            }
            if (total > 0) {
              int at = (int)b0 + incl - mine;
#else
            if (total > 0) {
              unsigned b0 = 0;
              if (lane == 0) b0 = atomicAdd(bcount, (unsigned)total);
              b0 = __shfl(b0, 0, kWave);
              int at = (int)b0 + incl - mine;
#endif
              for (unsigned int t = w; t; t &= t - 1) {
                const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
                const Index d = a.optr[v + 1] - a.optr[v];
                if (d >= kBigDeg)
                  for (int k = 0; k < (heavy ? 1 : (d + kBigChunk - 1) / kBigChunk); ++k, ++at)
                    if (at < a.big_cap) publish(reinterpret_cast<unsigned long long*>(&big_list[at]),
                                                ((unsigned long long)(unsigned)k << 32) | (unsigned)v);
              }
            }
          }
          if (!do_expand) continue;
          __syncthreads();
          const int nm = s_nmed < kMed ? s_nmed : kMed;
          for (int k = wave; k < nm; k += W) {
            const Index v = s_med[k];
            const Index e = a.optr[v + 1];
            for (Index p = a.optr[v] + lane; p < e; p += kWave) push_visit(a, label, V, Fn, a.oind[p], new_label, c);
          }
          __syncthreads();
          if (tid == 0) s_nmed = 0;
          __syncthreads();
        }
        };
        if (heavy) scan(true, false); else scan(nbig > 0, true);
        if (nbig > 0) {
          stamp();
          if (!grid_sync_at(&st->bar, gen, bid, G, false)) return -1;
          stamp();
          int nent = (int)__hip_atomic_load(bcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (nent > a.big_cap) nent = a.big_cap;
          if (!heavy) {
            for (int e = bid; e < nent; e += G) {
              const unsigned long long eb = fresh(reinterpret_cast<const unsigned long long*>(&big_list[e]));
              const int2 ent = make_int2((int)(eb & 0xffffffffull), (int)(eb >> 32));
              const Index pe = a.optr[ent.x + 1];
#pragma unroll
              for (int t = 0; t < kBigChunk / T; ++t) {
                const Index p = a.optr[ent.x] + ent.y * kBigChunk + t * T + tid;
                if (p < pe) push_visit(a, label, V, Fn, a.oind[p], new_label, c);
              }
            }
          } else {
            // ---- owner-computes: this workgroup's ranges, one after the other.  The pieces of a wave's 64 list
            // entries are laid end to end and dealt to the lanes 256 edges at a time (prefix sums in the LDS the
            // pull levels use for their leftovers), so a step costs one chain of memory latencies whatever the
            // piece lengths are; the range's new bits go out with one atomicOr per changed word -- the other
            // workgroups push the small vertices of the frontier with atomics meanwhile.
            for (int b = bid; b < a.oc_nb; b += G) {
              const Index v0 = a.oc_bounds[b];
              const int w0 = (int)(v0 >> 5);
              int nw = (int)((a.oc_bounds[b + 1] - v0 + 31) >> 5);
              if (w0 + nw > nwords) nw = nwords - w0;
              __syncthreads();
              for (int i = tid; i < nw; i += T) s_u.oc.ocw[i] = 0u;
              __syncthreads();
              for (int e0 = 0; e0 < nent; e0 += T) {
                const int e = e0 + tid;
                Index o0 = 0, o1 = 0;
                if (e < nent) {
                  const Index u = (Index)(fresh(reinterpret_cast<const unsigned long long*>(&big_list[e])) & 0xffffffffull);
                  const int r = a.oc_bigidx[u];
                  o0 = a.oc_off[(size_t)b * a.oc_nrows + r];
                  o1 = a.oc_off[(size_t)(b + 1) * a.oc_nrows + r];
                }
                const Index len = o1 - o0;
                Index inc = len;
inc = (Index)wave_incl_scan_u32((unsigned)inc);
                const Index total = (Index)__builtin_amdgcn_readlane((int)inc, kWave - 1);
                if (total == 0) continue;
                __builtin_amdgcn_wave_barrier();
                s_u.oc.row[wave][lane] = make_int2(inc - len, o0);
                __builtin_amdgcn_wave_barrier();
                for (Index at0 = 0; at0 < total; at0 += 4 * kWave) {
                  Index q[4];
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const Index at = at0 + j * kWave + lane;
                    q[j] = -1;
                    if (at < total) {
                      int r = 0;                           // the last entry whose first edge is <= at
#pragma unroll
                      for (int step = kWave / 2; step > 0; step >>= 1)
                        if (s_u.oc.row[wave][r + step].x <= at) r += step;
                      q[j] = s_u.oc.row[wave][r].y + (at - s_u.oc.row[wave][r].x);
                    }
                  }
                  Index d[4];
#pragma unroll
                  for (int j = 0; j < 4; ++j) d[j] = q[j] >= 0 ? a.oind[q[j]] : -1;
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (d[j] >= 0) atomicOr(&s_u.oc.ocw[(d[j] >> 5) - w0], 1u << (d[j] & 31));
                }
                __builtin_amdgcn_wave_barrier();
              }
              __syncthreads();
              for (int i = tid; i < nw; i += T) {
                const unsigned int acc = s_u.oc.ocw[i];
                if (!acc) continue;
                unsigned int newb = acc & ~fresh(&V[w0 + i]);
                if (!newb) continue;
                newb &= ~atomicOr(&V[w0 + i], newb);
                if (newb) {
                  atomicOr(&Fn[w0 + i], newb);
                  for (; newb; newb &= newb - 1)
                    discovered(a, label, (Index)((w0 + i) * 32) + (__ffs((int)newb) - 1), new_label, c);
                }
              }
            }
            stamp();
            scan(false, true);                             // the small and medium rows of a heavy level
          }
        }
      }
      last_dir = 0;
    } else {
      // ================= pull =================
      GRB_PHASE_START();
      // The barriers of this kernel do not invalidate; push levels read other workgroups' words
      // with fresh().  A pull level probes the visited bitmap millions of times, which is
      // faster through L1 with ordinary loads, so it pays the invalidate itself, once.
      const unsigned int* vin = Vp(cur);
      unsigned int* vout = Vp(cur ^ 1);
      float* const label = ph(tp)->label;
      const Index* hint = a.count_inspected ? nullptr : a.hint;
      const Index nchunks = (n + kWave - 1) / kWave;
      const Index nblocks = (nchunks + kPB - 1) / kPB;
      const Index nwaves = (Index)G * W;
      const unsigned long long lt_mask = (1ull << lane) - 1ull;
      typename PersistLds<T>::Pull& L = s_u.pull[wave];
      // Few vertices are left to discover (the levels after the big one): the dense walk below would carry 512-vertex
      // blocks with a handful of live lanes through its stages.  Here a wave numbers the active bits of kSparseWords
      // bitmap words (wave_for_each_bit) and takes them 64 at a time, one vertex per lane.  Same discoveries, same
      // accounting.
      const bool sparse_act = GRB_BFS_SPARSE_PULL && a.n_in >= 0 &&
                              (a.n_in - reached) * GRB_BFS_SPARSE_DIV < (long long)n;
#if GRB_BFS_PULL_DYN
      const Index pull_per = (nblocks + (Index)G - 1) / (Index)G;
      const Index pull_b0 = (Index)bid * pull_per;
      const Index pull_b1 = pull_b0 + pull_per < nblocks ? pull_b0 + pull_per : nblocks;
      if (tid == 0) s_pull_next = 0;
#endif
      if (!sparse_act || !GRB_BFS_SPARSE_FRESH) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
      }
      if (sparse_act) {
        constexpr bool kF = GRB_BFS_SPARSE_FRESH != 0;     // few probes: agent-scope loads instead of the invalidate
        const Index ngroups = (Index)((nwords + kSparseWords - 1) / kSparseWords);
        for (Index g = (Index)bid * W + wave; g < ngroups; g += nwaves) {
          const Index wi = g * kSparseWords + lane;
          const bool has_word = lane < kSparseWords && wi < nwords;
          unsigned int vw = 0xffffffffu, act = 0u;
          if (has_word) { vw = probe_word<kF>(vin, wi); act = ~(vw | a.skip[wi]); }
          if (__ballot(act != 0u) == 0ull) {
            if (has_word) publish(&vout[wi], vw);
            continue;
          }
          if (lane < kSparseWords) L.fresh_bits[lane] = 0u;
          wave_for_each_bit(&L.bits, act, lane, [&](int wl, int bit) {
            const bool on = wl >= 0;
            const Index v = on ? (g * kSparseWords + wl) * 32 + bit : 0;
            const Index hv = hint ? hint[v] : 0;
            const Index p = a.iptr[v], e = a.iptr[v + 1];
            bool found = false;
            // off lanes carry v = 0, and hint[0] is -1 when vertex 0 has no in-edges: probe word 0 then, never word -1
            if (hint) found = on && ((probe_word<kF>(vin, on ? (hv >> 5) : 0) >> (hv & 31)) & 1u);   // (an active vertex has in-edges: hv >= 0)
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
              pull_queue_run<kF, kQR, kQD>(a.iind, a.nnz, vin, L, lane, __popcll(um), c.inspected);
              if (und && ((L.found[lane >> 5] >> (lane & 31)) & 1u)) found = true;
              __builtin_amdgcn_wave_barrier();
            }
            if (found) {
              atomicOr(&L.fresh_bits[wl], 1u << bit);
              Index d;
              if (GRB_BFS_SYM && a.out_is_in) d = e - p;
              else d = a.optr[v + 1] - a.optr[v];
              ++c.found;
              c.deg += (unsigned long long)d;
              if (d >= kBigDeg) ++c.big;
              if (direct) label[v] = new_label;
            }
          });
          __builtin_amdgcn_wave_barrier();
          if (has_word) {
            const unsigned int nb = L.fresh_bits[lane];
            publish(&vout[wi], vw | nb);
            if (nb) publish(&Fn[wi], nb);
          }
          __builtin_amdgcn_wave_barrier();
        }
      } else
      // One wave owns a block of kPB chunks of 64 vertices and runs every stage for all of them at once, so a
      // stage costs one memory latency per block instead of one per chunk:  words -> hint probe (the row pointers
      // travel with it) -> the rows it did not settle, queued and taken dense (pull_queue_run) -> outputs.
#if GRB_BFS_PULL_DYN
      // the blocks of a workgroup's share are handed out as its waves ask for them (an LDS counter): a wave that drew
      // cheap blocks takes more of them, and the workgroup reaches the level's barrier when its work is done, not when
      // the wave with the two dearest blocks is
      for (;;) {
        Index blk = 0;
        if (lane == 0) blk = pull_b0 + (Index)atomicAdd(&s_pull_next, 1);
        blk = (Index)__builtin_amdgcn_readfirstlane((int)blk);
        if (blk >= pull_b1) break;
#else
      for (Index blk = (Index)bid * W + wave; blk < nblocks; blk += nwaves) {
#endif
        // ---- stage 0: the block's words; a lane's vertices are vbase + 64 j
        const Index wi = blk * (2 * kPB) + lane;
        const bool has_word = lane < 2 * kPB && wi < nwords;
        unsigned int vw = 0xffffffffu, inact = 0xffffffffu;
        if (has_word) { vw = vin[wi]; inact = vw | a.skip[wi]; }
        unsigned int act = 0;
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
          const unsigned int wj = __shfl(inact, 2 * j + (lane >> 5), kWave);
          act |= ((~wj >> (lane & 31)) & 1u) << j;
        }
        if (__ballot(act != 0u) == 0ull) {
          if (has_word) publish(&vout[wi], vw);
          continue;
        }
        const Index vbase = blk * (kPB * kWave) + lane;
        unsigned int fnd = 0;
        // ---- stage 1: the hinted in-neighbour of every active vertex; the row pointers travel
        // with it (coalesced, and needed by whoever the hint does not settle)
        Index p[kPB], e[kPB];
        {
          Index hv[kPB];
#pragma unroll
          for (int j = 0; j < kPB; ++j) {
            const Index vj = ((act >> j) & 1u) ? vbase + kWave * j : 0;
            hv[j] = hint ? hint[vj] : 0;
            p[j] = a.iptr[vj];
            e[j] = a.iptr[vj + 1];
          }
          if (hint) {
#pragma unroll
            for (int j = 0; j < kPB; ++j) {
              const unsigned int on = (act >> j) & 1u;
              const unsigned int w = vin[on ? (hv[j] >> 5) : 0];
              fnd |= (on & (w >> (hv[j] & 31)) & 1u) << j;
            }
          }
        }
        unsigned int und = act & ~fnd;
#pragma unroll
        for (int j = 0; j < kPB; ++j)
          if (p[j] >= e[j]) und &= ~(1u << j);
        if (__ballot(und != 0u)) {
          // ---- the undecided rows, queued in lane order
          if (lane < 2 * kPB) L.found[lane] = 0u;
          const int mine = __popc(und);
          int incl = mine;
incl = (int)wave_incl_scan_u32((unsigned)incl);
          const int Tq = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
          int at = incl - mine;
#pragma unroll
          for (int j = 0; j < kPB; ++j)
            if ((und >> j) & 1u) {
              L.row[at] = make_int2(p[j], e[j]);
              L.id[at] = (unsigned short)(j * kWave + lane);
              ++at;
            }
          __builtin_amdgcn_wave_barrier();
          pull_queue_run<false, kQR, kQD>(a.iind, a.nnz, vin, L, lane, Tq, c.inspected);
#pragma unroll
          for (int j = 0; j < kPB; ++j) fnd |= ((L.found[2 * j + (lane >> 5)] >> (lane & 31)) & 1u) << j;
          __builtin_amdgcn_wave_barrier();
        }
        // ---- outputs: new words, labels, accounting
        unsigned int nb = 0;
#pragma unroll
        for (int j = 0; j < kPB; ++j) {
          const unsigned long long fb = __ballot((fnd >> j) & 1u);
          if ((lane >> 1) == j) nb = (lane & 1) ? (unsigned int)(fb >> 32) : (unsigned int)(fb & 0xffffffffull);
        }
        if (has_word) {
          publish(&vout[wi], vw | nb);
          if (nb) publish(&Fn[wi], nb);
        }
        if (__ballot(fnd != 0u)) {
          if (GRB_BFS_SYM && a.out_is_in) {
            // the out-degree is the in-degree: no second pair of row pointers, no dependent load at the end of the step
#pragma unroll
            for (int j = 0; j < kPB; ++j)
              if ((fnd >> j) & 1u) {
                const Index d = e[j] - p[j];
                ++c.found;
                c.deg += (unsigned long long)d;
                if (d >= kBigDeg) ++c.big;
                if (direct) label[vbase + kWave * j] = new_label;
              }
          } else {
            Index d0[kPB], d1[kPB];
#pragma unroll
            for (int j = 0; j < kPB; ++j) {
              const bool f = (fnd >> j) & 1u;
              const Index vj = f ? vbase + kWave * j : 0;
              d0[j] = a.optr[vj];
              d1[j] = a.optr[vj + 1];
              if (f && direct) label[vj] = new_label;
            }
#pragma unroll
            for (int j = 0; j < kPB; ++j)
              if ((fnd >> j) & 1u) {
                const Index d = d1[j] - d0[j];
                ++c.found;
                c.deg += (unsigned long long)d;
                if (d >= kBigDeg) ++c.big;
              }
          }
        }
      }
      last_dir = 1;
    }

    stamp();
    if (a.trace && tid == 0 && levels < 12 && bid < 512) a.trace[256 + levels * 512 + bid] = wall_clock64() - t_level;
    // ---- level totals: one atomic per value per workgroup into this XCD group's line
    GRB_PHASE_START();
    // A wave's share of a level's totals fits 32 bits (every one of them is bounded by nnz, and Index is 32-bit); most
    // waves of most levels have nothing to report at all and skip the sums (a wave-uniform branch).
    unsigned long long r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#if GRB_BFS_DPP_TOTALS
    if (__ballot((c.found | c.inspected) != 0ull)) {
      r0 = wave_sum_u32((unsigned)c.found);
      r1 = wave_sum_u32((unsigned)c.deg);
      r2 = wave_sum_u32((unsigned)c.inspected);
      r3 = wave_sum_u32((unsigned)c.big);
    }
#else
    auto add = [](unsigned long long x, unsigned long long y) { return x + y; };
    r0 = wave_reduce(c.found, add); r1 = wave_reduce(c.deg, add);
    r2 = wave_reduce(c.inspected, add); r3 = wave_reduce(c.big, add);
#endif
    if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; }
    __syncthreads();
    unsigned long long* acc = &st->acc[iter % 3][0][0];
#if GRB_BFS_FINE_TRACE
    stamp();                                             // reduced + workgroup barrier
#endif
    if (tid < 4) {
      unsigned long long t = 0;
      for (int w = 0; w < W; ++w) t += s_red[w][tid];
      if (t) __hip_atomic_fetch_add(&acc[(bid & 7) * 16 + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#if GRB_BFS_FINE_TRACE
    {                                                    // grid_sync, stamped step by step (workgroup 0, thread 0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp();                                           // totals' adds have landed
      __syncthreads();
      stamp();                                           // every wave of this workgroup has arrived
      if (tid == 0) {
        const unsigned g = gen + 1;
        const unsigned x = bid & 7u;
        const unsigned groups = G < 8 ? (unsigned)G : 8u;
        const unsigned members = ((unsigned)G - x + 7u) / 8u;
        PersistState* const stl = st;
        const unsigned arr = __hip_atomic_fetch_add(&stl->bar.xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arr + 1u == members * g) (void)__hip_atomic_fetch_add(&stl->bar.top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp();                                         // own arrival returned
        unsigned npoll = 0;
        while (__hip_atomic_load(&stl->bar.top_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * g) { __builtin_amdgcn_s_sleep(1); ++npoll; if (npoll > kSpinLimit) break; }
        stamp();                                         // released
        if (a.trace && gtid == 0 && ntrace < 255) a.trace[1 + ntrace++] = npoll;   // (a count, not a time)
      }
      __syncthreads();
      ++gen;
    }
#else
    if (!grid_sync_at(&st->bar, gen, bid, G, false)) return -1;
#endif
    stamp();
    if (wave == 0) {
      unsigned long long q = 0;
      if (lane < 32) q = __hip_atomic_load(&acc[(lane >> 2) * 16 + (lane & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __shfl_xor(q, 4, kWave);
      q += __shfl_xor(q, 8, kWave);
      q += __shfl_xor(q, 16, kWave);
      if (lane < 4) s_tot[lane] = q;
    }
    __syncthreads();
    const unsigned long long tot_found = s_tot[0], tot_deg = s_tot[1], tot_insp = s_tot[2], tot_big = s_tot[3];
    stamp();
    if (gtid == 0 && levels < a.rec_cap) {
      grb_bfs_level& L = ph(gp)->rec[levels];
      L.direction = f1_dense ? 1 : 0;
      L.frontier = nf;
      L.frontier_edges = f1_dense ? (a.count_inspected ? (int64_t)tot_insp : 0) : (int64_t)mf;
      L.discovered = (int32_t)tot_found;
      L.ms = (float)(wall_clock64() - t_level) * a.ticks_to_ms;
    }
    ++levels;
    reached += (long long)tot_found;
    edges_cum += tot_deg;
    if (f1_dense) cur ^= 1;
    const float tmp = ratio_f1; ratio_f1 = ratio_f2; ratio_f2 = tmp;
    nf = (Index)tot_found;
    mf = tot_deg;
    nbig = tot_big;
    if (nf == 0) break;
  }

  // ---- the depth vector, written once and coalesced: level L + 1 for the vertices of F[L] (the kept bitmaps
  // are disjoint), 0 for everything never reached; a vertex that is visited but in none of the kept bitmaps was
  // discovered by a level >= kKeep and labelled there.  (V[cur] and every F are final after the last barrier.)
  {
    GRB_PHASE_START();
    // One bitmap word (32 vertices) per lane: the visited word and the word of every kept level go out together
    // (agent-scope loads: no invalidate to wait for), so the pass is one memory latency deep; a lane then writes its
    // 32 labels as eight 16-byte stores (a whole 128-byte line per lane).
    const unsigned int* Vf = Vp(cur);
    float* const label = ph(tp)->label;
    const int kept = levels + 1 < kKeep ? levels + 1 : kKeep;      // F[0 .. kept)
    const bool label_aligned = (reinterpret_cast<unsigned long long>(label) & 15ull) == 0ull;
#if GRB_BFS_LABEL_COAL
    // 32 words per wave step (every wave of the grid has one at n = 4 Mi); a word's six planes are computed by the
    // lane that loaded it and handed to the eight lanes that store its labels: a store instruction then writes
    // 64 x 16 consecutive bytes (eight whole lines) instead of 16 bytes in each of 64 lines.
    const long long nwave_all = (long long)G * (T / kWave);
    for (long long wb = ((long long)bid * (T / kWave) + wave) * 32; wb < nwords; wb += nwave_all * 32) {
      const long long wi0 = wb + (lane & 31);
      const bool have = wi0 < nwords;
      const long long wi = have ? wi0 : (long long)nwords - 1;
      const unsigned int vis = have ? fresh(&Vf[wi]) : 0u;
      unsigned int f[kKeep];
#pragma unroll
      for (int L0 = 0; L0 < kKeep; L0 += 8) {
        if (L0 < kept) {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = (L0 + u < kept && have) ? fresh(&Fp(L0 + u)[wi]) : 0u;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = 0u;
        }
      }
      unsigned int pl[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
      for (int L = 0; L < kKeep; ++L) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (((L + 1) >> k) & 1) pl[k] |= f[L];
      }
      const unsigned int keepm = vis & ~(pl[0] | pl[1] | pl[2] | pl[3] | pl[4] | pl[5]);
      const bool whole = label_aligned && (wb + 32) * 32 <= (long long)n && __ballot(keepm != 0u) == 0ull;
      if (whole) {
        LabelQuad* out = reinterpret_cast<LabelQuad*>(label + wb * 32) + lane;
        const int b0 = (lane & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int src = q * 8 + (lane >> 3);
          unsigned int lab[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const unsigned int pk = (unsigned int)__shfl((int)pl[k], src, kWave) >> b0;
#pragma unroll
            for (int t = 0; t < 4; ++t) lab[t] |= ((pk >> t) & 1u) << k;
          }
          LabelQuad x;
          x.x = (float)lab[0]; x.y = (float)lab[1]; x.z = (float)lab[2]; x.w = (float)lab[3];
          out[q * 64] = x;
        }
      } else if (have && lane < 32) {
        const long long v0 = wi * 32;
        for (int b = 0; b < 32 && v0 + b < (long long)n; ++b) {
          unsigned int lab = 0u;
#pragma unroll
          for (int k = 0; k < 6; ++k) lab |= ((pl[k] >> b) & 1u) << k;
          if (!((keepm >> b) & 1u)) label[v0 + b] = (float)lab;
        }
      }
    }
  }
#else
    for (long long wi = gtid; wi < nwords; wi += gthreads) {
      const unsigned int vis = fresh(&Vf[wi]);
      unsigned int f[kKeep];
#pragma unroll
      for (int L0 = 0; L0 < kKeep; L0 += 8) {
        if (L0 < kept) {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = L0 + u < kept ? fresh(&Fp(L0 + u)[wi]) : 0u;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) f[L0 + u] = 0u;
        }
      }
      // vertical counters: bit b of plane k = bit k of the label of vertex 32 wi + b (levels < kKeep = 32: 6 planes)
      unsigned int pl[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
      for (int L = 0; L < kKeep; ++L) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (((L + 1) >> k) & 1) pl[k] |= f[L];
      }
      const long long v0 = wi * 32;
      // a vertex that is visited but in no kept bitmap was labelled by a level >= kKeep: its label stays
      const unsigned int keepm = vis & ~(pl[0] | pl[1] | pl[2] | pl[3] | pl[4] | pl[5]);
      if (v0 + 32 <= (long long)n && keepm == 0u && label_aligned) {
        float4* out = reinterpret_cast<float4*>(label + v0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float x[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int b = q * 4 + t;
            unsigned int lab = 0u;
#pragma unroll
            for (int k = 0; k < 6; ++k) lab |= ((pl[k] >> b) & 1u) << k;
            x[t] = (float)lab;
          }
          out[q] = make_float4(x[0], x[1], x[2], x[3]);
        }
      } else {
        for (int b = 0; b < 32 && v0 + b < (long long)n; ++b) {
          unsigned int lab = 0u;
#pragma unroll
          for (int k = 0; k < 6; ++k) lab |= ((pl[k] >> b) & 1u) << k;
          if (!((keepm >> b) & 1u)) label[v0 + b] = (float)lab;
        }
      }
    }
  }
#endif
  stamp();
  if (a.trace && gtid == 0) a.trace[0] = (unsigned long long)ntrace;
  // ---- the end: this block's level count for whoever clears it, the next traversal's number, the record.  In a chained
  // launch a barrier comes first -- the next traversal clears a block and must find nobody in this one's bitmaps -- and
  // the record is written behind it: every workgroup's label stores have completed by then.
  int next = 0x7fffffff;
  if (gtid == 0) publish(&ph(gp)->rot[kHostRot ? 0 : 1 + blk], (unsigned int)levels);
  if (chained) {
    if (gtid == 0) publish(&st->next_idx[0], (unsigned)n_grids + (atomicAdd(ctr, 1u) - ctr_base));
    if (!grid_sync_at(&st->bar, gen, bid, G, false)) return -1;
    next = (int)fresh(&st->next_idx[0]);
  }
  if (gtid == 0) {
    const unsigned long long tag = (unsigned long long)(unsigned int)ph(tp)->seq << 32;
    const float ms = (float)(wall_clock64() - t_start) * a.ticks_to_ms;
    const unsigned int vals[8] = {(unsigned int)levels, (unsigned int)last_dir, (unsigned int)reached,
                                  (unsigned int)(edges_cum & 0xffffffffull), (unsigned int)(edges_cum >> 32),
                                  (unsigned int)nf, (unsigned int)(iter > a.max_niter ? 1 : 0),
                                  __float_as_uint(ms)};
    unsigned long long* mail = ph(tp)->mail;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      __hip_atomic_store(&mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return next;
#undef st
#undef V0
}

// ---- the launch: one grid per traversal in flight, several traversals per grid ---------------------------------------
// A traversal is barriers and dependent-load chains for half of its time (about 50 of 105 us move no
// bytes), and none of that gets shorter with more CUs.  A launch therefore carries up to kCoTrain queued traversals
// and runs n_grids of them (1 .. kCoMax) side by side: the launch is n_grids sub-grids of G = CUs workgroups of
// T = 1024 / 512 / 256 threads, workgroup b of the launch is workgroup b % G of sub-grid b / G (so a workgroup keeps the
// XCD its number implies, and -- the dispatcher placing workgroups in order -- a CU holds one workgroup of every
// sub-grid).  Every traversal has the whole device's CUs, L1s and LDS bandwidth; its barriers, totals and latency
// chains are filled with the other traversals' waves by the hardware scheduler, and nothing depends on how the runtime
// maps streams to hardware queues (grb_bfs_set_lanes does).  The sub-grids are independent: private state blocks,
// bitmaps, lists, barrier counters (a lane's worth each, BfsLane below); a sub-grid that finishes a traversal draws
// the next one from the launch's counter.  The argument blocks sit in the kernarg segment and a workgroup reads its own
// through the segment pointer (scalar loads from constant memory, what a by-value parameter compiles to when it is
// not indexed at run time -- a by-value table that is would be copied to scratch).
#ifndef GRB_CO_MAX
#define GRB_CO_MAX 12
#endif
constexpr int kCoMax = GRB_CO_MAX;            // sub-grids per launch at most (two waves each on a CU: GRB_CO_LEAN_WPE x 4 / 2)
constexpr int kCoTrain = 48;
struct LaunchArgs {
  PersistArgs a;
  GridArgs g[kCoMax];
  TravArgs t[kCoTrain];
  unsigned int* ctr;                // the launch's counter (monotonic over launches; ctr_base: where it stood)
  unsigned ctr_base;
  int ntrav, n_grids, G;
};
static_assert(sizeof(LaunchArgs) <= 4096, "the kernarg segment holds 4 KiB");
typedef const __attribute__((address_space(4))) LaunchArgs* LaunchArgsPtr;
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T > GRB_CO_LEAN_FROM ? 4 : GRB_CO_LEAN_WPE, T > GRB_CO_LEAN_FROM ? 4 : GRB_CO_LEAN_WPE))) void bfs_persistent_kernel(LaunchArgs la_) {
  const LaunchArgsPtr la = (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (T == kPThreads) {
    // one traversal on the launch's whole grid: every argument at a fixed place of the segment
    (void)bfs_persistent_body<T>(&la->a, &la->g[0], &la->t[0], (int)blockIdx.x, (int)gridDim.x, 0u, false, nullptr, 0u, 1);
    return;
  }
  const int G = la->G;
  const int j = (int)blockIdx.x / G;
  const int bid = (int)blockIdx.x - j * G;
  const auto* g = &la->g[j];
  const int ntrav = la->ntrav;
  const bool chained = ntrav > la->n_grids;
  unsigned trot = fresh(&g->rot[0]);                     // (nobody writes it before the launch's last traversal of this grid ends)
  int idx = j;
  while (idx < ntrav) {
    idx = bfs_persistent_body<T>(&la->a, g, &la->t[idx], bid, G, trot, chained, la->ctr, la->ctr_base, la->n_grids);
    if (idx < 0) return;
    ++trot;
  }
  if (bid == 0 && threadIdx.x == 0) publish(&g->rot[0], trot);
}

// which rows are big (>= kBigDeg entries), their numbers and their list, on the device (round 5: the host walked its
// mirror of the row pointers and uploaded a 16 MB table -- 4 of a first traversal's 12 ms, more on a busy host)
__global__ void oc_big_flag_kernel(const Index* __restrict__ ptr, Index nrows, unsigned int* __restrict__ flag /* [nrows + 1] */) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v <= nrows; v += stride)
    flag[v] = (v < nrows && ptr[v + 1] - ptr[v] >= kBigDeg) ? 1u : 0u;
}
__global__ void oc_big_place_kernel(const Index* __restrict__ ptr, Index nrows, const unsigned int* __restrict__ before,
                                    int* __restrict__ bigidx, Index* __restrict__ rows) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v < nrows; v += stride) {
    const bool big = ptr[v + 1] - ptr[v] >= kBigDeg;
    bigidx[v] = big ? (int)before[v] : -1;
    if (big) rows[before[v]] = v;
  }
}
// do two pointer arrays hold the same numbers?  (*differ is raised when not)
__global__ void ptr_differ_kernel(const Index* __restrict__ a, const Index* __restrict__ b, Index n1, unsigned int* __restrict__ differ) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  bool d = false;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) d |= a[i] != b[i];
  if (__ballot(d) && (threadIdx.x & (kWave - 1)) == 0) *differ = 1u;
}

// destinations of the big rows' entries, counted in bins of kOcBin vertices (one wave per row).  With at most kOcLdsBins
// bins (RMAT-22: 16 Ki) a workgroup counts in LDS and adds its non-empty bins to memory once -- 55 M global atomics on
// 16 Ki words were 3 ms of a matrix's first traversal.
constexpr int kOcLdsBins = 16384;
constexpr Index kOcLongRow = 8192;
__global__ __launch_bounds__(1024) void oc_mass_kernel(const Index* __restrict__ optr, const Index* __restrict__ oind,
                                                       const Index* __restrict__ rows, int nrows, int nbins, unsigned int* __restrict__ bins) {
  __shared__ unsigned int h[kOcLdsBins];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const bool lds = nbins <= kOcLdsBins;
  if (lds) {
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) h[i] = 0u;
    __syncthreads();
  }
  const int nwaves = gridDim.x * waves;
  // a row of kOcLongRow entries and more is walked by a whole workgroup (one wave alone on RMAT-22's longest row was
  // 1.2 ms, this kernel's whole time), any other by a wave
  for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
    const Index u = rows[r];
    const Index s0 = optr[u], e = optr[u + 1];
    if (e - s0 < kOcLongRow) continue;
    for (Index p = s0 + (Index)threadIdx.x; p < e; p += (Index)blockDim.x) {
      const int b = oind[p] / kOcBin;
      if (lds) atomicAdd(&h[b], 1u); else atomicAdd(&bins[b], 1u);
    }
  }
  for (int r = blockIdx.x * waves + wave; r < nrows; r += nwaves) {
    const Index u = rows[r];
    const Index s0 = optr[u], e = optr[u + 1];
    if (e - s0 >= kOcLongRow) continue;
    for (Index p = s0 + lane; p < e; p += kWave) {
      const int b = oind[p] / kOcBin;
      if (lds) atomicAdd(&h[b], 1u); else atomicAdd(&bins[b], 1u);
    }
  }
  if (lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x)
      if (h[i]) atomicAdd(&bins[i], h[i]);
  }
}

#ifndef GRB_OC_PARTS
#define GRB_OC_PARTS 8
#endif
constexpr int kOcParts = GRB_OC_PARTS;
// where big row rows[r] enters range b: off[b * nrows + r] = the first entry of the row with a destination >= bounds[b]
__global__ void oc_range_off_kernel(const Index* __restrict__ optr, const Index* __restrict__ oind, const Index* __restrict__ rows,
                                    int nrows, int R, const Index* __restrict__ bounds, Index* __restrict__ off) {
  // a thread per big row walks the bounds in order, each search starting where the last one ended (a gallop, then a
  // bisection of the bracket): a row's entries are read about once, and a step's stores are consecutive over the rows.
  // (A thread per (range, row) bisecting the whole row, the first version: 18 M independent searches, 3.6 ms -- kept for
  // the rows of kOcLongRow entries and more, oc_range_off_long_kernel: one thread walking 257 bounds through 300 000
  // entries is a chain of 5 000 dependent loads.)
  // (round 6: kOcParts threads per row, each walking its share of the bounds -- the first of them found by a bisection of
  // the whole row: a chain an eighth as long, 0.91 -> see docs/experiments.md R6.3)
  const int per = (R + 1 + kOcParts - 1) / kOcParts;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows * kOcParts; i += gridDim.x * blockDim.x) {
    const int r = i % nrows, part = i / nrows;                 // (neighbouring threads: neighbouring rows, the same bounds -- their stores stay consecutive)
    const Index u = rows[r];
    const Index e = optr[u + 1];
    Index p = optr[u];
    if (e - p >= kOcLongRow) continue;
    const int b0 = part * per, b1 = b0 + per < R + 1 ? b0 + per : R + 1;
    for (int b = b0; b < b1; ++b) {
      const Index key = bounds[b];
      Index lo = p, hi = p, step = 1;
      if (b == b0 && part > 0) {
        hi = e;                                                // this thread's first bound: anywhere in the row
      } else {
        while (hi < e && oind[hi] < key) { lo = hi + 1; hi += step; step <<= 1; }
        if (hi > e) hi = e;
      }
      while (lo < hi) {
        const Index mid = lo + (hi - lo) / 2;
        if (oind[mid] < key) lo = mid + 1; else hi = mid;
      }
      p = lo;
      off[(long long)b * nrows + r] = p;
    }
  }
}
__global__ void oc_range_off_long_kernel(const Index* __restrict__ optr, const Index* __restrict__ oind, const Index* __restrict__ rows,
                                         int nrows, int R, const Index* __restrict__ bounds, Index* __restrict__ off) {
  // a wave per long row, a lane per bound (64 at a time): independent bisections of the whole row
  const int lane = threadIdx.x & (kWave - 1);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nrows; r += nwaves) {
    const Index u = rows[r];
    const Index s0 = optr[u], e = optr[u + 1];
    if (e - s0 < kOcLongRow) continue;
    for (int b = lane; b <= R; b += kWave) {
      const Index key = bounds[b];
      Index lo = s0, hi = e;
      while (lo < hi) {
        const Index mid = lo + (hi - lo) / 2;
        if (oind[mid] < key) lo = mid + 1; else hi = mid;
      }
      off[(long long)b * nrows + r] = lo;
    }
  }
}

}  // namespace grb

using namespace grb;

// The tables of the owner-computes push for the rows [0, nrows) of a CSR whose columns are [0, ncols): the rows of
// >= kBigDeg entries numbered (bigidx, -1 for the others), destination ranges cut at equal mass of those rows' own
// destinations (counted on the device in bins of kOcBin vertices; at most kOcWords words wide), and where each big
// row enters each range ([nb + 1][nbig]).  Leaves *d_off null when there is nothing to gain (no big rows, a table
// beyond 64 M entries).  Also used by the partitioned traversal (bfs_part_run.hip) for a rank's out-edge shard.
grb_info grb::oc_tables_build(const Index* d_ptr, const Index* d_ind, const std::vector<Index>& optr, Index nrows, Index ncols, int G,
                              Index** d_bounds, Index** d_off, int** d_bigidx, int* nb, int* nbig, int max_words) {
  hipStream_t s = ctx().stream;
  *d_bounds = nullptr; *d_off = nullptr; *d_bigidx = nullptr; *nb = 0; *nbig = 0;
  const Index n = ncols;
  (void)optr;
  if (nrows <= 0 || n < 2 * kOcBin) return GRB_SUCCESS;
  // the big rows: flags, their exclusive scan (= a big row's number), the list
  // (temporaries from the stream-ordered allocator: a hipFree waits for the whole device, three of them were 0.2 ms of a
  // matrix's first traversal)
  unsigned int* d_flag = nullptr;
  GRB_HIP_TRY(hipMallocAsync((void**)&d_flag, 4 * ((size_t)nrows + 1), s));
  struct FreeFlag { void* p; hipStream_t s; ~FreeFlag() { (void)hipFreeAsync(p, s); } } free_flag{d_flag, s};
  hipLaunchKernelGGL(oc_big_flag_kernel, dim3(stream_grid((long long)nrows + 1, kBlock)), dim3(kBlock), 0, s, d_ptr, nrows, d_flag);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_exclusive_scan_u32(d_flag, (long long)nrows + 1));
  unsigned int nbig_u = 0;
  GRB_HIP_TRY(hipMemcpyAsync(&nbig_u, d_flag + nrows, 4, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  const size_t nrows_big = (size_t)nbig_u;
  if (nrows_big == 0) return GRB_SUCCESS;
  const int nbins = (int)(((long long)n + kOcBin - 1) / kOcBin);
  std::vector<unsigned int> bins((size_t)nbins, 0u);
  void *p_rows = nullptr, *p_bins = nullptr;
  int* d_big = nullptr;
  struct FreeTemp { void** p; hipStream_t s; ~FreeTemp() { if (*p) (void)hipFreeAsync(*p, s); } } free_rows{&p_rows, s}, free_bins{&p_bins, s};   // on every way out
  GRB_HIP_TRY(hipMallocAsync(&p_rows, sizeof(Index) * nrows_big, s));
  GRB_HIP_TRY(hipMalloc((void**)&d_big, sizeof(int) * (size_t)nrows));
  struct FreeBig { int** p; ~FreeBig() { if (*p) (void)hipFree(*p); } } free_big{&d_big};   // (handed over on success)
  GRB_HIP_TRY(hipMallocAsync(&p_bins, 4 * (size_t)nbins, s));
  GRB_HIP_TRY(hipMemsetAsync(p_bins, 0, 4 * (size_t)nbins, s));
  hipLaunchKernelGGL(oc_big_place_kernel, dim3(stream_grid(nrows, kBlock)), dim3(kBlock), 0, s, d_ptr, nrows, (const unsigned int*)d_flag,
                     d_big, (Index*)p_rows);
  GRB_HIP_TRY(hipGetLastError());
  {
    const int waves_needed = (int)nrows_big;
    int mgrid = (waves_needed + 15) / 16;
    if (mgrid > ctx().num_cu) mgrid = ctx().num_cu;
    if (mgrid < 1) mgrid = 1;
    hipLaunchKernelGGL(oc_mass_kernel, dim3(mgrid), dim3(1024), 0, s, d_ptr, d_ind, (const Index*)p_rows, (int)nrows_big, nbins,
                       (unsigned int*)p_bins);
  }
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipMemcpyAsync(bins.data(), p_bins, 4 * (size_t)nbins, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  (void)hipFreeAsync(p_bins, s);
  p_bins = nullptr;
  long long total = 0;
  for (unsigned int x : bins) total += (long long)x;
  long long target = std::max<long long>(1, total / (1ll * G));
  std::vector<Index> bounds;
  // At most one range per workgroup (GRB_BFS_OC_FIT=0: whatever the first cut gives, 5 % more than workgroups on
  // RMAT-22): the workgroups that own two ranges are the owner phase's critical path, twice everybody else's.
  static const bool fit = !getenv("GRB_BFS_OC_FIT") || atoi(getenv("GRB_BFS_OC_FIT")) != 0;
  for (int attempt = 0; attempt < 64; ++attempt) {
    bounds.assign(1, 0);
    static const int cap_env = getenv("GRB_BFS_OC_WIDTH") ? atoi(getenv("GRB_BFS_OC_WIDTH")) : 0;   // vertices per range at most
    int max_bins = max_words * 32 / kOcBin;                // a range's slice of the visited bitmap fits the LDS buffer
    if (cap_env >= kOcBin && cap_env / kOcBin < max_bins) max_bins = cap_env / kOcBin;
    long long acc = 0;
    int first = 0;
    for (int b = 0; b < nbins; ++b) {
      acc += (long long)bins[(size_t)b];
      if (acc >= target || b + 1 - first >= max_bins || b + 1 == nbins) {
        bounds.push_back((Index)std::min<long long>((long long)n, (long long)(b + 1) * kOcBin));
        first = b + 1;
        acc = 0;
      }
    }
    if (bounds.back() != n) bounds.push_back(n);
    if (!fit || (long long)bounds.size() - 1 <= G || (nbins + max_bins - 1) / max_bins > G) break;   // fits, or never can
    target += std::max<long long>(1, target / 32);
  }
  const long long R = (long long)bounds.size() - 1;
  if (R < 2 || (long long)nrows_big * (R + 1) > (64ll << 20)) return GRB_SUCCESS;
  GRB_HIP_TRY(hipMalloc((void**)d_bounds, sizeof(Index) * bounds.size()));
  GRB_HIP_TRY(hipMalloc((void**)d_off, sizeof(Index) * nrows_big * (size_t)(R + 1)));
  GRB_HIP_TRY(hipMemcpyAsync(*d_bounds, bounds.data(), sizeof(Index) * bounds.size(), hipMemcpyHostToDevice, s));
  *d_bigidx = d_big;
  d_big = nullptr;                                         // (the caller's now)
  hipLaunchKernelGGL(oc_range_off_kernel, dim3(stream_grid((long long)nrows_big * kOcParts, 64)), dim3(64), 0, s, d_ptr, d_ind,
                     (const Index*)p_rows, (int)nrows_big, (int)R, (const Index*)*d_bounds, *d_off);
  hipLaunchKernelGGL(oc_range_off_long_kernel, dim3(stream_grid((long long)nrows_big * kWave, kBlock)), dim3(kBlock), 0, s, d_ptr, d_ind,
                     (const Index*)p_rows, (int)nrows_big, (int)R, (const Index*)*d_bounds, *d_off);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipStreamSynchronize(s));                    // the host vectors above go out of scope
  *nb = (int)R;
  *nbig = (int)nrows_big;
  return GRB_SUCCESS;
}

// ---- host side ---------------------------------------------------------------------------------------------------
// A traversal is one launch queued on the library's stream and one 8-granule record in pinned host memory that the
// kernel's last instruction writes.  Launching and
// reading the record are separate steps (bfs_persistent_launch / bfs_persistent_collect): grb_bfs_fused does one after
// the other, grb_bfs_fused_enqueue / grb_bfs_wait let the caller queue K traversals and wait once.
namespace {
constexpr int kRing = 256;                       // records (= traversals in flight) at most
struct BfsTicket {
  int state = 0;                                 // 0 free, 1 in flight, 2 already complete (ran synchronously), 3 waiting for a
                                                 // co-scheduled launch to fill, 4 could not be launched (the wait runs it)
  int seq = 0;
  int lane = 0;
  grb_vector v = nullptr;
  grb_matrix A = nullptr;
  grb_descriptor desc = nullptr;
  grb_index source = 0;
  int max_niter = 0;                             // as the descriptor stood when the traversal was queued (the wait's unlabel pass)
  grb_bfs_result res = {};
};
// A lane = what one traversal in flight needs for itself: a stream, the two state blocks, V1, the big-vertex list, the
// level records, the level-count word.  Lane 0 is the library's own stream and scratch slots (the blocking call's path,
// and the only lane unless grb_bfs_set_lanes asks for more); lanes 1.. own their memory.  With L lanes a queued
// traversal runs on num_cu / L workgroups, and L of them are resident at once: a traversal is latency and barriers for
// two thirds of its time (section 5 of DESIGN.md), so two on half the device each finish in 1.4 x the time of one on
// all of it, four in 2 x.
constexpr int kMaxLanes = 8;
struct BfsLane {
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_done = nullptr;
  void *zero = nullptr, *v1 = nullptr, *big = nullptr, *rec = nullptr;
  size_t zero_cap = 0, v1_cap = 0, big_cap = 0, rec_cap = 0;
  size_t clean_bytes = 0;                        // zero_bytes the blocks were last cleared for (0: not clean)
  unsigned int* d_rot = nullptr;                  // PersistArgs::rot
  int block = 0;                                  // one traversal per launch: which of the two blocks the next one runs on
  unsigned long long fenced_epoch = ~0ull;       // ApiScope::epoch when this lane last fenced against the library's stream
};
struct BfsRules {                                // the descriptor fields a traversal runs under, as they stood when it was queued
  int mode = 0, max_niter = 0;
  float switchpoint = 0.f, edgeswitch = 0.f;
  bool operator==(const BfsRules& o) const {
    return mode == o.mode && max_niter == o.max_niter && switchpoint == o.switchpoint && edgeswitch == o.edgeswitch;
  }
};
static BfsRules bfs_rules_of(grb_descriptor desc) {
  BfsRules r;
  r.mode = desc->desc[GRB_MXVMODE]; r.max_niter = desc->max_niter; r.switchpoint = desc->switchpoint; r.edgeswitch = desc->edgeswitch;
  return r;
}
struct CoPend {                                  // a traversal that has its ticket and waits for company (co-scheduling)
  int slot = 0, seq = 0;
  grb_vector v = nullptr;
  grb_matrix A = nullptr;
  grb_index source = 0;
  grb_descriptor desc = nullptr;
  BfsRules rules;                                // (the descriptor's setters do not launch what has gathered: a traversal keeps
                                                 // the rules it was queued under, and one launch serves one set of rules)
};
struct BfsRing {
  int co_width = 1;                              // traversals per launch (grb_bfs_set_coschedule); 1: every traversal its own launch
  int co_n = 0;                                  // ... and the ones that wait for the launch to fill (ticket state 3)
  CoPend co[kCoTrain];
  BfsLane lane[kMaxLanes + 1 + kCoMax];          // [0]: the library's stream (blocking calls, one lane); [1 ..]: the lanes proper;
                                                 // [kMaxLanes + 1 ..]: the sub-grids of the launches of several traversals
  int lanes = 1, next_lane = 0;
  bool lanes_active = false;                     // the launch being queued is one of several in flight (set around enqueue)
  unsigned long long* h = nullptr;               // pinned, host-coherent: kRing x 8 granules {value, seq}
  unsigned long long* d = nullptr;               // the device-side address of h
  unsigned int* d_rot = nullptr;                 // PersistArgs::rot of lane 0 (the library's scratch slots)
  int block = 0;                                 // ... and which of its two blocks the next traversal runs on
  unsigned int* d_ctr = nullptr;                 // LaunchArgs::ctr: the chained launches' counter, and where it stands
  unsigned ctr_base = 0;
  bool ctr_dirty = false;                        // a chained launch did not finish: the counter is anywhere
  // grb_bfs_coschedule_profile: HIP events around the launches of several traversals (measurement passes only)
  bool co_profile = false;
  std::vector<hipEvent_t>* co_ev = nullptr;      // pairs (on the heap, never freed: the ring stays trivially destructible --
                                                 // nothing of it may run at exit, when the HIP runtime may be gone)
  size_t co_ev_used = 0;
  int co_prof_trav = 0;
  BfsTicket t[kRing];
  int next = 0;
  int poisoned_upto = 0;                         // records with seq <= this were queued behind a traversal that failed
  double enqueue_us = 0, wait_us = 0;
  long long calls = 0;
};
BfsRing g_ring;
grb_info ring_init() {
  BfsRing& r = g_ring;
  if (r.h) return GRB_SUCCESS;
  GRB_TRY(ctx_init());
  GRB_HIP_TRY(hipHostMalloc((void**)&r.h, sizeof(unsigned long long) * 8 * kRing, hipHostMallocMapped | hipHostMallocCoherent));
  memset(r.h, 0, sizeof(unsigned long long) * 8 * kRing);
  GRB_HIP_TRY(hipHostGetDevicePointer((void**)&r.d, r.h, 0));
  GRB_HIP_TRY(hipMalloc((void**)&r.d_rot, 512));
  GRB_HIP_TRY(hipMemset(r.d_rot, 0, 512));
  r.d_ctr = r.d_rot + 64;
  return GRB_SUCCESS;
}
// the record in slot `slot` once it carries tag `seq`: spins, then (after 5 ms) waits for the stream the traversal was
// queued on (its lane's) and looks again
grb_info ring_wait(int slot, int seq, unsigned int* out, hipStream_t stream) {
  const unsigned long long* hg = g_ring.h + 8 * (size_t)slot;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  bool synced = false;
  for (;;) {
    bool ok = true;
    for (int k = 0; k < 8; ++k) {
      const unsigned long long g = __atomic_load_n(&hg[k], __ATOMIC_ACQUIRE);
      if ((int)(g >> 32) != seq) { ok = false; break; }
      out[k] = (unsigned int)(g & 0xffffffffull);
    }
    if (ok) return GRB_SUCCESS;
    if (synced) {
      fprintf(stderr, "libgrb_hip: traversal record not published after stream sync\n");
      return GRB_PANIC;
    }
    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
      GRB_HIP_TRY(hipStreamSynchronize(stream));
      synced = true;
    }
  }
}
}  // namespace

// Queues one traversal on the library's stream; its record will appear in ring slot `slot` under
// tag *seq_out.
static grb_info lane_buffer(void** p, size_t* cap, size_t bytes, hipStream_t s) {
  if (*cap >= bytes) return GRB_SUCCESS;
  if (*p) { GRB_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(*p); *p = nullptr; *cap = 0; }
  const size_t want = (bytes + bytes / 4 + 255) & ~(size_t)255;
  GRB_HIP_TRY(hipMalloc(p, want));
  *cap = want;
  return GRB_SUCCESS;
}

// What a launch needs besides its argument block: whose buffers it runs on and what to note once it is queued.
struct LaunchCtx {
  int lane_id = 0;
  bool co = false;               // one of the sub-grids of ONE launch on the library's stream (lane_id = its number there)
  hipStream_t s = nullptr;
  int G = 0;
  void* p_zero = nullptr;
  size_t zero_bytes = 0, block_bytes = 0;
  int* p_blocksel = nullptr;     // one traversal per launch: the host's side of the rotation
};

// Fills the argument block of one (sub-)grid: the lane's buffers (lane 0: the library's scratch slots), the once-per-
// matrix facts and tables.  Queues at most memsets on lc->s.
static grb_info bfs_persistent_args(grb_matrix A, const BfsRules& rules, int profile, int lane_id, bool co, PersistArgs* out,
                                    GridArgs* gout, LaunchCtx* lc, void** p_rec_out, unsigned long long** trace_out, int oc_words = kOcWords,
                                    int g_mult = 1) {
  GRB_TRY(ring_init());
  Context& c = ctx();
  BfsLane& ln = g_ring.lane[lane_id];
  if (lane_id > 0 && !ln.d_rot) {
    GRB_HIP_TRY(hipMalloc((void**)&ln.d_rot, 256));
    GRB_HIP_TRY(hipMemset(ln.d_rot, 0, 256));
  }
  if (lane_id > 0 && !co && !ln.stream) {
    GRB_HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    GRB_HIP_TRY(hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
    GRB_HIP_TRY(hipEventCreateWithFlags(&ln.ev_done, hipEventDisableTiming));
  }
  hipStream_t s = (lane_id > 0 && !co) ? ln.stream : c.stream;
  unsigned int* d_rot = lane_id > 0 ? ln.d_rot : g_ring.d_rot;
  const Index n = A->nrows;
  const int nwords = 2 * ceil_div(n, 64);
  int wgs_per_cu = 1;
  if (const char* e = getenv("GRB_BFS_WGS_PER_CU")) wgs_per_cu = atoi(e) >= 2 ? 2 : 1;
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, bfs_persistent_kernel<kPThreads>, kPThreads, 0));
    if (max_per_cu < 1) return GRB_PANIC;
  }
  if (wgs_per_cu > max_per_cu) wgs_per_cu = max_per_cu;
  const int G_full = c.num_cu * wgs_per_cu;
  // lanes > 1: a queued traversal takes its share of the CUs (the blocking call, lane 0 alone, the whole device);
  // a sub-grid of a co-scheduled launch has a workgroup on every CU
  const int G = co ? c.num_cu * g_mult
                   : (g_ring.lanes > 1 && profile == 0 && g_ring.lanes_active) ? (c.num_cu / g_ring.lanes > 0 ? c.num_cu / g_ring.lanes : 1)
                                                                               : G_full;
  const int rec_cap = 1 << 15;
  const int big_cap = (int)(A->nvals / kBigDeg) + 2;

  // one allocation: three blocks [state | V0 | F0 .. F(kKeep + 2)], used in rotation (PersistArgs::blocks)
  const size_t st_bytes = (sizeof(PersistState) + 255) & ~(size_t)255;
  const size_t block_bytes = (st_bytes + 4 * (size_t)(1 + kKeep + 3) * (size_t)nwords + 255) & ~(size_t)255;
  const size_t zero_bytes = (co ? 3 : 2) * block_bytes;
  void *p_zero, *p_v1, *p_big, *p_rec;
  int* p_blocksel = lane_id > 0 ? &ln.block : &g_ring.block;
  if (lane_id == 0) {
    GRB_TRY(scratch(7, zero_bytes, &p_zero));
    GRB_TRY(scratch(8, 4 * (size_t)nwords, &p_v1));
    GRB_TRY(scratch(2, sizeof(int2) * (size_t)big_cap, &p_big));
    GRB_TRY(scratch(11, sizeof(grb_bfs_level) * (size_t)rec_cap, &p_rec));
    // every block is clear (and the rotation words say "nothing to clear") when somebody else has had the slot
    if (c.bfs_prezero_ptr != p_zero || c.bfs_prezero_bytes != zero_bytes) {
      GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
      GRB_HIP_TRY(hipMemsetAsync(d_rot, 0, 16, s));
      g_ring.block = 0;
    }
    c.bfs_prezero_ptr = nullptr;
  } else {
    const size_t had = ln.zero_cap;
    GRB_TRY(lane_buffer(&ln.zero, &ln.zero_cap, zero_bytes, s));
    GRB_TRY(lane_buffer(&ln.v1, &ln.v1_cap, 4 * (size_t)nwords, s));
    GRB_TRY(lane_buffer(&ln.big, &ln.big_cap, sizeof(int2) * (size_t)big_cap, s));
    GRB_TRY(lane_buffer(&ln.rec, &ln.rec_cap, sizeof(grb_bfs_level) * (size_t)rec_cap, s));
    p_zero = ln.zero; p_v1 = ln.v1; p_big = ln.big; p_rec = ln.rec;
    if (had != ln.zero_cap || ln.clean_bytes != zero_bytes) {     // new memory, or a graph of another size: clear every block
      GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
      GRB_HIP_TRY(hipMemsetAsync(d_rot, 0, 16, s));
      ln.block = 0;
    }
    ln.clean_bytes = 0;
  }

  static float ticks_to_ms = 0.f;
  if (ticks_to_ms == 0.f) {
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ticks_to_ms = khz > 0 ? 1.0f / (float)khz : 1e-5f;
  }

  PersistArgs& a = *out;
  a.optr = A->csr.ptr; a.oind = A->csr.ind;
  a.iptr = A->csc.ptr; a.iind = A->csc.ind;
  a.skip = A->d_no_in_edges;
  a.hint = A->d_pull_hint;
  a.n = n;
  a.nnz = A->nvals;
  // once per matrix: how many vertices have in-edges at all (the complement of the skip bitmap), and whether the two
  // pointer arrays hold the same numbers (the same array, or equal host mirrors).  On the LIBRARY's stream, whatever
  // lane the traversal goes to: the skip bitmap is made there (ensure_empty_rows), and waiting for it here also puts
  // every other first-use preparation of this matrix (the pull hint) in front of the lane's launch.
  if (A->bfs_n_in < 0) {
    hipStream_t sp = c.stream;
    std::vector<unsigned int> h_skip((size_t)nwords);
    GRB_HIP_TRY(hipMemcpyAsync(h_skip.data(), A->d_no_in_edges, 4 * (size_t)nwords, hipMemcpyDeviceToHost, sp));
    GRB_HIP_TRY(hipStreamSynchronize(sp));
    long long empty = 0;
    for (unsigned int w : h_skip) empty += __builtin_popcount(w);
    A->bfs_n_in = (long long)nwords * 32 - empty;           // the padding bits beyond n are set in the skip bitmap
    A->bfs_out_is_in = A->csr.ptr == A->csc.ptr;
    if (!A->bfs_out_is_in && A->csr.ptr && A->csc.ptr && A->nrows == A->ncols) {
      // compared on the device (two 17 MB host mirrors were compared here: 2 ms, and mirrors need not exist)
      unsigned int* d_differ = nullptr;
      GRB_HIP_TRY(hipMalloc((void**)&d_differ, 4));
      GRB_HIP_TRY(hipMemsetAsync(d_differ, 0, 4, sp));
      hipLaunchKernelGGL(ptr_differ_kernel, dim3(stream_grid((long long)n + 1, kBlock)), dim3(kBlock), 0, sp, A->csr.ptr, A->csc.ptr,
                         n + 1, d_differ);
      unsigned int differ = 1u;
      const hipError_t e1 = hipMemcpyAsync(&differ, d_differ, 4, hipMemcpyDeviceToHost, sp);
      const hipError_t e2 = hipStreamSynchronize(sp);
      (void)hipFree(d_differ);
      GRB_HIP_TRY(e1);
      GRB_HIP_TRY(e2);
      A->bfs_out_is_in = differ == 0u;
    }
  }
  a.n_in = A->bfs_n_in;
  a.out_is_in = A->bfs_out_is_in ? 1 : 0;
  a.mode = rules.mode;
  a.switchpoint = rules.switchpoint;
  a.edgeswitch = rules.edgeswitch;
  a.max_niter = rules.max_niter;
  a.count_inspected = (profile & 2) ? 1 : 0;
  gout->blocks = (char*)p_zero;
  a.block_bytes = (unsigned long long)block_bytes;
  a.st_bytes = (unsigned long long)st_bytes;
  gout->v1 = (unsigned int*)p_v1;
  gout->rot = d_rot;
  gout->big_list = (int2*)p_big;
  a.big_cap = big_cap;
  // owner-computes push: the tables are made once per matrix (ranges of equal in-edge mass, at most kOcWords words
  // wide, about two per workgroup; the big rows; where each big row enters each range)
  a.oc_bounds = nullptr; a.oc_off = nullptr; a.oc_bigidx = nullptr; a.oc_nb = 0; a.oc_nrows = 0; a.oc_min_edges = ~0ull;
  {
    const char* e = getenv("GRB_BFS_OC_MIN");              // frontier out-edges from which a push level uses it; 0 = off
    const long long oc_min = e ? atoll(e) : 262144;
    const bool narrow = G != G_full || oc_words != kOcWords;   // a lane's grid, or narrower slices: its own tables (the ranges are cut per workgroup)
    const int oc2_key = G + (oc_words << 12);
    if (!narrow) {
      if (oc_min > 0 && A->oc_state == 0) {
        A->oc_state = -1;
        if (A->nvals > 0 && (Index)A->h_csr_ptr.size() == n + 1) {
          GRB_TRY(oc_tables_build(A->csr.ptr, A->csr.ind, A->h_csr_ptr, n, n, G, &A->d_oc_bounds, &A->d_oc_off, &A->d_oc_bigidx,
                                  &A->oc_nb, &A->oc_nrows));
          if (A->d_oc_off) A->oc_state = 1;
        }
      }
      if (oc_min > 0 && A->oc_state == 1) {
        a.oc_bounds = A->d_oc_bounds;
        a.oc_off = A->d_oc_off;
        a.oc_bigidx = A->d_oc_bigidx;
        a.oc_nb = A->oc_nb;
        a.oc_nrows = A->oc_nrows;
        a.oc_min_edges = (unsigned long long)oc_min;
      }
    } else {
      if (oc_min > 0 && (A->oc2_state == 0 || A->oc2_grid != oc2_key)) {
        if (A->d_oc2_bounds || A->d_oc2_off || A->d_oc2_bigidx) {
          GRB_HIP_TRY(hipDeviceSynchronize());               // (traversals of other lanes may be reading the old tables)
          (void)hipFree(A->d_oc2_bounds); (void)hipFree(A->d_oc2_off); (void)hipFree(A->d_oc2_bigidx);
          A->d_oc2_bounds = nullptr; A->d_oc2_off = nullptr; A->d_oc2_bigidx = nullptr;
        }
        A->oc2_state = -1;
        A->oc2_grid = oc2_key;
        if (A->nvals > 0 && (Index)A->h_csr_ptr.size() == n + 1) {
          GRB_TRY(oc_tables_build(A->csr.ptr, A->csr.ind, A->h_csr_ptr, n, n, G, &A->d_oc2_bounds, &A->d_oc2_off, &A->d_oc2_bigidx,
                                  &A->oc2_nb, &A->oc2_nrows, oc_words));
          if (A->d_oc2_off) A->oc2_state = 1;
          GRB_HIP_TRY(hipStreamSynchronize(c.stream));       // (built on the library's stream; the lanes read them)
        }
      }
      if (oc_min > 0 && A->oc2_state == 1) {
        a.oc_bounds = A->d_oc2_bounds;
        a.oc_off = A->d_oc2_off;
        a.oc_bigidx = A->d_oc2_bigidx;
        a.oc_nb = A->oc2_nb;
        a.oc_nrows = A->oc2_nrows;
        a.oc_min_edges = (unsigned long long)oc_min;
      }
    }
  }
  gout->rec = (grb_bfs_level*)p_rec;
  a.rec_cap = co ? 0 : rec_cap;                              // (per-level records: the blocking call's profiling runs read them)
  a.ticks_to_ms = ticks_to_ms;
  *p_rec_out = p_rec;
  *trace_out = nullptr;
  static const bool want_trace = getenv("GRB_BFS_TRACE") != nullptr;
  a.trace = nullptr;
  if (want_trace && lane_id == 0 && !co) {
    void* p_tr;
    GRB_TRY(scratch(10, (256 + 12 * 512) * sizeof(unsigned long long), &p_tr));
    GRB_HIP_TRY(hipMemsetAsync(p_tr, 0, (256 + 12 * 512) * sizeof(unsigned long long), s));
    a.trace = (unsigned long long*)p_tr;
    *trace_out = a.trace;
  }
  lc->lane_id = lane_id; lc->co = co; lc->s = s; lc->G = G; lc->p_zero = p_zero; lc->zero_bytes = zero_bytes;
  lc->block_bytes = block_bytes; lc->p_blocksel = p_blocksel;
  return GRB_SUCCESS;
}
// the launch has been queued: the lane's blocks are in the state the rotation words describe
static grb_info bfs_persistent_queued(const LaunchCtx& lc) {
  Context& c = ctx();
  if (!lc.co) *lc.p_blocksel ^= 1;                          // the next traversal runs on the other block and clears this one
  if (lc.lane_id == 0) {
    c.bfs_prezero_ptr = lc.p_zero;
    c.bfs_prezero_bytes = lc.zero_bytes;
  } else {
    BfsLane& ln = g_ring.lane[lc.lane_id];
    ln.clean_bytes = lc.zero_bytes;
    if (!lc.co) GRB_HIP_TRY(hipEventRecord(ln.ev_done, lc.s));   // what the library's stream waits for before it touches v
  }
  return GRB_SUCCESS;
}

// A launch that needs the whole device resident (the blocking traversal, a co-scheduled group; also the other one-launch
// algorithms, through grb::bfs_lanes_fence) must not meet a lane's narrower grid half-way: both would spin at their
// barriers until the bound.  It waits for what the lanes have in flight.
grb_info grb::bfs_lanes_fence(hipStream_t s) {
  if (g_ring.lanes <= 1) return GRB_SUCCESS;
  for (int l = 1; l <= kMaxLanes; ++l) {
    BfsLane& ln = g_ring.lane[l];
    if (ln.stream && ln.ev_done && ln.stream != s) GRB_HIP_TRY(hipStreamWaitEvent(s, ln.ev_done, 0));
  }
  return GRB_SUCCESS;
}
// the library's stream has been given work that a lane's next launch must come after (the wait path's unlabel / re-run)
void grb::bfs_lanes_unfence() {
  for (int l = 1; l <= kMaxLanes; ++l) g_ring.lane[l].fenced_epoch = ~0ull;
}

// Queues one traversal (on the library's stream, or on its lane's); its record will appear in ring slot `slot` under
// tag *seq_out.
static grb_info bfs_persistent_launch(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int profile, int slot,
                                      int* seq_out, void** p_rec_out, unsigned long long** trace_out, int lane_id = 0, int seq_in = 0,
                                      const BfsRules* rules_in = nullptr) {
  Context& c = ctx();
  LaunchArgs la;
  memset(&la, 0, sizeof(la));
  LaunchCtx lc;
  GRB_TRY(bfs_persistent_args(A, rules_in ? *rules_in : bfs_rules_of(desc), profile, lane_id, false, &la.a, &la.g[0], &lc, p_rec_out, trace_out));
  la.t[0].label = (float*)v->d_val;
  la.t[0].block = (char*)lc.p_zero + (size_t)(*lc.p_blocksel) * lc.block_bytes;
  la.t[0].clean = (char*)lc.p_zero + (size_t)((*lc.p_blocksel) ^ 1) * lc.block_bytes;
  la.t[0].mail = g_ring.d + 8 * (size_t)slot;
  la.t[0].source = source;
  la.t[0].seq = seq_in ? seq_in : ++c.mail_seq;
  la.ctr = g_ring.d_ctr;
  la.ntrav = 1; la.n_grids = 1; la.G = lc.G;
  *seq_out = la.t[0].seq;
  hipStream_t s = lc.s;
  BfsLane& ln = g_ring.lane[lane_id];
  // whatever OTHER entry points have queued on the library's stream since this lane last looked (a fill of v, a build of
  // A) comes first; the traversal queue's own calls do not count -- lane 0's traversals live on that stream
  if (lane_id > 0 && ln.fenced_epoch != ApiScope::epoch) {
    GRB_HIP_TRY(hipEventRecord(ln.ev_in, c.stream));
    GRB_HIP_TRY(hipStreamWaitEvent(s, ln.ev_in, 0));
    ln.fenced_epoch = ApiScope::epoch;
  }
  if (lane_id == 0 && !g_ring.lanes_active) GRB_TRY(bfs_lanes_fence(s));   // a whole-device grid
  if (profile & 1) GRB_HIP_TRY(hipEventRecord(c.ev0, s));
  // The grid barrier needs every workgroup resident at once.  GRB_BFS_COOPERATIVE=1 asks the runtime to
  // guarantee that (hipLaunchCooperativeKernel fails fast when it cannot); the default launch relies on the
  // occupancy query above and on the bounded spins of the barrier.  Either way a traversal that cannot run
  // here is reported as GRB_NOT_IMPLEMENTED / GRB_PANIC and grb_bfs_fused re-runs it with the host-driven loop.
  static const bool cooperative = [] { const char* e = getenv("GRB_BFS_COOPERATIVE"); return e && atoi(e) != 0; }();
  static const bool force_fallback = [] { const char* e = getenv("GRB_BFS_FORCE_FALLBACK"); return e && atoi(e) != 0; }();
  if (force_fallback) return GRB_NOT_IMPLEMENTED;            // test hook: behave as if the launch had been refused
  if (cooperative) {
    void* kargs[] = {&la};
    if (hipLaunchCooperativeKernel(reinterpret_cast<void*>(bfs_persistent_kernel<kPThreads>), dim3(lc.G), dim3(kPThreads), kargs, 0, s) !=
        hipSuccess) {
      (void)hipGetLastError();
      return GRB_NOT_IMPLEMENTED;
    }
  } else {
    hipLaunchKernelGGL(bfs_persistent_kernel<kPThreads>, dim3(lc.G), dim3(kPThreads), 0, s, la);
    GRB_HIP_TRY(hipGetLastError());
  }
  if (profile & 1) GRB_HIP_TRY(hipEventRecord(c.ev1, s));
  return bfs_persistent_queued(lc);
}

// ntrav traversals (2 .. kCoTrain) of one matrix under one descriptor in ONE launch on the library's stream: n_grids
// sub-grids (sub-grid j on lane j's buffers) that draw the traversals from the launch's table.
template <int T>
static grb_info co_kernel_fits(int k) {
  static int per_cu = -1;
  if (per_cu < 0) {
    int m = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, bfs_persistent_kernel<T>, T, 0) != hipSuccess) { (void)hipGetLastError(); m = 0; }
    per_cu = m;
  }
  return per_cu >= k ? GRB_SUCCESS : GRB_NOT_IMPLEMENTED;
}
static grb_info bfs_co_launch(int ntrav, const CoPend* pend, int width, int g_mult = 1) {
  Context& c = ctx();
  if (ntrav < (g_mult > 1 ? 1 : 2) || ntrav > kCoTrain) return GRB_INVALID_VALUE;
  static const bool force_fallback = [] { const char* e = getenv("GRB_BFS_FORCE_FALLBACK"); return e && atoi(e) != 0; }();
  if (force_fallback) return GRB_NOT_IMPLEMENTED;
  int n_grids = width < ntrav ? width : ntrav;
  if (n_grids > kCoMax) n_grids = kCoMax;
  // two sub-grids: 512-thread workgroups; three or four: 256; up to twelve: 128 (built for six waves per SIMD)
  const int T = n_grids <= 2 ? 512 : n_grids <= 4 ? 256 : 128;   // (a launch the CU cannot hold is refused: co_kernel_fits)
  GRB_TRY(T == 512 ? co_kernel_fits<512>(n_grids * g_mult) : T == 256 ? co_kernel_fits<256>(n_grids * g_mult) : co_kernel_fits<128>(n_grids * g_mult));
  const int oc_words = T >= 256 ? kOcWords : kOcWords / 4;
  LaunchArgs la;
  memset(&la, 0, sizeof(la));
  LaunchCtx lc[kCoMax];
  for (int j = 0; j < n_grids; ++j) {
    void* p_rec = nullptr;
    unsigned long long* trace = nullptr;
    GRB_TRY(bfs_persistent_args(pend[0].A, pend[0].rules, 0, kMaxLanes + 1 + j, true, &la.a, &la.g[j], &lc[j], &p_rec, &trace, oc_words, g_mult));
  }
  for (int i = 0; i < ntrav; ++i) {
    la.t[i].label = (float*)pend[i].v->d_val;
    la.t[i].mail = g_ring.d + 8 * (size_t)pend[i].slot;
    la.t[i].source = pend[i].source;
    la.t[i].seq = pend[i].seq;
  }
  if (g_ring.ctr_dirty) {
    GRB_HIP_TRY(hipMemsetAsync(g_ring.d_ctr, 0, 4, c.stream));
    g_ring.ctr_base = 0;
    g_ring.ctr_dirty = false;
  }
  la.ctr = g_ring.d_ctr;
  la.ctr_base = g_ring.ctr_base;
  la.ntrav = ntrav; la.n_grids = n_grids; la.G = lc[0].G;
  GRB_TRY(bfs_lanes_fence(c.stream));
  if (g_ring.co_profile) {
    if (!g_ring.co_ev) g_ring.co_ev = new std::vector<hipEvent_t>();
    while (g_ring.co_ev->size() < g_ring.co_ev_used + 2) {
      hipEvent_t e;
      GRB_HIP_TRY(hipEventCreate(&e));
      g_ring.co_ev->push_back(e);
    }
    GRB_HIP_TRY(hipEventRecord((*g_ring.co_ev)[g_ring.co_ev_used], c.stream));
  }
  if (T == 512) hipLaunchKernelGGL(bfs_persistent_kernel<512>, dim3(n_grids * la.G), dim3(512), 0, c.stream, la);
  else if (T == 256) hipLaunchKernelGGL(bfs_persistent_kernel<256>, dim3(n_grids * la.G), dim3(256), 0, c.stream, la);
  else hipLaunchKernelGGL(bfs_persistent_kernel<128>, dim3(n_grids * la.G), dim3(128), 0, c.stream, la);
  GRB_HIP_TRY(hipGetLastError());
  if (g_ring.co_profile) {
    GRB_HIP_TRY(hipEventRecord((*g_ring.co_ev)[g_ring.co_ev_used + 1], c.stream));
    g_ring.co_ev_used += 2;
    g_ring.co_prof_trav += ntrav;
  }
  if (ntrav > n_grids) g_ring.ctr_base += (unsigned)ntrav;   // a chained launch draws once per traversal
  for (int j = 0; j < n_grids; ++j) GRB_TRY(bfs_persistent_queued(lc[j]));
  return GRB_SUCCESS;
}

// Waits for the record of a queued traversal and unpacks it.
static grb_info bfs_persistent_collect(int slot, int seq, int profile, void* p_rec, unsigned long long* trace,
                                       grb_bfs_level* levels_out, int max_levels, int* levels, int* last_dir,
                                       long long* reached, unsigned long long* edges, Index* nf_left, bool* hit_cap,
                                       float* tight_ms, int lane_id = 0) {
  Context& c = ctx();
  hipStream_t s = lane_id > 0 && g_ring.lane[lane_id].stream ? g_ring.lane[lane_id].stream : c.stream;
  const int rec_cap = 1 << 15;
  unsigned int gv[8];
  if (seq <= g_ring.poisoned_upto) return GRB_PANIC;            // queued behind a traversal that did not finish
  {
    const grb_info wi = ring_wait(slot, seq, gv, s);
    if (wi != GRB_SUCCESS) {
      // the kernel left early (its barrier gave up) without leaving its level count: the next launch cleared too little
      // of this block, and every traversal queued since ran on whatever that left
      c.bfs_prezero_ptr = nullptr;
      for (int l = 1; l < kMaxLanes + 1 + kCoMax; ++l) g_ring.lane[l].clean_bytes = 0;
      g_ring.poisoned_upto = c.mail_seq;
      g_ring.ctr_dirty = true;
      return wi;
    }
  }
  *levels = (int)gv[0];
  *last_dir = (int)gv[1];
  *reached = (long long)gv[2];
  *edges = ((unsigned long long)gv[4] << 32) | gv[3];
  *nf_left = (Index)gv[5];
  *hit_cap = gv[6] != 0;
  float ms;
  memcpy(&ms, &gv[7], 4);             // the kernel's own wall clock, first to last instruction
  if (profile & 1) {                  // profiling runs report the HIP-event time of the launch instead
    GRB_HIP_TRY(hipEventSynchronize(c.ev1));
    GRB_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  }
  *tight_ms = ms;
  if (trace) {
    unsigned long long h[256];
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    const double tick_us = khz > 0 ? 1e3 / (double)khz : 1e-2;
    GRB_HIP_TRY(hipMemcpyAsync(h, trace, sizeof(h), hipMemcpyDeviceToHost, s));
    GRB_HIP_TRY(hipStreamSynchronize(s));
    fprintf(stderr, "bfs trace (us since kernel start; init, then per level: expanded / barrier / totals):");
    for (unsigned long long i = 0; i < h[0] && i < 255; ++i) fprintf(stderr, " %.1f", (double)h[1 + i] * tick_us);
    fprintf(stderr, "\n");
    // when each workgroup had finished its share of a level (us since the level began): min / median / p90 / max
    {
      std::vector<unsigned long long> w((size_t)12 * 512);
      GRB_HIP_TRY(hipMemcpyAsync(w.data(), trace + 256, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      GRB_HIP_TRY(hipStreamSynchronize(s));
      const int G = c.num_cu < 512 ? c.num_cu : 512;
      fprintf(stderr, "bfs workgroup finish times per level (min/median/p90/max us):");
      for (int L = 0; L < 12 && L < *levels; ++L) {
        std::vector<double> t;
        for (int b = 0; b < G; ++b) t.push_back((double)w[(size_t)L * 512 + b] * tick_us);
        std::sort(t.begin(), t.end());
        fprintf(stderr, " [%.1f %.1f %.1f %.1f]", t.front(), t[t.size() / 2], t[t.size() * 9 / 10], t.back());
      }
      fprintf(stderr, "\n");
    }
  }
  if (levels_out && max_levels > 0) {
    const int k = *levels < max_levels ? (*levels < rec_cap ? *levels : rec_cap) : max_levels;
    if (k > 0) {
      GRB_HIP_TRY(hipMemcpyAsync(levels_out, p_rec, sizeof(grb_bfs_level) * (size_t)k, hipMemcpyDeviceToHost, s));
      GRB_HIP_TRY(hipStreamSynchronize(s));
    }
  }
  return GRB_SUCCESS;
}

// ---- tickets: a traversal that has been queued and not yet waited for -----------------------------------------------
// A free record of the ring (its previous traversal has been waited for); GRB_INSUFFICIENT_SPACE when kRing are in flight.
grb_info grb::bfs_ticket_take(int* slot) {
  GRB_TRY(ring_init());
  for (int k = 0; k < kRing; ++k) {
    const int i = (g_ring.next + k) % kRing;
    if (g_ring.t[i].state == 0) { g_ring.next = (i + 1) % kRing; *slot = i; return GRB_SUCCESS; }
  }
  return GRB_INSUFFICIENT_SPACE;
}
grb_info grb::bfs_persistent_enqueue(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int slot, int* seq) {
  void* p_rec = nullptr;
  unsigned long long* trace = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  GRB_TRY(ring_init());
  BfsTicket& t = g_ring.t[slot];
  if (g_ring.co_width > 1 && g_ring.lanes == 1) {
    // co-scheduling: the traversal gets its ticket now and its launch when somebody waits for a ticket, when any other
    // entry point is called (bfs_co_flush), or when kCoTrain of them have gathered -- a launch runs co_width of them at a
    // time and hands out the rest as its sub-grids come free, so the more it carries the less its tail weighs; the host
    // queues a ticket in a microsecond, so gathering costs the device nothing it notices.  One launch serves one matrix
    // and one set of descriptor fields.
    const BfsRules rules = bfs_rules_of(desc);
    if (g_ring.co_n > 0 && (g_ring.co[0].A != A || !(g_ring.co[0].rules == rules))) GRB_TRY(bfs_co_flush());
    *seq = ++ctx().mail_seq;
    CoPend& p = g_ring.co[g_ring.co_n++];
    p.slot = slot; p.seq = *seq; p.v = v; p.A = A; p.source = source; p.desc = desc; p.rules = rules;
    t.state = 3; t.seq = *seq; t.lane = 0; t.v = v; t.A = A; t.desc = desc; t.source = source; t.max_niter = rules.max_niter;
    grb_info fi = GRB_SUCCESS;
    if (g_ring.co_n >= kCoTrain) fi = bfs_co_flush();       // the launch's table is full
    g_ring.enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ++g_ring.calls;
    return fi;
  }
  // lanes: the queued traversals go round the lanes, each lane's launches in order on its own stream
  // (lane 0 is the library's stream -- measured: with it as one of the lanes four launches overlap, with four created
  // streams only two or three do, whatever GPU_MAX_HW_QUEUES says)
  int lane = 0;
  if (g_ring.lanes > 1) { lane = g_ring.next_lane; g_ring.next_lane = (g_ring.next_lane + 1) % g_ring.lanes; }
  g_ring.lanes_active = g_ring.lanes > 1;
  const grb_info li = bfs_persistent_launch(v, A, source, desc, 0, slot, seq, &p_rec, &trace, lane);
  g_ring.lanes_active = false;
  GRB_TRY(li);
  g_ring.enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  ++g_ring.calls;
  t.state = 1; t.seq = *seq; t.lane = lane; t.v = v; t.A = A; t.desc = desc; t.source = source; t.max_niter = desc->max_niter;
  return GRB_SUCCESS;
}
// Launches what waits: all of it in one launch when there are two or more (co_width sub-grids: 512-thread workgroups for
// two, 256 for three or four), a lone one as an ordinary traversal.  A launch that is refused leaves its tickets in state 4: grb_bfs_wait
// runs those traversals itself.
bool grb::bfs_co_pending() { return g_ring.co_n > 0; }
grb_info grb::bfs_co_flush() {
  const int k = g_ring.co_n;
  if (k == 0) return GRB_SUCCESS;
  CoPend pend[kCoTrain];
  for (int j = 0; j < k; ++j) pend[j] = g_ring.co[j];
  g_ring.co_n = 0;
  grb_info li = GRB_NOT_IMPLEMENTED;
  // (GRB_BFS_SINGLE_WIDE=m, a measurement hook: a lone traversal on m workgroups of 512 threads per CU -- needs a build
  // whose 512-thread instance fits m per CU, -DGRB_CO_LEAN_FROM=512; tools/bfs_single_wide_probe.py.  Three per CU, 24
  // waves, measured 0.111 ms against 0.098 for the 1024-thread kernel: one traversal is not short of waves)
  static const int single_wide = getenv("GRB_BFS_SINGLE_WIDE") ? atoi(getenv("GRB_BFS_SINGLE_WIDE")) : 0;
  if (k >= 2) li = bfs_co_launch(k, pend, g_ring.co_width);
  else if (single_wide > 1) li = bfs_co_launch(1, pend, 1, single_wide);
  if (li == GRB_SUCCESS) {
    for (int j = 0; j < k; ++j) g_ring.t[pend[j].slot].state = 1;
    return GRB_SUCCESS;
  }
  grb_info worst = GRB_SUCCESS;
  for (int j = 0; j < k; ++j) {                            // one by one (a lone traversal, or the group launch was refused)
    void* p_rec = nullptr;
    unsigned long long* trace = nullptr;
    int seq = 0;
    const grb_info si = bfs_persistent_launch(pend[j].v, pend[j].A, pend[j].source, pend[j].desc, 0, pend[j].slot, &seq, &p_rec, &trace, 0,
                                              pend[j].seq, &pend[j].rules);
    g_ring.t[pend[j].slot].state = si == GRB_SUCCESS ? 1 : 4;
    if (si != GRB_SUCCESS && si != GRB_NOT_IMPLEMENTED && si != GRB_PANIC) worst = si;
  }
  return worst;
}
// HIP events around the launches of several traversals: on != 0 starts collecting (and forgets what was collected);
// on == 0 stops, waits for the launches and reports their summed duration, their number and the traversals they ran.
grb_info grb::bfs_co_profile(int on, double* ms_total, int* launches, int* traversals) {
  GRB_TRY(ring_init());
  if (on) {
    GRB_TRY(bfs_co_flush());
    g_ring.co_profile = true;
    g_ring.co_ev_used = 0;
    g_ring.co_prof_trav = 0;
    return GRB_SUCCESS;
  }
  GRB_TRY(bfs_co_flush());
  g_ring.co_profile = false;
  double tot = 0;
  for (size_t i = 0; i + 1 < g_ring.co_ev_used; i += 2) {
    float ms = 0.f;
    GRB_HIP_TRY(hipEventSynchronize((*g_ring.co_ev)[i + 1]));
    GRB_HIP_TRY(hipEventElapsedTime(&ms, (*g_ring.co_ev)[i], (*g_ring.co_ev)[i + 1]));
    tot += (double)ms;
  }
  if (ms_total) *ms_total = tot;
  if (launches) *launches = (int)(g_ring.co_ev_used / 2);
  if (traversals) *traversals = g_ring.co_prof_trav;
  g_ring.co_ev_used = 0;
  g_ring.co_prof_trav = 0;
  return GRB_SUCCESS;
}
// Traversals per launch (1 .. kCoMax).  Everything queued so far is launched and waited for first.  Returns the previous value.
int grb::bfs_co_setting(int set) {
  const int before = g_ring.co_width;
  if (set >= 1 && set != before) {
    (void)bfs_co_flush();
    (void)hipDeviceSynchronize();
    g_ring.co_width = set > kCoMax ? kCoMax : set;
  }
  return before;
}
// Traversals in flight at once (1 .. 8): n lanes of num_cu / n workgroups each.  Everything queued so far is waited for
// first (a launch needs its whole grid resident: lanes of different widths must not meet).  Returns the previous value.
int grb::bfs_lanes_setting(int set) {
  const int before = g_ring.lanes;
  if (set >= 1 && set != before) {
    (void)bfs_co_flush();
    (void)hipDeviceSynchronize();
    int l = 1;
    while (2 * l <= set && 2 * l <= kMaxLanes) l *= 2;     // powers of two: a grid of num_cu / 3 workgroups measured 3 x slower per launch
    g_ring.lanes = l;
    g_ring.next_lane = 0;
  }
  return before;
}
// A traversal that ran synchronously (a path the ring does not serve) parks its result in a ticket all the same.
void grb::bfs_ticket_store(int slot, int seq, const grb_bfs_result& res) {
  BfsTicket& t = g_ring.t[slot];
  t.state = 2; t.seq = seq; t.res = res;
}
int grb::bfs_ticket_state(int slot, int seq, grb_vector* v, grb_matrix* A, grb_descriptor* desc, grb_index* source,
                          grb_bfs_result* parked, int* max_niter) {
  if (slot < 0 || slot >= kRing || !g_ring.h) return 0;
  const BfsTicket& t = g_ring.t[slot];
  if (t.state == 0 || t.seq != seq) return 0;
  if (v) *v = t.v;
  if (A) *A = t.A;
  if (desc) *desc = t.desc;
  if (source) *source = t.source;
  if (parked) *parked = t.res;
  if (max_niter) *max_niter = t.max_niter;
  return t.state;
}
void grb::bfs_ticket_release(int slot) { g_ring.t[slot].state = 0; }
grb_info grb::bfs_persistent_wait(int slot, int seq, int* levels, int* last_dir, long long* reached, unsigned long long* edges,
                                  Index* nf_left, bool* hit_cap, float* tight_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  const int lane = g_ring.t[slot].lane;
  const grb_info r = bfs_persistent_collect(slot, seq, 0, nullptr, nullptr, nullptr, 0, levels, last_dir, reached, edges, nf_left,
                                            hit_cap, tight_ms, lane);
  // a lane's launch is not ordered against the library's stream by itself: whatever is queued there from now on
  // (reading the labels, say) waits for the lane's last launch -- the record is written by ONE workgroup's last
  // instruction, others may still be storing labels
  if (r == GRB_SUCCESS && lane > 0 && g_ring.lane[lane].ev_done) (void)hipStreamWaitEvent(ctx().stream, g_ring.lane[lane].ev_done, 0);
  // ... and a caller who has taken the vector's storage (grb_vector_device_ptrs: zero-copy interop) may read it on a stream
  // the library knows nothing about: for such a vector the wait is for the LAUNCH, whose end also writes the labels back
  // from the L2s they were stored through
  if (r == GRB_SUCCESS && g_ring.t[slot].v && g_ring.t[slot].v->exposed) {
    hipStream_t ts = lane > 0 && g_ring.lane[lane].stream ? g_ring.lane[lane].stream : ctx().stream;
    if (hipStreamSynchronize(ts) != hipSuccess) return GRB_PANIC;
  }
  g_ring.wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return r;
}
void grb::bfs_host_times(double* enqueue_us, double* wait_us, long long* calls, bool reset) {
  if (enqueue_us) *enqueue_us = g_ring.enqueue_us;
  if (wait_us) *wait_us = g_ring.wait_us;
  if (calls) *calls = g_ring.calls;
  if (reset) { g_ring.enqueue_us = g_ring.wait_us = 0; g_ring.calls = 0; }
}

// Runs the persistent traversal and waits for it.  Outputs mirror what the fused loop keeps on the host.
grb_info bfs_persistent_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int profile,
                            grb_bfs_level* levels_out, int max_levels, int* levels, int* last_dir,
                            long long* reached, unsigned long long* edges, Index* nf_left, bool* hit_cap,
                            float* tight_ms) {
  int slot = 0, seq = 0;
  GRB_TRY(bfs_ticket_take(&slot));
  void* p_rec = nullptr;
  unsigned long long* trace = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  GRB_TRY(bfs_persistent_launch(v, A, source, desc, profile, slot, &seq, &p_rec, &trace));
  const auto t1 = std::chrono::steady_clock::now();
  const grb_info r = bfs_persistent_collect(slot, seq, profile, p_rec, trace, levels_out, max_levels, levels, last_dir, reached,
                                            edges, nf_left, hit_cap, tight_ms);
  const auto t2 = std::chrono::steady_clock::now();
  g_ring.enqueue_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
  g_ring.wait_us += std::chrono::duration<double, std::micro>(t2 - t1).count();
  ++g_ring.calls;
  return r;
}

