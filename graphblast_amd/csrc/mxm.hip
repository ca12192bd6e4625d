// mxm.hip -- masked SpGEMM (the triangle-counting path) and the two
// matrix helpers its driver needs.
//
//   grb_mxm        C<mask> = A (+.x) B for the mask's nonzeros only: backend/cuda/spgemm.hpp:22-110
//                  + kernels/spgemm.hpp:17-79 (spgemmMaskedKernel).  With GrB_INP1 = GrB_TRAN the
//                  "columns of B" are B's CSR rows, so C[i,j] = (+)_k A[i,k] (x) B[j,k] -- for
//                  B = A = L this is |N(i) n N(j)| on every edge of L, whose sum is the triangle
//                  count.  The reference gives a 32-lane warp to every row and binary-searches
//                  every A entry in the B column; here the mask entries are one flat range, a
//                  64-lane wave takes 64 consecutive entries, short dot products are done by
//                  a lane each and long ones by the whole wave (walk the shorter list, binary
//                  search the longer: min(d_i, d_j) * log max(d_i, d_j) per dot product).
//                  A's own row pointers are used (the
//                  reference walks A with the MASK's row pointers, kernels/spgemm.hpp:35-36,51-56,
//                  which is only meaningful when the two share structure -- as they do in tc()).
//   grb_matrix_tril  lower triangle on the host, as the reference (tri.hpp:21-48, sequential only)
//   grb_reduce_matrix_scalar  reduce.hpp:81-91
#include "common.hpp"

#include <algorithm>

namespace grb {

constexpr int kLaneDotMax = 16;     // dot products whose shorter list is longer go to the whole wave

// row of every mask entry (the entries are processed as one flat, evenly split range)
__global__ void entry_rows_kernel(const Index* __restrict__ ptr, Index nrows, Index nvals, Index* __restrict__ row_of) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index e = (Index)blockIdx.x * blockDim.x + threadIdx.x; e < nvals; e += stride) {
    Index lo = 0, hi = nrows;               // largest r with ptr[r] <= e
    while (hi - lo > 1) {
      const Index mid = lo + ((hi - lo) >> 1);
      if (ptr[mid] <= e) lo = mid; else hi = mid;
    }
    row_of[e] = lo;
  }
}

// sum of int32 entries in 64 bits (the triangle count of a com-Orkut-sized graph does not fit an int)
__global__ __launch_bounds__(kBlock) void sum_i32_wide_kernel(const int* __restrict__ d, Index n, long long* __restrict__ out) {
  long long acc = 0;
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += d[i];
  acc = wave_reduce(acc, [](long long x, long long y) { return x + y; });
  if (lane_id() == 0 && acc) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)acc);
}

__device__ inline Index lower_bound_dev(const Index* __restrict__ a, Index lo, Index hi, Index key) {
  while (lo < hi) {
    const Index mid = lo + ((hi - lo) >> 1);
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// One 64-lane wave takes 64 consecutive mask entries.  A lane computes its own dot product
// when the shorter of the two lists has at most kLaneDotMax entries (walk it, binary-search
// the longer one, searches only move right); the others are done one after the other by the
// whole wave: the shorter list spread over the lanes, each lane binary-searching the longer,
// partial sums folded with the semiring's add.  Work per dot product is
// min(d_i, d_j) * log max(d_i, d_j) either way, but no lane of a wave is left walking a hub
// row alone (RMAT-19 triangle count: 1074 ms with a lane per entry -> see DESIGN.md).
// Whether the products of one entry may be folded in any order: true commutative, associative monoids.  The
// comparison "monoids" of stddef.hpp (and whatever an application registers) are folded exactly as spgemmMasked's
// loop does -- add(mul(a, b), acc) over the common columns in ascending order -- by one lane per entry.
template <int SR>
constexpr bool mxm_order_free() {
  if constexpr (SR == GRB_RUNTIME_SR) return false;
  else {
    constexpr int op = MonoidTraits<SemiringTraits<SR>::monoid>::op;
    return op == OP_PLUS || op == OP_TIMES || op == OP_MIN || op == OP_MAX || op == OP_LOR || op == OP_LAND;
  }
}

template <int SR>
constexpr bool mxm_plus_monoid() {
  if constexpr (SR == GRB_RUNTIME_SR) return false;
  else return MonoidTraits<SemiringTraits<SR>::monoid>::op == OP_PLUS;
}

template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void spgemm_masked_kernel(
    T* __restrict__ c_val, const Index* __restrict__ m_row, const Index* __restrict__ m_ind,
    const void* __restrict__ m_val, int mask_f32, const Index* __restrict__ a_ptr, const Index* __restrict__ a_ind,
    const T* __restrict__ a_val, const Index* __restrict__ b_ptr, const Index* __restrict__ b_ind,
    const T* __restrict__ b_val, Index nvals, int only_b_longer /* 1: only the entries whose row of B is longer than
                                                                   their row of A (the others were done elsewhere) */) {
  typedef Semiring<SR, T> S;
  const int lane = lane_id();
  const Index wave_global = (Index)blockIdx.x * kWavesPerBlock + wave_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index base = wave_global * kWave; base < nvals; base += nwaves * kWave) {
    const Index e = base + lane;
    bool mine = e < nvals;
    if (mine && only_b_longer) {
      const Index row = m_row[e], col = m_ind[e];
      mine = b_ptr[col + 1] - b_ptr[col] > a_ptr[row + 1] - a_ptr[row];
    }
    const bool valid = mine && mask_nonzero(m_val, mask_f32, e);
    Index ss = 0, se = 0, ls = 0, le = 0;
    bool a_short = true;
    if (valid) {
      const Index row = m_row[e], col = m_ind[e];
      const Index as = a_ptr[row], ae = a_ptr[row + 1];
      const Index bs = b_ptr[col], be = b_ptr[col + 1];
      a_short = (ae - as) <= (be - bs);
      ss = a_short ? as : bs; se = a_short ? ae : be;
      ls = a_short ? bs : as; le = a_short ? be : ae;
    }
    T acc = S::identity();
    const bool heavy = mxm_order_free<SR>() && valid && (se - ss) > kLaneDotMax;
    if (valid && !heavy) {
      const Index* s_ind = a_short ? a_ind : b_ind;
      const Index* l_ind = a_short ? b_ind : a_ind;
      Index hint = ls;
      for (Index p = ss; p < se && hint < le; ++p) {
        const Index key = s_ind[p];
        const Index lo = lower_bound_dev(l_ind, hint, le, key);
        hint = lo;
        if (lo < le && l_ind[lo] == key) {
          const T av = a_short ? a_val[p] : a_val[lo];
          const T bv = a_short ? b_val[lo] : b_val[p];
          acc = S::add(S::mul(av, bv), acc);
        }
      }
    }
    unsigned long long todo = __ballot(heavy);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const Index css = __shfl(ss, src, kWave), cse = __shfl(se, src, kWave);
      const Index cls = __shfl(ls, src, kWave), cle = __shfl(le, src, kWave);
      const bool c_a_short = __shfl((int)a_short, src, kWave) != 0;
      const Index* s_ind = c_a_short ? a_ind : b_ind;
      const Index* l_ind = c_a_short ? b_ind : a_ind;
      T part = S::identity();
      for (Index p = css + lane; p < cse; p += kWave) {
        const Index key = s_ind[p];
        const Index lo = lower_bound_dev(l_ind, cls, cle, key);
        if (lo < cle && l_ind[lo] == key) {
          const T av = c_a_short ? a_val[p] : a_val[lo];
          const T bv = c_a_short ? b_val[lo] : b_val[p];
          part = S::add(S::mul(av, bv), part);
        }
      }
      part = wave_reduce(part, [](T x, T y) { return S::add(x, y); });
      if (lane == src) acc = part;
    }
    if (mine) c_val[e] = acc;
  }
}

// ---- the pivot-driven form ("wavefront per row, the row of L cached in LDS") ---------------------
// The entry-driven kernel above walks the shorter list of a mask entry and binary-searches the longer one in global
// memory, every lane on its own chain of dependent loads.  Here the LONGER list of an entry is a pivot that goes into
// an LDS hash table (column -> value) once, and the shorter lists of all the entries that share the pivot are
// streamed past it: coalesced reads, one LDS probe per element, no searching.
//   pass 1, pivot = row i of A      the entries (i, j) of the mask's row i whose row j of B is not longer than row i
//   pass 2, pivot = row j of B      the entries (i, j) of the mask's COLUMN j (its CSC) whose row i of A is shorter
// so every entry is done exactly once, at min(d_i, d_j) probes.  Inside a pivot the elements of all its partner lists
// form ONE index space dealt to the lanes (prefix sums of the partner lengths in LDS): a lane's work does not depend
// on how long "its" list is, consecutive lanes read consecutive elements, and the per-entry sums are accumulated in
// LDS with the monoid's atomic.  Pivots of up to kWaveCap entries are a wave's (64 partners per tile); longer ones
// take a 1024-thread workgroup (1024 partners per tile) whose table is 64 KiB of LDS (two workgroups per CU): 4096
// (key, value) slots -- or 8192 keys when the pivot side holds one value throughout, as a pattern matrix does -- and
// pivots beyond 128 KiB of table are taken in column-range segments.  Without a CSC of the mask pass 2 falls back to the
// entry-driven kernel for its entries.
#ifndef GRB_TC_EXP
#define GRB_TC_EXP 0
#endif
#ifndef GRB_TC_COMPACT
#define GRB_TC_COMPACT 1                   // the keys the filter lets through are queued and probed 64 at a time (0: each where it is found)
#endif
#ifndef GRB_TC_FILTER
#define GRB_TC_FILTER 1                    // key-only tables: a 16 KiB Bloom filter in front of the table
#endif
#ifndef GRB_TC_LOAD_INV
#define GRB_TC_LOAD_INV 4                  // table slots per pivot entry aimed at (2 = half load is the guaranteed minimum)
#endif
#ifndef GRB_TC_DEPTH
#define GRB_TC_DEPTH 2                     // chunks of a partner stream in flight behind the one being probed
#endif
#ifndef GRB_TC_WAVE_CAP
#define GRB_TC_WAVE_CAP 256
#endif
constexpr int kWaveCap = GRB_TC_WAVE_CAP;  // pivot entries a wave's table holds (1024 slots x 8 B = 8 KiB per wave)
constexpr int kWaveSlots = 2 * kWaveCap;
constexpr unsigned int kEmptyKey = 0xffffffffu;

struct HashSlot { unsigned int key; unsigned int val; };
struct KeySlot { unsigned int key; };                      // the pivot side's values are all equal: nothing to store

__device__ inline unsigned int tc_hash(unsigned int c) { return c * 2654435761u; }

// Tables are probed sixteen bytes at a time (four keys, or two (key, value) pairs): a probe is one LDS round trip and
// almost always the only one -- slot by slot, the lanes of a wave wait for the longest chain among them, and that
// chain of dependent LDS reads, not any memory traffic, was what bounded the kernel.  Insertion fills the first free
// slot of the key's group, then of the following groups; a lookup stops at the first group with a free slot.
typedef unsigned int TcWord4 __attribute__((ext_vector_type(4)));

__device__ inline void tc_insert(HashSlot* tab, unsigned int mask, unsigned int col, unsigned int vbits) {
  unsigned int s = (tc_hash(col) >> 7) & mask & ~1u;
  for (;;) {
    for (int j = 0; j < 2; ++j) {
      const unsigned int old = atomicCAS(&tab[s + j].key, kEmptyKey, col);
      if (old == kEmptyKey || old == col) {
        __hip_atomic_store(&tab[s + j].val, vbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
    }
    s = (s + 2) & mask;
  }
}
__device__ inline void tc_insert(KeySlot* tab, unsigned int mask, unsigned int col, unsigned int) {
  unsigned int s = (tc_hash(col) >> 7) & mask & ~3u;
  for (;;) {
    for (int j = 0; j < 4; ++j) {
      const unsigned int old = atomicCAS(&tab[s + j].key, kEmptyKey, col);
      if (old == kEmptyKey || old == col) return;
    }
    s = (s + 4) & mask;
  }
}
__device__ inline TcWord4 tc_group(const void* tab, unsigned int slot_bytes_index) {
  return *reinterpret_cast<const TcWord4*>(reinterpret_cast<const char*>(tab) + slot_bytes_index);
}
__device__ inline bool tc_find(const KeySlot* tab, unsigned int mask, unsigned int col, unsigned int*) {
  unsigned int s = (tc_hash(col) >> 7) & mask & ~3u;
  for (;;) {
    const TcWord4 g = tc_group(tab, s * 4u);
    if (g.x == col || g.y == col || g.z == col || g.w == col) return true;
    if (g.x == kEmptyKey || g.y == kEmptyKey || g.z == kEmptyKey || g.w == kEmptyKey) return false;
    s = (s + 4) & mask;
  }
}
__device__ inline bool tc_find(const HashSlot* tab, unsigned int mask, unsigned int col, unsigned int* vbits) {
  unsigned int s = (tc_hash(col) >> 7) & mask & ~1u;
  for (;;) {
    const TcWord4 g = tc_group(tab, s * 8u);
    if (g.x == col) { *vbits = g.y; return true; }
    if (g.z == col) { *vbits = g.w; return true; }
    if (g.x == kEmptyKey || g.z == kEmptyKey) return false;
    s = (s + 2) & mask;
  }
}

// first group of a key, and the verdict of one group: 1 found (value in *vbits), 0 absent, -1 look at the next group
__device__ inline unsigned int tc_home(const KeySlot*, unsigned int mask, unsigned int col) { return (tc_hash(col) >> 7) & mask & ~3u; }
__device__ inline unsigned int tc_home(const HashSlot*, unsigned int mask, unsigned int col) { return (tc_hash(col) >> 7) & mask & ~1u; }
__device__ inline TcWord4 tc_load(const KeySlot* tab, unsigned int s) { return tc_group(tab, s * 4u); }
__device__ inline TcWord4 tc_load(const HashSlot* tab, unsigned int s) { return tc_group(tab, s * 8u); }
__device__ inline int tc_verdict(const KeySlot*, TcWord4 g, unsigned int col, unsigned int*) {
  if (g.x == col || g.y == col || g.z == col || g.w == col) return 1;
  if (g.x == kEmptyKey || g.y == kEmptyKey || g.z == kEmptyKey || g.w == kEmptyKey) return 0;
  return -1;
}
__device__ inline int tc_verdict(const HashSlot*, TcWord4 g, unsigned int col, unsigned int* vbits) {
  if (g.x == col) { *vbits = g.y; return 1; }
  if (g.z == col) { *vbits = g.w; return 1; }
  if (g.x == kEmptyKey || g.z == kEmptyKey) return 0;
  return -1;
}
__device__ inline unsigned int tc_step(const KeySlot*) { return 4u; }
__device__ inline unsigned int tc_step(const HashSlot*) { return 2u; }

// monoid-specific atomic combine of a per-entry sum in LDS
template <int SR, typename T>
__device__ inline void tc_accumulate(T* addr, T v) {
  typedef Semiring<SR, T> S;
  constexpr bool int_sum = [] {
    if constexpr (SR == GRB_RUNTIME_SR) return false;
    else return MonoidTraits<SemiringTraits<SR>::monoid>::op == OP_PLUS && std::is_same<T, int>::value;
  }();
  if constexpr (int_sum) {
    atomicAdd(addr, v);
  } else {
    unsigned int* a = reinterpret_cast<unsigned int*>(addr);
    unsigned int old = *a, assumed;
    do {
      assumed = old;
      T cur;
      memcpy(&cur, &assumed, 4);
      const T nv = S::add(v, cur);
      unsigned int nb;
      memcpy(&nb, &nv, 4);
      old = atomicCAS(a, assumed, nb);
    } while (old != assumed);
  }
}

// The lanes of a wave hold products for partners `key` (non-decreasing across the lanes: they read consecutive
// elements): a segmented scan folds each partner's run inside the wave, and only the last lane of a run touches the
// partner's sum in LDS.  (One atomic per lane instead: most lanes of a wave hit the SAME word, and the LDS serialises
// same-address atomics -- 65 cycles per wave instruction, tools/probes/lds_atomic_probe.hip.)
template <int SR, typename T>
__device__ inline void tc_commit_runs(T* acc, int key, T c, bool active, int lane) {
  typedef Semiring<SR, T> S;
  if (!active) key = -1 - lane;                            // a run of its own, never committed
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const T y = __shfl_up(c, o, kWave);
    const int ky = __shfl_up(key, o, kWave);
    if (lane >= o && ky == key) c = S::add(y, c);
  }
  const int knext = __shfl_down(key, 1, kWave);
  if (active && (lane == kWave - 1 || knext != key)) tc_accumulate<SR, T>(&acc[key], c);
}

// What the two passes differ in.  `major` arrays: the pivot's side (pass 1: A, pass 2: B); `minor`: the partners'.
struct PivotView {
  const Index *piv_ptr, *piv_ind;   const void* piv_val;     // pivot lists
  const Index *par_ptr, *par_ind;   const void* par_val;     // partner lists
  const Index *ent_ptr, *ent_ind;                            // the pivot's entries: mask CSR (pass 1) / CSC (pass 2)
  const Index *m_ptr, *m_ind;       const void* m_val;       // the mask's CSR (output positions, mask values)
  int mask_f32;
  int cols;                                                  // 0: pass 1, 1: pass 2
  unsigned int iso_bits;                                     // the pivot side's one value (key-only tables)
  int par_iso;                                               // the partner side holds one value throughout ...
  unsigned int par_iso_bits;                                 // ... this one: its value array is not read
};

// entry t of pivot r: is it this pass's, its partner list, its position in C
__device__ inline bool tc_entry_of(const PivotView& v, Index r, Index t, Index dpiv, Index* ps, Index* pe, Index* out) {
  const Index other = v.ent_ind[t];
  const Index s = v.par_ptr[other], e = v.par_ptr[other + 1];
  Index pos = t;
  if (v.cols) {                                            // (i = other, j = r): where is j in the mask's row i
    if (!(e - s < dpiv)) return false;                     // pass 1 has it
    pos = lower_bound_dev(v.m_ind, v.m_ptr[other], v.m_ptr[other + 1], r);
  } else if (!(e - s <= dpiv)) {
    return false;
  }
  if (!mask_nonzero(v.m_val, v.mask_f32, pos)) return false;
  *ps = s; *pe = e; *out = pos;
  return true;
}

// c = C's values; entries this pass does not own are left alone (the other pass, or the initialisation, has them)
// kKeyOnly: the pivot side holds one value throughout (a pattern matrix): the table stores keys only -- twice the
// slots in the same LDS -- and, with an integer plus-monoid and a one-valued partner side as well (triangle
// counting), the keys that pass the filter are QUEUED per wave with the partner they belong to and probed 64 at a
// time with every lane active, a hit being one ds_add_u32 on the partner's counter (as in the block kernel below):
// no segmented scan per 64 elements, no table probe with a sixth of the lanes.
template <int SR, typename T, bool kKeyOnly>
__global__ __launch_bounds__(kBlock) void spgemm_pivot_wave_kernel(T* __restrict__ c_val, PivotView v, Index npivots) {
  typedef Semiring<SR, T> S;
  typedef typename std::conditional<kKeyOnly, KeySlot, HashSlot>::type WSlot;
  constexpr int kWSlots = kKeyOnly ? 2 * kWaveSlots : kWaveSlots;      // the same 8 KiB per wave
  constexpr bool kCompactW = GRB_TC_COMPACT != 0 && GRB_TC_FILTER != 0 && kKeyOnly && std::is_integral<T>::value && mxm_plus_monoid<SR>();
  constexpr int kCqW = kCompactW ? 2 * kWave : 1;
  __shared__ unsigned int s_ckey[kCompactW ? kWavesPerBlock : 1][kCqW];
  __shared__ unsigned char s_csrc[kCompactW ? kWavesPerBlock : 1][kCqW];
  __shared__ unsigned int s_cnt[kCompactW ? kWavesPerBlock : 1][kCompactW ? kWave : 1];
  __shared__ WSlot s_tab[kWavesPerBlock][kWSlots];
  constexpr bool kFilter = GRB_TC_FILTER != 0;             // a Bloom filter in front of the table, as in the block kernel
  constexpr int kFiltWords = kFilter ? kWaveCap / 2 : 1;   // 16 bits per key at capacity
  __shared__ unsigned int s_filt[kWavesPerBlock][kFiltWords];
  __shared__ Index s_off[kWavesPerBlock][kWave + 1];
  __shared__ Index s_start[kWavesPerBlock][kWave];
  __shared__ T s_acc[kWavesPerBlock][kWave];
  const int lane = lane_id(), wave = wave_id();
  WSlot* tab = s_tab[wave];
  Index* off = s_off[wave];
  const T* __restrict__ par_val = reinterpret_cast<const T*>(v.par_val);
  const T* __restrict__ piv_val = reinterpret_cast<const T*>(v.piv_val);
  T par_one;
  memcpy(&par_one, &v.par_iso_bits, 4);
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index r = (Index)blockIdx.x * kWavesPerBlock + wave; r < npivots; r += nwaves) {
    const Index es = v.ent_ptr[r], ee = v.ent_ptr[r + 1];
    if (ee == es) continue;
    const Index as = v.piv_ptr[r], ae = v.piv_ptr[r + 1];
    const Index da = ae - as;
    if (da > kWaveCap || da == 0) continue;               // the workgroup kernel's / nothing to intersect with
    bool built = false;
    unsigned int tmask = 0, fmask = 0;
    unsigned int* filt = s_filt[wave];
    for (Index t0 = es; t0 < ee; t0 += kWave) {
      // ---- this tile's partners: one per lane
      const Index t = t0 + lane;
      Index ps = 0, pe = 0, out = 0;
      const bool mine = t < ee && tc_entry_of(v, r, t, da, &ps, &pe, &out);
      const Index len = mine ? pe - ps : 0;
      Index incl = len;
incl = (Index)wave_incl_scan_u32((unsigned)incl);
      const Index total = (Index)__builtin_amdgcn_readlane((int)incl, kWave - 1);
      if (__ballot(mine) == 0ull) continue;
      if (!built) {                                        // the pivot's table, once, and only if somebody needs it
        unsigned int slots = 64;
        while ((Index)slots < (kKeyOnly ? 4 : 2) * da) slots <<= 1;   // key-only: a quarter load (the block kernel's reasoning)
        tmask = slots - 1;
        for (unsigned int i = lane; i < slots; i += kWave) tab[i].key = kEmptyKey;
        if constexpr (kFilter) {
          unsigned int fwords = 16;
          while (2 * (Index)fwords < da && fwords < (unsigned int)kFiltWords) fwords <<= 1;
          fmask = fwords - 1;
          for (unsigned int i = lane; i < fwords; i += kWave) filt[i] = 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (Index p = as + lane; p < ae; p += kWave) {
          unsigned int vb;
          const T av = piv_val[p];
          memcpy(&vb, &av, 4);
          const unsigned int col = (unsigned int)v.piv_ind[p];
          tc_insert(tab, tmask, col, vb);
          if constexpr (kFilter) {
            const unsigned int hh = tc_hash(col);
            atomicOr(&filt[(hh >> 12) & fmask], (1u << (hh & 31u)) | (1u << ((hh >> 5) & 31u)));
          }
        }
        built = true;
      }
      off[lane] = incl - len;
      if (lane == kWave - 1) off[kWave] = total;
      s_start[wave][lane] = ps;
      s_acc[wave][lane] = S::identity();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- the partners' elements as one index space
      if constexpr (kCompactW) {
        if (v.par_iso != 0) {                                // (wave-uniform)
          s_cnt[wave][lane] = 0u;
          int qlen = 0;
          auto probe_round = [&](int na) {                   // candidates [qlen - na, qlen) of the queue, one per lane
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool on = lane < na;
            const unsigned int key = on ? s_ckey[wave][qlen - na + lane] : 0u;
            const unsigned int src = on ? (unsigned int)s_csrc[wave][qlen - na + lane] : 0u;
            if (on && tc_find(tab, tmask, key, nullptr)) atomicAdd(&s_cnt[wave][src], 1u);
            qlen -= na;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          };
          for (Index x0 = 0; x0 < total; x0 += kWave) {
            const Index x = x0 + lane;
            const bool live = x < total;
            int lo = 0;                                      // last partner whose offset is <= x
            bool pass = false;
            unsigned int col = 0u;
            if (live) {
#pragma unroll
              for (int step = kWave / 2; step > 0; step >>= 1)
                if (off[lo + step] <= x) lo += step;
              col = (unsigned int)v.par_ind[s_start[wave][lo] + (x - off[lo])];
              const unsigned int hh = tc_hash(col);
              const unsigned int pat = (1u << (hh & 31u)) | (1u << ((hh >> 5) & 31u));
              pass = (filt[(hh >> 12) & fmask] & pat) == pat;
            }
            const unsigned long long pm = __ballot(pass);
            if (pm) {
              if (pass) {
                const int at = qlen + __popcll(pm & ((1ull << lane) - 1ull));
                s_ckey[wave][at] = col;
                s_csrc[wave][at] = (unsigned char)lo;
              }
              qlen += __popcll(pm);
              if (qlen >= kWave) probe_round(kWave);
            }
          }
          if (qlen > 0) probe_round(qlen);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          T pv;
          const unsigned int vb1 = v.iso_bits;
          memcpy(&pv, &vb1, 4);
          const T one = v.cols ? S::mul(par_one, pv) : S::mul(pv, par_one);
          if (mine) c_val[out] = (T)s_cnt[wave][lane] * one;   // an integer sum of `count` equal products
          __builtin_amdgcn_wave_barrier();
          continue;
        }
      }
      for (Index x0 = 0; x0 < total; x0 += kWave) {
        const Index x = x0 + lane;
        const bool live = x < total;
        int lo = 0;                                        // last partner whose offset is <= x
        T prod = S::identity();
        if (live) {
#pragma unroll
          for (int step = kWave / 2; step > 0; step >>= 1)
            if (off[lo + step] <= x) lo += step;
          const Index q = s_start[wave][lo] + (x - off[lo]);
          unsigned int vb;
          const unsigned int col = (unsigned int)v.par_ind[q];
          bool maybe = true;
          if constexpr (kFilter) {
            const unsigned int hh = tc_hash(col);
            const unsigned int pat = (1u << (hh & 31u)) | (1u << ((hh >> 5) & 31u));
            maybe = (filt[(hh >> 12) & fmask] & pat) == pat;
          }
          vb = v.iso_bits;                                   // (a key-only table stores no value: the pivot side's one)
          if (maybe && tc_find(tab, tmask, col, &vb)) {
            T pv;
            memcpy(&pv, &vb, 4);
            const T bv = v.par_iso ? par_one : par_val[q];
            prod = v.cols ? S::mul(bv, pv) : S::mul(pv, bv);
          }
        }
        tc_commit_runs<SR, T>(s_acc[wave], lo, prod, live, lane);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (mine) c_val[out] = s_acc[wave][lane];
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// pivots longer than kWaveCap: a 1024-thread workgroup per run of entries (listed in big), table in LDS; a pivot of
// more than kCap entries is taken kCap entries at a time (kSegments)
struct PivotItem { Index pivot, e0, e1; };                // a run of a long pivot's entries (whole tiles of 1024)

template <int SR, typename T, bool kSegments, typename Slot, int kTableBytes>
__global__ __launch_bounds__(1024) void spgemm_pivot_block_kernel(T* __restrict__ c_val, PivotView v,
                                                                  const PivotItem* __restrict__ big, int nbig,
                                                                  Index min_len, Index max_len,
                                                                  unsigned long long* __restrict__ trace) {
  typedef Semiring<SR, T> S;
  // kTableBytes of LDS table at half load: 64 KiB = two workgroups per CU; the 128 KiB instantiation takes the
  // pivots too long for that one, and with kSegments those too long for any LDS table
  constexpr int kCap = kTableBytes / (int)sizeof(Slot) / 2;
  const unsigned long long t_begin = wall_clock64();
  unsigned long long t_items = 0;
  int n_items = 0;
  __shared__ Slot s_tab[2 * kCap];
  // Most probes miss (RMAT-22 ef 28: one in seven finds its key), and what this kernel waits for is the LDS executing
  // a wave's random 16-byte reads (GRB_TC_EXP: the probes, not the stream, are its time).  A Bloom filter in front of
  // the table settles the misses with one 4-byte read: two bits of one word per key, 16 bits per key at the
  // capacity of the 64 KiB table (8 at the 128 KiB one's), so a few percent of the misses still look at the table.
  constexpr bool kFilter = GRB_TC_FILTER != 0;
  constexpr int kFiltWords = kFilter ? 4096 : 1;
  __shared__ unsigned int s_filt[kFiltWords];
  // GRB_TC_COMPACT: one key in six passes the filter, and an LDS instruction with a sixth of its lanes active costs
  // nearly what a full one costs (GRB_TC_EXP=3).  So the keys that pass are QUEUED per wave, with the partner they
  // belong to, and probed 64 at a time with every lane active; a hit is one ds_add_u32 on the partner's counter.
  // Key-only tables, an integer plus-monoid and a partner side that holds one value: an entry's result is then
  // count x (the one product).
  constexpr bool kCompact = GRB_TC_COMPACT != 0 && kFilter && std::is_same<Slot, KeySlot>::value && std::is_integral<T>::value &&
                            mxm_plus_monoid<SR>();
  constexpr int kCq = kCompact ? 2 * kWave : 1;
  __shared__ unsigned int s_ckey[kCompact ? 16 : 1][kCq];
  __shared__ unsigned char s_csrc[kCompact ? 16 : 1][kCq];
  __shared__ unsigned int s_cnt[kCompact ? 16 : 1][kCompact ? kWave : 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* __restrict__ par_val = reinterpret_cast<const T*>(v.par_val);
  const T* __restrict__ piv_val = reinterpret_cast<const T*>(v.piv_val);
  T par_one;
  memcpy(&par_one, &v.par_iso_bits, 4);
  for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const Index r = big[bi].pivot;
    const Index es = big[bi].e0, ee = big[bi].e1;
    const Index as = v.piv_ptr[r], ae = v.piv_ptr[r + 1];
    const Index da = ae - as;
    if (da > max_len || (kSegments ? da <= min_len : (da > kCap || da <= min_len))) continue;   // another instantiation's / kernel's
    const unsigned long long t_item = wall_clock64();
    ++n_items;
    Slot* tab = s_tab;
    // A pivot too long for the table is taken in SEGMENTS of kCap consecutive entries (= a column range, the lists
    // are sorted): the table holds one segment, every partner is restricted to that column range by two binary
    // searches, and an entry's result is folded over the segments.  (The first form of this probed a table in
    // global memory instead: 251 GB of HBM traffic per launch on RMAT-22 ef 28 -- it ran at the HBM rate on hash
    // probes, 97 ms for a few hundred rows; in segments 45 ms.)
    const Index nseg = kSegments ? (da + kCap - 1) / kCap : 1;
    for (Index seg = 0; seg < nseg; ++seg) {
    const Index seg_s = as + seg * (Index)kCap;
    const Index seg_e = (kSegments && seg_s + kCap < ae) ? seg_s + (Index)kCap : ae;
    const Index c_lo = kSegments ? v.piv_ind[seg_s] : 0, c_hi = kSegments ? v.piv_ind[seg_e - 1] : 0;
    // the table at a QUARTER load where the pivot allows it (half load is what the capacity guarantees): with four keys
    // per group and two expected, one group in seven is full and a probe walks on -- a dependent LDS read the whole wave
    // waits for, for each of its four keys in turn.  The isolating runs (GRB_TC_EXP) say the probes, not the stream,
    // are this kernel's time, and the LDS executes a wave's random 16-byte reads at ~70 cycles each.
    unsigned int slots = 1024;
    while ((Index)slots < GRB_TC_LOAD_INV * (seg_e - seg_s) && slots < 2u * (unsigned int)kCap) slots <<= 1;
    while ((Index)slots < 2 * (seg_e - seg_s)) slots <<= 1;
    const unsigned int tmask = slots - 1;
    unsigned int fwords = 64;
    while (kFilter && 2 * (Index)fwords < seg_e - seg_s && fwords < (unsigned int)kFiltWords) fwords <<= 1;   // 16 bits per key
    const unsigned int fmask = fwords - 1;
    __syncthreads();
    for (unsigned int i = tid; i < slots; i += 1024) tab[i].key = kEmptyKey;
    if constexpr (kFilter)
      for (unsigned int i = tid; i < fwords; i += 1024) s_filt[i] = 0u;
    __syncthreads();
    for (Index p = seg_s + tid; p < seg_e; p += 1024) {
      unsigned int vb;
      const T av = piv_val[p];
      memcpy(&vb, &av, 4);
      tc_insert(tab, tmask, (unsigned int)v.piv_ind[p], vb);
      if constexpr (kFilter) {
        const unsigned int hh = tc_hash((unsigned int)v.piv_ind[p]);
        atomicOr(&s_filt[(hh >> 12) & fmask], (1u << (hh & 31u)) | (1u << ((hh >> 5) & 31u)));
      }
    }
    __syncthreads();
    // A wave per partner: the pivots here are long, and so are their partners on average (RMAT-22 ef 28: 450
    // entries) -- the lanes stride one list, count in registers, fold once.  No LDS bookkeeping, no atomics: the
    // flat-index form this replaces issued ~400 instructions per 64 elements and was bound by that.  The partners'
    // descriptors (list bounds, output position: three to twenty dependent loads each) are fetched 64 at a time,
    // one per lane, and handed round with shuffles -- fetched by the wave as scalars they cost a latency chain each.
    for (Index t0 = es + wave * kWave; t0 < ee; t0 += 1024) {
      const Index t = t0 + lane;
      Index ps = 0, pe = 0, out = 0;
      const bool mine = t < ee && tc_entry_of(v, r, t, da, &ps, &pe, &out);
      if (kSegments && mine) {                               // the part of the partner inside this segment's columns
        ps = lower_bound_dev(v.par_ind, ps, pe, c_lo);
        pe = lower_bound_dev(v.par_ind, ps, pe, c_hi + 1);
      }
      T result = S::identity();
      const bool compact = kCompact && v.par_iso != 0;       // wave-uniform
      int qlen = 0;                                          // queued candidates (compact)
      if constexpr (kCompact) {
        if (compact) {
          s_cnt[wave][lane] = 0u;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      // candidates [qlen - na, qlen) of the wave's queue, one per lane
      auto probe_round = [&](int na) {
        if constexpr (kCompact) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const bool on = lane < na;
          const unsigned int key = on ? s_ckey[wave][qlen - na + lane] : 0u;
          const unsigned int src = on ? (unsigned int)s_csrc[wave][qlen - na + lane] : 0u;
          if (on && tc_find(reinterpret_cast<const KeySlot*>(tab), tmask, key, nullptr)) atomicAdd(&s_cnt[wave][src], 1u);
          qlen -= na;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      };
      // The partners' lists as one sequence of 256-element chunks, software-pipelined: the keys of the next chunk
      // (the same partner's, or the next partner's first) are in flight while the current chunk is probed -- a
      // partner is a couple of dependent memory steps otherwise, and a wave does hundreds of them one after the other.
      unsigned long long todo = __ballot(mine && pe > ps);
      // chunk descriptors: the partner (lane of the batch) a chunk belongs to (-1: none left), its first element, the
      // partner's end.  GRB_TC_DEPTH chunks are in flight behind the one being probed: every element comes from
      // HBM exactly once (the partners' lists are far larger than the caches), and with one chunk -- 1 KiB -- ahead
      // per wave the stream ran at 0.8 TB/s.
      struct Chunk { int src; Index q, ce; };
      auto advance = [&](Chunk d) -> Chunk {
        if (d.src < 0) return d;
        if (d.q + 4 * kWave < d.ce) { d.q += 4 * kWave; return d; }
        if (todo) {
          d.src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          d.q = __shfl(ps, d.src, kWave);
          d.ce = __shfl(pe, d.src, kWave);
        } else {
          d.src = -1;
        }
        return d;
      };
      auto fetch = [&](const Chunk& d, unsigned int (&k)[4]) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const Index q = d.q + h * kWave + lane;
#if GRB_TC_EXP == 2   // isolating experiment: no stream from memory, the keys are made up (results wrong)
          k[h] = (d.src >= 0 && q < d.ce) ? (unsigned int)(q * 2654435761u) >> 12 : kEmptyKey;
#else
          k[h] = (d.src >= 0 && q < d.ce) ? (unsigned int)v.par_ind[q] : kEmptyKey;   // the empty key is in no table
#endif
        }
      };
      Chunk d0 = {-1, 0, 0};
      if (todo) {
        d0.src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        d0.q = __shfl(ps, d0.src, kWave);
        d0.ce = __shfl(pe, d0.src, kWave);
      }
      constexpr int kDepth = GRB_TC_DEPTH;
      Chunk dq[kDepth + 1];
      unsigned int kq[kDepth + 1][4];
      dq[0] = d0;
      fetch(dq[0], kq[0]);
#pragma unroll
      for (int i = 1; i <= kDepth; ++i) {
        dq[i] = advance(dq[i - 1]);
        fetch(dq[i], kq[i]);
      }
      T acc = S::identity();
      while (dq[0].src >= 0) {
        // the oldest chunk becomes the current one ...
        unsigned int kc[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) kc[h] = kq[0][h];
        const int src_c = dq[0].src;
        const Index q_c = dq[0].q;
        const bool last_of_partner = dq[0].q + 4 * kWave >= dq[0].ce;
        // ... and one more is requested before this one is looked at
#pragma unroll
        for (int i = 0; i < kDepth; ++i) {
          dq[i] = dq[i + 1];
#pragma unroll
          for (int h = 0; h < 4; ++h) kq[i][h] = kq[i + 1][h];
        }
        dq[kDepth] = advance(dq[kDepth]);
        fetch(dq[kDepth], kq[kDepth]);
        // the four keys' home groups are read together (one LDS round trip); a key whose group is full and does
        // not hold it -- rare at half load -- walks on alone
        TcWord4 gr[4];
        unsigned int home[4];
        if constexpr (kFilter) {
          unsigned int fw[4], pat[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const unsigned int hh = tc_hash(kc[h]);
            pat[h] = (1u << (hh & 31u)) | (1u << ((hh >> 5) & 31u));
            fw[h] = s_filt[(hh >> 12) & fmask];
          }
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if ((fw[h] & pat[h]) != pat[h]) kc[h] = kEmptyKey;     // not in the table: as if the slot were empty
          if constexpr (kCompact) {
            if (compact) {
#pragma unroll
              for (int h = 0; h < 4; ++h) {
                const bool pos = kc[h] != kEmptyKey;
                const unsigned long long pm = __ballot(pos);
                if (pm) {
                  if (pos) {
                    const int at = qlen + __popcll(pm & ((1ull << lane) - 1ull));
                    s_ckey[wave][at] = kc[h];
                    s_csrc[wave][at] = (unsigned char)src_c;
                  }
                  qlen += __popcll(pm);
                  if (qlen >= kWave) probe_round(kWave);
                }
              }
              continue;                                      // nothing per partner here: the counters hold the results
            }
          }
#if GRB_TC_EXP == 3   // isolating experiment: the filter's reads alone, nobody goes on to the table (results wrong)
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            if (kc[h] != kEmptyKey && (fw[h] & 1u)) acc = S::add(acc, (T)1);
            kc[h] = kEmptyKey;
          }
#endif
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          home[h] = tc_home(tab, tmask, kc[h] != kEmptyKey ? kc[h] : 0u);
#if GRB_TC_EXP == 1   // isolating experiment: the stream is consumed without probing the table (results wrong)
          gr[h] = TcWord4{kc[h], 0u, 0u, 0u};
#else
          if (!kFilter || kc[h] != kEmptyKey) gr[h] = tc_load(tab, home[h]);
#endif
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (kc[h] == kEmptyKey) continue;
#if GRB_TC_EXP == 1
          if ((kc[h] & 1023u) == 7u) acc = S::add(acc, (T)1);
          continue;
#endif
          unsigned int vb = v.iso_bits;
          int verdict = tc_verdict(tab, gr[h], kc[h], &vb);
          for (unsigned int sl = home[h]; verdict < 0;) {
            sl = (sl + tc_step(tab)) & tmask;
            verdict = tc_verdict(tab, tc_load(tab, sl), kc[h], &vb);
          }
          if (verdict > 0) {
            T pv;
            memcpy(&pv, &vb, 4);
            const T bv = v.par_iso ? par_one : par_val[q_c + h * kWave + lane];
            acc = S::add(v.cols ? S::mul(bv, pv) : S::mul(pv, bv), acc);
          }
        }
        if (last_of_partner) {
          acc = wave_reduce(acc, [](T x, T y) { return S::add(x, y); });
          if (lane == src_c) result = acc;
          acc = S::identity();
        }
      }
      if constexpr (kCompact) {
        if (compact) {
          if (qlen > 0) probe_round(qlen);
          T pv;
          const unsigned int vb = v.iso_bits;
          memcpy(&pv, &vb, 4);
          const T one = v.cols ? S::mul(par_one, pv) : S::mul(pv, par_one);
          result = (T)s_cnt[wave][lane] * one;               // an integer sum of `count` equal products
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (mine) c_val[out] = (kSegments && seg > 0) ? S::add(result, c_val[out]) : result;
    }
    }
    const unsigned long long dt = wall_clock64() - t_item;
    t_items = dt > t_items ? dt : t_items;
  }
  if (trace && threadIdx.x == 0) {
    trace[3 * blockIdx.x] = wall_clock64() - t_begin;
    trace[3 * blockIdx.x + 1] = t_items;
    trace[3 * blockIdx.x + 2] = (unsigned long long)n_items;
  }
}

// ---- long pivots as BITMAPS (triangle counting: both sides one-valued, an integer plus-monoid) -------------------------
// A hash table makes a probe a filter read, a queue write and -- for the one key in six that passes -- a 16-byte table
// read; its size follows the pivot's length, so the longest pivots need 128 KiB tables or several table loads.  A
// bitmap over a COLUMN RANGE needs none of that: kBytes of LDS hold 8 x kBytes columns (128 KiB: one Mi columns -- four
// ranges cover RMAT-22 whatever the pivot's length), a probe is ONE 4-byte LDS read and exact, a hit is a bit.  The
// pivot's entries inside the range set their bits; every partner list is cut to the range by two binary searches (it is
// sorted), streamed through the same software pipeline as the block kernel's, and its hits are counted in registers,
// folded once per partner and range (DPP sum).  An entry's result is the sum over the ranges.
// What a partner costs before its first element is streamed -- its list bounds and output position (tc_entry_of: three
// to twenty dependent loads) and two binary searches per range -- is paid ONCE per item: the first pass over an item's
// entries writes each partner's cut points at all the range boundaries (the searches of one partner run side by side)
// and its output position to a scratch block of the workgroup's, and every range reads them back with coalesced loads.
constexpr int kBitsMaxSeg = 8;                             // ranges per pivot whose cut points are cached (more: searched per range)
constexpr int kBitsItemEntries = 4 * 1024;                 // entries per item (run_pass cuts the lists so)
constexpr int kBitsRec = kBitsMaxSeg + 2;                  // per entry: cut points [0 .. nseg], then the output position (-1: not this pass's)
template <int SR, typename T, int kBytes>
__global__ __launch_bounds__(1024) void spgemm_pivot_bitmap_kernel(T* __restrict__ c_val, PivotView v,
                                                                   const PivotItem* __restrict__ big, int nbig, Index min_len,
                                                                   Index* __restrict__ cuts_all) {
  typedef Semiring<SR, T> S;
  constexpr int kWords = kBytes / 4;
  constexpr long long kCols = (long long)kWords * 32;
  __shared__ unsigned int s_bits[kWords];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T par_one, pv;
  memcpy(&par_one, &v.par_iso_bits, 4);
  memcpy(&pv, &v.iso_bits, 4);
  const T one = v.cols ? S::mul(par_one, pv) : S::mul(pv, par_one);       // every product of this launch
  for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const Index r = big[bi].pivot;
    const Index es = big[bi].e0, ee = big[bi].e1;
    const Index as = v.piv_ptr[r], ae = v.piv_ptr[r + 1];
    const Index da = ae - as;
    if (da <= min_len) continue;                           // the table kernels'
    const long long seg_first = (long long)v.piv_ind[as] / kCols, seg_last = (long long)v.piv_ind[ae - 1] / kCols;
    const int nseg = (int)(seg_last - seg_first + 1);
    const bool cached = nseg <= kBitsMaxSeg && ee - es <= kBitsItemEntries && cuts_all != nullptr;
    Index* cuts = cuts_all + (size_t)blockIdx.x * kBitsItemEntries * kBitsRec;
    if (cached) {
      __syncthreads();                                     // (the previous item's readers are done)
      for (Index t0 = es + wave * kWave; t0 < ee; t0 += 1024) {
        const Index t = t0 + lane;
        Index ps = 0, pe = 0, out = 0;
        const bool mine = t < ee && tc_entry_of(v, r, t, da, &ps, &pe, &out);
        // cut k = the first element with a column >= (seg_first + k) * kCols, k = 0 .. nseg (elements outside the pivot's
        // ranges belong to no range); the searches of one partner run side by side
        Index lo[kBitsMaxSeg + 1], hi[kBitsMaxSeg + 1];
#pragma unroll
        for (int k = 0; k <= kBitsMaxSeg; ++k) { lo[k] = ps; hi[k] = (mine && k <= nseg) ? pe : ps; }
        bool any = true;
        while (any) {
          any = false;
#pragma unroll
          for (int k = 0; k <= kBitsMaxSeg; ++k) {
            if (lo[k] < hi[k]) {
              const Index mid = lo[k] + (hi[k] - lo[k]) / 2;
              const long long key = (seg_first + k) * kCols;
              if ((long long)v.par_ind[mid] < key) lo[k] = mid + 1; else hi[k] = mid;
              any = true;
            }
          }
          any = __ballot(any) != 0ull;
        }
        if (t < ee) {
          Index* rec = cuts + (size_t)(t - es) * kBitsRec;
#pragma unroll
          for (int k = 0; k <= kBitsMaxSeg; ++k) rec[k] = lo[k];
          rec[kBitsMaxSeg + 1] = mine ? out : -1;
        }
      }
      __syncthreads();
    }
    bool wrote = false;                                    // (the same for the whole workgroup)
    for (long long seg = seg_first; seg <= seg_last; ++seg) {
      const long long lo64 = seg * kCols, hi64 = lo64 + kCols;                  // columns [lo, hi)
      const Index c_lo = (Index)lo64;
      const Index c_end = hi64 > 0x7fffffffll ? 0x7fffffff : (Index)hi64;       // (no column id reaches 2^31 - 1)
      const Index p0 = lower_bound_dev(v.piv_ind, as, ae, c_lo);
      const Index p1 = lower_bound_dev(v.piv_ind, p0, ae, c_end);
      if (p1 == p0) continue;
      __syncthreads();
      for (int i = tid; i < kWords / 4; i += 1024) reinterpret_cast<uint4*>(s_bits)[i] = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
      for (Index p = p0 + tid; p < p1; p += 1024) {
        const unsigned int c = (unsigned int)(v.piv_ind[p] - c_lo);
        atomicOr(&s_bits[c >> 5], 1u << (c & 31u));
      }
      __syncthreads();
      for (Index t0 = es + wave * kWave; t0 < ee; t0 += 1024) {
        const Index t = t0 + lane;
        Index ps = 0, pe = 0, out = 0;
        bool mine = false;
        if (cached) {
          if (t < ee) {
            const Index* rec = cuts + (size_t)(t - es) * kBitsRec;
            const int k = (int)(seg - seg_first);
            ps = rec[k];
            pe = rec[k + 1];
            out = rec[kBitsMaxSeg + 1];
            mine = out >= 0;
          }
        } else {
          mine = t < ee && tc_entry_of(v, r, t, da, &ps, &pe, &out);
          if (mine) {                                        // the part of the partner inside this range
            ps = lower_bound_dev(v.par_ind, ps, pe, c_lo);
            pe = lower_bound_dev(v.par_ind, ps, pe, c_end);
          }
        }
        unsigned int result = 0u;
        unsigned long long todo = __ballot(mine && pe > ps);
        struct Chunk { int src; Index q, ce; };
        auto advance = [&](Chunk d) -> Chunk {
          if (d.src < 0) return d;
          if (d.q + 4 * kWave < d.ce) { d.q += 4 * kWave; return d; }
          if (todo) {
            d.src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            d.q = __shfl(ps, d.src, kWave);
            d.ce = __shfl(pe, d.src, kWave);
          } else {
            d.src = -1;
          }
          return d;
        };
        auto fetch = [&](const Chunk& d, unsigned int (&k)[4]) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const Index q = d.q + h * kWave + lane;
            k[h] = (d.src >= 0 && q < d.ce) ? (unsigned int)v.par_ind[q] : kEmptyKey;
          }
        };
        Chunk d0 = {-1, 0, 0};
        if (todo) {
          d0.src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          d0.q = __shfl(ps, d0.src, kWave);
          d0.ce = __shfl(pe, d0.src, kWave);
        }
        constexpr int kDepth = GRB_TC_DEPTH;
        Chunk dq[kDepth + 1];
        unsigned int kq[kDepth + 1][4];
        dq[0] = d0;
        fetch(dq[0], kq[0]);
#pragma unroll
        for (int i = 1; i <= kDepth; ++i) {
          dq[i] = advance(dq[i - 1]);
          fetch(dq[i], kq[i]);
        }
        unsigned int acc = 0u;
        while (dq[0].src >= 0) {
          unsigned int kc[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) kc[h] = kq[0][h];
          const int src_c = dq[0].src;
          const bool last_of_partner = dq[0].q + 4 * kWave >= dq[0].ce;
#pragma unroll
          for (int i = 0; i < kDepth; ++i) {
            dq[i] = dq[i + 1];
#pragma unroll
            for (int h = 0; h < 4; ++h) kq[i][h] = kq[i + 1][h];
          }
          dq[kDepth] = advance(dq[kDepth]);
          fetch(dq[kDepth], kq[kDepth]);
          unsigned int w[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const unsigned int c = kc[h] != kEmptyKey ? kc[h] - (unsigned int)c_lo : 0u;
            w[h] = s_bits[c >> 5] >> (c & 31u);
          }
#pragma unroll
          for (int h = 0; h < 4; ++h) acc += (kc[h] != kEmptyKey) ? (w[h] & 1u) : 0u;
          if (last_of_partner) {
            const unsigned int tot = wave_sum_u32(acc);
            if (lane == src_c) result = tot;
            acc = 0u;
          }
        }
        if (mine) {
          const T add = (T)result * one;                     // an integer sum of `result` equal products
          c_val[out] = wrote ? S::add(add, c_val[out]) : add;
        }
      }
      wrote = true;
    }
  }
}

// min and max of the raw 4-byte values (out preset to {~0, 0}): equal = one value throughout
__global__ __launch_bounds__(kBlock) void value_range_kernel(const unsigned int* __restrict__ val, Index n, unsigned int* __restrict__ out) {
  unsigned int lo = 0xffffffffu, hi = 0u;
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned int x = val[i];
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
  lo = wave_reduce(lo, [](unsigned int a, unsigned int b) { return a < b ? a : b; });
  hi = wave_reduce(hi, [](unsigned int a, unsigned int b) { return a > b ? a : b; });
  if (lane_id() == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

// C's values start as the semiring's identity (an entry neither pass owns -- a zero in the mask -- keeps it)
template <typename T>
__global__ void fill_value_kernel(T* __restrict__ d, Index n, T v) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = v;
}

// Stored values (x) a scalar, or (x) a vector entry picked by the row (by_major) or by the stored
// index (kernels/ewisemult.hpp:160-237: eWiseMultKernel scalar overload, eWiseMultCSRKernel,
// eWiseMultCSCKernel -- 32-lane warp per row there, a 64-lane wave per row here).
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void matrix_scale_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                              T* __restrict__ val, Index nmajor, const T* __restrict__ vec,
                                                              T scalar, int mode /*0 scalar, 1 by major, 2 by minor*/) {
  typedef Semiring<SR, T> S;
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index r = (Index)blockIdx.x * kWavesPerBlock + wave_id(); r < nmajor; r += nwaves) {
    const Index e = ptr[r + 1];
    const T bm = mode == 1 ? vec[r] : scalar;
    for (Index p = ptr[r] + lane; p < e; p += kWave) {
      const T b = mode == 2 ? vec[ind[p]] : bm;
      val[p] = S::mul(val[p], b);
    }
  }
}

// trace(A (+).(x) B^T) = sum over rows i of (+)_k A(i,k) (x) B(i,k)   (traceKernel, kernels/trace.hpp:7-67).
// One wave per row: lanes stride over A's row and binary-search B's; as there, a missing B entry
// contributes mul(a, identity), B's value passes through an Index-typed temporary (truncated towards
// zero, trace.hpp:44-46), rows are folded with the semiring's add and the row sums are then ADDED
// (atomicAdd whatever the semiring, trace.hpp:60-61).
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void trace_kernel(T* __restrict__ out, Index nrows, const Index* __restrict__ a_ptr,
                                                       const Index* __restrict__ a_ind, const T* __restrict__ a_val,
                                                       const Index* __restrict__ b_ptr, const Index* __restrict__ b_ind,
                                                       const T* __restrict__ b_val) {
  using S = Semiring<SR, T>;
  const int lane = threadIdx.x & (kWave - 1);
  const Index wave = (Index)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / kWave);
  const Index nwaves = (Index)((gridDim.x * (unsigned)blockDim.x) / kWave);
  T total = 0;
  for (Index row = wave; row < nrows; row += nwaves) {
    const Index ab = a_ptr[row], ae = a_ptr[row + 1], bb = b_ptr[row], be = b_ptr[row + 1];
    T sum = S::identity();
    for (Index p = ab + lane; p < ae; p += kWave) {
      const Index lo = lower_bound_dev(b_ind, bb, be, a_ind[p]);
      Index bv = (Index)S::identity();
      if (lo < be && b_ind[lo] == a_ind[p]) bv = (Index)b_val[lo];
      sum = S::add(sum, S::mul(a_val[p], (T)bv));
    }
    sum = wave_reduce(sum, [](T x, T y) { return S::add(x, y); });
    total += sum;
  }
  if (lane == 0 && total != 0) atomicAdd(out, total);
}

}  // namespace grb

using namespace grb;

static grb_info matrix_scale(grb_matrix A, int op, grb_vector B, double scalar, bool by_row) {
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (!A->owned) return GRB_INVALID_OBJECT;             // adopted storage belongs to the caller
  hipStream_t s = ctx().stream;
  const void* vec = nullptr;
  if (B) {
    if (B->dtype != A->dtype) return GRB_DOMAIN_MISMATCH;
    if (B->nsize != (by_row ? A->nrows : A->ncols)) return GRB_DIMENSION_MISMATCH;
    if (B->vec_type != GRB_DENSE) return GRB_NOT_IMPLEMENTED;   // callers densify (extractTuples semantics)
    vec = B->d_val;
  }
  grb_info info = dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    for (int o = 0; o < 2; ++o) {
      const CsrArrays& M = o == 0 ? A->csr : A->csc;
      if (!M.ptr || M.nvals == 0) continue;
      // CSR rows are matrix rows, CSC "rows" are matrix columns
      const int mode = !B ? 0 : ((o == 0) == by_row ? 1 : 2);
      hipLaunchKernelGGL((matrix_scale_kernel<SR, T>), dim3(stream_grid((long long)M.n * kWave, kBlock)), dim3(kBlock), 0,
                         s, M.ptr, M.ind, (T*)M.val, M.n, (const T*)vec, (T)scalar, mode);
      GRB_HIP_TRY(hipGetLastError());
    }
    return GRB_SUCCESS;
  });
  A->h_csr_val.clear(); A->h_csr_ind.clear();           // host mirrors are re-read on demand
  A->h_csc_val.clear(); A->h_csc_ind.clear();
  A->nonneg_values = -1; A->mean_value = -1.0; A->small_int_values = -1;
  matrix_values_changed(A);
  return info;
}

extern "C" {

grb_info grb_mxm(grb_matrix C, grb_matrix mask, grb_accum accum, grb_semiring op, grb_matrix A, grb_matrix B,
                 grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!C || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built || !B->built) return GRB_UNINITIALIZED_OBJECT;
  if (!mask) return GRB_NOT_IMPLEMENTED;                 // unmasked SpGEMM is a cuSPARSE call in the reference
  if (!mask->built) return GRB_UNINITIALIZED_OBJECT;
  if (C == A || C == B || C == mask) return GRB_NOT_IMPLEMENTED;
  if (A->dtype != B->dtype || C->dtype != A->dtype) return GRB_DOMAIN_MISMATCH;
  const bool tran_a = desc->desc[GRB_INP0] == GRB_TRAN;
  const bool tran_b = desc->desc[GRB_INP1] == GRB_TRAN;
  const CsrArrays& Aa = tran_a ? A->csc : A->csr;
  const CsrArrays& Bb = tran_b ? B->csr : B->csc;        // "columns of B"
  if (!Aa.ptr || !Bb.ptr || !mask->csr.ptr) return GRB_INVALID_OBJECT;
  if (Aa.n != mask->nrows || C->nrows != mask->nrows || C->ncols != mask->ncols) return GRB_DIMENSION_MISMATCH;
  hipStream_t s = ctx().stream;
  // C takes the mask's structure (C->dup(&mask->sparse_), spgemm.hpp:78-79)
  // whatever C held before goes, with the per-graph side arrays that described it (skip bitmaps,
  // pull hint, plans): a later traversal of C must not see hints of another graph
  // ... but its three arrays stay when C already holds a result of this shape (the same output matrix in a loop of
  // products: three hipFree and three hipMalloc of 4 nvals bytes a call otherwise, each a trip through the driver)
  CsrArrays kept;
#ifndef GRB_MXM_EAGER_RESULT                     // (build flag for the A/B: the round-4 behaviour -- fresh arrays, plan built here)
  if (C->built && C->owned && C != mask && C != A && C != B && !C->csc.ptr && C->csr.ptr && C->csr.ind && C->csr.val &&
      C->nrows == mask->nrows && C->csr.nvals == mask->nvals) {
    kept = C->csr;
    C->csr = CsrArrays();
  }
#endif
  matrix_release_device(C);
  C->owned = true;
  C->nvals = mask->nvals;
  const size_t cap = mask->nvals > 0 ? (size_t)mask->nvals : 1;
  if (kept.ptr) {
    C->csr.ptr = kept.ptr; C->csr.ind = kept.ind; C->csr.val = kept.val;
  } else {
    GRB_HIP_TRY(hipMalloc((void**)&C->csr.ptr, 4 * ((size_t)mask->nrows + 1)));
    GRB_HIP_TRY(hipMalloc((void**)&C->csr.ind, 4 * cap));
    GRB_HIP_TRY(hipMalloc(&C->csr.val, 4 * cap));
  }
  GRB_HIP_TRY(hipMemcpyAsync(C->csr.ptr, mask->csr.ptr, 4 * ((size_t)mask->nrows + 1), hipMemcpyDeviceToDevice, s));
  if (mask->nvals > 0)
    GRB_HIP_TRY(hipMemcpyAsync(C->csr.ind, mask->csr.ind, 4 * (size_t)mask->nvals, hipMemcpyDeviceToDevice, s));
  C->csr.n = mask->nrows;
  C->csr.nvals = mask->nvals;
  C->h_csr_ptr = mask->h_csr_ptr;
  C->h_csr_ind.clear(); C->h_csr_val.clear();
  C->h_csc_ptr.clear(); C->h_csc_ind.clear(); C->h_csc_val.clear();
#ifdef GRB_MXM_EAGER_RESULT
  GRB_TRY(build_spmv_plan(C->h_csr_ptr, C->nrows, C->ncols, &C->plan_csr));
#else
  C->plan_csr_pending = true;                    // mxv on the result works: its plan is built when one is asked for;
#endif
  C->built = true;                               // no CSC is made (as little as the reference's C->dup has one):
  if (mask->nvals == 0) return GRB_SUCCESS;      // products on the transpose return GrB_INVALID_OBJECT
  // ---- pivot-driven passes when the host mirrors of the row pointers are there (they list the long pivots);
  // GRB_MXM_PIVOT=0 keeps the entry-driven kernel alone
  static const bool pivot_ok = [] { const char* e = getenv("GRB_MXM_PIVOT"); return !e || atoi(e) != 0; }();
  const std::vector<Index>& hpa = tran_a ? A->h_csc_ptr : A->h_csr_ptr;
  const std::vector<Index>& hpb = tran_b ? B->h_csr_ptr : B->h_csc_ptr;
  const bool use_pivot = pivot_ok && (Index)hpa.size() == Aa.n + 1 && (Index)hpb.size() == Bb.n + 1 &&
                         (Index)mask->h_csr_ptr.size() == mask->nrows + 1 && Bb.n == mask->ncols;
  const bool have_csc = use_pivot && mask->csc.ptr && !mask->csc_alias && (Index)mask->h_csc_ptr.size() == mask->ncols + 1;
  const int grid = stream_grid(mask->nvals, kBlock);
  void* p_rows = nullptr;
  auto ensure_rows = [&]() -> grb_info {
    if (p_rows) return GRB_SUCCESS;
    GRB_TRY(scratch(9, 4 * (size_t)mask->nvals, &p_rows));            // not 4 / 5: those hold the push path's state
    hipLaunchKernelGGL(entry_rows_kernel, dim3(stream_grid(mask->nvals, kBlock)), dim3(kBlock), 0, s, mask->csr.ptr,
                       mask->nrows, mask->nvals, (Index*)p_rows);
    GRB_HIP_TRY(hipGetLastError());
    return GRB_SUCCESS;
  };
  if (!use_pivot || !have_csc) GRB_TRY(ensure_rows());
  return dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
    using T = decltype(t);
    constexpr int SR = decltype(tag)::value;
    auto entry_driven = [&](int only_b_longer) -> grb_info {
      hipLaunchKernelGGL((spgemm_masked_kernel<SR, T>), dim3(grid), dim3(kBlock), 0, s, (T*)C->csr.val,
                         (const Index*)p_rows, mask->csr.ind, mask->csr.val, mask->dtype == GRB_F32, Aa.ptr, Aa.ind,
                         (const T*)Aa.val, Bb.ptr, Bb.ind, (const T*)Bb.val, mask->nvals, only_b_longer);
      GRB_HIP_TRY(hipGetLastError());
      return GRB_SUCCESS;
    };
    if (!use_pivot) return entry_driven(0);
    if constexpr (!mxm_order_free<SR>()) {                // the fold's order is part of the answer: a lane per entry
      GRB_TRY(ensure_rows());
      return entry_driven(0);
    }
    hipLaunchKernelGGL((fill_value_kernel<T>), dim3(stream_grid(mask->nvals, kBlock)), dim3(kBlock), 0, s, (T*)C->csr.val,
                       mask->nvals, Semiring<SR, T>::identity());
    GRB_HIP_TRY(hipGetLastError());
    // a pass's list is the source of an asynchronous copy: both lists live until the stream has been waited for, once, on
    // the way out (the guard is destroyed first) -- the host's work for the second pass (4 M row lengths, a sort) then runs
    // while the device is busy with the first, not after it
    TcCoreDev core;                                        // (declared first: freed after the stream has been waited for)
    std::vector<PivotItem> big_store[4];
    struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{s};
    int pass_no = 0;
    T* c_out = (T*)C->csr.val;                             // where a pass writes (the dense core's T-part: a buffer of its own)
    auto run_pass = [&](const PivotView& v, Index npiv, bool piv_iso, const std::vector<Index>& hp_piv,
                        const std::vector<Index>& hp_ent, int scratch_list) -> grb_info {
      // the long pivots' entries in runs of <= 4 tiles, heaviest first (a run costs about entries x pivot length:
      // its partners are no longer than the pivot), dealt round-robin: the few giant rows do not become the tail
      std::vector<PivotItem>& big = big_store[pass_no++ & 3];
      big.clear();
      Index longest = 0;
      for (Index r = 0; r < npiv; ++r) {
        const Index d = hp_piv[(size_t)r + 1] - hp_piv[r];
        if (d <= kWaveCap || hp_ent[(size_t)r + 1] <= hp_ent[r]) continue;
        longest = d > longest ? d : longest;
        for (Index e0 = hp_ent[r]; e0 < hp_ent[(size_t)r + 1]; e0 += 4 * 1024) {
          const Index e1 = e0 + 4 * 1024 < hp_ent[(size_t)r + 1] ? e0 + 4 * 1024 : hp_ent[(size_t)r + 1];
          big.push_back(PivotItem{r, e0, e1});
        }
      }
      std::stable_sort(big.begin(), big.end(), [&](const PivotItem& x, const PivotItem& y) {
        const double cx = (double)(x.e1 - x.e0) * (double)(hp_piv[(size_t)x.pivot + 1] - hp_piv[x.pivot]);
        const double cy = (double)(y.e1 - y.e0) * (double)(hp_piv[(size_t)y.pivot + 1] - hp_piv[y.pivot]);
        return cx > cy;
      });
      if (piv_iso)
        hipLaunchKernelGGL((spgemm_pivot_wave_kernel<SR, T, true>), dim3(stream_grid((long long)npiv * kWave, kBlock)), dim3(kBlock), 0, s,
                           c_out, v, npiv);
      else
        hipLaunchKernelGGL((spgemm_pivot_wave_kernel<SR, T, false>), dim3(stream_grid((long long)npiv * kWave, kBlock)), dim3(kBlock), 0, s,
                           c_out, v, npiv);
      GRB_HIP_TRY(hipGetLastError());
      if (big.empty()) return GRB_SUCCESS;
      // two workgroups per CU (64 KiB tables); is the pivot side one value throughout?  then the tables hold keys only
      const int bgrid = (int)big.size() < 2 * ctx().num_cu ? (int)big.size() : 2 * ctx().num_cu;
      void* p_big;
      GRB_TRY(scratch(scratch_list, sizeof(PivotItem) * big.size() + 64, &p_big));
      GRB_HIP_TRY(hipMemcpyAsync(p_big, big.data(), sizeof(PivotItem) * big.size(), hipMemcpyHostToDevice, s));
      const bool iso = piv_iso;
      const PivotView& vv = v;
      auto launch = [&](auto slot_tag) -> grb_info {
        using Slot = decltype(slot_tag);
        static const bool want_trace = getenv("GRB_MXM_TRACE") != nullptr;
        unsigned long long* d_trace = nullptr;
        if (want_trace) {
          void* p_tr;
          GRB_TRY(scratch(2, 24 * (size_t)bgrid, &p_tr));
          d_trace = (unsigned long long*)p_tr;
        }
        auto dump = [&](const char* what) -> grb_info {
          if (!want_trace) return GRB_SUCCESS;
          std::vector<unsigned long long> h(3 * (size_t)bgrid);
          GRB_HIP_TRY(hipMemcpy(h.data(), d_trace, 24 * (size_t)bgrid, hipMemcpyDeviceToHost));
          double sum = 0, mx = 0, mxi = 0, items = 0;
          for (int g = 0; g < bgrid; ++g) {
            sum += (double)h[3 * g]; mx = (double)h[3 * g] > mx ? (double)h[3 * g] : mx;
            mxi = (double)h[3 * g + 1] > mxi ? (double)h[3 * g + 1] : mxi; items += (double)h[3 * g + 2];
          }
          fprintf(stderr, "mxm pivot block kernel (%s, pass %d): %d workgroups, %.0f items; busy ticks (100 MHz) mean %.0f max %.0f; "
                  "longest item %.0f\n", what, vv.cols + 1, bgrid, items, sum / bgrid, mx, mxi);
          return GRB_SUCCESS;
        };
        const Index cap64 = 65536 / (Index)sizeof(Slot) / 2, cap128 = 131072 / (Index)sizeof(Slot) / 2;
        // pivots longer than `bits_from` entries go to the bitmap kernel instead (key-only tables, an integer
        // plus-monoid, a one-valued partner side: triangle counting); GRB_TC_BITMAP_MIN moves the line, 0 = never
        Index max_tab = 0x7fffffff;
        if constexpr (std::is_same<Slot, KeySlot>::value && std::is_integral<T>::value && mxm_plus_monoid<SR>()) {
          static const long long bits_env = getenv("GRB_TC_BITMAP_MIN") ? atoll(getenv("GRB_TC_BITMAP_MIN")) : 2048;
          if (vv.par_iso && bits_env > 0 && longest > (Index)bits_env) max_tab = (Index)bits_env;
        }
        if (want_trace) GRB_HIP_TRY(hipMemsetAsync(d_trace, 0, 24 * (size_t)bgrid, s));
        hipLaunchKernelGGL((spgemm_pivot_block_kernel<SR, T, false, Slot, 65536>), dim3(bgrid), dim3(1024), 0, s, c_out, vv,
                           (const PivotItem*)p_big, (int)big.size(), (Index)0, max_tab, d_trace);
        GRB_HIP_TRY(hipGetLastError());
        GRB_TRY(dump("64 KiB LDS tables"));
        if (longest > cap64 && max_tab > cap64) {
          if (want_trace) GRB_HIP_TRY(hipMemsetAsync(d_trace, 0, 24 * (size_t)bgrid, s));
          hipLaunchKernelGGL((spgemm_pivot_block_kernel<SR, T, false, Slot, 131072>), dim3(bgrid), dim3(1024), 0, s, c_out, vv,
                             (const PivotItem*)p_big, (int)big.size(), cap64, max_tab, d_trace);
          GRB_HIP_TRY(hipGetLastError());
          GRB_TRY(dump("128 KiB LDS tables"));
        }
        if (longest > cap128 && max_tab > cap128) {
          if (want_trace) GRB_HIP_TRY(hipMemsetAsync(d_trace, 0, 24 * (size_t)bgrid, s));
          hipLaunchKernelGGL((spgemm_pivot_block_kernel<SR, T, true, Slot, 131072>), dim3(bgrid), dim3(1024), 0, s, c_out, vv,
                             (const PivotItem*)p_big, (int)big.size(), cap128, max_tab, d_trace);
          GRB_HIP_TRY(hipGetLastError());
          GRB_TRY(dump("128 KiB LDS tables, pivot in segments"));
        }
        if constexpr (std::is_same<Slot, KeySlot>::value && std::is_integral<T>::value && mxm_plus_monoid<SR>()) {
          if (max_tab != 0x7fffffff) {
            // 128 KiB of LDS: one workgroup per CU, four ranges on RMAT-22 (64 KiB -- two per CU, eight ranges -- measured
            // 151 against 134 ms)
            const int ggrid = (int)big.size() < ctx().num_cu ? (int)big.size() : ctx().num_cu;
            void* p_cuts;
            GRB_TRY(scratch(3, sizeof(Index) * (size_t)ggrid * kBitsItemEntries * kBitsRec, &p_cuts));
            hipLaunchKernelGGL((spgemm_pivot_bitmap_kernel<SR, T, 131072>), dim3(ggrid), dim3(1024), 0, s, c_out, vv,
                               (const PivotItem*)p_big, (int)big.size(), max_tab, (Index*)p_cuts);
            GRB_HIP_TRY(hipGetLastError());
          }
        }
        return GRB_SUCCESS;
      };
      if (iso) GRB_TRY(launch(KeySlot{}));
      else GRB_TRY(launch(HashSlot{}));
#ifdef GRB_MXM_EAGER_RESULT
      GRB_HIP_TRY(hipStreamSynchronize(s));               // (round 4: the stream waited for after every pass)
#endif
      return GRB_SUCCESS;
    };
    PivotView v1;
    v1.piv_ptr = Aa.ptr; v1.piv_ind = Aa.ind; v1.piv_val = Aa.val;
    v1.par_ptr = Bb.ptr; v1.par_ind = Bb.ind; v1.par_val = Bb.val;
    v1.ent_ptr = mask->csr.ptr; v1.ent_ind = mask->csr.ind;
    v1.m_ptr = mask->csr.ptr; v1.m_ind = mask->csr.ind; v1.m_val = mask->csr.val;
    v1.mask_f32 = mask->dtype == GRB_F32;
    v1.cols = 0;
    // is a side one value throughout (a pattern matrix)?  then its tables hold keys only and its value array is not read
    unsigned int rng[4] = {0xffffffffu, 0u, 0xffffffffu, 0u};
    {
      void* p_rng;
      GRB_TRY(scratch(3, 64, &p_rng));
      GRB_HIP_TRY(hipMemcpyAsync(p_rng, rng, 16, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(value_range_kernel, dim3(stream_grid(Aa.nvals, kBlock * 8)), dim3(kBlock), 0, s,
                         (const unsigned int*)Aa.val, Aa.nvals, (unsigned int*)p_rng);
      hipLaunchKernelGGL(value_range_kernel, dim3(stream_grid(Bb.nvals, kBlock * 8)), dim3(kBlock), 0, s,
                         (const unsigned int*)Bb.val, Bb.nvals, (unsigned int*)p_rng + 2);
      GRB_HIP_TRY(hipGetLastError());
      GRB_HIP_TRY(hipMemcpyAsync(rng, p_rng, 16, hipMemcpyDeviceToHost, s));
      GRB_HIP_TRY(hipStreamSynchronize(s));
    }
    const bool iso_a = rng[0] == rng[1] && Aa.nvals > 0, iso_b = rng[2] == rng[3] && Bb.nvals > 0;
    v1.iso_bits = rng[0];
    v1.par_iso = iso_b ? 1 : 0;
    v1.par_iso_bits = rng[2];
    // ---- the dense core (mxm_core.hip), for the triangle count's product C<L> = L (+.x) L^T: among the longest rows the
    // lists are dense enough to be bit rows.  An entry (i, j) between two core rows is taken out of the passes over the
    // whole mask (its copy of the mask value is zeroed) and computed as
    //     hits among the core's columns  (the bit rows: AND + popcount per entry, or MFMA on the denser tiles)
    //   + hits outside them              (the same pivot passes, on the two rows' lists WITHOUT the core vertices)
    // OFF by default (GRB_TC_CORE_K = rows of the core; measured on the config-5 stand-in, docs/experiments.md: the
    // product takes 123-140 ms with a core of 2 Ki ... 32 Ki rows against 112 without -- the tail of the degree
    // distribution is long, a core large enough to take a third of the streamed list elements off the pivot kernels is
    // too sparse for bit rows, and a small one does not pay for its set-up).  GRB_TC_CORE_MFMA_FROM moves the line between
    // the two bit-row kernels (entries per 128 x 128 tile; 0 = popcount only, 1 = MFMA only).
    bool use_core = false;
    if constexpr (std::is_same<T, int>::value && mxm_plus_monoid<SR>()) {
      // (read per call, as the reference reads its own environment switches per call: tests move them)
      const int core_k = getenv("GRB_TC_CORE_K") ? atoi(getenv("GRB_TC_CORE_K")) : 0;
      const int core_from = getenv("GRB_TC_CORE_MFMA_FROM") ? atoi(getenv("GRB_TC_CORE_MFMA_FROM")) : 2048;
      const long long core_min = getenv("GRB_TC_CORE_MIN_NVALS") ? atoll(getenv("GRB_TC_CORE_MIN_NVALS")) : (1ll << 22);
      if (core_k > 0 && iso_a && iso_b && have_csc && mask == A && A == B && !tran_a && tran_b && mask->nrows == mask->ncols &&
          mask->dtype == GRB_I32 && (long long)mask->nvals >= core_min) {
        GRB_TRY(tc_core_rows(Aa.ptr, Aa.n, core_k, &core));
        if (core.K >= 2 * 128) {
          void* p_mv;
          GRB_TRY(tc_core_alloc(&core, &p_mv, 4 * (size_t)mask->nvals));   // (not a scratch slot: the push path keeps state in those)
          GRB_HIP_TRY(hipMemcpyAsync(p_mv, mask->csr.val, 4 * (size_t)mask->nvals, hipMemcpyDeviceToDevice, s));
          GRB_TRY(tc_core_bits(Aa.ptr, Aa.ind, &core, true));
          if (core.nent > 0) {
            GRB_TRY(tc_core_tiles(&core, core_from == 0 ? 0 : core_from == 1 ? 1 : 2, core_from, nullptr));
            GRB_TRY(tc_core_split(Aa.ptr, Aa.ind, &core, (unsigned int*)p_mv));
            v1.m_val = p_mv;                               // the passes over the whole mask skip the entries between core rows
            use_core = true;
          }
        }
      }
    }
    auto core_part = [&]() -> grb_info {
      if (!use_core) return GRB_SUCCESS;
      if constexpr (std::is_same<T, int>::value && mxm_plus_monoid<SR>()) {
        const size_t ne = ((size_t)core.nent + 63) & ~(size_t)63;
        void* p_cw;
        GRB_TRY(tc_core_alloc(&core, &p_cw, 4 * 3 * ne + 64));
        int* ch = (int*)p_cw;
        T* ct = (T*)p_cw + ne;
        T* ones = (T*)p_cw + 2 * ne;
        unsigned long long* tot = (unsigned long long*)((T*)p_cw + 3 * ne);
        GRB_HIP_TRY(hipMemsetAsync(ct, 0, 4 * ne, s));       // (the plus-monoid's identity)
        GRB_HIP_TRY(hipMemsetAsync(tot, 0, 16, s));
        hipLaunchKernelGGL((fill_value_kernel<T>), dim3(stream_grid((long long)core.nent, kBlock)), dim3(kBlock), 0, s, ones, (Index)core.nent, (T)1);
        GRB_HIP_TRY(hipGetLastError());
        GRB_TRY(tc_core_hproduct(&core, ch, tot));
        // the lists without the core vertices, the entries between core rows as the mask: the same two passes
        PivotView t1 = v1;
        t1.piv_ptr = (const Index*)core.tptr; t1.piv_ind = core.tind;
        t1.par_ptr = (const Index*)core.tptr; t1.par_ind = core.tind;
        t1.ent_ptr = (const Index*)core.rowstart; t1.ent_ind = core.ccind;
        t1.m_ptr = (const Index*)core.rowstart; t1.m_ind = core.ccind; t1.m_val = ones;
        t1.mask_f32 = 0;
        c_out = ct;
        GRB_TRY(run_pass(t1, (Index)core.K, true, core.h_tptr, core.h_mptr, 6));
        PivotView t2 = t1;
        t2.ent_ptr = (const Index*)core.cscptr; t2.ent_ind = core.cscind;
        t2.cols = 1;
        t2.iso_bits = rng[2];
        t2.par_iso_bits = rng[0];
        GRB_TRY(run_pass(t2, (Index)core.K, true, core.h_tptr, core.h_cscptr, 6));
        c_out = (T*)C->csr.val;
        T a_one, b_one;
        memcpy(&a_one, &rng[0], 4);
        memcpy(&b_one, &rng[2], 4);
        const T one = Semiring<SR, T>::mul(a_one, b_one);
        unsigned int one_bits;
        memcpy(&one_bits, &one, 4);
        GRB_TRY(tc_core_combine(GRB_I32, C->csr.val, &core, ch, ct, one_bits, mask->csr.val, mask->dtype == GRB_F32));
      }
      return GRB_SUCCESS;
    };
    GRB_TRY(run_pass(v1, Aa.n, iso_a, hpa, mask->h_csr_ptr, 6));
    if (!have_csc) return entry_driven(1);                // the entries whose row of B is the longer list
    PivotView v2 = v1;
    v2.piv_ptr = Bb.ptr; v2.piv_ind = Bb.ind; v2.piv_val = Bb.val;
    v2.par_ptr = Aa.ptr; v2.par_ind = Aa.ind; v2.par_val = Aa.val;
    v2.ent_ptr = mask->csc.ptr; v2.ent_ind = mask->csc.ind;
    v2.cols = 1;
    v2.iso_bits = rng[2];
    v2.par_iso = iso_a ? 1 : 0;
    v2.par_iso_bits = rng[0];
    GRB_TRY(run_pass(v2, Bb.n, iso_b, hpb, mask->h_csc_ptr, 6));
    return core_part();
  });
}

// eWiseMult, matrix (x) broadcast scalar (operations.hpp:206-228), in place (C == A)
grb_info grb_matrix_eWiseMult_scalar(grb_matrix C, grb_semiring op, grb_matrix A, double val) { GRB_API_ENTER();
  if (!C || !A) return GRB_UNINITIALIZED_OBJECT;
  if (C != A) return GRB_NOT_IMPLEMENTED;
  return matrix_scale(A, op, nullptr, val, true);
}

// eWiseMult, matrix (x) broadcast vector (operations.hpp:240-267): C(i,j) = A(i,j) (x) B(i), or
// B(j) with GrB_INP1 = GrB_TRAN; in place (C == A)
grb_info grb_matrix_eWiseMult_vector(grb_matrix C, grb_semiring op, grb_matrix A, grb_vector B, grb_descriptor desc) { GRB_API_ENTER();
  if (!C || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (C != A) return GRB_NOT_IMPLEMENTED;
  if (desc->desc[GRB_INP0] != GRB_DEFAULT) return GRB_INVALID_VALUE;
  return matrix_scale(A, op, B, 0.0, desc->desc[GRB_INP1] != GRB_TRAN);
}

grb_info grb_reduce_matrix_scalar(double* val, grb_accum accum, grb_monoid op, grb_matrix A, grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!val || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (desc->struconly) { *val = (double)A->nvals; return GRB_SUCCESS; }    // reduce.hpp:86-87
  return k_reduce(op, A->dtype, A->csr.val, A->nvals, val);
}

grb_info grb_matrix_tril(grb_matrix C, grb_matrix A, grb_descriptor desc) { GRB_API_ENTER();
  if (!C || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (C->nrows != A->nrows || C->ncols != A->ncols) return GRB_DIMENSION_MISMATCH;
  const grb_index *ptr, *ind;
  const void* val;
  GRB_TRY(grb_matrix_host_csr(A, &ptr, &ind, &val));
  const uint32_t* v = (const uint32_t*)val;
  std::vector<Index> nptr((size_t)A->nrows + 1, 0), nind;
  std::vector<uint32_t> nval;
  for (Index r = 0; r < A->nrows; ++r) {
    for (Index p = ptr[r]; p < ptr[r + 1]; ++p)
      if (ind[p] <= r) { nind.push_back(ind[p]); nval.push_back(v[p]); }    // keep row >= col (tri.hpp:33-40)
    nptr[(size_t)r + 1] = (Index)nind.size();
  }
  static const Index kZero = 0;
  static const uint32_t kZeroV = 0;
  return grb_matrix_build_csr(C, nptr.data(), nind.empty() ? &kZero : nind.data(), nval.empty() ? &kZeroV : nval.data(),
                              (Index)nind.size(), nullptr, nullptr, nullptr);
}

// traceMxmTranspose (extension, operations.hpp:698-711 -> backend :1076-1108, trace.hpp:10-52)
grb_info grb_trace_mxm_transpose(double* val, grb_semiring op, grb_matrix A, grb_matrix B, grb_descriptor desc) { GRB_API_ENTER();
  if (!val || !A || !B) return GRB_UNINITIALIZED_OBJECT;
  (void)desc;
  if (!A->built || !B->built) return GRB_UNINITIALIZED_OBJECT;
  if (A->dtype != B->dtype) return GRB_DOMAIN_MISMATCH;
  if (A->nrows != B->nrows) return GRB_DIMENSION_MISMATCH;
  Context& c = ctx();
  void* d_out;
  GRB_TRY(scratch(10, 256, &d_out));
  GRB_HIP_TRY(hipMemsetAsync(d_out, 0, 8, c.stream));
  const Index n = A->nrows;
  if (n > 0) {
    const int grid = stream_grid((long long)n * kWave, kBlock);
    GRB_TRY(dispatch_semiring(op, A->dtype, [&](auto tag, auto t) -> grb_info {
      using T = decltype(t);
      constexpr int SR = decltype(tag)::value;
      hipLaunchKernelGGL((trace_kernel<SR, T>), dim3(grid), dim3(kBlock), 0, c.stream, (T*)d_out, n, A->csr.ptr,
                         A->csr.ind, (const T*)A->csr.val, B->csr.ptr, B->csr.ind, (const T*)B->csr.val);
      GRB_HIP_TRY(hipGetLastError());
      return GRB_SUCCESS;
    }));
  }
  int bits = 0;
  GRB_TRY(fetch_ints((const int*)d_out, 1, &bits));
  if (A->dtype == GRB_F32) { float f; memcpy(&f, &bits, 4); *val = (double)f; }
  else *val = (double)bits;
  return GRB_SUCCESS;
}

// algorithm::tc (algorithm/tc.hpp:15-54): B = (A x A^T) .* A on the lower triangle, ntris = sum(B)
grb_info grb_tc(int64_t* ntris, grb_matrix A, grb_matrix B, grb_descriptor desc, grb_algo_result* result) { GRB_API_ENTER();
  if (!ntris || !A || !B || !desc) return GRB_UNINITIALIZED_OBJECT;
  float ms = 0.f;
  grb_descriptor_toggle(desc, GRB_INP1);
  grb_info info = grb_timer_start();
  double sum = 0;
  long long wide = 0;
  // the count without the product (tc_count.hip), when grb_mxm would have accepted the call and A is what tc.hpp says it is
  bool counted = false;
  if (info == GRB_SUCCESS && A->built && A != B && A->dtype == B->dtype && desc->desc[GRB_INP0] != GRB_TRAN &&
      desc->desc[GRB_INP1] == GRB_TRAN && A->csr.ptr && A->csc.ptr && B->nrows == A->nrows && B->ncols == A->ncols)
    info = tc_count_try(A, &wide, &counted);
  if (counted) {
    sum = (double)wide;
  } else if (info == GRB_SUCCESS) {
    info = grb_mxm(B, A, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, A, A, desc);
  }
  if (counted) {
  } else if (info == GRB_SUCCESS && B->dtype == GRB_I32 && !desc->struconly) {
    // reduce<int, int>(ntris, ...) in the reference (tc.hpp:41-42) wraps beyond 2^31 triangles; this entry
    // point returns the count in 64 bits (the frontend's int* overload keeps the reference's int)
    void* p_sum;
    info = scratch(10, 8, &p_sum);
    if (info == GRB_SUCCESS && hipMemsetAsync(p_sum, 0, 8, ctx().stream) != hipSuccess) info = GRB_PANIC;
    if (info == GRB_SUCCESS && B->nvals > 0) {
      hipLaunchKernelGGL(sum_i32_wide_kernel, dim3(stream_grid(B->nvals, kBlock * 8)), dim3(kBlock), 0, ctx().stream,
                         (const int*)B->csr.val, B->nvals, (long long*)p_sum);
      if (hipGetLastError() != hipSuccess) info = GRB_PANIC;
    }
    if (info == GRB_SUCCESS && (hipMemcpyAsync(&wide, p_sum, 8, hipMemcpyDeviceToHost, ctx().stream) != hipSuccess ||
                                hipStreamSynchronize(ctx().stream) != hipSuccess))
      info = GRB_PANIC;
    sum = (double)wide;
  } else if (info == GRB_SUCCESS) {
    info = grb_reduce_matrix_scalar(&sum, GRB_ACCUM_NULL, GRB_PLUS_MONOID, B, desc);
    wide = (long long)sum;
  }
  if (info == GRB_SUCCESS) info = grb_timer_stop(&ms);
  // (the reference leaves GrB_INP1 toggled: tc.hpp:23 has no matching toggle back)
  *ntris = (int64_t)wide;
  if (result) { result->iterations = 1; result->tight_ms = ms; result->last_value = sum; }
  return info;
}

}  // extern "C"
