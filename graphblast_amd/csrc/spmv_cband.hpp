// spmv_cband.hpp -- the SpMV's second matrix format: row bands whose entries are sorted by COLUMN and coded in
// 16 + 16 bits (included by spmv.hip after its tuning constants).
//
// What bounds the CSR kernel of spmv.hip on a power-law matrix is not the 8 bytes per entry it streams but the L2
// request rate of the gathers of u: in CSR order the 64 lanes of a wave gather from ~64 different lines (RMAT-22:
// 61 M cold gathers at ~226 G/s are 0.27 of its 0.59 ms).  Sorting the entries of a whole BAND of
// rows by column makes neighbouring lanes gather from the same or neighbouring lines -- the request count drops by
// the number of entries that share a line -- at the price of a scattered accumulation, which stays on chip:
//
//   band        <= 16 Ki rows whose sums live in LDS (128 KiB of 8-byte accumulators, see below).  LIGHT bands are
//               ranges of consecutive rows; the <= 16 Ki rows of largest degree form ONE hub band (RMAT-22: 43 % of
//               the entries), whose column-sorted list is so dense that a wave's 64 gathers fall into one or two lines.
//   entry       one 32-bit word: (column - group base) << 16 | pad << 15 | row within the band; values in a second
//               array -- or nowhere when every stored value is the same (an "iso" matrix: PageRank's pattern times a
//               row scale, BFS / CC patterns): 4 bytes per entry instead of CSR's 8.
//   group       64 consecutive entries of a band (one per lane) that share the upper 16 bits of their column, with
//               the base in a side array; a (band, 65536-column block) segment is padded to whole groups.
//   item        a band, or a run of <= 4096 groups of the hub band; dealt to the workgroups once, at preparation,
//               largest first.  A workgroup accumulates an item's products into its LDS slice with LDS atomics,
//               then writes the band's rows (light: coalesced, mask / accumulate epilogue applied) or its partial
//               slice of the hub band, which a second small kernel folds in slot order.
//
// Columns are the RANK-renamed ones of the hub packing (u packed by descending reference count), so a band's list
// starts with the hottest columns.  The accumulating atomic is chosen by what the LDS executes natively
// (tools/probes/lds_atomic_probe.hip, cycles per wave instruction per CU): ds_add_u32 / ds_min / ds_max 13;
// ds_add_f32 193 -- serialised lane by lane whatever the addresses -- against 25 for ds_add_f64 and 27 for a
// compare-and-swap loop.  So a float sum is accumulated in DOUBLE (products formed in float as the semiring says,
// added in double, rounded once when the row is written): faster than the float atomic by 8x, and the result no
// longer depends on the order of the additions in any bit a float keeps, bar ties at a rounding boundary.
// Integer sums, min, max, or, and are exact in any order.  Only true commutative monoids take this path; the
// comparison "monoids" of stddef.hpp keep the CSR kernel (the reference's own reduction order is unpinned as well,
// SURVEY.md 8(c)).
//
// Replaces mgpu::SpmvCsrBinary (backend/cuda/spmv.hpp:178-220) like the CSR kernel does.
#pragma once

namespace grb {

// (build flags for the A/B of band size against waves per CU: tools/spmv_cband_occupancy.sh, docs/experiments.md R6.6)
#ifndef GRB_CB_THREADS
#define GRB_CB_THREADS 1024
#endif
#ifndef GRB_CB_ROWS
#define GRB_CB_ROWS 16384
#endif
#ifndef GRB_CB_WG_PER_CU
#define GRB_CB_WG_PER_CU 1
#endif
constexpr int kCbThreads = GRB_CB_THREADS;
constexpr int kCbWaves = kCbThreads / kWave;
constexpr int kCbRows = GRB_CB_ROWS;      // rows per band: 16 Ki accumulators of <= 8 bytes = 128 KiB of LDS
constexpr int kCbWgPerCu = GRB_CB_WG_PER_CU;   // persistent workgroups per CU the groups are dealt to
#ifndef GRB_CB_WPE
#define GRB_CB_WPE 4                      // waves per SIMD the product kernel is built for (= kCbWgPerCu x kCbWaves / 4)
#endif
#ifndef GRB_CB_ITEM_GROUPS
#define GRB_CB_ITEM_GROUPS 4096
#endif
constexpr int kCbItemGroups = GRB_CB_ITEM_GROUPS;   // groups per work item of the hub band
constexpr int kCbLightGroups = 16384;     // a light band is cut when it reaches this many groups
constexpr unsigned int kCbPad = 0x8000u;
#ifndef GRB_CB_UNROLL
#define GRB_CB_UNROLL 8
#endif
constexpr int kCbUnroll = GRB_CB_UNROLL;  // groups of one pipeline stage of a wave

struct CbBand {
  int row0, nrows;                        // light: rows [row0, row0 + nrows); hub: the hub list [0, nrows)
  int hub;
};
struct CbItem {
  int band;
  int g0, g1;                             // groups
  int slot_off;                           // -1: the band is this workgroup's alone (rows written directly); else
};                                        // the element offset of its partial slice

struct CbArgs {
  const CbBand* bands;
  const CbItem* items;                    // grouped by workgroup; a workgroup's hub items are consecutive
  const int* wg_ptr;                      // [grid + 1]
  const int* fin_band;                    // bands several workgroups share ...
  const int* fin_ptr;                     // ... their partial slices [fin_ptr[f], fin_ptr[f + 1]) ...
  const int* fin_off;                     // ... as element offsets
  const unsigned int* pack;               // [ngroups * 64]
  const void* val;                        // [ngroups * 64], null for an iso matrix
  const unsigned int* gbase;              // [ngroups]
  const Index* hub_rows;                  // [nhub]
  const unsigned int* hub_bits;           // bitmap over the rows
  unsigned int iso_bits;
  int nhub;
};


struct SpmvCBand {
  CbArgs args;
  unsigned int nhot = 0;                  // column codes below this index the packed hot prefix (plan.d_u2)
  int grid = 0, nbands = 0, nitems = 0, nslots = 0, nfin = 0, max_fin_rows = 0;
  long long partial_elems = 0;
  long long ngroups = 0, entries = 0;
  bool iso = false;
  void* d_partials = nullptr;             // [nslots][nhub]
  std::vector<int> wg_groups, wg_hub_groups, wg_items;   // per workgroup, for GRB_SPMV_TRACE
  std::vector<void*> owned;
};

inline void free_spmv_cband(SpmvCBand* b) {
  if (!b) return;
  for (void* p : b->owned)
    if (p) (void)hipFree(p);
  delete b;
}

// ---- preparation kernels ---------------------------------------------------------------------------------------
// Rows -> hubs, bands, a place for every row: on the device (round 5; the host did this over a 17 MB copy of the row
// pointers -- 10 of the preparation's 25 ms on a good day, 100 on a box whose host cores were busy).
//   cband_degree_hist_kernel   degrees counted in 65 537 bins (>= 65 536 share the last): the hub threshold is the
//                              (kCbRows + 1)-th largest degree, read off the histogram by the host (256 KB)
//   cband_light_kernel         light[r] = the row's entries if it is no hub (else 0), flag[r] = hub; hub bitmap.
//                              Exclusive scans of both (build.hip) give every row the light entries in front of it and
//                              every hub its number
//   cband_next_kernel          next[r] = where the band that starts at row r ends: kCbRows rows on, or in front of the first
//                              row that takes the band's light entries past the limit (a search in the prefix sums)
//   cband_chase_kernel         one thread follows next[] from row 0: the bands' first rows (a few hundred dependent loads)
//   cband_place_kernel         band and place of every row; the hub list; the bands' first sorted positions
constexpr int kCbDegBins = 65536;
constexpr int kCbDegLds = 16384;          // bins counted in LDS (64 KiB); longer rows go to memory one atomic each
__global__ __launch_bounds__(1024) void cband_degree_hist_kernel(const Index* __restrict__ ptr, Index n,
                                                                 unsigned int* __restrict__ hist /* [kCbDegBins + 1], zeroed */) {
  __shared__ unsigned int h[kCbDegLds];
  for (int i = threadIdx.x; i < kCbDegLds; i += blockDim.x) h[i] = 0u;
  __syncthreads();
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index r = (Index)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    const Index d = ptr[r + 1] - ptr[r];
    if (d < kCbDegLds) atomicAdd(&h[d], 1u);
    else atomicAdd(&hist[d < kCbDegBins ? d : kCbDegBins], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCbDegLds; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(kBlock) void cband_light_kernel(const Index* __restrict__ ptr, Index n, Index hub_above,
                                                             unsigned int* __restrict__ light /* [n + 1] */,
                                                             unsigned int* __restrict__ flag /* [n + 1] */,
                                                             unsigned int* __restrict__ hub_bits /* [(n + 31) / 32] */) {
  // a wave takes 64 consecutive rows (two words of the bitmap); rows past n are no hubs
  const Index nchunks = (n + kWave) / kWave;              // covers index n too (the scans' closing zero)
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  const int lane = lane_id();
  for (Index c = (Index)blockIdx.x * kWavesPerBlock + wave_id(); c < nchunks; c += nwaves) {
    const Index r = c * kWave + lane;
    Index d = 0;
    if (r < n) d = ptr[r + 1] - ptr[r];
    const bool hub = r < n && d > hub_above;
    if (r <= n) { light[r] = hub ? 0u : (unsigned int)d; flag[r] = hub ? 1u : 0u; }
    const unsigned long long m = __ballot(hub);
    const Index w = 2 * c + (lane == 32 ? 1 : 0);
    if ((lane == 0 || lane == 32) && (long long)w * 32 < (long long)n)
      hub_bits[w] = lane == 0 ? (unsigned int)(m & 0xffffffffull) : (unsigned int)(m >> 32);
  }
}

__global__ __launch_bounds__(kBlock) void cband_next_kernel(const unsigned int* __restrict__ P /* [n + 1] light entries in front */,
                                                            Index n, unsigned int max_entries, int max_rows,
                                                            Index* __restrict__ next) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index r = (Index)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    // the host's rule, row by row: a band that began at r is cut in front of row q > r when it holds max_rows rows or
    // when row q's entries would take it past max_entries, i.e. P[q + 1] - P[r] > max_entries
    const unsigned long long lim = (unsigned long long)P[r] + max_entries;
    Index lo = r + 1, hi = r + max_rows < n ? r + max_rows : n;     // the answer lies in [lo, hi]
    while (lo < hi) {
      const Index mid = lo + ((hi - lo) >> 1);
      if ((unsigned long long)P[mid + 1] > lim) hi = mid; else lo = mid + 1;
    }
    next[r] = lo;
  }
}

__global__ void cband_chase_kernel(const Index* __restrict__ next, Index n, Index* __restrict__ starts, int cap,
                                   int* __restrict__ count) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int k = 0;
  for (Index r = 0; r < n; r = next[r]) {
    if (k < cap) starts[k] = r;
    ++k;
  }
  if (k < cap) starts[k] = n;
  *count = k;
}

__global__ __launch_bounds__(kBlock) void cband_place_kernel(const Index* __restrict__ starts, int nlight, unsigned int band0,
                                                             const unsigned int* __restrict__ P, const unsigned int* __restrict__ H,
                                                             const unsigned int* __restrict__ hub_bits, Index n,
                                                             unsigned long long hub_entries,
                                                             unsigned int* __restrict__ row_band, unsigned short* __restrict__ row_loc,
                                                             Index* __restrict__ hub_rows, long long* __restrict__ band_first) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index r = (Index)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    if ((hub_bits[r >> 5] >> (r & 31)) & 1u) {
      const unsigned int i = H[r];
      row_band[r] = 0u;
      row_loc[r] = (unsigned short)i;
      hub_rows[i] = r;
    } else {
      int lo = 0, hi = nlight - 1;                         // the last band that starts at or in front of r
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (starts[mid] <= r) lo = mid; else hi = mid - 1;
      }
      row_band[r] = band0 + (unsigned int)lo;
      row_loc[r] = (unsigned short)(r - starts[lo]);
    }
    if (r < (Index)nlight) band_first[r] = (long long)hub_entries + (long long)P[starts[r]];
  }
}

// The sort key of entry p:  ((band of its row) << colbits | code of its column) << 16 | the row's position inside its band.
// Only the bits above the low 16 are sorted on; the position rides along, and the payload is the entry's VALUE -- what
// the emit pass needs is then in the sorted arrays themselves, read front to back (the first version sorted entry
// indices and gathered position and value through them afterwards: 16 GB of line fetches for RMAT-22).
constexpr int kCbKeyLow = 16;
// A wave takes 64 consecutive rows: their entries are one contiguous piece of the CSR arrays, walked 64 at a time
// (coalesced, every lane busy; a wave per ROW -- the first version -- left most lanes idle on the short rows that make up
// a power-law matrix).  A lane finds its entry's row among the wave's 65 row pointers (LDS).  Hub rows are stepped over
// here and taken by cband_keys_hub_kernel, a 1024-thread workgroup per hub row: one wave walking a row of 300 000
// entries was the whole kernel's time (3.5 of 4.4 ms on RMAT-22).
// The column's code is its rank when that is below nhot, else nhot + the column: from the CSR kernel's renamed column
// ids when they exist (ind2: no gather at all), else through the rank array (codes: one 4-byte gather per entry).
__device__ inline void cband_key_store(Index p, const Index* __restrict__ ind, const unsigned int* __restrict__ val,
                                       const Index* __restrict__ codes, const Index* __restrict__ ind2, unsigned int nhot,
                                       int colbits, unsigned int band, unsigned int loc, unsigned long long* __restrict__ keys,
                                       unsigned int* __restrict__ pay) {
  const unsigned int col = (unsigned int)ind[p];
  unsigned int code = col;
  if (ind2) {
    const unsigned int rk = (unsigned int)ind2[p];
    code = rk < nhot ? rk : nhot + col;
  } else if (codes) {
    code = (unsigned int)codes[col];
  }
  keys[p] = ((((unsigned long long)band << colbits) | (unsigned long long)code) << kCbKeyLow) | (unsigned long long)loc;
  pay[p] = val ? val[p] : 0u;
}

__global__ __launch_bounds__(kBlock) void cband_keys_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                            const unsigned int* __restrict__ val /* nullable */,
                                                            Index n, const unsigned int* __restrict__ row_band,
                                                            const unsigned short* __restrict__ row_loc,
                                                            const unsigned int* __restrict__ hub_bits,
                                                            const Index* __restrict__ codes /* nullable: natural order, or ind2 */,
                                                            const Index* __restrict__ ind2 /* nullable */, unsigned int nhot,
                                                            int colbits, unsigned long long* __restrict__ keys,
                                                            unsigned int* __restrict__ pay) {
  __shared__ Index s_ptr[kWavesPerBlock][kWave + 1];
  __shared__ unsigned int s_band[kWavesPerBlock][kWave];
  __shared__ unsigned int s_loc[kWavesPerBlock][kWave];
  const int lane = lane_id(), wv = wave_id();
  const Index nchunks = (n + kWave - 1) / kWave;
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index c = (Index)blockIdx.x * kWavesPerBlock + wv; c < nchunks; c += nwaves) {
    const Index r0 = c * kWave;
    const bool have = r0 + lane < n;
    const Index r = have ? r0 + lane : n;
    __builtin_amdgcn_wave_barrier();
    s_ptr[wv][lane] = ptr[r];                              // rows past n start where the matrix ends: never chosen
    if (lane == 0) s_ptr[wv][kWave] = ptr[r0 + kWave < n ? r0 + kWave : n];
    s_band[wv][lane] = have ? row_band[r] : 0u;
    s_loc[wv][lane] = have ? (unsigned int)row_loc[r] : 0u;
    const unsigned long long hubs = __ballot(have && ((hub_bits[r >> 5] >> (r & 31)) & 1u));
    __builtin_amdgcn_wave_barrier();
    const Index p1 = s_ptr[wv][kWave];
    Index base = s_ptr[wv][0];
    while (base < p1) {
      const Index p = base + lane < p1 ? base + lane : p1 - 1;
      int j = 0;                                           // the last row of the chunk that starts at or in front of p
#pragma unroll
      for (int step = kWave / 2; step > 0; step >>= 1)
        if (s_ptr[wv][j + step] <= p) j += step;
      const int j0 = __builtin_amdgcn_readfirstlane(j);    // the row this step begins in
      if ((hubs >> j0) & 1ull) { base = s_ptr[wv][j0 + 1]; continue; }
      if (base + lane < p1 && !((hubs >> j) & 1ull))
        cband_key_store(p, ind, val, codes, ind2, nhot, colbits, s_band[wv][j], s_loc[wv][j], keys, pay);
      base += kWave;
    }
  }
}

// the hub rows' entries: hub i (row hub_rows[i]) is band 0, place i; a workgroup per hub, 1024 entries a step
__global__ __launch_bounds__(1024) void cband_keys_hub_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                              const unsigned int* __restrict__ val /* nullable */,
                                                              const Index* __restrict__ hub_rows, int nhub,
                                                              const Index* __restrict__ codes, const Index* __restrict__ ind2,
                                                              unsigned int nhot, int colbits, unsigned long long* __restrict__ keys,
                                                              unsigned int* __restrict__ pay) {
  for (int i = blockIdx.x; i < nhub; i += gridDim.x) {
    const Index r = hub_rows[i];
    const Index e = ptr[r + 1];
    for (Index p = ptr[r] + (Index)threadIdx.x; p < e; p += (Index)blockDim.x)
      cband_key_store(p, ind, val, codes, ind2, nhot, colbits, 0u, (unsigned int)i, keys, pay);
  }
}

// first sorted position of every (band, 65536-column block): out[b * (ncb + 1) + cb]
__global__ void cband_segments_kernel(const unsigned long long* __restrict__ keys, const long long* __restrict__ band_start,
                                      int nbands, int ncb, int colbits, long long* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nbands * (ncb + 1)) return;
  const int b = (int)(t / (ncb + 1)), cb = (int)(t % (ncb + 1));
  long long lo = band_start[b], hi = band_start[b + 1];
  const unsigned long long want = ((unsigned long long)(unsigned)b << colbits) + ((unsigned long long)cb << 16);
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((keys[mid] >> kCbKeyLow) < want) lo = mid + 1; else hi = mid;
  }
  out[t] = lo;
}

// one wave per group: the coded entries, their values, the group's base.  Inside a chunk of four groups (chunks are
// aligned to the band's first group; bands hold whole chunks) the entries are stored TRANSPOSED: lane L's 16-byte
// word holds entry L of each of the four groups, so a lane streams 16 bytes while gather instruction k of a wave
// still covers the 64 consecutive entries of group k (few lines) and the group's base is wave-uniform.
__global__ __launch_bounds__(kBlock) void cband_emit_kernel(const unsigned long long* __restrict__ keys,
                                                            const unsigned int* __restrict__ pay,
                                                            const long long* __restrict__ seg_entry /* [nseg + 1] */,
                                                            const long long* __restrict__ seg_group /* [nseg + 1] */,
                                                            const unsigned int* __restrict__ seg_base,
                                                            const long long* __restrict__ seg_band_g0 /* first group of the segment's band */,
                                                            int nseg, long long ngroups, int colbits, unsigned int* __restrict__ pack,
                                                            unsigned int* __restrict__ val2, unsigned int* __restrict__ gbase) {
  const int lane = lane_id();
  const long long nwaves = (long long)gridDim.x * kWavesPerBlock;
  const unsigned long long colmask = (1ull << colbits) - 1ull;
  for (long long g = (long long)blockIdx.x * kWavesPerBlock + wave_id(); g < ngroups; g += nwaves) {
    int lo = 0, hi = nseg - 1;                         // last segment whose first group is <= g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg_group[mid] <= g) lo = mid; else hi = mid - 1;
    }
    // groups past the segment's last one are the band's padding up to a whole chunk
    const long long q = seg_entry[lo] + (g - seg_group[lo]) * kWave + lane;
    const long long seg_groups = (seg_entry[lo + 1] - seg_entry[lo] + kWave - 1) / kWave;
    const bool valid = g - seg_group[lo] < seg_groups && q < seg_entry[lo + 1];
    unsigned int pk = kCbPad, v = 0;
    if (valid) {
      const unsigned long long key = keys[q];
      pk = (((unsigned int)((key >> kCbKeyLow) & colmask) - seg_base[lo]) << 16) | (unsigned int)(key & 0xffffull);
      if (val2) v = pay[q];
    }
    const long long rel = g - seg_band_g0[lo];
    const long long at = (seg_band_g0[lo] + (rel & ~3ll)) * kWave + (long long)lane * 4 + (rel & 3ll);
    pack[at] = pk;
    if (val2) val2[at] = v;
    if (lane == 0) gbase[g] = seg_base[lo];
  }
}

// are all stored values the same 4 bytes?  out[0] = min, out[1] = max of the raw words (out preset to {~0, 0})
__global__ __launch_bounds__(kBlock) void cband_iso_kernel(const unsigned int* __restrict__ val, long long n,
                                                           unsigned int* __restrict__ out) {
  unsigned int lo = 0xffffffffu, hi = 0u;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const unsigned int x = val[i];
    lo = x < lo ? x : lo;
    hi = x > hi ? x : hi;
  }
  lo = wave_reduce(lo, [](unsigned int a, unsigned int b) { return a < b ? a : b; });
  hi = wave_reduce(hi, [](unsigned int a, unsigned int b) { return a > b ? a : b; });
  if (lane_id() == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

__global__ void cband_invert_kernel(const Index* __restrict__ order, Index npacked, Index* __restrict__ rank) {
  const Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npacked) rank[order[i]] = i;
}

// ---- the product -------------------------------------------------------------------------------------------------
// what a row's sum is accumulated in
template <int M, typename T>
struct CbAcc {
  typedef typename std::conditional<std::is_same<T, float>::value && MonoidTraits<M>::op == OP_PLUS, double, T>::type type;
};

template <int M, typename T, typename A>
__device__ inline void cband_combine(A* addr, T v) {
  constexpr int op = MonoidTraits<M>::op;
  if constexpr (op == OP_PLUS) {
    atomicAdd(addr, (A)v);                             // ds_add_f64 / ds_add_u32
  } else if constexpr (op == OP_MIN) {
    // floats through the integer unit (ds_min_i32 / ds_max_u32: IEEE order is integer order for v >= 0 and the
    // reverse unsigned order below); a NaN product fails the test and is dropped, as fminf drops it
    if (v < *addr) {
      if constexpr (std::is_same<T, float>::value) {
        if (__float_as_int(v) >= 0) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));   // by sign BIT: -0.0 goes below
        else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
      } else {
        atomicMin(addr, v);
      }
    }
  } else if constexpr (op == OP_MAX) {
    if (v > *addr) {
      if constexpr (std::is_same<T, float>::value) {
        if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
        else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
      } else {
        atomicMax(addr, v);
      }
    }
  } else if constexpr (op == OP_LOR) {
    if (v != (T)0) *addr = (T)1;                       // idempotent store: benign race
  } else if constexpr (op == OP_LAND) {
    if (v == (T)0) *addr = (T)0;
  } else {
    unsigned int* a = reinterpret_cast<unsigned int*>(addr);
    unsigned int old = *a, assumed;
    do {
      assumed = old;
      T cur;
      memcpy(&cur, &assumed, 4);
      const T nv = Monoid<M, T>::add(cur, v);
      unsigned int nb;
      memcpy(&nb, &nv, 4);
      old = atomicCAS(a, assumed, nb);
    } while (old != assumed);
  }
}

// One lane takes one entry of each of FOUR groups (one 16-byte load of the coded words, one of the values; the
// chunk is stored transposed, see cband_emit_kernel): 4-byte loads reach 4.4 TB/s on this stream, and the guide
// prices 8-byte accesses at 0.54-0.70 of the 16-byte rate.
constexpr int kCbChunk = 4;               // groups per chunk
#ifndef GRB_CB_STAGE
#define GRB_CB_STAGE 2
#endif
constexpr int kCbStage = GRB_CB_STAGE;    // chunks per pipeline stage of a wave

typedef unsigned int CbWord4 __attribute__((ext_vector_type(4)));

template <int SR, typename T, bool kIso>
__global__ __launch_bounds__(kCbThreads) __attribute__((amdgpu_waves_per_eu(GRB_CB_WPE, GRB_CB_WPE))) void spmv_cband_kernel(CbArgs a, const T* __restrict__ u_hot,
                                                                const T* __restrict__ u_nat, unsigned int nhot,
                                                                const void* __restrict__ mask, int mask_f32, int scmp,
                                                                int accum, T* w, void* __restrict__ partials_raw,
                                                                unsigned long long* __restrict__ trace) {
  typedef Semiring<SR, T> S;
  typedef typename CbAcc<S::monoid, T>::type Acc;
  __shared__ unsigned long long ys_raw[kCbRows];
  Acc* ys = reinterpret_cast<Acc*>(ys_raw);
  Acc* __restrict__ partials = reinterpret_cast<Acc*>(partials_raw);
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = a.wg_ptr[blockIdx.x], i1 = a.wg_ptr[blockIdx.x + 1];
  const CbWord4* __restrict__ pack4 = reinterpret_cast<const CbWord4*>(a.pack);
  const CbWord4* __restrict__ val4 = reinterpret_cast<const CbWord4*>(a.val);
  const CbWord4* __restrict__ gbase4 = reinterpret_cast<const CbWord4*>(a.gbase);
  if (trace && tid == 0) trace[2 * blockIdx.x] = wall_clock64();
  T iso;
  memcpy(&iso, &a.iso_bits, 4);
  CbBand B = {0, 0, 0};
  auto flush = [&](int slot_off) {
    __syncthreads();
    if (slot_off >= 0) {
      Acc* out = partials + slot_off;
      for (int i = tid; i < B.nrows; i += kCbThreads) out[i] = ys[i];
    } else {
      for (int i = tid; i < B.nrows; i += kCbThreads) {
        const Index row = B.hub ? a.hub_rows[i] : B.row0 + i;
        if (B.hub || !((a.hub_bits[row >> 5] >> (row & 31)) & 1u)) spmv_store<SR, T>(w, row, (T)ys[i], mask, mask_f32, scmp, accum);
      }
    }
    __syncthreads();
  };
  // a column code below nhot is a position in the packed hot prefix; above, nhot + the column itself
  auto gather = [&](unsigned int code) -> T {
#if defined(GRB_CB_EXP_NOGATHER)   // isolating experiment (tools/spmv_cband_variants.sh): every gather hits one hot KiB
    return u_hot[code & 255u];
#else
    const T* p = code < nhot ? u_hot + code : u_nat + (code - nhot);
    return *p;
#endif
  };
  for (int it = i0; it < i1; ++it) {
    const CbItem I = a.items[it];                        // one run of one band: clear, accumulate, write
    B = a.bands[I.band];
    for (int i = tid; i < B.nrows; i += kCbThreads) ys[i] = (Acc)S::identity();
    __syncthreads();
    // Three stages in flight per wave, two chunks each: the coded entries (and values) of step i + 2 are streaming
    // in while the gathers of step i + 1 are outstanding and step i is accumulated -- the only waits on memory are
    // for loads issued two steps earlier.
    const int nch = (I.g1 - I.g0 + kCbChunk - 1) / kCbChunk;
    constexpr int kStep = kCbWaves * kCbStage;
    CbWord4 pk0[kCbStage], pk1[kCbStage], av0[kCbStage], av1[kCbStage];
    CbWord4 base0[kCbStage], base1[kCbStage];       // the four groups' bases: wave-uniform
    T x0[kCbStage][4];
#define GRB_CB_STREAM(C, PK, AV, BASE)                                                          \
  _Pragma("unroll") for (int k = 0; k < kCbStage; ++k) {                                         \
    const int cc = (C) + k * kCbWaves;                                                           \
    const int cl = cc < nch ? cc : nch - 1;                                                      \
    const size_t at = ((size_t)(I.g0 + kCbChunk * cl) * kWave) / 4 + lane;                       \
    (PK)[k] = __builtin_nontemporal_load(&pack4[at]);                                            \
    if (!kIso) (AV)[k] = __builtin_nontemporal_load(&val4[at]);                                  \
    (BASE)[k] = gbase4[(I.g0 + kCbChunk * cl) / 4];                                              \
    if (cc >= nch) (PK)[k] = CbWord4{kCbPad, kCbPad, kCbPad, kCbPad};                            \
  }
#define GRB_CB_GATHER(PK, BASE, X)                                   \
  _Pragma("unroll") for (int k = 0; k < kCbStage; ++k) {              \
    (X)[k][0] = gather((BASE)[k].x + ((PK)[k].x >> 16));             \
    (X)[k][1] = gather((BASE)[k].y + ((PK)[k].y >> 16));             \
    (X)[k][2] = gather((BASE)[k].z + ((PK)[k].z >> 16));             \
    (X)[k][3] = gather((BASE)[k].w + ((PK)[k].w >> 16));             \
  }
    auto accumulate = [&](unsigned int pk, unsigned int vbits, T x) {
      if (pk & kCbPad) return;
      T av;
      memcpy(&av, &vbits, 4);
      const T prod = S::mul(kIso ? iso : av, x);
#if defined(GRB_CB_EXP_NOATOMIC)   // isolating experiment: the products are consumed without touching the LDS slice
      if (prod == (T)12345) ys[pk & 0x7fffu] = (Acc)1;
#else
      cband_combine<S::monoid, T, Acc>(&ys[pk & 0x7fffu], prod);
#endif
    };
    int c = wave;
    if (c < nch) {
      GRB_CB_STREAM(c, pk0, av0, base0)
      asm volatile("" ::: "memory");
      GRB_CB_STREAM(c + kStep, pk1, av1, base1)
      asm volatile("" ::: "memory");
      GRB_CB_GATHER(pk0, base0, x0)
      for (; c < nch; c += kStep) {
        CbWord4 pk2[kCbStage], av2[kCbStage], base2[kCbStage];
        T x1[kCbStage][4];
        asm volatile("" ::: "memory");
        GRB_CB_STREAM(c + 2 * kStep, pk2, av2, base2)
        asm volatile("" ::: "memory");
        GRB_CB_GATHER(pk1, base1, x1)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < kCbStage; ++k) {
          accumulate(pk0[k].x, av0[k].x, x0[k][0]);
          accumulate(pk0[k].y, av0[k].y, x0[k][1]);
          accumulate(pk0[k].z, av0[k].z, x0[k][2]);
          accumulate(pk0[k].w, av0[k].w, x0[k][3]);
        }
#pragma unroll
        for (int k = 0; k < kCbStage; ++k) {
          pk0[k] = pk1[k]; base0[k] = base1[k];
          pk1[k] = pk2[k]; base1[k] = base2[k];
          if (!kIso) { av0[k] = av1[k]; av1[k] = av2[k]; }
#pragma unroll
          for (int j = 0; j < 4; ++j) x0[k][j] = x1[k][j];
        }
      }
    }
#undef GRB_CB_STREAM
#undef GRB_CB_GATHER
    flush(I.slot_off);
  }
  if (trace && tid == 0) trace[2 * blockIdx.x + 1] = wall_clock64();
}

// the rows of the bands several workgroups worked on: partial slices folded by sixteen lanes per row (wave q of a
// 1024-thread workgroup takes slots s0 + q, s0 + q + 16, ..., four loads in flight each: the hub band of RMAT-22 has
// ~210 slices, and with four waves per row block the kernel ran at 2.5 TB/s on 27 MB), then the epilogue.
// blockIdx.y = shared band, blockIdx.x = 64 of its rows.
constexpr int kCbFoldWaves = 16;
template <int SR, typename T>
__global__ __launch_bounds__(kCbFoldWaves * kWave) void spmv_cband_fold_kernel(CbArgs a, const void* __restrict__ partials_raw,
                                                                             const void* __restrict__ mask, int mask_f32, int scmp,
                                                                             int accum, T* w) {
  typedef Semiring<SR, T> S;
  typedef typename CbAcc<S::monoid, T>::type Acc;
  const Acc* __restrict__ partials = reinterpret_cast<const Acc*>(partials_raw);
  const int f = blockIdx.y;
  const CbBand B = a.bands[a.fin_band[f]];
  if ((int)blockIdx.x * kWave >= B.nrows) return;
  const int s0 = a.fin_ptr[f], s1 = a.fin_ptr[f + 1];
  const int l = threadIdx.x & (kWave - 1), q = threadIdx.x >> 6;     // wave q takes slots s0 + q, s0 + q + 16, ...
  const int i = blockIdx.x * kWave + l;
  __shared__ unsigned long long s_part[kCbFoldWaves][kWave];
  Acc acc = (Acc)S::identity();
  if (i < B.nrows) {
    auto fold = [&](Acc p) {
      if constexpr (std::is_same<Acc, T>::value) acc = S::add(acc, p);
      else acc += p;
    };
    int s = s0 + q;
    for (; s + 3 * kCbFoldWaves < s1; s += 4 * kCbFoldWaves) {       // four slices in flight
      const Acc p0 = partials[(size_t)a.fin_off[s] + i];
      const Acc p1 = partials[(size_t)a.fin_off[s + kCbFoldWaves] + i];
      const Acc p2 = partials[(size_t)a.fin_off[s + 2 * kCbFoldWaves] + i];
      const Acc p3 = partials[(size_t)a.fin_off[s + 3 * kCbFoldWaves] + i];
      fold(p0); fold(p1); fold(p2); fold(p3);
    }
    for (; s < s1; s += kCbFoldWaves) fold(partials[(size_t)a.fin_off[s] + i]);
  }
  *reinterpret_cast<Acc*>(&s_part[q][l]) = acc;
  __syncthreads();
  if (q == 0 && i < B.nrows) {
    Acc tot = acc;
    for (int k = 1; k < kCbFoldWaves; ++k) {
      const Acc p = *reinterpret_cast<Acc*>(&s_part[k][l]);
      if constexpr (std::is_same<Acc, T>::value) tot = S::add(tot, p);
      else tot += p;
    }
    const Index row = B.hub ? a.hub_rows[i] : B.row0 + i;
    if (B.hub || !((a.hub_bits[row >> 5] >> (row & 31)) & 1u)) spmv_store<SR, T>(w, row, (T)tot, mask, mask_f32, scmp, accum);
  }
}

// the hot prefix of the packed vector: u_hot[i] = u[order[i]] for the nhot most referenced columns
// (four elements per lane: one 16-byte read of the order, four gathers in flight, one 16-byte store)
template <typename T>
__global__ void cband_pack_kernel(const T* __restrict__ u, const Index* __restrict__ order, Index nhot, T* __restrict__ u_hot) {
  const Index i4 = ((Index)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < nhot) {
    const int4 o = *reinterpret_cast<const int4*>(order + i4);
    struct alignas(16) V4 { T x, y, z, w; };
    V4 v;
    v.x = u[o.x]; v.y = u[o.y]; v.z = u[o.z]; v.w = u[o.w];
    *reinterpret_cast<V4*>(u_hot + i4) = v;
  } else {
    for (Index i = i4; i < nhot; ++i) u_hot[i] = u[order[i]];
  }
}

// column -> code: its rank when that is below nhot, else nhot + the column itself
__global__ void cband_codes_kernel(Index* __restrict__ rank_to_code, Index m, unsigned int nhot) {
  const Index c = (Index)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < m) {
    const unsigned int r = (unsigned int)rank_to_code[c];
    rank_to_code[c] = (Index)(r < nhot ? r : nhot + (unsigned int)c);
  }
}

}  // namespace grb
