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

constexpr int kCbThreads = 1024;
constexpr int kCbWaves = kCbThreads / kWave;
constexpr int kCbRows = 16384;            // rows per band: 16 Ki accumulators of <= 8 bytes = 128 KiB of LDS
constexpr int kCbItemGroups = 4096;       // groups per work item of the hub band
constexpr int kCbLightGroups = 16384;     // a light band is cut when it reaches this many groups
constexpr unsigned int kCbPad = 0x8000u;
constexpr int kCbUnroll = 8;              // groups a wave has in flight

struct CbBand {
  int row0, nrows;                        // light: rows [row0, row0 + nrows); hub: the hub list [0, nrows)
  int hub;
};
struct CbItem {
  int band;
  int g0, g1;                             // groups
};

struct CbArgs {
  const CbBand* bands;
  const CbItem* items;                    // grouped by workgroup; a workgroup's hub items are consecutive
  const int* wg_ptr;                      // [grid + 1]
  const int* wg_slot;                     // [grid] partial slot of the workgroup's hub items, -1: none
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
  int grid = 0, nbands = 0, nitems = 0, nslots = 0;
  long long ngroups = 0, entries = 0;
  bool iso = false;
  void* d_partials = nullptr;             // [nslots][nhub]
  std::vector<void*> owned;
};

inline void free_spmv_cband(SpmvCBand* b) {
  if (!b) return;
  for (void* p : b->owned)
    if (p) (void)hipFree(p);
  delete b;
}

// ---- preparation kernels ---------------------------------------------------------------------------------------
// key of entry p: (band of its row) << 32 | rank of its column; payload p; and the row's position inside its band
__global__ __launch_bounds__(kBlock) void cband_keys_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                            Index n, const unsigned int* __restrict__ row_band,
                                                            const unsigned short* __restrict__ row_loc,
                                                            const Index* __restrict__ rank /* nullable: natural order */,
                                                            unsigned long long* __restrict__ keys,
                                                            unsigned int* __restrict__ pay, unsigned short* __restrict__ eloc) {
  const int lane = lane_id();
  const Index nwaves = (Index)gridDim.x * kWavesPerBlock;
  for (Index r = (Index)blockIdx.x * kWavesPerBlock + wave_id(); r < n; r += nwaves) {
    const unsigned long long hi = (unsigned long long)row_band[r] << 32;
    const unsigned short loc = row_loc[r];
    const Index e = ptr[r + 1];
    for (Index p = ptr[r] + lane; p < e; p += kWave) {
      const Index c = ind[p];
      keys[p] = hi | (unsigned long long)(unsigned int)(rank ? rank[c] : c);
      pay[p] = (unsigned int)p;
      eloc[p] = loc;
    }
  }
}

// first sorted position of every (band, 65536-column block): out[b * (ncb + 1) + cb]
__global__ void cband_segments_kernel(const unsigned long long* __restrict__ keys, const long long* __restrict__ band_start,
                                      int nbands, int ncb, long long* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nbands * (ncb + 1)) return;
  const int b = (int)(t / (ncb + 1)), cb = (int)(t % (ncb + 1));
  long long lo = band_start[b], hi = band_start[b + 1];
  const unsigned long long want = ((unsigned long long)(unsigned)b << 32) | ((unsigned long long)cb << 16);
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  out[t] = lo;
}

// one wave per group: the coded entries, their values, the group's base
__global__ __launch_bounds__(kBlock) void cband_emit_kernel(const unsigned long long* __restrict__ keys,
                                                            const unsigned int* __restrict__ pay,
                                                            const unsigned short* __restrict__ eloc,
                                                            const unsigned int* __restrict__ val /* nullable */,
                                                            const long long* __restrict__ seg_entry /* [nseg + 1] */,
                                                            const long long* __restrict__ seg_group /* [nseg + 1] */,
                                                            const unsigned int* __restrict__ seg_base, int nseg,
                                                            long long ngroups, unsigned int* __restrict__ pack,
                                                            unsigned int* __restrict__ val2, unsigned int* __restrict__ gbase) {
  const int lane = lane_id();
  const long long nwaves = (long long)gridDim.x * kWavesPerBlock;
  for (long long g = (long long)blockIdx.x * kWavesPerBlock + wave_id(); g < ngroups; g += nwaves) {
    int lo = 0, hi = nseg - 1;                         // last segment whose first group is <= g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg_group[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const long long q = seg_entry[lo] + (g - seg_group[lo]) * kWave + lane;
    const bool valid = q < seg_entry[lo + 1];
    unsigned int pk = kCbPad, v = 0;
    if (valid) {
      const unsigned int p = pay[q];
      pk = (((unsigned int)(keys[q] & 0xffffffffull) - seg_base[lo]) << 16) | (unsigned int)eloc[p];
      if (val) v = val[p];
    }
    pack[g * kWave + lane] = pk;
    if (val2) val2[g * kWave + lane] = v;
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
    if (v < *addr) atomicMin(addr, v);
  } else if constexpr (op == OP_MAX) {
    if (v > *addr) atomicMax(addr, v);
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

template <int SR, typename T, bool kIso>
__global__ __launch_bounds__(kCbThreads) void spmv_cband_kernel(CbArgs a, const T* __restrict__ u,
                                                                const void* __restrict__ mask, int mask_f32, int scmp,
                                                                int accum, T* w, void* __restrict__ partials_raw) {
  typedef Semiring<SR, T> S;
  typedef typename CbAcc<S::monoid, T>::type Acc;
  __shared__ unsigned long long ys_raw[kCbRows];
  Acc* ys = reinterpret_cast<Acc*>(ys_raw);
  Acc* __restrict__ partials = reinterpret_cast<Acc*>(partials_raw);
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = a.wg_ptr[blockIdx.x], i1 = a.wg_ptr[blockIdx.x + 1];
  const T* __restrict__ val = reinterpret_cast<const T*>(a.val);
  T iso;
  memcpy(&iso, &a.iso_bits, 4);
  int cur = -1;
  CbBand B = {0, 0, 0};
  auto flush = [&]() {
    __syncthreads();
    if (B.hub) {
      Acc* out = partials + (size_t)a.wg_slot[blockIdx.x] * (size_t)a.nhub;
      for (int i = tid; i < B.nrows; i += kCbThreads) out[i] = ys[i];
    } else {
      for (int i = tid; i < B.nrows; i += kCbThreads) {
        const Index row = B.row0 + i;
        if (!((a.hub_bits[row >> 5] >> (row & 31)) & 1u)) spmv_store<SR, T>(w, row, (T)ys[i], mask, mask_f32, scmp, accum);
      }
    }
    __syncthreads();
  };
  for (int it = i0; it < i1; ++it) {
    const CbItem I = a.items[it];
    if (I.band != cur) {
      if (cur >= 0) flush();
      B = a.bands[I.band];
      cur = I.band;
      for (int i = tid; i < B.nrows; i += kCbThreads) ys[i] = (Acc)S::identity();
      __syncthreads();
    }
    for (int g = I.g0 + wave; g < I.g1; g += kCbWaves * kCbUnroll) {
      unsigned int pk[kCbUnroll], base[kCbUnroll];
      T av[kCbUnroll], x[kCbUnroll];
#pragma unroll
      for (int k = 0; k < kCbUnroll; ++k) {
        const int gg = g + k * kCbWaves;
        const int gc = gg < I.g1 ? gg : I.g1 - 1;
        pk[k] = stream_load(&a.pack[(size_t)gc * kWave + lane]);
        if (!kIso) av[k] = stream_load(&val[(size_t)gc * kWave + lane]);
        base[k] = a.gbase[gc];
        if (gg >= I.g1) pk[k] = kCbPad;
      }
#pragma unroll
      for (int k = 0; k < kCbUnroll; ++k) x[k] = u[base[k] + (pk[k] >> 16)];
#pragma unroll
      for (int k = 0; k < kCbUnroll; ++k)
        if (!(pk[k] & kCbPad)) cband_combine<S::monoid, T, Acc>(&ys[pk[k] & 0x7fffu], S::mul(kIso ? iso : av[k], x[k]));
    }
  }
  if (cur >= 0) flush();
}

// the hub band's rows: partial slices folded in slot order, then the epilogue
template <int SR, typename T>
__global__ __launch_bounds__(kBlock) void spmv_cband_hub_kernel(const void* __restrict__ partials_raw, int nslots, int nhub,
                                                                const Index* __restrict__ hub_rows,
                                                                const void* __restrict__ mask, int mask_f32, int scmp,
                                                                int accum, T* w) {
  typedef Semiring<SR, T> S;
  typedef typename CbAcc<S::monoid, T>::type Acc;
  const Acc* __restrict__ partials = reinterpret_cast<const Acc*>(partials_raw);
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nhub) return;
  Acc acc = (Acc)S::identity();
  for (int s = 0; s < nslots; ++s) {
    const Acc p = partials[(size_t)s * nhub + i];
    if constexpr (std::is_same<Acc, T>::value) acc = S::add(acc, p);
    else acc += p;
  }
  spmv_store<SR, T>(w, hub_rows[i], (T)acc, mask, mask_f32, scmp, accum);
}

}  // namespace grb
