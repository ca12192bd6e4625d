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
// with list(p) in an LDS hash table of the workgroup that owns the pivot p and the lists of p's partners streamed past it
// a wave to a list, 16 bytes per lane per step.
#include "common.hpp"

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

// Every list starts at a multiple of four entries of D and is filled up to one with kTcPad, a number no list holds: a
// lane's 16 bytes are four entries of ONE list or four to skip, and no entry needs a range check of its own (the kernels
// are bound by their vector instructions as much as by the stream: 4 cycles of a SIMD per instruction and wave, ~20
// instructions per element before this).
//
// A wave streams the lists of up to 64 partners (lane l holds partner l's {first element, length}): 64 lanes x 16 bytes a
// step, kDepth steps' loads in flight -- the same partner's next 256 elements or the next partner's first ones, whichever
// follows -- before the oldest step's elements are looked up: a wave's chain is its look-ups, not its loads.
// look(x): one entry of a partner's list (or kTcPad).
template <int kDepth, typename F>
__device__ __forceinline__ void tc_stream_batch(const int* __restrict__ D, const int2 my, const int nb, const int lane, F&& look) {
  // the batch's steps, counted once: the loop below is a counted loop whose control is the scalar unit's alone
  const int total = (int)wave_sum_u32((unsigned)(my.y + 4 * kWave - 1) / (4u * kWave));
  int gi = -1, gk = 0, ge1 = 0;                 // the generator: partner gi, the step after the last issued starts at gk (of [.., ge1))
  int4 v[kDepth];
  int se1[kDepth], sk[kDepth];
  auto issue = [&](int d) {                     // (only called while a step is left: partners have at least one element)
    if (gk >= ge1) {
      ++gi;
      gk = __builtin_amdgcn_readlane(my.x, gi);
      ge1 = gk + __builtin_amdgcn_readlane(my.y, gi);
    }
    se1[d] = ge1; sk[d] = gk;
    if (gk + 4 * lane < ge1) v[d] = *reinterpret_cast<const int4*>(D + gk + 4 * lane);
    gk += 4 * kWave;
  };
#pragma unroll
  for (int d = 0; d < kDepth; ++d)
    if (d < total) issue(d);
  for (int st = 0; st < total; st += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      if (st + d >= total) break;
      if (sk[d] + 4 * lane < se1[d]) {
        look((unsigned)v[d].x); look((unsigned)v[d].y); look((unsigned)v[d].z); look((unsigned)v[d].w);
      }
      if (st + d + kDepth < total) issue(d);
    }
  }
}

// One task = a pivot and up to a few hundred of its partners: {pivot, first partner, partners, -}.  A pivot with many
// partners is cut into tasks that each build the table again (<= 2 len LDS operations against thousands of streamed
// elements).  A WAVE streams a partner: 64 lanes x 16 bytes a step, the load of the following step -- the same partner's
// or the next one's, the 64 partner descriptors of a batch sit in the lanes' registers -- issued before this step's
// elements are looked up, so a wave's chain is its look-ups, not its loads.
template <int kThreads, int kTable>
__global__ __launch_bounds__(kThreads) void tc_count_pivot_kernel(const int* __restrict__ D, const int* __restrict__ Dptr,
                                                                   const int2* __restrict__ P, const int4* __restrict__ tasks,
                                                                   unsigned long long* total) {
  __shared__ unsigned table[kTable];
  __shared__ int next;
  __shared__ unsigned part[kThreads / kWave];
  constexpr int kWaves = kThreads / kWave;
  const int tid = threadIdx.x, lane = lane_id();
  const int4 task = tasks[blockIdx.x];
  const int s = Dptr[task.x], len = Dptr[task.x + 1] - s;
  int need = len * GRB_TC_SLOTS - 1 > 63 ? len * GRB_TC_SLOTS - 1 : 63;
  if (need > kTable - 1) need = kTable - 1;
  const int lg = 32 - __clz(need);             // a table of 2^lg >= 4 len slots (2 len for the longest lists), at least 64
  const unsigned mask = (1u << lg) - 1u;
  const int shift = 32 - lg;
  for (unsigned i = tid; i <= mask; i += kThreads) table[i] = kTcEmpty;
  if (tid == 0) next = kWaves;
  __syncthreads();
  for (int i = tid; i < len; i += kThreads) {
    const unsigned x = (unsigned)D[s + i];
    if (x == kTcPad) continue;                  // (len counts the list's room: up to three entries of filling)
    unsigned slot = (x * 0x9E3779B1u) >> shift;
    while (atomicCAS(&table[slot], kTcEmpty, x) != kTcEmpty) slot = (slot + 1) & mask;
  }
  __syncthreads();
  unsigned count = 0;
  const int nbatch = (task.z + kWave - 1) / kWave;
  int b = kWaves > 1 ? wave_id() : 0;
  while (b < nbatch) {
    const int nb = task.z - b * kWave < kWave ? task.z - b * kWave : kWave;
    const int2 my = lane < nb ? P[task.y + b * kWave + lane] : make_int2(0, 0);
    tc_stream_batch<GRB_TC_DEPTH>(D, my, nb, lane, [&](unsigned x) {
      unsigned slot = (x * 0x9E3779B1u) >> shift;
      unsigned t = table[slot];
      while (t != x && t != kTcEmpty) { slot = (slot + 1) & mask; t = table[slot]; }
      count += t == x;
    });
    if (kWaves > 1) {
      if (lane == 0) b = atomicAdd(&next, 1);
      b = __builtin_amdgcn_readfirstlane(b);
    } else {
      ++b;
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

// The pivots numbered up to kBits (vertices are NUMBERED BY RANK, 0 = the highest degree, so list(p) holds numbers below
// p): the pivot's list is a BITMAP of p bits -- one LDS read and a bit test per streamed element, no hashing, no walk.
template <int kThreads, int kBits>
__global__ __launch_bounds__(kThreads) void tc_count_bitmap_kernel(const int* __restrict__ D, const int* __restrict__ Dptr,
                                                                    const int2* __restrict__ P, const int4* __restrict__ tasks,
                                                                    unsigned long long* total) {
  __shared__ unsigned bits[kBits / 32 + 1];
  __shared__ int next;
  __shared__ unsigned part[kThreads / kWave];
  constexpr int kWaves = kThreads / kWave;
  const int tid = threadIdx.x, lane = lane_id();
  const int4 task = tasks[blockIdx.x];
  const int s = Dptr[task.x], len = Dptr[task.x + 1] - s;
  const int nwords = (task.x + 31) / 32;        // task.x <= kBits
  for (int i = tid; i <= nwords; i += kThreads) bits[i] = 0u;     // word nwords stays zero: every number from 32 nwords on looks there
  if (tid == 0) next = kWaves;
  __syncthreads();
  for (int i = tid; i < len; i += kThreads) {
    const unsigned x = (unsigned)D[s + i];
    if (x != kTcPad) atomicOr(&bits[x >> 5], 1u << (x & 31));     // (len counts the list's room: up to three entries of filling)
  }
  __syncthreads();
  const unsigned top = (unsigned)nwords;
  unsigned count = 0;
  const int nbatch = (task.z + kWave - 1) / kWave;
  int b = wave_id();
  while (b < nbatch) {
    const int nb = task.z - b * kWave < kWave ? task.z - b * kWave : kWave;
    const int2 my = lane < nb ? P[task.y + b * kWave + lane] : make_int2(0, 0);
    tc_stream_batch<GRB_TC_DEPTH>(D, my, nb, lane, [&](unsigned x) {
      const unsigned w = x >> 5;
      count += (bits[w < top ? w : top] >> (x & 31)) & 1u;
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
constexpr int kTcChunk[3] = {256, 2048, 2048};   // partners per task, by kernel

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

// (pbase: the degree of the vertex numbered r -- room for its partners, whoever they turn out to be; scanned afterwards)
__global__ __launch_bounds__(kBlock) void tc_number_kernel(const unsigned* __restrict__ order, const unsigned long long* __restrict__ key,
                                                           Index n, int* __restrict__ number, unsigned* __restrict__ pbase) {
  const Index r = (Index)blockIdx.x * kBlock + threadIdx.x;
  if (r < n) { number[order[r]] = (int)r; pbase[r] = 0xffffffffu - (unsigned)key[r]; }
}

// pass A: every entry (i, j) as {lower-ranked end, higher-ranked end} in the new numbers; the lower end's list grows by one.
// bad: an entry on or above the diagonal, or a value that is not 1 -- the sum of the product is then not a count
__global__ __launch_bounds__(kBlock) void tc_orient_kernel(const int* __restrict__ erow, const Index* __restrict__ ind,
                                                           const unsigned* __restrict__ val, unsigned one, long long nnz,
                                                           const int* __restrict__ number, int* __restrict__ elo, int* __restrict__ ehi,
                                                           unsigned* __restrict__ cnt, int* __restrict__ bad) {
  const long long stride = (long long)gridDim.x * kBlock;
  bool wrong = false;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += stride) {
    const int i = erow[e], j = (int)ind[e];
    wrong |= j >= i || val[e] != one;
    const int a = number[i], b = number[j];
    const int lo = a > b ? a : b, hi = a > b ? b : a;
    elo[e] = lo; ehi[e] = hi;
    atomicAdd(&cnt[lo], 1u);
  }
  if (__any(wrong) && lane_id() == 0) atomicOr(bad, 1);
}

// a list's length, and its room in D: the next multiple of four entries (before the scan that makes the counts positions)
__global__ __launch_bounds__(kBlock) void tc_lengths_kernel(unsigned* __restrict__ cnt, Index n, int* __restrict__ len, int* __restrict__ longest) {
  const Index v = (Index)blockIdx.x * kBlock + threadIdx.x;
  int l = 0;
  if (v < n) { l = (int)cnt[v]; len[v] = l; cnt[v] = (unsigned)(l + 3) & ~3u; }
#pragma unroll
  for (int off = kWave / 2; off; off >>= 1) { const int o = __shfl_down(l, off); l = o > l ? o : l; }
  if (lane_id() == 0 && l > 0) atomicMax(longest, l);
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
// pivot's partners as {first element of the list in D, length}, in the room its degree reserves (dcur / pcur start as
// copies of Dptr / pbase: the atomics hand out absolute positions)
__global__ __launch_bounds__(kBlock) void tc_lists_kernel(const int* __restrict__ elo, const int* __restrict__ ehi, long long nnz,
                                                          const unsigned* __restrict__ Dptr, const int* __restrict__ len,
                                                          unsigned* __restrict__ dcur, int* __restrict__ D,
                                                          unsigned* __restrict__ pcur, int2* __restrict__ P) {
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += stride) {
    const int lo = elo[e], hi = ehi[e];
    D[atomicAdd(&dcur[lo], 1u)] = hi;
    int pivot, partner, plen;
    tc_roles(lo, hi, len, &pivot, &partner, &plen);
    if (plen > 0) P[atomicAdd(&pcur[pivot], 1u)] = make_int2((int)Dptr[partner], plen);
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

struct TcPrep {
  int state = 0;                              // 0 not tried, 1 ready, -1 the sum of this matrix's product is not such a count
  int* D = nullptr;                           // the lists, end to end (+ 8 entries: a step reads whole 16-byte groups)
  int* Dptr = nullptr;                        // [n + 1]
  int2* P = nullptr;                          // the partners, pivot by pivot
  int4* tasks[3] = {nullptr, nullptr, nullptr};
  int ntasks[3] = {0, 0, 0};
  int longest = 0;
  float prep_ms = 0.f;
};

static int g_tc_product = -1;                 // grb_tc_set_product
static struct { int path; float prep_ms, count_ms; int longest; int ntasks[3]; } g_tc_last = {0, 0.f, 0.f, 0, {0, 0, 0}};

void tc_prep_free(grb_matrix_s* A) {
  TcPrep* t = (TcPrep*)A->tc_prep;
  if (!t) return;
  for (void* q : {(void*)t->D, (void*)t->Dptr, (void*)t->P, (void*)t->tasks[0], (void*)t->tasks[1], (void*)t->tasks[2]})
    if (q) (void)hipFree(q);
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
  const long long nnz = A->nvals;
  t->state = -1;
  if (A->csc_alias || !A->csc.ptr || !A->csr.val || n != A->ncols || nnz < 1 || nnz > 0x7ffffff0ll) return GRB_SUCCESS;
  TcTemps tmp;
  int *erow, *number, *elo, *ehi, *len, *flags;
  unsigned long long* key;
  unsigned *order, *cur, *pptr, *pcur, *c0, *c1, *c2;
  GRB_TRY(tmp.get(&erow, (size_t)nnz));
  GRB_TRY(tmp.get(&elo, (size_t)nnz));
  GRB_TRY(tmp.get(&ehi, (size_t)nnz));
  GRB_TRY(tmp.get(&number, (size_t)n));
  GRB_TRY(tmp.get(&len, (size_t)n));
  GRB_TRY(tmp.get(&key, (size_t)n));
  GRB_TRY(tmp.get(&order, (size_t)n));
  GRB_TRY(tmp.get(&cur, (size_t)n));
  GRB_TRY(tmp.get(&pptr, (size_t)n + 1));
  GRB_TRY(tmp.get(&pcur, (size_t)n));
  GRB_TRY(tmp.get(&c0, (size_t)n + 1));
  GRB_TRY(tmp.get(&c1, (size_t)n + 1));
  GRB_TRY(tmp.get(&c2, (size_t)n + 1));
  GRB_TRY(tmp.get(&flags, 2));                // {bad, longest list}
  GRB_HIP_TRY(hipMalloc((void**)&t->Dptr, 4 * ((size_t)n + 1)));
  unsigned* const dptr = (unsigned*)t->Dptr;
  const int vgrid = (int)((n + kBlock - 1) / kBlock), egrid = stream_grid(nnz, kBlock * 4);
  // the numbering: by degree, the highest first, ties by the caller's number (the sort is stable)
  hipLaunchKernelGGL(tc_degree_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const Index*)A->csr.ptr, (const Index*)A->csc.ptr, n, key, order);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_sort_pairs(key, order, n, 32, 0));
  GRB_HIP_TRY(hipMemsetAsync(pptr + n, 0, 4, s));
  hipLaunchKernelGGL(tc_number_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const unsigned*)order, (const unsigned long long*)key, n, number, pptr);
  hipLaunchKernelGGL(tc_rows_kernel, dim3(stream_grid((long long)n * 16, kBlock)), dim3(kBlock), 0, s, (const Index*)A->csr.ptr, n, erow);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipMemsetAsync(dptr, 0, 4 * ((size_t)n + 1), s));
  GRB_HIP_TRY(hipMemsetAsync(flags, 0, 8, s));
  const unsigned one = A->dtype == GRB_F32 ? 0x3f800000u : 1u;
  hipLaunchKernelGGL(tc_orient_kernel, dim3(egrid), dim3(kBlock), 0, s, (const int*)erow, (const Index*)A->csr.ind,
                     (const unsigned*)A->csr.val, one, nnz, (const int*)number, elo, ehi, dptr, flags);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(tc_lengths_kernel, dim3(vgrid), dim3(kBlock), 0, s, dptr, n, len, flags + 1);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_exclusive_scan_u32(dptr, (long long)n + 1));
  GRB_TRY(device_exclusive_scan_u32(pptr, (long long)n + 1));
  int h_flags[2] = {0, 0};
  unsigned room = 0;
  GRB_HIP_TRY(hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(&room, dptr + n, 4, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  if (h_flags[0]) return GRB_SUCCESS;         // not a strictly lower triangle of ones
  t->longest = h_flags[1];
  GRB_HIP_TRY(hipMalloc((void**)&t->D, 4 * ((size_t)room + 8)));
  GRB_HIP_TRY(hipMalloc((void**)&t->P, 8 * (2 * (size_t)nnz + 1)));          // (a vertex's room: its degree)
  GRB_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)t->D, (int)kTcPad, (size_t)room + 8, s));
  GRB_HIP_TRY(hipMemcpyAsync(cur, dptr, 4 * (size_t)n, hipMemcpyDeviceToDevice, s));
  GRB_HIP_TRY(hipMemcpyAsync(pcur, pptr, 4 * (size_t)n, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(tc_lists_kernel, dim3(egrid), dim3(kBlock), 0, s, (const int*)elo, (const int*)ehi, nnz, (const unsigned*)dptr,
                     (const int*)len, cur, t->D, pcur, t->P);
  GRB_HIP_TRY(hipGetLastError());
  // the tasks (GRB_TC_BITMAP_UPTO: tests send the pivots beyond a smaller number to the hash-table kernel)
  int bitmap_upto = kTcBits;
  if (const char* e = getenv("GRB_TC_BITMAP_UPTO")) { const int v = atoi(e); if (v >= 0 && v < kTcBits) bitmap_upto = v; }
  for (unsigned* c : {c0, c1, c2}) GRB_HIP_TRY(hipMemsetAsync(c + n, 0, 4, s));
  hipLaunchKernelGGL(tc_task_count_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const int*)len, (const unsigned*)pptr, (const unsigned*)pcur, n, bitmap_upto, c0, c1, c2);
  GRB_HIP_TRY(hipGetLastError());
  unsigned nt[3] = {0, 0, 0};
  int k = 0;
  for (unsigned* c : {c0, c1, c2}) {
    GRB_TRY(device_exclusive_scan_u32(c, (long long)n + 1));
    GRB_HIP_TRY(hipMemcpyAsync(&nt[k++], c + n, 4, hipMemcpyDeviceToHost, s));
  }
  GRB_HIP_TRY(hipStreamSynchronize(s));
  if (nt[2] > 0 && t->longest > kTcHashLen) return GRB_SUCCESS;     // a list no table here holds
  for (k = 0; k < 3; ++k) {
    t->ntasks[k] = (int)nt[k];
    GRB_HIP_TRY(hipMalloc((void**)&t->tasks[k], 16 * ((size_t)nt[k] + 1)));
  }
  hipLaunchKernelGGL(tc_task_fill_kernel, dim3(vgrid), dim3(kBlock), 0, s, (const int*)len, (const unsigned*)pptr, (const unsigned*)pcur, n, bitmap_upto, (const unsigned*)c0,
                     (const unsigned*)c1, (const unsigned*)c2, t->tasks[0], t->tasks[1], t->tasks[2]);
  GRB_HIP_TRY(hipGetLastError());
  GRB_HIP_TRY(hipStreamSynchronize(s));
  t->state = 1;
  return GRB_SUCCESS;
}

int tc_product_setting(int set) {
  if (g_tc_product < 0) { const char* e = getenv("GRB_TC_PRODUCT"); g_tc_product = e && atoi(e) != 0 ? 1 : 0; }
  const int prev = g_tc_product;
  if (set >= 0) g_tc_product = set ? 1 : 0;
  return prev;
}

// *done = true: *count is the sum of L (+.x) L^T over the entries of L.  false: A is not a strictly lower triangle of ones
// (or holds a list no table here takes) -- the caller forms the product.
grb_info tc_count_try(grb_matrix_s* A, long long* count, bool* done) {
  *done = false;
  g_tc_last.path = 0;
  if (tc_product_setting(-1)) return GRB_SUCCESS;
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
    const grb_info info = tc_prepare(A, t);
    if (info != GRB_SUCCESS || t->state != 1) {
      // (whatever was allocated goes; the verdict stays: the next call does not try again)
      for (void** q : {(void**)&t->D, (void**)&t->Dptr, (void**)&t->P, (void**)&t->tasks[0], (void**)&t->tasks[1], (void**)&t->tasks[2]})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
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
  if (t->ntasks[1] > 0)
    hipLaunchKernelGGL((tc_count_bitmap_kernel<512, kTcBits>), dim3(t->ntasks[1]), dim3(512), 0, s, (const int*)t->D, (const int*)t->Dptr,
                       (const int2*)t->P, (const int4*)t->tasks[1], slots);
  if (t->ntasks[2] > 0)
    hipLaunchKernelGGL((tc_count_pivot_kernel<512, 2 * kTcHashLen>), dim3(t->ntasks[2]), dim3(512), 0, s, (const int*)t->D, (const int*)t->Dptr,
                       (const int2*)t->P, (const int4*)t->tasks[2], slots);
  if (t->ntasks[0] > 0)
    hipLaunchKernelGGL((tc_count_pivot_kernel<64, 2 * kTcWaveLen>), dim3(t->ntasks[0]), dim3(64), 0, s, (const int*)t->D, (const int*)t->Dptr,
                       (const int2*)t->P, (const int4*)t->tasks[0], slots);
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

extern "C" grb_info grb_tc_last(grb_tc_info* out) { GRB_API_ENTER_HOST();
  if (!out) return GRB_NULL_POINTER;
  out->path = g_tc_last.path;
  out->prep_ms = g_tc_last.path ? g_tc_last.prep_ms : 0.f;
  out->count_ms = g_tc_last.path ? g_tc_last.count_ms : 0.f;
  out->longest_list = g_tc_last.path ? g_tc_last.longest : 0;
  for (int k = 0; k < 3; ++k) out->tasks[k] = g_tc_last.path ? g_tc_last.ntasks[k] : 0;
  return GRB_SUCCESS;
}
