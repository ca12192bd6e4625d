// algorithms_ext.hip -- the remaining drivers of graphblas/algorithm/ (SURVEY.md 8(f)4):
// maximal independent set, graph colouring (IS / MIS / Jones-Plassmann), local graph
// clustering and BFS eccentricity ("diameter"), plus the two extension operations only they
// use: scatter (operations.hpp:748-761) and graphColor (operations.hpp:816-826).
//
// mis / gc / lgc: the reference's loops give a result that is a function of the graph and the
// weight vector only when every vector stays dense (its --mxvmode 2).  In its push and
// push-pull modes the weight vector -- which is mask and input of the same vxm -- is
// converted to sparse storage in place and the loop then runs through "not implemented"
// branches (sparse mask, sparse-sparse eWiseAdd) that leave the frontier stale: restated op
// by op (oracle/algorithms.py) it does not terminate.  The drivers here therefore pin
// GrB_MXVMODE to GrB_PULLONLY for the duration of the call and restore it; applications
// that compile the reference's own algorithm/mis.hpp against the C++ frontend get the
// reference's behaviour for whatever mode they select.
#include <climits>
#include <cmath>
#include <vector>

#include "common.hpp"

using namespace grb;

namespace {
struct VecGuard {               // frees temporaries on every exit path
  std::vector<grb_vector> v;
  ~VecGuard() { for (grb_vector x : v) grb_vector_free(x); }
  grb_info make(grb_vector* out, grb_dtype dt, Index n) {
    grb_info i = grb_vector_new(out, dt, n);
    if (i == GRB_SUCCESS) v.push_back(*out);
    return i;
  }
};
struct ModeGuard {              // GrB_MXVMODE pinned for a scope
  grb_descriptor d;
  int saved = GRB_PUSHPULL;
  explicit ModeGuard(grb_descriptor desc, int mode) : d(desc) {
    grb_descriptor_get(d, GRB_MXVMODE, &saved);
    grb_descriptor_set(d, GRB_MXVMODE, mode);
  }
  ~ModeGuard() { grb_descriptor_set(d, GRB_MXVMODE, saved); }
};

// w[(Index)u[k]] = val for 0 < u[k] < bound   (scatterKernel, kernels/scatter.hpp:7-21)
template <typename T, typename U>
__global__ void scatter_const_kernel(T* __restrict__ w, Index bound, const U* __restrict__ u, Index n, T val) {
  for (Index k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const Index i = static_cast<Index>(u[k]);
    if (i > 0 && i < bound) w[i] = val;
  }
}

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// ---- colouring by rounds of local maxima, as a dataflow over the priority DAG ----
// v is chosen in the round after its last BLOCKER was: a blocker of v is a neighbour u != v that is
// a candidate and has priority(u) >= priority(v) (the reference's test is w[v] > max w[u], so an
// equal neighbour blocks as well -- for ever).  blockers[v] counts them; a round colours its
// frontier (the vertices whose count reached 0 in the round before) and releases the neighbours
// they block, appending those that reach 0 to the next frontier.  Every edge is looked at twice in
// total (count, release) instead of once per round, and a round costs what its frontier costs.
//   kMode 0: colour = the round number                                  (algorithm::gcIS, gc.hpp:43-149)
//   kMode 1: colour = the smallest colour no neighbour holds            (Jones-Plassmann; graphColor)
//   kMode 2: ONE colour per round, the smallest held by no coloured neighbour of ANY frontier vertex
//            (algorithm::gcJP, gc.hpp:258-421).  A vertex stores its round; the round's colour is
//            settled by the last workgroup to finish the round (a bitmap of the colours seen, OR-ed by
//            everybody) and read through round_colour[] from then on.
// Frontier vertices of one round are never adjacent (one would block the other), so first-fit reads
// settled colours only.  Frontier sizes live on the device in three rotating counters (read / appended
// to / being zeroed): the host queues a batch of rounds and looks at the sizes once per batch.
constexpr int kColourWindow = 2048;                       // colours examined per first-fit pass
constexpr int kColourBatch = 16;                          // rounds queued between two host looks
constexpr int kNbrUnroll = 4;                             // neighbour loads in flight per lane
constexpr int kLdsSeenColours = 65536;                    // gcJP: colours a wave can collect in LDS (8 KiB)
constexpr int kSplitItems = 2048;                         // frontier positions with a shared forbidden set

// priority of u against v: does u block v?  weights: the reference's strict test; hash: total order
__device__ __forceinline__ bool blocks(unsigned pu, Index u, unsigned pv, Index v, bool by_weight) {
  if (by_weight) return pu != 0u && pu >= pv;             // weight 0 = not a candidate (gc.hpp:66-70)
  return pu > pv || (pu == pv && u > v);
}
__device__ __forceinline__ unsigned priority(const int* weights, Index v) {
  return weights ? (unsigned)weights[v] : mix32((unsigned)v);
}

// blockers[v] for every vertex; the vertices with none (and a non-zero weight) form round 1's frontier
__global__ __launch_bounds__(kBlock) void colour_init_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                             Index n, const int* __restrict__ weights,
                                                             int* __restrict__ blockers, Index* __restrict__ first,
                                                             int* __restrict__ first_count) {
  __shared__ Index requeue[kWavesPerBlock][kWave];
  int nrequeue = 0;
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = threadIdx.x / kWave;
  const Index wave = (Index)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / kWave);
  const Index nwaves = (Index)((gridDim.x * (unsigned)blockDim.x) / kWave);
  for (Index v = wave; v < n; v += nwaves) {
    const unsigned pv = priority(weights, v);
    const Index b = ptr[v], e = ptr[v + 1];
    int cnt = 0;
    for (Index p = b + lane; p < e; p += kWave * kNbrUnroll) {
      Index u[kNbrUnroll];
      unsigned pu[kNbrUnroll];
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) u[k] = p + k * kWave < e ? ind[p + k * kWave] : v;
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) pu[k] = priority(weights, u[k]);
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) cnt += (u[k] != v && blocks(pu[k], u[k], pv, v, weights != nullptr)) ? 1 : 0;
    }
    cnt = wave_reduce(cnt, [](int a, int c) { return a + c; });
    cnt = __shfl(cnt, 0, kWave);
    const bool never = weights && pv == 0u;
    if (lane == 0) blockers[v] = never ? INT_MAX : cnt;
    if (cnt == 0 && !never) {
      if (lane == 0) requeue[wib][nrequeue] = v;
      if (++nrequeue == kWave) {
        int pos = 0;
        if (lane == 0) pos = atomicAdd(first_count, kWave);
        pos = __shfl(pos, 0, kWave);
        __builtin_amdgcn_wave_barrier();
        first[pos + lane] = requeue[wib][lane];
        __builtin_amdgcn_wave_barrier();
        nrequeue = 0;
      }
    }
  }
  if (nrequeue) {
    int pos = 0;
    if (lane == 0) pos = atomicAdd(first_count, nrequeue);
    pos = __shfl(pos, 0, kWave);
    __builtin_amdgcn_wave_barrier();
    if (lane < nrequeue) first[pos + lane] = requeue[wib][lane];
  }
}

template <int kMode>
__global__ __launch_bounds__(kBlock) void colour_round_kernel(
    const Index* __restrict__ ptr, const Index* __restrict__ ind, const int* __restrict__ weights,
    int* __restrict__ colour, int* __restrict__ blockers, const Index* __restrict__ cur, Index* __restrict__ next,
    int* __restrict__ list_count /* [3] rotating */, int* __restrict__ frontier_size /* per round of the batch */,
    unsigned int* __restrict__ shared_forb /* [kSplitItems][64], zero between rounds */,
    int* __restrict__ arrive /* [kSplitItems], zero between rounds */,
    unsigned int* __restrict__ seen /* kMode 2: max_colors bits, zero between rounds */,
    int* __restrict__ round_colour /* kMode 2 */, int max_colors, unsigned int* __restrict__ tickets, int round,
    int slot) {
  constexpr bool kFirstFit = kMode == 1;
  constexpr int kLdsSeenWords = kMode == 2 ? kLdsSeenColours / 32 : 1;
  __shared__ unsigned int lds_seen[kWavesPerBlock][kLdsSeenWords];
  const bool use_lds_seen = kMode == 2 && max_colors <= kLdsSeenColours;
  bool touched_seen = false;
  if (use_lds_seen) {
    for (int k = threadIdx.x & (kWave - 1); k < (max_colors + 31) / 32; k += kWave) lds_seen[threadIdx.x / kWave][k] = 0u;
    __builtin_amdgcn_wave_barrier();
  }
  __shared__ unsigned int forb[kWavesPerBlock][kColourWindow / 32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = threadIdx.x / kWave;
  const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / kWave);
  const int nwaves = (int)((gridDim.x * (unsigned)blockDim.x) / kWave);
  const int ncur = list_count[round % 3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    list_count[(round + 2) % 3] = 0;                      // the round after next appends there
    frontier_size[slot] = ncur;
  }
  int* next_count = &list_count[(round + 1) % 3];
  // a small frontier leaves most of the launch idle while one wave walks a hub's list, so the list
  // is cut into `split` interleaved slices, one wave each.  Releasing neighbours needs nothing from
  // the other slices; first-fit does: the slices OR the colours they see into a forbidden set in
  // global memory and the last one to arrive (a ticket per frontier position) picks the colour.
  int split = 1;
  while (split < 32 && (long long)ncur * split * 2 <= nwaves && (!kFirstFit || ncur <= kSplitItems)) split *= 2;
  const long long nitems = (long long)ncur * split;
  for (long long item = wave; item < nitems; item += nwaves) {
    const Index v = cur[item / split];
    const int slice = (int)(item % split);
    const unsigned pv = priority(weights, v);
    const Index b = ptr[v], e = ptr[v + 1];
    // only lists long enough to be worth it are cut (a slice = at least one full pass of the wave)
    const int vsplit = min(split, max(1, (int)((e - b + kWave * kNbrUnroll - 1) / (kWave * kNbrUnroll))));
    if (slice >= vsplit) continue;
    int mine = round;
    int first_base = 1;
    bool choose = slice == 0;                              // this wave writes colour[v]
    if constexpr (kFirstFit) {
      if (vsplit > 1) {
        const int fi = (int)(item / split);
        unsigned int* fb = shared_forb + (size_t)fi * (kColourWindow / 32);
        // collected in LDS first: the colours of a hub's neighbours fall into a handful of words, and
        // same-address global atomics serialise
        forb[wib][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (Index p = b + (Index)slice * kWave * kNbrUnroll + lane; p < e; p += (Index)vsplit * kWave * kNbrUnroll) {
          int cu[kNbrUnroll];
#pragma unroll
          for (int k = 0; k < kNbrUnroll; ++k) cu[k] = p + k * kWave < e ? colour[ind[p + k * kWave]] : 0;
#pragma unroll
          for (int k = 0; k < kNbrUnroll; ++k)
            if (cu[k] >= 1 && cu[k] <= kColourWindow) atomicOr(&forb[wib][(cu[k] - 1) >> 5], 1u << ((cu[k] - 1) & 31));
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned seen = forb[wib][lane];
        if (seen) atomicOr(&fb[lane], seen);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ORs above are L2 atomics: drained = visible
        int ticket = 0;
        if (lane == 0) ticket = atomicAdd(&arrive[fi], 1);
        ticket = __shfl(ticket, 0, kWave);
        choose = ticket == vsplit - 1;
        if (choose) {
          const unsigned freebits = ~__hip_atomic_load(&fb[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fb[lane] = 0u;                                   // left clean for the next round
          if (lane == 0) arrive[fi] = 0;
          const unsigned long long any = __ballot(freebits != 0u);
          if (any) {
            const int wl = __ffsll((long long)any) - 1;
            const unsigned fbw = __shfl(freebits, wl, kWave);
            mine = wl * 32 + __ffs((int)fbw);              // colours from 1
            first_base = 0;                                // settled: skip the windowed search below
          } else {
            first_base = 1 + kColourWindow;                // every colour of the first window is taken
          }
        } else {
          first_base = 0;
        }
      }
      if (first_base > 0) mine = 0;
      for (int base = first_base; first_base > 0 && mine == 0; base += kColourWindow) {
        for (int k = lane; k < kColourWindow / 32; k += kWave) forb[wib][k] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (Index p = b + lane; p < e; p += kWave * kNbrUnroll) {
          int cu[kNbrUnroll];
#pragma unroll
          for (int k = 0; k < kNbrUnroll; ++k) cu[k] = p + k * kWave < e ? colour[ind[p + k * kWave]] : 0;
#pragma unroll
          for (int k = 0; k < kNbrUnroll; ++k)
            if (cu[k] >= base && cu[k] < base + kColourWindow)
              atomicOr(&forb[wib][(cu[k] - base) >> 5], 1u << ((cu[k] - base) & 31));
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned freebits = ~forb[wib][lane];        // kColourWindow / 32 == kWave words
        const unsigned long long any = __ballot(freebits != 0u);
        if (any) {
          const int wl = __ffsll((long long)any) - 1;
          const unsigned fb = __shfl(freebits, wl, kWave);
          mine = base + wl * 32 + (__ffs((int)fb) - 1);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (lane == 0 && choose) colour[v] = mine;
    // release the neighbours v was blocking
    for (Index p = b + (Index)slice * kWave * kNbrUnroll + lane; p < e; p += (Index)vsplit * kWave * kNbrUnroll) {
      Index u[kNbrUnroll];
      unsigned pu[kNbrUnroll];
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) u[k] = p + k * kWave < e ? ind[p + k * kWave] : v;
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) pu[k] = priority(weights, u[k]);
      if constexpr (kMode == 2) {
        // colours around the frontier: neighbours coloured in an earlier round (gc.hpp:356-364; only
        // colours in (0, max_colors) reach the dense array there).  Collected per wave in LDS (bits only
        // ever get set, so a set bit is not set again) and flushed once at the end of the kernel.
        int ru[kNbrUnroll];
#pragma unroll
        for (int k = 0; k < kNbrUnroll; ++k) ru[k] = u[k] != v ? colour[u[k]] : 0;
#pragma unroll
        for (int k = 0; k < kNbrUnroll; ++k) {
          if (ru[k] > 0 && ru[k] < round) {
            const int cu = round_colour[ru[k]];
            if (cu > 0 && cu < max_colors) {
              if (use_lds_seen) {
                if (!((lds_seen[wib][cu >> 5] >> (cu & 31)) & 1u)) atomicOr(&lds_seen[wib][cu >> 5], 1u << (cu & 31));
              } else if (!((seen[cu >> 5] >> (cu & 31)) & 1u)) {
                atomicOr(&seen[cu >> 5], 1u << (cu & 31));
              }
            }
          }
        }
        touched_seen = true;
      }
#pragma unroll
      for (int k = 0; k < kNbrUnroll; ++k) {
        // v blocks u  <=>  blocks(pv, v, pu[k], u[k]); a weight-0 u holds INT_MAX and never reaches 0
        const bool freed = u[k] != v && blocks(pv, v, pu[k], u[k], weights != nullptr) &&
                           atomicSub(&blockers[u[k]], 1) == 1;
        const unsigned long long m = __ballot(freed);      // one append per wave, not per vertex
        if (m) {
          int pos = 0;
          if (lane == __ffsll((long long)m) - 1) pos = atomicAdd(next_count, __popcll(m));
          pos = __shfl(pos, __ffsll((long long)m) - 1, kWave);
          if (freed) next[pos + __popcll(m & ((1ull << lane) - 1ull))] = u[k];
        }
      }
    }
  }
  if constexpr (kMode == 2) {
    if (use_lds_seen && __any(touched_seen)) {               // wave-uniform: every lane owns words to flush
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < (max_colors + 31) / 32; k += kWave) {
        const unsigned int bits = lds_seen[wib][k];
        if (bits) atomicOr(&seen[k], bits);
      }
    }
    // the last workgroup to get here settles this round's colour: the smallest index in [1, max_colors)
    // nobody saw, max_colors if there is none (min_array[0] = max_colors, gc.hpp:380-385)
    __shared__ int s_last, s_best[kWavesPerBlock];
    // everything the settle step reads was written by device-scope atomics (they execute at L2), so
    // draining this wave's outstanding operations is all the ordering the ticket needs -- a
    // __threadfence() here is an L2 write-back per workgroup per round
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = last_workgroup_arrives(tickets) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const int nw = (max_colors + 31) / 32;
    int best = max_colors;
    for (int w = threadIdx.x; w < nw; w += kBlock) {
      unsigned int bits = __hip_atomic_load(&seen[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      seen[w] = 0u;                                        // clean for the next round
      if (w == 0) bits |= 1u;                              // colour 0 is "uncoloured"
      if (~bits) {
        const int c = w * 32 + (__ffs((int)~bits) - 1);
        if (c < best) best = c;
      }
    }
    best = wave_reduce(best, [](int a, int b) { return a < b ? a : b; });
    if (lane_id() == 0) s_best[wave_id()] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kWavesPerBlock; ++w) best = s_best[w] < best ? s_best[w] : best;
      round_colour[round] = best < max_colors ? best : max_colors;
    }
  }
}

__global__ void clip_colours_kernel(int* __restrict__ colour, Index n, int max_colour) {
  for (Index v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
    if (colour[v] > max_colour) colour[v] = 0;
}

__global__ void colour_extract_kernel(const int* __restrict__ colour, const int* __restrict__ round_colour,
                                      int* __restrict__ out, Index n, int minus) {
  for (Index v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const int c = colour[v];
    out[v] = (round_colour && c > 0 ? round_colour[c] : c) - minus;
  }
}

// Runs rounds until a frontier is empty (or max_rounds is reached).  colours_out: int vector, colours
// from 1 (0 = never coloured) minus `minus`.  rounds_out = the reference's `iter` (1 + rounds that
// coloured something).
template <int kMode>
grb_info colour_by_rounds(grb_matrix A, const int* d_weights, int max_rounds, grb_vector colours_out, int minus,
                          int* rounds_out, int max_colors = 0) {
  constexpr bool kFirstFit = kMode == 1;
  static_assert(kColourWindow / 32 == kWave, "one forbidden-set word per lane");
  const Index n = A->nrows;
  Context& c = ctx();
  VecGuard g;
  grb_vector col, blk, la, lb, cnt;
  GRB_TRY(g.make(&col, GRB_I32, n));
  GRB_TRY(g.make(&blk, GRB_I32, n));
  GRB_TRY(g.make(&la, GRB_I32, n));
  GRB_TRY(g.make(&lb, GRB_I32, n));
  GRB_TRY(g.make(&cnt, GRB_I32, 64));
  GRB_TRY(grb_vector_fill(col, 0.0));
  GRB_TRY(grb_vector_fill(cnt, 0.0));
  grb_vector sf = nullptr, arr = nullptr;
  if (kFirstFit) {
    GRB_TRY(g.make(&sf, GRB_I32, kSplitItems * (kColourWindow / 32)));
    GRB_TRY(g.make(&arr, GRB_I32, kSplitItems));
    GRB_TRY(grb_vector_fill(sf, 0.0));
    GRB_TRY(grb_vector_fill(arr, 0.0));
  }
  grb_vector seen = nullptr, rcol = nullptr;
  if (kMode == 2) {
    GRB_TRY(g.make(&seen, GRB_I32, (max_colors + 31) / 32 + 1));
    GRB_TRY(g.make(&rcol, GRB_I32, 65536));
    GRB_TRY(grb_vector_fill(seen, 0.0));
    GRB_TRY(grb_vector_fill(rcol, 0.0));
  }
  int* counters = (int*)cnt->d_val;                        // [0..2] frontier sizes, [8..8+batch) per round
  int round = 1, iter = 1;
  bool done = n == 0;
  if (!done) {
    // round 1 reads list la through counter 1 % 3
    hipLaunchKernelGGL(colour_init_kernel, dim3(stream_grid((long long)n * kWave, kBlock)), dim3(kBlock), 0, c.stream,
                       A->csr.ptr, A->csr.ind, n, d_weights, (int*)blk->d_val, (Index*)la->d_val, counters + 1);
    GRB_HIP_TRY(hipGetLastError());
  }
  long long remaining = n;
  while (!done && round <= max_rounds) {
    GRB_HIP_TRY(hipMemsetAsync(counters + 8, 0, sizeof(int) * kColourBatch, c.stream));
    const int first = round;
    // a frontier can hold every remaining vertex at most; later batches shrink with it
    int grid = stream_grid((remaining > 0 ? remaining : 1) * kWave, kBlock);
    if (grid < c.num_cu) grid = c.num_cu;                  // idle waves are what long lists get split over
    for (int k = 0; k < kColourBatch && round <= max_rounds; ++k, ++round) {
      const Index* cur = (const Index*)((round & 1) ? la->d_val : lb->d_val);
      Index* nxt = (Index*)((round & 1) ? lb->d_val : la->d_val);
      hipLaunchKernelGGL((colour_round_kernel<kMode>), dim3(grid), dim3(kBlock), 0, c.stream, A->csr.ptr, A->csr.ind,
                         d_weights, (int*)col->d_val, (int*)blk->d_val, cur, nxt, counters, counters + 8,
                         sf ? (unsigned int*)sf->d_val : nullptr, arr ? (int*)arr->d_val : nullptr,
                         seen ? (unsigned int*)seen->d_val : nullptr, rcol ? (int*)rcol->d_val : nullptr, max_colors,
                         c.d_tickets, round, k);
      GRB_HIP_TRY(hipGetLastError());
    }
    int h[kColourBatch];
    GRB_TRY(fetch_ints(counters + 8, kColourBatch, h));
    for (int k = 0; k < round - first; ++k) {
      if (h[k] == 0) { done = true; break; }               // succ == 0: the loop of gc.hpp:117-120 breaks
      ++iter;
      remaining -= h[k];
    }
  }
  hipLaunchKernelGGL(colour_extract_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, c.stream, (const int*)col->d_val,
                     rcol ? (const int*)rcol->d_val : (const int*)nullptr, (int*)colours_out->d_val, n, minus);
  GRB_HIP_TRY(hipGetLastError());
  if (rounds_out) *rounds_out = iter;
  return GRB_SUCCESS;
}
}  // namespace

extern "C" {

// scatter (extension; operations.hpp:748-761 -> backend/cuda/operations.hpp:1110-1142 + scatter.hpp:10-82).
// The dense variant of the reference passes u's length as the bound of w (scatter.hpp:38); a target
// past w's own end would be an out-of-bounds store there, so w's size bounds it here as well.
grb_info grb_scatter(grb_vector w, grb_vector mask, grb_vector u, double val, grb_descriptor desc) { GRB_API_ENTER();
  if (!w || !u) return GRB_UNINITIALIZED_OBJECT;
  (void)desc;
  const int ut = u->vec_type;
  GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
  if (ut != GRB_SPARSE && ut != GRB_DENSE) return GRB_UNINITIALIZED_OBJECT;
  if (mask) return GRB_SUCCESS;                            // "Masked variant scatter not implemented yet"
  const void* src = ut == GRB_DENSE ? u->d_val : u->s_val;
  const Index n = ut == GRB_DENSE ? u->nsize : u->s_nvals;
  const Index bound = ut == GRB_DENSE ? (u->nsize < w->nsize ? u->nsize : w->nsize) : w->nsize;
  if (n <= 0) return GRB_SUCCESS;
  Context& c = ctx();
  const dim3 grid(stream_grid(n)), block(kBlock);
  if (w->dtype == GRB_F32 && u->dtype == GRB_F32)
    hipLaunchKernelGGL((scatter_const_kernel<float, float>), grid, block, 0, c.stream, (float*)w->d_val, bound,
                       (const float*)src, n, (float)val);
  else if (w->dtype == GRB_F32)
    hipLaunchKernelGGL((scatter_const_kernel<float, int>), grid, block, 0, c.stream, (float*)w->d_val, bound,
                       (const int*)src, n, (float)val);
  else if (u->dtype == GRB_F32)
    hipLaunchKernelGGL((scatter_const_kernel<int, float>), grid, block, 0, c.stream, (int*)w->d_val, bound,
                       (const float*)src, n, (int)val);
  else
    hipLaunchKernelGGL((scatter_const_kernel<int, int>), grid, block, 0, c.stream, (int*)w->d_val, bound,
                       (const int*)src, n, (int)val);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

// graphColor (operations.hpp:816-826 -> backend/cuda/color.hpp:18-88).  The reference hands the CSR
// to cuSPARSE csrcolor (closed source, not in the tree); what its callers rely on
// (example/ggc_cusparse.cu:94-99) is a proper colouring with colours counted from 0.  This is
// Jones-Plassmann with a hashed priority and first-fit colours (colour_by_rounds above).  w: int or float vector of length nrows.
grb_info grb_graph_color(grb_vector w, grb_matrix A, grb_descriptor desc, int* ncolors) { GRB_API_ENTER();
  if (!w || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  const Index n = A->nrows;
  if (A->ncols != w->nsize) return GRB_DIMENSION_MISMATCH;
  if (!A->csr.ptr) return GRB_INVALID_OBJECT;
  VecGuard g;
  grb_vector colours;
  GRB_TRY(g.make(&colours, GRB_I32, n));
  GRB_TRY(grb_vector_set_storage(colours, GRB_DENSE));
  GRB_TRY(colour_by_rounds<1>(A, nullptr, 65534, colours, 1, nullptr));
  double mx = 0;
  GRB_TRY(grb_reduce_vector(&mx, GRB_ACCUM_NULL, GRB_MAXIMUM_MONOID, colours, desc));
  if (ncolors) *ncolors = n > 0 ? (int)mx + 1 : 0;
  if (w->dtype == GRB_I32) return grb_vector_dup(w, colours);
  // float output (example/ggc_cusparse.cu colours a Vector<int>; kept for completeness)
  std::vector<int> h(n > 0 ? n : 1);
  grb_index nn = n;
  GRB_TRY(grb_vector_extract_tuples_dense(colours, h.data(), &nn));
  std::vector<float> f(h.begin(), h.end());
  return grb_vector_build_dense(w, f.data(), n);
}

// ---- maximal independent set -------------------------------------------------------------
// misInner (algorithm/mis.hpp:22-111): v = 1 on the members; w (candidate weights), f, m clobbered.
static grb_info mis_inner(grb_vector v, grb_vector w, grb_vector f, grb_vector m, grb_matrix A, grb_descriptor desc,
                          int* rounds_out) {
  GRB_TRY(grb_vector_fill(v, 0.0));
  double succ = 0;
  int rounds = 0;
  const Index n = A->nrows;
  do {
    ++rounds;
    // a failing vxm is not fatal in the reference either (no CHECK, mis.hpp:62-63, :86-87)
    grb_vxm(m, w, GRB_ACCUM_NULL, GRB_MAXIMUM_MULTIPLIES, w, A, desc);
    GRB_TRY(grb_eWiseAdd(f, nullptr, GRB_ACCUM_NULL, GRB_GREATER_PLUS, w, m, desc));
    GRB_TRY(grb_assign(v, f, GRB_ACCUM_NULL, 1.0, desc));
    GRB_TRY(grb_assign(w, f, GRB_ACCUM_NULL, 0.0, desc));
    GRB_TRY(grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, f, desc));
    if (succ == 0) break;
    grb_vxm(m, w, GRB_ACCUM_NULL, GRB_LOGICAL_OR_AND, f, A, desc);
    GRB_TRY(grb_assign(w, m, GRB_ACCUM_NULL, 0.0, desc));
    if (rounds > n + 1) return GRB_PANIC;                  // cannot happen with dense vectors
  } while (succ > 0);
  if (rounds_out) *rounds_out = rounds;
  return GRB_SUCCESS;
}

// The weight vector of mis / gc*: the caller's, or -- as the reference draws it on the host,
// apply(set_random<int>) under GrB_SEQUENTIAL (algorithm/common.hpp:8-20, mis.hpp:128-133) --
// srand(seed) then rand() per vertex in index order.
static grb_info load_weights(grb_vector w, grb_vector weights, int seed, Index n) {
  if (weights) {
    if (weights->dtype != GRB_I32 || weights->nsize != n) return GRB_DIMENSION_MISMATCH;
    GRB_TRY(grb_vector_dup(w, weights));
    return grb_vector_set_storage(w, GRB_DENSE);
  }
  std::vector<int> h(n > 0 ? n : 1);
  srand((unsigned)seed);
  for (Index i = 0; i < n; ++i) h[i] = rand();
  return grb_vector_build_dense(w, h.data(), n);
}

grb_info grb_mis(grb_vector v, grb_matrix A, int seed, grb_vector weights, grb_descriptor desc,
                 grb_algo_result* result) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_I32 || A->dtype != GRB_I32) return GRB_DOMAIN_MISMATCH;
  const Index n = A->nrows;
  VecGuard g;
  grb_vector w, f, m;
  for (grb_vector* p : {&w, &f, &m}) GRB_TRY(g.make(p, GRB_I32, n));
  GRB_TRY(load_weights(w, weights, seed, n));
  ModeGuard pin(desc, GRB_PULLONLY);
  int rounds = 0;
  float ms = 0.f;
  GRB_TRY(grb_timer_start());
  GRB_TRY(mis_inner(v, w, f, m, A, desc, &rounds));
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = rounds; result->tight_ms = ms; result->last_value = 0; }
  return GRB_SUCCESS;
}

// algorithm::gcJP / gcMIS / gcIS (algorithm/gc.hpp:258-421, :152-255, :43-149); algo as --gcalgo:
// 0 Jones-Plassmann, 1 maximal-independent-set per colour, 2 independent-set per colour.
// v = colours from 1 (0 = left uncoloured when max_niter rounds ran out); iterations = `iter`.
grb_info grb_gc(grb_vector v, grb_matrix A, int seed, grb_vector weights, int max_colors, int algo,
                grb_descriptor desc, grb_algo_result* result) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_I32 || A->dtype != GRB_I32) return GRB_DOMAIN_MISMATCH;
  if (algo < 0 || algo > 2 || max_colors < 2) return GRB_INVALID_VALUE;
  const Index n = A->nrows;
  VecGuard g;
  grb_vector f, w, m, nn, temp_w, d = nullptr, ascending = nullptr, min_array = nullptr;
  for (grb_vector* p : {&f, &w, &m, &nn, &temp_w}) GRB_TRY(g.make(p, GRB_I32, n));
  if (algo == 0) {
    for (grb_vector* p : {&d, &ascending, &min_array}) GRB_TRY(g.make(p, GRB_I32, max_colors));
    GRB_TRY(grb_vector_fill_ascending(ascending, max_colors));
  }
  GRB_TRY(grb_vector_fill(v, 0.0));
  GRB_TRY(load_weights(w, weights, seed, n));
  ModeGuard pin(desc, GRB_PULLONLY);
  int iter = 1;
  double succ = 0;
  float ms = 0.f;
  static const bool fused_ok = [] { const char* e = getenv("GRB_GC_FUSED"); return !e || atoi(e) != 0; }();
  if (algo == 0 && fused_ok) {
    // gcJP: the rounds (and their frontiers) are gcIS's; what differs is the colour a round hands out
    GRB_TRY(grb_timer_start());
    GRB_TRY(colour_by_rounds<2>(A, (const int*)w->d_val, desc->max_niter, v, 0, &iter, max_colors));
    GRB_TRY(grb_timer_stop(&ms));
    if (result) { result->iterations = iter; result->tight_ms = ms; result->last_value = 0; }
    return GRB_SUCCESS;
  }
  if (algo == 1 && fused_ok) {
    // gcMIS: colour c is the maximal independent set Luby's rounds find among the vertices left after
    // colours < c, and with fixed weights that set is the greedy one in decreasing weight order -- so the
    // whole algorithm is first-fit colouring in that order: v takes the smallest colour none of its
    // heavier neighbours holds, and can do so as soon as they all have one (the same dataflow, colours
    // chosen as in graphColor).  Equal adjacent weights block each other for ever in both formulations.
    // The reference hands out at most max_niter colours (gc.hpp:244-247).
    GRB_TRY(grb_timer_start());
    GRB_TRY(colour_by_rounds<1>(A, (const int*)w->d_val, 65534, v, 0, nullptr));
    GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
    double mx = 0;
    GRB_TRY(grb_reduce_vector(&mx, GRB_ACCUM_NULL, GRB_MAXIMUM_MONOID, v, desc));
    int ncol = (int)mx;
    if (ncol > desc->max_niter) {
      hipLaunchKernelGGL(clip_colours_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, ctx().stream, (int*)v->d_val, n,
                         desc->max_niter);
      GRB_HIP_TRY(hipGetLastError());
      ncol = desc->max_niter;
    }
    GRB_TRY(grb_timer_stop(&ms));
    if (result) { result->iterations = ncol + 1; result->tight_ms = ms; result->last_value = 0; }
    return GRB_SUCCESS;
  }
  if (algo == 2 && fused_ok) {
    // gcIS as rounds of one kernel each (colour_by_rounds): the same vertices take the same colour
    // in the same round as with the op sequence below -- v wins round r iff w[v] > 0 and w[v]
    // exceeds the weight of every neighbour uncoloured when r began.  The reference runs at most
    // max_niter rounds that colour something (gc.hpp:139-142).
    GRB_TRY(grb_timer_start());
    GRB_TRY(colour_by_rounds<0>(A, (const int*)w->d_val, desc->max_niter, v, 0, &iter));   // <= 65534 rounds
    GRB_TRY(grb_timer_stop(&ms));
    if (result) { result->iterations = iter; result->tight_ms = ms; result->last_value = 0; }
    return GRB_SUCCESS;
  }
  GRB_TRY(grb_timer_start());
  do {
    double colour = iter;
    if (algo == 1) {
      GRB_TRY(grb_vector_dup(temp_w, w));
      GRB_TRY(mis_inner(f, temp_w, nn, m, A, desc, nullptr));
    } else {
      // JP masks the product by the candidates themselves, IS does not (gc.hpp:339-340, :106-107)
      grb_vxm(m, algo == 0 ? w : nullptr, GRB_ACCUM_NULL, GRB_MAXIMUM_MULTIPLIES, w, A, desc);
      GRB_TRY(grb_eWiseAdd(f, nullptr, GRB_ACCUM_NULL, GRB_GREATER_PLUS, w, m, desc));
    }
    GRB_TRY(grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, f, desc));
    if (succ == 0) break;
    if (algo == 0) {
      // smallest colour unused around the whole frontier (gc.hpp:356-385)
      grb_vxm(m, v, GRB_ACCUM_NULL, GRB_LOGICAL_OR_AND, f, A, desc);
      GRB_TRY(grb_eWiseMult(nn, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, m, v, desc));
      GRB_TRY(grb_vector_fill(d, 0.0));
      GRB_TRY(grb_scatter(d, nullptr, nn, (double)max_colors, desc));
      GRB_TRY(grb_eWiseMult(min_array, nullptr, GRB_ACCUM_NULL, GRB_MINIMUM_PLUS, d, ascending, desc));
      GRB_TRY(grb_vector_set_element(min_array, (double)max_colors, 0));
      GRB_TRY(grb_reduce_vector(&colour, GRB_ACCUM_NULL, GRB_MINIMUM_MONOID, min_array, desc));
    }
    GRB_TRY(grb_assign(v, f, GRB_ACCUM_NULL, colour, desc));
    GRB_TRY(grb_assign(w, f, GRB_ACCUM_NULL, 0.0, desc));
    ++iter;
    if (iter > desc->max_niter) break;
  } while (succ > 0);
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = iter; result->tight_ms = ms; result->last_value = succ; }
  return GRB_SUCCESS;
}

// algorithm::lgc (algorithm/lgc.hpp:14-176): approximate personalised PageRank from s by
// residual pushes; p and A are float.  iterations = loop passes, last_value = last frontier size.
grb_info grb_lgc(grb_vector p, grb_matrix A, grb_index s, double alpha, double eps, grb_descriptor desc,
                 grb_algo_result* result) { GRB_API_ENTER();
  if (!p || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (p->dtype != GRB_F32 || A->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
  const Index n = A->nrows;
  if (s < 0 || s >= n) return GRB_INVALID_INDEX;
  VecGuard g;
  grb_vector degrees, r, r2, eps_vector, degrees_eps, f, alpha_vector, alpha_vector2;
  for (grb_vector* q : {&degrees, &r, &r2, &eps_vector, &degrees_eps, &f, &alpha_vector, &alpha_vector2})
    GRB_TRY(g.make(q, GRB_F32, n));
  // pinned to the dense mode, as mis / gc above: the reference's push path has no accum
  // (spmspv.hpp:28-33), so r = r + A r2 loses r there, and its sparse masked eWiseMult with w == u
  // (lgc.hpp:116-117) rewrites the index array other threads are still searching
  ModeGuard pin(desc, GRB_PULLONLY);
  GRB_TRY(grb_reduce_matrix_rows(degrees, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MONOID, A, desc));
  GRB_TRY(grb_vector_fill(p, 0.0));
  GRB_TRY(grb_vector_fill(r2, 0.0));
  const grb_index idx[1] = {s};
  const float one[1] = {1.f};
  if (desc->desc[GRB_MXVMODE] == GRB_PULLONLY) {
    GRB_TRY(grb_vector_fill(r, 0.0));
    GRB_TRY(grb_vector_set_element(r, 1.0, s));
  } else {
    GRB_TRY(grb_vector_build_sparse(r, idx, one, 1));
  }
  GRB_TRY(grb_vector_fill(eps_vector, (double)(float)eps));
  GRB_TRY(grb_eWiseMult(degrees_eps, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, degrees, eps_vector, desc));
  GRB_TRY(grb_vector_build_sparse(f, idx, one, 1));
  GRB_TRY(grb_vector_fill(alpha_vector, (double)(float)alpha));
  GRB_TRY(grb_vector_fill(alpha_vector2, (double)(float)((1. - alpha) / 2.)));
  int iter = 1;
  double succ = 0;
  float ms = 0.f;
  GRB_TRY(grb_timer_start());
  do {
    // none of these is CHECKed in the reference (lgc.hpp:108-135): an op that reports an error
    // leaves its output as it was and the loop goes on
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_eWiseMult(r2, f, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, r, alpha_vector, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_eWiseAdd(p, nullptr, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, p, r2, desc);
    grb_eWiseMult(r, f, GRB_ACCUM_NULL, GRB_PLUS_MULTIPLIES, r, alpha_vector2, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_eWiseMult(r2, f, GRB_ACCUM_NULL, GRB_PLUS_DIVIDES, r, degrees, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    grb_mxv(r, nullptr, GRB_ACCUM_PRESENT, GRB_PLUS_MULTIPLIES, A, r2, desc);
    grb_eWiseMult(f, nullptr, GRB_ACCUM_NULL, GRB_PLUS_GREATER, r, degrees_eps, desc);
    GRB_TRY(grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, f, desc));
    ++iter;
    if (iter > desc->max_niter) break;
  } while (succ > 0);
  GRB_TRY(grb_timer_stop(&ms));
  if (result) { result->iterations = iter - 1; result->tight_ms = ms; result->last_value = succ; }
  return GRB_SUCCESS;
}

// algorithm::diameter (algorithm/diameter.hpp:14-59): BFS eccentricity of each source in
// [s_start, s_end); *diameter_max = the largest, *diameter_ind = the last source attaining it.
grb_info grb_diameter(grb_vector v, grb_matrix A, grb_index s_start, grb_index s_end, grb_descriptor desc,
                      int* diameter_max, int* diameter_ind) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_F32 || A->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
  const Index n = A->nrows;
  VecGuard g;
  grb_vector q1, q2;
  GRB_TRY(g.make(&q1, GRB_F32, n));
  GRB_TRY(g.make(&q2, GRB_F32, n));
  int dmax = 0, dind = -1;
  for (grb_index s = s_start; s < s_end; ++s) {
    GRB_TRY(grb_vector_fill(v, 0.0));
    const grb_index idx[1] = {s};
    const float one[1] = {1.f};
    grb_vector_build_sparse(q1, idx, one, 1);              // unchecked there too (diameter.hpp:36)
    int iter = 1;
    double succ = 0;
    do {
      grb_assign(v, q1, GRB_ACCUM_NULL, (double)iter, desc);
      grb_descriptor_toggle(desc, GRB_MASK);
      grb_vxm(q2, v, GRB_ACCUM_NULL, GRB_LOGICAL_OR_AND, q1, A, desc);
      grb_descriptor_toggle(desc, GRB_MASK);
      grb_vector_swap(q2, q1);
      GRB_TRY(grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, q1, desc));
      ++iter;
      if (iter > n + 2) return GRB_PANIC;
    } while (succ > 0);
    if (iter - 2 > dmax) dmax = iter - 2;
    if (iter - 2 == dmax) dind = (int)s;
  }
  if (diameter_max) *diameter_max = dmax;
  if (diameter_ind) *diameter_ind = dind;
  return GRB_SUCCESS;
}

}  // extern "C"
