// sssp_persist.hip -- algorithm::sssp (graphblas/algorithm/sssp.hpp:15-103) as ONE launch.
//
// The reference runs frontier-filtered Bellman-Ford, six GraphBLAS calls per round
//   f2 = f1 (min.+) A ;  m = f2 < v ;  v = min(v, f2) ;  f2<!m> = inf ;  swap ;  succ = reduce(m)
// and stops when no distance improved.  Rounds are synchronous: round r relaxes the edges
// of the vertices improved in round r-1 with the distances they had at the END of round r-1.
// This kernel keeps exactly those rounds (same iteration count, same distances after every
// round, hence the same result under a max_niter cap) with ONE grid barrier per round:
//
//   D_r    distances at the start of round r, in three rotating buffers.  Round r only reads
//          D_r and only writes D_(r+1) -- with atomicMin on the float's bit pattern
//          (distances are non-negative, so unsigned order is float order).
//   F_r    bitmap of the vertices improved by round r-1 (the reference's f1 pattern), four
//          rotating buffers (F_(r-1), F_r read; F_(r+1) written; one being cleared).
//
// The buffer that becomes D_(r+1) last held D_(r-2); the two differ from D_r exactly on
// F_(r-1) and F_r, so round r first lowers those entries to D_r -- also with atomicMin, which
// makes the order against concurrent relaxations of the same entry irrelevant.  A relaxation
// counts as an improvement iff nd < D_r[v] (stable data: exact), and the first one to set v's
// bit in F_(r+1) (atomicOr return value) counts v.
//
// The fused loop is used for non-negative weights only (checked once per matrix); anything
// else runs the op-by-op driver in algorithms.hip.
#include "persist_common.hpp"

namespace grb {

constexpr int kSsspSmall = 16;
constexpr int kSsspBig = 512;
constexpr int kSsspChunk = 1024;
constexpr int kSsspMedCap = 4096;

struct SsspState {                  // zeroed by the host before every launch
  GridBarrier bar;
  unsigned big_count[2][32];
  unsigned long long acc[3][8][16]; // per (set, XCD group): improved, improved with degree >= kSsspBig
};

struct SsspArgs {
  const Index *optr, *oind;
  const float* oval;
  Index n;
  Index source;
  int max_niter;
  unsigned long long bail_found;    // leave the loop when a round improves more vertices than this
  unsigned long long bail_edges;    // ... and only when they also carry more out-edges than this (a share of nnz)
  float* D[3];                      // D[0] is the result vector; all FLT_MAX, D[1][source] = 0 (host)
  unsigned int* F[4];               // F[1] has the source bit, the rest is zero (host)
  int2* big_list;
  int big_cap;
  SsspState* st;
  unsigned long long* mail;
  int seq;
  float ticks_to_ms;
  grb_algo_iter* rec;               // per-round records (nullable): what sssp.hpp prints under --timing 1
  int rec_cap;
};

struct RoundCounters {
  unsigned long long improved = 0, big = 0, deg = 0;   // deg: out-degree sum of the improved vertices
};

__device__ inline void relax(const SsspArgs& a, const float* Dc, float* Dn, unsigned int* Fn, float du, Index p,
                             RoundCounters& c) {
  const Index v = a.oind[p];
  const float nd = du + a.oval[p];
  if (!(nd < fresh(&Dc[v]))) return;
  atomicMin(reinterpret_cast<unsigned int*>(&Dn[v]), __float_as_uint(nd));
  const unsigned int bit = 1u << (v & 31);
  if (fresh(&Fn[v >> 5]) & bit) return;
  const unsigned int old = atomicOr(&Fn[v >> 5], bit);
  if (old & bit) return;
  ++c.improved;
  const Index dv = a.optr[v + 1] - a.optr[v];
  c.deg += (unsigned long long)dv;
  if (dv >= kSsspBig) ++c.big;
}

// Up to N edges of one vertex with their dependent steps issued stage by stage (loads, then
// distance reads, then atomics ...): a short adjacency list costs one chain of memory
// latencies instead of one per edge.  This is what bounds rounds with tiny frontiers.
template <int N>
__device__ inline void relax_batch(const SsspArgs& a, const float* Dc, float* Dn, unsigned int* Fn, float du, Index p0,
                                   Index e, RoundCounters& c) {
  Index v[N];
  float nd[N], dv[N];
  bool ok[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    ok[j] = p0 + j < e;
    const Index p = ok[j] ? p0 + j : p0;
    v[j] = a.oind[p];
    nd[j] = du + a.oval[p];
  }
#pragma unroll
  for (int j = 0; j < N; ++j) dv[j] = fresh(&Dc[v[j]]);
  unsigned int fw[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    ok[j] = ok[j] && nd[j] < dv[j];
    if (ok[j]) atomicMin(reinterpret_cast<unsigned int*>(&Dn[v[j]]), __float_as_uint(nd[j]));
    fw[j] = ok[j] ? fresh(&Fn[v[j] >> 5]) : 0xffffffffu;
  }
  unsigned int old[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned int bit = 1u << (v[j] & 31);
    old[j] = 0xffffffffu;
    if (!(fw[j] & bit)) old[j] = atomicOr(&Fn[v[j] >> 5], bit);
  }
  Index d0[N], d1[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned int bit = 1u << (v[j] & 31);
    const bool won = !(old[j] & bit);
    d0[j] = a.optr[won ? v[j] : 0];
    d1[j] = a.optr[won ? v[j] + 1 : 0];
    if (!won) d1[j] = d0[j] - 1;          // marks "not an improvement" (degree -1)
  }
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (d1[j] - d0[j] >= 0) {
      ++c.improved;
      c.deg += (unsigned long long)(d1[j] - d0[j]);
      if (d1[j] - d0[j] >= kSsspBig) ++c.big;
    }
}

__global__ __launch_bounds__(kPThreads) void sssp_persistent_kernel(SsspArgs a) {
  __shared__ WaveBits s_bits[kPWaves];
  __shared__ WaveBits4 s_bits4[kPWaves];
  __shared__ unsigned long long s_red[kPWaves][4];
  __shared__ unsigned long long s_tot[4];
  __shared__ Index s_med[kSsspMedCap];
  __shared__ int s_nmed;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int G = gridDim.x;
  const long long gtid = (long long)blockIdx.x * kPThreads + tid;
  const long long gthreads = (long long)G * kPThreads;
  const int nwords = 2 * ((a.n + 63) / 64);
  SsspState* st = a.st;
  unsigned gen = 0;
  const unsigned long long t_start = wall_clock64();

  int iter = 1;
  unsigned long long succ = 1, nbig = (a.optr[a.source + 1] - a.optr[a.source] >= kSsspBig) ? 1 : 0;
  int last_round = 0, bailed = 0;
  for (; iter <= a.max_niter; ++iter) {
    const unsigned long long t_round = wall_clock64();
    const float* Dc = a.D[iter % 3];
    float* Dn = a.D[(iter + 1) % 3];
    const unsigned int* Fp = a.F[(iter + 3) % 4];
    const unsigned int* Fc = a.F[iter % 4];
    unsigned int* Fn = a.F[(iter + 1) % 4];
    unsigned* bcount = &st->big_count[iter & 1][0];
    for (long long i = gtid; i < nwords; i += gthreads) publish(&a.F[(iter + 2) % 4][i], 0u);
    if (gtid == 0) publish(&st->big_count[(iter + 1) & 1][0], 0u);
    if (blockIdx.x == 0 && tid < 32) publish(&st->acc[(iter + 1) % 3][tid >> 2][tid & 3], 0ull);

    // ---- bring the write buffer up to D_r where it is behind (F_(r-1) and F_r), and list
    // the big frontier vertices as 1024-edge entries when the totals announced any
    for (long long base = 0; base < nwords; base += gthreads) {
      const long long i = base + gtid;
      const unsigned int wc = (i < nwords) ? fresh(&Fc[i]) : 0u;
      const unsigned int wp = (i < nwords) ? fresh(&Fp[i]) : 0u;
      wave_for_each_bit(&s_bits[wave], wc | wp, lane, [&](int L, int bit) {
        if (L < 0) return;
        const Index v = (Index)(i - lane + L) * 32 + bit;     // the words of a wave are consecutive
        atomicMin(reinterpret_cast<unsigned int*>(&Dn[v]), __float_as_uint(fresh(&Dc[v])));
      });
      if (nbig > 0) {
        int mine = 0;
        for (unsigned int t = wc; t; t &= t - 1) {
          const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
          const Index d = a.optr[v + 1] - a.optr[v];
          if (d >= kSsspBig) mine += (d + kSsspChunk - 1) / kSsspChunk;
        }
        int incl = mine;
incl = (int)wave_incl_scan_u32((unsigned)incl);
        const int total = (int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        if (total > 0) {
          unsigned b0 = 0;
          if (lane == 0) b0 = atomicAdd(bcount, (unsigned)total);
          b0 = __shfl(b0, 0, kWave);
          int at = (int)b0 + incl - mine;
          for (unsigned int t = wc; t; t &= t - 1) {
            const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
            const Index d = a.optr[v + 1] - a.optr[v];
            if (d >= kSsspBig)
              for (int k = 0; k < (d + kSsspChunk - 1) / kSsspChunk; ++k, ++at)
                if (at < a.big_cap) publish(reinterpret_cast<unsigned long long*>(&a.big_list[at]),
                                            ((unsigned long long)(unsigned)k << 32) | (unsigned)v);
          }
        }
      }
    }
    const unsigned long long t_copy = wall_clock64();
    RoundCounters c;
    if (nbig > 0) {
      if (!grid_sync(&st->bar, gen, false)) return;
      int nent = (int)__hip_atomic_load(bcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nent > a.big_cap) nent = a.big_cap;
      for (int e = blockIdx.x; e < nent; e += G) {
        const unsigned long long eb = fresh(reinterpret_cast<const unsigned long long*>(&a.big_list[e]));
        const int2 ent = make_int2((int)(eb & 0xffffffffull), (int)(eb >> 32));
        const Index p = a.optr[ent.x] + ent.y * kSsspChunk + tid;
        if (p < a.optr[ent.x + 1]) relax(a, Dc, Dn, Fn, fresh(&Dc[ent.x]), p, c);
      }
    }
    // ---- relax the rest of the frontier: words interleaved over the workgroups
    if (tid == 0) s_nmed = 0;
    __syncthreads();
    for (long long base = 0; base < nwords; base += kBitsWords * gthreads) {
      // a wave reads 64 consecutive words per load (two cache lines); consecutive 64-word chunks go to different
      // workgroups, a lane takes four words per pass.  (One word per lane with stride G, the first version, made every
      // lane of a load a cache line of its own: 750 K line requests per round for a road network's 3 MB bitmap.)
      const long long c0 = base / kWave + (long long)wave * G + blockIdx.x;     // word k of a lane: chunk c0 + k * kPWaves * G
      unsigned int w[kBitsWords];
#pragma unroll
      for (int k = 0; k < kBitsWords; ++k) {
        const long long i = (c0 + (long long)k * kPWaves * G) * kWave + lane;
        w[k] = (i < nwords) ? fresh(&Fc[i]) : 0u;
      }
      wave_for_each_bit4(&s_bits4[wave], w, lane, [&](int L, int k, int bit) {
        // every lane walks relax_batch together (a lane without a vertex with an empty range)
        Index v = 0, s = 0, e = 0;
        if (L >= 0) {
          v = (Index)((c0 + (long long)k * kPWaves * G) * kWave + L) * 32 + bit;
          s = a.optr[v];
          e = a.optr[v + 1];
          const Index d = e - s;
          if (d >= kSsspBig) e = s;
          else if (d >= kSsspSmall) {
            const int slot = atomicAdd(&s_nmed, 1);
            if (slot < kSsspMedCap) { s_med[slot] = v; e = s; }
          }
        }
        const float du = e > s ? fresh(&Dc[v]) : 0.f;
        for (Index p = s; p < e; p += 4) relax_batch<4>(a, Dc, Dn, Fn, du, p, e, c);
      });
      __syncthreads();
      const int nm = s_nmed < kSsspMedCap ? s_nmed : kSsspMedCap;
      for (int k = wave; k < nm; k += kPWaves) {
        const Index v = s_med[k];
        const Index e = a.optr[v + 1];
        const float du = fresh(&Dc[v]);
        for (Index p = a.optr[v] + lane; p < e; p += kWave) relax(a, Dc, Dn, Fn, du, p, c);
      }
      __syncthreads();
      if (tid == 0) s_nmed = 0;
      __syncthreads();
    }

    // ---- totals
    const unsigned long long t_relax = wall_clock64();
    const unsigned long long r0 = wave_sum_u64(c.improved), r1 = wave_sum_u64(c.big);
    const unsigned long long r2 = wave_sum_u64(c.deg);
    if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; }
    __syncthreads();
    unsigned long long* acc = &st->acc[iter % 3][0][0];
    if (tid < 3) {
      unsigned long long t = 0;
      for (int w = 0; w < kPWaves; ++w) t += s_red[w][tid];
      if (t) __hip_atomic_fetch_add(&acc[(blockIdx.x & 7) * 16 + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_sync(&st->bar, gen, false)) return;
    if (wave == 0) {
      unsigned long long q = 0;
      if (lane < 32) q = __hip_atomic_load(&acc[(lane >> 2) * 16 + (lane & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __shfl_xor(q, 4, kWave);
      q += __shfl_xor(q, 8, kWave);
      q += __shfl_xor(q, 16, kWave);
      if (lane < 4) s_tot[lane] = q;
    }
    __syncthreads();
    succ = s_tot[0];
    nbig = s_tot[1];
    const unsigned long long next_edges = s_tot[2];
    __syncthreads();
    last_round = iter;
    if (a.rec && gtid == 0 && iter <= a.rec_cap) {
      grb_algo_iter& R = a.rec[iter - 1];
      R.iteration = iter;
      R.direction = GRB_PUSHONLY;
      R.value = (double)succ;
      R.ms = (float)(wall_clock64() - t_round) * a.ticks_to_ms;
      // diagnostic (workgroup 0's clock): tenths of a microsecond spent copying forward / relaxing, packed 16 : 16
      const float to_tenth_us = a.ticks_to_ms * 1e4f;
      const unsigned int t_a = (unsigned int)((float)(t_copy - t_round) * to_tenth_us);
      const unsigned int t_b = (unsigned int)((float)(t_relax - t_copy) * to_tenth_us);
      R.reserved = (int)(((t_a > 0xffffu ? 0xffffu : t_a) << 16) | (t_b > 0xffffu ? 0xffffu : t_b));
    }
    if (succ == 0) break;           // f1.nvals == 0 / reduce(m) == 0, sssp.hpp:88-90
    // hand over to the op-by-op rounds (whose dense product costs about nnz) only when the next frontier is
    // dense in EDGES: a road network's wave front can pass switchpoint * n vertices and still carry a sliver
    // of the graph's edges -- relaxing it here costs microseconds, a pull over the whole matrix does not
    if (succ > a.bail_found && next_edges > a.bail_edges && iter < a.max_niter) { bailed = 1; break; }
  }

  // the distances after the last round live in D_(last+1); the result vector is buffer 0
  {
    const int res = (last_round + 1) % 3;
    if (res != 0) {
      const float* Dr = a.D[res];
      for (long long i = gtid; i < a.n; i += gthreads) a.D[0][i] = fresh(&Dr[i]);
    }
  }
  if (gtid == 0) {
    const unsigned long long tag = (unsigned long long)(unsigned int)a.seq << 32;
    const float ms = (float)(wall_clock64() - t_start) * a.ticks_to_ms;
    const unsigned int vals[4] = {(unsigned int)(iter > a.max_niter ? a.max_niter + 1 : iter), (unsigned int)succ,
                                  __float_as_uint(ms), (unsigned int)bailed};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __hip_atomic_store(&a.mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// hand-over to the op-by-op rounds: the frontier as the dense vector f1 of sssp.hpp
__global__ void sssp_handover_kernel(const unsigned int* __restrict__ F, const float* __restrict__ D, Index n,
                                     float* __restrict__ f1) {
  const Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f1[i] = ((F[i >> 5] >> (i & 31)) & 1u) ? D[i] : FLT_MAX;
}

__global__ void sssp_seed_kernel(float* D1, unsigned int* F1, Index source) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    D1[source] = 0.f;
    F1[source >> 5] = 1u << (source & 31);
  }
}

}  // namespace grb

using namespace grb;

// Runs the fused loop; `v` must be a dense f32 vector of size n.  Returns GRB_NOT_IMPLEMENTED
// when the matrix is not eligible (the caller then runs the op-by-op driver).  The loop relaxes
// with atomics (push); once a round improves more than switchpoint * n vertices the frontier is
// dense enough for the pull product (SpMV) of the op-by-op rounds to be faster, so the kernel
// stops there and, if f1_dense is given, leaves the frontier in it: *handed_over = true and
// *iterations = rounds done.  GrB_PUSHONLY never hands over.
grb_info sssp_persistent_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* iterations,
                             double* succ, float* tight_ms, grb_vector f1_dense, bool* handed_over) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n = A->nrows;
  if (A->dtype != GRB_F32 || v->dtype != GRB_F32 || !A->csr.ptr || A->nrows != A->ncols || A->nvals == 0)
    return GRB_NOT_IMPLEMENTED;
  if (A->nonneg_values < 0) {       // unknown: one reduction over the stored values, cached
    double mn = 0;
    GRB_TRY(k_reduce(GRB_MINIMUM_MONOID, GRB_F32, A->csr.val, A->nvals, &mn));
    A->nonneg_values = mn >= 0.0 ? 1 : 0;
  }
  if (!A->nonneg_values) return GRB_NOT_IMPLEMENTED;

  // road-like graphs: the work-efficient order first (sssp_nearfar.hip) -- same distances, same round count, or it
  // declines and the rounds below run
  {
    const grb_info nf = sssp_nearfar_run(v, A, source, desc, iterations, succ, tight_ms, nullptr);
    if (nf == GRB_SUCCESS) { *handed_over = false; return GRB_SUCCESS; }
    if (nf != GRB_NOT_IMPLEMENTED) return nf;
  }

  const int nwords = 2 * ceil_div(n, 64);
  GRB_TRY(bfs_lanes_fence(ctx().stream));   // a whole-device grid must not meet a BFS lane's narrower one half-way (bfs_persist.hip)
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_persistent_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_PANIC;
  }
  const int G = c.num_cu;
  const int big_cap = (int)(A->nvals / kSsspBig) + 2;
  const size_t st_bytes = (sizeof(SsspState) + 255) & ~(size_t)255;
  const size_t zero_bytes = st_bytes + 16 * (size_t)nwords;
  void *p_zero, *p_c, *p_big;
  GRB_TRY(scratch(7, zero_bytes, &p_zero));
  ctx().bfs_prezero_ptr = nullptr;          // this slot is about to be overwritten
  GRB_TRY(scratch(8, 8 * (size_t)n + 8, &p_c));
  GRB_TRY(scratch(2, sizeof(int2) * (size_t)big_cap, &p_big));
  static float ticks_to_ms = 0.f;
  if (ticks_to_ms == 0.f) {
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ticks_to_ms = khz > 0 ? 1.0f / (float)khz : 1e-5f;
  }
  SsspArgs a;
  a.optr = A->csr.ptr; a.oind = A->csr.ind; a.oval = (const float*)A->csr.val;
  a.n = n;
  a.source = source;
  a.max_niter = desc->max_niter;
  a.bail_found = (f1_dense && desc->desc[GRB_MXVMODE] != GRB_PUSHONLY)
                     ? (unsigned long long)((double)desc->switchpoint * (double)n)
                     : ~0ull;
  a.bail_edges = (unsigned long long)(A->nvals / 8);
  a.D[0] = (float*)v->d_val;
  a.D[1] = (float*)p_c;
  a.D[2] = a.D[1] + n;
  a.F[0] = (unsigned int*)((char*)p_zero + st_bytes);
  for (int k = 1; k < 4; ++k) a.F[k] = a.F[k - 1] + nwords;
  a.big_list = (int2*)p_big;
  a.big_cap = big_cap;
  a.st = (SsspState*)p_zero;
  a.mail = c.d_hgran;
  a.seq = ++c.mail_seq;
  a.ticks_to_ms = ticks_to_ms;
  const int rec_cap = 1 << 16;
  a.rec = nullptr;
  a.rec_cap = 0;
  desc->iter_log.clear();
  if (desc->timing != 0) {
    void* p_rec;
    GRB_TRY(scratch(11, sizeof(grb_algo_iter) * (size_t)rec_cap, &p_rec));
    a.rec = (grb_algo_iter*)p_rec;
    a.rec_cap = rec_cap;
  }

  GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
  GRB_TRY(k_fill(GRB_F32, a.D[0], (double)FLT_MAX, n));
  GRB_TRY(k_fill(GRB_F32, a.D[1], (double)FLT_MAX, 2 * (size_t)n <= 0x7fffffffull ? 2 * n : n));
  if (2 * (size_t)n > 0x7fffffffull) GRB_TRY(k_fill(GRB_F32, a.D[2], (double)FLT_MAX, n));
  hipLaunchKernelGGL(sssp_seed_kernel, dim3(1), dim3(64), 0, s, a.D[1], a.F[1], source);
  GRB_HIP_TRY(hipGetLastError());
  static const bool force_fallback = [] { const char* e = getenv("GRB_SSSP_FORCE_FALLBACK"); return e && atoi(e) != 0; }();
  if (force_fallback) return GRB_PANIC;                        // test hook: as if the grid barrier had given up
  hipLaunchKernelGGL(sssp_persistent_kernel, dim3(G), dim3(kPThreads), 0, s, a);
  GRB_HIP_TRY(hipGetLastError());
  unsigned int gv[4];
  GRB_TRY(wait_granules(a.seq, 4, gv));
  *iterations = (int)gv[0];
  *succ = (double)gv[1];
  float ms;
  memcpy(&ms, &gv[2], 4);
  *tight_ms = ms;
  *handed_over = gv[3] != 0;
  if (a.rec) {
    int rounds = *iterations > desc->max_niter ? desc->max_niter : *iterations;
    if (rounds > rec_cap) rounds = rec_cap;
    if (rounds > 0) {
      desc->iter_log.resize((size_t)rounds);
      GRB_HIP_TRY(hipMemcpyAsync(desc->iter_log.data(), a.rec, sizeof(grb_algo_iter) * (size_t)rounds,
                                 hipMemcpyDeviceToHost, s));
      GRB_HIP_TRY(hipStreamSynchronize(s));
    }
  }
  if (*handed_over) {
    const unsigned int* Fn = a.F[(*iterations + 1) % 4];       // improved by the last round done
    hipLaunchKernelGGL(sssp_handover_kernel, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, s, Fn, (const float*)a.D[0], n,
                       (float*)f1_dense->d_val);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}
