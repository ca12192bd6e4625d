// sssp_persist.hip -- algorithm::sssp (graphblas/algorithm/sssp.hpp:15-103) as ONE launch.
//
// The reference runs frontier-filtered Bellman-Ford, six GraphBLAS calls per round
//   f2 = f1 (min.+) A ;  m = f2 < v ;  v = min(v, f2) ;  f2<!m> = inf ;  swap ;  succ = reduce(m)
// and stops when no distance improved.  Rounds are synchronous: round r relaxes the edges
// of the vertices improved in round r-1 with the distances they had at the END of round r-1.
// This kernel keeps exactly those rounds (same iteration count, same distances after every
// round, hence the same result under a max_niter cap) on three arrays:
//
//   D[n]  best distance at the start of the round (the reference's v)
//   C[n]  candidates of the round being relaxed, FLT_MAX elsewhere (the reference's f2)
//   F     bitmap of the vertices improved by the last round (the reference's f1 pattern),
//         three buffers in rotation as in bfs_persist.hip
//
//   phase A  every frontier word has one owner: D[v] = C[v], C[v] = FLT_MAX for its vertices;
//            vertices of degree >= 512 are cut into 1024-edge entries of a global list
//   barrier
//   phase B  relax: nd = D[u] + w(u,v); if nd < D[v]: old = atomicMin(C[v], nd) on the float's
//            bit pattern (distances are non-negative, so unsigned order is float order); the
//            first improver of v (old == FLT_MAX) sets v's bit in the next frontier and counts it
//   barrier  + totals (number improved = the reference's succ)
//
// The fused loop is used for non-negative weights only (checked once per matrix); anything
// else runs the op-by-op driver in algorithms.hip.
#include "persist_common.hpp"

namespace grb {

constexpr int kSsspSmall = 16;
constexpr int kSsspBig = 512;
constexpr int kSsspChunk = 1024;
constexpr int kSsspMedCap = 4096;
constexpr unsigned int kInfBits = 0x7f7fffffu;    // FLT_MAX

struct SsspState {                  // zeroed by the host before every launch
  GridBarrier bar;
  unsigned big_count[2][32];
  unsigned long long acc[3][8][16]; // per (set, XCD group): improved
};

struct SsspArgs {
  const Index *optr, *oind;
  const float* oval;
  Index n;
  Index source;
  int max_niter;
  float* D;                         // the result vector, FLT_MAX-filled by the host
  float* C;                         // FLT_MAX-filled by the host, C[source] = 0
  unsigned int* F[3];               // F[0] has the source bit, F[1], F[2] are zero (host)
  int2* big_list;
  int big_cap;
  SsspState* st;
  unsigned long long* mail;
  int seq;
  float ticks_to_ms;
};

__device__ inline void relax(const SsspArgs& a, unsigned int* Fn, float du, Index p, unsigned long long& improved) {
  const Index v = a.oind[p];
  const float nd = du + a.oval[p];
  if (!(nd < a.D[v])) return;
  const unsigned int old = atomicMin(reinterpret_cast<unsigned int*>(&a.C[v]), __float_as_uint(nd));
  if (old == kInfBits) {
    atomicOr(&Fn[v >> 5], 1u << (v & 31));
    ++improved;
  }
}

__global__ __launch_bounds__(kPThreads) void sssp_persistent_kernel(SsspArgs a) {
  __shared__ unsigned long long s_red[kPWaves];
  __shared__ unsigned long long s_tot;
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

  int fcur = 0, iter = 1;
  unsigned long long succ = 1;
  for (; iter <= a.max_niter; ++iter) {
    const int fnext = (fcur + 1) % 3, fzero = (fcur + 2) % 3;
    const unsigned int* Fc = a.F[fcur];
    unsigned int* Fn = a.F[fnext];
    unsigned* bcount = &st->big_count[iter & 1][0];
    for (long long i = gtid; i < nwords; i += gthreads) publish(&a.F[fzero][i], 0u);
    if (gtid == 0) publish(&st->big_count[(iter + 1) & 1][0], 0u);
    if (blockIdx.x == 0 && tid < 8) publish(&st->acc[(iter + 1) % 3][tid][0], 0ull);

    // ---- phase A: commit the last round's improvements, list the big frontier vertices
    for (long long base = 0; base < nwords; base += gthreads) {
      const long long i = base + gtid;
      const unsigned int w = (i < nwords) ? Fc[i] : 0u;
      int mine = 0;
      for (unsigned int t = w; t; t &= t - 1) {
        const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
        publish(&a.D[v], a.C[v]);
        publish(&a.C[v], __uint_as_float(kInfBits));
        const Index d = a.optr[v + 1] - a.optr[v];
        if (d >= kSsspBig) mine += (d + kSsspChunk - 1) / kSsspChunk;
      }
      int incl = mine;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int y = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += y;
      }
      const int total = __shfl(incl, kWave - 1, kWave);
      if (total > 0) {
        unsigned b0 = 0;
        if (lane == 0) b0 = atomicAdd(bcount, (unsigned)total);
        b0 = __shfl(b0, 0, kWave);
        int at = (int)b0 + incl - mine;
        for (unsigned int t = w; t; t &= t - 1) {
          const Index v = (Index)i * 32 + (__ffs((int)t) - 1);
          const Index d = a.optr[v + 1] - a.optr[v];
          if (d >= kSsspBig)
            for (int k = 0; k < (d + kSsspChunk - 1) / kSsspChunk; ++k, ++at)
              if (at < a.big_cap) publish(reinterpret_cast<unsigned long long*>(&a.big_list[at]),
                                          ((unsigned long long)(unsigned)k << 32) | (unsigned)v);
        }
      }
    }
    if (!grid_sync(&st->bar, gen)) return;

    // ---- phase B: relax the frontier's edges
    unsigned long long improved = 0;
    {
      int nent = (int)__hip_atomic_load(bcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nent > a.big_cap) nent = a.big_cap;
      for (int e = blockIdx.x; e < nent; e += G) {
        const int2 ent = a.big_list[e];
        const Index p = a.optr[ent.x] + ent.y * kSsspChunk + tid;
        if (p < a.optr[ent.x + 1]) relax(a, Fn, a.D[ent.x], p, improved);
      }
    }
    if (tid == 0) s_nmed = 0;
    __syncthreads();
    for (long long base = 0; base < nwords; base += gthreads) {
      const long long i = (base / G + tid) * G + blockIdx.x;
      unsigned int w = (i < nwords) ? Fc[i] : 0u;
      for (; w; w &= w - 1) {
        const Index v = (Index)i * 32 + (__ffs((int)w) - 1);
        const Index s = a.optr[v], e = a.optr[v + 1];
        const Index d = e - s;
        if (d >= kSsspBig) continue;
        if (d >= kSsspSmall) {
          const int slot = atomicAdd(&s_nmed, 1);
          if (slot < kSsspMedCap) { s_med[slot] = v; continue; }
        }
        const float du = a.D[v];
        for (Index p = s; p < e; ++p) relax(a, Fn, du, p, improved);
      }
      __syncthreads();
      const int nm = s_nmed < kSsspMedCap ? s_nmed : kSsspMedCap;
      for (int k = wave; k < nm; k += kPWaves) {
        const Index v = s_med[k];
        const Index e = a.optr[v + 1];
        const float du = a.D[v];
        for (Index p = a.optr[v] + lane; p < e; p += kWave) relax(a, Fn, du, p, improved);
      }
      __syncthreads();
      if (tid == 0) s_nmed = 0;
      __syncthreads();
    }

    // ---- totals
    improved = wave_reduce(improved, [](unsigned long long x, unsigned long long y) { return x + y; });
    if (lane == 0) s_red[wave] = improved;
    __syncthreads();
    unsigned long long* acc = &st->acc[iter % 3][0][0];
    if (tid == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < kPWaves; ++w) t += s_red[w];
      if (t) __hip_atomic_fetch_add(&acc[(blockIdx.x & 7) * 16], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_sync(&st->bar, gen)) return;
    if (wave == 0) {
      unsigned long long q = 0;
      if (lane < 8) q = __hip_atomic_load(&acc[lane * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __shfl_xor(q, 1, kWave);
      q += __shfl_xor(q, 2, kWave);
      q += __shfl_xor(q, 4, kWave);
      if (lane == 0) s_tot = q;
    }
    __syncthreads();
    succ = s_tot;
    __syncthreads();
    fcur = fnext;
    if (succ == 0) break;           // f1.nvals == 0 / reduce(m) == 0, sssp.hpp:88-90
  }

  // the last round's improvements are part of v (v = min(v, f2) happens inside the round);
  // only pending when the iteration cap ended the loop
  if (succ != 0) {
    const unsigned int* Fc = a.F[fcur];
    for (long long i = gtid; i < nwords; i += gthreads)
      for (unsigned int w = Fc[i]; w; w &= w - 1) {
        const Index v = (Index)i * 32 + (__ffs((int)w) - 1);
        a.D[v] = a.C[v];
      }
  }
  if (gtid == 0) {
    const unsigned long long tag = (unsigned long long)(unsigned int)a.seq << 32;
    const float ms = (float)(wall_clock64() - t_start) * a.ticks_to_ms;
    const unsigned int vals[4] = {(unsigned int)(iter > a.max_niter ? a.max_niter + 1 : iter), (unsigned int)succ,
                                  __float_as_uint(ms), 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __hip_atomic_store(&a.mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void sssp_seed_kernel(float* C, unsigned int* F0, Index source) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    C[source] = 0.f;
    F0[source >> 5] = 1u << (source & 31);
  }
}

}  // namespace grb

using namespace grb;

// Runs the fused loop; `v` must be a dense f32 vector of size n.  Returns GRB_NOT_IMPLEMENTED
// when the matrix is not eligible (the caller then runs the op-by-op driver).
grb_info sssp_persistent_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* iterations,
                             double* succ, float* tight_ms) {
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

  const int nwords = 2 * ceil_div(n, 64);
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_persistent_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_PANIC;
  }
  const int G = c.num_cu;
  const int big_cap = (int)(A->nvals / kSsspBig) + 2;
  const size_t st_bytes = (sizeof(SsspState) + 255) & ~(size_t)255;
  const size_t zero_bytes = st_bytes + 12 * (size_t)nwords;
  void *p_zero, *p_c, *p_big;
  GRB_TRY(scratch(7, zero_bytes, &p_zero));
  GRB_TRY(scratch(8, 4 * (size_t)n + 4, &p_c));
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
  a.D = (float*)v->d_val;
  a.C = (float*)p_c;
  a.F[0] = (unsigned int*)((char*)p_zero + st_bytes); a.F[1] = a.F[0] + nwords; a.F[2] = a.F[1] + nwords;
  a.big_list = (int2*)p_big;
  a.big_cap = big_cap;
  a.st = (SsspState*)p_zero;
  a.mail = c.d_hgran;
  a.seq = ++c.mail_seq;
  a.ticks_to_ms = ticks_to_ms;

  GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
  GRB_TRY(k_fill(GRB_F32, a.D, (double)FLT_MAX, n));
  GRB_TRY(k_fill(GRB_F32, a.C, (double)FLT_MAX, n));
  hipLaunchKernelGGL(sssp_seed_kernel, dim3(1), dim3(64), 0, s, a.C, a.F[0], source);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(sssp_persistent_kernel, dim3(G), dim3(kPThreads), 0, s, a);
  GRB_HIP_TRY(hipGetLastError());
  unsigned int gv[4];
  GRB_TRY(wait_granules(a.seq, 4, gv));
  *iterations = (int)gv[0];
  *succ = (double)gv[1];
  float ms;
  memcpy(&ms, &gv[2], 4);
  *tight_ms = ms;
  return GRB_SUCCESS;
}
