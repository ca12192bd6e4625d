// sssp_nearfar.hip -- algorithm::sssp on long-diameter, low-degree graphs (road networks) without the
// Bellman-Ford rounds' redundant work, and still the reference's answer.
//
// The reference relaxes, every round, the out-edges of every vertex whose distance improved in the previous round
// (graphblas/algorithm/sssp.hpp:53-90).  On a road network with weights 1..64 that frontier is a band of 0.4-1.0 M
// vertices for almost all of 8 880 rounds: a vertex is improved 175 times on average before its distance is final
// (4.2 G improvements for 24 M vertices; sssp_persist.hip runs those rounds exactly, 77 us each).  What the caller
// can observe of that loop is (a) the distances after min(max_niter, convergence) rounds and (b) the number of
// rounds.  Both follow from the fixed point alone when the loop converges within max_niter:
//   * d[v]: the fixed point of d[v] = min_u fl(d[u] + w(u, v)) is unique for non-negative weights and is what the
//     synchronous rounds converge to -- any label-correcting order reaches the same floats;
//   * rounds: the synchronous loop gives v its final distance in round h(v) = the fewest edges among the paths
//     that attain it, and stops one round after the last improvement: iterations = max_v h(v) + 1.  h is the
//     second component of the fixed point of the LEXICOGRAPHIC relaxation on (distance, hops).
// So this kernel relaxes 64-bit keys (float bits of the distance << 32 | hops) with one atomicMin each, in a
// near / far order: only the "dirty" vertices (key changed since their edges were last relaxed) whose distance is
// below a moving threshold are expanded; when none is left the threshold jumps to the smallest dirty distance
// + delta (delta = 32 x the mean edge weight: measured flat between 16 x and 64 x on the road-like stand-in -- the
// number of passes cannot fall below the hop depth of the shortest-path tree, a wider band only adds rework).  If the fixed point
// says the reference would have been cut off by max_niter (iterations > max_niter), or per-round records were asked
// for (--timing), the caller runs the round-exact kernel instead: nothing observable changes.
//
// One launch, one grid barrier per pass (persist_common.hpp).  A pass: every workgroup walks its words of the dirty
// bitmap, one dirty vertex per lane per step (wave_for_each_bit4); near ones clear their bit FIRST, then read their
// key, then relax -- whoever lowers the key afterwards sets the bit again.
#include "persist_common.hpp"

namespace grb {

typedef unsigned long long u64;

struct NfState {                    // zeroed by the host before every launch
  GridBarrier bar;
  u64 acc[3][8][16];                // per (set, XCD group): expanded, left far, made dirty, out-edges of the expanded
  unsigned int minfar[3][32];       // smallest distance (float bits) among the dirty vertices left far; host sets all ones
  unsigned int maxhops[32];
  unsigned int maxdist[32];         // float bits of the largest finite distance
};

struct NfArgs {
  const Index *optr, *oind;
  const float* oval;
  Index n;
  Index source;
  float delta;
  u64* K;                           // keys, all (FLT_MAX, ~0) but the source's (0, 0)
  unsigned int* dirty;              // the source's bit
  float* D;                         // result
  NfState* st;
  u64* mail;
  int seq;
  float ticks_to_ms;
  int max_passes;
  int inner;                        // walks of a wave over its words per pass
};

constexpr int kNfWide = 64;         // out-degree from which the whole wave expands a vertex

__device__ inline u64 nf_key(float d, unsigned int hops) { return ((u64)__float_as_uint(d) << 32) | hops; }

// one out-edge: the target's key lowered, the target marked dirty by whoever lowered it
__device__ inline void nf_relax(const NfArgs& a, float du, unsigned int hu, Index p, unsigned int& made) {
  const Index v = a.oind[p];
  const u64 nk = nf_key(du + a.oval[p], hu + 1u);
  if (!(nk < fresh(&a.K[v]))) return;
  const u64 old = atomicMin(&a.K[v], nk);
  if (!(nk < old)) return;
  const unsigned int bit = 1u << (v & 31);
  if (fresh(&a.dirty[v >> 5]) & bit) return;
  if (!(atomicOr(&a.dirty[v >> 5], bit) & bit)) ++made;
}

// up to N edges of one vertex, their dependent steps issued stage by stage
template <int N>
__device__ inline void nf_relax_batch(const NfArgs& a, float du, unsigned int hu, Index p0, Index e, unsigned int& made) {
  Index v[N];
  u64 nk[N];
  bool ok[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    ok[j] = p0 + j < e;
    const Index p = ok[j] ? p0 + j : p0;
    v[j] = a.oind[p];
    nk[j] = nf_key(du + a.oval[p], hu + 1u);
  }
  // no peek at the target's key first: a pass of this kernel is a chain of dependent memory steps, not a
  // throughput problem (a few thousand vertices per pass), and the atomicMin answers the question itself
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (ok[j]) ok[j] = nk[j] < atomicMin(&a.K[v[j]], nk[j]);
  unsigned int fw[N];
#pragma unroll
  for (int j = 0; j < N; ++j) fw[j] = ok[j] ? fresh(&a.dirty[v[j] >> 5]) : 0xffffffffu;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned int bit = 1u << (v[j] & 31);
    if (!(fw[j] & bit) && !(atomicOr(&a.dirty[v[j] >> 5], bit) & bit)) ++made;
  }
}

__global__ __launch_bounds__(kPThreads) void sssp_nearfar_kernel(NfArgs a) {
  __shared__ WaveBits4 s_bits4[kPWaves];
  __shared__ u64 s_red[kPWaves][4];
  __shared__ unsigned int s_min[kPWaves];
  __shared__ u64 s_tot[4];
  __shared__ unsigned int s_minfar;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int G = gridDim.x;
  const long long gtid = (long long)blockIdx.x * kPThreads + tid;
  const long long gthreads = (long long)G * kPThreads;
  const int nwords = 2 * ((a.n + 63) / 64);
  NfState* st = a.st;
  unsigned gen = 0;
  const u64 t_start = wall_clock64();
  float T = a.delta;                // the source (distance 0) is near
  int pass = 1;
  int converged = 0;
  u64 cum_expanded = 0, cum_made = 0, cum_relaxed = 0;   // over all passes: what the kernel's algorithmic bytes are priced on
  for (; pass <= a.max_passes; ++pass) {
    // the next pass's totals (nobody touches them during this one)
    if (blockIdx.x == 0 && tid < 32) publish(&st->acc[(pass + 1) % 3][tid / 4][tid % 4], 0ull);
    if (blockIdx.x == 0 && tid == 32) publish(&st->minfar[(pass + 1) % 3][0], 0xffffffffu);
    unsigned int expanded = 0, far = 0, made = 0, mymin = 0xffffffffu;
    u64 relaxed = 0;                  // out-edges of the vertices expanded (accounting only)
    // A wave walks its words up to `inner` (2) times per pass, as long as the last walk expanded something: what
    // it (or anybody) made dirty and near in its own words meanwhile moves on without waiting for the barrier.
    for (int walk = 0; walk < a.inner; ++walk) {
    const unsigned int expanded_before = expanded;
    for (long long base = 0; base < nwords; base += kBitsWords * gthreads) {
      const long long q0 = base / G + tid;               // workgroup b owns the words = b (mod G)
      unsigned int w[kBitsWords];
#pragma unroll
      for (int k = 0; k < kBitsWords; ++k) {
        const long long i = (q0 + (long long)k * kPThreads) * G + blockIdx.x;
        w[k] = (i < nwords) ? fresh(&a.dirty[i]) : 0u;
      }
      wave_for_each_bit4(&s_bits4[wave], w, lane, [&](int L, int k, int bit) {
        Index v = 0, s = 0, e = 0;
        float du = 0.f;
        unsigned int hu = 0;
        if (L >= 0) {
          const long long word = (q0 + (L - lane) + (long long)k * kPThreads) * G + blockIdx.x;
          v = (Index)word * 32 + bit;
          const unsigned int dbits = (unsigned int)(fresh(&a.K[v]) >> 32);
          if (__uint_as_float(dbits) < T) {
            // clear first -- an acquire, so that the read of the key below cannot be served before it: a later
            // improvement marks the vertex again, and this read cannot miss an earlier one
            (void)__hip_atomic_fetch_and(&a.dirty[word], ~(1u << bit), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            const u64 key = fresh(&a.K[v]);
            du = __uint_as_float((unsigned int)(key >> 32));
            hu = (unsigned int)key;
            s = a.optr[v];
            e = a.optr[v + 1];
            ++expanded;
            relaxed += (u64)(e - s);
          } else {
            ++far;
            mymin = dbits < mymin ? dbits : mymin;
          }
        }
        const bool wide = e - s >= kNfWide;
        if (!wide)
          for (Index p = s; p < e; p += 4) nf_relax_batch<4>(a, du, hu, p, e, made);
        for (u64 todo = __ballot(wide); todo; todo &= todo - 1) {      // a hub: the whole wave takes its edges
          const int src = __ffsll((long long)todo) - 1;
          const Index s2 = __shfl(s, src, kWave), e2 = __shfl(e, src, kWave);
          const float d2 = __shfl(du, src, kWave);
          const unsigned int h2 = __shfl(hu, src, kWave);
          for (Index p = s2 + lane; p < e2; p += kWave) nf_relax(a, d2, h2, p, made);
        }
      });
    }
    if (__ballot(expanded != expanded_before) == 0ull) break;
    }
    // ---- totals
    auto add = [](u64 x, u64 y) { return x + y; };
    const u64 r0 = wave_reduce((u64)expanded, add), r1 = wave_reduce((u64)far, add), r2 = wave_reduce((u64)made, add);
    const u64 r3 = wave_reduce(relaxed, add);
    const unsigned int rm = wave_reduce(mymin, [](unsigned int x, unsigned int y) { return x < y ? x : y; });
    if (lane == 0) { s_red[wave][0] = r0; s_red[wave][1] = r1; s_red[wave][2] = r2; s_red[wave][3] = r3; s_min[wave] = rm; }
    __syncthreads();
    u64* acc = &st->acc[pass % 3][0][0];
    if (tid < 4) {
      u64 t = 0;
      for (int w2 = 0; w2 < kPWaves; ++w2) t += s_red[w2][tid];
      if (t) __hip_atomic_fetch_add(&acc[(blockIdx.x & 7) * 16 + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 4) {
      unsigned int m = 0xffffffffu;
      for (int w2 = 0; w2 < kPWaves; ++w2) m = s_min[w2] < m ? s_min[w2] : m;
      if (m != 0xffffffffu) atomicMin(&st->minfar[pass % 3][0], m);
    }
    if (!grid_sync(&st->bar, gen, false)) return;
    if (wave == 0) {
      u64 q = 0;
      if (lane < 32) q = __hip_atomic_load(&acc[(lane >> 2) * 16 + (lane & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __shfl_xor(q, 4, kWave);
      q += __shfl_xor(q, 8, kWave);
      q += __shfl_xor(q, 16, kWave);
      if (lane < 4) s_tot[lane] = q;
      if (lane == 0) s_minfar = fresh(&st->minfar[pass % 3][0]);
    }
    __syncthreads();
    const u64 t_expanded = s_tot[0], t_far = s_tot[1], t_made = s_tot[2];
    cum_expanded += t_expanded; cum_made += t_made; cum_relaxed += s_tot[3];
    const unsigned int t_minfar = s_minfar;
    __syncthreads();
    // vertices made dirty in this pass may be near or far: only "nothing expanded, nothing dirty" is the end
    if (t_expanded == 0 && t_far == 0 && t_made == 0) { converged = 1; break; }
    if (t_expanded == 0 && t_made == 0 && t_minfar != 0xffffffffu) {
      // at least the next float above the smallest dirty distance: delta can vanish in the sum at large distances
      const float up = __uint_as_float(t_minfar + 1u);
      T = __uint_as_float(t_minfar) + a.delta;
      T = T > up ? T : up;
    }
  }

  // ---- distances out, and the largest hop count of a reached vertex (the reference's round count - 1)
  unsigned int mh = 0, md = 0;
  for (long long i = gtid; i < a.n; i += gthreads) {
    const u64 key = fresh(&a.K[i]);
    a.D[i] = __uint_as_float((unsigned int)(key >> 32));
    const unsigned int h = (unsigned int)key;
    if (h != 0xffffffffu) {
      mh = h > mh ? h : mh;
      md = (unsigned int)(key >> 32) > md ? (unsigned int)(key >> 32) : md;
    }
  }
  auto umax = [](unsigned int x, unsigned int y) { return x > y ? x : y; };
  mh = wave_reduce(mh, umax);
  md = wave_reduce(md, umax);
  if (lane == 0 && mh) atomicMax(&st->maxhops[0], mh);
  if (lane == 0 && md) atomicMax(&st->maxdist[0], md);
  if (!grid_sync(&st->bar, gen, false)) return;
  if (gtid == 0) {
    const u64 tag = (u64)(unsigned int)a.seq << 32;
    const float ms = (float)(wall_clock64() - t_start) * a.ticks_to_ms;
    const unsigned int vals[8] = {fresh(&st->maxhops[0]), (unsigned int)pass, __float_as_uint(ms), (unsigned int)converged,
                                  fresh(&st->maxdist[0]), (unsigned int)(cum_expanded > 0xffffffffull ? 0xffffffffull : cum_expanded),
                                  (unsigned int)(cum_relaxed >> 4 > 0xffffffffull ? 0xffffffffull : cum_relaxed >> 4),
                                  (unsigned int)(cum_made > 0xffffffffull ? 0xffffffffull : cum_made)};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      __hip_atomic_store(&a.mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// stored values that are not integers in [0, 2^20]: with none, every path sum below 2^24 is exact
__global__ void nf_count_inexact_kernel(const float* __restrict__ val, Index nvals, unsigned int* __restrict__ out) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  unsigned int bad = 0;
  for (Index p = (Index)blockIdx.x * blockDim.x + threadIdx.x; p < nvals; p += stride) {
    const float x = val[p];
    if (!(x >= 0.f && x <= 1048576.f && x == floorf(x))) ++bad;
  }
  bad = wave_reduce(bad, [](unsigned int x, unsigned int y) { return x + y; });
  if (lane_id() == 0 && bad) atomicAdd(out, bad);
}

__global__ void nf_init_kernel(u64* K, Index n, Index source) {
  const Index stride = (Index)gridDim.x * blockDim.x;
  for (Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    K[i] = i == source ? 0ull : (((u64)__float_as_uint(FLT_MAX) << 32) | 0xffffffffull);
}
__global__ void nf_seed_kernel(unsigned int* dirty, unsigned int* minfar, Index source) {
  if (threadIdx.x == 0) dirty[source >> 5] = 1u << (source & 31);
  if (threadIdx.x < 3) minfar[threadIdx.x * 32] = 0xffffffffu;
}

}  // namespace grb

using namespace grb;

// Whether grb_sssp should try this kernel first.  GRB_SSSP_NEARFAR=0 never; by default for graphs that look like
// road networks (fewer than 8 stored entries per row) whose weights are small integers -- every path sum is then
// exact and the round count derived from the hop counts IS the synchronous loop's; =1 also for other weights
// (distances still identical; the reported round count can differ from the reference's where float rounding lets a
// not-yet-final distance produce a final one).  Never when per-round records are requested (--timing).
static int g_nearfar_mode = -2;     // -2: not read yet; -1 auto, 0 off, 1 on
int grb::sssp_nearfar_setting(int set, bool apply) {
  if (g_nearfar_mode == -2) { const char* e = getenv("GRB_SSSP_NEARFAR"); g_nearfar_mode = e ? (atoi(e) > 0 ? 1 : atoi(e) == 0 ? 0 : -1) : -1; }
  if (apply) g_nearfar_mode = set > 0 ? 1 : set == 0 ? 0 : -1;
  return g_nearfar_mode;
}
static int nearfar_env() { return sssp_nearfar_setting(0, false); }
static int g_barrier_failures = 0;   // consecutive launches whose grid barrier gave up
static long long g_last_work[3] = {0, 0, 0};
static int g_last_order = 0;         // what the last grb_sssp of this process ran: 0 synchronous rounds, else near / far (its passes)
void grb::sssp_last_work(long long* out3) { for (int i = 0; i < 3; ++i) out3[i] = g_last_work[i]; }
int grb::sssp_last_order(int set) {
  if (set >= 0) g_last_order = set;
  return g_last_order;
}
static grb_info nearfar_wanted(grb_matrix A, grb_descriptor desc, bool* yes) {
  *yes = false;
  const int env = nearfar_env();
  if (env == 0 || desc->timing != 0) return GRB_SUCCESS;
  if (env > 0) { *yes = true; return GRB_SUCCESS; }
  if (A->nvals >= 8ll * (long long)A->nrows) return GRB_SUCCESS;
  if (A->small_int_values < 0) {
    void* p;
    GRB_TRY(scratch(9, 64, &p));
    GRB_HIP_TRY(hipMemsetAsync(p, 0, 4, ctx().stream));
    hipLaunchKernelGGL(nf_count_inexact_kernel, dim3(stream_grid(A->nvals, kBlock)), dim3(kBlock), 0, ctx().stream,
                       (const float*)A->csr.val, A->nvals, (unsigned int*)p);
    GRB_HIP_TRY(hipGetLastError());
    unsigned int bad = 1;
    GRB_HIP_TRY(hipMemcpyAsync(&bad, p, 4, hipMemcpyDeviceToHost, ctx().stream));
    GRB_HIP_TRY(hipStreamSynchronize(ctx().stream));
    A->small_int_values = bad == 0 ? 1 : 0;
  }
  *yes = A->small_int_values == 1;
  return GRB_SUCCESS;
}

// GRB_SUCCESS: v holds the distances and *iterations the reference's round count.  GRB_NOT_IMPLEMENTED: not
// eligible, or the reference would have stopped at max_niter before converging -- run the round-exact kernel.
grb_info grb::sssp_nearfar_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* iterations,
                               double* succ, float* tight_ms, int* passes) {
  bool wanted = false;
  g_last_order = 0;
  if (g_barrier_failures >= 3) return GRB_NOT_IMPLEMENTED;   // the device is evidently shared: stop paying for the attempts
  GRB_TRY(nearfar_wanted(A, desc, &wanted));
  if (!wanted) return GRB_NOT_IMPLEMENTED;
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n = A->nrows;
  if (A->mean_value < 0.0) {
    double sum = 0;
    GRB_TRY(k_reduce(GRB_PLUS_MONOID, GRB_F32, A->csr.val, A->nvals, &sum));
    A->mean_value = A->nvals > 0 ? sum / (double)A->nvals : 0.0;
  }
  const int nwords = 2 * ceil_div(n, 64);
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_nearfar_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_NOT_IMPLEMENTED;
  }
  const size_t st_bytes = (sizeof(NfState) + 255) & ~(size_t)255;
  const size_t zero_bytes = st_bytes + 4 * (size_t)nwords;
  void *p_zero, *p_k;
  GRB_TRY(scratch(7, zero_bytes, &p_zero));
  c.bfs_prezero_ptr = nullptr;              // this slot is about to be overwritten
  GRB_TRY(scratch(8, 8 * (size_t)n + 8, &p_k));
  static float ticks_to_ms = 0.f;
  if (ticks_to_ms == 0.f) {
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ticks_to_ms = khz > 0 ? 1.0f / (float)khz : 1e-5f;
  }
  NfArgs a;
  a.optr = A->csr.ptr; a.oind = A->csr.ind; a.oval = (const float*)A->csr.val;
  a.n = n;
  a.source = source;
  static const double delta_factor = getenv("GRB_SSSP_DELTA") ? atof(getenv("GRB_SSSP_DELTA")) : 32.0;
  a.delta = (float)(delta_factor * A->mean_value);
  if (!(a.delta > 0.f)) a.delta = 1.f;       // all-zero weights: any positive width
  a.K = (u64*)p_k;
  a.st = (NfState*)p_zero;
  a.dirty = (unsigned int*)((char*)p_zero + st_bytes);
  a.D = (float*)v->d_val;
  a.mail = c.d_hgran;
  a.seq = ++c.mail_seq;
  a.ticks_to_ms = ticks_to_ms;
  // a pass either expands a vertex or raises the threshold past one: far fewer than n of each are ever needed;
  // the cap only turns a logic error into "not converged" (the rounds then run) instead of an endless launch
  const long long pass_cap = 8ll * (long long)n + 1024;
  a.max_passes = pass_cap > 0x7fffff00ll ? 0x7fffff00 : (int)pass_cap;
  static const int inner = getenv("GRB_SSSP_INNER") ? atoi(getenv("GRB_SSSP_INNER")) : 2;
  a.inner = inner < 1 ? 1 : inner;
  GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
  hipLaunchKernelGGL(nf_init_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a.K, n, (Index)source);
  hipLaunchKernelGGL(nf_seed_kernel, dim3(1), dim3(64), 0, s, a.dirty, &a.st->minfar[0][0], (Index)source);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(sssp_nearfar_kernel, dim3(c.num_cu), dim3(kPThreads), 0, s, a);
  GRB_HIP_TRY(hipGetLastError());
  unsigned int gv[8];
  if (wait_granules(a.seq, 8, gv) != GRB_SUCCESS) {       // the barrier gave up: round-exact path
    ++g_barrier_failures;
    return GRB_NOT_IMPLEMENTED;
  }
  g_barrier_failures = 0;
  if (!gv[3]) return GRB_NOT_IMPLEMENTED;
  float maxdist;
  memcpy(&maxdist, &gv[4], 4);
  if (nearfar_env() <= 0 && !(maxdist < 16777216.f)) return GRB_NOT_IMPLEMENTED;   // sums no longer exact: rounds could differ
  const long long rounds = (long long)gv[0] + 1;        // the synchronous loop's count: last improving round + 1
  if (rounds > (long long)desc->max_niter) return GRB_NOT_IMPLEMENTED;       // it would have been cut off
  *iterations = (int)rounds;
  *succ = 0.0;
  memcpy(tight_ms, &gv[2], 4);
  if (passes) *passes = (int)gv[1];
  desc->iter_log.clear();
  g_last_order = (int)gv[1] > 0 ? (int)gv[1] : 1;       // the passes it took
  g_last_work[0] = (long long)gv[5];                    // vertices expanded, out-edges relaxed, vertices made dirty: all passes
  g_last_work[1] = (long long)gv[6] << 4;
  g_last_work[2] = (long long)gv[7];
  return GRB_SUCCESS;
}
