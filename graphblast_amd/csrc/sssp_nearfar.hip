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
      // a wave reads 64 consecutive words per load (two cache lines); consecutive 64-word chunks go to different
      // workgroups.  (One word per lane with stride G, the first version, made every lane of a load a line of its own.)
      const long long c0 = base / kWave + (long long)wave * G + blockIdx.x;     // word k of a lane: chunk c0 + k * kPWaves * G
      unsigned int w[kBitsWords];
#pragma unroll
      for (int k = 0; k < kBitsWords; ++k) {
        const long long i = (c0 + (long long)k * kPWaves * G) * kWave + lane;
        w[k] = (i < nwords) ? fresh(&a.dirty[i]) : 0u;
      }
      wave_for_each_bit4(&s_bits4[wave], w, lane, [&](int L, int k, int bit) {
        Index v = 0, s = 0, e = 0;
        float du = 0.f;
        unsigned int hu = 0;
        if (L >= 0) {
          const long long word = (c0 + (long long)k * kPWaves * G) * kWave + L;
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
    // (DPP sums, common.hpp: a pass of a road network is a few microseconds, and five shuffle reductions were one of them)
    const u64 r0 = wave_sum_u64((u64)expanded), r1 = wave_sum_u64((u64)far), r2 = wave_sum_u64((u64)made);
    const u64 r3 = wave_sum_u64(relaxed);
    const unsigned int rm = wave_min_u32(mymin);
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
  mh = wave_max_u32(mh);
  md = wave_max_u32(md);
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

// ---- the same order from queues ------------------------------------------------------------------------------------
// A pass of the kernel above is a chain of dependent memory steps, not a throughput problem: a road network's near
// set is a few thousand vertices, and finding them costs a walk over the whole dirty bitmap (3 MB for 24 M vertices)
// before the first key is read -- 30 us per pass, 6 947 passes.  Here the near set IS a list, and every step that
// list would add to the chain is designed away:
//   * an entry is (distance bits, vertex), written once per successful atomicMin.  No "queued" bit per vertex and
//     no clearing of it: a vertex lowered twice has two entries, and the one whose distance is no longer the key's
//     is dropped when it is taken out (one compare against the key that is read anyway);
//   * an entry at or above the threshold goes to the far pile instead; when a pass leaves the next queue empty the
//     threshold moves to the smallest far distance + delta and one pass deals the pile out again (live near ones to
//     the queue, live far ones to the other pile);
//   * joiners are staged in LDS and each workgroup asks for its slots in the next queue ONCE per pass -- one
//     global atomic per workgroup instead of one per wave step on the same word (those serialise: measured 10.3 us
//     per pass with them, whatever the number of workgroups);
//   * the count of the next queue is the low half of the very word the barrier's last arrivals add to: the poll
//     that sees everybody arrive has read it.
// A wave takes 16 queue entries per step and gives each of them 4 lanes, one per out-edge (a road network's rows
// hold 2-4 entries): every stage -- entry, key + row bounds, edges, atomicMin -- is one wave instruction for all
// the edges of 16 vertices.  Chain per pass: entry, key / bounds, edges, atomicMin, [LDS], slots, stores, arrive, poll.
constexpr int kNfqStageNear = 2048; // entries a workgroup stages per pass before it appends one by one
constexpr int kNfqStageFar = 1024;

// (vertex, float bits of the distance it was given, its row's bounds): everything the pass that takes the entry
// out needs to issue the edge loads at once -- the key is read beside them, only to see whether the entry is
// still the vertex's latest and for the hop count
struct NfqEntry {
  u64 vd;                           // distance bits << 32 | vertex
  u64 se;                           // row end << 32 | row start
};

struct NfqState {                   // zeroed by the host before every launch
  unsigned int xcd_count[3][8][32]; // barrier b uses set b % 3 (one 128 B line per counter)
  u64 top[3][16];                   // high half: arrivals; low half: entries of the queue the pass filled
  unsigned int fcount[2][32];       // entries of the far piles
  unsigned int minfar[2][32];       // smallest distance (float bits) that went to the pile; host sets all ones
  unsigned int stop[32];            // 1: a barrier gave up; 2: a list is full (the caller takes the bitmap form)
  unsigned int maxhops[32];
  unsigned int maxdist[32];
  u64 work[4][16];                  // all passes: entries expanded, out-edges relaxed, entries written
  u64 tally[2][16];                 // as a traversal: vertices reached, their out-degrees
};

struct NfqArgs {
  const Index *optr, *oind;
  const float* oval;
  Index n;
  float delta;
  u64* K;
  NfqEntry* qn[2];                  // near queues, near_cap entries each
  NfqEntry* qf[2];                  // far piles, far_cap entries each
  unsigned int near_cap, far_cap;
  float* D;
  NfqState* st;
  u64* mail;
  int seq;
  float ticks_to_ms;
  int max_passes;
  int unit;                         // every weight is 1 (the value array is not read): the passes are BFS levels
  int as_bfs;                       // the result is algorithm::bfs's: depth labels (source 1, unreached 0), reached / edge totals
  int inner;                        // sub-steps a workgroup runs on the near entries it staged itself before the pass's grid barrier
};

constexpr int kNfqSlots = 16;       // queue entries a wave takes per step
constexpr int kNfqLanes = kWave / kNfqSlots;

__device__ inline void nfq_put(NfqEntry* p, const NfqEntry& x) { publish(&p->vd, x.vd); publish(&p->se, x.se); }
__device__ inline NfqEntry nfq_get(const NfqEntry* p) { NfqEntry x; x.vd = fresh(&p->vd); x.se = fresh(&p->se); return x; }

// Barrier number b of the launch; returns false when it was abandoned.  *lo = the low half of its word.
// (One flat counter for all 256 workgroups would be one step less for the last arrivals, and measured 15 % slower
// per pass: 256 pollers and the arrivals share the one word.  Letting the idle workgroups poll 16x less often
// changed neither form.)
__device__ inline bool nfq_sync(NfqState* st, unsigned int b, unsigned int* lo) {
  __shared__ u64 s_word;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int G = gridDim.x, x = blockIdx.x & 7u;
    const unsigned int groups = G < 8u ? G : 8u, members = (G - x + 7u) / 8u;
    const unsigned int set = b % 3u, next = (b + 1u) % 3u;
    if (blockIdx.x == 0) {
      // the set after this one was last used two barriers ago: everybody has left it
      for (int i = 0; i < 8; ++i) publish(&st->xcd_count[next][i][0], 0u);
      publish(&st->top[next][0], 0ull);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned int a = __hip_atomic_fetch_add(&st->xcd_count[set][x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1u == members) (void)__hip_atomic_fetch_add(&st->top[set][0], 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int want = groups;
    unsigned int spins = 0;
    u64 w;
    while ((unsigned int)((w = __hip_atomic_load(&st->top[set][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit || fresh(&st->stop[0]) == 1u) {
        publish(&st->stop[0], 1u);
        w = ~0ull;
        break;
      }
    }
    s_word = w;
  }
  __syncthreads();
  const u64 w = s_word;
  *lo = (unsigned int)w;
  return w != ~0ull;
}

__global__ __launch_bounds__(kPThreads) void sssp_nfq_kernel(NfqArgs a) {
  __shared__ NfqEntry s_near2[2][kNfqStageNear];          // the staged near entries; two buffers in turn (sub-steps, below)
  __shared__ NfqEntry s_far[kNfqStageFar];
  int nsel = 0;                                           // the buffer being staged into
#define s_near (s_near2[nsel])
  __shared__ unsigned int s_cnt[4];                       // staged near, staged far, smallest far distance
  __shared__ unsigned int s_base[2];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int sub = lane & (kNfqLanes - 1), slot = lane / kNfqLanes;
  const int G = gridDim.x;
  const long long gtid = (long long)blockIdx.x * kPThreads + tid;
  const long long gthreads = (long long)G * kPThreads;
  const long long gwave = gtid >> 6, nwaves = gthreads >> 6;
  NfqState* st = a.st;
  const u64 t_start = wall_clock64();
  float T = a.delta;
  int pass = 1, converged = 0, fsel = 0;
  int ftarget = 0;                                        // the pile this pass adds to
  unsigned int bidx = 1;                                  // the barrier that ends this pass; the queue it fills is counted in that barrier's word
  unsigned int ncur = 1;                                  // the host queued the source
  u64 my_expanded = 0, my_relaxed = 0, my_queued = 0;
  if (tid == 0) { s_cnt[0] = 0u; s_cnt[1] = 0u; s_cnt[2] = 0xffffffffu; }
  __syncthreads();

  // the joiners of one wave step, staged; what does not fit goes to the list itself, a wave at a time
  auto stage = [&](bool joins, const NfqEntry& entry, NfqEntry* staged, unsigned int* staged_n, unsigned int room, NfqEntry* list,
                   bool near_list) {
    const u64 m = __ballot(joins);
    if (m == 0ull) return;
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(staged_n, (unsigned int)__popcll(m));
    base = __shfl(base, 0, kWave);
    const unsigned int pos = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    const bool over = joins && pos >= room;
    if (joins && !over) staged[pos] = entry;
    if (joins) ++my_queued;
    const u64 mo = __ballot(over);
    if (mo == 0ull) return;
    const unsigned int cnt = (unsigned int)__popcll(mo);
    unsigned int gb = 0;
    if (lane == 0)
      gb = near_list ? (unsigned int)__hip_atomic_fetch_add(&st->top[bidx % 3u][0], (u64)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                     : atomicAdd(&st->fcount[ftarget][0], cnt);
    gb = __shfl(gb, 0, kWave);
    const unsigned int cap = near_list ? a.near_cap : a.far_cap;
    if (gb + cnt > cap) { if (lane == 0) publish(&st->stop[0], 2u); return; }
    if (over) nfq_put(&list[gb + (unsigned int)__popcll(mo & ((1ull << lane) - 1ull))], entry);
  };
  // what the workgroup staged, into the lists: one request for slots per list
  auto flush = [&](NfqEntry* near_list) {
    NfqEntry* far_list = a.qf[ftarget];
    const int far_sel = ftarget;
    __syncthreads();
    const unsigned int sn = s_cnt[0], sf = s_cnt[1], mn = s_cnt[2];
    if (sn == 0u && sf == 0u && mn == 0xffffffffu) return;     // (the same for the whole workgroup)
    const unsigned int nn = sn < (unsigned int)kNfqStageNear ? sn : (unsigned int)kNfqStageNear;
    const unsigned int nf = sf < (unsigned int)kNfqStageFar ? sf : (unsigned int)kNfqStageFar;
    if (tid == 0 && nn)
      s_base[0] = (unsigned int)__hip_atomic_fetch_add(&st->top[bidx % 3u][0], (u64)nn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == kWave && nf) s_base[1] = atomicAdd(&st->fcount[far_sel][0], nf);
    if (tid == 2 * kWave && mn != 0xffffffffu) atomicMin(&st->minfar[far_sel][0], mn);
    __syncthreads();
    if (nn) {
      const unsigned int b0 = s_base[0];
      if (b0 + nn > a.near_cap) { if (tid == 0) publish(&st->stop[0], 2u); }
      else for (unsigned int i = tid; i < nn; i += kPThreads) nfq_put(&near_list[b0 + i], s_near[i]);
    }
    if (nf) {
      const unsigned int b1 = s_base[1];
      if (b1 + nf > a.far_cap) { if (tid == 0) publish(&st->stop[0], 2u); }
      else for (unsigned int i = tid; i < nf; i += kPThreads) nfq_put(&far_list[b1 + i], s_far[i]);
    }
    __syncthreads();
    if (tid == 0) { s_cnt[0] = 0u; s_cnt[1] = 0u; s_cnt[2] = 0xffffffffu; }
  };
  auto stage_far = [&](bool jf, const NfqEntry& entry) {
    if (__ballot(jf) == 0ull) return;
    const unsigned int db = jf ? (unsigned int)(entry.vd >> 32) : 0xffffffffu;
    const unsigned int rm = wave_min_u32(db);
    if (lane == 0) atomicMin(&s_cnt[2], rm);
    stage(jf, entry, s_far, &s_cnt[1], kNfqStageFar, a.qf[ftarget], false);
  };
  // one out-edge per active lane
  auto relax = [&](bool act, float du, unsigned int hu, Index p, NfqEntry* qnext) {
    bool jn = false, jf = false;
    NfqEntry entry = {0ull, 0ull};
    if (act) {
      const Index t = a.oind[p];
      const float dn = du + (a.unit ? 1.f : a.oval[p]);
      const u64 nk = nf_key(dn, hu + 1u);
      // the target's row bounds travel with its entry: asked for beside the atomic, not after it
      const Index ts = a.optr[t], te = a.optr[t + 1];
      if (nk < atomicMin(&a.K[t], nk)) {
        entry.vd = ((u64)__float_as_uint(dn) << 32) | (u64)(unsigned int)t;
        entry.se = ((u64)(unsigned int)te << 32) | (u64)(unsigned int)ts;
        jn = dn < T;
        jf = !jn;
      }
    }
    stage(jn, entry, s_near, &s_cnt[0], kNfqStageNear, qnext, true);
    stage_far(jf, entry);
  };

  for (; pass <= a.max_passes; ++pass) {
    NfqEntry* qnext = a.qn[(pass + 1) & 1];
    if (ncur == 0u) {
      // ---- nothing near: the threshold moves, the pile is dealt out
      const unsigned int nfar = fresh(&st->fcount[fsel][0]);
      if (nfar == 0u) { converged = 1; break; }
      const unsigned int mf = fresh(&st->minfar[fsel][0]);
      if (mf != 0xffffffffu) {
        // at least the next float above the smallest far distance: delta can vanish in the sum at large distances
        const float up = __uint_as_float(mf + 1u);
        T = __uint_as_float(mf) + a.delta;
        T = T > up ? T : up;
      } else {
        T = FLT_MAX;
      }
      const NfqEntry* pile = a.qf[fsel];
      ftarget = fsel ^ 1;
      for (long long b = gwave * kWave; b < (long long)nfar; b += nwaves * kWave) {
        const long long i = b + lane;
        bool jn = false, jf = false;
        NfqEntry entry = {0ull, 0ull};
        if (i < (long long)nfar) {
          entry = nfq_get(&pile[i]);
          const unsigned int db = (unsigned int)(entry.vd >> 32);
          const bool live = (unsigned int)(fresh(&a.K[(unsigned int)entry.vd]) >> 32) == db;
          jn = live && __uint_as_float(db) < T;
          jf = live && !jn;
        }
        stage(jn, entry, s_near, &s_cnt[0], kNfqStageNear, qnext, true);
        stage_far(jf, entry);
      }
      flush(qnext);
      if (!nfq_sync(st, bidx++, &ncur)) return;
      if (fresh(&st->stop[0]) != 0u) break;
      fsel ^= 1;
      if (gtid == 0) { publish(&st->fcount[fsel ^ 1][0], 0u); publish(&st->minfar[fsel ^ 1][0], 0xffffffffu); }
      continue;
    }
    // ---- a pass over the near queue
    const NfqEntry* qcur = a.qn[pass & 1];
    ftarget = fsel;
    // one wave step: the entry of this lane's slot (an all-zero entry is an empty slot), its edges, what they lower
    auto expand = [&](const NfqEntry& entry) {
      const Index s = (Index)(unsigned int)entry.se, e = (Index)(unsigned int)(entry.se >> 32);
      const u64 key = fresh(&a.K[(unsigned int)entry.vd]);           // in flight with the edges below
      const bool wide = e - s >= kNfWide;
      const float du = __uint_as_float((unsigned int)(entry.vd >> 32));
      // narrow rows: the first edge of every lane is loaded before the key is looked at
      Index t0 = 0;
      float w0 = 0.f;
      const bool first = !wide && s + sub < e;
      if (first) { t0 = a.oind[s + sub]; w0 = a.unit ? 1.f : a.oval[s + sub]; }
      const bool live = e > s && (unsigned int)(key >> 32) == (unsigned int)(entry.vd >> 32);   // else: lowered since, and queued again then
      const unsigned int hu = (unsigned int)key;
      if (live && sub == 0) { ++my_expanded; my_relaxed += (u64)(e - s); }
      {
        bool jn = false, jf = false;
        NfqEntry out = {0ull, 0ull};
        if (first && live) {
          const float dn = du + w0;
          const u64 nk = nf_key(dn, hu + 1u);
          const Index ts = a.optr[t0], te = a.optr[t0 + 1];
          if (nk < atomicMin(&a.K[t0], nk)) {
            out.vd = ((u64)__float_as_uint(dn) << 32) | (u64)(unsigned int)t0;
            out.se = ((u64)(unsigned int)te << 32) | (u64)(unsigned int)ts;
            jn = dn < T;
            jf = !jn;
          }
        }
        stage(jn, out, s_near, &s_cnt[0], kNfqStageNear, qnext, true);
        stage_far(jf, out);
      }
      for (Index off = sub + kNfqLanes; __ballot(live && !wide && s + off < e) != 0ull; off += kNfqLanes)
        relax(live && !wide && s + off < e, du, hu, s + off, qnext);
      for (u64 todo = __ballot(live && wide && sub == 0); todo; todo &= todo - 1) {       // a hub: the whole wave takes its edges
        const int src = __ffsll((long long)todo) - 1;
        const Index s2 = __shfl(s, src, kWave), e2 = __shfl(e, src, kWave);
        const float d2 = __shfl(du, src, kWave);
        const unsigned int h2 = __shfl(hu, src, kWave);
        for (Index p0 = s2; p0 < e2; p0 += kWave) relax(p0 + lane < e2, d2, h2, p0 + lane, qnext);
      }
    };
    for (long long b = gwave * kNfqSlots; b < (long long)ncur; b += nwaves * kNfqSlots) {
      const long long i = b + slot;
      NfqEntry entry = {0ull, 0ull};
      if (i < (long long)ncur) entry = nfq_get(&qcur[i]);            // (the 4 lanes of a slot read the same 16 bytes)
      expand(entry);
    }
    // ---- sub-steps: the near entries this workgroup staged are expanded here and now, and what THEY stage too, a.inner
    // times over, before anything goes to the global queue and the grid meets at the barrier.  Any order of relaxations
    // reaches the same fixed point of (distance, hops); a pass's price is its chain of dependent steps plus the barrier,
    // and a sub-step is the chain alone, its entries already in LDS.  (Not for a traversal -- unit weights, as_bfs: there
    // a pass is a level and a vertex is queued once, which running ahead would give up.)
    for (int subs = 0; subs < a.inner; ++subs) {
      __syncthreads();
      const unsigned int staged = s_cnt[0];
      const unsigned int nloc = staged < (unsigned int)kNfqStageNear ? staged : (unsigned int)kNfqStageNear;
      if (nloc == 0u) break;                                         // (the same for the whole workgroup)
      __syncthreads();
      const NfqEntry* mine = s_near2[nsel];
      nsel ^= 1;
      if (tid == 0) s_cnt[0] = 0u;
      __syncthreads();
      for (unsigned int b = (unsigned int)(tid >> 6) * kNfqSlots; b < nloc; b += kPWaves * kNfqSlots) {
        const unsigned int i = b + (unsigned int)slot;
        NfqEntry entry = {0ull, 0ull};
        if (i < nloc) entry = mine[i];
        expand(entry);
      }
    }
    flush(qnext);
    if (!nfq_sync(st, bidx++, &ncur)) return;
    if (fresh(&st->stop[0]) != 0u) break;
  }

  // ---- distances out, and the largest hop count of a reached vertex (the reference's round count - 1)
  unsigned int mh = 0, md = 0;
  u64 n_reached = 0, n_edges = 0;
  if (converged)
    for (long long i = gtid; i < a.n; i += gthreads) {
      const u64 key = fresh(&a.K[i]);
      const unsigned int h = (unsigned int)key;
      const float d = __uint_as_float((unsigned int)(key >> 32));
      if (a.as_bfs) {
        a.D[i] = h != 0xffffffffu ? d + 1.f : 0.f;          // depth labels: the source is 1, unreached vertices 0
        if (h != 0xffffffffu) { ++n_reached; n_edges += (u64)(a.optr[i + 1] - a.optr[i]); }
      } else {
        a.D[i] = d;
      }
      if (h != 0xffffffffu) {
        mh = h > mh ? h : mh;
        md = (unsigned int)(key >> 32) > md ? (unsigned int)(key >> 32) : md;
      }
    }
  mh = wave_max_u32(mh);
  md = wave_max_u32(md);
  const u64 w0 = wave_sum_u64(my_expanded), w1 = wave_sum_u64(my_relaxed), w2 = wave_sum_u64(my_queued);
  const u64 r0 = a.as_bfs ? wave_sum_u64(n_reached) : 0ull, r1 = a.as_bfs ? wave_sum_u64(n_edges) : 0ull;
  if (lane == 0) {
    if (mh) atomicMax(&st->maxhops[0], mh);
    if (md) atomicMax(&st->maxdist[0], md);
    if (w0) __hip_atomic_fetch_add(&st->work[0][0], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w1) __hip_atomic_fetch_add(&st->work[1][0], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w2) __hip_atomic_fetch_add(&st->work[2][0], w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r0) __hip_atomic_fetch_add(&st->tally[0][0], r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r1) __hip_atomic_fetch_add(&st->tally[1][0], r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned int unused;
  if (!nfq_sync(st, bidx++, &unused)) return;
  if (gtid == 0) {
    const u64 tag = (u64)(unsigned int)a.seq << 32;
    const float ms = (float)(wall_clock64() - t_start) * a.ticks_to_ms;
    u64 c0 = fresh(&st->work[0][0]), c1 = fresh(&st->work[1][0]) >> 4;
    const u64 c2 = fresh(&st->work[2][0]);
    if (a.as_bfs) { c0 = fresh(&st->tally[0][0]); c1 = fresh(&st->tally[1][0]); }     // (an edge total fits: nnz < 2^31)
    const unsigned int how = converged ? 1u : fresh(&st->stop[0]) == 2u ? 2u : 0u;     // 2: a list was full
    const unsigned int vals[8] = {fresh(&st->maxhops[0]), (unsigned int)pass, __float_as_uint(ms), how,
                                  fresh(&st->maxdist[0]), (unsigned int)(c0 > 0xffffffffull ? 0xffffffffull : c0),
                                  (unsigned int)(c1 > 0xffffffffull ? 0xffffffffull : c1),
                                  (unsigned int)(c2 > 0xffffffffull ? 0xffffffffull : c2)};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      __hip_atomic_store(&a.mail[k], tag | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#undef s_near

__global__ void nfq_seed_kernel(NfqEntry* q1, NfqState* st, Index source, const Index* optr) {
  if (threadIdx.x == 0) {                                        // distance 0
    q1[0].vd = (u64)(unsigned int)source;
    q1[0].se = ((u64)(unsigned int)optr[source + 1] << 32) | (u64)(unsigned int)optr[source];
  }
  if (threadIdx.x < 2) st->minfar[threadIdx.x][0] = 0xffffffffu;
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
  GRB_TRY(bfs_lanes_fence(ctx().stream));   // a whole-device grid must not meet a BFS lane's narrower one half-way (bfs_persist.hip)
  static int max_per_cu = 0;
  if (!max_per_cu) {
    int m2 = 0;
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_nearfar_kernel, kPThreads, 0));
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&m2, sssp_nfq_kernel, kPThreads, 0));
    if (m2 < max_per_cu) max_per_cu = m2;
    if (max_per_cu < 1) return GRB_NOT_IMPLEMENTED;
  }
  static float ticks_to_ms = 0.f;
  if (ticks_to_ms == 0.f) {
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ticks_to_ms = khz > 0 ? 1.0f / (float)khz : 1e-5f;
  }
  // the band: GRB_SSSP_DELTA x the mean edge weight.  Measured flat from 32 to 128 for the queue form (85-88 ms on
  // the road-like stand-in: fewer passes, each a little longer) and from 16 to 64 for the bitmap form
  static const double delta_env = getenv("GRB_SSSP_DELTA") ? atof(getenv("GRB_SSSP_DELTA")) : 0.0;
  // a pass either expands a vertex or raises the threshold past one: far fewer than n of each are ever needed;
  // the cap only turns a logic error into "not converged" (the rounds then run) instead of an endless launch
  const long long pass_cap = 8ll * (long long)n + 1024;
  const int max_passes = pass_cap > 0x7fffff00ll ? 0x7fffff00 : (int)pass_cap;
  // GRB_SSSP_QUEUE=0: the near set found by walking the dirty bitmap (the first form of this kernel) instead of kept in queues
  static const int use_queue = getenv("GRB_SSSP_QUEUE") ? atoi(getenv("GRB_SSSP_QUEUE")) : 1;
  void *p_zero, *p_k;
  c.bfs_prezero_ptr = nullptr;              // slot 7 is about to be overwritten
  GRB_TRY(scratch(8, 8 * (size_t)n + 8, &p_k));
  int seq = 0;
  unsigned int gv[8];
  for (int attempt = 0; attempt < 2; ++attempt) {
  const bool queue_form = use_queue != 0 && attempt == 0;
  float delta = (float)((delta_env > 0.0 ? delta_env : queue_form ? 64.0 : 32.0) * A->mean_value);
  if (!(delta > 0.f)) delta = 1.f;           // all-zero weights: any positive width
  if (queue_form) {
    const size_t st_bytes = (sizeof(NfqState) + 255) & ~(size_t)255;
    // one entry (16 bytes) per successful atomicMin: a pass writes a few thousand on a road network, and cannot
    // write more than it relaxes edges; the far pile keeps entries that have gone stale until it is dealt out.
    // Neither list is sized for the worst case -- when one fills up the kernel says so and the bitmap form runs
    size_t near_cap = (size_t)n + (size_t)A->nvals / 4 + 64, far_cap = 2 * (size_t)n + (size_t)A->nvals / 4 + 64;
    static const long long cap_env = getenv("GRB_SSSP_QUEUE_CAP") ? atoll(getenv("GRB_SSSP_QUEUE_CAP")) : 0;   // (tests: make the lists overflow)
    if (cap_env > 0) { near_cap = (size_t)cap_env; far_cap = (size_t)cap_env; }
    if (far_cap > 0xfffffff0ull) continue;
    void* p_q;
    GRB_TRY(scratch(7, st_bytes, &p_zero));
    GRB_TRY(scratch(11, sizeof(NfqEntry) * (2 * near_cap + 2 * far_cap), &p_q));
    NfqArgs a;
    a.optr = A->csr.ptr; a.oind = A->csr.ind; a.oval = (const float*)A->csr.val;
    a.n = n;
    a.delta = delta;
    a.K = (u64*)p_k;
    a.st = (NfqState*)p_zero;
    a.qn[0] = (NfqEntry*)p_q; a.qn[1] = a.qn[0] + near_cap;
    a.qf[0] = a.qn[1] + near_cap; a.qf[1] = a.qf[0] + far_cap;
    a.near_cap = (unsigned int)near_cap; a.far_cap = (unsigned int)far_cap;
    a.D = (float*)v->d_val;
    a.mail = c.d_hgran;
    a.seq = seq = ++c.mail_seq;
    a.ticks_to_ms = ticks_to_ms;
    a.max_passes = max_passes;
    a.unit = 0;
    a.as_bfs = 0;
    static const int nfq_inner = getenv("GRB_SSSP_SUBSTEPS") ? atoi(getenv("GRB_SSSP_SUBSTEPS")) : 16;
    a.inner = nfq_inner < 0 ? 0 : nfq_inner;
    GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, st_bytes, s));
    hipLaunchKernelGGL(nf_init_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a.K, n, (Index)source);
    hipLaunchKernelGGL(nfq_seed_kernel, dim3(1), dim3(64), 0, s, a.qn[1], a.st, (Index)source, a.optr);
    GRB_HIP_TRY(hipGetLastError());
    // (every CU takes part: the passes of a road network would be as fast with 32 workgroups -- their cost is the
    // chain of dependent steps, measured equal from 32 to 256 -- but a wide frontier wants the whole machine)
    hipLaunchKernelGGL(sssp_nfq_kernel, dim3(c.num_cu), dim3(kPThreads), 0, s, a);
    GRB_HIP_TRY(hipGetLastError());
  } else {
    const size_t st_bytes = (sizeof(NfState) + 255) & ~(size_t)255;
    const size_t zero_bytes = st_bytes + 4 * (size_t)nwords;
    GRB_TRY(scratch(7, zero_bytes, &p_zero));
    NfArgs a;
    a.optr = A->csr.ptr; a.oind = A->csr.ind; a.oval = (const float*)A->csr.val;
    a.n = n;
    a.source = source;
    a.delta = delta;
    a.K = (u64*)p_k;
    a.st = (NfState*)p_zero;
    a.dirty = (unsigned int*)((char*)p_zero + st_bytes);
    a.D = (float*)v->d_val;
    a.mail = c.d_hgran;
    a.seq = seq = ++c.mail_seq;
    a.ticks_to_ms = ticks_to_ms;
    a.max_passes = max_passes;
    static const int inner = getenv("GRB_SSSP_INNER") ? atoi(getenv("GRB_SSSP_INNER")) : 2;
    a.inner = inner < 1 ? 1 : inner;
    GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, zero_bytes, s));
    hipLaunchKernelGGL(nf_init_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a.K, n, (Index)source);
    hipLaunchKernelGGL(nf_seed_kernel, dim3(1), dim3(64), 0, s, a.dirty, &a.st->minfar[0][0], (Index)source);
    GRB_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sssp_nearfar_kernel, dim3(c.num_cu), dim3(kPThreads), 0, s, a);
    GRB_HIP_TRY(hipGetLastError());
  }
  if (wait_granules(seq, 8, gv) != GRB_SUCCESS) {       // the barrier gave up: round-exact path
    ++g_barrier_failures;
    return GRB_NOT_IMPLEMENTED;
  }
  g_barrier_failures = 0;
  if (queue_form && gv[3] == 2u) continue;              // a list was full: the bitmap form needs none
  break;
  }
  if (gv[3] != 1u) return GRB_NOT_IMPLEMENTED;
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

// a road network (few entries per row, many vertices) pushed level by level: the traversal bfs_queue_run serves
bool grb::bfs_queue_wanted(grb_matrix A, grb_descriptor desc) {
  static const int use = getenv("GRB_BFS_QUEUE") ? atoi(getenv("GRB_BFS_QUEUE")) : -1;   // -1 auto, 0 never, 1 whenever possible
  if (use == 0 || g_barrier_failures >= 3) return false;
  const Index n = A->nrows;
  if (use < 0 && !(A->nvals < 8ll * (long long)n && n >= (1 << 16))) return false;
  return desc->desc[GRB_MXVMODE] != GRB_PULLONLY;
}

// algorithm::bfs on a long-diameter, low-degree graph (a road network) through the same queues: with every weight 1
// a pass IS a level, a vertex is queued exactly once, and a level of a few thousand vertices costs the 7.6 us of a
// pass instead of the 26-31 us the bitmap kernel of bfs_persist.hip pays for walking and recycling 3 MB of bitmap
// (4896^2 grid, 8 134 levels: 215 -> see DESIGN.md 5.0).  GRB_SUCCESS: v holds the depth labels (source 1, unreached
// 0) and the totals are the one-launch traversal's; GRB_NOT_IMPLEMENTED: not this kind of graph / call, or the
// traversal would have been cut off by max_niter -- the caller runs the bitmap kernel.
grb_info grb::bfs_queue_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int* levels,
                            long long* reached, unsigned long long* edges, float* tight_ms) {
  if (!bfs_queue_wanted(A, desc)) return GRB_NOT_IMPLEMENTED;
  const Index n = A->nrows;
  Context& c = ctx();
  hipStream_t s = c.stream;
  GRB_TRY(bfs_lanes_fence(ctx().stream));   // a whole-device grid must not meet a BFS lane's narrower one half-way (bfs_persist.hip)
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_nfq_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_NOT_IMPLEMENTED;
  }
  static float ticks_to_ms = 0.f;
  if (ticks_to_ms == 0.f) {
    int khz = 0, dev = 0;
    GRB_HIP_TRY(hipGetDevice(&dev));
    GRB_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    ticks_to_ms = khz > 0 ? 1.0f / (float)khz : 1e-5f;
  }
  const size_t st_bytes = (sizeof(NfqState) + 255) & ~(size_t)255;
  const size_t near_cap = (size_t)n + 64, far_cap = 64;   // a vertex is queued once (its first key is its last); nothing is far
  void *p_zero, *p_k, *p_q;
  c.bfs_prezero_ptr = nullptr;              // slot 7 is about to be overwritten
  GRB_TRY(scratch(7, st_bytes, &p_zero));
  GRB_TRY(scratch(8, 8 * (size_t)n + 8, &p_k));
  GRB_TRY(scratch(11, sizeof(NfqEntry) * (2 * near_cap + 2 * far_cap), &p_q));
  NfqArgs a;
  a.optr = A->csr.ptr; a.oind = A->csr.ind; a.oval = nullptr;
  a.n = n;
  a.delta = FLT_MAX;                        // every distance is near: no far pile, no threshold
  a.K = (u64*)p_k;
  a.st = (NfqState*)p_zero;
  a.qn[0] = (NfqEntry*)p_q; a.qn[1] = a.qn[0] + near_cap;
  a.qf[0] = a.qn[1] + near_cap; a.qf[1] = a.qf[0] + far_cap;
  a.near_cap = (unsigned int)near_cap; a.far_cap = (unsigned int)far_cap;
  a.D = (float*)v->d_val;
  a.mail = c.d_hgran;
  a.seq = ++c.mail_seq;
  a.ticks_to_ms = ticks_to_ms;
  const long long pass_cap = (long long)n + 1024;
  a.max_passes = pass_cap > 0x7fffff00ll ? 0x7fffff00 : (int)pass_cap;
  a.unit = 1;
  a.as_bfs = 1;
  a.inner = 0;                              // a pass is a level
  GRB_HIP_TRY(hipMemsetAsync(p_zero, 0, st_bytes, s));
  hipLaunchKernelGGL(nf_init_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a.K, n, (Index)source);
  hipLaunchKernelGGL(nfq_seed_kernel, dim3(1), dim3(64), 0, s, a.qn[1], a.st, (Index)source, a.optr);
  GRB_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(sssp_nfq_kernel, dim3(c.num_cu), dim3(kPThreads), 0, s, a);
  GRB_HIP_TRY(hipGetLastError());
  unsigned int gv[8];
  if (wait_granules(a.seq, 8, gv) != GRB_SUCCESS) {
    ++g_barrier_failures;
    return GRB_NOT_IMPLEMENTED;
  }
  g_barrier_failures = 0;
  if (gv[3] != 1u) return GRB_NOT_IMPLEMENTED;
  const long long lv = (long long)gv[0] + 1;            // the level that finds nothing is counted, as the loop counts it
  if (lv > (long long)desc->max_niter) return GRB_NOT_IMPLEMENTED;   // it would have been cut off: the exact loop runs
  *levels = (int)lv;
  *reached = (long long)gv[5];
  *edges = (unsigned long long)gv[6];
  memcpy(tight_ms, &gv[2], 4);
  return GRB_SUCCESS;
}
