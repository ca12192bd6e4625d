// sssp_part_run.hip -- algorithm::sssp on the 1-D vertex partition in its FRONTIER form, the round loop on the
// device.  Round 2's partitioned SSSP relaxed every stored in-edge in every round (a dense
// MinimumPlus product over the whole shard: 8 880 full SpMVs on the road_usa-sized stand-in) and ran its loop in
// Python with torch arithmetic.  Here a round costs what its frontier costs:
//
//   state per rank   D[n]      tentative distances: the rank's OWNED entries are authoritative, every other entry is
//                              the smallest candidate this rank has sent or seen for that vertex (a filter)
//                    dcur[]    the owned distances as the current round started (rounds are synchronous: a round
//                              relaxes with the distances of the round before, graphblas/algorithm/sssp.hpp:53-90)
//                    queues    owned vertices improved in the previous round (being expanded) / in this one
//   launch           apply     the (vertex, candidate) pairs every rank sent last launch: atomicMin into D; an owned
//                              vertex improved for the first time this round joins the next queue
//                    -- grid barrier --
//                    finish    when every rank had finished expanding: the round ends -- dcur := D on the queue,
//                              queues swap, the round counter advances; a second barrier
//                    expand    the queue's out-edges: owned targets settled at once, others that beat the filter go
//                              to the outbox (wave-aggregated reservation, one atomic per 64 edges)
//   collective       all-gather of the fixed-size outboxes (header + <= cap pairs; RCCL on the communication stream)
//
// An outbox that fills up does not fail the round: the wave that could not reserve space records where it stopped
// (vertex, edge offset), the rank reports "not finished" in its header, and the next launch continues the SAME round
// from those points while the other ranks only apply -- so the number of rounds, and the distances after each, are
// the reference's whatever the capacity.  Termination lags one exchange (the improved counts travel in the headers);
// the host-side launch rule is grb_bfs_part_run's: launch k goes out when launch k - 2 has reported, every rank
// enqueues the same number of collectives.  With one rank several rounds run per launch.
//
// Weights must be non-negative (distances compare as unsigned integers in the atomics), as for grb_sssp's persistent
// kernel.  The reference has no multi-GPU code (backend/cuda/descriptor.hpp:242).
#include "persist_common.hpp"

#include <chrono>

namespace grb {

constexpr unsigned int kSsspNone = 0xffffffffu;       // an outbox slot without a pair
constexpr int kSsspHdr = 16;                          // header words of an outbox

struct SsspCarry {                                    // what the round loop carries from launch to launch
  int round;                                          // rounds finished
  int qsel;                                           // queue being expanded: Q[qsel]; improved vertices go to Q[qsel ^ 1]
  int csel;                                           // resume list being consumed: C[csel]
  unsigned qn;                                        // entries of Q[qsel]
  unsigned pos;                                       // entries of it already dealt out
  unsigned nc;                                        // entries of C[csel]
  long long local_changed;                            // owned vertices the last finished round improved
  int fresh_round;                                    // 1: a round finished in the last launch (its count is in the header)
  int done, done_at, hit_cap;
  int iterations;                                     // the reference's loop counter at exit
  int pad;
};

struct SsspState {
  GridBarrier bar[2];
  unsigned cursor[32];                                // outbox pairs reserved in this launch
  unsigned full[32];
  unsigned qcount[2][32];                             // entries appended to Q[0] / Q[1]
  unsigned ccount[2][32];                             // entries appended to C[0] / C[1]
  unsigned panic[32];
  SsspCarry carry[2];
};

struct SsspArgs {
  const Index *optr, *oind;                           // out-edges of the owned vertices (local rows, global columns)
  const float* oval;                                  // their weights
  Index n, lo, n_local;
  int world, rank;
  int max_niter;
  unsigned int* D;                                    // [n] float bits
  unsigned int* dcur;                                 // [n_local] float bits
  Index* Q[2];                                        // [n_local] each
  int2* C[2];                                         // resume points (vertex, edge offset)
  int ccap;
  unsigned int* outbox;                               // [kSsspHdr + 2 cap] this rank's send buffer
  const unsigned int* inboxes;                        // world x the same (== outbox when world == 1)
  int cap;                                            // pairs an outbox holds
  SsspState* st;
  unsigned long long* mail;
  SsspCarry* result;
  int launch, nsteps;
};

constexpr int kStageQ = 8192;           // queue joiners a workgroup stages per phase


__global__ __launch_bounds__(kPThreads) void sssp_part_kernel(SsspArgs a) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int G = gridDim.x;
  const long long gtid = (long long)blockIdx.x * kPThreads + tid;
  const long long gthreads = (long long)G * kPThreads;
  const long long gwave = gtid >> 6, nwaves = gthreads >> 6;
  const int k = a.launch;
  SsspState* st = a.st;
  GridBarrier* bar = &st->bar[k & 1];
  unsigned gen = 0;
  const Index lo = a.lo, hi = a.lo + a.n_local;
  if (blockIdx.x == 0) {
    unsigned* z = reinterpret_cast<unsigned*>(&st->bar[(k + 1) & 1]);
    for (int i = tid; i < (int)(sizeof(GridBarrier) / sizeof(unsigned)); i += kPThreads) publish(&z[i], 0u);
  }
  SsspCarry cy = st->carry[k & 1];
  auto report = [&](const SsspCarry& y, bool bad) {
    const unsigned long long hi32 = bad ? 0xffffffffull : (unsigned long long)(unsigned)(y.done_at + 1);
    __hip_atomic_store(a.mail, (hi32 << 32) | (unsigned long long)(unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  const bool panicked = fresh(&st->panic[0]) != 0u;
  if (cy.done || panicked) {
    if (gtid == 0) {
      st->carry[(k + 1) & 1] = cy;
      // an idle launch still sends a well-formed, empty outbox
      publish(&a.outbox[0], 0u); publish(&a.outbox[1], 1u); publish(&a.outbox[2], 0u); publish(&a.outbox[3], 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      report(cy, panicked);
    }
    return;
  }
  auto give_up = [&]() { if (tid == 0) publish(&st->panic[0], 1u); };
  const int stride = kSsspHdr + 2 * a.cap;
  // joiners of the next queue are staged per workgroup and get their slots with ONE atomic per phase: a wave step's
  // own atomicAdd on the queue's counter serialises with everybody else's (0.4-1 M vertices per round on a road
  // network: 60 K same-address atomics, ~0.75 ms of a round)
  __shared__ Index s_q[kStageQ];
  __shared__ unsigned s_qn, s_qbase;
  if (tid == 0) s_qn = 0u;
  __syncthreads();
  auto stage_q = [&](bool joins, Index val, unsigned* qcnt, Index* qnext) {
    const unsigned long long m = __ballot(joins);
    if (m == 0ull) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&s_qn, (unsigned)__popcll(m));
    base = __shfl(base, 0, kWave);
    if (!joins) return;
    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < (unsigned)kStageQ) s_q[pos] = val;
    else publish(&qnext[atomicAdd(qcnt, 1u)], val);
  };
  auto flush_q = [&](unsigned* qcnt, Index* qnext) {
    __syncthreads();
    const unsigned staged = s_qn < (unsigned)kStageQ ? s_qn : (unsigned)kStageQ;
    if (staged != 0u) {                                  // (the same for the whole workgroup)
      if (tid == 0) s_qbase = atomicAdd(qcnt, staged);
      __syncthreads();
      const unsigned b0 = s_qbase;
      for (unsigned i = tid; i < staged; i += kPThreads) publish(&qnext[b0 + i], s_q[i]);
      __syncthreads();
      if (tid == 0) s_qn = 0u;
    }
  };
  constexpr int kSlots = 16, kSubLanes = kWave / kSlots;   // queue vertices per wave step; lanes per vertex
  const int sub = lane & (kSubLanes - 1), slot = lane / kSubLanes;

  for (int step = 0; step < a.nsteps; ++step) {
    const bool first = (k == 0 && step == 0);            // the host seeded round 0's result: nothing to apply
    bool all_complete = true;
    long long prev_changed = 0;                          // improved vertices of the round that finished last launch
    int prev_fresh = 1;
    if (gtid == 0) { publish(&st->cursor[0], 0u); publish(&st->full[0], 0u); }
    if (!first) {
      // ---- apply: the pairs of every other rank (this rank's own pairs went into its D when they were sent)
      for (int q = 0; q < a.world; ++q) {
        const unsigned int* box = a.inboxes + (size_t)q * stride;
        const unsigned np = fresh(&box[0]);
        all_complete = all_complete && fresh(&box[1]) != 0u;
        prev_fresh = prev_fresh && fresh(&box[2]) != 0u;
        prev_changed += (long long)fresh(&box[3]);
        if (a.world == 1 || q == a.rank) continue;
        for (long long i0 = 0; i0 < (long long)np; i0 += gthreads) {
          const long long i = i0 + gtid;
          bool joins = false;
          Index lv = 0;
          if (i < (long long)np) {
            const unsigned v = fresh(&box[kSsspHdr + 2 * i]);
            const unsigned c = fresh(&box[kSsspHdr + 2 * i + 1]);
            if (v != kSsspNone) {
              if ((Index)v >= lo && (Index)v < hi) {
                lv = (Index)v - lo;
                if (c < fresh(&a.D[v])) {
                  const unsigned old = atomicMin(&a.D[v], c);
                  joins = c < old && old == fresh(&a.dcur[lv]);
                }
              } else if (c < fresh(&a.D[v])) {
                atomicMin(&a.D[v], c);
              }
            }
          }
          stage_q(joins, lv, &st->qcount[cy.qsel ^ 1][0], a.Q[cy.qsel ^ 1]);
        }
      }
      flush_q(&st->qcount[cy.qsel ^ 1][0], a.Q[cy.qsel ^ 1]);
      if (!grid_sync(bar, gen, false)) { give_up(); return; }
    }
    // ---- the loop's exit tests (sssp.hpp:53, :86-88), one exchange after the round they are about: the headers read
    // above say whether a round ended in the previous launch and how many vertices it improved, everywhere
    if (!first && cy.round >= 1 && prev_fresh) {
      if (prev_changed == 0) { cy.done = 1; cy.done_at = k; cy.iterations = cy.round; break; }
      if (cy.round >= a.max_niter) { cy.done = 1; cy.done_at = k; cy.hit_cap = 1; cy.iterations = a.max_niter + 1; break; }
    }
    // ---- the round ends when every rank has expanded all of its queue
    if (!first && all_complete) {
      const unsigned qnext = fresh(&st->qcount[cy.qsel ^ 1][0]);
      for (long long i = gtid; i < (long long)qnext; i += gthreads) {
        const Index lv = fresh(&a.Q[cy.qsel ^ 1][i]);
        publish(&a.dcur[lv], fresh(&a.D[lo + lv]));
      }
      cy.round += 1;
      cy.local_changed = (long long)qnext;
      cy.fresh_round = 1;
      cy.qsel ^= 1;
      cy.qn = qnext;
      cy.pos = 0;
      cy.nc = 0;
      if (gtid == 0) publish(&st->qcount[cy.qsel ^ 1][0], 0u);
      if (!grid_sync(bar, gen, false)) { give_up(); return; }
    } else if (!first) {
      cy.fresh_round = 0;
    }
    // ---- expand: resume points first, then the part of the queue not dealt out yet
    const bool capped = cy.round >= a.max_niter;         // the loop may not start another round: nothing is expanded
    int2* cin = a.C[cy.csel];
    int2* cout = a.C[cy.csel ^ 1];
    unsigned* ccnt = &st->ccount[cy.csel ^ 1][0];
    unsigned* qcnt = &st->qcount[cy.qsel ^ 1][0];
    auto relax = [&](Index u, Index p0) {
      const unsigned du = fresh(&a.dcur[u]);
      const float duf = __uint_as_float(du);
      const Index e = a.optr[u + 1];
      for (Index pb = p0; pb < e; pb += kWave) {
        if (a.world > 1 && fresh(&st->full[0]) != 0u) {
          if (lane == 0) { const unsigned at = atomicAdd(ccnt, 1u); if ((int)at < a.ccap) publish(reinterpret_cast<unsigned long long*>(&cout[at]), ((unsigned long long)(unsigned)pb << 32) | (unsigned)u); }
          return;
        }
        const Index p = pb + lane;
        const bool valid = p < e;
        const Index v = valid ? a.oind[p] : 0;
        const unsigned c = valid ? __float_as_uint(duf + a.oval[p]) : 0u;
        const bool owned = valid && v >= lo && v < hi;
        bool joins = false;
        if (owned && c < fresh(&a.D[v])) {
          const unsigned old = atomicMin(&a.D[v], c);
          joins = c < old && old == fresh(&a.dcur[v - lo]);
        }
        stage_q(joins, v - lo, qcnt, a.Q[cy.qsel ^ 1]);
        if (a.world > 1) {
          const bool emit = valid && !owned && c < fresh(&a.D[v]);
          const unsigned long long m = __ballot(emit);
          if (m != 0ull) {
            const int cnt = __popcll(m);
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&st->cursor[0], (unsigned)cnt);
            base = __shfl(base, 0, kWave);
            if (base + (unsigned)cnt > (unsigned)a.cap) {
              // no room: the slots reserved below the capacity stay empty, the rest of the vertex waits
              if (lane == 0) {
                publish(&st->full[0], 1u);
                const unsigned atc = atomicAdd(ccnt, 1u);
                if ((int)atc < a.ccap) publish(reinterpret_cast<unsigned long long*>(&cout[atc]), ((unsigned long long)(unsigned)pb << 32) | (unsigned)u);
              }
              for (unsigned s = base + lane; s < (unsigned)a.cap && s < base + (unsigned)cnt; s += kWave)
                publish(&a.outbox[kSsspHdr + 2 * s], kSsspNone);
              return;
            }
            if (emit) {
              const unsigned s = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
              publish(&a.outbox[kSsspHdr + 2 * s], (unsigned)v);
              publish(&a.outbox[kSsspHdr + 2 * s + 1], c);
              atomicMin(&a.D[v], c);
            }
          }
        }
      }
    };
    if (!capped) {
      for (long long i = gwave; i < (long long)cy.nc; i += nwaves) {
        const unsigned long long e = fresh(reinterpret_cast<const unsigned long long*>(&cin[i]));
        relax((Index)(e & 0xffffffffull), (Index)(e >> 32));
      }
      // the queue: 16 vertices per wave step, 4 lanes each (a road network's rows hold 2-4 entries: a wave per
      // vertex left 60 lanes idle and took 3.7 ms per round on the 4896^2 grid); rows of 64 and more entries
      // get the whole wave, one after the other
      for (long long b = (long long)cy.pos + gwave * kSlots; b < (long long)cy.qn; b += nwaves * kSlots) {
        const long long i = b + slot;
        const bool have = i < (long long)cy.qn;
        const Index u = have ? fresh(&a.Q[cy.qsel][i]) : 0;
        const Index s = have ? a.optr[u] : 0, e = have ? a.optr[u + 1] : 0;
        const bool wide = e - s >= kWave;
        const float duf = __uint_as_float(have ? fresh(&a.dcur[u]) : 0u);
        bool stopped = false;
        for (Index off = 0; !stopped && __ballot(!wide && s + off + sub < e) != 0ull; off += kSubLanes) {
          if (a.world > 1 && fresh(&st->full[0]) != 0u) {
            // no room in the outbox: what is left of every row of this step waits for the next launch
            if (sub == 0 && !wide && s + off < e) {
              const unsigned at = atomicAdd(ccnt, 1u);
              if ((int)at < a.ccap) publish(reinterpret_cast<unsigned long long*>(&cout[at]), ((unsigned long long)(unsigned)(s + off) << 32) | (unsigned)u);
            }
            stopped = true;
            break;
          }
          const Index p = s + off + sub;
          const bool valid = !wide && p < e;
          const Index v = valid ? a.oind[p] : 0;
          const unsigned c = valid ? __float_as_uint(duf + a.oval[p]) : 0u;
          const bool owned = valid && v >= lo && v < hi;
          bool joins = false;
          if (owned && c < fresh(&a.D[v])) {
            const unsigned old = atomicMin(&a.D[v], c);
            joins = c < old && old == fresh(&a.dcur[v - lo]);
          }
          stage_q(joins, v - lo, qcnt, a.Q[cy.qsel ^ 1]);
          if (a.world > 1) {
            const bool emit = valid && !owned && c < fresh(&a.D[v]);
            const unsigned long long m = __ballot(emit);
            if (m != 0ull) {
              const int cnt = __popcll(m);
              unsigned base = 0;
              if (lane == 0) base = atomicAdd(&st->cursor[0], (unsigned)cnt);
              base = __shfl(base, 0, kWave);
              if (base + (unsigned)cnt > (unsigned)a.cap) {
                if (lane == 0) publish(&st->full[0], 1u);
                if (sub == 0 && !wide && s + off < e) {
                  const unsigned atc = atomicAdd(ccnt, 1u);
                  if ((int)atc < a.ccap) publish(reinterpret_cast<unsigned long long*>(&cout[atc]), ((unsigned long long)(unsigned)(s + off) << 32) | (unsigned)u);
                }
                for (unsigned q = base + lane; q < (unsigned)a.cap && q < base + (unsigned)cnt; q += kWave)
                  publish(&a.outbox[kSsspHdr + 2 * q], kSsspNone);
                stopped = true;
                break;
              }
              if (emit) {
                const unsigned q = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                publish(&a.outbox[kSsspHdr + 2 * q], (unsigned)v);
                publish(&a.outbox[kSsspHdr + 2 * q + 1], c);
                atomicMin(&a.D[v], c);
              }
            }
          }
        }
        for (unsigned long long todo = __ballot(wide && sub == 0); todo; todo &= todo - 1) {
          const int src = __ffsll((long long)todo) - 1;
          const Index u2 = __shfl(u, src, kWave), s2 = __shfl(s, src, kWave);
          relax(u2, s2);
        }
      }
    }
    // ---- what this launch leaves behind
    flush_q(qcnt, a.Q[cy.qsel ^ 1]);
    if (!grid_sync(bar, gen, false)) { give_up(); return; }
    const unsigned left = fresh(&st->ccount[cy.csel ^ 1][0]);
    unsigned sent = fresh(&st->cursor[0]);
    if (sent > (unsigned)a.cap) sent = (unsigned)a.cap;
    if (left > (unsigned)a.ccap) { give_up(); return; }
    cy.pos = cy.qn;
    cy.csel ^= 1;
    cy.nc = left;
    if (gtid == 0) {
      publish(&st->ccount[cy.csel ^ 1][0], 0u);
      publish(&a.outbox[0], sent);
      publish(&a.outbox[1], left == 0u ? 1u : 0u);
      publish(&a.outbox[2], cy.fresh_round ? 1u : 0u);
      publish(&a.outbox[3], cy.fresh_round ? (unsigned)cy.local_changed : 0u);
    }
    // several steps in one launch (one rank only): everything above must have landed before the next step reads it
    if (step + 1 < a.nsteps && !grid_sync(bar, gen, false)) { give_up(); return; }
  }
  if (gtid == 0) {
    if (cy.done) {                                       // an ending launch leaves an empty, finished outbox too
      publish(&a.outbox[0], 0u); publish(&a.outbox[1], 1u); publish(&a.outbox[2], 0u); publish(&a.outbox[3], 0u);
    }
    st->carry[(k + 1) & 1] = cy;
    if (cy.done) *a.result = cy;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    report(cy, false);
  }
}

// D = FLT_MAX everywhere, 0 at the source; dcur likewise; the source in its owner's queue
__global__ void sssp_part_init_kernel(unsigned int* __restrict__ D, Index n, unsigned int* __restrict__ dcur, Index lo,
                                      Index n_local, Index source, Index* __restrict__ q0, SsspState* st) {
  const unsigned inf = __float_as_uint(FLT_MAX);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    D[i] = i == source ? 0u : inf;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_local; i += (long long)gridDim.x * blockDim.x)
    dcur[i] = (lo + i) == source ? 0u : inf;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    SsspCarry cy = {};
    cy.done_at = -1;
    if (source >= lo && source < lo + n_local) { q0[0] = source - lo; cy.qn = 1; cy.local_changed = 1; }
    cy.fresh_round = 1;
    st->carry[0] = cy;
  }
}

__global__ void sssp_part_copy_kernel(const unsigned int* __restrict__ D_own, float* __restrict__ out, Index n_local) {
  const Index i = (Index)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_local) out[i] = __uint_as_float(D_own[i]);
}

constexpr int kLoopMaxRanksS = 16;
struct LoopbackPtrsS {
  const unsigned int* send[kLoopMaxRanksS];
  unsigned int* recv[kLoopMaxRanksS];
};
__global__ __launch_bounds__(kBlock) void sssp_loopback_allgather_kernel(LoopbackPtrsS p, int world, int nwords) {
  const long long total = (long long)world * nwords;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int r = (int)(i / nwords);
    const unsigned int x = p.send[r][i - (long long)r * nwords];
    for (int q = 0; q < world; ++q) p.recv[q][i] = x;
  }
}

}  // namespace grb

using namespace grb;

struct grb_part_sssp_s {
  int rank = 0, world = 1;
  Index n = 0, lo = 0, n_local = 0;
  grb_matrix A_out = nullptr;            // rows = owned vertices, values = weights (f32)
  int cap = 0, ccap = 0;
  char* d_block = nullptr;               // state | D | dcur | Q0 | Q1 | C0 | C1 | outbox
  size_t st_bytes = 0;
  unsigned int *D = nullptr, *dcur = nullptr, *outbox = nullptr, *inboxes = nullptr;
  Index* Q[2] = {nullptr, nullptr};
  int2* C[2] = {nullptr, nullptr};
  unsigned long long *h_mail = nullptr, *d_mail = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

void sssp_mail_peek(grb_part_sssp p, int* launches_done, int* done_at) {
  const unsigned long long g = __atomic_load_n(p->h_mail, __ATOMIC_ACQUIRE);
  *launches_done = (int)(g & 0xffffffffull);
  const unsigned int hi = (unsigned int)(g >> 32);
  *done_at = hi == 0xffffffffu ? -2 : (int)hi - 1;
}

grb_info part_sssp_run(grb_part_sssp* ps, int nranks, grb_index source, int max_niter, int rounds_per_launch,
                       float* const* d_dist_local, grb_part_sssp_result* res) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  grb_part_sssp p0 = ps[0];
  const bool loopback = nranks > 1;
  const int world = p0->world;
  if (loopback && (world != nranks || nranks > kLoopMaxRanksS)) return GRB_INVALID_VALUE;
  if (source < 0 || source >= p0->n) return GRB_INVALID_INDEX;
  if (max_niter < 1) return GRB_INVALID_VALUE;
  if (!loopback && world > 1) {
    int r = -1, w = 0;
    grb_comm_info(&r, &w);
    if (w != world || r != p0->rank) return GRB_UNINITIALIZED_OBJECT;
  }
  if (world > 1 || rounds_per_launch < 1) rounds_per_launch = 1;
  GRB_TRY(bfs_lanes_fence(ctx().stream));   // a whole-device grid must not meet a BFS lane's narrower one half-way (bfs_persist.hip)
  static int max_per_cu = 0;
  if (!max_per_cu) {
    GRB_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_per_cu, sssp_part_kernel, kPThreads, 0));
    if (max_per_cu < 1) return GRB_PANIC;
  }
  const int G = c.num_cu;
  const int words = kSsspHdr + 2 * p0->cap;
  std::vector<SsspArgs> args(nranks);
  LoopbackPtrsS lp = {};
  for (int r = 0; r < nranks; ++r) {
    grb_part_sssp p = ps[r];
    if (p->cap != p0->cap) return GRB_INVALID_VALUE;
    SsspArgs& a = args[r];
    a.optr = p->A_out->csr.ptr; a.oind = p->A_out->csr.ind; a.oval = (const float*)p->A_out->csr.val;
    a.n = p->n; a.lo = p->lo; a.n_local = p->n_local;
    a.world = world; a.rank = p->rank;
    a.max_niter = max_niter;
    a.D = p->D; a.dcur = p->dcur;
    a.Q[0] = p->Q[0]; a.Q[1] = p->Q[1];
    a.C[0] = p->C[0]; a.C[1] = p->C[1];
    a.ccap = p->ccap;
    a.outbox = p->outbox;
    a.inboxes = world > 1 ? p->inboxes : p->outbox;
    a.cap = p->cap;
    a.st = (SsspState*)p->d_block;
    a.mail = p->d_mail;
    a.result = reinterpret_cast<SsspCarry*>(p->d_mail + 8);
    a.launch = 0;
    a.nsteps = rounds_per_launch;
    __atomic_store_n(p->h_mail, 0ull, __ATOMIC_RELEASE);
    GRB_HIP_TRY(hipMemsetAsync(p->d_block, 0, p->st_bytes, s));
    GRB_HIP_TRY(hipMemsetAsync(p->outbox, 0, 4 * (size_t)kSsspHdr, s));
    hipLaunchKernelGGL(sssp_part_init_kernel, dim3(stream_grid(p->n)), dim3(kBlock), 0, s, p->D, p->n, p->dcur, p->lo,
                       p->n_local, (Index)source, p->Q[0], a.st);
    GRB_HIP_TRY(hipGetLastError());
    if (loopback) { lp.send[r] = p->outbox; lp.recv[r] = p->inboxes; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  GRB_HIP_TRY(hipEventRecord(p0->ev0, s));
  int k = 0;
  for (;;) {
    if (k >= 2) {                                        // the launch rule of grb_bfs_part_run
      unsigned spins = 0;
      bool synced = false;
      int ld = 0, done_at = -1;
      for (;;) {
        sssp_mail_peek(p0, &ld, &done_at);
        if (ld >= k - 1) break;
        if ((++spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
          if (synced) return GRB_PANIC;
          GRB_HIP_TRY(hipStreamSynchronize(s));
          synced = true;
        }
      }
      if (done_at == -2) return GRB_PANIC;
      if (done_at >= 0 && done_at <= k - 2) break;
    }
    for (int r = 0; r < nranks; ++r) {
      args[r].launch = k;
      hipLaunchKernelGGL(sssp_part_kernel, dim3(G), dim3(kPThreads), 0, s, args[r]);
      GRB_HIP_TRY(hipGetLastError());
    }
    if (loopback) {
      hipLaunchKernelGGL(sssp_loopback_allgather_kernel, dim3(stream_grid((long long)nranks * words)), dim3(kBlock), 0, s, lp,
                         nranks, words);
      GRB_HIP_TRY(hipGetLastError());
    } else if (world > 1) {
      GRB_TRY(grb_comm_allgather(p0->outbox, p0->inboxes, 4 * (size_t)words));
      GRB_TRY(grb_comm_wait());
    }
    ++k;
    if (k > (1 << 28)) return GRB_PANIC;
  }
  for (int r = 0; r < nranks; ++r) {
    grb_part_sssp p = ps[r];
    if (p->n_local > 0 && d_dist_local[r]) {
      hipLaunchKernelGGL(sssp_part_copy_kernel, dim3(ceil_div(p->n_local, kBlock)), dim3(kBlock), 0, s, p->D + p->lo, d_dist_local[r],
                         p->n_local);
      GRB_HIP_TRY(hipGetLastError());
    }
  }
  GRB_HIP_TRY(hipEventRecord(p0->ev1, s));
  GRB_HIP_TRY(hipEventSynchronize(p0->ev1));
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, p0->ev0, p0->ev1));
  for (int r = 0; r < nranks; ++r) {
    grb_part_sssp p = ps[r];
    int ld = 0, done_at = -1;
    sssp_mail_peek(p, &ld, &done_at);
    SsspCarry cy;
    memcpy(&cy, p->h_mail + 8, sizeof(cy));
    if (done_at < 0 || !cy.done) return GRB_PANIC;
    if (res) {
      res[r].iterations = cy.iterations;
      res[r].rounds = cy.round;
      res[r].launches = k;
      res[r].hit_cap = cy.hit_cap;
      res[r].ms = ms;
    }
  }
  return GRB_SUCCESS;
}

}  // namespace

extern "C" {

grb_info grb_part_sssp_new(grb_part_sssp* out, int rank, int world, grb_index n_global, grb_index lo, grb_matrix A_out,
                           int outbox_pairs) { GRB_API_ENTER();
  if (!out) return GRB_NULL_POINTER;
  if (!A_out || !A_out->built) return GRB_UNINITIALIZED_OBJECT;
  if (A_out->dtype != GRB_F32 || (A_out->nvals > 0 && !A_out->csr.val)) return GRB_DOMAIN_MISMATCH;
  if (world < 1 || rank < 0 || rank >= world || lo < 0 || outbox_pairs < 64) return GRB_INVALID_VALUE;
  if (A_out->ncols != n_global || lo + A_out->nrows > n_global) return GRB_DIMENSION_MISMATCH;
  GRB_TRY(ctx_init());
  grb_part_sssp p = new grb_part_sssp_s();
  p->rank = rank; p->world = world;
  p->n = n_global; p->lo = lo; p->n_local = A_out->nrows;
  p->A_out = A_out;
  p->cap = outbox_pairs;
  const size_t nl = (size_t)(p->n_local > 0 ? p->n_local : 1);
  p->ccap = (int)nl + 4096;
  p->st_bytes = (sizeof(SsspState) + 255) & ~(size_t)255;
  const size_t words = kSsspHdr + 2 * (size_t)p->cap;
  const size_t bytes = p->st_bytes + 4 * (size_t)n_global + 4 * nl + 2 * 4 * nl + 2 * 8 * (size_t)p->ccap + 4 * words + 256;
  auto fail = [&](grb_info i) { grb_part_sssp_free(p); return i; };
  if (hipMalloc((void**)&p->d_block, bytes) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  char* q = p->d_block + p->st_bytes;
  p->D = (unsigned int*)q; q += 4 * (size_t)n_global;
  p->dcur = (unsigned int*)q; q += 4 * nl;
  p->Q[0] = (Index*)q; q += 4 * nl;
  p->Q[1] = (Index*)q; q += 4 * nl;
  q = (char*)(((uintptr_t)q + 7) & ~(uintptr_t)7);
  p->C[0] = (int2*)q; q += 8 * (size_t)p->ccap;
  p->C[1] = (int2*)q; q += 8 * (size_t)p->ccap;
  p->outbox = (unsigned int*)q;
  if (world > 1 && hipMalloc((void**)&p->inboxes, 4 * words * (size_t)world) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  if (hipHostMalloc((void**)&p->h_mail, 256, hipHostMallocMapped) != hipSuccess) return fail(GRB_OUT_OF_MEMORY);
  memset(p->h_mail, 0, 256);
  static_assert(sizeof(SsspCarry) <= 256 - 64, "the result record follows the mail word in pinned memory");
  if (hipHostGetDevicePointer((void**)&p->d_mail, p->h_mail, 0) != hipSuccess) return fail(GRB_PANIC);
  if (hipEventCreate(&p->ev0) != hipSuccess || hipEventCreate(&p->ev1) != hipSuccess) return fail(GRB_PANIC);
  *out = p;
  return GRB_SUCCESS;
}

grb_info grb_part_sssp_free(grb_part_sssp p) { GRB_API_ENTER();
  if (!p) return GRB_SUCCESS;
  (void)hipStreamSynchronize(ctx().stream);
  if (p->d_block) (void)hipFree(p->d_block);
  if (p->inboxes) (void)hipFree(p->inboxes);
  if (p->h_mail) (void)hipHostFree(p->h_mail);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  delete p;
  return GRB_SUCCESS;
}

grb_info grb_sssp_part_run(grb_part_sssp p, grb_index source, int max_niter, int rounds_per_launch, float* d_dist_local,
                           grb_part_sssp_result* result) { GRB_API_ENTER();
  if (!p) return GRB_UNINITIALIZED_OBJECT;
  float* out[1] = {d_dist_local};
  return part_sssp_run(&p, 1, source, max_niter, rounds_per_launch, out, result);
}

grb_info grb_sssp_part_run_group(grb_part_sssp* parts, int nranks, grb_index source, int max_niter,
                                 float* const* d_dist_local, grb_part_sssp_result* results) { GRB_API_ENTER();
  if (!parts || !d_dist_local || nranks < 1) return GRB_NULL_POINTER;
  for (int r = 0; r < nranks; ++r)
    if (!parts[r] || parts[r]->rank != r) return GRB_INVALID_VALUE;
  return part_sssp_run(parts, nranks, source, max_niter, 1, d_dist_local, results);
}

}  // extern "C"
