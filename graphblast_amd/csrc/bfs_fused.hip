// bfs_fused.hip -- direction-optimised BFS with a device-resident level loop.
//
// Same algorithm and same push/pull decisions as graphblas/algorithm/bfs.hpp:14-89 run
// through vxm (backend/cuda/operations.hpp:80-209 + vector.hpp:291-323 `convert`), but
// the per-level GraphBLAS call sequence  assign -> vxm(convert, spmspv|spmv) -> swap ->
// reduce  is collapsed to one expansion per level on a representation chosen for
// MI355X:
//
//   visited set     bitmap, n/8 bytes (512 KiB at RMAT-22: lives in every XCD's L2), not
//                   the reference's n x 4-byte float vector used as a 1-bit flag; a second
//                   per-matrix bitmap marks vertices without in-edges (never discoverable),
//                   so pull levels skip them without touching the matrix
//   frontier        never stored as a dense vector: the vertices a level discovered are
//                   exactly  visited_now & ~visited_before ; an ordered, atomic-free bitmap
//                   compaction (count / scan / list) turns that into the sorted index queue
//                   of the next push level, and its scan total is the frontier size the host
//                   needs for the direction decision
//   push level      degree scan + chunk partition + edge-balanced expansion
//                   (push_common.hpp); atomicOr on the bitmap deduplicates, the winner
//                   writes the label.  No queue append, hence no same-address atomics
//   pull level      one wave per 64 vertices = 2 bitmap words; lanes probe their own
//                   in-neighbour list serially (early exit), leftovers are finished by the
//                   whole wave with coalesced reads + ballot; an unvisited vertex is
//                   discoverable iff ANY in-neighbour is visited as of the start of the level;
//                   visited is double-buffered (read `in`, write `out` = in | new): no atomics
//   level control   one small D2H mailbox read per level (the reference synchronises after
//                   every runtime call and returns three scan totals per level)
//
// Labels are the reference's: v[i] = level at which i was discovered, source = 1,
// unreachable = 0, float32 (bit-exact vs SimpleReferenceBfs, test_bfs.hpp:11-61).
#include "bfs_kernels.hpp"

namespace grb {

__global__ void bfs_seed_kernel(unsigned int* visited, float* label, Index* queue, Index source,
                                const Index* __restrict__ out_ptr, unsigned long long* edges_acc,
                                unsigned int* ticket) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    visited[source >> 5] |= 1u << (source & 31);
    label[source] = 1.f;
    queue[0] = source;
    *edges_acc = (unsigned long long)(out_ptr[source + 1] - out_ptr[source]);
    *ticket = 0u;
  }
}

// labels == bad -> 0 (frontier discovered by the last allowed iteration is never assigned
// by the reference loop, bfs.hpp:48-66)
__global__ void bfs_unlabel_kernel(float* __restrict__ label, Index n, float bad) {
  for (Index i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (label[i] == bad) label[i] = 0.f;
}

}  // namespace grb

using namespace grb;

grb_info bfs_persistent_run(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, int profile,
                            grb_bfs_level* levels_out, int max_levels, int* levels, int* last_dir,
                            long long* reached, unsigned long long* edges, Index* nf_left, bool* hit_cap,
                            float* tight_ms);

// TEPS numerator recomputed from the labels (only needed when max_niter cut the search)
static grb_info bfs_tally_labels(const float* label, const Index* ptr, Index n, int64_t* edges, int32_t* reached) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  void* p_tally;
  GRB_TRY(scratch(10, 64 * sizeof(unsigned long long), &p_tally));
  GRB_HIP_TRY(hipMemsetAsync(p_tally, 0, 64 * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(bfs_tally_kernel, dim3(stream_grid(n, kBlock * 8)), dim3(kBlock), 0, s, label, ptr, n,
                     (unsigned long long*)p_tally);
  GRB_HIP_TRY(hipGetLastError());
  unsigned long long h_tally[64];
  GRB_HIP_TRY(hipMemcpyAsync(h_tally, p_tally, sizeof(h_tally), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  unsigned long long e = 0, r = 0;
  for (int i = 0; i < 32; ++i) { e += h_tally[2 * i]; r += h_tally[2 * i + 1]; }
  *edges = (int64_t)e;
  *reached = (int32_t)r;
  return GRB_SUCCESS;
}

// what the host does with the record of a one-launch traversal: lastmxv_, the labels of a search that max_niter cut
// short (the last frontier is never assigned by the reference loop), the result block, the vector's count
static grb_info bfs_one_launch_finish(grb_vector v, grb_matrix A, grb_descriptor desc, int p_levels, int p_dir, long long p_reached,
                                      unsigned long long p_edges, Index p_nf, bool p_cap, float p_ms, grb_bfs_result* result,
                                      int max_niter_queued = -1) {
  hipStream_t s = ctx().stream;
  const Index n = A->nrows;
  const int cap_niter = max_niter_queued >= 0 ? max_niter_queued : desc->max_niter;   // (a queued traversal: the cap it ran under)
  desc->lastmxv = p_dir ? GRB_PULLONLY : GRB_PUSHONLY;
  if (p_cap && p_nf > 0) {
    bfs_lanes_unfence();                                    // (a lane's next launch into v comes after this)
    hipLaunchKernelGGL(bfs_unlabel_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, (float*)v->d_val, n,
                       (float)(cap_niter + 1));
    GRB_HIP_TRY(hipGetLastError());
    int64_t e2 = 0; int32_t r2 = 0;
    GRB_TRY(bfs_tally_labels((const float*)v->d_val, A->csr.ptr, n, &e2, &r2));
    p_edges = (unsigned long long)e2; p_reached = r2;
  }
  if (result) {
    result->levels = p_levels;
    result->tight_ms = p_ms;
    result->edges_traversed = (int64_t)p_edges;
    result->reached = (int32_t)p_reached;
  }
  v->d_nnz = (Index)p_reached;
  return GRB_SUCCESS;
}

static bool bfs_use_persistent() {
  static const bool use = [] { const char* e = getenv("GRB_BFS_PERSISTENT"); return !e || atoi(e) != 0; }();
  return use;
}
static int g_persistent_failures = 0;   // three barrier give-ups in a row (each costs seconds of bounded spinning): the device is
                                        // evidently shared; stop trying for the rest of the process

// ---- queue now, wait later (grb_hip.h) ------------------------------------------------------------------------------
// K traversals into K vectors are K kernel pairs queued back to back on the library's stream, each with its own
// record in pinned host memory; the host looks at the records when it wants the results.  Nothing between two
// traversals waits for the host, so a host that is slow, descheduled or busy (the per-level host round trips of
// bfs.hpp:42-88 are the extreme case) costs nothing as long as it queues faster than the device traverses.
extern "C" grb_info grb_bfs_fused_enqueue(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                                          grb_bfs_ticket* ticket) { GRB_API_ENTER_BFSQ();
  if (!ticket) return GRB_NULL_POINTER;
  *ticket = 0;
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built || !A->csr.ptr || !A->csc.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
  if (A->nrows != A->ncols || v->nsize != A->nrows) return GRB_DIMENSION_MISMATCH;
  if (source < 0 || source >= A->nrows) return GRB_INVALID_INDEX;
  GRB_TRY(ctx_init());
  int slot = 0, seq = 0;
  GRB_TRY(bfs_ticket_take(&slot));
  const bool one_launch = A->format == 0 && bfs_use_persistent() && g_persistent_failures < 3 && !bfs_queue_wanted(A, desc);
  if (one_launch) {
    hipStream_t s = ctx().stream;
    const bool first_use = !A->d_no_in_edges || !A->d_pull_hint;
    GRB_TRY(ensure_empty_rows(&A->d_no_in_edges, A->csc, s));
    GRB_TRY(ensure_pull_hint(&A->d_pull_hint, A->csc, A->csr.ptr, s));
    // the matrix's skip bitmap and pull hint were just queued on the library's stream: a lane's stream is not ordered
    // behind that (its fence looks at other entry points only), so wait here, once per matrix
    if (first_use && bfs_lanes_setting(0) > 1) GRB_HIP_TRY(hipStreamSynchronize(s));
    GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
    const grb_info li = bfs_persistent_enqueue(v, A, source, desc, slot, &seq);
    if (li == GRB_SUCCESS) {
      *ticket = ((grb_bfs_ticket)(unsigned int)seq << 8) | (grb_bfs_ticket)slot;
      return GRB_SUCCESS;
    }
    if (li != GRB_NOT_IMPLEMENTED && li != GRB_PANIC) return li;
  }
  // a traversal the ring does not serve (road-network queues, CSR-only format, the host-driven fallback): run it now
  // and park the result; the ticket behaves the same
  grb_bfs_result res = {};
  GRB_TRY(grb_bfs_fused(v, A, source, desc, &res, nullptr, 0, 0));
  seq = ++ctx().mail_seq;
  bfs_ticket_store(slot, seq, res);
  *ticket = ((grb_bfs_ticket)(unsigned int)seq << 8) | (grb_bfs_ticket)slot;
  return GRB_SUCCESS;
}

extern "C" grb_info grb_bfs_wait(grb_bfs_ticket ticket, grb_bfs_result* result) { GRB_API_ENTER_BFSQ();
  const int slot = (int)(ticket & 0xff), seq = (int)(ticket >> 8);
  grb_vector v = nullptr; grb_matrix A = nullptr; grb_descriptor desc = nullptr; grb_index source = 0;
  grb_bfs_result parked = {};
  int max_niter_queued = -1;
  const int state = bfs_ticket_state(slot, seq, &v, &A, &desc, &source, &parked, &max_niter_queued);
  if (state == 0) return GRB_INVALID_VALUE;                 // never issued, or waited for already
  if (state == 2) {
    if (result) *result = parked;
    bfs_ticket_release(slot);
    return GRB_SUCCESS;
  }
  if (state == 3) {                                         // still waiting for its co-scheduled launch to fill: launch what there is
    GRB_TRY(bfs_co_flush());
    if (bfs_ticket_state(slot, seq, nullptr, nullptr, nullptr, nullptr, nullptr) == 4) {
      bfs_ticket_release(slot);                             // the launch was refused: the same traversal, now
      return grb_bfs_fused(v, A, source, desc, result, nullptr, 0, 0);
    }
  } else if (state == 4) {
    bfs_ticket_release(slot);
    return grb_bfs_fused(v, A, source, desc, result, nullptr, 0, 0);
  }
  int p_levels = 0, p_dir = 0;
  long long p_reached = 0;
  unsigned long long p_edges = 0;
  Index p_nf = 0;
  bool p_cap = false;
  float p_ms = 0.f;
  const grb_info pi = bfs_persistent_wait(slot, seq, &p_levels, &p_dir, &p_reached, &p_edges, &p_nf, &p_cap, &p_ms);
  bfs_ticket_release(slot);
  if (pi == GRB_PANIC) {
    // the launch did not run to its end (its grid barrier gave up), or it was queued behind one that did not: the same
    // traversal now, through grb_bfs_fused (which falls back to the host-driven level loop by itself)
    // (the whole device: the ticket may belong to a lane whose kernel is still writing v and its state blocks, and
    // what the re-run queues on the library's stream must come before the lanes' next launches)
    ++g_persistent_failures;
    GRB_HIP_TRY(hipDeviceSynchronize());
    bfs_lanes_unfence();
    return grb_bfs_fused(v, A, source, desc, result, nullptr, 0, 0);
  }
  GRB_TRY(pi);
  g_persistent_failures = 0;
  return bfs_one_launch_finish(v, A, desc, p_levels, p_dir, p_reached, p_edges, p_nf, p_cap, p_ms, result, max_niter_queued);
}

// Traversals in flight at once: with n > 1 the traversals queued by grb_bfs_fused_enqueue go round n lanes, each lane a
// stream of its own whose launches take num_cu / n workgroups, so that n launches are resident together.  n < 1 only
// queries.  Returns the previous value.
extern "C" int grb_bfs_set_lanes(int n) { GRB_API_ENTER_NOINFO();
  return bfs_lanes_setting(n);
}

// Traversals per launch: with k > 1 the traversals queued by grb_bfs_fused_enqueue are launched k at a time, side by
// side in one grid (bfs_persist.hip: bfs_co_kernel).  k < 1 only queries.  Returns the previous value.
extern "C" int grb_bfs_set_coschedule(int k) { grb::ApiScope api_scope__; (void)api_scope__.enter(false, false);   // (one of the queue's own)
  return bfs_co_setting(k);
}

// Measurement passes: HIP events (on the library's stream) around the launches of several traversals.  on != 0 starts
// collecting; on == 0 stops and reports the launches' summed duration, their number and the traversals they ran.
extern "C" grb_info grb_bfs_coschedule_profile(int on, double* launch_ms_total, int* launches, int* traversals) { GRB_API_ENTER_BFSQ();
  return bfs_co_profile(on, launch_ms_total, launches, traversals);
}

// Host time spent inside the one-launch traversal's two halves since the last reset: queueing the launches
// (argument block, two kernel launches) and waiting for / unpacking the record.  calls = traversals.
extern "C" grb_info grb_bfs_host_times(double* enqueue_us, double* wait_us, long long* calls, int reset) { GRB_API_ENTER_HOST();
  bfs_host_times(enqueue_us, wait_us, calls, reset != 0);
  return GRB_SUCCESS;
}

extern "C" grb_info grb_bfs_fused(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                                  grb_bfs_result* result, grb_bfs_level* levels_out, int max_levels, int profile) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built || !A->csr.ptr || !A->csc.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (v->dtype != GRB_F32) return GRB_DOMAIN_MISMATCH;
  if (A->nrows != A->ncols || v->nsize != A->nrows) return GRB_DIMENSION_MISMATCH;
  if (source < 0 || source >= A->nrows) return GRB_INVALID_INDEX;
  if (A->format != 0) {
    // GRB_SPARSE_MATRIX_FORMAT = 1: no CSC, vxm pushes whatever the mxvmode says (operations.hpp:131-133);
    // the call sequence of algorithm/bfs.hpp over the ops reproduces exactly that
    grb_info i = grb_bfs(v, A, source, desc, result);
    if (i == GRB_SUCCESS && result) GRB_TRY(bfs_tally_labels((const float*)v->d_val, A->csr.ptr, A->nrows,
                                                              &result->edges_traversed, &result->reached));
    return i;
  }
  GRB_TRY(ctx_init());
  Context& c = ctx();
  hipStream_t s = c.stream;
  const Index n = A->nrows;
  const int nwords = 2 * ceil_div(n, 64);            // two words per 64-vertex chunk
  const int btiles = ceil_div(nwords, kBlock);
  const int mode = desc->desc[GRB_MXVMODE];
  const long long max_edges = A->nvals;
  const long long max_chunks = max_edges / kEdgeChunk + 2;

  // vertices without in-edges, cached per matrix
  GRB_TRY(ensure_empty_rows(&A->d_no_in_edges, A->csc, s));
  GRB_TRY(ensure_pull_hint(&A->d_pull_hint, A->csc, A->csr.ptr, s));

  const bool use_persistent = bfs_use_persistent();
  int& persistent_failures = g_persistent_failures;
  if (use_persistent && persistent_failures < 3) {
    GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
    int p_levels = 0, p_dir = 0;
    long long p_reached = 0;
    unsigned long long p_edges = 0;
    Index p_nf = 0;
    bool p_cap = false;
    float p_ms = 0.f;
    // a road network (few entries per row, thousands of levels): the levels from queues (sssp_nearfar.hip) -- unless
    // per-level records were asked for, which only the bitmap kernel keeps
    grb_info pi = GRB_NOT_IMPLEMENTED;
    if (profile == 0 && !levels_out) pi = bfs_queue_run(v, A, source, desc, &p_levels, &p_reached, &p_edges, &p_ms);
    if (pi != GRB_SUCCESS)
      pi = bfs_persistent_run(v, A, source, desc, profile, levels_out, max_levels, &p_levels, &p_dir,
                              &p_reached, &p_edges, &p_nf, &p_cap, &p_ms);
    if (pi == GRB_PANIC) ++persistent_failures; else persistent_failures = 0;
    if (pi == GRB_PANIC || pi == GRB_NOT_IMPLEMENTED) {
      // the one-launch traversal could not run to its end here (launch refused, or its grid barrier gave up
      // because the grid was not co-resident -- other work on the device): same traversal, level loop driven
      // from the host (below), instead of an error
      static bool told = false;
      if (!told) fprintf(stderr, "libgrb_hip: one-launch BFS unavailable (Info %d), using the host-driven level loop\n", pi);
      told = true;
      GRB_HIP_TRY(hipStreamSynchronize(s));
      c.bfs_prezero_ptr = nullptr;
      goto host_loop;
    }
    GRB_TRY(pi);
    return bfs_one_launch_finish(v, A, desc, p_levels, p_dir, p_reached, p_edges, p_nf, p_cap, p_ms, result);
  }

host_loop:
  void *p_va, *p_vb, *p_q, *p_scan, *p_rs, *p_tiles, *p_bt;
  GRB_TRY(scratch(7, 4 * (size_t)nwords, &p_va));
  ctx().bfs_prezero_ptr = nullptr;          // this slot is about to be overwritten
  GRB_TRY(scratch(8, 4 * (size_t)nwords, &p_vb));
  GRB_TRY(scratch(9, 4 * (size_t)n + 4, &p_q));
  GRB_TRY(scratch(2, 4 * (size_t)n + 4, &p_scan));
  GRB_TRY(scratch(11, 4 * (size_t)n + 4, &p_rs));
  const int max_tiles = ceil_div(n, kDegTile);
  GRB_TRY(scratch(3, 4 * (size_t)(2 * max_tiles + 2), &p_tiles));
  GRB_TRY(scratch(6, 4 * (size_t)(2 * btiles + 4 + max_chunks + 2) + 8 * (size_t)(btiles + 2), &p_bt));
  unsigned int* vis = (unsigned int*)p_va;            // current visited set
  unsigned int* vis_alt = (unsigned int*)p_vb;        // the set before the last level
  Index* queue = (Index*)p_q;
  int* local_scan = (int*)p_scan;
  Index* row_start = (Index*)p_rs;
  int* tile_sums = (int*)p_tiles;
  int* tile_off = tile_sums + max_tiles;
  int* btile_counts = (int*)p_bt;
  int* btile_off = btile_counts + btiles;             // btiles + 1 entries
  Index* chunk_owner = (Index*)(btile_off + btiles + 2);
  unsigned long long* btile_deg =
      (unsigned long long*)(((uintptr_t)(chunk_owner + max_chunks + 2) + 7) & ~(uintptr_t)7);
  int* d_state = c.d_mail + 8;    // [1] expanded edges, [2..3] inspected (u64)
  unsigned long long* d_edges_acc = reinterpret_cast<unsigned long long*>(c.d_mail + 40);
  unsigned int* d_ticket = reinterpret_cast<unsigned int*>(c.d_mail + 44);

  GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));
  float* label = (float*)v->d_val;
  GRB_TRY(k_fill(GRB_F32, label, 0.0, n));
  GRB_HIP_TRY(hipMemsetAsync(vis, 0, 4 * (size_t)nwords, s));
  GRB_HIP_TRY(hipMemsetAsync(d_state, 0, 4 * sizeof(int), s));
  hipLaunchKernelGGL(bfs_seed_kernel, dim3(1), dim3(64), 0, s, vis, label, queue, source, A->csr.ptr, d_edges_acc,
                     d_ticket);
  GRB_HIP_TRY(hipGetLastError());

  // profile bit 0: HIP events around every level's expansion kernels (cheap, reusable pool)
  // profile bit 1: pull levels additionally count the edges they inspect
  static std::vector<hipEvent_t> pool;
  size_t used = 0;
  const bool count_inspected = (profile & 2) != 0;
  auto mark = [&]() -> grb_info {
    if (!profile) return GRB_SUCCESS;
    if (used == pool.size()) {
      hipEvent_t e;
      GRB_HIP_TRY(hipEventCreate(&e));
      pool.push_back(e);
    }
    GRB_HIP_TRY(hipEventRecord(pool[used++], s));
    return GRB_SUCCESS;
  };

  // state of the two frontier Vector objects of bfs.hpp (f1, f2): storage + ratio_
  bool f1_dense = (mode == GRB_PULLONLY);
  float ratio_f1 = 0.f, ratio_f2 = 0.f;
  bool have_queue = true;          // queue holds the current frontier (level 1: the source)
  Index nf = 1;
  int iter = 1, levels = 0;
  uint64_t edges_cum = 0;
  int64_t reached = 1;
  GRB_HIP_TRY(hipEventRecord(c.ev0, s));
  for (; iter <= desc->max_niter; ++iter) {
    // ---- vxm's direction decision on u = f1 (operations.hpp:131-140, vector.hpp:291-323)
    if (mode == GRB_PUSHPULL) {
      const float ratio = (float)nf / (float)n;
      if (!f1_dense) {
        if (ratio > desc->switchpoint && ratio > ratio_f1) f1_dense = true; else ratio_f1 = ratio;
      } else {
        if (ratio <= desc->switchpoint && ratio < ratio_f1) f1_dense = false; else ratio_f1 = ratio;
      }
    } else {
      f1_dense = (mode == GRB_PULLONLY);
    }
    if (count_inspected) GRB_HIP_TRY(hipMemsetAsync(d_state + 2, 0, 2 * sizeof(int), s));
    int mf = -1;
    if (!f1_dense) {
      if (!have_queue) {
        // the frontier is what the previous level discovered: list (vis & ~vis_alt), ordered,
        // with the tile offsets the count + scan of that level already produced
        hipLaunchKernelGGL(push_scan_tiles_kernel, dim3(1), dim3(kBlock), 0, s, btile_counts, btiles, btile_off,
                           d_state);
        GRB_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(bitmap_list_kernel, dim3(btiles), dim3(kBlock), 0, s, vis, vis_alt, nwords, btile_off,
                           queue);
        GRB_HIP_TRY(hipGetLastError());
        have_queue = true;
      }
      GRB_TRY(mark());
      GRB_TRY(lb_prepare(s, A->csr, queue, nf, local_scan, row_start, tile_sums, tile_off, d_state + 1));
      if (mode == GRB_PUSHPULL && desc->edgeswitch > 0.f && nf >= 32) {
        // extension: the reference switches on the frontier's VERTEX count only, so a 1 % frontier
        // that contains the hubs is expanded edge by edge; leave push when its out-edges exceed
        // edgeswitch * nnz (costs one extra mailbox read on such levels)
        GRB_TRY(fetch_ints(d_state + 1, 1, &mf));
        if ((double)mf > (double)desc->edgeswitch * (double)max_edges) {
          f1_dense = true;
          if (profile) --used;          // the prepare work is not part of the pull kernel's time
        }
      }
    }
    if (!f1_dense) {
      GRB_HIP_TRY(hipMemcpyAsync(vis_alt, vis, 4 * (size_t)nwords, hipMemcpyDeviceToDevice, s));
      BfsPushVisitor vis_fn{vis, label, (float)(iter + 1)};
      GRB_TRY(lb_run(s, A->csr, nf, max_edges, local_scan, row_start, tile_off, chunk_owner, vis_fn));
      GRB_TRY(mark());
      desc->lastmxv = GRB_PUSHONLY;
    } else {
      const int grid = stream_grid((long long)ceil_div(n, kWave) * kWave, kBlock);
      GRB_TRY(mark());
      if (count_inspected)
        // accounting runs walk the lists in storage order (no hint) so that the inspected-edge
        // count is the sequential early-exit count the oracle defines
        hipLaunchKernelGGL((bfs_pull_kernel<true>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           vis, A->d_no_in_edges, (const Index*)nullptr, vis_alt, 0, label, (float)(iter + 1),
                           reinterpret_cast<unsigned long long*>(d_state + 2));
      else
        hipLaunchKernelGGL((bfs_pull_kernel<false>), dim3(grid), dim3(kBlock), 0, s, A->csc.ptr, A->csc.ind, n, vis,
                           vis, A->d_no_in_edges, A->d_pull_hint, vis_alt, 0, label, (float)(iter + 1),
                           (unsigned long long*)nullptr);
      GRB_HIP_TRY(hipGetLastError());
      GRB_TRY(mark());
      std::swap(vis, vis_alt);       // vis = new set, vis_alt = set before this level
      desc->lastmxv = GRB_PULLONLY;
    }
    // close the level: discovered = |vis & ~vis_alt|, published by the kernel itself
    const int seq = ++c.mail_seq;
    hipLaunchKernelGGL(bfs_level_tail_kernel, dim3(btiles), dim3(kBlock), 0, s, vis, vis_alt, nwords, A->csr.ptr, n,
                       btile_counts, btile_deg, d_ticket, d_state, d_edges_acc, c.d_hgran, seq);
    GRB_HIP_TRY(hipGetLastError());
    unsigned int gv[6];
    GRB_TRY(wait_granules(seq, count_inspected ? 6 : 4, gv));
    int h[4] = {(int)gv[0], (int)gv[1], count_inspected ? (int)gv[4] : 0, count_inspected ? (int)gv[5] : 0};
    edges_cum = ((uint64_t)gv[3] << 32) | gv[2];
    reached += h[0];
    have_queue = false;
    if (!(f1_dense && count_inspected)) { h[2] = h[3] = 0; }
    if (levels_out && levels < max_levels) {
      grb_bfs_level& L = levels_out[levels];
      L.direction = f1_dense ? 1 : 0;
      L.frontier = nf;
      L.frontier_edges = f1_dense ? (count_inspected ? (int64_t)(((uint64_t)(uint32_t)h[3] << 32) | (uint32_t)h[2]) : 0)
                                  : (int64_t)h[1];
      L.discovered = h[0];
      L.ms = 0.f;
    }
    ++levels;
    // f2.swap(&f1): storage follows the output of this level, ratio_ slots exchange
    std::swap(ratio_f1, ratio_f2);
    nf = h[0];
    if (nf == 0) break;             // reduce(succ) == 0, bfs.hpp:75-76
  }
  GRB_HIP_TRY(hipEventRecord(c.ev1, s));
  const bool hit_cap = iter > desc->max_niter;
  if (hit_cap && nf > 0) {
    hipLaunchKernelGGL(bfs_unlabel_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, label, n,
                       (float)(desc->max_niter + 1));
    GRB_HIP_TRY(hipGetLastError());
  }
  GRB_HIP_TRY(hipEventSynchronize(c.ev1));
  unsigned long long tot_e = edges_cum, tot_r = (unsigned long long)reached;
  if (hit_cap && nf > 0) {
    // the unlabelled last frontier is not "reached": recount from the labels
    int64_t e2 = 0; int32_t r2 = 0;
    GRB_TRY(bfs_tally_labels(label, A->csr.ptr, n, &e2, &r2));
    tot_e = (unsigned long long)e2; tot_r = (unsigned long long)r2;
  }
  float ms = 0.f;
  GRB_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  if (result) {
    result->levels = levels;
    result->tight_ms = ms;
    result->edges_traversed = (int64_t)tot_e;
    result->reached = (int32_t)tot_r;
  }
  if (profile) {
    for (size_t i = 0; i + 1 < used; i += 2) {
      float lm = 0.f;
      (void)hipEventElapsedTime(&lm, pool[i], pool[i + 1]);
      size_t lv = i / 2;
      if (levels_out && (int)lv < max_levels && (int)lv < levels) levels_out[lv].ms = lm;
    }
  }
  v->d_nnz = (Index)tot_r;
  return GRB_SUCCESS;
}
