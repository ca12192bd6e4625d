// mxm_core.hip -- the DENSE CORE of the masked product C<L> = L (+.x) L^T (triangle counting, algorithm/tc.hpp:15-54;
// the reference's kernel: backend/cuda/kernels/spgemm.hpp:17-79, one sorted-list intersection per mask entry).
//
// In a power-law graph much of that product sits between a few thousand rows: on the RMAT-22 ef-28 stand-in of config 5
// the 16 Ki longest rows of L (0.4 % of the vertices) hold 11 % of the mask's entries between them and 39 % of the
// 6.94e9 hits, the 32 Ki longest 20 % and 61 %.  Among those rows the lists are dense enough to be BIT rows: H[r][c] = 1
// when the r-th core row has the c-th core row's vertex as an entry (ranks follow the vertex order, so H is strictly
// lower triangular like L).
// The product restricted to the core is then
//       C_H(i, j) = sum_k H[i][k] * H[j][k]        for the mask's entries (i, j) = the set bits of H itself,
// which is a dense K x K x K problem with two ways to run it on a CU:
//   popcount   a lane per MASK ENTRY, 32 columns of k per AND + v_bcnt: work = entries x K / 32, nothing is computed
//              for the pairs (i, j) that are not entries
//   MFMA       v_mfma_i32_16x16x64_i8 on 0/1 bytes expanded from the bit rows in registers: every pair of a 128 x 128
//              tile is computed, 16 384 multiply-adds per instruction, and the mask is applied to the finished tile
// The first wins where the mask is sparse, the second where it is dense; a tile's entry count decides (method 2).
// Both read the same bit rows through the same LDS staging and write the same per-entry results, so the A/B is a
// matter of calling grb_tc_dense_core twice (tools/tc_core_ab.py; tests/test_gpu_mxm.py compares both with numpy).
#include "common.hpp"

namespace grb {

constexpr int kCoreTile = 128;        // rows (and columns) of an output tile: a 256-thread workgroup, 2 x 2 waves of 64 x 64
constexpr int kCoreChunkW = 32;       // words of a bit row staged per step: 1024 values of k
constexpr int kCorePitch = kCoreChunkW + 1;   // (odd pitch: the 16 rows a wave instruction touches fall into 16 banks)
constexpr int kCoreBatch = 4096;      // mask entries of a tile the popcount kernel carries at once (16 per thread)
typedef int CoreV4 __attribute__((ext_vector_type(4)));

// row lengths, capped, as a histogram (LDS-private per workgroup): the host reads the threshold that leaves <= K rows
__global__ __launch_bounds__(1024) void core_len_hist_kernel(const Index* __restrict__ ptr, Index n, unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) h[i] = 0u;
  __syncthreads();
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (Index)gridDim.x * blockDim.x) {
    const Index d = ptr[v + 1] - ptr[v];
    atomicAdd(&h[d < 16383 ? d : 16383], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 16384; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}
__global__ void core_flag_kernel(const Index* __restrict__ ptr, Index n, Index theta, unsigned int* __restrict__ flag /* [n + 1] */) {
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += (Index)gridDim.x * blockDim.x)
    flag[v] = (v < n && ptr[v + 1] - ptr[v] >= theta) ? 1u : 0u;
}
__global__ void core_rank_kernel(const Index* __restrict__ ptr, Index n, Index theta, const unsigned int* __restrict__ before,
                                 int* __restrict__ rank, Index* __restrict__ rows) {
  for (Index v = (Index)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (Index)gridDim.x * blockDim.x) {
    const bool core = ptr[v + 1] - ptr[v] >= theta;
    rank[v] = core ? (int)before[v] : -1;
    if (core) rows[before[v]] = v;
  }
}
// the bit rows: a wave per core row walks the row's entries, an entry that is a core vertex sets its bit
__global__ __launch_bounds__(kBlock) void core_fill_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                           const Index* __restrict__ rows, int K, const int* __restrict__ rank,
                                                           unsigned int* __restrict__ H, int Wr, unsigned int* __restrict__ colcnt /* nullable */) {
  const int lane = threadIdx.x & (kWave - 1);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < K; r += nw) {
    const Index x = rows[r];
    const Index e = ptr[x + 1];
    for (Index p = ptr[x] + lane; p < e; p += kWave) {
      const int c = rank[ind[p]];
      if (c >= 0) {
        atomicOr(&H[(size_t)r * Wr + (c >> 5)], 1u << (c & 31));
        if (colcnt) atomicAdd(&colcnt[c], 1u);
      }
    }
  }
}
// per row: how many bits stand before each word (the per-entry results are stored in row-major order of the set bits)
__global__ __launch_bounds__(kBlock) void core_prefix_kernel(const unsigned int* __restrict__ H, int K, int Wr,
                                                             unsigned short* __restrict__ pre, unsigned int* __restrict__ rowcnt /* [K + 1] */,
                                                             const Index* __restrict__ ptr, const Index* __restrict__ rows,
                                                             unsigned int* __restrict__ tcnt /* [K + 1], nullable: entries outside the core */) {
  const int lane = threadIdx.x & (kWave - 1);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r <= K; r += nw) {
    if (r == K) { if (lane == 0) { rowcnt[K] = 0u; if (tcnt) tcnt[K] = 0u; } continue; }
    unsigned int run = 0u;
    for (int w0 = 0; w0 < Wr; w0 += kWave) {
      const int w = w0 + lane;
      const unsigned int c = w < Wr ? (unsigned int)__popc(H[(size_t)r * Wr + w]) : 0u;
      const unsigned int incl = wave_incl_scan_u32(c);
      if (w < Wr) pre[(size_t)r * Wr + w] = (unsigned short)(run + incl - c);
      run += (unsigned int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
    }
    if (lane == 0) {
      rowcnt[r] = run;
      if (tcnt) tcnt[r] = (unsigned int)(ptr[rows[r] + 1] - ptr[rows[r]]) - run;
    }
  }
}
// mask entries per 128 x 128 tile (bi >= bj), tile t = bi (bi + 1) / 2 + bj
__global__ __launch_bounds__(kBlock) void core_tile_count_kernel(const unsigned int* __restrict__ H, int K, int Wr, int nt,
                                                                 unsigned int* __restrict__ tile_cnt) {
  const int ntile = nt * (nt + 1) / 2;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    int bi = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int bj = t - bi * (bi + 1) / 2;
    unsigned int c = 0u;
    for (int q = threadIdx.x; q < kCoreTile * (kCoreTile / 32); q += blockDim.x) {
      const int row = bi * kCoreTile + (q >> 2), w = bj * (kCoreTile / 32) + (q & 3);
      if (row < K) c += (unsigned int)__popc(H[(size_t)row * Wr + w]);
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(&tile_cnt[t], c);
  }
}

struct CoreTile { unsigned short bi, bj; };

// where the result of entry (row i, column j) goes
__device__ inline unsigned int core_out_pos(const unsigned int* __restrict__ H, const unsigned short* __restrict__ pre,
                                            const unsigned int* __restrict__ rowstart, int Wr, int i, int j) {
  const size_t at = (size_t)i * Wr + (j >> 5);
  return rowstart[i] + (unsigned int)pre[at] + (unsigned int)__popc(H[at] & ((1u << (j & 31)) - 1u));
}

// stages `nrows` bit rows starting at row0, words [w0, w0 + kCoreChunkW), into s[row][kCorePitch]
__device__ inline void core_stage(unsigned int* s, const unsigned int* __restrict__ H, int K, int Wr, int row0, int w0, int tid) {
  for (int q = tid; q < kCoreTile * kCoreChunkW; q += 256) {
    const int r = q / kCoreChunkW, w = q % kCoreChunkW;
    const int row = row0 + r;
    s[r * kCorePitch + w] = (row < K && w0 + w < Wr) ? H[(size_t)row * Wr + w0 + w] : 0u;
  }
}

// ---- popcount: a thread per mask entry of the tile, 16 entries per thread at a time
__global__ __launch_bounds__(256) void core_popc_kernel(const unsigned int* __restrict__ H, const unsigned short* __restrict__ pre,
                                                        const unsigned int* __restrict__ rowstart, int K, int Wr,
                                                        const CoreTile* __restrict__ tiles, int ntiles, int* __restrict__ out,
                                                        unsigned long long* __restrict__ total) {
  __shared__ unsigned int sA[kCoreTile * kCorePitch], sB[kCoreTile * kCorePitch];
  __shared__ unsigned int sM[kCoreTile][kCoreTile / 32];
  __shared__ unsigned int sStart[kCoreTile + 1];
  __shared__ unsigned short sEnt[kCoreBatch];
  const int tid = threadIdx.x;
  unsigned long long mine = 0ull;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int bi = tiles[t].bi, bj = tiles[t].bj;
    const int row0 = bi * kCoreTile, col0 = bj * kCoreTile;
    __syncthreads();
    for (int q = tid; q < kCoreTile * (kCoreTile / 32); q += 256) {
      const int row = row0 + (q >> 2);
      sM[q >> 2][q & 3] = row < K ? H[(size_t)row * Wr + (col0 >> 5) + (q & 3)] : 0u;
    }
    __syncthreads();
    if (tid < kWave) {                                       // the tile's entries numbered row-major: row starts by one wave
      unsigned int run = 0u;
      for (int r0 = 0; r0 < kCoreTile; r0 += kWave) {
        const int r = r0 + tid;
        const unsigned int c = (unsigned int)(__popc(sM[r][0]) + __popc(sM[r][1]) + __popc(sM[r][2]) + __popc(sM[r][3]));
        const unsigned int incl = wave_incl_scan_u32(c);
        sStart[r] = run + incl - c;
        run += (unsigned int)__builtin_amdgcn_readlane((int)incl, kWave - 1);
      }
      if (tid == 0) sStart[kCoreTile] = run;
    }
    __syncthreads();
    const int E = (int)sStart[kCoreTile];
    const int nchunk = (col0 + kCoreTile + 32 * kCoreChunkW - 1) / (32 * kCoreChunkW);   // bits of row j stand below column j
    for (int e0 = 0; e0 < E; e0 += kCoreBatch) {
      __syncthreads();
      if (tid < kCoreTile) {                                 // this batch's entries, (row << 8 | column) inside the tile
        int at = (int)sStart[tid] - e0;
#pragma unroll
        for (int w = 0; w < kCoreTile / 32; ++w)
          for (unsigned int b = sM[tid][w]; b; b &= b - 1u, ++at)
            if (at >= 0 && at < kCoreBatch) sEnt[at] = (unsigned short)((tid << 8) | (w * 32 + __ffs((int)b) - 1));
      }
      __syncthreads();
      const int nb = E - e0 < kCoreBatch ? E - e0 : kCoreBatch;
      unsigned int acc[kCoreBatch / 256];
#pragma unroll
      for (int q = 0; q < kCoreBatch / 256; ++q) acc[q] = 0u;
      for (int c = 0; c < nchunk; ++c) {
        __syncthreads();
        core_stage(sA, H, K, Wr, row0, c * kCoreChunkW, tid);
        core_stage(sB, H, K, Wr, col0, c * kCoreChunkW, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kCoreBatch / 256; ++q) {
          const int e = q * 256 + tid;
          if (e < nb) {
            const unsigned int ij = sEnt[e];
            const unsigned int* a = &sA[(ij >> 8) * kCorePitch];
            const unsigned int* b = &sB[(ij & 255u) * kCorePitch];
            unsigned int s = acc[q];
#pragma unroll
            for (int w = 0; w < kCoreChunkW; ++w) s += (unsigned int)__popc(a[w] & b[w]);
            acc[q] = s;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kCoreBatch / 256; ++q) {
        const int e = q * 256 + tid;
        if (e < nb) {
          const unsigned int ij = sEnt[e];
          out[core_out_pos(H, pre, rowstart, Wr, row0 + (int)(ij >> 8), col0 + (int)(ij & 255u))] = (int)acc[q];
          mine += acc[q];
        }
      }
    }
  }
  mine = wave_sum_u64(mine);
  if ((tid & (kWave - 1)) == 0 && mine) atomicAdd(total, mine);
}

// ---- MFMA: sixteen 0/1 bytes from sixteen bits (four bits to four bytes by one multiply: n + (n << 7) + (n << 14) +
// (n << 21) puts bit b of the nibble at bit 8 b)
__device__ __forceinline__ CoreV4 core_expand16(unsigned int h) {
  CoreV4 d;
  d.x = (int)(((h & 0xfu) * 0x00204081u) & 0x01010101u);
  d.y = (int)((((h >> 4) & 0xfu) * 0x00204081u) & 0x01010101u);
  d.z = (int)((((h >> 8) & 0xfu) * 0x00204081u) & 0x01010101u);
  d.w = (int)((((h >> 12) & 0xfu) * 0x00204081u) & 0x01010101u);
  return d;
}
// v_mfma_i32_16x16x64_i8: lane l holds, of A (16 x 64), row l % 16 and the sixteen values k = 16 (l / 16) .. + 15; of B
// (64 x 16) column l % 16 and the same sixteen k; of C / D (16 x 16) column l % 16 and rows 4 (l / 16) .. + 3
// (tests/test_gpu_mxm.py::test_dense_core checks the kernel built on this against numpy).
__global__ __launch_bounds__(256) void core_mfma_kernel(const unsigned int* __restrict__ H, const unsigned short* __restrict__ pre,
                                                        const unsigned int* __restrict__ rowstart, int K, int Wr,
                                                        const CoreTile* __restrict__ tiles, int ntiles, int* __restrict__ out,
                                                        unsigned long long* __restrict__ total) {
  __shared__ unsigned int sA[kCoreTile * kCorePitch], sB[kCoreTile * kCorePitch];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;                   // this wave's 64 x 64 quarter of the tile
  const int l16 = lane & 15, kq = lane >> 4;                 // row / column inside a 16 x 16 block; which sixteen of the 64 k
  unsigned long long mine = 0ull;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int bi = tiles[t].bi, bj = tiles[t].bj;
    const int row0 = bi * kCoreTile, col0 = bj * kCoreTile;
    const int nchunk = (col0 + kCoreTile + 32 * kCoreChunkW - 1) / (32 * kCoreChunkW);
    CoreV4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = CoreV4{0, 0, 0, 0};
    for (int c = 0; c < nchunk; ++c) {
      __syncthreads();
      core_stage(sA, H, K, Wr, row0, c * kCoreChunkW, tid);
      core_stage(sB, H, K, Wr, col0, c * kCoreChunkW, tid);
      __syncthreads();
#pragma unroll 2
      for (int ks = 0; ks < kCoreChunkW / 2; ++ks) {         // 64 values of k per step: two words of every row
        const int w = 2 * ks + (kq >> 1), sh = 16 * (kq & 1);
        CoreV4 fa[4], fb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = core_expand16((sA[(wi * 64 + a * 16 + l16) * kCorePitch + w] >> sh) & 0xffffu);
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = core_expand16((sB[(wj * 64 + b * 16 + l16) * kCorePitch + w] >> sh) & 0xffffu);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    // the mask: only the pairs that are entries of L are results
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = col0 + wj * 64 + b * 16 + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + wi * 64 + a * 16 + 4 * kq + r;
          if (i < K && j < K) {
            const size_t at = (size_t)i * Wr + (j >> 5);
            const unsigned int word = H[at];
            if ((word >> (j & 31)) & 1u) {
              const int val = acc[a][b][r];
              out[rowstart[i] + (unsigned int)pre[at] + (unsigned int)__popc(word & ((1u << (j & 31)) - 1u))] = val;
              mine += (unsigned long long)(unsigned int)val;
            }
          }
        }
      }
  }
  mine = wave_sum_u64(mine);
  if (lane == 0 && mine) atomicAdd(total, mine);
}

// position-weighted sum of the per-entry results: equal for two runs only if (almost surely) every entry is
__global__ __launch_bounds__(kBlock) void core_checksum_kernel(const int* __restrict__ out, long long n, unsigned long long* __restrict__ sum) {
  unsigned long long s = 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += (unsigned long long)(unsigned int)out[i] * (unsigned long long)((unsigned int)i * 2654435761u | 1u);
  s = wave_sum_u64(s);
  if ((threadIdx.x & (kWave - 1)) == 0 && s) atomicAdd(sum, s);
}


// The second walk over the core rows' lists, when the core is one part of a whole product (mxm.hip): an entry that is a
// core vertex is a MASK ENTRY BETWEEN CORE ROWS -- its result comes from the bit rows, so it is noted (where it sits in
// the mask, its column, its place in the column's list) and its copy of the mask value is zeroed, which takes it out of
// the pivot kernels' passes over the whole mask; every other entry goes to the row's T-list (the list without the core
// vertices, in order), against which the same pivot kernels intersect the T-lists of the entries noted here.
__global__ __launch_bounds__(kBlock) void core_split_kernel(const Index* __restrict__ ptr, const Index* __restrict__ ind,
                                                            const Index* __restrict__ rows, int K, const int* __restrict__ rank,
                                                            const unsigned int* __restrict__ H, const unsigned short* __restrict__ pre,
                                                            const unsigned int* __restrict__ rowstart, int Wr,
                                                            Index* __restrict__ pos, Index* __restrict__ ccind,
                                                            const unsigned int* __restrict__ tptr, Index* __restrict__ tind,
                                                            const unsigned int* __restrict__ cscptr, unsigned int* __restrict__ cscfill,
                                                            Index* __restrict__ cscind, unsigned int* __restrict__ mval2) {
  const int lane = threadIdx.x & (kWave - 1);
  const int nw = gridDim.x * (blockDim.x >> 6);
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < K; r += nw) {
    const Index x = rows[r];
    const Index s0 = ptr[x], e = ptr[x + 1];
    unsigned int tat = tptr[r];
    for (Index p0 = s0; p0 < e; p0 += kWave) {
      const Index p = p0 + lane;
      const bool have = p < e;
      const Index col = have ? ind[p] : 0;
      const int c = have ? rank[col] : -1;
      const bool isT = have && c < 0;
      const unsigned long long tm = __ballot(isT);
      if (isT) tind[tat + (unsigned int)__popcll(tm & lt)] = col;
      tat += (unsigned int)__popcll(tm);
      if (have && c >= 0) {
        const unsigned int en = core_out_pos(H, pre, rowstart, Wr, r, c);
        pos[en] = p;
        ccind[en] = (Index)c;
        cscind[cscptr[c] + atomicAdd(&cscfill[c], 1u)] = (Index)r;
        if (mval2) mval2[p] = 0u;
      }
    }
  }
}
// C at a core entry = (hits among the core's columns + hits outside them) x the one product
template <typename T>
__global__ __launch_bounds__(kBlock) void core_combine_kernel(T* __restrict__ c_val, const Index* __restrict__ pos, const int* __restrict__ ch,
                                                              const T* __restrict__ ct, unsigned int nent, T one,
                                                              const void* __restrict__ m_val, int mask_f32) {
  for (unsigned int en = blockIdx.x * blockDim.x + threadIdx.x; en < nent; en += gridDim.x * blockDim.x) {
    const Index p = pos[en];
    if (!mask_nonzero(m_val, mask_f32, p)) continue;        // (an entry the mask's own value switches off stays the identity)
    c_val[p] = (T)ch[en] * one + ct[en];
  }
}

}  // namespace grb

using namespace grb;

TcCoreDev::~TcCoreDev() { for (void* q : owned) (void)hipFree(q); }
static grb_info core_malloc(TcCoreDev* d, void** p, size_t bytes);
grb_info grb::tc_core_alloc(TcCoreDev* d, void** p, size_t bytes) { return core_malloc(d, p, bytes); }
static grb_info core_malloc(TcCoreDev* d, void** p, size_t bytes) {
  GRB_HIP_TRY(hipMalloc(p, bytes ? bytes : 4));
  d->owned.push_back(*p);
  return GRB_SUCCESS;
}

// the core rows: the k_want longest (all rows of a length are taken or none: theta is a length), ranked in vertex order
grb_info grb::tc_core_rows(const Index* ptr, Index n, int k_want, TcCoreDev* d) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  d->K = 0; d->n = n;
  if (k_want > 65535) k_want = 65535;                       // (the per-word prefix counts are 16-bit)
  unsigned int* d_hist = nullptr;
  GRB_TRY(core_malloc(d, (void**)&d_hist, 4 * 16384));
  GRB_HIP_TRY(hipMemsetAsync(d_hist, 0, 4 * 16384, s));
  hipLaunchKernelGGL(core_len_hist_kernel, dim3(c.num_cu), dim3(1024), 0, s, ptr, n, d_hist);
  GRB_HIP_TRY(hipGetLastError());
  std::vector<unsigned int> hist(16384);
  GRB_HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, 4 * 16384, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  Index theta = 16383;
  long long have = hist[16383];
  if (have > k_want) return GRB_NOT_IMPLEMENTED;             // more than k_want rows beyond the histogram's last bin
  while (theta > 1 && have + (long long)hist[theta - 1] <= (long long)k_want) { --theta; have += hist[theta]; }
  d->K = (int)have;
  d->theta = theta;
  if (d->K < 2) return GRB_SUCCESS;
  unsigned int* d_flag = nullptr;
  GRB_TRY(core_malloc(d, (void**)&d_flag, 4 * ((size_t)n + 1)));
  GRB_TRY(core_malloc(d, (void**)&d->rank, 4 * (size_t)n));
  GRB_TRY(core_malloc(d, (void**)&d->rows, 4 * (size_t)d->K));
  hipLaunchKernelGGL(core_flag_kernel, dim3(stream_grid((long long)n + 1)), dim3(kBlock), 0, s, ptr, n, theta, d_flag);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_exclusive_scan_u32(d_flag, (long long)n + 1));
  hipLaunchKernelGGL(core_rank_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, ptr, n, theta, (const unsigned int*)d_flag, d->rank, d->rows);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}
// the bit rows (padded to whole staging chunks), the per-word prefix counts, the rows' first result; with split: the
// entries outside the core per row and the entries per core column as well
grb_info grb::tc_core_bits(const Index* ptr, const Index* ind, TcCoreDev* d, bool split) {
  hipStream_t s = ctx().stream;
  const int K = d->K;
  d->nt = (K + kCoreTile - 1) / kCoreTile;
  d->Wr = ((d->nt * kCoreTile / 32 + kCoreChunkW - 1) / kCoreChunkW) * kCoreChunkW;
  const int Wr = d->Wr;
  GRB_TRY(core_malloc(d, (void**)&d->H, 4 * (size_t)K * Wr));
  GRB_TRY(core_malloc(d, (void**)&d->pre, 2 * (size_t)K * Wr));
  GRB_TRY(core_malloc(d, (void**)&d->rowstart, 4 * ((size_t)K + 1)));
  if (split) {
    GRB_TRY(core_malloc(d, (void**)&d->tptr, 4 * ((size_t)K + 1)));
    GRB_TRY(core_malloc(d, (void**)&d->cscptr, 4 * ((size_t)K + 1)));
    GRB_HIP_TRY(hipMemsetAsync(d->cscptr, 0, 4 * ((size_t)K + 1), s));
  }
  GRB_HIP_TRY(hipMemsetAsync(d->H, 0, 4 * (size_t)K * Wr, s));
  hipLaunchKernelGGL(core_fill_kernel, dim3(stream_grid((long long)K * kWave)), dim3(kBlock), 0, s, ptr, ind, (const Index*)d->rows, K,
                     (const int*)d->rank, d->H, Wr, split ? d->cscptr : nullptr);
  hipLaunchKernelGGL(core_prefix_kernel, dim3(stream_grid((long long)(K + 1) * kWave)), dim3(kBlock), 0, s, (const unsigned int*)d->H, K, Wr,
                     d->pre, d->rowstart, ptr, (const Index*)d->rows, split ? d->tptr : nullptr);
  GRB_HIP_TRY(hipGetLastError());
  GRB_TRY(device_exclusive_scan_u32(d->rowstart, (long long)K + 1));
  if (split) {
    GRB_TRY(device_exclusive_scan_u32(d->tptr, (long long)K + 1));
    GRB_TRY(device_exclusive_scan_u32(d->cscptr, (long long)K + 1));
  }
  GRB_HIP_TRY(hipMemcpyAsync(&d->nent, d->rowstart + K, 4, hipMemcpyDeviceToHost, s));
  if (split) GRB_HIP_TRY(hipMemcpyAsync(&d->nt_elems, d->tptr + K, 4, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  return GRB_SUCCESS;
}
// the tiles: entry counts decide who takes a tile; heaviest (most columns of k) first
grb_info grb::tc_core_tiles(TcCoreDev* d, int method, int dense_from, grb_tc_core_result* res) {
  hipStream_t s = ctx().stream;
  const int nt = d->nt;
  const int ntile_all = nt * (nt + 1) / 2;
  unsigned int* d_tcnt = nullptr;
  GRB_TRY(core_malloc(d, (void**)&d_tcnt, 4 * (size_t)ntile_all));
  GRB_HIP_TRY(hipMemsetAsync(d_tcnt, 0, 4 * (size_t)ntile_all, s));
  hipLaunchKernelGGL(core_tile_count_kernel, dim3(ntile_all < 4096 ? ntile_all : 4096), dim3(kBlock), 0, s, (const unsigned int*)d->H, d->K, d->Wr,
                     nt, d_tcnt);
  GRB_HIP_TRY(hipGetLastError());
  std::vector<unsigned int> tcnt((size_t)ntile_all);
  GRB_HIP_TRY(hipMemcpyAsync(tcnt.data(), d_tcnt, 4 * (size_t)ntile_all, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  std::vector<CoreTile> t_popc, t_mfma;
  const unsigned int from = method == 0 ? 0xffffffffu : method == 1 ? 1u : (unsigned int)(dense_from > 0 ? dense_from : 1);
  for (int bi = nt - 1; bi >= 0; --bi)
    for (int bj = bi; bj >= 0; --bj) {                       // (columns of k a tile needs grow with bj)
      const unsigned int e = tcnt[(size_t)bi * (bi + 1) / 2 + bj];
      if (!e) continue;
      if (res) {
        ++res->tiles;
        const int decile = (int)((unsigned long long)e * 10ull / (unsigned long long)(kCoreTile * kCoreTile));
        ++res->tiles_by_density[decile > 9 ? 9 : decile];
      }
      (e >= from ? t_mfma : t_popc).push_back(CoreTile{(unsigned short)bi, (unsigned short)bj});
    }
  auto by_work = [](const CoreTile& x, const CoreTile& y) { return x.bj > y.bj; };
  std::stable_sort(t_popc.begin(), t_popc.end(), by_work);
  std::stable_sort(t_mfma.begin(), t_mfma.end(), by_work);
  d->n_popc = (int)t_popc.size();
  d->n_mfma = (int)t_mfma.size();
  if (res) res->tiles_mfma = d->n_mfma;
  GRB_TRY(core_malloc(d, &d->tiles_popc, sizeof(CoreTile) * t_popc.size()));
  GRB_TRY(core_malloc(d, &d->tiles_mfma, sizeof(CoreTile) * t_mfma.size()));
  // (pageable host vectors: the copies complete before the calls return)
  if (!t_popc.empty()) GRB_HIP_TRY(hipMemcpy(d->tiles_popc, t_popc.data(), sizeof(CoreTile) * t_popc.size(), hipMemcpyHostToDevice));
  if (!t_mfma.empty()) GRB_HIP_TRY(hipMemcpy(d->tiles_mfma, t_mfma.data(), sizeof(CoreTile) * t_mfma.size(), hipMemcpyHostToDevice));
  return GRB_SUCCESS;
}
// out[e] = sum_k H[i][k] H[j][k] for every entry e = (i, j) between core rows; *total += their sum
grb_info grb::tc_core_hproduct(const TcCoreDev* d, int* out, unsigned long long* total) {
  Context& c = ctx();
  hipStream_t s = c.stream;
  const int wg = 2 * c.num_cu;
  if (d->n_popc) {
    hipLaunchKernelGGL(core_popc_kernel, dim3(d->n_popc < wg ? d->n_popc : wg), dim3(256), 0, s, (const unsigned int*)d->H,
                       (const unsigned short*)d->pre, (const unsigned int*)d->rowstart, d->K, d->Wr, (const CoreTile*)d->tiles_popc, d->n_popc, out,
                       total);
    GRB_HIP_TRY(hipGetLastError());
  }
  if (d->n_mfma) {
    hipLaunchKernelGGL(core_mfma_kernel, dim3(d->n_mfma < wg ? d->n_mfma : wg), dim3(256), 0, s, (const unsigned int*)d->H,
                       (const unsigned short*)d->pre, (const unsigned int*)d->rowstart, d->K, d->Wr, (const CoreTile*)d->tiles_mfma, d->n_mfma, out,
                       total);
    GRB_HIP_TRY(hipGetLastError());
  }
  return GRB_SUCCESS;
}
// the entries between core rows as a mask of their own (CSR: rowstart / ccind, CSC: cscptr / cscind), where each sits in
// the whole mask (pos), the core rows' lists without the core vertices (tptr / tind); mval2 (nullable): a copy of the
// whole mask's values in which those entries are zeroed
grb_info grb::tc_core_split(const Index* ptr, const Index* ind, TcCoreDev* d, unsigned int* mval2) {
  hipStream_t s = ctx().stream;
  const int K = d->K;
  unsigned int* d_fill = nullptr;
  GRB_TRY(core_malloc(d, (void**)&d->pos, 4 * (size_t)d->nent));
  GRB_TRY(core_malloc(d, (void**)&d->ccind, 4 * (size_t)d->nent));
  GRB_TRY(core_malloc(d, (void**)&d->cscind, 4 * (size_t)d->nent));
  GRB_TRY(core_malloc(d, (void**)&d->tind, 4 * (size_t)d->nt_elems));
  GRB_TRY(core_malloc(d, (void**)&d_fill, 4 * ((size_t)K + 1)));
  GRB_HIP_TRY(hipMemsetAsync(d_fill, 0, 4 * ((size_t)K + 1), s));
  hipLaunchKernelGGL(core_split_kernel, dim3(stream_grid((long long)K * kWave)), dim3(kBlock), 0, s, ptr, ind, (const Index*)d->rows, K,
                     (const int*)d->rank, (const unsigned int*)d->H, (const unsigned short*)d->pre, (const unsigned int*)d->rowstart, d->Wr, d->pos,
                     d->ccind, (const unsigned int*)d->tptr, d->tind, (const unsigned int*)d->cscptr, d_fill, d->cscind, mval2);
  GRB_HIP_TRY(hipGetLastError());
  d->h_mptr.resize((size_t)K + 1); d->h_tptr.resize((size_t)K + 1); d->h_cscptr.resize((size_t)K + 1);
  GRB_HIP_TRY(hipMemcpyAsync(d->h_mptr.data(), d->rowstart, 4 * ((size_t)K + 1), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(d->h_tptr.data(), d->tptr, 4 * ((size_t)K + 1), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipMemcpyAsync(d->h_cscptr.data(), d->cscptr, 4 * ((size_t)K + 1), hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  return GRB_SUCCESS;
}
grb_info grb::tc_core_combine(int dtype, void* c_val, const TcCoreDev* d, const int* ch, const void* ct, unsigned int one_bits, const void* m_val,
                              int mask_f32) {
  hipStream_t s = ctx().stream;
  if (!d->nent) return GRB_SUCCESS;
  if (dtype != GRB_I32) return GRB_NOT_IMPLEMENTED;
  int one;
  memcpy(&one, &one_bits, 4);
  hipLaunchKernelGGL((core_combine_kernel<int>), dim3(stream_grid((long long)d->nent)), dim3(kBlock), 0, s, (int*)c_val, (const Index*)d->pos, ch,
                     (const int*)ct, d->nent, one, m_val, mask_f32);
  GRB_HIP_TRY(hipGetLastError());
  return GRB_SUCCESS;
}

extern "C" grb_info grb_tc_dense_core(grb_matrix L, int k_want, int method, int dense_from, grb_tc_core_result* res) { GRB_API_ENTER();
  if (!L || !res) return GRB_NULL_POINTER;
  if (!L->built || !L->csr.ptr) return GRB_UNINITIALIZED_OBJECT;
  if (L->nrows != L->ncols) return GRB_DIMENSION_MISMATCH;
  if (k_want < 1 || method < 0 || method > 2) return GRB_INVALID_VALUE;
  memset(res, 0, sizeof(*res));
  GRB_TRY(ctx_init());
  hipStream_t s = ctx().stream;
  hipEvent_t ev[3];
  for (auto& e : ev) GRB_HIP_TRY(hipEventCreate(&e));
  struct EvFree { hipEvent_t* e; ~EvFree() { for (int i = 0; i < 3; ++i) (void)hipEventDestroy(e[i]); } } ev_free{ev};
  TcCoreDev d;
  GRB_HIP_TRY(hipEventRecord(ev[0], s));
  GRB_TRY(tc_core_rows(L->csr.ptr, L->nrows, k_want, &d));
  res->core_rows = d.K;
  res->min_row_length = d.theta;
  if (d.K < 2) return GRB_SUCCESS;
  GRB_TRY(tc_core_bits(L->csr.ptr, L->csr.ind, &d, false));
  res->core_entries = (int64_t)d.nent;
  GRB_TRY(tc_core_tiles(&d, method, dense_from, res));
  int* d_out = nullptr;
  unsigned long long* d_tot = nullptr;
  GRB_TRY(core_malloc(&d, (void**)&d_out, 4 * (size_t)d.nent));
  GRB_TRY(core_malloc(&d, (void**)&d_tot, 16));
  GRB_HIP_TRY(hipMemsetAsync(d_out, 0xff, 4 * (size_t)d.nent, s));   // (an entry nobody wrote shows in the checksum)
  GRB_HIP_TRY(hipMemsetAsync(d_tot, 0, 16, s));
  GRB_HIP_TRY(hipEventRecord(ev[1], s));
  GRB_TRY(tc_core_hproduct(&d, d_out, d_tot));
  GRB_HIP_TRY(hipEventRecord(ev[2], s));
  if (d.nent) {
    hipLaunchKernelGGL(core_checksum_kernel, dim3(stream_grid((long long)d.nent)), dim3(kBlock), 0, s, (const int*)d_out, (long long)d.nent, d_tot + 1);
    GRB_HIP_TRY(hipGetLastError());
  }
  unsigned long long tot[2] = {0ull, 0ull};
  GRB_HIP_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, s));
  GRB_HIP_TRY(hipStreamSynchronize(s));
  res->count = (int64_t)tot[0];
  res->checksum = tot[1];
  GRB_HIP_TRY(hipEventElapsedTime(&res->build_ms, ev[0], ev[1]));
  GRB_HIP_TRY(hipEventElapsedTime(&res->product_ms, ev[1], ev[2]));
  return GRB_SUCCESS;
}
