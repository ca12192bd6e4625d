// mtx.hip -- MatrixMarket coordinate files parsed on the device (the last host-side
// piece of ingest).  The reference's readMtx (graphblas/util.hpp:363-430) reads the banner and the
// size line with mmio and then fscanf's `nvals` tuples (readTuples, util.hpp:197-258): token by token,
// "%d %d" for pattern files, "%d %d %d" / "%d %d %f" for integer / real ones, 1-based indices, then
// removeSelfloop + customSort.  Here the host reads the file, parses banner / comments / size line (a
// few hundred bytes) and hands the rest to three kernels: count the tokens per block, list their
// offsets, parse entry e from tokens [k e, k e + k).  The coordinate list goes straight into the device
// ingest of build.hip (reverse edges for symmetric files, self loops and duplicates dropped).
//
// Values: integer files parse exactly; real files parse up to 19 significant digits into a double and
// scale by a power of ten, which can differ from fscanf's correctly rounded float in the last bit for
// long decimal strings.  Where the loader drops entries of a VALUED file the reference leaves the
// values array unshifted (util.hpp:311-323, a bug it flags itself); this loader keeps each surviving
// entry's own value.
#include <cstdio>
#include <string>
#include <vector>

#include "common.hpp"

namespace grb {

__device__ __forceinline__ bool is_space(unsigned char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

// tokens starting in [chunk begin, chunk end): a non-space byte whose predecessor is a space (or file start)
__global__ __launch_bounds__(kBlock) void mtx_count_tokens_kernel(const unsigned char* __restrict__ text, long long n,
                                                                 long long chunk, int* __restrict__ counts) {
  __shared__ int smem[kWavesPerBlock];
  const long long b = (long long)blockIdx.x * chunk, e = b + chunk < n ? b + chunk : n;
  int c = 0;
  for (long long i = b + threadIdx.x; i < e; i += kBlock)
    c += (!is_space(text[i]) && (i == 0 || is_space(text[i - 1]))) ? 1 : 0;
  c = wave_reduce(c, [](int a, int d) { return a + d; });
  if (lane_id() == 0) smem[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w];
    counts[blockIdx.x] = t;
  }
}

// ordered list of token offsets; a block walks its chunk in 256-byte steps (ballot-free block scan)
__global__ __launch_bounds__(kBlock) void mtx_list_tokens_kernel(const unsigned char* __restrict__ text, long long n,
                                                                long long chunk, const long long* __restrict__ block_off,
                                                                long long* __restrict__ token_off) {
  __shared__ int smem[kWavesPerBlock];
  const long long b = (long long)blockIdx.x * chunk, e = b + chunk < n ? b + chunk : n;
  long long base = block_off[blockIdx.x];
  for (long long s = b; s < e; s += kBlock) {
    const long long i = s + threadIdx.x;
    const int is_start = (i < e && !is_space(text[i]) && (i == 0 || is_space(text[i - 1]))) ? 1 : 0;
    int total;
    const int pos = block_exclusive_scan(is_start, smem, total);
    if (is_start) token_off[base + pos] = i;
    base += total;
  }
}

__device__ inline long long parse_int(const unsigned char* __restrict__ t, long long i, long long n) {
  bool neg = false;
  if (i < n && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; ++i; }
  long long v = 0;
  while (i < n && t[i] >= '0' && t[i] <= '9') { v = v * 10 + (t[i] - '0'); ++i; }
  return neg ? -v : v;
}

__device__ inline double parse_real(const unsigned char* __restrict__ t, long long i, long long n) {
  bool neg = false;
  if (i < n && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; ++i; }
  // "%f" accepts inf / infinity / nan in any case (C11 7.22.1.3): parse them as such, not as 0
  if (i + 2 < n) {
    const unsigned char a = t[i] | 0x20, b = t[i + 1] | 0x20, c = t[i + 2] | 0x20;
    if (a == 'i' && b == 'n' && c == 'f') return neg ? -__builtin_huge_val() : __builtin_huge_val();
    if (a == 'n' && b == 'a' && c == 'n') return __builtin_nan("");
  }
  unsigned long long mant = 0;
  int digits = 0, exp10 = 0;
  bool seen_point = false;
  for (; i < n; ++i) {
    const unsigned char c = t[i];
    if (c >= '0' && c <= '9') {
      if (digits < 19) { mant = mant * 10 + (c - '0'); if (mant) ++digits; if (seen_point) --exp10; }
      else if (!seen_point) ++exp10;
    } else if (c == '.' && !seen_point) {
      seen_point = true;
    } else {
      break;
    }
  }
  if (i < n && (t[i] == 'e' || t[i] == 'E')) exp10 += (int)parse_int(t, i + 1, n);
  double v = (double)mant;
  double p = 10.0;
  int k = exp10 < 0 ? -exp10 : exp10;
  double scale = 1.0;
  while (k) { if (k & 1) scale *= p; p *= p; k >>= 1; }
  v = exp10 < 0 ? v / scale : v * scale;
  return neg ? -v : v;
}

// entry e = tokens [k e, k e + k): row, col (1-based in the file), value.  kind: 0 pattern, 1 integer, 2 real
template <typename T>
__global__ void mtx_parse_entries_kernel(const unsigned char* __restrict__ text, long long n,
                                         const long long* __restrict__ token_off, long long nentries, int kind,
                                         Index* __restrict__ rows, Index* __restrict__ cols, T* __restrict__ vals) {
  const int k = kind == 0 ? 2 : 3;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < nentries; e += (long long)gridDim.x * blockDim.x) {
    rows[e] = (Index)(parse_int(text, token_off[k * e], n) - 1);
    cols[e] = (Index)(parse_int(text, token_off[k * e + 1], n) - 1);
    if (kind == 0) vals[e] = (T)1;
    else if (kind == 1) vals[e] = (T)(int)parse_int(text, token_off[k * e + 2], n);        // "%d" then cast
    else vals[e] = (T)(float)parse_real(text, token_off[k * e + 2], n);                    // "%f" then cast
  }
}

struct MtxHeader {
  long long data_begin = 0;
  Index nrows = 0, ncols = 0;
  long long nvals = 0;
  int kind = 0;          // 0 pattern, 1 integer, 2 real
  bool symmetric = false;
};

// mm_read_banner + mm_read_mtx_crd_size (graphblas/mmio.hpp) on the in-memory text
static grb_info parse_header(const std::vector<unsigned char>& text, MtxHeader* h) {
  size_t pos = 0;
  auto line = [&](std::string* out) -> bool {
    if (pos >= text.size()) return false;
    size_t e = pos;
    while (e < text.size() && text[e] != '\n') ++e;
    out->assign((const char*)&text[pos], e - pos);
    pos = e < text.size() ? e + 1 : e;
    return true;
  };
  std::string l;
  if (!line(&l)) return GRB_INVALID_VALUE;
  char banner[64], mtx[64], crd[64], dt[64], st[64];
  if (sscanf(l.c_str(), "%63s %63s %63s %63s %63s", banner, mtx, crd, dt, st) != 5) return GRB_INVALID_VALUE;
  auto lower = [](char* s) { for (; *s; ++s) if (*s >= 'A' && *s <= 'Z') *s += 'a' - 'A'; };
  lower(mtx); lower(crd); lower(dt); lower(st);
  if (std::string(banner) != "%%MatrixMarket" || std::string(mtx) != "matrix") return GRB_INVALID_VALUE;
  if (std::string(crd) != "coordinate") return GRB_NOT_IMPLEMENTED;          // readMtx reads tuples only
  const std::string sdt(dt), sst(st);
  if (sdt == "pattern") h->kind = 0; else if (sdt == "integer") h->kind = 1; else if (sdt == "real") h->kind = 2;
  else return GRB_NOT_IMPLEMENTED;                                           // complex: readMtx reads nothing
  if (sst == "general") h->symmetric = false; else if (sst == "symmetric") h->symmetric = true;
  else return GRB_NOT_IMPLEMENTED;
  do {
    if (!line(&l)) return GRB_INVALID_VALUE;
  } while (!l.empty() && l[0] == '%');
  long long r = 0, c = 0, nz = 0;
  while (sscanf(l.c_str(), "%lld %lld %lld", &r, &c, &nz) != 3)
    if (!line(&l)) return GRB_INVALID_VALUE;
  if (r < 0 || c < 0 || nz < 0 || r > 0x7fffffffLL || c > 0x7fffffffLL) return GRB_INVALID_VALUE;
  h->nrows = (Index)r; h->ncols = (Index)c; h->nvals = nz; h->data_begin = (long long)pos;
  return GRB_SUCCESS;
}

}  // namespace grb

using namespace grb;

extern "C" {

// Extension: readMtx + Matrix::build in one call, the text parsed on the device.  directed as readMtx:
// 0 = symmetric iff the banner says so, 1 = force directed, 2 = force undirected.  *A is created here
// (grb_matrix_free it); dims_out (nullable) = {nrows, ncols, nvals after the loader}.
grb_info grb_matrix_load_mtx(grb_matrix* A, const char* path, grb_dtype dtype, int directed, grb_index* dims_out) { GRB_API_ENTER();
  if (!A || !path) return GRB_NULL_POINTER;
  GRB_TRY(ctx_init());
  FILE* f = fopen(path, "rb");
  if (!f) return GRB_INVALID_VALUE;
  std::vector<unsigned char> text;
  fseek(f, 0, SEEK_END);
  const long long fsz = ftell(f);
  fseek(f, 0, SEEK_SET);
  text.resize((size_t)(fsz > 0 ? fsz : 0));
  const size_t got = text.empty() ? 0 : fread(text.data(), 1, text.size(), f);
  fclose(f);
  if (got != text.size()) return GRB_INVALID_VALUE;
  MtxHeader h;
  GRB_TRY(parse_header(text, &h));
  const bool undirected = directed == 1 ? false : (h.symmetric || directed == 2);
  Context& c = ctx();
  hipStream_t s = c.stream;
  const long long n = (long long)text.size() - h.data_begin;
  const int k = h.kind == 0 ? 2 : 3;
  grb_info info = GRB_SUCCESS;
  unsigned char* d_text = nullptr;
  int* d_counts = nullptr;
  long long *d_block_off = nullptr, *d_tok = nullptr;
  Index *d_rows = nullptr, *d_cols = nullptr;
  void* d_vals = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_text); (void)hipFree(d_counts); (void)hipFree(d_block_off); (void)hipFree(d_tok);
    (void)hipFree(d_rows); (void)hipFree(d_cols); (void)hipFree(d_vals);
  };
#define MTX_HIP(call) do { if ((call) != hipSuccess) { cleanup(); return GRB_PANIC; } } while (0)
  long long nentries = 0;
  if (n > 0 && h.nvals > 0) {
    const long long chunk = 1 << 16;
    const int nblocks = (int)((n + chunk - 1) / chunk);
    MTX_HIP(hipMalloc((void**)&d_text, (size_t)n));
    MTX_HIP(hipMemcpyAsync(d_text, text.data() + h.data_begin, (size_t)n, hipMemcpyHostToDevice, s));
    MTX_HIP(hipMalloc((void**)&d_counts, sizeof(int) * (size_t)nblocks));
    MTX_HIP(hipMalloc((void**)&d_block_off, sizeof(long long) * (size_t)nblocks));
    hipLaunchKernelGGL(mtx_count_tokens_kernel, dim3(nblocks), dim3(kBlock), 0, s, d_text, n, chunk, d_counts);
    MTX_HIP(hipGetLastError());
    std::vector<int> counts((size_t)nblocks);
    MTX_HIP(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * (size_t)nblocks, hipMemcpyDeviceToHost, s));
    MTX_HIP(hipStreamSynchronize(s));
    std::vector<long long> off((size_t)nblocks);
    long long ntok = 0;
    for (int b = 0; b < nblocks; ++b) { off[(size_t)b] = ntok; ntok += counts[(size_t)b]; }
    nentries = ntok / k;
    if (nentries > h.nvals) nentries = h.nvals;                     // fscanf stops after nvals tuples
    if (nentries < h.nvals) fprintf(stdout, "Error: Not enough rows in mtx file!\n");   // util.hpp:217-219
    // 32-bit Index like the reference: the count (doubled when reverse entries are added) must fit
    if (nentries > 0x7fffffffll || (undirected && 2 * nentries > 0x7fffffffll)) { cleanup(); return GRB_OUT_OF_MEMORY; }
    if (nentries > 0) {
      MTX_HIP(hipMemcpyAsync(d_block_off, off.data(), sizeof(long long) * (size_t)nblocks, hipMemcpyHostToDevice, s));
      MTX_HIP(hipMalloc((void**)&d_tok, sizeof(long long) * (size_t)(ntok > 0 ? ntok : 1)));
      hipLaunchKernelGGL(mtx_list_tokens_kernel, dim3(nblocks), dim3(kBlock), 0, s, d_text, n, chunk, d_block_off, d_tok);
      MTX_HIP(hipGetLastError());
      MTX_HIP(hipMalloc((void**)&d_rows, sizeof(Index) * (size_t)nentries));
      MTX_HIP(hipMalloc((void**)&d_cols, sizeof(Index) * (size_t)nentries));
      MTX_HIP(hipMalloc(&d_vals, 4 * (size_t)nentries));
      const int grid = stream_grid(nentries);
      if (dtype == GRB_F32)
        hipLaunchKernelGGL((mtx_parse_entries_kernel<float>), dim3(grid), dim3(kBlock), 0, s, d_text, n, d_tok, nentries, h.kind,
                           d_rows, d_cols, (float*)d_vals);
      else
        hipLaunchKernelGGL((mtx_parse_entries_kernel<int>), dim3(grid), dim3(kBlock), 0, s, d_text, n, d_tok, nentries, h.kind,
                           d_rows, d_cols, (int*)d_vals);
      MTX_HIP(hipGetLastError());
    }
  }
#undef MTX_HIP
  info = grb_matrix_new(A, dtype, h.nrows, h.ncols);
  if (info == GRB_SUCCESS) {
    // removeSelfloop: reverse edges when undirected, self loops (GRB_UTIL_REMOVE_SELFLOOP, default on)
    // and duplicates dropped (util.hpp:263-329)
    const char* env = getenv("GRB_UTIL_REMOVE_SELFLOOP");
    const int flags = (undirected ? 1 : 0) | ((!env || atoi(env) != 0) ? 2 : 0) | 4;
    info = grb_matrix_ingest_device(*A, d_rows, d_cols, d_vals, (grb_index)nentries, flags);
    if (info != GRB_SUCCESS) { grb_matrix_free(*A); *A = nullptr; }
  }
  if (hipStreamSynchronize(s) != hipSuccess && info == GRB_SUCCESS) info = GRB_PANIC;
  cleanup();
  if (info == GRB_SUCCESS && dims_out) { dims_out[0] = h.nrows; dims_out[1] = h.ncols; dims_out[2] = (*A)->nvals; }
  return info;
}

}  // extern "C"
