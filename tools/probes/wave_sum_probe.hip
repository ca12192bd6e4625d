// Checks the DPP wave primitives of common.hpp (wave_sum_u32, wave_incl_scan_u32, wave_sum_u64, wave_min_u32, wave_max_u32,
// wave_or_u32 / _u64: row shifts inside the 16-lane rows + two row broadcasts) against serial loops for random inputs,
// including zeros and values near 2^32.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I graphblast_amd/csrc -I include
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace grb;
struct Out { unsigned sum, mn, mx, orr; unsigned long long sum64, or64; };
__global__ void k(const unsigned* in, const unsigned long long* in64, Out* out, unsigned* scan) {
  const unsigned v = in[blockIdx.x * 64 + threadIdx.x];
  const unsigned long long w = in64[blockIdx.x * 64 + threadIdx.x];
  Out o;
  o.sum = wave_sum_u32(v); o.mn = wave_min_u32(v); o.mx = wave_max_u32(v); o.orr = wave_or_u32(v);
  o.sum64 = wave_sum_u64(w); o.or64 = wave_or_u64(w);
  scan[blockIdx.x * 64 + threadIdx.x] = wave_incl_scan_u32(v);
  if (threadIdx.x == (blockIdx.x % 64)) out[blockIdx.x] = o;       // any lane: the results are wave-uniform
}
int main() {
  const int nb = 4096;
  std::vector<unsigned> h(nb * 64), scan(nb * 64);
  std::vector<unsigned long long> h64(nb * 64);
  std::vector<Out> got(nb);
  srand(7);
  for (int b = 0; b < nb; ++b)
    for (int l = 0; l < 64; ++l) {
      unsigned x = (unsigned)rand() * 2u + (unsigned)(rand() & 1);
      if (b % 3 == 0) x = (rand() % 8 == 0) ? x % 5 : 0;
      if (b % 7 == 1) x >>= 7;
      h[b * 64 + l] = x;
      h64[b * 64 + l] = ((unsigned long long)(unsigned)rand() << 33) ^ ((unsigned long long)x << (b % 5));
    }
  unsigned *d_in, *d_scan; unsigned long long* d_in64; Out* d_out;
  if (hipMalloc(&d_in, h.size() * 4) != hipSuccess || hipMalloc(&d_scan, h.size() * 4) != hipSuccess ||
      hipMalloc(&d_in64, h64.size() * 8) != hipSuccess || hipMalloc(&d_out, nb * sizeof(Out)) != hipSuccess) return 2;
  (void)hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_in64, h64.data(), h64.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d_in, d_in64, d_out, d_scan);
  if (hipMemcpy(got.data(), d_out, nb * sizeof(Out), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  (void)hipMemcpy(scan.data(), d_scan, scan.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; ++b) {
    unsigned s = 0, mn = ~0u, mx = 0, orr = 0; unsigned long long s64 = 0, o64 = 0;
    bool ok = true;
    for (int l = 0; l < 64; ++l) {
      const unsigned x = h[b * 64 + l];
      s += x; mn = x < mn ? x : mn; mx = x > mx ? x : mx; orr |= x;
      s64 += h64[b * 64 + l]; o64 |= h64[b * 64 + l];
      ok = ok && scan[b * 64 + l] == s;
    }
    ok = ok && got[b].sum == s && got[b].mn == mn && got[b].mx == mx && got[b].orr == orr && got[b].sum64 == s64 && got[b].or64 == o64;
    bad += !ok;
  }
  printf("wave primitives: %d of %d waves wrong\n", bad, nb);
  return bad != 0;
}
