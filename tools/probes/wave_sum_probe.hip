// Checks wave_sum_u32 (common.hpp: the DPP row-shift / row-broadcast sum of a wave's 64 lanes) against a serial sum for
// random inputs, including zeros and values near 2^31.  Build: hipcc --offload-arch=gfx950 -I graphblast_amd/csrc -I include
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace grb;
__global__ void k(const unsigned* in, unsigned* out) {
  const unsigned v = in[blockIdx.x * 64 + threadIdx.x];
  const unsigned r = wave_sum_u32(v);
  if (threadIdx.x == 17) out[blockIdx.x] = r;       // any lane: the result is wave-uniform
}
int main() {
  const int nb = 4096;
  std::vector<unsigned> h(nb * 64), want(nb), got(nb);
  srand(7);
  for (int b = 0; b < nb; ++b) {
    unsigned s = 0;
    for (int l = 0; l < 64; ++l) {
      unsigned x = (unsigned)rand();
      if (b % 3 == 0) x = (rand() % 8 == 0) ? x % 5 : 0;
      if (b % 7 == 1) x >>= 7;
      h[b * 64 + l] = x; s += x;
    }
    want[b] = s;
  }
  unsigned *d_in, *d_out;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, nb * 4);
  hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d_in, d_out);
  hipMemcpy(got.data(), d_out, nb * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < nb; ++b) bad += got[b] != want[b];
  printf("wave_sum_u32: %d of %d waves wrong\n", bad, nb);
  return bad != 0;
}
