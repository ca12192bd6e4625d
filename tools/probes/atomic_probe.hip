// Throughput of scattered 4-byte operations on a 512 KiB bitmap from the whole chip:
// agent-scope atomicOr with / without return, workgroup-scope atomicOr, plain store, plain load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline unsigned rng(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 4; }

template <int MODE>
__global__ void probe(unsigned* bm, unsigned nwords, int per_thread, unsigned* sink) {
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  unsigned acc = 0;
  for (int i = 0; i < per_thread; i += 4) {
    unsigned w[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { unsigned r = rng(s); w[j] = r % nwords; b[j] = 1u << (r & 31); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) acc += __hip_atomic_fetch_or(&bm[w[j]], b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 1) (void)__hip_atomic_fetch_or(&bm[w[j]], b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 2) acc += __hip_atomic_fetch_or(&bm[w[j]], b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 3) bm[w[j]] = b[j];
      if (MODE == 4) acc += bm[w[j]];
      if (MODE == 5) acc += __hip_atomic_load(&bm[w[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 6) __hip_atomic_store(&bm[w[j]], b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int MODE>
int run(const char* name, unsigned* bm, unsigned nwords, unsigned* sink) {
  const int blocks = 256 * 8, threads = 256, per_thread = 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE><<<blocks, threads>>>(bm, nwords, per_thread, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, threads>>>(bm, nwords, per_thread, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = 5.0 * blocks * threads * per_thread;
  printf("%-34s %8.1f us per launch, %7.1f G ops/s\n", name, ms / 5 * 1e3, ops / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  for (unsigned nwords : {131072u, 4194304u}) {
    unsigned *bm, *sink;
    CK(hipMalloc(&bm, 4ull * nwords)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(bm, 0, 4ull * nwords));
    printf("---- %u words (%u KiB)\n", nwords, nwords / 256);
    if (run<0>("atomicOr agent, returning", bm, nwords, sink)) return 1;
    if (run<1>("atomicOr agent, no return", bm, nwords, sink)) return 1;
    if (run<2>("atomicOr workgroup, returning", bm, nwords, sink)) return 1;
    if (run<3>("plain store", bm, nwords, sink)) return 1;
    if (run<6>("agent-scope (write-through) store", bm, nwords, sink)) return 1;
    if (run<4>("plain load", bm, nwords, sink)) return 1;
    if (run<5>("agent-scope load", bm, nwords, sink)) return 1;
    CK(hipFree(bm)); CK(hipFree(sink));
  }
  return 0;
}
