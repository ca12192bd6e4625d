// lds_atomic_probe.hip -- what an LDS atomic costs on gfx950 as a function of the operation and of how the 64
// lanes' addresses fall on the banks (the SpMV's column-sorted format accumulates its row sums with them).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_probe lds_atomic_probe.hip && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kRows = 32768, kThreads = 1024, kIters = 4096;

__device__ inline unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// mode 0 random rows; 1 each half-wave hits 32 distinct banks; 2 all lanes one bank (32-way); 3 consecutive rows
template <int OP>
__global__ __launch_bounds__(kThreads) void probe(int mode, float* out) {
  __shared__ float ys[kRows];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kRows; i += kThreads) ys[i] = 0.f;
  __syncthreads();
  unsigned s = hash(blockIdx.x * kThreads + tid + 1);
  for (int it = 0; it < kIters; ++it) {
    s = s * 1664525u + 1013904223u;
    unsigned r = (s >> 8) & (kRows - 1);
    if (mode == 1) r = (r & ~31u) | (unsigned)(lane & 31);
    if (mode == 2) r = r & ~31u;
    if (mode == 3) r = ((s >> 8) & (kRows - 64)) + lane;
    const float v = (float)(s & 7);
    if (OP == 0) atomicAdd(&ys[r], v);
    if (OP == 1) atomicAdd(reinterpret_cast<unsigned*>(&ys[r]), s & 7u);
    if (OP == 2) atomicMin(reinterpret_cast<unsigned*>(&ys[r]), s);
    if (OP == 3) { ys[r] = v; }                                        // plain store for scale
    if (OP == 4) { const float o = ys[r]; if (v > o) ys[r] = v; }      // read + conditional store
    if (OP == 5) (void)atomicAdd(reinterpret_cast<unsigned long long*>(&ys[r & ~1u]), (unsigned long long)(s & 7u));
    if (OP == 6) {                                                     // float add as a compare-and-swap loop
      unsigned* a = reinterpret_cast<unsigned*>(&ys[r]);
      unsigned old = *a, assumed;
      do {
        assumed = old;
        old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
      } while (old != assumed);
    }
    if (OP == 7) atomicAdd(reinterpret_cast<double*>(&ys[r & ~1u]), (double)v);
    if (OP == 8) { const float o = atomicAdd(&ys[r], v); if (o == 12345.6f) ys[0] = o; }   // returning float add
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = tid; i < kRows; i += kThreads) acc += ys[i];
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* ops[] = {"ds_add_f32", "ds_add_u32", "ds_min_u32", "ds_write_b32", "read+cond write", "ds_add_u64",
                       "cas float add", "ds_add_f64", "ds_add_rtn_f32"};
  const char* modes[] = {"random", "bank-balanced halves", "one bank", "consecutive"};
  for (int op = 0; op < 9; ++op)
    for (int mode = 0; mode < 4; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (op) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 5: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 6: hipLaunchKernelGGL(probe<6>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          case 7: hipLaunchKernelGGL(probe<7>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
          default: hipLaunchKernelGGL(probe<8>, dim3(256), dim3(kThreads), 0, 0, mode, d); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      // per CU: 16 waves x kIters wave-instructions
      const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * kIters);
      printf("%-16s %-22s %8.3f ms  %6.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", ops[op], modes[mode], ms, cyc);
    }
  return 0;
}
