// Micro-benchmark: LDS f32 atomic add (ds_add_f32) vs plain LDS read-modify-write throughput.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
  __shared__ float t[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) t[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int base = (wv * 997) & 8191;
  for (int it = 0; it < iters; ++it) {
    const int a = (base + lane * stride) & 8191;
    if (MODE == 0) atomicAdd(&t[a], 1.0f);
    else if (MODE == 1) { float v = t[a]; t[a] = v + 1.0f; }
    else { __hip_atomic_fetch_add(&t[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    base = (base + 64 * stride + 32) & 8191;
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < 8192; i += 256) s += t[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096, blocks = 1024;
  for (int stride : {1, 2, 32}) for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) {
        double winstr = (double)blocks * 4 * iters;
        printf("stride %2d mode %d (%s): %.3f ms  %.2f ns/wave-instr/CU  (%.1f cycles @2.4GHz)\n", stride, mode,
               mode == 0 ? "atomicAdd" : mode == 1 ? "read+write" : "hip_atomic relaxed wg",
               ms, ms * 1e6 / (winstr / 256), ms * 1e6 / (winstr / 256) * 2.4);
      }
    }
  }
  return 0;
}
