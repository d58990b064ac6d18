// Probe of ds_read_b64_tr_b16 (gfx950): which element reaches which lane.
// LDS holds halves with value = index; lane l points at byte address addr[l] (8-B aligned).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + addr[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  int h_addr[64], h_out[256], *d_addr, *d_out;
  // experiment 1: canonical (lane * 4 halves).  experiment 2: rows of 32 halves (row stride free):
  // lane i of a 16-lane group -> row (i / 4), column quad (i % 4); groups at +512 halves
  for (int e = 0; e < 2; ++e) {
    for (int l = 0; l < 64; ++l) {
      const int g = l / 16, i = l % 16;
      h_addr[l] = e == 0 ? l * 4 : g * 512 + (i / 4) * 32 + (i % 4) * 4;
    }
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("experiment %d\n", e);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d (addr %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
