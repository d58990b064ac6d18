// Checks the operand layout assumed by the owner-tile backward for v_mfma_f32_32x32x16_bf16:
//   A: lane l holds row i = l & 31, 8 k-values of group g = l >> 5
//   B: lane l holds col j = l & 31, 8 k-values of group g = l >> 5 (same k slots as A)
//   D: reg r of lane l is element (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l & 31)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __bf16 tobf(float f) { unsigned u = __float_as_uint(f); unsigned short h = u >> 16; return __builtin_bit_cast(__bf16, h); }
__global__ void k(const float* A, const float* B, float* D) {   // A[32][16], B[16][32], D[32][32]
  const int l = threadIdx.x, g = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = tobf(A[(l & 31) * 16 + g * 8 + j]); b[j] = tobf(B[(g * 8 + j) * 32 + (l & 31)]); }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + (l & 31)] = c[r];
}
int main() {
  float hA[512], hB[512], hD[1024], ref[1024];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6); hB[i] = (float)((i * 5 + 1) % 11 - 5); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
  printf("mfma layout max err %g  -> %s\n", e, e == 0 ? "OK" : "MISMATCH");
  return e != 0;
}
