// Reductions behind the Linear layers' weight / bias gradients (unibev_amd/linear.py):
//   * grad_bias[n]  = sum over the M = bs*Nq rows of grad_out[M, N]          (column sum)
//   * grad_weight   = sum over the S split-K slices of the strided-batched GEMM partials [S, N*K]
// As framework reductions these were 96 launches and 1.7 ms per training step (9 % of the kernel
// time, profiles/r01_v7_*): a generic reduce kernel per tensor at 17.7 us average.  Here one launch
// serves both outputs of one Linear; the column sum streams grad_out once with 16-byte loads.
#include "ubv_common.h"

namespace ubv {

template <typename T> struct red_vec { static constexpr int kVec = 16 / elem<T>::kBytes; };

// Fat blocks: every block ends with one atomic per column on the SAME N addresses, and same-address
// atomics serialise at ~23 ns each (measured: 512 blocks 22.9 us, 1024 blocks 34.5 us for a 41 MB
// column sum), so the number of blocks — not the bytes — sets the tail.
constexpr int kRedThreads = 1024;

// blocks [0, col_blocks): column sums of go[rows, N] into gb (f32, zeroed by the caller, one atomic
// per column per block); blocks [col_blocks, ...): out[i] = sum_s part[s][i], 16 bytes per thread.
template <typename T>
__global__ __launch_bounds__(kRedThreads) void linear_grad_reduce_kernel(
    const T* __restrict__ go, long rows, int N, float* __restrict__ gb, int col_blocks,
    int rows_per_block, const T* __restrict__ part, int S, long NK, float* __restrict__ gw) {
  constexpr int VEC = red_vec<T>::kVec;
  __shared__ float red[kRedThreads][VEC];
  if ((int)blockIdx.x < col_blocks) {
    const int lpr = N / VEC;                       // threads per row
    const int G = kRedThreads / lpr;               // rows per block step
    const int sub = threadIdx.x / lpr, c = (threadIdx.x - sub * lpr) * VEC;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    if (sub < G) {
      const long r0 = (long)blockIdx.x * rows_per_block;
      const long r1 = min(rows, r0 + rows_per_block);
      for (long r = r0 + sub; r < r1; r += 4 * G) {     // four independent rows in flight
        float a[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long ru = r + (long)u * G;
          if (ru < r1) vec_io<T, VEC>::load(go + ru * N + c, a[u]);
          else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) a[u][i] = 0.0f;
          }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += (a[0][i] + a[1][i]) + (a[2][i] + a[3][i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    for (int col = threadIdx.x; col < N; col += kRedThreads) {
      const int cl = col / VEC, e = col - cl * VEC;
      float s = 0.0f;
      for (int g = 0; g < G; ++g) s += red[g * lpr + cl][e];
      atomic_add_f32(gb + col, s);
    }
    return;
  }
  const long i0 = ((long)(blockIdx.x - col_blocks) * kRedThreads + threadIdx.x) * VEC;
  if (i0 >= NK) return;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
  for (int s = 0; s < S; ++s) {
    float a[VEC];
    vec_io<T, VEC>::load(part + (long)s * NK + i0, a);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += a[i];
  }
#pragma unroll
  for (int i = 0; i < VEC; i += 4)
    *reinterpret_cast<float4*>(gw + i0 + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
}

template <typename T>
static void linear_grad_launch(const void* go, long rows, int N, float* gb, const void* part, int S,
                               long NK, float* gw, hipStream_t st) {
  constexpr int VEC = red_vec<T>::kVec;
  int col_blocks = 0, rpb = 0;
  if (go != nullptr && rows > 0) {
    // one fat block per CU (see kRedThreads)
    rpb = (int)max(128L, (rows + 255) / 256);
    col_blocks = (int)((rows + rpb - 1) / rpb);
  }
  const int sum_blocks = (part != nullptr) ? (int)((NK / VEC + kRedThreads - 1) / kRedThreads) : 0;
  if (col_blocks + sum_blocks == 0) return;
  hipLaunchKernelGGL((linear_grad_reduce_kernel<T>), dim3(col_blocks + sum_blocks), dim3(kRedThreads), 0, st,
                     (const T*)go, rows, N, gb, col_blocks, rpb, (const T*)part, S, NK, gw);
}

// out = a + b, f32, plain (unpacked) adds: the sum of two gradients of one tensor inside the encoders' two-stream window,
// where a framework add kernel (packed f32) must not run
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, long n4, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    // (v_add_f32 spelled out: the backend pairs adjacent f32 adds of a float4 into v_pk_add_f32 even without the SLP
    //  vectoriser, and tests/test_build_isa.py allows no packed f32 instruction in the library)
    float r[4];
    const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[e]) : "v"(xs[e]), "v"(ys[e]));
    reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (i == 0)
    for (long k = 4 * n4; k < n; ++k) {                   // (the tail too: the loop vectoriser would pair these)
      float r;
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a[k]), "v"(b[k]));
      out[k] = r;
    }
}

// out[r * ld + c] = sum_s slices[s][r * cols + c]: the batch sum of S row blocks written into a column slice of a wider
// matrix (the positional fold's gradient slot, encoders._EncoderBase._fold_pos_terms).  One float4 per thread.
__global__ __launch_bounds__(256) void slice_sum_kernel(const float* __restrict__ slices, int S, long rows, int cols4,
                                                        float* __restrict__ out, long ld) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const long r = i / cols4;
  const int c = (int)(i - r * cols4) * 4;
  const long n = rows * cols4 * 4;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int s = 0; s < S; ++s) {
    float a[4];
    vec_io<float, 4>::load(slices + (long)s * n + i * 4, a);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += a[k];
  }
  *reinterpret_cast<float4*>(out + r * ld + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

}  // namespace ubv

extern "C" int ubv_slice_sum_f32(const float* slices, int S, int64_t rows, int cols, float* out, int64_t ld, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(slices != nullptr && out != nullptr && S > 0 && rows > 0 && cols > 0, "slice_sum: bad arguments");
  UBV_CHECK_ARG(cols % 4 == 0 && ld % 4 == 0 && ld >= cols && ((uintptr_t)slices % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "slice_sum: cols=%d and the row stride must be multiples of 4 floats, operands 16-byte aligned", cols);
  const long n4 = rows * (cols / 4);
  hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), slices, S,
                     (long)rows, cols / 4, out, (long)ld);
  UBV_CHECK_LAUNCH("slice_sum");
  return UBV_OK;
}

extern "C" int ubv_add2_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(a != nullptr && b != nullptr && out != nullptr && n > 0, "add2: bad arguments");
  UBV_CHECK_ARG(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "add2: operands must be 16-byte aligned");
  const long n4 = n / 4;
  const long blocks = (n4 > 0 ? n4 : 1) / 256 + 1;
  hipLaunchKernelGGL(add2_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, b, out, n4, (long)n);
  UBV_CHECK_LAUNCH("add2");
  return UBV_OK;
}

extern "C" int ubv_linear_grad_reduce(const void* grad_out, int64_t rows, int N, float* grad_bias,
                                      const void* partials, int S, int64_t NK, float* grad_weight,
                                      int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "linear_grad_reduce: unknown dtype %d", dtype);
  const int vec = dtype == UBV_F32 ? 4 : 8;
  if (grad_out != nullptr) {
    UBV_CHECK_ARG(grad_bias != nullptr && rows >= 0 && N > 0, "linear_grad_reduce: bad bias arguments");
    UBV_CHECK_ARG(N % vec == 0 && N / vec <= 256, "linear_grad_reduce: N=%d must be a multiple of %d and <= %d",
                  N, vec, 256 * vec);
    UBV_CHECK_ARG(((uintptr_t)grad_out % 16) == 0, "linear_grad_reduce: grad_out must be 16-byte aligned");
  }
  if (partials != nullptr) {
    UBV_CHECK_ARG(grad_weight != nullptr && S > 0 && NK > 0 && NK % vec == 0,
                  "linear_grad_reduce: bad split-K arguments");
    UBV_CHECK_ARG(((uintptr_t)partials % 16) == 0 && ((uintptr_t)grad_weight % 16) == 0,
                  "linear_grad_reduce: partials / grad_weight must be 16-byte aligned");
  }
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32: linear_grad_launch<float>(grad_out, rows, N, grad_bias, partials, S, NK, grad_weight, st); break;
    case UBV_F16: linear_grad_launch<f16_t>(grad_out, rows, N, grad_bias, partials, S, NK, grad_weight, st); break;
    default: linear_grad_launch<bf16_t>(grad_out, rows, N, grad_bias, partials, S, NK, grad_weight, st); break;
  }
  UBV_CHECK_LAUNCH("linear_grad_reduce");
  return UBV_OK;
}
