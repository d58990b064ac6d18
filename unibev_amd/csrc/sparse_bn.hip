// BatchNorm1d (+ ReLU) over the ACTIVE rows of a sparse feature matrix [N, C] — the norm / activation between the
// sparse convolutions of the LiDAR middle encoder ([ext] mmdet3d
// make_sparse_convmodule / SparseBasicBlock: SparseSequential(conv, BatchNorm1d(eps 1e-3, momentum 0.01), ReLU)).
//
// N is 55 k - 185 k rows, C is 16 - 128 channels: a row is 64 - 512 bytes.  The framework's batch norm treats the
// matrix as a channels-last image and spends 73 us on the statistics and 85 us on the backward reduction of each of
// the 21 layers — latency-bound reductions over few channels (4 ms of a 25 ms encoder pass).  Here:
// The basic block's tail  relu(bn2(conv2) + identity)  is the same pass with a residual operand (forward: added before
// the ReLU; backward: the mask comes from the stored output, the masked gradient is also the identity's gradient).
//   forward   bn_stats_kernel (one streaming pass: per-block partial sums of x and x^2, 16-byte loads, fixed block ->
//             rows map) + bn_finalize_kernel (partials summed IN BLOCK ORDER: deterministic; mean, 1 / sqrt(var +
//             eps), running statistics with the unbiased variance as torch does) + bn_apply_kernel (normalise,
//             affine, optional ReLU, one pass)
//   backward  bn_bwd_reduce_kernel (partials of sum dy' and sum dy' * xhat, dy' = dy masked by the ReLU) +
//             bn_bwd_finalize_kernel (d gamma, d beta) + bn_bwd_apply_kernel
//             dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat))
// f32 statistics and accumulation; data f32 (the encoder's default) or 16-bit.
#include "ubv_common.h"

namespace ubv {

constexpr int kBnBlocks = 1024;        // upper bound of stats blocks (partials buffer: kBnBlocks x 2C floats)

template <typename T>
__device__ __forceinline__ void bn_load4(const T* p, float (&v)[4]) { vec_io<T, 4>::load(p, v); }

// Thread t: channel quad cq = t % (C / 4), row lane rl = t / (C / 4); rows [r0, r1) of the block in steps of RL.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ yout,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         long N, int C, int relu, long rows_per_block,
                                                         float* __restrict__ partial) {
  __shared__ float red[2][256][4];
  const int CQ = C / 4, RL = 256 / CQ;
  const int cq = threadIdx.x % CQ, rl = threadIdx.x / CQ;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float mu[4], rs[4], g[4], b[4];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu[i] = mean[cq * 4 + i]; rs[i] = rstd[cq * 4 + i]; g[i] = gamma[cq * 4 + i]; b[i] = beta[cq * 4 + i]; }
  }
  // forward statistics are sums of x - x[0] (per channel): E[x^2] - mean^2 of the raw values cancels when |mean| is much
  // larger than the spread; shifted by a sample of the channel it does not (bn_finalize_kernel adds the shift back)
  float sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (!BWD) bn_load4<T>(x + cq * 4, sh);
  if (rl < RL) {
    for (long r = r0 + rl; r < r1; r += RL) {
      float v[4];
      bn_load4<T>(x + r * C + cq * 4, v);
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float u = v[i] - sh[i]; s1[i] += u; s2[i] = fmaf(u, u, s2[i]); }
      } else {
        float d[4], yo[4] = {1.f, 1.f, 1.f, 1.f};
        bn_load4<T>(dy + r * C + cq * 4, d);
        if (yout != nullptr) bn_load4<T>(yout + r * C + cq * 4, yo);       // (residual form: the mask is the stored output)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (v[i] - mu[i]) * rs[i];
          const float yv = yout != nullptr ? yo[i] : fmaf(xh, g[i], b[i]);
          const float dd = (relu && yv <= 0.0f) ? 0.0f : d[i];
          s1[i] += dd;
          s2[i] = fmaf(dd, xh, s2[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
  __syncthreads();
  if (threadIdx.x < CQ) {            // fixed order over the row lanes: deterministic
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < RL; ++l)
#pragma unroll
      for (int i = 0; i < 4; ++i) { a1[i] += red[0][l * CQ + threadIdx.x][i]; a2[i] += red[1][l * CQ + threadIdx.x][i]; }
    float* p = partial + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[threadIdx.x * 4 + i] = a1[i]; p[C + threadIdx.x * 4 + i] = a2[i]; }
  }
}

// Sum of the per-block partials of channel c: one 256-thread block per channel — thread t adds blocks t, t + 256, ...
// (at most 4), then a fixed-shape tree in LDS: deterministic.  (One thread walking all 1024 partials was a
// 1024-deep chain of dependent loads: 117 us per call.)
__device__ __forceinline__ void bn_block_sums(const float* __restrict__ partial, int blocks, int C, int c, double& s1,
                                              double& s2) {
  __shared__ double red[2][256];
  double a1 = 0.0, a2 = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) {
    a1 += partial[(long)b * 2 * C + c];
    a2 += partial[(long)b * 2 * C + C + c];
  }
  red[0][threadIdx.x] = a1; red[1][threadIdx.x] = a2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
    __syncthreads();
  }
  s1 = red[0][0]; s2 = red[1][0];
}

// forward: mean / rstd (+ running statistics); one block per channel
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int blocks, long N, int C,
                                                          float eps, float momentum, const T* __restrict__ x,
                                                          float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var) {
  const int c = blockIdx.x;
  double s1, s2;
  bn_block_sums(partial, blocks, C, c, s1, s2);
  if (threadIdx.x != 0) return;
  const double ms = s1 / (double)N;                            // mean of x - x[0]
  double var = s2 / (double)N - ms * ms;
  var = var > 0.0 ? var : 0.0;
  const double m = ms + (double)elem<T>::to_float(x[c]);
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// backward: sums -> d gamma (sum dy' xhat), d beta (sum dy')
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int C,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma) {
  const int c = blockIdx.x;
  double s1, s2;
  bn_block_sums(partial, blocks, C, c, s1, s2);
  if (threadIdx.x != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
}

// MODE 0: y = act(xhat * gamma + beta (+ res)).  MODE 1: dx = gamma * rstd * (dy' - s1 / N - xhat * s2 / N), and with a
// residual (res = the stored forward output, the ReLU mask) dres = dy'
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const T* __restrict__ res,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ s1, const float* __restrict__ s2,
                                                       long N, int C, int relu, T* __restrict__ out, T* __restrict__ dres) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;              // one 4-channel piece per thread
  const long total = N * (C / 4);
  if (i4 >= total) return;
  const int c0 = (int)(i4 % (C / 4)) * 4;
  float v[4], o[4];
  bn_load4<T>(x + i4 * 4, v);
  float d[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, dr[4];
  if (MODE == 1) bn_load4<T>(dy + i4 * 4, d);
  if (res != nullptr) bn_load4<T>(res + i4 * 4, rr);
  const float invn = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    const float xh = (v[i] - mean[c]) * rstd[c];
    const float y = fmaf(xh, gamma[c], beta[c]);
    if (MODE == 0) {
      const float yr = y + rr[i];
      o[i] = (relu && yr <= 0.0f) ? 0.0f : yr;
    } else {
      const float yv = res != nullptr ? rr[i] : y;
      const float dd = (relu && yv <= 0.0f) ? 0.0f : d[i];
      dr[i] = dd;
      o[i] = gamma[c] * rstd[c] * (dd - s1[c] * invn - xh * s2[c] * invn);
    }
  }
  vec_io<T, 4>::store(out + i4 * 4, o);
  if (MODE == 1 && dres != nullptr) vec_io<T, 4>::store(dres + i4 * 4, dr);
}

static int bn_blocks(long N) {
  long b = (N + 255) / 256;
  return (int)(b < 1 ? 1 : (b > kBnBlocks ? kBnBlocks : b));
}

template <typename T>
static void bn_forward_T(const void* x, const void* res, const float* gamma, const float* beta, float* rm, float* rv,
                         float* mean, float* rstd, float* partial, void* y, long N, int C, float eps, float momentum,
                         int relu, int training, hipStream_t st) {
  if (training) {
    const int blocks = bn_blocks(N);
    const long rpb = (N + blocks - 1) / blocks;
    hipLaunchKernelGGL((bn_partial_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, nullptr, nullptr, N, C, 0, rpb, partial);
    hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3(C), dim3(256), 0, st, partial, blocks, N, C, eps, momentum,
                       (const T*)x, mean, rstd, rm, rv);
  }
  const long pieces = N * (C / 4);
  hipLaunchKernelGGL((bn_apply_kernel<T, 0>), dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const T*)x,
                     (const T*)nullptr, (const T*)res, mean, rstd, gamma, beta, nullptr, nullptr, N, C, relu, (T*)y,
                     (T*)nullptr);
}

template <typename T>
static void bn_backward_T(const void* x, const void* dy, const void* yout, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, float* partial, float* dgamma, float* dbeta, void* dx,
                          void* dres, long N, int C, int relu, hipStream_t st) {
  const int blocks = bn_blocks(N);
  const long rpb = (N + blocks - 1) / blocks;
  hipLaunchKernelGGL((bn_partial_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)dy,
                     (const T*)yout, mean, rstd, gamma, beta, N, C, relu, rpb, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, partial, blocks, C, dbeta, dgamma);
  const long pieces = N * (C / 4);
  hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const T*)x,
                     (const T*)dy, (const T*)yout, mean, rstd, gamma, beta, dbeta, dgamma, N, C, relu, (T*)dx, (T*)dres);
}

}  // namespace ubv

extern "C" int64_t ubv_rows_bn_partial_elems(int C) { return (int64_t)ubv::kBnBlocks * 2 * C; }

extern "C" int ubv_rows_bn_forward(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* mean, float* rstd, float* partial, void* y, int64_t N,
                                   int C, float eps, float momentum, int relu, int training, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && gamma && beta && mean && rstd && y && N > 0 && C > 0, "rows_bn_forward: bad arguments");
  UBV_CHECK_ARG(!training || partial != nullptr, "rows_bn_forward: training mode needs the partials buffer");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "rows_bn_forward: unknown dtype %d", dtype);
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0 || ((uintptr_t)x % 8) != 0 || ((uintptr_t)y % 8) != 0) {
    set_error("rows_bn_forward: C=%d must be a multiple of 4 with C / 4 dividing 256, rows 8-byte aligned", C);
    return UBV_ERR_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  if (dtype == UBV_F32) bn_forward_T<float>(x, residual, gamma, beta, running_mean, running_var, mean, rstd, partial, y, N, C, eps, momentum, relu, training, st);
  else if (dtype == UBV_F16) bn_forward_T<f16_t>(x, residual, gamma, beta, running_mean, running_var, mean, rstd, partial, y, N, C, eps, momentum, relu, training, st);
  else bn_forward_T<bf16_t>(x, residual, gamma, beta, running_mean, running_var, mean, rstd, partial, y, N, C, eps, momentum, relu, training, st);
  UBV_CHECK_LAUNCH("rows_bn_forward");
  return UBV_OK;
}

extern "C" int ubv_rows_bn_backward(const void* x, const void* grad_y, const void* y_out, const float* gamma,
                                    const float* beta, const float* mean, const float* rstd, float* partial,
                                    float* grad_gamma, float* grad_beta, void* grad_x, void* grad_residual, int64_t N,
                                    int C, int relu, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && grad_y && gamma && beta && mean && rstd && partial && grad_gamma && grad_beta && grad_x && N > 0 &&
                    C > 0, "rows_bn_backward: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "rows_bn_backward: unknown dtype %d", dtype);
  UBV_CHECK_ARG((y_out == nullptr) == (grad_residual == nullptr), "rows_bn_backward: y_out and grad_residual come together");
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) {
    set_error("rows_bn_backward: C=%d must be a multiple of 4 with C / 4 dividing 256", C);
    return UBV_ERR_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  if (dtype == UBV_F32) bn_backward_T<float>(x, grad_y, y_out, gamma, beta, mean, rstd, partial, grad_gamma, grad_beta, grad_x, grad_residual, N, C, relu, st);
  else if (dtype == UBV_F16) bn_backward_T<f16_t>(x, grad_y, y_out, gamma, beta, mean, rstd, partial, grad_gamma, grad_beta, grad_x, grad_residual, N, C, relu, st);
  else bn_backward_T<bf16_t>(x, grad_y, y_out, gamma, beta, mean, rstd, partial, grad_gamma, grad_beta, grad_x, grad_residual, N, C, relu, st);
  UBV_CHECK_LAUNCH("rows_bn_backward");
  return UBV_OK;
}
