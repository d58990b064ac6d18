// Gradient clipping + AdamW over FLAT f32 buffers (gfx950).
//
// The training step the bench times ends with mmcv's optimizer hook: clip_grad_norm_(max_norm = 35) and AdamW
// (reference config: optimizer AdamW lr 2e-4 weight_decay 0.01, optimizer_config grad_clip max_norm 35,
// projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:455-462).  As framework calls that is a
// multi-tensor norm (4 launches), a stack, a clamp, a multi-tensor scale and 5 multi-tensor AdamW launches over
// ~200 parameter tensors: 0.4 ms per step.  With parameters, gradients and both moments each living in ONE
// flat f32 buffer (the gradients already do: dp.FlatGradients) it is two streaming passes:
//   1. ubv_sumsq_f32      partial sums of squares per block, then one block adds them in a fixed order
//                         (deterministic; f64 accumulation of the block partials)
//   2. ubv_adamw_flat     p, m, v updated in place from g; the clip coefficient
//                         min(1, max_norm / (sqrt(sumsq) + 1e-6)) and the bias corrections are computed on the
//                         device from the sum of squares and a device-side step counter — nothing is read back,
//                         so the pair can also be captured in a HIP graph.
// Semantics: torch.nn.utils.clip_grad_norm_ (norm_type 2) followed by torch.optim.AdamW (amsgrad off,
// maximize off): p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//                p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
#include <math.h>

#include "ubv_common.h"

namespace ubv {

constexpr int kSqBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long n, double* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.0f;
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    } else {
      for (long j = i; j < n; ++j) acc = fmaf(x[j], x[j], acc);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, int nblocks, float* __restrict__ out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) a += part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

// Parameter groups (torch.optim param_groups; mmcv's paramwise_cfg lr_mult / decay_mult make them): consecutive
// element ranges of the flat buffers with their own learning rate and weight decay.  end[k] is the first element
// past range k; the clip coefficient and the step counter are shared, as in one torch optimizer.
constexpr int kMaxAdamGroups = 16;
struct AdamGroups {
  int count;
  long end[kMaxAdamGroups];
  float lr[kMaxAdamGroups], wd[kMaxAdamGroups];
};

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const AdamGroups grp, float b1, float b2, float eps,
                                                         const long long* __restrict__ step, const float* __restrict__ sumsq,
                                                         float max_norm) {
  const float t = (float)(*step);
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  float clip = 1.0f;
  if (sumsq != nullptr && max_norm > 0.0f) {
    const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    clip = c < 1.0f ? c : 1.0f;
  }
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    const int cnt = (i + 3 < n) ? 4 : (int)(n - i);
    int gi = 0;
    while (gi + 1 < grp.count && i >= grp.end[gi]) ++gi;      // the range of element i; a boundary may fall inside the four
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
      const float4 c = *reinterpret_cast<const float4*>(m + i), d = *reinterpret_cast<const float4*>(v + i);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int k = 0; k < cnt; ++k) { pv[k] = p[i + k]; gv[k] = g[i + k]; mv[k] = m[i + k]; vv[k] = v[i + k]; }
    }
    for (int k = 0; k < cnt; ++k) {
      while (gi + 1 < grp.count && i + k >= grp.end[gi]) ++gi;
      const float lr = grp.lr[gi], step_size = lr / bc1, decay = 1.0f - lr * grp.wd[gi];
      const float gk = gv[k] * clip;
      pv[k] *= decay;
      mv[k] = b1 * mv[k] + (1.0f - b1) * gk;
      vv[k] = b2 * vv[k] + (1.0f - b2) * gk * gk;
      pv[k] -= step_size * mv[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
    }
    if (cnt == 4) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (int k = 0; k < cnt; ++k) { p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k]; }
    }
  }
}

__global__ void step_inc_kernel(long long* step) { *step += 1; }

}  // namespace ubv

extern "C" int64_t ubv_sumsq_workspace(void) { return (int64_t)ubv::kSqBlocks * sizeof(double); }

extern "C" int ubv_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && out && workspace && n >= 0 && ((uintptr_t)x % 16) == 0, "sumsq_f32: bad arguments");
  hipStream_t st = as_stream(stream);
  long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > kSqBlocks) blocks = kSqBlocks;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)n, (double*)workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, (int)blocks, out);
  UBV_CHECK_LAUNCH("sumsq_f32");
  return UBV_OK;
}

static int adamw_run(float* p, const float* g, float* m, float* v, int64_t n, const ubv::AdamGroups& grp, float beta1,
                     float beta2, float eps, int64_t* step, const float* sumsq, float max_norm, void* stream);

extern "C" int ubv_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t* step, const float* sumsq,
                              float max_norm, void* stream) {
  ubv::AdamGroups grp{};
  grp.count = 1;
  grp.end[0] = (long)n;
  grp.lr[0] = lr;
  grp.wd[0] = weight_decay;
  return adamw_run(p, g, m, v, n, grp, beta1, beta2, eps, step, sumsq, max_norm, stream);
}

extern "C" int ubv_adamw_flat_max_groups(void) { return ubv::kMaxAdamGroups; }

extern "C" int ubv_adamw_flat_groups(float* p, const float* g, float* m, float* v, int64_t n, int n_groups,
                                     const int64_t* group_end, const float* group_lr, const float* group_weight_decay,
                                     float beta1, float beta2, float eps, int64_t* step, const float* sumsq,
                                     float max_norm, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(group_end && group_lr && group_weight_decay && n_groups >= 1, "adamw_flat_groups: bad arguments");
  if (n_groups > kMaxAdamGroups) {
    set_error("adamw_flat_groups: %d groups (ranges of one lr / weight decay), at most %d", n_groups, kMaxAdamGroups);
    return UBV_ERR_UNSUPPORTED;
  }
  AdamGroups grp{};
  grp.count = n_groups;
  int64_t prev = 0;
  for (int k = 0; k < n_groups; ++k) {
    UBV_CHECK_ARG(group_end[k] >= prev && group_end[k] <= n, "adamw_flat_groups: ends must ascend inside [0, n]");
    prev = group_end[k];
    grp.end[k] = (long)group_end[k];
    grp.lr[k] = group_lr[k];
    grp.wd[k] = group_weight_decay[k];
  }
  UBV_CHECK_ARG(prev == n, "adamw_flat_groups: the last range must end at n");
  return adamw_run(p, g, m, v, n, grp, beta1, beta2, eps, step, sumsq, max_norm, stream);
}

static int adamw_run(float* p, const float* g, float* m, float* v, int64_t n, const ubv::AdamGroups& grp, float beta1,
                     float beta2, float eps, int64_t* step, const float* sumsq, float max_norm, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(p && g && m && v && step && n >= 0, "adamw_flat: bad arguments");
  UBV_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                "adamw_flat: buffers must be 16-byte aligned");
  UBV_CHECK_ARG(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps > 0.0f, "adamw_flat: bad hyper-parameters");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, (long long*)step);
  if (n > 0) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long)n, grp, beta1, beta2,
                       eps, (const long long*)step, sumsq, max_norm);
  }
  UBV_CHECK_LAUNCH("adamw_flat");
  return UBV_OK;
}
