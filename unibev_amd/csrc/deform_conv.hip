// Modulated deformable convolution (DCNv2) — SURVEY.md section 8 row f4: the one custom kernel of the image
// backbone (ResNet-101 stages 3 and 4 under `dcn=dict(type='DCNv2', deform_groups=1)`, config :229-236).
//
// Reference: [ext] mmcv ops/csrc/common/cuda/modulated_deform_conv_cuda_kernel.cuh
// (modulated_deformable_im2col / col2im / col2im_coord) + the GEMMs around them.  mmcv works on NCHW: its
// im2col thread owns one (channel, output pixel) and walks the taps, a column matrix [C*kh*kw, N*Ho*Wo].  Here
// the feature map is channels-last and the column matrix is [N*Ho*Wo, kh*kw*C] with the TAP outermost, so a
// lane is 16 bytes of consecutive channels of one (pixel, tap): every gather and every column store is a
// coalesced row segment, the convolution itself is ubv_gemm_nt on the row-major columns (weights permuted once
// to [Cout, kh*kw*C]) and the result is born channels-last.  Backward: dCol = dY . W (ubv_gemm_nt); d(input) on
// owner tiles with MFMA (dcn_owner.inl) for the samples near their tap, f32 atomics here for the rest; d(offset) /
// d(mask) reduced over the channels of the deformable group inside the wave; dW is ubv_gemm_wgrad over
// (dY, columns).
//
// Sampling semantics (dmcn_im2col_bilinear): position p = (ho*s - pad + i*dil + dy, wo*s - pad + j*dil + dx);
// a position with p <= -1 or p >= size contributes 0, otherwise the four corners with corners outside the map
// read as 0.  offset[n, 2*(g*K + k)] is dy, [.. + 1] is dx, mask[n, g*K + k] the modulation.
#include "ubv_common.h"

namespace ubv {

struct DcnGeom {
  int N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
};

struct DcnTap {
  bool live;           // position inside (-1, H) x (-1, W)
  int h0, w0;          // floor of the position
  float lh, lw;        // fractional parts
  float dy, dx;        // the raw offsets
};

template <typename T>
__device__ __forceinline__ DcnTap dcn_tap(const DcnGeom& g, const T* __restrict__ offset, int n, int grp, int k,
                                          int ho, int wo) {
  const int K = g.kh * g.kw, i = k / g.kw, j = k - i * g.kw;
  const long plane = (long)g.Ho * g.Wo;
  const long ob = (((long)n * g.dg + grp) * 2 * K + 2 * k) * plane + (long)ho * g.Wo + wo;
  const float dy = elem<T>::to_float(offset[ob]), dx = elem<T>::to_float(offset[ob + plane]);
  const float hp = (float)(ho * g.sh - g.ph + i * g.dh) + dy;
  const float wp = (float)(wo * g.sw - g.pw + j * g.dw) + dx;
  DcnTap t;
  t.dy = dy; t.dx = dx;
  t.live = hp > -1.0f && wp > -1.0f && hp < (float)g.H && wp < (float)g.W;
  const float hf = floorf(hp), wf = floorf(wp);
  t.h0 = t.live ? (int)hf : 0;
  t.w0 = t.live ? (int)wf : 0;
  t.lh = t.live ? hp - hf : 0.0f;
  t.lw = t.live ? wp - wf : 0.0f;
  return t;
}

// columns[m, k*C + c] = mask * bilinear(x[n, :, :, c]) — one lane = VEC channels of one (m, k)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const T* __restrict__ x, const T* __restrict__ offset,
                                                         const T* __restrict__ mask, T* __restrict__ col,
                                                         const DcnGeom g) {
  const int CV = g.C / VEC, K = g.kh * g.kw;
  const long total = (long)g.N * g.Ho * g.Wo * K * CV;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int cv = (int)(t % CV);
  long r = t / CV;
  const int k = (int)(r % K);
  const long m = r / K;
  const int wo = (int)(m % g.Wo), ho = (int)((m / g.Wo) % g.Ho), n = (int)(m / ((long)g.Wo * g.Ho));
  const int c = cv * VEC, grp = c / (g.C / g.dg);
  const DcnTap tp = dcn_tap<T>(g, offset, n, grp, k, ho, wo);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
  if (tp.live) {
    const float mk = elem<T>::to_float(mask[(((long)n * g.dg + grp) * K + k) * g.Ho * g.Wo + (long)ho * g.Wo + wo]);
    const float wgt[4] = {(1.0f - tp.lh) * (1.0f - tp.lw), (1.0f - tp.lh) * tp.lw, tp.lh * (1.0f - tp.lw), tp.lh * tp.lw};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hc = tp.h0 + (q >> 1), wc = tp.w0 + (q & 1);
      if (hc < 0 || wc < 0 || hc >= g.H || wc >= g.W) continue;
      float v[VEC];
      vec_io<T, VEC>::load(x + (((long)n * g.H + hc) * g.W + wc) * g.C + c, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt[q] * mk, v[e], acc[e]);
    }
  }
  vec_io<T, VEC>::store(col + m * ((long)K * g.C) + (long)k * g.C + c, acc);
}

// One wave per (m, k, group): d(input) by f32 atomics, d(offset) / d(mask) reduced over the group's channels.
// 4 channels per lane for every data type (with 8, a 256-channel group kept half the wave idle and the 16-bit
// launch took twice the f32 one).  The kernel is bound by its atomics — 36 per input element, neighbouring waves
// on the same lines: 890 us at 12 x 256 x 16x44, against 44 us for the im2col.  Two LDS-window variants (8x8 output
// tile x 64 channels, window flushed once) were measured and dropped: ds_add_f32 runs at ~190 clocks per wave
// instruction (692 us), and wave-private channels with plain read-modify-writes need 16x more wave iterations
// (instruction-bound, 1 100 us); walking the taps outermost so that concurrent waves hit different lines changed
// nothing (1 166 vs 1 175 us forward + backward): it is the atomic RATE (~88 G/s), not contention.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const T* __restrict__ gcol, const T* __restrict__ x,
                                                         const T* __restrict__ offset, const T* __restrict__ mask,
                                                         float* __restrict__ gx, float* __restrict__ goff,
                                                         float* __restrict__ gmask, const DcnGeom g, int reach) {
  const int K = g.kh * g.kw, Cg = g.C / g.dg, CV = Cg / VEC;
  const long waves = (long)g.N * g.Ho * g.Wo * K * g.dg;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= waves) return;
  const int lane = threadIdx.x & 63;
  const int grp = (int)(wave % g.dg);
  long r = wave / g.dg;
  const int k = (int)(r % K);
  const long m = r / K;
  const int wo = (int)(m % g.Wo), ho = (int)((m / g.Wo) % g.Ho), n = (int)(m / ((long)g.Wo * g.Ho));
  const DcnTap tp = dcn_tap<T>(g, offset, n, grp, k, ho, wo);
  const long plane = (long)g.Ho * g.Wo, pix = (long)ho * g.Wo + wo;
  const long mi = (((long)n * g.dg + grp) * K + k) * plane + pix;
  const long oi = (((long)n * g.dg + grp) * 2 * K + 2 * k) * plane + pix;
  float s_mask = 0.0f, s_h = 0.0f, s_w = 0.0f;
  // reach >= 0: the owner-tile pass (dcn_owner.inl) took every sample with |dy|, |dx| <= reach; only the others
  // are scattered here (same predicate on the same values)
  const bool scatter = reach < 0 || !(fabsf(tp.dy) <= (float)reach && fabsf(tp.dx) <= (float)reach);
  if (tp.live) {                                                     // wave-uniform
    const float mk = elem<T>::to_float(mask[mi]);
    const float hh = 1.0f - tp.lh, hw = 1.0f - tp.lw;
    const float wgt[4] = {hh * hw, hh * tp.lw, tp.lh * hw, tp.lh * tp.lw};
    const float dh[4] = {-hw, -tp.lw, hw, tp.lw}, dw[4] = {-hh, hh, -tp.lh, tp.lh};
    for (int cv = lane; cv < CV; cv += 64) {
      const int c = grp * Cg + cv * VEC;
      float gc[VEC];
      vec_io<T, VEC>::load(gcol + m * ((long)K * g.C) + (long)k * g.C + c, gc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int hc = tp.h0 + (q >> 1), wc = tp.w0 + (q & 1);
        if (hc < 0 || wc < 0 || hc >= g.H || wc >= g.W) continue;   // wave-uniform
        const long xo = (((long)n * g.H + hc) * g.W + wc) * g.C + c;
        float v[VEC];
        vec_io<T, VEC>::load(x + xo, v);
        float dot = 0.0f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          dot = fmaf(gc[e], v[e], dot);
          if (scatter) atomic_add_f32(gx + xo + e, gc[e] * mk * wgt[q]);
        }
        s_mask = fmaf(dot, wgt[q], s_mask);
        s_h = fmaf(dot, dh[q] * mk, s_h);
        s_w = fmaf(dot, dw[q] * mk, s_w);
      }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      s_mask += __shfl_xor(s_mask, sft, 64);
      s_h += __shfl_xor(s_h, sft, 64);
      s_w += __shfl_xor(s_w, sft, 64);
    }
  }
  if (lane == 0) {
    gmask[mi] = s_mask;
    goff[oi] = s_h;
    goff[oi + plane] = s_w;
  }
}

static int dcn_check(const DcnGeom& g, int dtype, const char* who) {
  UBV_CHECK_ARG(g.N >= 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.Ho > 0 && g.Wo > 0 && g.kh > 0 && g.kw > 0 &&
                    g.sh > 0 && g.sw > 0 && g.dh > 0 && g.dw > 0 && g.dg > 0 && g.ph >= 0 && g.pw >= 0,
                "%s: bad geometry", who);
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "%s: unknown dtype %d", who, dtype);
  const int vec = dtype == UBV_F32 ? 4 : 8;
  UBV_CHECK_ARG(g.C % g.dg == 0 && (g.C / g.dg) % vec == 0,
                "%s: channels per deformable group (%d / %d) must be a multiple of %d", who, g.C, g.dg, vec);
  return UBV_OK;
}

}  // namespace ubv

extern "C" int ubv_dcn_im2col(const void* x, const void* offset, const void* mask, void* columns, int N, int H,
                              int W, int C, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                              int dw, int deform_groups, int dtype, void* stream) {
  using namespace ubv;
  const DcnGeom g{N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, deform_groups};
  int rc = dcn_check(g, dtype, "dcn_im2col");
  if (rc) return rc;
  if (N == 0) return UBV_OK;
  UBV_CHECK_ARG(x && offset && mask && columns, "dcn_im2col: null pointer");
  const int vec = dtype == UBV_F32 ? 4 : 8;
  const long total = (long)N * Ho * Wo * kh * kw * (C / vec);
  UBV_CHECK_ARG((total + 255) / 256 < (1L << 31), "dcn_im2col: too many elements");
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32: hipLaunchKernelGGL((dcn_im2col_kernel<float, 4>), grid, blk, 0, st, (const float*)x, (const float*)offset, (const float*)mask, (float*)columns, g); break;
    case UBV_F16: hipLaunchKernelGGL((dcn_im2col_kernel<f16_t, 8>), grid, blk, 0, st, (const f16_t*)x, (const f16_t*)offset, (const f16_t*)mask, (f16_t*)columns, g); break;
    default: hipLaunchKernelGGL((dcn_im2col_kernel<bf16_t, 8>), grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)offset, (const bf16_t*)mask, (bf16_t*)columns, g); break;
  }
  UBV_CHECK_LAUNCH("dcn_im2col");
  return UBV_OK;
}

extern "C" int ubv_dcn_col2im(const void* grad_columns, const void* x, const void* offset, const void* mask,
                              float* grad_x, float* grad_offset, float* grad_mask, int N, int H, int W, int C,
                              int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                              int deform_groups, int dtype, void* stream) {
  using namespace ubv;
  const DcnGeom g{N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, deform_groups};
  int rc = dcn_check(g, dtype, "dcn_col2im");
  if (rc) return rc;
  if (N == 0) return UBV_OK;
  UBV_CHECK_ARG(grad_columns && x && offset && mask && grad_x && grad_offset && grad_mask, "dcn_col2im: null pointer");
  const long waves = (long)N * Ho * Wo * kh * kw * deform_groups;
  UBV_CHECK_ARG((waves + 3) / 4 < (1L << 31), "dcn_col2im: too many elements");
  const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
  hipStream_t st = as_stream(stream);
  // d(input): owner tiles on the matrix cores for the samples within kDcnReach of their tap (plain stores, every
  // pixel once), the rest — and everything, when the shape is outside that kernel's reach — by atomics below
  const bool owned = dcn_owner_launch(grad_columns, offset, mask, grad_x, N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw,
                                      dh, dw, deform_groups, dtype, st);
  const int reach = owned ? 3 : -1;
  switch (dtype) {
    case UBV_F32: hipLaunchKernelGGL((dcn_col2im_kernel<float, 4>), grid, blk, 0, st, (const float*)grad_columns, (const float*)x, (const float*)offset, (const float*)mask, grad_x, grad_offset, grad_mask, g, reach); break;
    case UBV_F16: hipLaunchKernelGGL((dcn_col2im_kernel<f16_t, 4>), grid, blk, 0, st, (const f16_t*)grad_columns, (const f16_t*)x, (const f16_t*)offset, (const f16_t*)mask, grad_x, grad_offset, grad_mask, g, reach); break;
    default: hipLaunchKernelGGL((dcn_col2im_kernel<bf16_t, 4>), grid, blk, 0, st, (const bf16_t*)grad_columns, (const bf16_t*)x, (const bf16_t*)offset, (const bf16_t*)mask, grad_x, grad_offset, grad_mask, g, reach); break;
  }
  UBV_CHECK_LAUNCH("dcn_col2im");
  return UBV_OK;
}
