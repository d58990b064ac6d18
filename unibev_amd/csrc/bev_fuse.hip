// Memory-bound glue on either side of the BEV encoders, each as one pass over HBM:
//   * flatten_embed: NCHW feature map -> token-major rows + camera / level embeddings
//     (UniBEVTransformer._pre_process_img_feats / _pre_process_pts_feats,
//      models/modules/transformer_fusion.py:230-278), LDS-tiled transpose.
//   * bev_fuse: channel (CNW) / spatial weighting, linear | avg | cat fusion and the final
//     (B,Nq,C) -> (Nq,B,C*s) permute (transformer_fusion.py:280-413, 549), 16-B vectors.
#include "ubv_common.h"

namespace ubv {

// ---------------------------------------------------------------------------------- flatten_embed
// grid (ceil(HW/32), ceil(C/32), N), block (32, 8)
template <typename T>
__global__ __launch_bounds__(256) void flatten_embed_fwd_kernel(
    const T* __restrict__ in, const float* __restrict__ embA, int groups,
    const float* __restrict__ embB, T* __restrict__ out, int C, int HW) {
#pragma clang fp contract(off)
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, hw = hw0 + tx;
    tile[k][tx] = (c < C && hw < HW) ? elem<T>::to_float(in[((long)n * C + c) * HW + hw]) : 0.0f;
  }
  __syncthreads();
  const int c = c0 + tx;
  if (c >= C) return;
  const float ea = embA ? embA[(long)(n % groups) * C + c] : 0.0f;
  const float eb = embB ? embB[c] : 0.0f;
  for (int k = ty; k < 32; k += 8) {
    const int hw = hw0 + k;
    if (hw < HW) {
      float v = tile[tx][k];
      if (embA) v = v + ea;
      if (embB) v = v + eb;
      out[((long)n * HW + hw) * C + c] = elem<T>::from_float(v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void flatten_embed_bwd_kernel(
    const T* __restrict__ gout, T* __restrict__ gin, float* __restrict__ gemb, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int k = ty; k < 32; k += 8) {
    const int hw = hw0 + k, c = c0 + tx;
    tile[k][tx] = (c < C && hw < HW) ? elem<T>::to_float(gout[((long)n * HW + hw) * C + c]) : 0.0f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, hw = hw0 + tx;
    if (c < C && hw < HW) gin[((long)n * C + c) * HW + hw] = elem<T>::from_float(tile[tx][k]);
  }
  if (gemb != nullptr && ty == 0) {
    const int c = c0 + tx;
    if (c < C) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 32; ++k) s += tile[k][tx];
      atomic_add_f32(gemb + (long)n * C + c, s);
    }
  }
}

// ---------------------------------------------------------------------------------- bev_fuse
// One wave per (q, b) row; lanes stride the C channels in 16-B vectors.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(
    const T* __restrict__ img, const T* __restrict__ pts, const float* __restrict__ cw_img,
    const float* __restrict__ cw_pts, const float* __restrict__ sw_img,
    const float* __restrict__ sw_pts, T* __restrict__ out, int B, int Nq, int C, int cat) {
#pragma clang fp contract(off)
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= (long)B * Nq) return;
  const int q = (int)(wave / B), b = (int)(wave - (long)q * B);     // output-row order (q, b)
  const long src = ((long)b * Nq + q) * C;
  const long dst = ((long)q * B + b) * (cat ? 2 * C : C);
  const float si = sw_img ? sw_img[q] : 1.0f, sp = sw_pts ? sw_pts[q] : 1.0f;
  for (int c = lane * VEC; c < C; c += 64 * VEC) {
    float a[VEC], p[VEC], r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a[i] = 0.0f; p[i] = 0.0f; }
    if (img) vec_io<T, VEC>::load(img + src + c, a);
    if (pts) vec_io<T, VEC>::load(pts + src + c, p);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      a[i] = a[i] * cw_img[c + i];
      p[i] = p[i] * cw_pts[c + i];
      if (sw_img) a[i] = a[i] * si;
      if (sw_pts) p[i] = p[i] * sp;
      r[i] = a[i] + p[i];
    }
    if (cat) {
      vec_io<T, VEC>::store(out + dst + c, a);
      vec_io<T, VEC>::store(out + dst + C + c, p);
    } else {
      vec_io<T, VEC>::store(out + dst + c, r);
    }
  }
}

constexpr int kFuseMaxChunks = 4;   // C <= 64 * VEC * 4

template <typename T, int VEC>
__global__ __launch_bounds__(256) void fuse_bwd_kernel(
    const T* __restrict__ gout, const T* __restrict__ img, const T* __restrict__ pts,
    const float* __restrict__ cw_img, const float* __restrict__ cw_pts,
    const float* __restrict__ sw_img, const float* __restrict__ sw_pts, T* __restrict__ gimg,
    T* __restrict__ gpts, float* __restrict__ gcw, float* __restrict__ gsw, int B, int Nq, int C,
    int cat, int rows_per_wave) {
  const long wave0 = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * rows_per_wave;
  const int lane = threadIdx.x & 63;
  float acc_i[kFuseMaxChunks][VEC], acc_p[kFuseMaxChunks][VEC];
#pragma unroll
  for (int k = 0; k < kFuseMaxChunks; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) { acc_i[k][i] = 0.0f; acc_p[k][i] = 0.0f; }

  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const long row = wave0 + rr;
    if (row >= (long)B * Nq) break;
    const int q = (int)(row / B), b = (int)(row - (long)q * B);
    const long src = ((long)b * Nq + q) * C;
    const long dst = ((long)q * B + b) * (cat ? 2 * C : C);
    const float si = sw_img ? sw_img[q] : 1.0f, sp = sw_pts ? sw_pts[q] : 1.0f;
    float rs_i = 0.0f, rs_p = 0.0f;
#pragma unroll
    for (int k = 0; k < kFuseMaxChunks; ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (c < C) {
        float gi[VEC], gp[VEC], a[VEC], p[VEC], o[VEC];
        vec_io<T, VEC>::load(gout + dst + c, gi);
        if (cat) vec_io<T, VEC>::load(gout + dst + C + c, gp);
        else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) gp[i] = gi[i];
        }
        if (img) {
          vec_io<T, VEC>::load(img + src + c, a);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            acc_i[k][i] = fmaf(gi[i] * a[i], si, acc_i[k][i]);
            rs_i = fmaf(gi[i] * a[i], cw_img[c + i], rs_i);
            o[i] = gi[i] * cw_img[c + i] * si;
          }
          if (gimg) vec_io<T, VEC>::store(gimg + src + c, o);
        }
        if (pts) {
          vec_io<T, VEC>::load(pts + src + c, p);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            acc_p[k][i] = fmaf(gp[i] * p[i], sp, acc_p[k][i]);
            rs_p = fmaf(gp[i] * p[i], cw_pts[c + i], rs_p);
            o[i] = gp[i] * cw_pts[c + i] * sp;
          }
          if (gpts) vec_io<T, VEC>::store(gpts + src + c, o);
        }
      }
    }
    if (gsw != nullptr) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        rs_i += __shfl_xor(rs_i, m, 64);
        rs_p += __shfl_xor(rs_p, m, 64);
      }
      if (lane == 0) {
        atomic_add_f32(gsw + q, rs_i);
        atomic_add_f32(gsw + Nq + q, rs_p);
      }
    }
  }
  if (gcw != nullptr) {
#pragma unroll
    for (int k = 0; k < kFuseMaxChunks; ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (c < C) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          atomic_add_f32(gcw + c + i, acc_i[k][i]);
          atomic_add_f32(gcw + C + c + i, acc_p[k][i]);
        }
      }
    }
  }
}

// Same gradients for the common widths where one row needs at most one wave (C/VEC divides 64,
// e.g. C = 256: 64 f32 lanes or 32 16-bit lanes): 64/LPR rows per wave step so no lane idles, two
// steps in flight, and the channel-weight partial sums of a block are reduced in LDS before ONE
// atomic per column per block (the chunked kernel above: 380 us for 80 000 x 256 bf16 rows,
// 1250 waves each walking 64 rows with half its lanes off).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fuse_bwd_rows_kernel(
    const T* __restrict__ gout, const T* __restrict__ img, const T* __restrict__ pts,
    const float* __restrict__ cw_img, const float* __restrict__ cw_pts,
    const float* __restrict__ sw_img, const float* __restrict__ sw_pts, T* __restrict__ gimg,
    T* __restrict__ gpts, float* __restrict__ gcw, float* __restrict__ gsw, int B, int Nq, int C,
    int cat, int rows_per_wave) {
  __shared__ float red[2][4][64][VEC];
  const int lpr = C / VEC, G = 64 / lpr;             // lanes per row, rows per wave step
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane / lpr, c = (lane - sub * lpr) * VEC;
  const long rows = (long)B * Nq;
  const long wave0 = ((long)blockIdx.x * 4 + wv) * rows_per_wave;
  float acc_i[VEC], acc_p[VEC], ci[VEC], cp[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc_i[i] = 0.0f; acc_p[i] = 0.0f;
    ci[i] = cw_img[c + i]; cp[i] = cw_pts[c + i];
  }
  for (int r0 = 0; r0 < rows_per_wave; r0 += G) {
    const long row = wave0 + r0 + sub;
    const bool ok = (r0 + sub < rows_per_wave) && row < rows;
    const long rc = ok ? row : 0;
    const int q = (int)(rc / B), b = (int)(rc - (long)q * B);
    const long src = ((long)b * Nq + q) * C + c;
    const long dst = ((long)q * B + b) * (cat ? 2 * C : C) + c;
    const float si = sw_img ? sw_img[q] : 1.0f, sp = sw_pts ? sw_pts[q] : 1.0f;
    float gi[VEC], gp[VEC], a[VEC], p[VEC], o[VEC];
    vec_io<T, VEC>::load(gout + dst, gi);
    if (cat) vec_io<T, VEC>::load(gout + dst + C, gp);
    if (img) vec_io<T, VEC>::load(img + src, a);
    if (pts) vec_io<T, VEC>::load(pts + src, p);
    const float live = ok ? 1.0f : 0.0f;
    float rs_i = 0.0f, rs_p = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      gi[i] *= live;
      gp[i] = cat ? gp[i] * live : gi[i];
    }
    if (img) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc_i[i] = fmaf(gi[i] * a[i], si, acc_i[i]);
        rs_i = fmaf(gi[i] * a[i], ci[i], rs_i);
        o[i] = gi[i] * ci[i] * si;
      }
      if (gimg && ok) vec_io<T, VEC>::store(gimg + src, o);
    }
    if (pts) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc_p[i] = fmaf(gp[i] * p[i], sp, acc_p[i]);
        rs_p = fmaf(gp[i] * p[i], cp[i], rs_p);
        o[i] = gp[i] * cp[i] * sp;
      }
      if (gpts && ok) vec_io<T, VEC>::store(gpts + src, o);
    }
    if (gsw != nullptr) {
      for (int m = lpr >> 1; m >= 1; m >>= 1) {
        rs_i += __shfl_xor(rs_i, m, 64);
        rs_p += __shfl_xor(rs_p, m, 64);
      }
      if (ok && c == 0) {
        atomic_add_f32(gsw + q, rs_i);
        atomic_add_f32(gsw + Nq + q, rs_p);
      }
    }
  }
  if (gcw != nullptr) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[0][wv][lane][i] = acc_i[i]; red[1][wv][lane][i] = acc_p[i]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
      const int which = t / C, col = t - which * C;
      const int cl = col / VEC, e = col - cl * VEC;
      float sum = 0.0f;
      for (int w = 0; w < 4; ++w)
        for (int g = 0; g < G; ++g) sum += red[which][w][g * lpr + cl][e];
      atomic_add_f32(gcw + which * C + col, sum);
    }
  }
}

template <typename T>
static int flatten_launch(bool bwd, const void* a0, const float* embA, int groups,
                          const float* embB, void* a1, float* gemb, int N, int C, int HW,
                          hipStream_t st) {
  dim3 grid((HW + 31) / 32, (C + 31) / 32, N), block(32, 8);
  if (!bwd)
    hipLaunchKernelGGL((flatten_embed_fwd_kernel<T>), grid, block, 0, st, (const T*)a0, embA,
                       groups, embB, (T*)a1, C, HW);
  else
    hipLaunchKernelGGL((flatten_embed_bwd_kernel<T>), grid, block, 0, st, (const T*)a0, (T*)a1,
                       gemb, C, HW);
  return 0;
}

}  // namespace ubv

extern "C" int ubv_flatten_embed_forward(const void* in, const float* embA, int groups,
                                         const float* embB, void* out, int N, int C, int HW,
                                         int dtype, void* stream) {
  UBV_CHECK_ARG(in && out, "flatten_embed_forward: null pointer");
  UBV_CHECK_ARG(N > 0 && C > 0 && HW > 0 && N <= 65535, "flatten_embed_forward: bad dimension");
  UBV_CHECK_ARG(embA == nullptr || groups > 0, "flatten_embed_forward: groups must be positive");
  hipStream_t st = ubv::as_stream(stream);
  switch (dtype) {
    case UBV_F32: ubv::flatten_launch<float>(false, in, embA, groups, embB, out, nullptr, N, C, HW, st); break;
    case UBV_F16: ubv::flatten_launch<ubv::f16_t>(false, in, embA, groups, embB, out, nullptr, N, C, HW, st); break;
    case UBV_BF16: ubv::flatten_launch<ubv::bf16_t>(false, in, embA, groups, embB, out, nullptr, N, C, HW, st); break;
    default: ubv::set_error("flatten_embed_forward: unknown dtype %d", dtype); return UBV_ERR_INVALID;
  }
  UBV_CHECK_LAUNCH("flatten_embed_forward");
  return UBV_OK;
}

extern "C" int ubv_flatten_embed_backward(const void* grad_out, void* grad_in, float* grad_emb,
                                          int N, int C, int HW, int dtype, void* stream) {
  UBV_CHECK_ARG(grad_out && grad_in, "flatten_embed_backward: null pointer");
  UBV_CHECK_ARG(N > 0 && C > 0 && HW > 0 && N <= 65535, "flatten_embed_backward: bad dimension");
  hipStream_t st = ubv::as_stream(stream);
  switch (dtype) {
    case UBV_F32: ubv::flatten_launch<float>(true, grad_out, nullptr, 1, nullptr, grad_in, grad_emb, N, C, HW, st); break;
    case UBV_F16: ubv::flatten_launch<ubv::f16_t>(true, grad_out, nullptr, 1, nullptr, grad_in, grad_emb, N, C, HW, st); break;
    case UBV_BF16: ubv::flatten_launch<ubv::bf16_t>(true, grad_out, nullptr, 1, nullptr, grad_in, grad_emb, N, C, HW, st); break;
    default: ubv::set_error("flatten_embed_backward: unknown dtype %d", dtype); return UBV_ERR_INVALID;
  }
  UBV_CHECK_LAUNCH("flatten_embed_backward");
  return UBV_OK;
}

namespace ubv {
template <typename T>
static void fuse_launch(bool bwd, const void* gout, const void* img, const void* pts,
                        const float* cwi, const float* cwp, const float* swi, const float* swp,
                        void* o0, void* o1, float* gcw, float* gsw, int B, int Nq, int C, int cat,
                        hipStream_t st) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  const long rows = (long)B * Nq;
  if (!bwd) {
    const int blocks = (int)((rows + 3) / 4);
    hipLaunchKernelGGL((fuse_fwd_kernel<T, VEC>), dim3(blocks), dim3(256), 0, st, (const T*)img,
                       (const T*)pts, cwi, cwp, swi, swp, (T*)o0, B, Nq, C, cat);
  } else {
    const int lpr = C / VEC;
    if (C % VEC == 0 && lpr <= 64 && 64 % lpr == 0) {
      const int rpw = 32;
      const long waves = (rows + rpw - 1) / rpw;
      const int blocks = (int)((waves + 3) / 4);
      hipLaunchKernelGGL((fuse_bwd_rows_kernel<T, VEC>), dim3(blocks), dim3(256), 0, st,
                         (const T*)gout, (const T*)img, (const T*)pts, cwi, cwp, swi, swp, (T*)o0,
                         (T*)o1, gcw, gsw, B, Nq, C, cat, rpw);
      return;
    }
    const int rpw = 64;
    const long waves = (rows + rpw - 1) / rpw;
    const int blocks = (int)((waves + 3) / 4);
    hipLaunchKernelGGL((fuse_bwd_kernel<T, VEC>), dim3(blocks), dim3(256), 0, st, (const T*)gout,
                       (const T*)img, (const T*)pts, cwi, cwp, swi, swp, (T*)o0, (T*)o1, gcw, gsw,
                       B, Nq, C, cat, rpw);
  }
}
}  // namespace ubv

static int fuse_check(int B, int Nq, int C, int dtype, const char* who) {
  UBV_CHECK_ARG(B > 0 && Nq > 0 && C > 0, "%s: non-positive dimension", who);
  const int vec = dtype == UBV_F32 ? 4 : 8;
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "%s: unknown dtype %d", who, dtype);
  UBV_CHECK_ARG(C % vec == 0 && C <= 64 * vec * ubv::kFuseMaxChunks,
                "%s: C=%d must be a multiple of %d and <= %d", who, C, vec,
                64 * vec * ubv::kFuseMaxChunks);
  return UBV_OK;
}

extern "C" int ubv_bev_fuse_forward(const void* img, const void* pts, const float* cw_img,
                                    const float* cw_pts, const float* sw_img, const float* sw_pts,
                                    void* out, int B, int Nq, int C, int cat, int dtype,
                                    void* stream) {
  UBV_CHECK_ARG((img || pts) && cw_img && cw_pts && out, "bev_fuse_forward: null pointer");
  int rc = fuse_check(B, Nq, C, dtype, "bev_fuse_forward");
  if (rc) return rc;
  hipStream_t st = ubv::as_stream(stream);
  switch (dtype) {
    case UBV_F32: ubv::fuse_launch<float>(false, nullptr, img, pts, cw_img, cw_pts, sw_img, sw_pts, out, nullptr, nullptr, nullptr, B, Nq, C, cat, st); break;
    case UBV_F16: ubv::fuse_launch<ubv::f16_t>(false, nullptr, img, pts, cw_img, cw_pts, sw_img, sw_pts, out, nullptr, nullptr, nullptr, B, Nq, C, cat, st); break;
    default: ubv::fuse_launch<ubv::bf16_t>(false, nullptr, img, pts, cw_img, cw_pts, sw_img, sw_pts, out, nullptr, nullptr, nullptr, B, Nq, C, cat, st); break;
  }
  UBV_CHECK_LAUNCH("bev_fuse_forward");
  return UBV_OK;
}

extern "C" int ubv_bev_fuse_backward(const void* grad_out, const void* img, const void* pts,
                                     const float* cw_img, const float* cw_pts, const float* sw_img,
                                     const float* sw_pts, void* grad_img, void* grad_pts,
                                     float* grad_cw, float* grad_sw, int B, int Nq, int C, int cat,
                                     int dtype, void* stream) {
  UBV_CHECK_ARG(grad_out && cw_img && cw_pts, "bev_fuse_backward: null pointer");
  int rc = fuse_check(B, Nq, C, dtype, "bev_fuse_backward");
  if (rc) return rc;
  hipStream_t st = ubv::as_stream(stream);
  switch (dtype) {
    case UBV_F32: ubv::fuse_launch<float>(true, grad_out, img, pts, cw_img, cw_pts, sw_img, sw_pts, grad_img, grad_pts, grad_cw, grad_sw, B, Nq, C, cat, st); break;
    case UBV_F16: ubv::fuse_launch<ubv::f16_t>(true, grad_out, img, pts, cw_img, cw_pts, sw_img, sw_pts, grad_img, grad_pts, grad_cw, grad_sw, B, Nq, C, cat, st); break;
    default: ubv::fuse_launch<ubv::bf16_t>(true, grad_out, img, pts, cw_img, cw_pts, sw_img, sw_pts, grad_img, grad_pts, grad_cw, grad_sw, B, Nq, C, cat, st); break;
  }
  UBV_CHECK_LAUNCH("bev_fuse_backward");
  return UBV_OK;
}
