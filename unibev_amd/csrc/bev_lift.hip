// Fused BEV query lifting (one feature level): Linear outputs -> softmax / sampling locations ->
// bilinear gather -> per-camera accumulation -> camera mean, in ONE kernel, forward and backward.
//
// What is fused away relative to the reference (spatial_cross_attention_img.py:141-212, 385-435;
// spatial_cross_attention_pts.py:383-442; decoder.py:299-327):
//   * sampling_locations [B,Nq,H,1,P,2] and attention_weights [B,Nq,H,1,P] (15-30 MB fp32 per
//     call) are never written: each (query, head) lane group turns its raw offsets / logits into
//     locations / softmax weights in registers.
//   * SCA-img: no nonzero() host syncs, no zero-padded per-camera re-batch, no scatter-add.  The
//     offsets / logits of a query do not depend on the camera, so one Linear over all Nq queries
//     replaces the Linear over 6 x max_len padded rows; the kernel loops over the cameras a query
//     is visible in (visibility of batch element 0, quirk q1), accumulates in camera order (the
//     order of the reference's `slots[j, idx] += ...`) and divides by the per-sample count (q2).
//   * backward: offsets are shared by all cameras, so d(offset) / d(logit) are accumulated in
//     registers across the camera loop and reduced over the Dh lanes with wave shuffles — no
//     atomics except on grad_value.
//
// Thread map: as k1 (Dh channels of a (query, head) on LP = Dh/VEC adjacent lanes, 16-B vectors, a
// wave64 holds 64/(H*LP) queries).  Work decomposition: a 256-thread block owns an 8x8 tile of
// the BEV query grid (neighbouring queries sample neighbouring pixels: L1/L2 reuse), and tiles
// are dealt to XCDs in contiguous bands so a band's value rows stay in one XCD's L2.
#include "ubv_common.h"

namespace ubv {

struct LiftArgs {
  const void* value; const float* offsets; long off_stride; const float* logits; long log_stride;
  const float* ref; const uint8_t* vis0; const float* count;
  void* out;                                     // fwd
  const void* gout; float* gvalue; float* goff; long goff_stride; float* glog; long glog_stride;
  int B, Nc, fh, fw, H, Nq, Z, qw, qh, tiles_x, tiles_per_sample, total_tiles, chunk;
};

// Decode (b, q, valid) of the query this lane works on in iteration `it`.
__device__ __forceinline__ bool lift_query(const LiftArgs& a, int item, int li, int& b, int& q) {
  b = item / a.tiles_per_sample;
  const int tile = item - b * a.tiles_per_sample;
  if (a.qw > 0) {
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int qy = ty * 8 + (li >> 3), qx = tx * 8 + (li & 7);
    q = qy * a.qw + qx;
    return qy < a.qh && qx < a.qw;
  }
  q = tile * 64 + li;
  return q < a.Nq;
}

template <int P>
__device__ __forceinline__ void load_row(const float* p, float (&v)[P]) {
#pragma unroll
  for (int i = 0; i < P; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
}
template <int P>
__device__ __forceinline__ void store_row(float* p, const float (&v)[P]) {
#pragma unroll
  for (int i = 0; i < P; i += 4)
    *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}

template <int P>
__device__ __forceinline__ void softmax_row(const float (&l)[P], float (&w)[P]) {
  float m = l[0];
#pragma unroll
  for (int i = 1; i < P; ++i) m = fmaxf(m, l[i]);
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < P; ++i) { w[i] = expf(l[i] - m); s += w[i]; }
#pragma unroll
  for (int i = 0; i < P; ++i) w[i] = w[i] / s;
}

template <typename T, int DH, int VEC, int P>
__global__ __launch_bounds__(256) void lift_fwd_kernel(const LiftArgs a) {
  constexpr int LP = DH / VEC;
  const int item = xcd_remap(blockIdx.x, a.chunk);
  if (item >= a.total_tiles) return;
  const int LQ = a.H * LP, QW = kWave / LQ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cg = lane % LP, h = (lane / LP) % a.H, sub = lane / LQ;
  const int S = a.fh * a.fw;
  const long row = (long)a.H * DH;
  const T* __restrict__ value = (const T*)a.value;
  T* __restrict__ out = (T*)a.out;
  const float fwf = (float)a.fw, fhf = (float)a.fh;

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    if (!lift_query(a, item, li0 + wv * QW + sub, b, q)) continue;
    const long bq = (long)b * a.Nq + q;
    float lg[P], w[P], off[2 * P];
    load_row<P>(a.logits + bq * a.log_stride + h * P, lg);
    load_row<2 * P>(a.offsets + bq * a.off_stride + h * 2 * P, off);
    softmax_row<P>(lg, w);

    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;

    for (int cam = 0; cam < a.Nc; ++cam) {
      if (a.vis0 != nullptr && a.vis0[(long)cam * a.Nq + q] == 0) continue;
      const float* rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
      const T* vb = value + ((long)b * a.Nc + cam) * S * row + h * DH + cg * VEC;
      int zi = 0;                                   // anchor of flat point p is p % Z (quirk q3)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float2 r = *reinterpret_cast<const float2*>(rp + zi * 2);
        zi = (zi + 1 == a.Z) ? 0 : zi + 1;
        const float lx = r.x + off[2 * p] / fwf;
        const float ly = r.y + off[2 * p + 1] / fhf;
        const Footprint f = make_footprint(lx, ly, a.fh, a.fw);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[VEC];
          vec_io<T, VEC>::load(vb + (long)f.idx[k] * row, v);
          const float c = w[p] * f.w[k];
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
        }
      }
    }
    if (a.count != nullptr) {
      const float cnt = a.count[bq];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = acc[i] / cnt;
    }
    vec_io<T, VEC>::store(out + bq * row + h * DH + cg * VEC, acc);
  }
}

template <typename T, int DH, int VEC, int P>
__global__ __launch_bounds__(256) void lift_bwd_kernel(const LiftArgs a) {
  constexpr int LP = DH / VEC;
  const int item = xcd_remap(blockIdx.x, a.chunk);
  if (item >= a.total_tiles) return;
  const int LQ = a.H * LP, QW = kWave / LQ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cg = lane % LP, h = (lane / LP) % a.H, sub = lane / LQ;
  const int S = a.fh * a.fw;
  const long row = (long)a.H * DH;
  const T* __restrict__ value = (const T*)a.value;
  const T* __restrict__ gout = (const T*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    const bool valid = lift_query(a, item, li0 + wv * QW + sub, b, q);
    if (!valid) { b = 0; q = 0; }             // keep every lane in the shuffles below
    const long bq = (long)b * a.Nq + q;
    float lg[P], w[P], off[2 * P];
    load_row<P>(a.logits + bq * a.log_stride + h * P, lg);
    load_row<2 * P>(a.offsets + bq * a.off_stride + h * 2 * P, off);
    softmax_row<P>(lg, w);

    float go[VEC];
    vec_io<T, VEC>::load(gout + bq * row + h * DH + cg * VEC, go);
    const float inv = valid ? 1.0f : 0.0f;
    const float cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) go[i] = (a.count != nullptr ? go[i] / cnt : go[i]) * inv;

    float gw[P], gx[P], gy[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { gw[p] = 0.0f; gx[p] = 0.0f; gy[p] = 0.0f; }

    for (int cam = 0; cam < a.Nc; ++cam) {
      if (a.vis0 != nullptr && a.vis0[(long)cam * a.Nq + q] == 0) continue;
      const float* rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
      const long vo = ((long)b * a.Nc + cam) * S * row + h * DH + cg * VEC;
      int zi = 0;                                   // anchor of flat point p is p % Z (quirk q3)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float2 r = *reinterpret_cast<const float2*>(rp + zi * 2);
        zi = (zi + 1 == a.Z) ? 0 : zi + 1;
        const float lx = r.x + off[2 * p] / fwf;
        const float ly = r.y + off[2 * p + 1] / fhf;
        const Footprint f = make_footprint(lx, ly, a.fh, a.fw);
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[VEC];
          const long o = vo + (long)f.idx[k] * row;
          vec_io<T, VEC>::load(value + o, v);
          float d = 0.0f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) d = fmaf(go[i], v[i], d);
          dot[k] = d * f.m[k];
          const float c = w[p] * f.w[k];
          if (valid && c != 0.0f) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) atomic_add_f32(a.gvalue + o + i, c * go[i]);
          }
        }
        const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
        gw[p] += hy * hx * dot[0] + hy * f.lx * dot[1] + f.ly * hx * dot[2] + f.ly * f.lx * dot[3];
        gx[p] += (dot[1] - dot[0]) * hy + (dot[3] - dot[2]) * f.ly;
        gy[p] += (dot[2] - dot[0]) * hx + (dot[3] - dot[1]) * f.lx;
      }
    }
    // reduce the Dh partial dot products over the LP lanes of this (query, head)
#pragma unroll
    for (int m = 1; m < LP; m <<= 1) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        gw[p] += __shfl_xor(gw[p], m, 64);
        gx[p] += __shfl_xor(gx[p], m, 64);
        gy[p] += __shfl_xor(gy[p], m, 64);
      }
    }
    if (valid && cg == 0) {
      // softmax backward: dlogit_p = w_p * (gw_p - sum_k w_k gw_k)
      float s = 0.0f;
#pragma unroll
      for (int p = 0; p < P; ++p) s = fmaf(w[p], gw[p], s);
      float gl[P], gofs[2 * P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        gl[p] = w[p] * (gw[p] - s);
        gofs[2 * p] = (w[p] * gx[p] * fwf) / fwf;      // d loc = w*g*W ; d off = d loc / W
        gofs[2 * p + 1] = (w[p] * gy[p] * fhf) / fhf;
      }
      store_row<P>(a.glog + bq * a.glog_stride + h * P, gl);
      store_row<2 * P>(a.goff + bq * a.goff_stride + h * 2 * P, gofs);
    }
  }
}

// ---- dispatch ----------------------------------------------------------------------------------------
template <typename T, int DH, int P>
static void lift_launch(const LiftArgs& a, bool bwd, hipStream_t st) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  const int blocks = 8 * a.chunk;
  if (!bwd)
    hipLaunchKernelGGL((lift_fwd_kernel<T, DH, VEC, P>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((lift_bwd_kernel<T, DH, VEC, P>), dim3(blocks), dim3(256), 0, st, a);
}

template <typename T>
static bool lift_dispatch_T(const LiftArgs& a, int Dh, int P, bool bwd, hipStream_t st) {
  if (Dh == 32 && P == 8) { lift_launch<T, 32, 8>(a, bwd, st); return true; }
  if (Dh == 32 && P == 4) { lift_launch<T, 32, 4>(a, bwd, st); return true; }
  if (Dh == 16 && P == 8) { lift_launch<T, 16, 8>(a, bwd, st); return true; }
  if (Dh == 16 && P == 4) { lift_launch<T, 16, 4>(a, bwd, st); return true; }
  return false;
}

static bool lift_shape_ok(int H, int Dh, int P, int dtype) {
  if (dtype < 0 || dtype > 2) return false;
  if (!(Dh == 32 || Dh == 16) || !(P == 4 || P == 8)) return false;
  const int vec = (dtype == UBV_F32) ? 4 : 8;
  const int lq = H * (Dh / vec);
  return lq >= 4 && lq <= 64 && (64 % lq) == 0;
}

static int lift_run(LiftArgs a, int Dh, int P, int dtype, bool bwd, void* stream) {
  UBV_CHECK_ARG(a.B > 0 && a.Nc > 0 && a.fh > 0 && a.fw > 0 && a.H > 0 && a.Nq > 0 && a.Z > 0,
                "bev_lift: non-positive dimension");
  UBV_CHECK_ARG(P % a.Z == 0, "bev_lift: num_points %d not a multiple of Z %d", P, a.Z);
  if (!lift_shape_ok(a.H, Dh, P, dtype)) {
    set_error("bev_lift: no kernel for H=%d Dh=%d P=%d dtype=%d", a.H, Dh, P, dtype);
    return UBV_ERR_UNSUPPORTED;
  }
  UBV_CHECK_ARG((a.off_stride % 4) == 0 && (a.log_stride % 4) == 0 &&
                    ((uintptr_t)a.offsets % 16) == 0 && ((uintptr_t)a.logits % 16) == 0,
                "bev_lift: offsets/logits rows must be 16-byte aligned");
  UBV_CHECK_ARG(((uintptr_t)a.ref % 8) == 0, "bev_lift: ref must be 8-byte aligned");
  if (bwd)
    UBV_CHECK_ARG((a.goff_stride % 4) == 0 && (a.glog_stride % 4) == 0 &&
                      ((uintptr_t)a.goff % 16) == 0 && ((uintptr_t)a.glog % 16) == 0,
                  "bev_lift: grad rows must be 16-byte aligned");
  if (a.qw > 0 && (long)a.qw * a.qh == a.Nq) {
    a.tiles_x = (a.qw + 7) / 8;
    a.tiles_per_sample = a.tiles_x * ((a.qh + 7) / 8);
  } else {
    a.qw = a.qh = 0;
    a.tiles_x = 0;
    a.tiles_per_sample = (a.Nq + 63) / 64;
  }
  a.total_tiles = a.B * a.tiles_per_sample;
  a.chunk = (a.total_tiles + 7) / 8;
  hipStream_t st = as_stream(stream);
  bool ok = false;
  switch (dtype) {
    case UBV_F32: ok = lift_dispatch_T<float>(a, Dh, P, bwd, st); break;
    case UBV_F16: ok = lift_dispatch_T<f16_t>(a, Dh, P, bwd, st); break;
    case UBV_BF16: ok = lift_dispatch_T<bf16_t>(a, Dh, P, bwd, st); break;
  }
  if (!ok) { set_error("bev_lift: dispatch failed"); return UBV_ERR_UNSUPPORTED; }
  UBV_CHECK_LAUNCH(bwd ? "bev_lift_backward" : "bev_lift_forward");
  return UBV_OK;
}

}  // namespace ubv

extern "C" int ubv_bev_lift_supported(int H, int Dh, int P, int dtype) {
  return ubv::lift_shape_ok(H, Dh, P, dtype) ? 1 : 0;
}

extern "C" int ubv_bev_lift_forward(const void* value, const float* offsets, int64_t off_stride,
                                    const float* logits, int64_t log_stride, const float* ref,
                                    const uint8_t* vis0, const float* count, void* out, int B,
                                    int Nc, int fh, int fw, int H, int Dh, int Nq, int P, int Z,
                                    int qgrid_w, int qgrid_h, int dtype, void* stream) {
  UBV_CHECK_ARG(value && offsets && logits && ref && out, "bev_lift_forward: null pointer");
  ubv::LiftArgs a{};
  a.value = value; a.offsets = offsets; a.off_stride = off_stride; a.logits = logits;
  a.log_stride = log_stride; a.ref = ref; a.vis0 = vis0; a.count = count; a.out = out;
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = Z; a.qw = qgrid_w;
  a.qh = qgrid_h;
  return ubv::lift_run(a, Dh, P, dtype, false, stream);
}

extern "C" int ubv_bev_lift_backward(const void* value, const float* offsets, int64_t off_stride,
                                     const float* logits, int64_t log_stride, const float* ref,
                                     const uint8_t* vis0, const float* count, const void* grad_out,
                                     float* grad_value, float* grad_offsets, int64_t goff_stride,
                                     float* grad_logits, int64_t glog_stride, int B, int Nc, int fh,
                                     int fw, int H, int Dh, int Nq, int P, int Z, int qgrid_w,
                                     int qgrid_h, int dtype, void* stream) {
  UBV_CHECK_ARG(value && offsets && logits && ref && grad_out && grad_value && grad_offsets &&
                    grad_logits, "bev_lift_backward: null pointer");
  ubv::LiftArgs a{};
  a.value = value; a.offsets = offsets; a.off_stride = off_stride; a.logits = logits;
  a.log_stride = log_stride; a.ref = ref; a.vis0 = vis0; a.count = count; a.gout = grad_out;
  a.gvalue = grad_value; a.goff = grad_offsets; a.goff_stride = goff_stride; a.glog = grad_logits;
  a.glog_stride = glog_stride;
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = Z; a.qw = qgrid_w;
  a.qh = qgrid_h;
  return ubv::lift_run(a, Dh, P, dtype, true, stream);
}
