// k1 — multi-scale deformable attention sampling, forward and backward, at the mmcv operator
// boundary (any L, P; explicit sampling locations and attention weights).
//
// Thread map (fast path): Dh channels of one (query, head) are spread over LP = Dh/VEC adjacent
// lanes, VEC channels (16 B, or 8 B for Dh < one 16-B vector) per lane, so each bilinear corner is
// ONE coalesced row-segment load per (query, head) and a wave64 covers 64/(H*LP) whole queries.
// Locations and weights of a (query, head) are contiguous and are read as the same address by its
// LP lanes (one broadcast request).  fp32 accumulation; value/out in f32, f16 or bf16.
// Backward: grad_value by hardware f32 atomics (global_atomic_add_f32); grad_loc / grad_weight are
// per-(query, head, point) dot products over Dh reduced across the LP lanes with wave shuffles.
//
// A one-thread-per-(b,q,h) fallback covers shapes the lane map cannot tile (H*LP not dividing 64).
#include "ubv_common.h"

namespace ubv {

constexpr int kMaxLevels = 8;

struct LevelTable {
  int h[kMaxLevels];
  int w[kMaxLevels];
  int start[kMaxLevels];
};

__device__ __forceinline__ void load_levels(const int64_t* __restrict__ ss,
                                            const int64_t* __restrict__ ls, int L, LevelTable& t) {
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l < L) {
      t.h[l] = (int)ss[2 * l];
      t.w[l] = (int)ss[2 * l + 1];
      t.start[l] = (int)ls[l];
    }
  }
}

// -------------------------------------------------------------------------------------------------
template <typename T, int DH, int VEC>
__global__ __launch_bounds__(256) void k1_fwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ ss, const int64_t* __restrict__ ls,
    const float* __restrict__ loc, const float* __restrict__ aw, T* __restrict__ out, int B, int S,
    int H, int L, int Nq, int P) {
  constexpr int LP = DH / VEC;
  const int LQ = H * LP;                     // lanes per query
  const int QW = kWave / LQ;                 // queries per wave
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int cg = lane % LP;
  const int h = (lane / LP) % H;
  const long bq = wave * QW + lane / LQ;     // flat (b, q)
  if (bq >= (long)B * Nq) return;
  const int b = (int)(bq / Nq);

  const float* lp = loc + (bq * H + h) * (long)L * P * 2;
  const float* wp = aw + (bq * H + h) * (long)L * P;
  const T* vb = value + (long)b * S * H * DH + h * DH + cg * VEC;

  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;

  for (int l = 0; l < L; ++l) {
    const int Hh = (int)ss[2 * l], Ww = (int)ss[2 * l + 1];
    const T* vl = vb + (long)ls[l] * H * DH;
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
      const float a = wp[l * P + p];
      const Footprint f = make_footprint(xy.x, xy.y, Hh, Ww);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[VEC];
        vec_io<T, VEC>::load(vl + (long)f.idx[k] * H * DH, v);
        const float c = a * f.w[k];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
      }
    }
  }
  vec_io<T, VEC>::store(out + bq * H * DH + h * DH + cg * VEC, acc);
}

// GV = false: grad_loc / grad_weight only (grad_value comes from the owner tiles, ubv_ms_deform_attn_backward_planned)
template <typename T, int DH, int VEC, bool GV = true>
__global__ __launch_bounds__(256) void k1_bwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ ss, const int64_t* __restrict__ ls,
    const float* __restrict__ loc, const float* __restrict__ aw, const T* __restrict__ gout,
    float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gaw, int B, int S,
    int H, int L, int Nq, int P) {
  constexpr int LP = DH / VEC;
  const int LQ = H * LP;
  const int QW = kWave / LQ;
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int cg = lane % LP;
  const int h = (lane / LP) % H;
  const long bq_raw = wave * QW + lane / LQ;
  const bool active = bq_raw < (long)B * Nq;
  const long bq = active ? bq_raw : 0;       // inactive lanes still join the shuffles
  const int b = (int)(bq / Nq);

  const float* lp = loc + (bq * H + h) * (long)L * P * 2;
  const float* wp = aw + (bq * H + h) * (long)L * P;
  const long voff = (long)b * S * H * DH + h * DH + cg * VEC;

  float go[VEC];
  vec_io<T, VEC>::load(gout + bq * H * DH + h * DH + cg * VEC, go);
  if (!active) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) go[i] = 0.0f;
  }

  for (int l = 0; l < L; ++l) {
    const int Hh = (int)ss[2 * l], Ww = (int)ss[2 * l + 1];
    const long lvl_off = voff + (long)ls[l] * H * DH;
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + (l * P + p) * 2);
      const float a = wp[l * P + p];
      const Footprint f = make_footprint(xy.x, xy.y, Hh, Ww);
      float dot[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[VEC];
        const long o = lvl_off + (long)f.idx[k] * H * DH;
        vec_io<T, VEC>::load(value + o, v);
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) d = fmaf(go[i], v[i], d);
        dot[k] = d * f.m[k];                       // corners outside the map read as zero
        const float c = a * f.w[k];
        if (GV && active && c != 0.0f) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) atomic_add_f32(gvalue + o + i, c * go[i]);
        }
      }
      const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
      float g_w = hy * hx * dot[0] + hy * f.lx * dot[1] + f.ly * hx * dot[2] + f.ly * f.lx * dot[3];
      float g_x = (dot[1] - dot[0]) * hy + (dot[3] - dot[2]) * f.ly;
      float g_y = (dot[2] - dot[0]) * hx + (dot[3] - dot[1]) * f.lx;
#pragma unroll
      for (int m = 1; m < LP; m <<= 1) {
        g_w += __shfl_xor(g_w, m, 64);
        g_x += __shfl_xor(g_x, m, 64);
        g_y += __shfl_xor(g_y, m, 64);
      }
      if (active && cg == 0) {
        const long pi = (bq * H + h) * (long)L * P + l * P + p;
        gaw[pi] = g_w;
        gloc[2 * pi] = a * g_x * (float)Ww;
        gloc[2 * pi + 1] = a * g_y * (float)Hh;
      }
    }
  }
}

// ---- shape-agnostic fallback: one thread per (b, q, h), loops over channels -------------------------
template <typename T>
__global__ void k1_fwd_any_kernel(const T* __restrict__ value, const int64_t* __restrict__ ss,
                                  const int64_t* __restrict__ ls, const float* __restrict__ loc,
                                  const float* __restrict__ aw, T* __restrict__ out, int B, int S,
                                  int H, int Dh, int L, int Nq, int P) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * Nq * H) return;
  const int h = (int)(t % H);
  const long bq = t / H;
  const int b = (int)(bq / Nq);
  for (int c = 0; c < Dh; ++c) {
    float acc = 0.0f;
    for (int l = 0; l < L; ++l) {
      const int Hh = (int)ss[2 * l], Ww = (int)ss[2 * l + 1];
      const T* vl = value + ((long)b * S + ls[l]) * H * Dh + h * Dh + c;
      for (int p = 0; p < P; ++p) {
        const long pi = t * L * P + l * P + p;
        const Footprint f = make_footprint(loc[2 * pi], loc[2 * pi + 1], Hh, Ww);
        float s = 0.0f;
        for (int k = 0; k < 4; ++k) s += f.w[k] * elem<T>::to_float(vl[(long)f.idx[k] * H * Dh]);
        acc = fmaf(aw[pi], s, acc);
      }
    }
    out[bq * H * Dh + h * Dh + c] = elem<T>::from_float(acc);
  }
}

template <typename T>
__global__ void k1_bwd_any_kernel(const T* __restrict__ value, const int64_t* __restrict__ ss,
                                  const int64_t* __restrict__ ls, const float* __restrict__ loc,
                                  const float* __restrict__ aw, const T* __restrict__ gout,
                                  float* __restrict__ gvalue, float* __restrict__ gloc,
                                  float* __restrict__ gaw, int B, int S, int H, int Dh, int L,
                                  int Nq, int P) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * Nq * H) return;
  const int h = (int)(t % H);
  const long bq = t / H;
  const int b = (int)(bq / Nq);
  for (int l = 0; l < L; ++l) {
    const int Hh = (int)ss[2 * l], Ww = (int)ss[2 * l + 1];
    const long base = ((long)b * S + ls[l]) * H * Dh + h * Dh;
    for (int p = 0; p < P; ++p) {
      const long pi = t * L * P + l * P + p;
      const float a = aw[pi];
      const Footprint f = make_footprint(loc[2 * pi], loc[2 * pi + 1], Hh, Ww);
      float dot[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < Dh; ++c) {
        const float g = elem<T>::to_float(gout[bq * H * Dh + h * Dh + c]);
        for (int k = 0; k < 4; ++k) {
          const long o = base + (long)f.idx[k] * H * Dh + c;
          dot[k] = fmaf(g, elem<T>::to_float(value[o]) * f.m[k], dot[k]);
          const float cw = a * f.w[k];
          if (cw != 0.0f) atomic_add_f32(gvalue + o, cw * g);
        }
      }
      const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
      gaw[pi] = hy * hx * dot[0] + hy * f.lx * dot[1] + f.ly * hx * dot[2] + f.ly * f.lx * dot[3];
      gloc[2 * pi] = a * ((dot[1] - dot[0]) * hy + (dot[3] - dot[2]) * f.ly) * (float)Ww;
      gloc[2 * pi + 1] = a * ((dot[2] - dot[0]) * hx + (dot[3] - dot[1]) * f.lx) * (float)Hh;
    }
  }
}

// ---- dispatch ----------------------------------------------------------------------------------------
struct K1Args {
  const void* value; const int64_t* ss; const int64_t* ls; const float* loc; const float* aw;
  void* out; const void* gout; float* gvalue; float* gloc; float* gaw;
  int B, S, H, Dh, L, Nq, P;
};

template <typename T, int DH, int VEC>
static void launch_fast(const K1Args& a, bool bwd, hipStream_t st) {
  constexpr int LP = DH / VEC;
  const int QW = kWave / (a.H * LP);
  const long waves = ((long)a.B * a.Nq + QW - 1) / QW;
  const int blocks = (int)((waves + 3) / 4);
  if (!bwd)
    hipLaunchKernelGGL((k1_fwd_kernel<T, DH, VEC>), dim3(blocks), dim3(256), 0, st,
                       (const T*)a.value, a.ss, a.ls, a.loc, a.aw, (T*)a.out, a.B, a.S, a.H, a.L,
                       a.Nq, a.P);
  else
    hipLaunchKernelGGL((k1_bwd_kernel<T, DH, VEC>), dim3(blocks), dim3(256), 0, st,
                       (const T*)a.value, a.ss, a.ls, a.loc, a.aw, (const T*)a.gout, a.gvalue,
                       a.gloc, a.gaw, a.B, a.S, a.H, a.L, a.Nq, a.P);
}

template <typename T>
static void launch_any(const K1Args& a, bool bwd, hipStream_t st) {
  const long n = (long)a.B * a.Nq * a.H;
  const int blocks = (int)((n + 255) / 256);
  if (!bwd)
    hipLaunchKernelGGL((k1_fwd_any_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)a.value,
                       a.ss, a.ls, a.loc, a.aw, (T*)a.out, a.B, a.S, a.H, a.Dh, a.L, a.Nq, a.P);
  else
    hipLaunchKernelGGL((k1_bwd_any_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)a.value,
                       a.ss, a.ls, a.loc, a.aw, (const T*)a.gout, a.gvalue, a.gloc, a.gaw, a.B,
                       a.S, a.H, a.Dh, a.L, a.Nq, a.P);
}

template <typename T>
static void dispatch_T(const K1Args& a, bool bwd, hipStream_t st) {
  constexpr int V16 = 16 / elem<T>::kBytes;     // channels in a 16-B vector
  auto fits = [&](int dh, int vec) { return a.Dh == dh && (kWave % (a.H * (dh / vec))) == 0 &&
                                            a.H * (dh / vec) <= kWave; };
  if (fits(32, V16)) return launch_fast<T, 32, V16>(a, bwd, st);
  if (fits(16, V16)) return launch_fast<T, 16, V16>(a, bwd, st);
  if (fits(64, V16)) return launch_fast<T, 64, V16>(a, bwd, st);
  if (fits(8, 4)) return launch_fast<T, 8, 4>(a, bwd, st);
  if (fits(4, 4)) return launch_fast<T, 4, 4>(a, bwd, st);
  launch_any<T>(a, bwd, st);
}

static int k1_dispatch(const K1Args& a, int dtype, bool bwd, void* stream) {
  UBV_CHECK_ARG(a.B > 0 && a.S > 0 && a.H > 0 && a.Dh > 0 && a.L > 0 && a.Nq >= 0 && a.P > 0,
                "ms_deform_attn: non-positive dimension (B=%d S=%d H=%d Dh=%d L=%d Nq=%d P=%d)",
                a.B, a.S, a.H, a.Dh, a.L, a.Nq, a.P);
  UBV_CHECK_ARG(a.L <= kMaxLevels, "ms_deform_attn: at most %d levels (got %d)", kMaxLevels, a.L);
  if (a.Nq == 0) return UBV_OK;                 // empty query set: nothing to do, nothing to check
  UBV_CHECK_ARG(a.value && a.ss && a.ls && a.loc && a.aw, "ms_deform_attn: null input pointer");
  UBV_CHECK_ARG(bwd ? (a.gout && a.gvalue && a.gloc && a.gaw) : (a.out != nullptr),
                "ms_deform_attn: null output / gradient pointer");
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32: dispatch_T<float>(a, bwd, st); break;
    case UBV_F16: dispatch_T<f16_t>(a, bwd, st); break;
    case UBV_BF16: dispatch_T<bf16_t>(a, bwd, st); break;
    default: set_error("ms_deform_attn: unknown dtype %d", dtype); return UBV_ERR_INVALID;
  }
  UBV_CHECK_LAUNCH(bwd ? "ms_deform_attn_backward" : "ms_deform_attn_forward");
  return UBV_OK;
}

template <typename T, int DH>
static void launch_query_only(const K1Args& a, hipStream_t st) {
  constexpr int VEC = 16 / elem<T>::kBytes, LP = DH / VEC;
  const int QW = kWave / (a.H * LP);
  const long waves = ((long)a.B * a.Nq + QW - 1) / QW;
  hipLaunchKernelGGL((k1_bwd_kernel<T, DH, VEC, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st,
                     (const T*)a.value, a.ss, a.ls, a.loc, a.aw, (const T*)a.gout, a.gvalue, a.gloc, a.gaw, a.B, a.S,
                     a.H, a.L, a.Nq, a.P);
}

}  // namespace ubv

extern "C" int64_t ubv_ms_deform_attn_backward_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P,
                                                         int dtype) {
  if (B <= 0 || Nq <= 0 || !ubv::k1_grid_ok(H, Dh, P, dtype, fh, fw)) return 0;
  return ubv::k1_grid_workspace(B, fh, fw, H, Dh, Nq, P, dtype);
}

extern "C" int ubv_ms_deform_attn_backward_planned(const void* value, const int64_t* spatial_shapes,
                                                   const int64_t* level_start, const float* sampling_loc,
                                                   const float* attn_weight, const void* grad_out, float* grad_value,
                                                   float* grad_sampling_loc, float* grad_attn_weight, int B, int S,
                                                   int H, int Dh, int L, int Nq, int P, int dtype, int fh, int fw,
                                                   void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace ubv;
  if (L != 1 || (long)fh * fw != S || !k1_grid_ok(H, Dh, P, dtype, fh, fw)) {
    set_error("ms_deform_attn_backward_planned: one level of fh x fw = S pixels, Dh in {16, 32}, P in {4, 8} "
              "(got L=%d S=%d %dx%d Dh=%d P=%d)", L, S, fh, fw, Dh, P);
    return UBV_ERR_UNSUPPORTED;
  }
  if (Nq == 0) return UBV_OK;
  UBV_CHECK_ARG(value && spatial_shapes && level_start && sampling_loc && attn_weight && grad_out && grad_value &&
                    grad_sampling_loc && grad_attn_weight, "ms_deform_attn_backward_planned: null pointer");
  K1Args a{value, spatial_shapes, level_start, sampling_loc, attn_weight, nullptr, grad_out, grad_value,
           grad_sampling_loc, grad_attn_weight, B, S, H, Dh, L, Nq, P};
  hipStream_t st = as_stream(stream);
#define UBV_K1Q(TT) do { if (Dh == 32) launch_query_only<TT, 32>(a, st); else launch_query_only<TT, 16>(a, st); } while (0)
  if (dtype == UBV_F32) UBV_K1Q(float); else if (dtype == UBV_F16) UBV_K1Q(f16_t); else UBV_K1Q(bf16_t);
#undef UBV_K1Q
  const int rc = k1_grid_value(sampling_loc, attn_weight, grad_out, grad_value, B, fh, fw, H, Dh, Nq, P, dtype,
                               workspace, workspace_bytes, st);
  if (rc != UBV_OK) return rc;
  UBV_CHECK_LAUNCH("ms_deform_attn_backward_planned");
  return UBV_OK;
}

extern "C" int ubv_ms_deform_attn_forward(const void* value, const int64_t* spatial_shapes,
                                          const int64_t* level_start, const float* sampling_loc,
                                          const float* attn_weight, void* out, int B, int S, int H,
                                          int Dh, int L, int Nq, int P, int dtype, int im2col_step,
                                          void* stream) {
  (void)im2col_step;
  ubv::K1Args a{value, spatial_shapes, level_start, sampling_loc, attn_weight, out, nullptr,
                nullptr, nullptr, nullptr, B, S, H, Dh, L, Nq, P};
  return ubv::k1_dispatch(a, dtype, false, stream);
}

extern "C" int ubv_ms_deform_attn_backward(const void* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start, const float* sampling_loc,
                                           const float* attn_weight, const void* grad_out,
                                           float* grad_value, float* grad_sampling_loc,
                                           float* grad_attn_weight, int B, int S, int H, int Dh,
                                           int L, int Nq, int P, int dtype, int im2col_step,
                                           void* stream) {
  (void)im2col_step;
  ubv::K1Args a{value, spatial_shapes, level_start, sampling_loc, attn_weight, nullptr, grad_out,
                grad_value, grad_sampling_loc, grad_attn_weight, B, S, H, Dh, L, Nq, P};
  return ubv::k1_dispatch(a, dtype, true, stream);
}

// The operator with a QUERY-GRID hint: the Nq queries are a qgrid_h x qgrid_w grid in row-major order (BEV queries —
// what the reference's call sites spatial_cross_attention_pts.py:439-442 and the encoder's self-attention pass) and the
// single level is fh x fw.  f32, 8 heads of 32 channels, 4 or 8 points: the TILE plan of csrc/bev_lift_tile.hip (one lane
// per sampling point, the tile's pixel box in LDS; backward: owner tiles, no f32 atomics).  Any other shape: the plain
// operator (forward) / UBV_ERR_UNSUPPORTED (backward: the caller takes ubv_ms_deform_attn_backward[_planned]).
extern "C" int ubv_ms_deform_attn_grid_supported(int H, int Dh, int L, int P, int dtype, int fh, int fw, int Nq, int qgrid_h,
                                                 int qgrid_w) {
  return (L == 1 && ubv::k1_tile_ok(H, Dh, P, dtype, fh, fw, Nq, qgrid_h, qgrid_w)) ? 1 : 0;
}

extern "C" int ubv_ms_deform_attn_forward_grid(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                               const float* sampling_loc, const float* attn_weight, void* out, int B, int S,
                                               int H, int Dh, int L, int Nq, int P, int dtype, int fh, int fw, int qgrid_h,
                                               int qgrid_w, void* stream) {
  using namespace ubv;
  if (L == 1 && (long)fh * fw == S && Nq > 0 && k1_tile_ok(H, Dh, P, dtype, fh, fw, Nq, qgrid_h, qgrid_w)) {
    UBV_CHECK_ARG(value && sampling_loc && attn_weight && out, "ms_deform_attn_forward_grid: null pointer");
    const int rc = k1_tile_forward(value, sampling_loc, attn_weight, out, B, fh, fw, H, Nq, P, qgrid_h, qgrid_w, as_stream(stream));
    if (rc != UBV_OK) return rc;
    UBV_CHECK_LAUNCH("ms_deform_attn_forward_grid");
    return UBV_OK;
  }
  return ubv_ms_deform_attn_forward(value, spatial_shapes, level_start, sampling_loc, attn_weight, out, B, S, H, Dh, L, Nq, P,
                                    dtype, 64, stream);
}

extern "C" int64_t ubv_ms_deform_attn_backward_grid_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype,
                                                              int qgrid_h, int qgrid_w) {
  if (B <= 0 || !ubv::k1_tile_ok(H, Dh, P, dtype, fh, fw, Nq, qgrid_h, qgrid_w)) return 0;
  return ubv::k1_tile_workspace(B, fh, fw, H, Nq, P, qgrid_h, qgrid_w);
}

extern "C" int ubv_ms_deform_attn_backward_grid(const void* value, const float* sampling_loc, const float* attn_weight,
                                                const void* grad_out, float* grad_value, float* grad_sampling_loc,
                                                float* grad_attn_weight, int B, int S, int H, int Dh, int L, int Nq, int P,
                                                int dtype, int fh, int fw, int qgrid_h, int qgrid_w, void* workspace,
                                                int64_t workspace_bytes, void* stream) {
  using namespace ubv;
  if (L != 1 || (long)fh * fw != S || !k1_tile_ok(H, Dh, P, dtype, fh, fw, Nq, qgrid_h, qgrid_w)) {
    set_error("ms_deform_attn_backward_grid: f32, one level of fh x fw = S pixels, 8 heads of 32 channels, P in {4, 8}, "
              "queries = qgrid_h x qgrid_w (got L=%d S=%d %dx%d H=%d Dh=%d P=%d Nq=%d grid %dx%d)", L, S, fh, fw, H, Dh, P, Nq,
              qgrid_h, qgrid_w);
    return UBV_ERR_UNSUPPORTED;
  }
  UBV_CHECK_ARG(value && sampling_loc && attn_weight && grad_out && grad_value && grad_sampling_loc && grad_attn_weight,
                "ms_deform_attn_backward_grid: null pointer");
  const int rc = k1_tile_backward(value, sampling_loc, attn_weight, grad_out, grad_value, grad_sampling_loc, grad_attn_weight,
                                  B, fh, fw, H, Nq, P, qgrid_h, qgrid_w, workspace, workspace_bytes, as_stream(stream));
  if (rc != UBV_OK) return rc;
  UBV_CHECK_LAUNCH("ms_deform_attn_backward_grid");
  return UBV_OK;
}
