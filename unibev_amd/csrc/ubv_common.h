// Shared device/host helpers for libunibev_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "unibev_hip.h"

namespace ubv {

constexpr int kWave = 64;   // CDNA wavefront

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define UBV_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::ubv::set_error(__VA_ARGS__);        \
      return UBV_ERR_INVALID;               \
    }                                       \
  } while (0)

#define UBV_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ::ubv::set_error("%s: %s", what, hipGetErrorString(e__));                  \
      return UBV_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

// ---- element types ------------------------------------------------------------------------------
struct bf16_t { uint16_t bits; };
using f16_t = _Float16;

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }
// f32 -> bf16, round to nearest even, NaN kept quiet: gfx950's v_cvt_pk_bf16_f32 (two values per
// instruction; the software rounding is five VALU operations and a NaN branch per value).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ uint32_t float_to_bf16_bits(float f) { return cvt_pk_bf16(f, f) & 0xffffu; }

template <typename T> struct elem;
template <> struct elem<float> {
  static constexpr int kBytes = 4;
  static __device__ __forceinline__ float to_float(float v) { return v; }
  static __device__ __forceinline__ float from_float(float v) { return v; }
};
template <> struct elem<f16_t> {
  static constexpr int kBytes = 2;
  static __device__ __forceinline__ float to_float(f16_t v) { return (float)v; }
  static __device__ __forceinline__ f16_t from_float(float v) { return (f16_t)v; }
};
template <> struct elem<bf16_t> {
  static constexpr int kBytes = 2;
  static __device__ __forceinline__ float to_float(bf16_t v) { return bf16_bits_to_float(v.bits); }
  static __device__ __forceinline__ bf16_t from_float(float v) {
    bf16_t r; r.bits = (uint16_t)float_to_bf16_bits(v); return r;
  }
};

// ---- 16-byte (or 8-byte) vector load/store of VEC elements, widened to f32 -----------------------
template <typename T, int VEC> struct vec_io;

template <> struct vec_io<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <> struct vec_io<float, 8> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const float4 t = *reinterpret_cast<const float4*>(p), u = *reinterpret_cast<const float4*>(p + 4);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

template <> struct vec_io<f16_t, 8> {
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[8]) {
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    const h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[8]) {
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    h8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (_Float16)v[i];
    *reinterpret_cast<h8*>(p) = t;
  }
};
template <> struct vec_io<f16_t, 4> {
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[4]) {
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    const h4 t = *reinterpret_cast<const h4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[4]) {
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    h4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (_Float16)v[i];
    *reinterpret_cast<h4*>(p) = t;
  }
};

template <> struct vec_io<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct vec_io<bf16_t, 4> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = cvt_pk_bf16(v[0], v[1]);
    t.y = cvt_pk_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = t;
  }
};

// Hardware f32 atomic add without a CAS loop (global_atomic_add_f32, no return).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- bilinear footprint of one sampling point ------------------------------------------------------
// Pixel coordinates x = loc_x*W - 0.5, y = loc_y*H - 0.5 (mmcv ms_deform_attn / grid_sample
// align_corners=False); corners outside the map contribute zero.  Corner indices are clamped so
// that every corner can be loaded unconditionally; a corner outside gets weight 0.
struct Footprint {
  int idx[4];      // y*W + x of the 4 (clamped) corners: 00, 01(x+1), 10(y+1), 11
  float w[4];      // bilinear weight of each corner, 0 when the corner is outside
  float lx, ly;    // fractional parts
  float m[4];      // 1 where the corner is inside, else 0 (for the gradient w.r.t. the location)
  int xc[2], yc[2];  // clamped corner columns / rows: corner k sits at (xc[k & 1], yc[k >> 1])
};

// Footprint from pixel coordinates (x, y) = (loc_x*W - 0.5, loc_y*H - 0.5).
__device__ __forceinline__ Footprint footprint_px(float x, float y, int Hh, int Ww) {
  Footprint f;
  const bool inside = (y > -1.0f) && (x > -1.0f) && (y < (float)Hh) && (x < (float)Ww);
  const float xf = floorf(x), yf = floorf(y);
  // NaN / huge locations: `inside` is false, keep the integer conversion defined.
  const int x0 = inside ? (int)xf : 0;
  const int y0 = inside ? (int)yf : 0;
  f.lx = inside ? x - xf : 0.0f;
  f.ly = inside ? y - yf : 0.0f;
  const float mx0 = (inside && x0 >= 0) ? 1.0f : 0.0f;
  const float mx1 = (inside && x0 + 1 <= Ww - 1) ? 1.0f : 0.0f;
  const float my0 = (inside && y0 >= 0) ? 1.0f : 0.0f;
  const float my1 = (inside && y0 + 1 <= Hh - 1) ? 1.0f : 0.0f;
  const int xc0 = min(max(x0, 0), Ww - 1), xc1 = min(max(x0 + 1, 0), Ww - 1);
  const int yc0 = min(max(y0, 0), Hh - 1), yc1 = min(max(y0 + 1, 0), Hh - 1);
  f.xc[0] = xc0; f.xc[1] = xc1; f.yc[0] = yc0; f.yc[1] = yc1;
  f.idx[0] = yc0 * Ww + xc0; f.idx[1] = yc0 * Ww + xc1;
  f.idx[2] = yc1 * Ww + xc0; f.idx[3] = yc1 * Ww + xc1;
  f.m[0] = my0 * mx0; f.m[1] = my0 * mx1; f.m[2] = my1 * mx0; f.m[3] = my1 * mx1;
  const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
  f.w[0] = hy * hx * f.m[0]; f.w[1] = hy * f.lx * f.m[1];
  f.w[2] = f.ly * hx * f.m[2]; f.w[3] = f.ly * f.lx * f.m[3];
  return f;
}

__device__ __forceinline__ Footprint make_footprint(float loc_x, float loc_y, int Hh, int Ww) {
  return footprint_px(loc_x * (float)Ww - 0.5f, loc_y * (float)Hh - 0.5f, Hh, Ww);
}

// XCD-aware block -> work-item remap: blocks are dispatched round-robin over the 8 XCDs
// (block b on XCD b % 8, observed, speed only), so give XCD x the contiguous item range
// [x*chunk, (x+1)*chunk): neighbouring tiles then share one XCD's 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int chunk) { return (bid & 7) * chunk + (bid >> 3); }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// DCNv2 d(input) by owner tiles (dcn_owner.inl, compiled with bev_lift.hip whose TileAcc it reuses); false when the
// shape is outside its reach.  Samples with an offset component beyond kDcnReach are left to the caller.
bool dcn_owner_launch(const void* gcol, const void* offset, const void* mask, float* gx, int N, int H, int W, int C,
                      int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                      int dtype, hipStream_t st);

// grad_value of the k1 operator on the GRID owner-tile plan (bev_lift.hip), called from msda_k1.hip
bool k1_grid_ok(int H, int Dh, int P, int dtype, int fh, int fw);
int64_t k1_grid_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype);
int k1_grid_value(const float* loc, const float* aw, const void* gout, float* gvalue, int B, int fh, int fw, int H,
                  int Dh, int Nq, int P, int dtype, void* ws, int64_t ws_bytes, hipStream_t st);
// the k1 operator on the TILE plan (bev_lift_tile.hip) when its queries are a qh x qw grid in row-major order
bool k1_tile_ok(int H, int Dh, int P, int dtype, int fh, int fw, int Nq, int qh, int qw);
int k1_tile_forward(const void* value, const float* loc, const float* aw, void* out, int B, int fh, int fw, int H, int Nq,
                    int P, int qh, int qw, hipStream_t st);
int64_t k1_tile_workspace(int B, int fh, int fw, int H, int Nq, int P, int qh, int qw);
int k1_tile_backward(const void* value, const float* loc, const float* aw, const void* gout, float* gvalue, float* gloc,
                     float* gaw, int B, int fh, int fw, int H, int Nq, int P, int qh, int qw, void* ws, int64_t ws_bytes,
                     hipStream_t st);

// RAII pair of HIP events around one kernel launch (a no-op unless ubv_profile_enable(1)).
class ProfScope {
 public:
  ProfScope(const char* name, hipStream_t st, double algorithmic_bytes);
  ~ProfScope();
 private:
  long idx_;
  hipStream_t st_;
};

// ---- dropout: stateless keep mask (shared by add_norm.hip and the GEMM epilogues) -----------------
// 32-bit mix of (seed, element index): keep iff hash >= threshold.
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
// The FFN activation (relu_dropout kernels and the GEMM epilogue) draws FOUR 16-bit decisions from one
// 64-bit mix, for the 4-element group the index belongs to: a quarter of the 64-bit multiplies, which a
// GEMM epilogue cannot hide behind memory time the way a streaming kernel does.  keep4(...)[e] for
// element 4 g + e; thresh16 = thresh >> 16 (p at a resolution of 2^-16).
__device__ __forceinline__ uint64_t drop_mix64(uint64_t seed, uint64_t group) {
  uint64_t z = group + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool drop_keep16(uint64_t mix, int e, uint32_t thresh) {
  return (uint32_t)((mix >> (16 * e)) & 0xffffull) >= (thresh >> 16);
}
static inline void drop_params(float p, uint32_t& thresh, float& scale) {
  if (p <= 0.0f) { thresh = 0u; scale = 1.0f; return; }
  const double t = (double)p * 4294967296.0;
  thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
  scale = 1.0f / (1.0f - p);
}

}  // namespace ubv
