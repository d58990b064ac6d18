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
#include <limits.h>
#include <stdlib.h>

#include "ubv_common.h"
#include "bev_lift_core.h"

namespace ubv {

// FAST (16-bit data paths): one reciprocal instead of P IEEE divisions — an ulp of f32 that the
// 16-bit values downstream cannot see.  f32 data keeps the reference's exact divisions.
template <int P, bool FAST = false>
__device__ __forceinline__ void softmax_row(const float (&l)[P], float (&w)[P]) {
  float m = l[0];
#pragma unroll
  for (int i = 1; i < P; ++i) m = fmaxf(m, l[i]);
  float s = 0.0f;
  // FAST: v_exp_f32 on a pre-scaled argument and v_rcp_f32 (about 2 ulp of f32, invisible to 16-bit
  // data) instead of expf's range reduction (~12 instructions each) and an IEEE division
#pragma unroll
  for (int i = 0; i < P; ++i) { w[i] = FAST ? __expf(l[i] - m) : expf(l[i] - m); s += w[i]; }
  const float inv = FAST ? __builtin_amdgcn_rcpf(s) : 1.0f / s;
#pragma unroll
  for (int i = 0; i < P; ++i) w[i] = FAST ? w[i] * inv : w[i] / s;
}

// a / b, or a * (1/b) with a precomputed reciprocal on the 16-bit data paths (see softmax_row)
template <bool FAST>
__device__ __forceinline__ float div_or_mul(float a, float b, float inv_b) {
  return FAST ? a * inv_b : a / b;
}

// Rows of offsets / logits / their gradients: f32, or (lowp) the value's 16-bit type.  Under
// autocast the producing Linear already emits 16-bit values, so reading them here directly is
// bit-identical to the f32 up-cast it replaces and saves a cast kernel + a 2x larger read.
template <typename T, int N>
__device__ __forceinline__ void load_ol(const void* base, long idx, bool lowp, float (&v)[N]) {
  if constexpr (sizeof(T) == 2) {
    if (lowp) {
      const T* p = (const T*)base + idx;
      if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 8) {
          float t[8];
          vec_io<T, 8>::load(p + i, t);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[i + k] = t[k];
        }
      } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
          float t[4];
          vec_io<T, 4>::load(p + i, t);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[i + k] = t[k];
        }
      }
      return;
    }
  }
  load_row<N>((const float*)base + idx, v);
}
template <typename T, int N>
__device__ __forceinline__ void store_ol(void* base, long idx, bool lowp, const float (&v)[N]) {
  if constexpr (sizeof(T) == 2) {
    if (lowp) {
      T* p = (T*)base + idx;
      if constexpr (N % 8 == 0) {
#pragma unroll
        for (int i = 0; i < N; i += 8) {
          float t[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] = v[i + k];
          vec_io<T, 8>::store(p + i, t);
        }
      } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
          float t[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] = v[i + k];
          vec_io<T, 4>::store(p + i, t);
        }
      }
      return;
    }
  }
  store_row<N>((float*)base + idx, v);
}
template <typename T>
__device__ __forceinline__ float ld_ol(const void* base, long idx, bool lowp) {
  if constexpr (sizeof(T) == 2) {
    if (lowp) return elem<T>::to_float(((const T*)base)[idx]);
  }
  return ((const float*)base)[idx];
}

template <typename T, int DH, int VEC, int P, bool OL16>
// Register budget: the compiler's schedule depends on the occupancy it is allowed to aim for.  Measured
// (bs = 2, bf16): P = 8 with at most 2 waves per SIMD 151 -> 126 us (SCA-pts), 183 -> 171 us (SCA-img);
// P = 4 unconstrained (152 VGPRs, 3 waves) 87 -> 67 us.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (P == 8 ? 2 : 8)))) void lift_fwd_kernel(const LiftArgs a) {
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
  constexpr bool FAST = sizeof(T) == 2;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;
  const int rowi = (int)row;                          // element offsets of a map fit 32 bits (checked)

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    if (!lift_query(a, item, li0 + wv * QW + sub, b, q)) continue;
    const long bq = (long)b * a.Nq + q;
    float lg[P], w[P], off[2 * P];
    load_ol<T, P>(a.logits, bq * a.log_stride + h * P, OL16, lg);
    load_ol<T, 2 * P>(a.offsets, bq * a.off_stride + h * 2 * P, OL16, off);
    softmax_row<P, FAST>(lg, w);

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
        const float lx = r.x + div_or_mul<FAST>(off[2 * p], fwf, inv_fw);
        const float ly = r.y + div_or_mul<FAST>(off[2 * p + 1], fhf, inv_fh);
        const Footprint f = make_footprint(lx, ly, a.fh, a.fw);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[VEC];
          vec_io<T, VEC>::load(vb + f.idx[k] * rowi, v);
          const float c = w[p] * f.w[k];
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
        }
      }
    }
    if (a.count != nullptr) {
      const float cnt = a.count[bq], inv_cnt = 1.0f / cnt;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = div_or_mul<FAST>(acc[i], cnt, inv_cnt);
    }
    vec_io<T, VEC>::store(out + bq * row + h * DH + cg * VEC, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, part 1 — per-query gradients (d offsets, d logits): the forward's gather plus dot
// products with grad_out, reduced over the Dh lanes with wave shuffles.  ATOMICS selects who
// scatters grad_value:
//   kAtomAll : this kernel, one hardware f32 atomic per (corner, channel) — the mmcv scheme; kept for
//              arbitrary reference points on large maps.
//   kAtomNone: never — the owner-tile kernels below take every corner without atomics.
// The launcher names its three plans with the same enum: kAtomAll (this kernel scatters), kPlanGrid
// (points binned by owner tile: lift_bin_kernel + lift_bwd_value_kernel) and kAtomNone for the
// CAMERA plan (per-camera lists: lift_bwd_value_camera_kernel).
enum { kAtomAll = 0, kPlanGrid = 1, kAtomNone = 2, kPlanMaps = 3 };


template <typename T, int DH, int VEC, int P, int ATOMICS, bool OL16>
// P = 4 runs best at 3 waves per SIMD (127 -> 96 us); P = 8 loses to its own cache footprint there.
__global__ __launch_bounds__(256, ((ATOMICS == kAtomNone && P == 4) ? 3 : 1)) void lift_bwd_query_kernel(const LiftArgs a) {
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
  constexpr bool FAST = sizeof(T) == 2;       // reciprocals instead of IEEE divisions (softmax_row)
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;
  const int rowi = (int)row;

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    const bool valid = lift_query(a, item, li0 + wv * QW + sub, b, q);
    if (!valid) { b = 0; q = 0; }             // keep every lane in the shuffles below
    const long bq = (long)b * a.Nq + q;
    float lg[P], w[P], off[2 * P];
    load_ol<T, P>(a.logits, bq * a.log_stride + h * P, OL16, lg);
    load_ol<T, 2 * P>(a.offsets, bq * a.off_stride + h * 2 * P, OL16, off);
    softmax_row<P, FAST>(lg, w);

    float go[VEC];
    vec_io<T, VEC>::load(gout + bq * row + h * DH + cg * VEC, go);
    const float inv = valid ? 1.0f : 0.0f;
    const float cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
    const float inv_cnt = 1.0f / cnt;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      go[i] = (a.count != nullptr ? div_or_mul<FAST>(go[i], cnt, inv_cnt) : go[i]) * inv;

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
        const float lx = r.x + div_or_mul<FAST>(off[2 * p], fwf, inv_fw);
        const float ly = r.y + div_or_mul<FAST>(off[2 * p + 1], fhf, inv_fh);
        const float xp = lx * fwf - 0.5f, yp = ly * fhf - 0.5f;
        const Footprint f = footprint_px(xp, yp, a.fh, a.fw);
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[VEC];
          const long o = vo + f.idx[k] * rowi;
          vec_io<T, VEC>::load(value + o, v);
          float d = 0.0f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) d = fmaf(go[i], v[i], d);
          dot[k] = d * f.m[k];
          if (ATOMICS == kAtomAll) {
            const float c = w[p] * f.w[k];
            if (valid && c != 0.0f) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) atomic_add_f32(a.gvalue + o + i, c * go[i]);
            }
          }
        }
        const float hx_ = 1.0f - f.lx, hy_ = 1.0f - f.ly;
        gw[p] += hy_ * hx_ * dot[0] + hy_ * f.lx * dot[1] + f.ly * hx_ * dot[2] + f.ly * f.lx * dot[3];
        gx[p] += (dot[1] - dot[0]) * hy_ + (dot[3] - dot[2]) * f.ly;
        gy[p] += (dot[2] - dot[0]) * hx_ + (dot[3] - dot[1]) * f.lx;
      }
    }
    // reduce the Dh partial dot products over the LP lanes of this (query, head)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (LP > 1) { gw[p] = add_xor<1>(gw[p]); gx[p] = add_xor<1>(gx[p]); gy[p] = add_xor<1>(gy[p]); }
      if (LP > 2) { gw[p] = add_xor<2>(gw[p]); gx[p] = add_xor<2>(gx[p]); gy[p] = add_xor<2>(gy[p]); }
      if (LP > 4) { gw[p] = add_xor<4>(gw[p]); gx[p] = add_xor<4>(gx[p]); gy[p] = add_xor<4>(gy[p]); }
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
        // d loc = w*g*W ; d off = d loc / W (the reference's rounding; elided on 16-bit paths)
        gofs[2 * p] = FAST ? w[p] * gx[p] : (w[p] * gx[p] * fwf) / fwf;
        gofs[2 * p + 1] = FAST ? w[p] * gy[p] : (w[p] * gy[p] * fhf) / fhf;
      }
      store_ol<T, P>(a.glog, bq * a.glog_stride + h * P, OL16, gl);
      store_ol<T, 2 * P>(a.goff, bq * a.goff_stride + h * 2 * P, OL16, gofs);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GRID plan, step 0 — bin the sampling points by OWNER TILE.  A point (query, head, slot) touches
// up to 4 corners in up to 4 of the 8x8-pixel tiles of its (sample, head) map; its record
// (x_pix, y_pix, w/count, query) is appended to the bucket of every tile that holds a corner with
// non-zero weight (1.3 buckets per point on average), so the owner-tile kernel below reads exactly
// the points it owns — the previous plan searched a 14x14 query window per slot and discarded 2/3
// of what it evaluated.  A wave is one (8x8 query tile, head): its points fall into <= ~4 tiles, so
// the appends are wave-aggregated: ranks come from LDS counters on an 8x8 torus of tile slots and
// each occupied slot costs ONE global atomic per sampling slot p, all of them in flight together
// (a leader loop with one returning atomic per distinct tile ran at 93 us, bound by the round trips).  Buckets have a fixed capacity (2x the expected load); what does not fit goes to an
// overflow list that lift_ovf_* scatter atomically, so the result is exact for ANY offsets.
struct TileArgs;

// MODE 0: fixed-capacity buckets + overflow list (GRID).  MODE 1 / 2: the two passes of an exact CSR
// build for the MAPS plan (count the records per bucket; after the scan, write them at
// bin_start[bucket] + cursor) — per-camera maps, where the load per tile varies 100:1 (the rows at the
// horizon receive 85 % of the projected pillars) and no fixed capacity fits.  The cameras are walked in
// the wave; one that none of the wave's 64 queries sees costs a ballot.
// K1: the operator-level backward (ubv_ms_deform_attn_backward_planned): a.offsets holds explicit sampling LOCATIONS
// [B, Nq, H, P, 2] (normalised x, y), a.logits the attention WEIGHTS [B, Nq, H, P] — no reference points, no softmax.
template <typename T, int DH, int P, int MODE, bool K1 = false>
__global__ __launch_bounds__(256) void lift_bin_kernel(const LiftArgs a, int tiles_x, int tiles) {
  // per wave: an 8x8 torus of tile slots — occupant tile, local count, global base
  __shared__ volatile int slot_tile[4][64];
  __shared__ int slot_cnt[4][64], slot_base[4][64];
  const int wv = wave_in_block(), lane = threadIdx.x & 63;
  slot_tile[wv][lane] = -1;
  slot_cnt[wv][lane] = 0;
  const long wave = (long)blockIdx.x * 4 + wv;
  if (wave >= (long)a.total_tiles * a.H) return;
  const int item = (int)(wave / a.H), h = (int)(wave - (long)item * a.H);   // head fastest
  int b, q;
  const bool valid = lift_query(a, item, lane, b, q);
  if (!valid) q = 0;                                   // b is wave-uniform either way
  const long bq = (long)b * a.Nq + q;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  float lg[P], w[P], off[2 * P];
  load_ol<T, P>(a.logits, bq * a.log_stride + h * P, K1 ? 0 : a.ol16, lg);
  load_ol<T, 2 * P>(a.offsets, bq * a.off_stride + h * 2 * P, K1 ? 0 : a.ol16, off);
  // 16-bit data: hardware exp / reciprocals like the other kernels of the path (the kernel is bound by its
  // per-point arithmetic: one lane = one point, ~150 instructions each with IEEE divisions)
  // (f32 data keeps the IEEE forms: hardware exp / reciprocals measured 36.4 vs 37.0 us, 63.3 vs 64.8 — not where the time is)
  constexpr bool FAST = sizeof(T) == 2 && !K1;
  if constexpr (K1) {
#pragma unroll
    for (int i = 0; i < P; ++i) w[i] = lg[i];
  } else {
    softmax_row<P, FAST>(lg, w);
  }
  const float cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
  const float inv_cnt = 1.0f / cnt, inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;
  for (int cam = 0; cam < a.Nc; ++cam) {
  const bool vis = valid && (a.vis0 == nullptr || a.vis0[(long)cam * a.Nq + q] != 0);
  if (a.Nc > 1 && __ballot(vis) == 0ull) continue;    // wave-uniform
  const float* rp = K1 ? nullptr : a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
  const int tile_base = ((b * a.Nc + cam) * a.H + h) * tiles;     // first bucket of this (map, head)
  int* __restrict__ cntp = (MODE == 2 ? a.bin_cur : a.bin_cnt) + tile_base;
  float4* __restrict__ binp = MODE == 0 ? a.bins + (long)tile_base * a.cap : a.bins;

  auto put = [&](int tile, int idx, const float4& rec) {
    if constexpr (MODE == 0) {
      if (idx < a.cap) {
        binp[(long)tile * a.cap + idx] = rec;
      } else {
        const int o = atomicAdd(a.ovf_n, 1);
        if (o < a.ovf_cap) { a.ovf_rec[o] = rec; a.ovf_tile[o] = tile_base + tile; }
      }
    } else if constexpr (MODE == 2) {
      binp[idx] = rec;                                // idx already includes bin_start
    }
  };

  int zi = 0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float lx, ly;
    if constexpr (K1) {
      lx = off[2 * p]; ly = off[2 * p + 1];
    } else {
      const float2 r = *reinterpret_cast<const float2*>(rp + zi * 2);
      zi = (zi + 1 == a.Z) ? 0 : zi + 1;
      lx = r.x + div_or_mul<FAST>(off[2 * p], fwf, inv_fw);
      ly = r.y + div_or_mul<FAST>(off[2 * p + 1], fhf, inv_fh);
    }
    const float xp = lx * fwf - 0.5f, yp = ly * fhf - 0.5f;
    const Footprint f = footprint_px(xp, yp, a.fh, a.fw);
    const float wn = div_or_mul<FAST>(w[p], cnt, inv_cnt);
    const float4 rec = make_float4(xp, yp, wn, __int_as_float(q));
    int tk[4], hs[4], rank[4];
    bool nz[4], lead[4], local[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      nz[k] = vis && wn != 0.0f && f.w[k] != 0.0f;
      const int tx = f.xc[k & 1] >> 3, ty = f.yc[k >> 1] >> 3;
      tk[k] = ty * tiles_x + tx;
      hs[k] = ((ty & 7) << 3) | (tx & 7);
    }
    // a tile receives the record once: through its first corner with non-zero weight
    lead[0] = nz[0];
    lead[1] = nz[1] && !(nz[0] && tk[1] == tk[0]);
    lead[2] = nz[2] && !(nz[0] && tk[2] == tk[0]) && !(nz[1] && tk[2] == tk[1]);
    lead[3] = nz[3] && !(nz[0] && tk[3] == tk[0]) && !(nz[1] && tk[3] == tk[1]) &&
              !(nz[2] && tk[3] == tk[2]);
    // 1. rank within the wave through LDS counters (a wave's points span a few tiles; two tiles
    //    that collide on the torus — never for coherent offsets — take the direct global path)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      local[k] = false;
      rank[k] = 0;
      if (lead[k]) {
        // claim the slot with plain LDS accesses: the lanes that see it free all write their tile, one write wins,
        // everybody re-reads (a wave's LDS operations execute in order; the slot arrays are per wave) — an atomicCAS
        // here serialised the 64 lanes on one address a second time, next to the rank counter below
        volatile int* st = &slot_tile[wv][hs[k]];
        if (*st == -1) *st = tk[k];
        local[k] = *st == tk[k];
        if (local[k]) rank[k] = atomicAdd(&slot_cnt[wv][hs[k]], 1);
        else {
          const int i = atomicAdd(cntp + tk[k], 1);
          put(tk[k], MODE == 2 ? a.bin_start[tile_base + tk[k]] + i : i, rec);
        }
      }
    }
    // 2. ONE global atomic per occupied slot, all slots in parallel (lane = slot)
    {
      const int c = slot_cnt[wv][lane];
      if (c > 0) {
        const int st = slot_tile[wv][lane];
        const int i = atomicAdd(cntp + st, c);
        slot_base[wv][lane] = MODE == 2 ? a.bin_start[tile_base + st] + i : i;
      }
    }
    // 3. write the records
    if constexpr (MODE != 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lead[k] && local[k]) put(tk[k], slot_base[wv][hs[k]] + rank[k], rec);
    }
    slot_tile[wv][lane] = -1;
    slot_cnt[wv][lane] = 0;
  }
  }
}

// Overflow handling (rare: a tile received more than `cap` points).  f32 grad_value (a.ovf_after): the owner tiles
// store first, then lift_ovf_scatter_kernel adds the overflow list atomically on top — one launch that exits at once
// when nothing overflowed.  16-bit grad_value (and the two-stream diagnostic mode) cannot add into the rounded output:
// step A zeroes the f32 scratch map of the overflowed tiles, step B scatters into it BEFORE the owner kernel, which
// adds its bucket sums on top for exactly those tiles and rounds once.
__global__ __launch_bounds__(256) void lift_ovf_zero_kernel(const LiftArgs a, int tiles_x, int tiles,
                                                            int Dh) {
  if (*a.ovf_n == 0) return;                         // (a small grid walks the tiles: the common case is this exit)
  const long total = (long)a.B * a.H * tiles;
  const int lane = threadIdx.x & 63;
  for (long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wave < total; wave += (long)gridDim.x * 4) {
    if (a.bin_cnt[wave] <= a.cap) continue;
    const int tl = (int)(wave % tiles), bh = (int)(wave / tiles);
    const int h = bh % a.H, b = bh / a.H;
    const int x0 = (tl % tiles_x) * 8, y0 = (tl / tiles_x) * 8;
    const long row = (long)a.H * Dh;
    float* gv = a.gvalue + (long)b * a.fh * a.fw * row + h * Dh;
    const int py = lane >> 3, px = lane & 7;
    if (y0 + py < a.fh && x0 + px < a.fw)
      for (int c = 0; c < Dh; ++c) gv[((long)(y0 + py) * a.fw + (x0 + px)) * row + c] = 0.0f;
  }
}

template <typename T, int DH>
__global__ __launch_bounds__(256) void lift_ovf_scatter_kernel(const LiftArgs a, int tiles_x, int tiles) {
  const int n = min(*a.ovf_n, a.ovf_cap);
  if (n == 0) return;
  const long row = (long)a.H * DH;
  const int c = threadIdx.x & 31;                      // channel; 32 threads per overflow entry
  const long stride = (long)gridDim.x * 8;
  for (long e = (long)blockIdx.x * 8 + (threadIdx.x >> 5); e < n; e += stride) {
    const float4 rec = a.ovf_rec[e];
    const int t = a.ovf_tile[e];
    const int tl = t % tiles, bh = t / tiles;
    const int h = bh % a.H, b = bh / a.H;
    const int tx = tl % tiles_x, ty = tl / tiles_x;
    const int q = __float_as_int(rec.w);
    const Footprint f = footprint_px(rec.x, rec.y, a.fh, a.fw);
    if (c >= DH) continue;
    const float go = elem<T>::to_float(((const T*)a.gout)[((long)b * a.Nq + q) * row + h * DH + c]);
    float* gv = a.gvalue + (long)b * a.fh * a.fw * row + h * DH + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float cw = rec.z * f.w[k];
      if (cw != 0.0f && (f.xc[k & 1] >> 3) == tx && (f.yc[k >> 1] >> 3) == ty)
        atomic_add_f32(gv + (long)f.idx[k] * row, cw * go);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, part 2 — grad_value without global atomics: "owner computes", scatter as a matmul.
//
// A wave OWNS a pixel tile (M = 32*RB pixels) of one (sample, camera) value map for one head.  The
// scatter  grad_value[pix, :] += w_p * bilinear_k * grad_out[q, head, :]  over the sampling points
// that touch the tile is the product
//        Tile[M pixels, Dh]  +=  A[M, K points]  .  G[K, Dh]
// with a SPARSE coefficient matrix A (<= 4 non-zeros per column: the point's owned corners) and the
// points' grad_out rows G.  Per round of 32 candidate points:
//   * each point's lane writes its (up to) 4 coefficients into its own column of A in LDS — a
//     conflict-free scatter, no atomics, no read-modify-write — and stages its grad_out row in LDS;
//   * every lane reads MFMA fragments of A and G from LDS (ds_read_b128) and
//     v_mfma_f32_32x32x16 accumulates the tile in REGISTERS (16 f32 per lane per 32 pixels);
//   * the lanes clear the coefficients they wrote.
// Why not LDS accumulation: ds_add_f32 retires ~1 lane per 3 cycles on gfx950 (193 cycles per wave
// instruction measured, tools/ubench/lds_atomic.hip, 24x a plain read+write pair), and a plain
// LDS read-modify-write per point is a ~60-instruction scalar loop bound by its LDS round trip
// (profiles/ notes).  16-bit operands: with f16 / bf16 data the coefficients are rounded to that
// type (the data's own precision); with f32 data both operands are split into bf16 hi + lo parts
// and three products are accumulated (hi*hi + lo*hi + hi*lo, relative error ~2^-16), f32
// accumulation throughout.
//
//   GRID   (BEV-grid queries, one map per sample: self-attention, SCA-pts): 8x8-pixel tiles; the
//          points of a tile come from its bucket, filled by lift_bin_kernel (above).
//   CAMERA (small per-camera maps, arbitrary projected reference points: SCA-img): the tile is a
//          band of full rows (the whole 8x22 map in one band), the points are a share of the
//          camera's compacted visible-query list; the shares of one camera write partial maps
//          ("slabs") that slab_reduce_kernel sums.
// Width of the query grid when the visible lists can be walked tile by tile (whole 8x8 tiles), else 0 (ascending
// order).  UBV_CAM_TILED=0 keeps the ascending order (A/B runs).
static inline int visible_tile_width(int Nq, int qw) {
  static const int tiled = getenv("UBV_CAM_TILED") ? atoi(getenv("UBV_CAM_TILED")) : 1;
  if (!tiled || qw <= 0 || qw % 8 != 0 || Nq % qw != 0 || (Nq / qw) % 8 != 0) return 0;
  return qw;
}

// Ordered compaction of each camera's visible queries (vis0[cam, q] != 0): list[cam, 0..n) holds the
// query indices in ascending order — or, when the queries are a qw-wide BEV grid of whole 8x8 tiles (qw > 0), in
// TILE-MAJOR order (tile by tile, row-major inside a tile): a batch of 32 list entries is then half a tile, a patch
// of the ground a few metres across that lands on a few neighbouring pixels of the camera's map, where 32 entries
// of one grid row are a 16 m line that crosses it.  One 1024-thread block per camera.
__global__ __launch_bounds__(1024) void compact_visible_kernel(const uint8_t* __restrict__ vis0,
                                                               int Nq, int qw, int* __restrict__ list,
                                                               int* __restrict__ n_out) {
  // FOUR walk positions per thread and step (Nq % 4 == 0, checked by the callers' plans: a BEV grid of whole tiles, or
  // any list whose length is a multiple of 4; else one position per thread): the four are consecutive queries in either
  // order — one 4-byte read of vis0 — and the block walks 40 000 queries in 10 steps of (wave scan, two barriers)
  // instead of 40 (the kernel is one block per camera: its time is the length of that chain, 38 -> 14 us).
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int cam = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const int tiles_x = qw >> 3;
  const bool four = (Nq & 3) == 0 && (vis0 == nullptr || (((uintptr_t)vis0) & 3) == 0);
  const int per = four ? 4 : 1;
  for (int q0 = 0; q0 < Nq; q0 += 1024 * per) {
    const int pos = q0 + (int)threadIdx.x * per;
    int q = pos;
    if (qw > 0 && pos < Nq) {                              // position of the tile-major walk -> query index
      const int tile = pos >> 6, in = pos & 63, ty = tile / tiles_x, tx = tile - ty * tiles_x;
      q = (ty * 8 + (in >> 3)) * qw + tx * 8 + (in & 7);
    }
    unsigned bits = 0u;                                     // bit j: query q + j is visible
    if (pos < Nq) {
      if (vis0 == nullptr) bits = four ? 0xfu : 1u;
      else if (four) {
        const uint32_t v4 = *reinterpret_cast<const uint32_t*>(vis0 + (long)cam * Nq + q);
        bits = ((v4 & 0xffu) != 0u) | (((v4 >> 8) & 0xffu) != 0u) << 1 | (((v4 >> 16) & 0xffu) != 0u) << 2 | ((v4 >> 24) != 0u) << 3;
      } else bits = vis0[(long)cam * Nq + q] != 0;
    }
    const int cnt = __popc(bits);
    int incl = cnt;                                         // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_cnt[wv] = incl;
    __syncthreads();
    int off = base_s + incl - cnt;
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    int* dst = list + (long)cam * Nq + off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (bits & (1u << j)) *dst++ = q + j;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; base_s += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) n_out[cam] = base_s;
}

struct TileGeom {
  int b, cam, h, ck, x0, y0, tw, th, npx;
};

__device__ __forceinline__ bool tile_decode(const LiftArgs& a, const TileArgs& t, TileGeom& g) {
  const int item = xcd_remap(blockIdx.x, t.chunk) * (int)(blockDim.x >> 6) + wave_in_block();
  if (item >= t.total) return false;
  // item -> (b, cam, tile_y, tile_x, chunk, head); head fastest: neighbours share rows of grad_out
  int r = item;
  g.h = r % a.H; r /= a.H;
  g.ck = r % t.chunks; r /= t.chunks;
  const int tx = r % t.tiles_x; r /= t.tiles_x;
  const int ty = r % t.tiles_y; r /= t.tiles_y;
  g.cam = r % a.Nc;
  g.b = r / a.Nc;
  g.x0 = tx * t.tile_w; g.y0 = ty * t.tile_h;
  g.tw = min(t.tile_w, a.fw - g.x0); g.th = min(t.tile_h, a.fh - g.y0);
  g.npx = t.tile_w * t.tile_h;
  return true;
}

// Ownership of the 4 corners of a footprint by tile g: fills lp (tile-local pixel index or -1) and
// cwt (coefficient).
__device__ __forceinline__ bool tile_own(const Footprint& f, float w, bool valid, const TileGeom& g,
                                         int tile_w, int (&lp)[4], float (&cwt)[4]) {
  bool any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cx = f.xc[k & 1], cy = f.yc[k >> 1];
    // bitwise, not short-circuit: keeps the four corners straight-line code
    int o = (int)valid & (int)(f.w[k] != 0.0f) & (int)((unsigned)(cx - g.x0) < (unsigned)g.tw) &
            (int)((unsigned)(cy - g.y0) < (unsigned)g.th);
    const bool own = o != 0;
    lp[k] = own ? (cy - g.y0) * tile_w + (cx - g.x0) : -1;
    cwt[k] = own ? w * f.w[k] : 0.0f;
    any = any || own;
  }
  return any;
}

// A rows are 32 point slots = 64 bytes with NO padding; the four 16-byte chunks of a row are XOR-swizzled with
// bits 2..3 of the row index, so the b128 fragment reads of any 16 consecutive rows touch 16 distinct bank groups
// (rows r and r + 4 share a 16-word window and get different chunks).  1 KB per plane less than 80-byte rows:
// an f32 wave's carve drops from 15 to 13 KB, which is what lets 12 waves share a CU.
constexpr int kAStride = 32;
__device__ __forceinline__ int a_index(int row, int slot) {
  return row * kAStride + ((((slot >> 3) ^ (row >> 2)) & 3) << 3) + (slot & 7);
}

// grad_out rows in LDS: DH + 8 halves per row.  With DH = 32 halves (64 bytes) a lane's row started on the same 4
// bank groups as every 4th other lane's: PMC showed 46 % of the f32 owner-tile kernel's LDS cycles as bank
// conflicts (8-way on the 8-byte hi / lo stores).  80-byte rows spread 16 consecutive rows over all 64 banks.
template <int DH> struct GRow { static constexpr int kStride = DH + 8; };

// Per-wave LDS carve (u16 units): A_hi[32*RB][kAStride] (+ A_lo), G_hi[32][DH + 8] (+ G_lo)
template <typename T, int DH, int RB>
struct TileLds {
  static constexpr bool kSplit = mma_traits<T>::kSplit;
  static constexpr int kA = 32 * RB * kAStride;
  static constexpr int kG = 32 * GRow<DH>::kStride;
  static constexpr int kWords = (kA + kG) * (kSplit ? 2 : 1);      // u16 per wave
};

// A grad_out row (held raw in `v`, Dh elements) as 16-bit MFMA operands: for f32 data the row is split into bf16
// hi + lo halves IN PLACE (v[0 .. NV/2) = hi, v[NV/2 .. NV) = lo; 8 channels per uint4) — register work the
// callers do once per batch, outside the per-half staging; 16-bit data stays as it is.
template <typename T, int NV>
__device__ __forceinline__ void split_row(uint4 (&v)[NV]) {
  if constexpr (mma_traits<T>::kSplit) {
    static_assert(NV % 2 == 0, "pairs of 4-channel loads");
    uint4 hi4[NV / 2], lo4[NV / 2];
#pragma unroll
    for (int i = 0; i < NV; i += 2) {
      const float f[8] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z),
                          __uint_as_float(v[i].w), __uint_as_float(v[i + 1].x), __uint_as_float(v[i + 1].y),
                          __uint_as_float(v[i + 1].z), __uint_as_float(v[i + 1].w)};
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = cvt_pk_bf16(f[2 * k], f[2 * k + 1]);
        lo[k] = cvt_pk_bf16(f[2 * k] - __uint_as_float(hi[k] << 16), f[2 * k + 1] - __uint_as_float(hi[k] & 0xffff0000u));
      }
      hi4[i / 2] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      lo4[i / 2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) { v[i] = hi4[i]; v[NV / 2 + i] = lo4[i]; }
  }
}

// Stores a row prepared by split_row in LDS slot `slot`.
template <typename T, int DH, int NV>
__device__ __forceinline__ void stage_row(uint16_t* __restrict__ g_hi, uint16_t* __restrict__ g_lo,
                                          int slot, const uint4 (&v)[NV]) {
  constexpr int GS = GRow<DH>::kStride;
  if constexpr (!mma_traits<T>::kSplit) {
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<uint4*>(g_hi + slot * GS)[i] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      reinterpret_cast<uint4*>(g_hi + slot * GS)[i] = v[i];
      reinterpret_cast<uint4*>(g_lo + slot * GS)[i] = v[NV / 2 + i];
    }
  }
}

// The wave's pending operand tiles: up to 32 point slots filled densely by the owned points of
// successive batches (slot = fill + rank among the batch's owned lanes); when they are full the
// MFMA round runs over all 32 slots and the coefficient tile is cleared.
template <typename T, int DH, int RB>
struct TileAcc {
  using M = mma_traits<T>;
  using L = TileLds<T, DH, RB>;
  uint16_t* a_hi; uint16_t* a_lo; uint16_t* g_hi; uint16_t* g_lo;
  f32x16_t acc[RB];
  int fill;

  __device__ __forceinline__ void init(uint16_t* lds, int lane) {
    a_hi = lds; a_lo = lds + L::kA;
    g_hi = lds + (M::kSplit ? 2 : 1) * L::kA; g_lo = g_hi + L::kG;
    static_assert(L::kWords % 8 == 0, "16-byte vectors");
    for (int i = lane; i < L::kWords / 8; i += 64) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    fill = 0;
  }

  // Tile += A . G over the 32 slots, then clear A.  (One wave: its LDS operations execute in
  // program order, so the fragment reads see the producers' writes without a barrier.)
  __device__ __forceinline__ void flush(int lane) {
    const int n = lane & 31, kg = lane >> 5;
    // B fragment = 8 consecutive slots (K) of one channel: two transposing LDS reads (ds_read_b64_tr_b16: in a
    // 16-lane group lane i addresses the 4-channel piece (row i / 4, quad i % 4) and receives channel i of the 4
    // rows) instead of 8 u16 reads + packing.  Dh = 16: columns 16..31 are don't-care (re-read 0..15).
    constexpr int GS = GRow<DH>::kStride;
    const int tr = (kg * 8 + ((lane & 15) >> 2)) * GS + (DH >= 32 ? ((lane >> 4) & 1) * 16 : 0) + (lane & 3) * 4;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const uint4 b_hi = tr16_frag(g_hi + kb * 16 * GS + tr, GS);
      uint4 b_lo = make_uint4(0, 0, 0, 0);
      if (M::kSplit) b_lo = tr16_frag(g_lo + kb * 16 * GS + tr, GS);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int ao = a_index(rb * 32 + n, kb * 16 + kg * 8);
        const uint4 f_hi = *reinterpret_cast<const uint4*>(a_hi + ao);
        acc[rb] = M::mma(f_hi, b_hi, acc[rb]);
        if (M::kSplit) {
          const uint4 f_lo = *reinterpret_cast<const uint4*>(a_lo + ao);
          acc[rb] = M::mma(f_lo, b_hi, acc[rb]);
          acc[rb] = M::mma(f_hi, b_lo, acc[rb]);
        }
      }
    }
    constexpr int kAVec = (M::kSplit ? 2 : 1) * L::kA / 8;          // 16-byte vectors of A
    for (int i = lane; i < kAVec; i += 64) reinterpret_cast<uint4*>(a_hi)[i] = make_uint4(0, 0, 0, 0);
    fill = 0;
  }

  static constexpr int kNV = DH * elem<T>::kBytes / 16;

  // The lane's grad_out row: grow_base (wave-uniform: sample, head) + grow_off elements (query * row).
  static __device__ __forceinline__ void gather(const T* __restrict__ grow_base, unsigned grow_off, bool want,
                                                uint4 (&grow)[kNV]) {
#pragma unroll
    for (int i = 0; i < kNV; ++i)
      grow[i] = want ? reinterpret_cast<const uint4*>(gather_ptr(grow_base, grow_off))[i] : make_uint4(0, 0, 0, 0);
  }

  // Adds the owned points of one batch (one point per lane) whose grad_out rows are already in registers.
  // The operand conversions (row split, coefficient encoding) run once for the 64 lanes; only the LDS stores sit
  // in the per-half loop (a half's instructions cost the same with 32 lanes masked off).
  __device__ __forceinline__ void add_rows(const int (&lp)[4], const float (&cwt)[4], bool any,
                                           uint4 (&grow)[kNV], int lane) {
    const unsigned long long m = __ballot(any);
    if (m == 0ull) return;
    split_row<T, kNV>(grow);
    uint16_t chi[4], clo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      chi[k] = M::enc(cwt[k]);
      clo[k] = M::kSplit ? M::enc(cwt[k] - M::dec(chi[k])) : (uint16_t)0;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const unsigned int hm = (unsigned int)(m >> (32 * half));
      if (hm == 0u) continue;
      const int cnt = __popc(hm);
      if (fill + cnt > 32) flush(lane);
      if (any && (lane >> 5) == half) {
        const int slot = fill + __popc(hm & ((1u << (lane & 31)) - 1u));
        stage_row<T, DH, kNV>(g_hi, g_lo, slot, grow);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (lp[k] >= 0) {
            const int ai = a_index(lp[k], slot);
            a_hi[ai] = chi[k];
            if (M::kSplit) a_lo[ai] = clo[k];
          }
        }
      }
      fill += cnt;
    }
  }

  // Same, fetching the rows here: the loads are issued first, a pending MFMA round (flush) overlaps their latency.
  __device__ __forceinline__ void add(const int (&lp)[4], const float (&cwt)[4], bool any,
                                      const T* __restrict__ grow_base, unsigned grow_off, int lane) {
    if (__ballot(any) == 0ull) return;
    uint4 grow[kNV];
    gather(grow_base, grow_off, any, grow);
    add_rows(lp, cwt, any, grow, lane);
  }
};

// GRID owner tiles.  One wave per (sample, head, 8x8-pixel tile), blockDim.x / 64 independent waves
// per block; registers capped for 3 waves per SIMD.  The wave walks its bucket (lift_bin_kernel) 64
// records at a time — record loads one batch ahead, contiguous — takes the corners that lie inside
// its tile and accumulates them with TileAcc.  It is the single writer of its pixels: a plain store
// of the finished tile (rounded once for 16-bit outputs), so grad_value needs no zeroing; only a
// tile whose bucket overflowed adds its sums to what lift_ovf_* scattered there.
template <typename T, int DH, int P, int RB>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 4 : 3)) void lift_bwd_value_kernel(const LiftArgs a, const TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  using L = TileLds<T, DH, RB>;
  TileGeom g;
  if (!tile_decode(a, t, g)) return;
  const int lane = threadIdx.x & 63;
  uint16_t* __restrict__ lds = lds_all + wave_in_block() * L::kWords;
  TileAcc<T, DH, RB> ta;
  ta.init(lds, lane);
  const long row = (long)a.H * DH;
  const T* __restrict__ gout = (const T*)a.gout;

  const int tiles = t.tiles_x * t.tiles_y;
  const long bucket = ((long)g.b * a.H + g.h) * tiles + (g.y0 >> 3) * t.tiles_x + (g.x0 >> 3);
  const int cnt_raw = a.bin_cnt[bucket];
  const int n = min(cnt_raw, a.cap);
  const float4* __restrict__ bp = a.bins + bucket * a.cap;
  // Two batches of records and one batch of grad_out rows in flight: a wave is a serial chain of dependent round
  // trips (record -> row of its query -> LDS -> MFMA) and only 2-3 waves fit a SIMD, so the row gather of batch
  // i + 1 is issued before batch i is processed (every record of the bucket has a corner in this tile: the
  // gather is unconditional).
  using TA = TileAcc<T, DH, RB>;
  const T* __restrict__ gbase = gout + (long)g.b * a.Nq * row + g.h * DH;
  // (loads are unconditional on clamped indices — lanes past the end re-read the last record and are masked by
  //  `valid`: predicated vector loads made the compiler wait for each one where it was issued)
  const int last = max(n - 1, 0);
  float4 rec1 = bp[min(lane, last)];
  float4 rec2 = bp[min(64 + lane, last)];
  uint4 rows1[TA::kNV];
  TA::gather(gbase, (unsigned)__float_as_int(rec1.w) * (unsigned)row, true, rows1);
  for (int e0 = 0; e0 < n; e0 += 64) {
    const float4 rec = rec1;
    uint4 rows[TA::kNV];
#pragma unroll
    for (int i = 0; i < TA::kNV; ++i) rows[i] = rows1[i];
    const bool valid = e0 + lane < n;
    rec1 = rec2;
    TA::gather(gbase, (unsigned)__float_as_int(rec1.w) * (unsigned)row, true, rows1);
    rec2 = bp[min(e0 + 128 + lane, last)];
    int lp[4];
    float cwt[4];
    const Footprint f = footprint_px(rec.x, rec.y, a.fh, a.fw);
    const bool any = tile_own(f, rec.z, valid, g, t.tile_w, lp, cwt);
    ta.add_rows(lp, cwt, any, rows, lane);
  }
  if (ta.fill > 0) ta.flush(lane);
  // ---- store the tile: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
  // tiles are 8 pixels wide, so the tile-local pixel index splits with a shift
  const bool rmw = cnt_raw > a.cap && !a.ovf_after;    // lift_ovf_* scattered part of this tile before this kernel
  // pixel (lx, ly) of accumulator element (rb, r): lx = (r & 3) + 4 (lane >> 5), ly = 4 rb + (r >> 2) — the lane
  // part (column, its half's 4-pixel shift) is one 32-bit offset, the (rb, r) part is wave-uniform: scalar
  // address arithmetic instead of a 64-bit multiply-add chain per element (it was a third of a self-attention
  // tile's instructions)
  const long mbase = (long)g.b * a.fh * a.fw * row + g.h * DH;
  const long tile0 = mbase + ((long)g.y0 * a.fw + g.x0) * row;         // wave-uniform
  const int col = lane & 31, lxh = 4 * (lane >> 5);
  if (col < DH) {
    const unsigned lane_off = (unsigned)(lxh * (int)row + col);
    const unsigned rowstride = (unsigned)(a.fw * (int)row);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lx = (r & 3) + lxh, ly = 4 * rb + (r >> 2);
        if (lx < g.tw && ly < g.th) {
          const unsigned uo = (unsigned)ly * rowstride + (unsigned)(r & 3) * (unsigned)row;     // uniform
          float v = ta.acc[rb][r];
          if (sizeof(T) == 2 && a.gvalue_lp != nullptr) {
            T* dst = (T*)a.gvalue_lp + tile0 + uo;
            if (rmw) v += (a.gvalue + tile0 + uo)[lane_off];
            dst[lane_off] = elem<T>::from_float(v);
          } else {
            float* dst = a.gvalue + tile0 + uo;
            if (rmw) v += dst[lane_off];
            dst[lane_off] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GRID owner tiles from QUERY records (round 6, f32 TILE instances; the query kernel's BINS = 2 mode).  A bucket holds the
// indices of the (query, head) pairs with a point in the tile — 5x fewer entries than points-in-tile records — and the
// pair's points come from qpts (written once, coalesced, by the query kernel instead of a 16-byte record per point and
// owner tile).  Per round of 32 entries, lanes l and l + 32 share an entry and split its P points (as the matrix-core
// camera kernels do): each builds the footprints of its half, adds the owned corners' coefficients into column l & 31 of
// A[64 pixels][32 slots] — ONE dword per coefficient, bf16 hi | lo << 16, read-modify-write by the one lane pair that owns
// the column (the halves take turns: two lanes of a pair may hit one pixel) — and stages its half of the entry's
// grad_out row.  One MFMA round (K = 32) per 32 ENTRIES where the point-record kernel runs one per 32 POINTS: a fifth of
// the row gathers, f32 -> bf16 splits, LDS stagings and MFMA rounds; the footprint arithmetic stays (every point of an
// entry is expanded, about half of them land in the tile).  Exact for any offsets: entries beyond a bucket's capacity and
// points outside the query block's 8 x 8 neighbourhood of tiles went to the overflow list as point records.
constexpr int kQStride = 32;                                // dwords per A row (32 slots)
__device__ __forceinline__ int aq_index(int row, int slot) {
  // 16-byte chunk c of row r sits at position c ^ (r & 7): the b128 fragment reads of 8 consecutive rows touch 8 bank groups
  return row * kQStride + ((((slot >> 2) ^ row) & 7) << 2) + (slot & 3);
}

template <int P>
__global__ __launch_bounds__(256, 3) void lift_bwd_value_q_kernel(const LiftArgs a, const TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  constexpr int DH = 32, RB = 2, PH = P / 2, GS = GRow<DH>::kStride;
  using M = mma_traits<bf16_t>;
  constexpr int kAWords = 64 * kQStride;                    // dwords of A
  constexpr int kWaveBytes = kAWords * 4 + 2 * 32 * GS * 2; // A + G_hi + G_lo
  TileGeom g;
  if (!tile_decode(a, t, g)) return;
  const int lane = threadIdx.x & 63, slot = lane & 31, half = lane >> 5;
  unsigned char* base = reinterpret_cast<unsigned char*>(lds_all) + wave_in_block() * kWaveBytes;
  uint32_t* A = reinterpret_cast<uint32_t*>(base);
  uint16_t* g_hi = reinterpret_cast<uint16_t*>(base + kAWords * 4);
  uint16_t* g_lo = g_hi + 32 * GS;
  for (int i = lane; i < kWaveBytes / 16; i += 64) reinterpret_cast<uint4*>(base)[i] = make_uint4(0u, 0u, 0u, 0u);
  f32x16_t acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
  const long row = (long)a.H * DH;
  const int tiles = t.tiles_x * t.tiles_y;
  const long bucket = ((long)g.b * a.H + g.h) * tiles + (g.y0 >> 3) * t.tiles_x + (g.x0 >> 3);
  const int cnt_raw = a.bin_cnt[bucket];
  const int n = min(cnt_raw, a.cap);
  const int* __restrict__ qb = reinterpret_cast<const int*>(a.bins) + bucket * a.cap;
  const float* __restrict__ gbase = (const float*)a.gout + (long)g.b * a.Nq * row + g.h * DH + half * 16;
  const float* __restrict__ pbase = a.qpts + ((long)g.b * a.Nq * a.H + g.h) * (P * 3) + half * (PH * 3);
  const int last = max(n - 1, 0);
  int qn = qb[min(slot, last)];
  for (int e0 = 0; e0 < n; e0 += 32) {
    const int q = qn;
    const bool valid = e0 + slot < n;
    qn = qb[min(e0 + 32 + slot, last)];                   // next round's entry: in flight during this one
    // this lane's half of the entry: PH points and 16 channels of its grad_out row
    float pt[PH * 3];
    {
      const float* pp = pbase + (long)q * a.H * (P * 3);
      if constexpr ((PH * 3) % 4 == 0) {
#pragma unroll
        for (int i = 0; i < PH * 3; i += 4) {
          const float4 v = *reinterpret_cast<const float4*>(pp + i);
          pt[i] = v.x; pt[i + 1] = v.y; pt[i + 2] = v.z; pt[i + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < PH * 3; i += 2) {
          const float2 v = *reinterpret_cast<const float2*>(pp + i);
          pt[i] = v.x; pt[i + 1] = v.y;
        }
      }
    }
    uint4 grow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) grow[i] = reinterpret_cast<const uint4*>(gather_ptr(gbase, (unsigned)q * (unsigned)row))[i];
    // coefficients: the halves take turns (both lanes of a pair write column `slot`)
    int lp[PH][4];
    float cw[PH][4];
#pragma unroll
    for (int i = 0; i < PH; ++i) {
      const Footprint f = footprint_px(pt[3 * i], pt[3 * i + 1], a.fh, a.fw);
      tile_own(f, pt[3 * i + 2], valid, g, t.tile_w, lp[i], cw[i]);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (half == hh) {
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          uint32_t e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = A[aq_index(max(lp[i][k], 0), slot)];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (lp[i][k] >= 0) A[aq_index(lp[i][k], slot)] = coef_add(e[k], cw[i][k]);
        }
      }
    }
    // this lane's 16 channels: f32 -> bf16 hi / lo, into row `slot` of G
    {
      const float f[16] = {__uint_as_float(grow[0].x), __uint_as_float(grow[0].y), __uint_as_float(grow[0].z), __uint_as_float(grow[0].w),
                           __uint_as_float(grow[1].x), __uint_as_float(grow[1].y), __uint_as_float(grow[1].z), __uint_as_float(grow[1].w),
                           __uint_as_float(grow[2].x), __uint_as_float(grow[2].y), __uint_as_float(grow[2].z), __uint_as_float(grow[2].w),
                           __uint_as_float(grow[3].x), __uint_as_float(grow[3].y), __uint_as_float(grow[3].z), __uint_as_float(grow[3].w)};
      uint4 hi[2], lo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float f8[8] = {f[8 * i], f[8 * i + 1], f[8 * i + 2], f[8 * i + 3], f[8 * i + 4], f[8 * i + 5], f[8 * i + 6], f[8 * i + 7]};
        split8(f8, hi[i], lo[i]);
        if (!valid) { hi[i] = make_uint4(0u, 0u, 0u, 0u); lo[i] = hi[i]; }
      }
      uint4* dh = reinterpret_cast<uint4*>(g_hi + slot * GS + half * 16);
      uint4* dl = reinterpret_cast<uint4*>(g_lo + slot * GS + half * 16);
      dh[0] = hi[0]; dh[1] = hi[1];
      dl[0] = lo[0]; dl[1] = lo[1];
    }
    // Tile += A . G over the 32 slots (one wave: its LDS operations execute in program order), then clear A
    {
      const int nn = lane & 31, kg = lane >> 5;
      const int tr = (kg * 8 + ((lane & 15) >> 2)) * GS + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const uint4 b_hi = tr16_frag(g_hi + kb * 16 * GS + tr, GS);
        const uint4 b_lo = tr16_frag(g_lo + kb * 16 * GS + tr, GS);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int r = rb * 32 + nn, s0 = kb * 16 + kg * 8;
          const uint4 e0v = *reinterpret_cast<const uint4*>(A + aq_index(r, s0));
          const uint4 e1v = *reinterpret_cast<const uint4*>(A + aq_index(r, s0 + 4));
          const uint4 f_hi = make_uint4((e0v.x & 0xffffu) | (e0v.y << 16), (e0v.z & 0xffffu) | (e0v.w << 16),
                                        (e1v.x & 0xffffu) | (e1v.y << 16), (e1v.z & 0xffffu) | (e1v.w << 16));
          const uint4 f_lo = make_uint4((e0v.x >> 16) | (e0v.y & 0xffff0000u), (e0v.z >> 16) | (e0v.w & 0xffff0000u),
                                        (e1v.x >> 16) | (e1v.y & 0xffff0000u), (e1v.z >> 16) | (e1v.w & 0xffff0000u));
          acc[rb] = M::mma(f_hi, b_hi, acc[rb]);
          acc[rb] = M::mma(f_lo, b_hi, acc[rb]);
          acc[rb] = M::mma(f_hi, b_lo, acc[rb]);
        }
      }
      for (int i = lane; i < kAWords / 4; i += 64) reinterpret_cast<uint4*>(A)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  // ---- store the tile (as lift_bwd_value_kernel: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  const long mbase = (long)g.b * a.fh * a.fw * row + g.h * DH;
  const long tile0 = mbase + ((long)g.y0 * a.fw + g.x0) * row;
  const int col = lane & 31, lxh = 4 * (lane >> 5);
  const unsigned lane_off = (unsigned)(lxh * (int)row + col);
  const unsigned rowstride = (unsigned)(a.fw * (int)row);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lx = (r & 3) + lxh, ly = 4 * rb + (r >> 2);
      if (lx < g.tw && ly < g.th) {
        const unsigned uo = (unsigned)ly * rowstride + (unsigned)(r & 3) * (unsigned)row;
        (a.gvalue + tile0 + uo)[lane_off] = acc[rb][r];
      }
    }
  }
}

#include "bev_lift_maps.inl"
#include "dcn_owner.inl"

// ------------------------------------------------------------------------------------------------
// CAMERA owner tiles, one lane per QUERY: a round trip to memory fetches everything 64 queries need
// (list entry -> offsets, logits, anchors, count, grad_out row), double-buffered in registers, and
// buys 64*P sampling points of work, so the fetch latency hides under the MFMA phase instead of
// being paid once per 64 points.  Per batch the 64 grad_out rows are staged in LDS once and their
// B fragments stay in registers for all P slots; per slot p the 64 points (one per lane) scatter
// their <= 4 coefficients into column `lane` of A[pixels][64], the touched 32-pixel row blocks run
// 4 MFMAs (K = 64) each, and the lanes clear exactly what they wrote.
// A camera's visible-query list is dealt evenly, in whole batches of 64, to `chunks` waves.
__device__ __forceinline__ int cam_chunk_len(int n, int chunks) {
  return ((n + chunks * 64 - 1) / (chunks * 64)) * 64;
}

// Balanced shares (matrix-core CAMERA plans).  A fixed number of chunks per camera gives the wave of a busy camera
// more queries than the wave of a quiet one — 9 526 against 6 076 visible queries between the front and a side camera
// of the nuScenes rig, and the kernel ends with its slowest wave.  Here the W = Nc x chunks waves of a (sample, head)
// cut every camera's list into pieces of ONE length Lq = T / (W - Nc) rounded up to 64 (T = all visible pairs), camera
// c taking ceil(n_c / Lq) consecutive waves: sum <= W, every piece <= Lq.  Waves past the sum have nothing to do.
struct CamShare { int cam, l0, ncand, first, count; };
__device__ __forceinline__ int cam_share_len(const LiftArgs& a, int W) {
  int T = 0;
  for (int c = 0; c < a.Nc; ++c) T += a.cam_n[c];
  const int den = W - a.Nc > 1 ? W - a.Nc : 1;
  const int lq = (((T + den - 1) / den + 63) / 64) * 64;
  return lq > 64 ? lq : 64;
}
// wave k of W -> its camera and list range; false: idle
__device__ __forceinline__ bool cam_share(const LiftArgs& a, int W, int k, CamShare& s) {
  const int lq = cam_share_len(a, W);
  int base = 0;
  for (int c = 0; c < a.Nc; ++c) {
    const int n = a.cam_n[c], w = (n + lq - 1) / lq;
    if (k < base + w) {
      s.cam = c; s.l0 = (k - base) * lq; s.ncand = min(lq, n - s.l0); s.first = base; s.count = w;
      return true;
    }
    base += w;
  }
  return false;
}
// the waves that hold partial maps of camera `cam`
__device__ __forceinline__ void cam_share_of(const LiftArgs& a, int W, int cam, int& first, int& count) {
  const int lq = cam_share_len(a, W);
  first = 0;
  for (int c = 0; c < cam; ++c) first += (a.cam_n[c] + lq - 1) / lq;
  count = (a.cam_n[cam] + lq - 1) / lq;
}

constexpr int kCStride = 72;      // u16 per A row: 64 query columns + 8 pad (144 B: conflict-free b128)

template <typename T, int DH, int RB>
struct CamLds {
  static constexpr bool kSplit = mma_traits<T>::kSplit;
  static constexpr int kA = 32 * RB * kCStride;
  static constexpr int kG = 64 * GRow<DH>::kStride;
  static constexpr int kWords = (kA + kG) * (kSplit ? 2 : 1);      // u16 per wave
};

template <typename T, int DH, int P, int RB>
__global__ __launch_bounds__(256) void lift_bwd_value_camera_kernel(const LiftArgs a, const TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  using M = mma_traits<T>;
  using L = CamLds<T, DH, RB>;
  constexpr int NV = DH * elem<T>::kBytes / 16;
  TileGeom g;
  if (!tile_decode(a, t, g)) return;
  const int lane = threadIdx.x & 63;
  const int cq = cam_chunk_len(a.cam_n[g.cam], t.chunks);
  const int l0 = g.ck * cq;                               // offset into this camera's list
  const int ncand = min(cq, a.cam_n[g.cam] - l0);
  if (ncand <= 0) return;                                 // nothing visible in this chunk
  uint16_t* __restrict__ lds = lds_all + (threadIdx.x >> 6) * L::kWords;
  uint16_t* __restrict__ a_hi = lds;
  uint16_t* __restrict__ a_lo = lds + L::kA;
  uint16_t* __restrict__ g_hi = lds + (M::kSplit ? 2 : 1) * L::kA;
  uint16_t* __restrict__ g_lo = g_hi + L::kG;
  for (int i = lane; i < L::kWords / 8; i += 64) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  f32x16_t acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;

  const long row = (long)a.H * DH;
  const T* __restrict__ gout = (const T*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const int n = lane & 31, kg = lane >> 5;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;

  struct Raw {
    float off[2 * P], lg[P];
    float2 ref[P];
    float cnt;
    uint4 grow[NV];
    bool valid;
  };
  auto fetch_q = [&](int c0, bool& valid) -> int {
    const int c = c0 + lane;
    valid = c < ncand;
    return a.cam_list[(long)g.cam * a.Nq + l0 + (valid ? c : 0)];
  };
  auto fetch_raw = [&](int q, bool valid, Raw& rw) {
    rw.valid = valid;
    const long bq = (long)g.b * a.Nq + q;
    load_ol<T, P>(a.logits, bq * a.log_stride + g.h * P, a.ol16, rw.lg);
    load_ol<T, 2 * P>(a.offsets, bq * a.off_stride + g.h * 2 * P, a.ol16, rw.off);
    const float* rp = a.ref + (((long)g.cam * a.B + g.b) * a.Nq + q) * a.Z * 2;
    int zi = 0;                                           // anchor of flat point p is p % Z
#pragma unroll
    for (int p = 0; p < P; ++p) {
      rw.ref[p] = *reinterpret_cast<const float2*>(rp + zi * 2);
      zi = (zi + 1 == a.Z) ? 0 : zi + 1;
    }
    rw.cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
    const uint4* gp = reinterpret_cast<const uint4*>(gout + bq * row + g.h * DH);
#pragma unroll
    for (int i = 0; i < NV; ++i) rw.grow[i] = gp[i];
  };

  bool v1, v2;
  Raw cur, nxt;
  const int q1 = fetch_q(0, v1);
  fetch_raw(q1, v1, nxt);
  int q2 = fetch_q(64, v2);
  for (int c0 = 0; c0 < ncand; c0 += 64) {
    cur = nxt;
    if (c0 + 64 < ncand) {
      fetch_raw(q2, v2, nxt);
      q2 = fetch_q(c0 + 128, v2);
    }
    // grad_out rows of the 64 queries -> LDS, then this lane's B fragments for the 4 K-blocks
    split_row<T, NV>(cur.grow);
    stage_row<T, DH, NV>(g_hi, g_lo, lane, cur.grow);
    uint4 b_hi[4], b_lo[4];
    {
      constexpr int GS = GRow<DH>::kStride;                // two transposing reads per fragment (see TileAcc::flush)
      const int tr = (kg * 8 + ((lane & 15) >> 2)) * GS + (DH >= 32 ? ((lane >> 4) & 1) * 16 : 0) + (lane & 3) * 4;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        b_hi[kb] = tr16_frag(g_hi + kb * 16 * GS + tr, GS);
        b_lo[kb] = M::kSplit ? tr16_frag(g_lo + kb * 16 * GS + tr, GS) : make_uint4(0, 0, 0, 0);
      }
    }
    float w[P];
    softmax_row<P>(cur.lg, w);
    const float inv_cnt = 1.0f / cur.cnt;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      int lp[4];
      float cwt[4];
      // reciprocal multiplies: the location may differ from the forward's by an ulp, to which the
      // scattered gradient is continuous
      const Footprint f = make_footprint(cur.ref[p].x + cur.off[2 * p] * inv_fw,
                                         cur.ref[p].y + cur.off[2 * p + 1] * inv_fh, a.fh, a.fw);
      const bool any = tile_own(f, w[p] * inv_cnt, cur.valid, g, t.tile_w, lp, cwt);
      if (__ballot(any) == 0ull) continue;
      unsigned touched = 0;                               // 32-pixel row blocks this lane writes
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (lp[k] >= 0) {
          const uint16_t hi = M::enc(cwt[k]);
          a_hi[lp[k] * kCStride + lane] = hi;
          if (M::kSplit) a_lo[lp[k] * kCStride + lane] = M::enc(cwt[k] - M::dec(hi));
          touched |= 1u << (lp[k] >> 5);
        }
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (__ballot((touched >> rb) & 1u) == 0ull) continue;       // wave-uniform skip
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int ao = (rb * 32 + n) * kCStride + kb * 16 + kg * 8;
          const uint4 f_hi = *reinterpret_cast<const uint4*>(a_hi + ao);
          acc[rb] = M::mma(f_hi, b_hi[kb], acc[rb]);
          if (M::kSplit) {
            const uint4 f_lo = *reinterpret_cast<const uint4*>(a_lo + ao);
            acc[rb] = M::mma(f_lo, b_hi[kb], acc[rb]);
            acc[rb] = M::mma(f_hi, b_lo[kb], acc[rb]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (lp[k] >= 0) {
          a_hi[lp[k] * kCStride + lane] = 0;
          if (M::kSplit) a_lo[lp[k] * kCStride + lane] = 0;
        }
      }
    }
  }
  // ---- this chunk's partial map -> its slab ([b][cam][h][chunk][S][Dh], plain stores);
  // slab_reduce_kernel sums the active chunks.  D layout: col = lane & 31,
  // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = lane & 31;
  if (col < DH) {
    float* __restrict__ slab = a.slab + ((((long)g.b * a.Nc + g.cam) * a.H + g.h) * t.chunks + g.ck) *
                                            ((long)a.fh * a.fw * DH);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // bands are whole rows (tile_w == fw, x0 == 0): the band-local pixel index is linear
        const int px = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (px < g.th * a.fw) slab[((long)g.y0 * a.fw + px) * DH + col] = acc[rb][r];
      }
    }
  }
}

// Sums the partial maps of a camera's active list chunks into grad_value (plain stores: every
// element of grad_value is written exactly once, so no zeroing and no atomics).
template <typename T>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const LiftArgs a, int chunks, int Dh, int balanced) {
  const long per_map = (long)a.fh * a.fw * Dh;                 // one (b, cam, h) map
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.Nc * a.H * per_map;
  if (t >= total) return;
  const long m = t / per_map, e = t - m * per_map;             // m = (b*Nc + cam)*H + h
  const int h = (int)(m % a.H);
  const int cam = (int)((m / a.H) % a.Nc);
  const int b = (int)(m / ((long)a.H * a.Nc));
  float s = 0.0f;
  if (balanced) {                                     // slabs [b][h][W waves], camera `cam` owns waves first .. first + count
    const int W = a.Nc * chunks;
    int first, count;
    cam_share_of(a, W, cam, first, count);
    const long sb = ((long)b * a.H + h) * W + first;
    for (int c = 0; c < count; ++c) s += a.slab[(sb + c) * per_map + e];
  } else {
    const int cq = cam_chunk_len(a.cam_n[cam], chunks);
    const int nact = cq > 0 ? min(chunks, (a.cam_n[cam] + cq - 1) / cq) : 0;
    for (int c = 0; c < nact; ++c) s += a.slab[(m * chunks + c) * per_map + e];
  }
  const long px = e / Dh, col = e - px * Dh;
  const long o = (((long)b * a.Nc + cam) * a.fh * a.fw + px) * ((long)a.H * Dh) + h * Dh + col;
  if (sizeof(T) == 2 && a.gvalue_lp != nullptr)
    ((T*)a.gvalue_lp)[o] = elem<T>::from_float(s);
  else
    a.gvalue[o] = s;
}

// grad_value f32 -> the value's 16-bit type (only the all-atomics plan needs this separate pass)
template <typename T>
__global__ __launch_bounds__(256) void narrow_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                     long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = elem<T>::from_float(src[i]);
}

#include "bev_lift_cam.inl"
#include "bev_lift_cam32.inl"
#include "bev_lift_shared.inl"
#include "bev_lift_win.inl"

// ---- dispatch ----------------------------------------------------------------------------------------
// Algorithmic (compulsory) bytes of each kernel: every operand read once, every result written
// once, gathers not counted.
struct LiftBytes {
  double value, value_f32, offlog, ref, vis, out, rec;
};
static LiftBytes lift_bytes(const LiftArgs& a, int Dh, int P, int esize) {
  LiftBytes b;
  const double C = (double)a.H * Dh, S = (double)a.fh * a.fw;
  b.value = a.B * a.Nc * S * C * esize;
  b.value_f32 = a.B * a.Nc * S * C * 4.0;
  b.offlog = (double)a.B * a.Nq * a.H * P * 3 * (a.ol16 ? 2 : 4);
  b.ref = (double)a.Nc * a.B * a.Nq * a.Z * 2 * 4;
  b.vis = a.Nc > 1 ? (double)a.Nc * a.Nq + (double)a.B * a.Nq * 4 : 0.0;
  b.out = (double)a.B * a.Nq * C * esize;
  b.rec = (double)a.B * a.Nq * a.H * P * 16;
  return b;
}

// CAMERA plan on the matrix cores (bev_lift_cam.inl; f32 data with split operands: bev_lift_cam32.inl): Dh = 32,
// P = 8, maps of <= 192 pixels.  UBV_CAM_MFMA=0 switches it off.  f32 data (UBV_CAM_MFMA32: 1 = default, 0 = off,
// 2 = backward kernels only), measured at 6 x 8x22, bs = 2 against the kernels it replaces: query gradient 167 -> 140 us,
// value gradient 279 -> 156 us, forward 150 -> 123 us (its first version, whole map per wave: 190 us).
static bool cam_mfma_ok(const LiftArgs& a, int Dh, int P, int dtype, bool fwd = false) {
  static const int env = getenv("UBV_CAM_MFMA") ? atoi(getenv("UBV_CAM_MFMA")) : 1;
  static const int env32 = getenv("UBV_CAM_MFMA32") ? atoi(getenv("UBV_CAM_MFMA32")) : 1;
  return env != 0 && (dtype != UBV_F32 || (env32 != 0 && !(fwd && env32 == 2))) && Dh == 32 && P == 8 && a.fh >= 1 && a.fw >= 1 &&
         a.fh <= 13 && cam_kpad(a.fh, a.fw) <= 16 * kCamKbMax;
}
// fragment-ordered copy of value: one buffer of 16-bit fragments, two (hi, lo) for f32 data
static size_t cam_vfrag_bytes(const LiftArgs& a, int dtype = UBV_BF16) {
  const size_t KB = cam_kpad(a.fh, a.fw) <= 14 * 16 ? 14 : 15;
  return (size_t)a.B * a.Nc * a.H * KB * 64 * 16 * (dtype == UBV_F32 ? 2 : 1);
}
// f32 backward: bf16 hi + lo copies of value in its own layout
static size_t cam_vnat_bytes(const LiftArgs& a, int Dh) {
  return (((size_t)a.B * a.Nc * a.fh * a.fw * a.H * Dh * 2) + 255) & ~(size_t)255;
}
static CamArgs cam_args(const LiftArgs& a, const void* vfrag, int dtype = UBV_BF16) {
  CamArgs c{};
  c.vfrag = vfrag;
  if (vfrag != nullptr && dtype == UBV_F32) c.vfrag_lo = (const char*)vfrag + cam_vfrag_bytes(a);
  c.KB = cam_kpad(a.fh, a.fw) <= 14 * 16 ? 14 : 15;       // 14 covers the 8x22 maps
  c.witems = a.total_tiles * 2 * a.H;                      // a wave is half a tile for one head
  c.chunk = (c.witems + 7) / 8;
  c.fh1 = a.fh + 1;
  c.mg = (65536u + (unsigned)c.fh1 - 1u) / (unsigned)c.fh1;
  return c;
}

// Heads per block of the shared-footprint kernels (bev_lift_shared.inl, shared_item): 0 = all heads in one block.
// Single large maps (self-attention, SCA-pts, the decoder's cross-attention) are walked one head group at a time
// so that an XCD's resident blocks gather from a slice of the map that fits its L2; the small per-camera maps stay in
// L2 anyway.  UBV_LIFT_HB = 0 | 1 | 2 | 4 forces a value.
static int shared_hb(const LiftArgs& a, int esize, int lp, bool bwd) {
  static const int env = getenv("UBV_LIFT_HB") ? atoi(getenv("UBV_LIFT_HB")) : -1;
  int hb = env;
  if (hb < 0) {
    // measured at bs = 2 on the 200x200 / 180x180 maps (profiles/r03_lift_hb_{fp32,bf16}.txt), us for HB = 0 / 1 / 2 / 4:
    //   f32  self-attn forward 96 / 82 / 89 / 93, query gradient 108 / 112 / 109 / 106
    //   bf16 self-attn forward 50 / - / 49 / 48, query gradient 52 / - / 44 / 43; SCA-pts 82 / - / 78 / 77, 94 / - / 81 / 78
    const double map_bytes = (double)a.fh * a.fw * a.H * 32 * esize;        // one sample's map (Dh <= 32)
    hb = (a.Nc == 1 && map_bytes > 8e6) ? ((esize == 4 && !bwd) ? 1 : 4) : 0;
  }
  if (hb <= 0 || hb >= a.H || a.H % hb != 0 || 64 % (hb * lp) != 0) return 0;
  return hb;
}

template <typename T, int DH, int VEC, int P, bool OL16>
static void launch_fwd_shared(const LiftArgs& a, hipStream_t st) {
  const int hb = shared_hb(a, sizeof(T), DH / VEC, false);
  const unsigned blocks = 8u * a.chunk * (hb ? a.H / hb : 1);
  if (hb == 1) hipLaunchKernelGGL((lift_fwd_shared_kernel<T, DH, VEC, P, OL16, 1>), dim3(blocks), dim3(256), 0, st, a);
  else if (hb == 2) hipLaunchKernelGGL((lift_fwd_shared_kernel<T, DH, VEC, P, OL16, 2>), dim3(blocks), dim3(256), 0, st, a);
  else if (hb == 4) hipLaunchKernelGGL((lift_fwd_shared_kernel<T, DH, VEC, P, OL16, 4>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((lift_fwd_shared_kernel<T, DH, VEC, P, OL16, 0>), dim3(8u * a.chunk), dim3(256), 0, st, a);
}

template <typename T, int DH, int VEC, int P, bool OL16>
static void launch_bwd_query_shared(const LiftArgs& a, hipStream_t st) {
  const int hb = shared_hb(a, sizeof(T), DH / VEC, true);
  const unsigned blocks = 8u * a.chunk * (hb ? a.H / hb : 1);
  if (hb == 1) hipLaunchKernelGGL((lift_bwd_query_shared_kernel<T, DH, VEC, P, OL16, 1>), dim3(blocks), dim3(256), 0, st, a);
  else if (hb == 2) hipLaunchKernelGGL((lift_bwd_query_shared_kernel<T, DH, VEC, P, OL16, 2>), dim3(blocks), dim3(256), 0, st, a);
  else if (hb == 4) hipLaunchKernelGGL((lift_bwd_query_shared_kernel<T, DH, VEC, P, OL16, 4>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((lift_bwd_query_shared_kernel<T, DH, VEC, P, OL16, 0>), dim3(8u * a.chunk), dim3(256), 0, st, a);
}

template <typename T, int DH, int P>
static void lift_launch(const LiftArgs& a, const TileArgs& t, int bwd_mode, bool cam_mfma, void* fwd_ws,
                        hipStream_t st) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  // shared-footprint kernels on f32 data: 8 channels per lane (two 16-byte loads per corner, 4 lanes per (query, head),
  // DPP broadcasts) for the camera instances — 183 -> 148 us forward, 196 -> 175 us backward at 6 x 8x22 — and 4
  // (8 lanes, ds_swizzle broadcasts) for the single-map ones, which lose 5 % with the wider lanes.
  // UBV_LIFT_F32_VEC = 4 | 8 forces one.
  constexpr int VECS = 8;
  static const int f32_vec = getenv("UBV_LIFT_F32_VEC") ? atoi(getenv("UBV_LIFT_F32_VEC")) : 0;
  const bool wide = sizeof(T) == 2 || (f32_vec == 0 ? a.Nc > 1 : f32_vec == 8);
  const int blocks = 8 * a.chunk;
  const LiftBytes nb = lift_bytes(a, DH, P, elem<T>::kBytes);
  char tag[96];
  auto name = [&](const char* k) {
    snprintf(tag, sizeof(tag), "%s<P=%d,Dh=%d,%dB> Nc=%d map=%dx%d Nq=%d B=%d", k, P, DH,
             elem<T>::kBytes, a.Nc, a.fh, a.fw, a.Nq, a.B);
    return tag;
  };
  if (bwd_mode < 0) {
    ProfScope ps(name("bev_lift_fwd"), st, nb.value + nb.offlog + nb.ref + nb.vis + nb.out);
    if constexpr (DH == 32 && P == 8 && sizeof(T) == 2) {
      if (cam_mfma) {
        const CamArgs c = cam_args(a, fwd_ws);
        const long nthreads = (long)a.B * a.Nc * a.H * c.KB * 64;
        hipLaunchKernelGGL(value_frags_kernel<T>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                           (const T*)a.value, (T*)fwd_ws, a.B * a.Nc, a.fh * a.fw, a.H, a.fh, a.fw, c.KB);
        const size_t lds = (size_t)32 * (c.KB * 16 + 4) * sizeof(uint16_t);
        const dim3 grid(8 * c.chunk), blk(64);
        if (c.KB == 14) {
          if (a.ol16) hipLaunchKernelGGL((lift_cam_fwd_kernel<T, 8, true, 14>), grid, blk, lds, st, a, c);
          else hipLaunchKernelGGL((lift_cam_fwd_kernel<T, 8, false, 14>), grid, blk, lds, st, a, c);
        } else {
          if (a.ol16) hipLaunchKernelGGL((lift_cam_fwd_kernel<T, 8, true, 15>), grid, blk, lds, st, a, c);
          else hipLaunchKernelGGL((lift_cam_fwd_kernel<T, 8, false, 15>), grid, blk, lds, st, a, c);
        }
        return;
      }
    }
    if constexpr (DH == 32 && P == 8 && sizeof(T) == 4) {
      if (cam_mfma) {                            // f32 data: split operands (bev_lift_cam32.inl)
        const CamArgs c = cam_args(a, fwd_ws, UBV_F32);
        const long nthreads = (long)a.B * a.Nc * a.H * c.KB * 64;
        hipLaunchKernelGGL(value_frags32_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st,
                           (const float*)a.value, (uint16_t*)c.vfrag, (uint16_t*)c.vfrag_lo, a.B * a.Nc, a.fh * a.fw,
                           a.H, a.fh, a.fw, c.KB);
        const size_t lds = (size_t)32 * kCam32AStr * sizeof(uint32_t);
        const dim3 grid(8 * c.chunk), blk(64);
        if (c.KB == 14) hipLaunchKernelGGL((lift_cam32_fwd_kernel<8, 14>), grid, blk, lds, st, a, c);
        else hipLaunchKernelGGL((lift_cam32_fwd_kernel<8, 15>), grid, blk, lds, st, a, c);
        return;
      }
    }
    static const int shared_env = getenv("UBV_LIFT_SHARED") ? atoi(getenv("UBV_LIFT_SHARED")) : 1;
    constexpr int DT = sizeof(T) == 4 ? UBV_F32 : std::is_same<T, f16_t>::value ? UBV_F16 : UBV_BF16;
    if (tile_ok(a, DH, P, DT)) {                 // one lane per (query, point), LDS window (bev_lift_tile.hip)
      tile_fwd_launch(a, P, st, false, DH, DT);
      return;
    }
    if (shared_env && win_ok<T, DH, P>(a)) {   // BEV-grid queries: corners served from an LDS window (bev_lift_win.inl)
      constexpr int HG = 128 / (DH * (int)sizeof(T));
      const int chunk = (int)(((long)a.total_tiles * (a.H / HG) + 7) / 8);
      const dim3 grid(8 * chunk);
      if (sizeof(T) == 2 && a.ol16) {
        if (wide) hipLaunchKernelGGL((lift_fwd_win_kernel<T, DH, VECS, P, sizeof(T) == 2, HG>), grid, dim3(256), kWinLds, st, a, chunk);
        else hipLaunchKernelGGL((lift_fwd_win_kernel<T, DH, VEC, P, sizeof(T) == 2, HG>), grid, dim3(256), kWinLds, st, a, chunk);
      } else {
        if (wide) hipLaunchKernelGGL((lift_fwd_win_kernel<T, DH, VECS, P, false, HG>), grid, dim3(256), kWinLds, st, a, chunk);
        else hipLaunchKernelGGL((lift_fwd_win_kernel<T, DH, VEC, P, false, HG>), grid, dim3(256), kWinLds, st, a, chunk);
      }
      return;
    }
    if (shared_env) {            // per-point arithmetic shared inside the lane group (bev_lift_shared.inl)
      if (sizeof(T) == 2 && a.ol16)
        {
          if (wide) launch_fwd_shared<T, DH, VECS, P, sizeof(T) == 2>(a, st);
          else launch_fwd_shared<T, DH, VEC, P, sizeof(T) == 2>(a, st);
        }
      else
        {
          if (wide) launch_fwd_shared<T, DH, VECS, P, false>(a, st);
          else launch_fwd_shared<T, DH, VEC, P, false>(a, st);
        }
      return;
    }
    if (sizeof(T) == 2 && a.ol16)
      hipLaunchKernelGGL((lift_fwd_kernel<T, DH, VEC, P, sizeof(T) == 2>), dim3(blocks), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((lift_fwd_kernel<T, DH, VEC, P, false>), dim3(blocks), dim3(256), 0, st, a);
    return;
  }
  // query kernel: value, offsets/logits, refs, grad_out in; d(offsets/logits) out (+ records)
  const double q_bytes = nb.value + 2 * nb.offlog + nb.ref + nb.vis + nb.out;
  // the whole backward op (every kernel launched below, scoped or not): op-level timing for the
  // roofline, with the op's compulsory bytes (operands and results once; scratch is overhead)
  ProfScope op_scope(name("bev_lift_bwd_op"), st,
                     nb.value + (a.gvalue_lp != nullptr ? nb.value : nb.value_f32) + 2 * nb.offlog + nb.ref +
                         nb.vis + nb.out);
  if (bwd_mode == kAtomAll) {
    ProfScope ps(name("bev_lift_bwd_query+atomics"), st, q_bytes + nb.value_f32);
    if (sizeof(T) == 2 && a.ol16)
      hipLaunchKernelGGL((lift_bwd_query_kernel<T, DH, VEC, P, kAtomAll, sizeof(T) == 2>), dim3(blocks), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((lift_bwd_query_kernel<T, DH, VEC, P, kAtomAll, false>), dim3(blocks), dim3(256), 0, st, a);
    if (sizeof(T) == 2 && a.gvalue_lp != nullptr) {
      const long n = (long)a.B * a.Nc * a.fh * a.fw * a.H * DH;
      hipLaunchKernelGGL(narrow_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                         a.gvalue, (T*)a.gvalue_lp, n);
    }
  } else if (bwd_mode == kPlanGrid) {
    // query gradients (the kernel also zeroes the tile counters), sampling points binned by owner tile, the owner
    // tiles — each stores its finished pixels exactly once — and whatever overflowed a bucket added on top
    // (normally nothing: that kernel exits at once).  4 launches; 16-bit grad_value keeps the older order, below.
    const int tiles = t.tiles_x * t.tiles_y;
    // Diagnostic only (UBV_LIFT_TWO_STREAM=1): the grad_value chain (bins -> overflow -> owner tiles)
    // on a side stream next to the query-gradient kernel, forked and joined with events inside this
    // call.  Not faster (both chains are issue-bound); kept to reproduce and
    // test the concurrent residency of the two kernels (tests/test_lift_gpu.py).
    const char* two_env = getenv("UBV_LIFT_TWO_STREAM");
    const bool two = two_env != nullptr && atoi(two_env) != 0;
    hipStream_t s2 = st;
    static hipStream_t side_stream[64] = {};
    static hipEvent_t side_fork[64] = {}, side_join[64] = {};
    int dev = 0;
    if (two) {
      (void)hipGetDevice(&dev);
      dev &= 63;
      if (side_stream[dev] == nullptr) {
        (void)hipStreamCreateWithFlags(&side_stream[dev], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&side_fork[dev], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&side_join[dev], hipEventDisableTiming);
      }
      s2 = side_stream[dev];
      (void)hipEventRecord(side_fork[dev], st);
      (void)hipStreamWaitEvent(s2, side_fork[dev], 0);
    }
    // a.ovf_after: the query-gradient kernel runs FIRST and zeroes the counters (a.cnt_words) the bin kernel appends
    // through — except the LDS-window kernel (SCA-pts, f32), which keeps a memset and bins -> query: with the zeroing
    // loop in front of it that kernel measured 177 us instead of 157 (same registers, same occupancy, either launch
    // order: the scheduler's placement of the window fill changed), the gather kernel does not care.  Without
    // ovf_after the caller's memset zeroed the counters and the order is bins -> overflow -> query -> owner tiles.
    // TILE plan (bev_lift_tile.hip): the query-gradient kernel also bins its points — no lift_bin_kernel
    // (16-bit grad_value, !ovf_after: the TILE query kernel — which bins — then has to run before the overflow pass)
    constexpr int DT = sizeof(T) == 4 ? UBV_F32 : std::is_same<T, f16_t>::value ? UBV_F16 : UBV_BF16;
    const bool tile = tile_ok(a, DH, P, DT, true) && !two;
    const bool qfirst = (a.ovf_after && !win_ok<T, DH, P>(a) && !tile) || (tile && !a.ovf_after);
    if (a.ovf_after && !qfirst) (void)hipMemsetAsync(a.bin_cnt, 0, (size_t)a.cnt_words * sizeof(int), st);
    LiftArgs aq = a;
    if (!qfirst) aq.cnt_words = 0;
    auto run_query = [&]() {
      {
        ProfScope ps(name("bev_lift_bwd_query"), st, q_bytes);
        if (tile) tile_bwd_query_launch(aq, P, true, t.tiles_x, tiles, st, false, DH, DT);
        else if (win_ok<T, DH, P>(a)) {
          constexpr int HG = 128 / (DH * (int)sizeof(T));
          const int chunk = (int)(((long)a.total_tiles * (a.H / HG) + 7) / 8);
          const dim3 grid(8 * chunk);
          if (sizeof(T) == 2 && a.ol16) {
            if (wide) hipLaunchKernelGGL((lift_bwd_query_win_kernel<T, DH, VECS, P, sizeof(T) == 2, HG>), grid, dim3(256), kWinLds, st, aq, chunk);
            else hipLaunchKernelGGL((lift_bwd_query_win_kernel<T, DH, VEC, P, sizeof(T) == 2, HG>), grid, dim3(256), kWinLds, st, aq, chunk);
          } else {
            if (wide) hipLaunchKernelGGL((lift_bwd_query_win_kernel<T, DH, VECS, P, false, HG>), grid, dim3(256), kWinLds, st, aq, chunk);
            else hipLaunchKernelGGL((lift_bwd_query_win_kernel<T, DH, VEC, P, false, HG>), grid, dim3(256), kWinLds, st, aq, chunk);
          }
        } else
        if (sizeof(T) == 2 && a.ol16)
          {
            if (wide) launch_bwd_query_shared<T, DH, VECS, P, sizeof(T) == 2>(aq, st);
            else launch_bwd_query_shared<T, DH, VEC, P, sizeof(T) == 2>(aq, st);
          }
        else
          {
            if (wide) launch_bwd_query_shared<T, DH, VECS, P, false>(aq, st);
            else launch_bwd_query_shared<T, DH, VEC, P, false>(aq, st);
          }
      }
    };
    if (qfirst) run_query();
    if (!tile) {
      const long waves = (long)a.total_tiles * a.H;
      ProfScope ps(name("bev_lift_bwd_bins"), s2, nb.offlog + nb.ref + nb.rec);
      hipLaunchKernelGGL((lift_bin_kernel<T, DH, P, 0>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                         s2, a, t.tiles_x, tiles);
    }
    if (!a.ovf_after) {
      const long tw = (long)a.B * a.H * tiles;
      const long zb = (tw + 3) / 4;
      hipLaunchKernelGGL(lift_ovf_zero_kernel, dim3((unsigned)(zb < 256 ? zb : 256)), dim3(256), 0, s2, a,
                         t.tiles_x, tiles, DH);
      hipLaunchKernelGGL((lift_ovf_scatter_kernel<T, DH>), dim3(256), dim3(256), 0, s2, a, t.tiles_x,
                         tiles);
    }
    if (!qfirst) run_query();
    constexpr int RB = 2;
    const size_t lds = (size_t)t.waves * TileLds<T, DH, RB>::kWords * sizeof(uint16_t);
    {
      ProfScope ps(name("bev_lift_bwd_value_grid"), s2,
                   nb.rec + nb.out + (a.gvalue_lp != nullptr ? nb.value : nb.value_f32));
      bool done = false;
      if constexpr (sizeof(T) == 4 && DH == 32) {
        if (a.qrec && tile) {                      // the query kernel wrote query records (bev_lift_tile.hip, BINS = 2)
          const size_t lq = (size_t)t.waves * (64 * kQStride * 4 + 2 * 32 * GRow<32>::kStride * 2);
          hipLaunchKernelGGL((lift_bwd_value_q_kernel<P>), dim3(8 * t.chunk), dim3(64 * t.waves), lq, s2, a, t);
          done = true;
        }
      }
      if (!done)
        hipLaunchKernelGGL((lift_bwd_value_kernel<T, DH, P, RB>), dim3(8 * t.chunk), dim3(64 * t.waves),
                           lds, s2, a, t);
    }
    if (a.ovf_after)
      hipLaunchKernelGGL((lift_ovf_scatter_kernel<T, DH>), dim3(256), dim3(256), 0, s2, a, t.tiles_x, tiles);
    if (two) {
      (void)hipEventRecord(side_join[dev], s2);
      (void)hipStreamWaitEvent(st, side_join[dev], 0);
    }
  } else if (bwd_mode == kPlanMaps) {
    // exact CSR of the sampling points by owner tile (count, scan, fill), one wave per work item,
    // slabs of multi-item buckets summed in order (bev_lift_maps.inl); counters zeroed by the caller
    const int tiles = t.tiles_x * t.tiles_y;
    const int nb_ = a.B * a.Nc * a.H * tiles;
    const long waves = (long)a.total_tiles * a.H;
    {
      ProfScope ps(name("bev_lift_bwd_bins"), st, 2 * (nb.offlog + nb.ref) + nb.rec);
      hipLaunchKernelGGL((lift_bin_kernel<T, DH, P, 1>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a,
                         t.tiles_x, tiles);
      hipLaunchKernelGGL(maps_scan_kernel, dim3(1), dim3(1024), 0, st, a, nb_);
      hipLaunchKernelGGL((lift_bin_kernel<T, DH, P, 2>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a,
                         t.tiles_x, tiles);
    }
    {
      ProfScope ps(name("bev_lift_bwd_query"), st, q_bytes);
      if (sizeof(T) == 2 && a.ol16)
        {
          if (wide) launch_bwd_query_shared<T, DH, VECS, P, sizeof(T) == 2>(a, st);
          else launch_bwd_query_shared<T, DH, VEC, P, sizeof(T) == 2>(a, st);
        }
      else
        {
          if (wide) launch_bwd_query_shared<T, DH, VECS, P, false>(a, st);
          else launch_bwd_query_shared<T, DH, VEC, P, false>(a, st);
        }
    }
    constexpr int RB = 2;
    const size_t lds = (size_t)4 * TileLds<T, DH, RB>::kWords * sizeof(uint16_t);
    {
      ProfScope ps(name("bev_lift_bwd_value_maps"), st,
                   nb.rec + nb.out + (a.gvalue_lp != nullptr ? nb.value : nb.value_f32));
      hipLaunchKernelGGL((lift_bwd_value_items_kernel<T, DH, P, RB>), dim3((unsigned)((a.max_items + 3) / 4)),
                         dim3(256), lds, st, a, t);
      hipLaunchKernelGGL((maps_reduce_kernel<T, DH>), dim3((unsigned)((nb_ + 3) / 4)), dim3(256), 0, st, a,
                         t.tiles_x, tiles, nb_);
    }
  } else {
    if (!a.ext_list)
      hipLaunchKernelGGL(compact_visible_kernel, dim3(a.Nc), dim3(1024), 0, st, a.vis0, a.Nq, visible_tile_width(a.Nq, a.qw),
                         a.cam_list, a.cam_n);
    constexpr int RB = 6;
    const bool rb3 = sizeof(T) == 4 && t.tile_h * a.fw <= 96;      // f32, half-height bands: 3 row blocks
    const size_t lds = (size_t)t.waves * (rb3 ? CamLds<T, DH, 3>::kWords : CamLds<T, DH, RB>::kWords) * sizeof(uint16_t);
    {
      ProfScope ps(name("bev_lift_bwd_value_camera"), st,
                   nb.offlog + nb.ref + nb.vis + nb.out + nb.value_f32);
      bool done = false;
      if constexpr (DH == 32 && P == 8 && sizeof(T) == 2) {
        if (cam_mfma && t.tiles_y == 1) {          // whole padded map per wave, one MFMA round per batch
          const CamArgs c = cam_args(a, nullptr);
          const int mbt = (c.KB + 1) / 2;
          const size_t l2 = (size_t)t.waves * (mbt * 32 * kCamVStride + 64 * 32) * sizeof(uint16_t);
          if (mbt == 7)
            hipLaunchKernelGGL((lift_cam_bwd_value_kernel<T, 8, 7>), dim3(8 * t.chunk), dim3(64 * t.waves), l2, st, a, t, c);
          else
            hipLaunchKernelGGL((lift_cam_bwd_value_kernel<T, 8, 8>), dim3(8 * t.chunk), dim3(64 * t.waves), l2, st, a, t, c);
          done = true;
        }
      }
      if constexpr (DH == 32 && P == 8 && sizeof(T) == 4) {
        if (cam_mfma && t.tiles_y == 1) {          // f32 data: 32 queries per batch, packed hi | lo coefficients
          const CamArgs c = cam_args(a, nullptr);
          const int mbt = (c.KB + 1) / 2;
          const size_t l2 = (size_t)t.waves * (kCam32Win * kCam32VStride + 32 * 32) * sizeof(uint32_t);
          if (mbt == 7)
            hipLaunchKernelGGL((lift_cam32_bwd_value_kernel<8, 7>), dim3(8 * t.chunk), dim3(64 * t.waves), l2, st, a, t, c);
          else
            hipLaunchKernelGGL((lift_cam32_bwd_value_kernel<8, 8>), dim3(8 * t.chunk), dim3(64 * t.waves), l2, st, a, t, c);
          done = true;
        }
      }
      if (!done) {
        if (rb3)
          hipLaunchKernelGGL((lift_bwd_value_camera_kernel<T, DH, P, 3>), dim3(8 * t.chunk),
                             dim3(64 * t.waves), lds, st, a, t);
        else
          hipLaunchKernelGGL((lift_bwd_value_camera_kernel<T, DH, P, RB>), dim3(8 * t.chunk),
                             dim3(64 * t.waves), lds, st, a, t);
      }
    }
    {
      const long n = (long)a.B * a.Nc * a.H * a.fh * a.fw * DH;
      hipLaunchKernelGGL(slab_reduce_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a,
                         t.chunks, DH, t.balanced);
    }
    ProfScope ps(name("bev_lift_bwd_query"), st, q_bytes);
    if constexpr (DH == 32 && P == 8 && sizeof(T) == 2) {
      if (cam_mfma) {
        const CamArgs c = cam_args(a, nullptr);
        const size_t lds = (size_t)32 * kCamDStr * sizeof(float);
        if (a.ol16)
          hipLaunchKernelGGL((lift_cam_bwd_query_kernel<T, 8, true>), dim3(8 * c.chunk), dim3(64), lds, st, a, c);
        else
          hipLaunchKernelGGL((lift_cam_bwd_query_kernel<T, 8, false>), dim3(8 * c.chunk), dim3(64), lds, st, a, c);
        return;
      }
    }
    if constexpr (DH == 32 && P == 8 && sizeof(T) == 4) {
      if (cam_mfma && a.vnat_hi != nullptr) {
        const CamArgs c = cam_args(a, nullptr);
        const long n8 = (long)a.B * a.Nc * a.fh * a.fw * a.H * DH / 8;
        hipLaunchKernelGGL(value_split32_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st,
                           (const float*)a.value, (uint16_t*)a.vnat_hi, (uint16_t*)a.vnat_lo, n8);
        const size_t lds = (size_t)32 * kCamDStr * sizeof(float);
        hipLaunchKernelGGL((lift_cam32_bwd_query_kernel<8>), dim3(8 * c.chunk), dim3(64), lds, st, a, c);
        return;
      }
    }
    if (sizeof(T) == 2 && a.ol16)
      {
        if (wide) launch_bwd_query_shared<T, DH, VECS, P, sizeof(T) == 2>(a, st);
        else launch_bwd_query_shared<T, DH, VEC, P, sizeof(T) == 2>(a, st);
      }
    else
      {
        if (wide) launch_bwd_query_shared<T, DH, VECS, P, false>(a, st);
        else launch_bwd_query_shared<T, DH, VEC, P, false>(a, st);
      }
  }
}

template <typename T>
static bool lift_dispatch_T(const LiftArgs& a, const TileArgs& t, int Dh, int P, int bwd_mode,
                            bool cam_mfma, void* fwd_ws, hipStream_t st) {
  if (Dh == 32 && P == 8) { lift_launch<T, 32, 8>(a, t, bwd_mode, cam_mfma, fwd_ws, st); return true; }
  if (Dh == 32 && P == 4) { lift_launch<T, 32, 4>(a, t, bwd_mode, cam_mfma, fwd_ws, st); return true; }
  if (Dh == 16 && P == 8) { lift_launch<T, 16, 8>(a, t, bwd_mode, cam_mfma, fwd_ws, st); return true; }
  if (Dh == 16 && P == 4) { lift_launch<T, 16, 4>(a, t, bwd_mode, cam_mfma, fwd_ws, st); return true; }
  return false;
}

static bool lift_shape_ok(int H, int Dh, int P, int dtype) {
  if (dtype < 0 || dtype > 2) return false;
  if (!(Dh == 32 || Dh == 16) || !(P == 4 || P == 8)) return false;
  const int vec = (dtype == UBV_F32) ? 4 : 8;
  const int lq = H * (Dh / vec);
  return lq >= 4 && lq <= 64 && (64 % lq) == 0;
}


// Chooses who scatters grad_value (see lift_bwd_value_kernel) and lays out the owner tiles.
static int plan_backward(const LiftArgs& a, int Dh, int P, int dtype, int ref_is_grid, TileArgs& t) {
  t = TileArgs{};
  // GRID: the BEV-grid instances (self-attention, SCA-pts), and any single-map instance whose map
  // is too large for the CAMERA plan's row bands — the object-query decoder's cross-attention
  // (900 queries anywhere on the 200x200 fused BEV map): binning does not care how the queries are
  // laid out, clustered queries simply spill into the (exact) overflow list.
  const int cband = a.fw <= 192 ? 192 / a.fw : 0;            // rows per CAMERA band
  const bool cam_fits = cband >= 1 && (a.fh + cband - 1) / cband <= 8;
  if (a.Nc == 1 && ((ref_is_grid && a.qw > 0) || !cam_fits || ref_is_grid == 2)) {     // (2: the k1 operator asks for GRID)
    t.mode = 1;
    t.tile_w = t.tile_h = 8;                 // 64 pixels = 2 MFMA row blocks
    t.tiles_x = (a.fw + 7) / 8;
    t.tiles_y = (a.fh + 7) / 8;
    t.chunks = 1;
    t.chunk_q = 0;
    static const int grid_waves = getenv("UBV_GRID_WAVES") ? atoi(getenv("UBV_GRID_WAVES")) : 0;
    // f32 tiles hold 13 KB of LDS per wave: blocks of 2 or 3 waves pack 12 waves on a CU (the register limit), blocks of 4 only 8
    t.waves = grid_waves > 0 ? grid_waves : (dtype == UBV_F32 ? 3 : 4);   // (f32: 1 / 2 / 3 waves measured 112 / 114 / 109 us on the self-attention shape)
    // bucket capacity: twice the expected records per tile (P points per query-head, Nq/S queries
    // per pixel, 1.3 tiles per point); the overflow list takes whatever concentrates beyond that
    const double expect = 1.3 * P * 64.0 * (double)a.Nq / ((double)a.fh * a.fw);
    t.cap = (((int)(2.0 * expect) + 63) / 64) * 64 + 64;
  } else {
    // CAMERA: bands of full rows, at most 6 MFMA row blocks (192 pixels) per band, and few enough
    // bands that re-walking the visible queries once per band stays cheap
    int band = 192 / a.fw;
    if (band > a.fh) band = a.fh;
    const int bands = band >= 1 ? (a.fh + band - 1) / band : 0;
    static const int maps_env = getenv("UBV_LIFT_MAPS") ? atoi(getenv("UBV_LIFT_MAPS")) : 1;
    if ((bands != 1 && (maps_env || band < 1 || bands > 8)) || (maps_env == 2 && a.Nc > 1 && !cam_mfma_ok(a, Dh, P, dtype))) {
      // MAPS: maps of more than one band (bev_lift_maps.inl); UBV_LIFT_MAPS=0 keeps the band walk for A/B runs
      t.mode = 3;
      t.tile_w = t.tile_h = 8;
      t.tiles_x = (a.fw + 7) / 8;
      t.tiles_y = (a.fh + 7) / 8;
      t.chunks = 1;
      t.waves = 4;
      t.total = a.B * a.Nc * t.tiles_y * t.tiles_x * a.H;
      t.chunk = ((t.total + t.waves - 1) / t.waves + 7) / 8;
      return kPlanMaps;
    }
    t.mode = 2;
    t.tile_w = a.fw;
    t.tile_h = band;
    t.tiles_x = 1;
    t.tiles_y = bands;
    // f32 data (hi + lo operand tiles: 65 KB of LDS per wave at 6 row blocks, two waves per CU): a single-band map
    // is cut into two bands of 3 row blocks when it fits — half the LDS per wave, four waves per CU; each band
    // re-walks the visible list (UBV_CAM_F32_BANDS=1 keeps one band)
    static const int f32_bands = getenv("UBV_CAM_F32_BANDS") ? atoi(getenv("UBV_CAM_F32_BANDS")) : 2;
    const bool cam32 = dtype == UBV_F32 && bands == 1 && cam_mfma_ok(a, Dh, P, dtype);   // whole padded map per wave (bev_lift_cam32.inl)
    if (dtype == UBV_F32 && bands == 1 && f32_bands == 2 && !cam32) {        // (four quarter bands: 756 us)
      const int half = (a.fh + 1) / 2;
      if (half * a.fw <= 96) { t.tile_h = half; t.tiles_y = 2; }
    }
    // each (sample, camera, band, head) list is dealt evenly (device side, cam_chunk_len) to
    // `chunks` waves, enough of them to give every SIMD of the chip one wave: a wave keeps its
    // partial map in registers across all of its batches and writes ONE slab
    static const int split_env = getenv("UBV_CAM_SPLIT") ? atoi(getenv("UBV_CAM_SPLIT")) : 0;
    // (one block per CU: its LDS holds 4 waves' operand tiles, or 2 with f32 data's hi + lo tiles;
    // one block too many would cost a whole second round, so round down)
    t.waves = (dtype == UBV_F32 && t.tile_h * a.fw > 96 && !cam32) ? 2 : 4;
    const int combos = a.B * a.Nc * t.tiles_y * a.H;
    int split = split_env > 0 ? split_env : (256 * t.waves) / combos;
    if (cam32) {
      // f32 matrix-core value gradient (bev_lift_cam32.inl): 22.5 KB of LDS per wave -> one-wave blocks, 7 resident per
      // CU, and list shares sized for exactly that one round.  Measured at 6 x 8x22, bs = 2 (value-gradient kernel, us)
      // for 4 / 5 / 6 / 7 / 8 / 10 / 14 waves per CU: 218 / 183 / 154 / 141 / 186 / 175 / 161; four-wave blocks: 210.
      static const int w32 = getenv("UBV_CAM32_WAVES") ? atoi(getenv("UBV_CAM32_WAVES")) : 1;
      static const int per_cu = getenv("UBV_CAM32_PER_CU") ? atoi(getenv("UBV_CAM32_PER_CU")) : 7;
      t.waves = w32 > 0 ? w32 : 1;
      if (split_env <= 0) split = (256 * per_cu) / combos;
    }
    split = max(1, min(split, (a.Nq + 63) / 64));
    t.chunk_q = 0;
    t.chunks = split;
    static const int bal_env = getenv("UBV_CAM_BALANCE") ? atoi(getenv("UBV_CAM_BALANCE")) : 1;
    t.balanced = (bal_env != 0 && t.tiles_y == 1 && cam_mfma_ok(a, Dh, P, dtype)) ? 1 : 0;
  }
  t.total = a.B * a.Nc * t.tiles_y * t.tiles_x * t.chunks * a.H;
  t.chunk = ((t.total + t.waves - 1) / t.waves + 7) / 8;       // blocks per XCD
  return t.mode == 1 ? kPlanGrid : kAtomNone;
}

static size_t lift_list_bytes(const LiftArgs& a) {
  return (((size_t)a.Nc * a.Nq + a.Nc + 4) * sizeof(int) + 255) & ~(size_t)255;
}
// GRID workspace: [tile counters + overflow counter][buckets][overflow records][overflow tiles];
// the overflow list is sized for the worst case (every point overflowing in all of its <= 4 tiles).
struct GridWs { size_t cnt_bytes, bins_off, ovf_rec_off, ovf_tile_off, qpts_off, total; long ovf_cap; };
static GridWs grid_ws(const LiftArgs& a, const TileArgs& t, int P) {
  GridWs w;
  const size_t tiles = (size_t)a.B * a.H * t.tiles_x * t.tiles_y;
  w.cnt_bytes = ((tiles + 1) * sizeof(int) + 255) & ~(size_t)255;
  w.bins_off = w.cnt_bytes;
  w.ovf_cap = 4L * a.B * a.Nq * a.H * P;
  w.ovf_rec_off = w.bins_off + tiles * t.cap * sizeof(float4);
  w.ovf_tile_off = w.ovf_rec_off + (size_t)w.ovf_cap * sizeof(float4);
  // query records (f32 TILE instances): every (query, head)'s points, 12 bytes each
  w.qpts_off = w.ovf_tile_off + (((size_t)w.ovf_cap * sizeof(int) + 255) & ~(size_t)255);
  w.total = w.qpts_off + (((size_t)a.B * a.Nq * a.H * P * 3 * sizeof(float) + 255) & ~(size_t)255);
  return w;
}

// query records instead of point records: f32 data and f32 grad_value on the TILE plan, module form.  OFF by default
// (UBV_LIFT_QREC=1): measured slower — profiles/r06_qrec_experiment.txt
static bool qrec_ok(const LiftArgs& a, int Dh, int P, int dtype) {
  static const int env = getenv("UBV_LIFT_QREC") ? atoi(getenv("UBV_LIFT_QREC")) : 0;
  return env != 0 && dtype == UBV_F32 && Dh == 32 && a.gvalue_lp == nullptr && a.ovf_after && tile_ok(a, Dh, P, dtype, true);
}
// MAPS workspace: [counts | cursors | n_items] (zeroed per call) [starts][first items][item buckets]
// [records][slabs].  Records and items are sized for the worst case — every query visible in every
// camera, every point in 4 tiles — which the host cannot narrow without reading the visibility back
// (2 GB of the 288 at the reference's 6 x 25x45 shape and bs = 2; only what is used is touched).
struct MapsWs { size_t zero_bytes, start_off, first_off, ibucket_off, bins_off, slab_off, total; long buckets, max_rec, max_items; };
static MapsWs maps_ws(const LiftArgs& a, const TileArgs& t, int Dh, int P) {
  MapsWs w;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  w.buckets = (long)a.B * a.Nc * a.H * t.tiles_x * t.tiles_y;
  w.max_rec = 4L * a.B * a.Nc * a.Nq * a.H * P;
  w.max_items = w.buckets + w.max_rec / kItemRecs;
  w.zero_bytes = al((2 * (size_t)w.buckets + 1) * sizeof(int));
  w.start_off = w.zero_bytes;
  w.first_off = w.start_off + al(((size_t)w.buckets + 1) * sizeof(int));
  w.ibucket_off = w.first_off + al(((size_t)w.buckets + 1) * sizeof(int));
  w.bins_off = w.ibucket_off + al((size_t)w.max_items * sizeof(int));
  w.slab_off = w.bins_off + al((size_t)w.max_rec * sizeof(float4));
  w.total = w.slab_off + al((size_t)w.max_items * 64 * Dh * sizeof(float));
  return w;
}
static size_t cam_slab_bytes(const LiftArgs& a, const TileArgs& t, int Dh) {
  return (((size_t)a.B * a.Nc * a.H * t.chunks * a.fh * a.fw * Dh * sizeof(float)) + 255) & ~(size_t)255;
}
static size_t lift_ws_bytes(int mode, const LiftArgs& a, const TileArgs& t, int Dh, int P, int dtype) {
  if (mode == kPlanGrid) return grid_ws(a, t, P).total;
  if (mode == kPlanMaps) return maps_ws(a, t, Dh, P).total;
  if (mode == kAtomNone)     // visible-query lists + one partial map per (b, cam, head, chunk) (+ f32 matrix-core plan: hi / lo copies of value)
    return lift_list_bytes(a) + cam_slab_bytes(a, t, Dh) +
           ((dtype == UBV_F32 && cam_mfma_ok(a, Dh, P, dtype)) ? 2 * cam_vnat_bytes(a, Dh) : 0);
  return 0;
}

static int lift_run(LiftArgs a, int Dh, int P, int dtype, bool bwd, int ref_is_grid, void* ws,
                    int64_t ws_bytes, void* stream) {
  UBV_CHECK_ARG(a.B > 0 && a.Nc > 0 && a.fh > 0 && a.fw > 0 && a.H > 0 && a.Nq > 0 && a.Z > 0,
                "bev_lift: non-positive dimension");
  UBV_CHECK_ARG(P % a.Z == 0, "bev_lift: num_points %d not a multiple of Z %d", P, a.Z);
  UBV_CHECK_ARG((long)a.fh * a.fw * a.H * Dh < (1L << 30) && (long)a.Nq * a.H * Dh < (1L << 30),
                "bev_lift: one value map / one sample's query rows must hold fewer than 2^30 elements (32-bit byte offsets)");
  if (!lift_shape_ok(a.H, Dh, P, dtype)) {
    set_error("bev_lift: no kernel for H=%d Dh=%d P=%d dtype=%d", a.H, Dh, P, dtype);
    return UBV_ERR_UNSUPPORTED;
  }
  UBV_CHECK_ARG(!a.ol16 || dtype != UBV_F32, "bev_lift: offlog_dtype must be f32 or equal dtype");
  const int ol_align = a.ol16 ? 8 : 4;            // elements per 16 bytes
  UBV_CHECK_ARG((a.off_stride % ol_align) == 0 && (a.log_stride % ol_align) == 0 &&
                    ((uintptr_t)a.offsets % 16) == 0 && ((uintptr_t)a.logits % 16) == 0,
                "bev_lift: offsets/logits rows must be 16-byte aligned");
  UBV_CHECK_ARG(a.gvalue_lp == nullptr || dtype != UBV_F32,
                "bev_lift_backward: grad_value_lowp needs a 16-bit dtype");
  UBV_CHECK_ARG(((uintptr_t)a.ref % 8) == 0, "bev_lift: ref must be 8-byte aligned");
  if (bwd)
    UBV_CHECK_ARG((a.goff_stride % ol_align) == 0 && (a.glog_stride % ol_align) == 0 &&
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
  auto magic = [](long nmax, int d) -> unsigned {
    if (d <= 1 || nmax * (long)d >= (1L << 32)) return 0u;      // d == 1: plain division folds away
    return (unsigned)((1UL << 32) / (unsigned long)d) + 1u;
  };
  a.mg_tps = magic(a.total_tiles, a.tiles_per_sample);
  a.mg_tx = a.tiles_x > 0 ? magic(a.tiles_per_sample, a.tiles_x) : 0u;
  hipStream_t st = as_stream(stream);
  TileArgs t{};
  int mode = -1;
  if (bwd) {
    mode = plan_backward(a, Dh, P, dtype, ref_is_grid, t);
    const size_t need = lift_ws_bytes(mode, a, t, Dh, P, dtype);
    UBV_CHECK_ARG(need == 0 || (ws != nullptr && ws_bytes >= (int64_t)need),
                  "bev_lift_backward: workspace of %lld bytes needed, got %lld", (long long)need,
                  (long long)ws_bytes);
    if (mode == kPlanGrid) {
      const GridWs w = grid_ws(a, t, P);
      a.bin_cnt = (int*)ws;
      a.ovf_n = a.bin_cnt + (size_t)a.B * a.H * t.tiles_x * t.tiles_y;
      a.bins = (float4*)((char*)ws + w.bins_off);
      a.ovf_rec = (float4*)((char*)ws + w.ovf_rec_off);
      a.ovf_tile = (int*)((char*)ws + w.ovf_tile_off);
      a.cap = t.cap;
      a.ovf_cap = (int)min(w.ovf_cap, (long)INT_MAX);
      static const bool two_env = getenv("UBV_LIFT_TWO_STREAM") != nullptr && atoi(getenv("UBV_LIFT_TWO_STREAM")) != 0;
      a.ovf_after = (a.gvalue_lp == nullptr && !two_env) ? 1 : 0;
      if (a.ovf_after) a.cnt_words = (int)(w.cnt_bytes / sizeof(int));      // zeroed by the query-gradient kernel
      else if (hipMemsetAsync(ws, 0, w.cnt_bytes, st) != hipSuccess) {
        set_error("bev_lift_backward: memset failed");
        return UBV_ERR_LAUNCH;
      }
      a.qpts = (float*)((char*)ws + w.qpts_off);
      a.qrec = qrec_ok(a, Dh, P, dtype) ? 1 : 0;
    }
    if (mode == kPlanMaps) {
      const MapsWs w = maps_ws(a, t, Dh, P);
      UBV_CHECK_ARG(w.max_rec < (1L << 31) && w.max_items < (1L << 31), "bev_lift_backward: too many sampling points for the MAPS plan");
      a.bin_cnt = (int*)ws;
      a.bin_cur = a.bin_cnt + w.buckets;
      a.n_items = a.bin_cur + w.buckets;
      a.bin_start = (int*)((char*)ws + w.start_off);
      a.item_first = (int*)((char*)ws + w.first_off);
      a.item_bucket = (int*)((char*)ws + w.ibucket_off);
      a.bins = (float4*)((char*)ws + w.bins_off);
      a.slab = (float*)((char*)ws + w.slab_off);
      a.max_items = (int)w.max_items;
      if (hipMemsetAsync(ws, 0, w.zero_bytes, st) != hipSuccess) {
        set_error("bev_lift_backward: memset failed");
        return UBV_ERR_LAUNCH;
      }
    }
    if (mode == kAtomNone) {
      if (!a.ext_list) {
        a.cam_list = (int*)ws;
        a.cam_n = (int*)ws + (size_t)a.Nc * a.Nq;
      }
      a.slab = (float*)((char*)ws + lift_list_bytes(a));
      if (dtype == UBV_F32 && cam_mfma_ok(a, Dh, P, dtype)) {
        a.vnat_hi = (char*)ws + lift_list_bytes(a) + cam_slab_bytes(a, t, Dh);
        a.vnat_lo = (const char*)a.vnat_hi + cam_vnat_bytes(a, Dh);
      }
    }
    if (mode == kAtomAll) {    // the owner-tile plans write every element of grad_value exactly once
      const size_t bytes = (size_t)a.B * a.Nc * a.fh * a.fw * a.H * Dh * sizeof(float);
      if (hipMemsetAsync(a.gvalue, 0, bytes, st) != hipSuccess) {
        set_error("bev_lift_backward: memset failed");
        return UBV_ERR_LAUNCH;
      }
    }
  }
  // small per-camera maps: gather / dot products on the matrix cores (bev_lift_cam.inl)
  bool cam_mfma = cam_mfma_ok(a, Dh, P, dtype, !bwd) && (!bwd || mode == kAtomNone);
  void* fwd_ws = nullptr;
  if (cam_mfma && !bwd) {
    if (ws != nullptr && ws_bytes >= (int64_t)cam_vfrag_bytes(a, dtype)) fwd_ws = ws;
    else cam_mfma = false;                       // no scratch given: the gather kernel needs none
  }
  bool ok = false;
  switch (dtype) {
    case UBV_F32: ok = lift_dispatch_T<float>(a, t, Dh, P, mode, cam_mfma, fwd_ws, st); break;
    case UBV_F16: ok = lift_dispatch_T<f16_t>(a, t, Dh, P, mode, cam_mfma, fwd_ws, st); break;
    case UBV_BF16: ok = lift_dispatch_T<bf16_t>(a, t, Dh, P, mode, cam_mfma, fwd_ws, st); break;
  }
  if (!ok) { set_error("bev_lift: dispatch failed"); return UBV_ERR_UNSUPPORTED; }
  UBV_CHECK_LAUNCH(bwd ? "bev_lift_backward" : "bev_lift_forward");
  return UBV_OK;
}

// ---- grad_value of the k1 OPERATOR (msda_k1.hip) on the GRID owner-tile plan: explicit locations / weights are binned
// by owner tile (lift_bin_kernel<..., K1>) and the owner tiles store every pixel once — no f32 atomics on grad_value.
// One level, the shapes bev_lift covers; anything else keeps the atomic kernel.
bool k1_grid_ok(int H, int Dh, int P, int dtype, int fh, int fw) {
  return lift_shape_ok(H, Dh, P, dtype) && fh >= 1 && fw >= 1;
}
static void k1_grid_args(LiftArgs& a, TileArgs& t, int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype) {
  a = LiftArgs{};
  a.B = B; a.Nc = 1; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = 1;
  a.tiles_per_sample = (Nq + 63) / 64;
  a.total_tiles = B * a.tiles_per_sample;
  a.chunk = (a.total_tiles + 7) / 8;
  (void)plan_backward(a, Dh, P, dtype, 2, t);
}
int64_t k1_grid_workspace(int B, int fh, int fw, int H, int Dh, int Nq, int P, int dtype) {
  LiftArgs a; TileArgs t;
  k1_grid_args(a, t, B, fh, fw, H, Dh, Nq, P, dtype);
  return (int64_t)grid_ws(a, t, P).total;
}
template <typename T, int DH, int P>
static void k1_grid_launch(const LiftArgs& a, const TileArgs& t, hipStream_t st) {
  const int tiles = t.tiles_x * t.tiles_y;
  const long waves = (long)a.total_tiles * a.H;
  hipLaunchKernelGGL((lift_bin_kernel<T, DH, P, 0, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a,
                     t.tiles_x, tiles);
  const long tw = (long)a.B * a.H * tiles, zb = (tw + 3) / 4;
  hipLaunchKernelGGL(lift_ovf_zero_kernel, dim3((unsigned)(zb < 256 ? zb : 256)), dim3(256), 0, st, a, t.tiles_x, tiles, DH);
  hipLaunchKernelGGL((lift_ovf_scatter_kernel<T, DH>), dim3(256), dim3(256), 0, st, a, t.tiles_x, tiles);
  constexpr int RB = 2;
  const size_t lds = (size_t)t.waves * TileLds<T, DH, RB>::kWords * sizeof(uint16_t);
  hipLaunchKernelGGL((lift_bwd_value_kernel<T, DH, P, RB>), dim3(8 * t.chunk), dim3(64 * t.waves), lds, st, a, t);
}
int k1_grid_value(const float* loc, const float* aw, const void* gout, float* gvalue, int B, int fh, int fw, int H,
                  int Dh, int Nq, int P, int dtype, void* ws, int64_t ws_bytes, hipStream_t st) {
  LiftArgs a; TileArgs t;
  k1_grid_args(a, t, B, fh, fw, H, Dh, Nq, P, dtype);
  const GridWs w = grid_ws(a, t, P);
  if (ws == nullptr || ws_bytes < (int64_t)w.total) {
    set_error("ms_deform_attn_backward_planned: workspace of %lld bytes needed, got %lld", (long long)w.total, (long long)ws_bytes);
    return UBV_ERR_INVALID;
  }
  a.offsets = loc; a.off_stride = (long)H * P * 2; a.logits = aw; a.log_stride = (long)H * P;
  a.gout = gout; a.gvalue = gvalue;
  a.bin_cnt = (int*)ws;
  a.ovf_n = a.bin_cnt + (size_t)B * H * t.tiles_x * t.tiles_y;
  a.bins = (float4*)((char*)ws + w.bins_off);
  a.ovf_rec = (float4*)((char*)ws + w.ovf_rec_off);
  a.ovf_tile = (int*)((char*)ws + w.ovf_tile_off);
  a.cap = t.cap;
  a.ovf_cap = (int)min(w.ovf_cap, (long)INT_MAX);
  if (hipMemsetAsync(ws, 0, w.cnt_bytes, st) != hipSuccess) { set_error("ms_deform_attn_backward_planned: memset failed"); return UBV_ERR_LAUNCH; }
#define UBV_K1G(TT) do { \
    if (Dh == 32 && P == 8) k1_grid_launch<TT, 32, 8>(a, t, st); else if (Dh == 32 && P == 4) k1_grid_launch<TT, 32, 4>(a, t, st); \
    else if (Dh == 16 && P == 8) k1_grid_launch<TT, 16, 8>(a, t, st); else k1_grid_launch<TT, 16, 4>(a, t, st); } while (0)
  if (dtype == UBV_F32) UBV_K1G(float); else if (dtype == UBV_F16) UBV_K1G(f16_t); else UBV_K1G(bf16_t);
#undef UBV_K1G
  return UBV_OK;
}

// ---- the k1 OPERATOR on the TILE plan: explicit sampling locations / attention weights, queries on a
// qh x qw grid (the caller says so: the mmcv operator's signature has no such notion).  Forward = lift_tile_fwd_kernel<P, K1>;
// backward = lift_tile_bwd_query_kernel<P, bins, K1> (d locations, d weights, records) + the owner tiles.
static void k1_tile_args(LiftArgs& a, int B, int fh, int fw, int H, int Nq, int qh, int qw) {
  a = LiftArgs{};
  a.B = B; a.Nc = 1; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = 1; a.qw = qw; a.qh = qh;
  a.tiles_x = (qw + 7) / 8;
  a.tiles_per_sample = a.tiles_x * ((qh + 7) / 8);
  a.total_tiles = B * a.tiles_per_sample;
  a.chunk = (a.total_tiles + 7) / 8;
  // (mg_tps / mg_tx = 0: plain divisions in the tile decode)
}
bool k1_tile_ok(int H, int Dh, int P, int dtype, int fh, int fw, int Nq, int qh, int qw) {
  if (qh <= 0 || qw <= 0 || (long)qh * qw != Nq || fh < 1 || fw < 1) return false;
  LiftArgs a;
  k1_tile_args(a, 1, fh, fw, H, Nq, qh, qw);
  return Dh == 32 && dtype == UBV_F32 && tile_ok(a, Dh, P, dtype) && (long)fh * fw * H * Dh < (1L << 30) && (long)Nq * H * Dh < (1L << 30);
}
int k1_tile_forward(const void* value, const float* loc, const float* aw, void* out, int B, int fh, int fw, int H, int Nq,
                    int P, int qh, int qw, hipStream_t st) {
  LiftArgs a;
  k1_tile_args(a, B, fh, fw, H, Nq, qh, qw);
  a.value = value; a.offsets = loc; a.off_stride = (long)H * P * 2; a.logits = aw; a.log_stride = (long)H * P; a.out = out;
  ProfScope ps("k1_tile_fwd", st, 0.0);
  tile_fwd_launch(a, P, st, true);
  return UBV_OK;
}
int64_t k1_tile_workspace(int B, int fh, int fw, int H, int Nq, int P, int qh, int qw) {
  LiftArgs a; TileArgs t{};
  k1_tile_args(a, B, fh, fw, H, Nq, qh, qw);
  (void)plan_backward(a, 32, P, UBV_F32, 2, t);
  return (int64_t)grid_ws(a, t, P).total;
}
int k1_tile_backward(const void* value, const float* loc, const float* aw, const void* gout, float* gvalue, float* gloc,
                     float* gaw, int B, int fh, int fw, int H, int Nq, int P, int qh, int qw, void* ws, int64_t ws_bytes,
                     hipStream_t st) {
  LiftArgs a; TileArgs t{};
  k1_tile_args(a, B, fh, fw, H, Nq, qh, qw);
  if (plan_backward(a, 32, P, UBV_F32, 2, t) != kPlanGrid) { set_error("ms_deform_attn_backward_grid: no owner-tile plan for this shape"); return UBV_ERR_UNSUPPORTED; }
  const GridWs w = grid_ws(a, t, P);
  if (ws == nullptr || ws_bytes < (int64_t)w.total) {
    set_error("ms_deform_attn_backward_grid: workspace of %lld bytes needed, got %lld", (long long)w.total, (long long)ws_bytes);
    return UBV_ERR_INVALID;
  }
  a.value = value; a.offsets = loc; a.off_stride = (long)H * P * 2; a.logits = aw; a.log_stride = (long)H * P;
  a.gout = gout; a.gvalue = gvalue; a.goff = gloc; a.goff_stride = (long)H * P * 2; a.glog = gaw; a.glog_stride = (long)H * P;
  a.bin_cnt = (int*)ws;
  a.ovf_n = a.bin_cnt + (size_t)B * H * t.tiles_x * t.tiles_y;
  a.bins = (float4*)((char*)ws + w.bins_off);
  a.ovf_rec = (float4*)((char*)ws + w.ovf_rec_off);
  a.ovf_tile = (int*)((char*)ws + w.ovf_tile_off);
  a.cap = t.cap;
  a.ovf_cap = (int)min(w.ovf_cap, (long)INT_MAX);
  a.ovf_after = 1;                                        // owner tiles store plainly, the overflow list is added afterwards
  if (hipMemsetAsync(ws, 0, w.cnt_bytes, st) != hipSuccess) { set_error("ms_deform_attn_backward_grid: memset failed"); return UBV_ERR_LAUNCH; }
  const int tiles = t.tiles_x * t.tiles_y;
  {
    ProfScope ps("k1_tile_bwd_query", st, 0.0);
    tile_bwd_query_launch(a, P, true, t.tiles_x, tiles, st, true);
  }
  constexpr int RB = 2;
  const size_t lds = (size_t)t.waves * TileLds<float, 32, RB>::kWords * sizeof(uint16_t);
  {
    ProfScope ps("k1_tile_bwd_value", st, 0.0);
    if (P == 4) hipLaunchKernelGGL((lift_bwd_value_kernel<float, 32, 4, RB>), dim3(8 * t.chunk), dim3(64 * t.waves), lds, st, a, t);
    else hipLaunchKernelGGL((lift_bwd_value_kernel<float, 32, 8, RB>), dim3(8 * t.chunk), dim3(64 * t.waves), lds, st, a, t);
  }
  hipLaunchKernelGGL((lift_ovf_scatter_kernel<float, 32>), dim3(256), dim3(256), 0, st, a, t.tiles_x, tiles);
  return UBV_OK;
}

}  // namespace ubv

extern "C" int64_t ubv_bev_lift_backward_workspace(int B, int Nc, int fh, int fw, int H, int Dh,
                                                   int Nq, int P, int qgrid_w, int qgrid_h,
                                                   int ref_is_grid) {
  ubv::LiftArgs a{};
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq;
  a.qw = (qgrid_w > 0 && (long)qgrid_w * qgrid_h == Nq) ? qgrid_w : 0;
  a.qh = a.qw ? qgrid_h : 0;
  // the plan can depend on the dtype (the matrix-core CAMERA plan is 16-bit only): the larger of the two
  size_t need = 0;
  for (int dt : {UBV_BF16, UBV_F32}) {
    ubv::TileArgs t{};
    const int mode = ubv::plan_backward(a, Dh, P, dt, ref_is_grid, t);
    const size_t b = ubv::lift_ws_bytes(mode, a, t, Dh, P, dt);
    need = b > need ? b : need;
  }
  return (int64_t)need;
}

extern "C" int64_t ubv_visible_lists_elems(int Nc, int Nq) { return (int64_t)Nc * Nq + Nc; }

extern "C" int ubv_compact_visible(const uint8_t* vis0, int Nc, int Nq, int32_t* lists, void* stream) {
  return ubv_compact_visible_grid(vis0, Nc, Nq, 0, lists, stream);
}

extern "C" int ubv_compact_visible_grid(const uint8_t* vis0, int Nc, int Nq, int qw, int32_t* lists, void* stream) {
  UBV_CHECK_ARG(lists != nullptr && Nc > 0 && Nq > 0 && qw >= 0, "compact_visible: bad arguments");
  hipLaunchKernelGGL(ubv::compact_visible_kernel, dim3(Nc), dim3(1024), 0, ubv::as_stream(stream), vis0, Nq,
                     ubv::visible_tile_width(Nq, qw), lists, lists + (size_t)Nc * Nq);
  UBV_CHECK_LAUNCH("compact_visible");
  return UBV_OK;
}

extern "C" int64_t ubv_bev_lift_forward_workspace(int B, int Nc, int fh, int fw, int H, int Dh, int P,
                                                  int dtype) {
  ubv::LiftArgs a{};
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H;
  return ubv::cam_mfma_ok(a, Dh, P, dtype, true) ? (int64_t)ubv::cam_vfrag_bytes(a, dtype) : 0;
}

extern "C" int ubv_bev_lift_supported(int H, int Dh, int P, int dtype) {
  return ubv::lift_shape_ok(H, Dh, P, dtype) ? 1 : 0;
}

extern "C" int ubv_bev_lift_forward(const void* value, const void* offsets, int64_t off_stride,
                                    const void* logits, int64_t log_stride, int offlog_dtype,
                                    const float* ref,
                                    const uint8_t* vis0, const float* count, void* out, int B,
                                    int Nc, int fh, int fw, int H, int Dh, int Nq, int P, int Z,
                                    int qgrid_w, int qgrid_h, int dtype, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  UBV_CHECK_ARG(value && offsets && logits && ref && out, "bev_lift_forward: null pointer");
  UBV_CHECK_ARG(offlog_dtype == UBV_F32 || offlog_dtype == dtype,
                "bev_lift: offlog_dtype %d must be f32 or equal dtype %d", offlog_dtype, dtype);
  ubv::LiftArgs a{};
  a.ol16 = (offlog_dtype != UBV_F32) ? 1 : 0;
  a.value = value; a.offsets = offsets; a.off_stride = off_stride; a.logits = logits;
  a.log_stride = log_stride; a.ref = ref; a.vis0 = vis0; a.count = count; a.out = out;
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = Z; a.qw = qgrid_w;
  a.qh = qgrid_h;
  return ubv::lift_run(a, Dh, P, dtype, false, 0, workspace, workspace_bytes, stream);
}

extern "C" int ubv_bev_lift_backward(const void* value, const void* offsets, int64_t off_stride,
                                     const void* logits, int64_t log_stride, int offlog_dtype,
                                     const float* ref, const uint8_t* vis0, const float* count,
                                     const float* slot_center, const void* grad_out,
                                     float* grad_value, void* grad_value_lowp, void* grad_offsets,
                                     int64_t goff_stride, void* grad_logits, int64_t glog_stride,
                                     int B, int Nc, int fh,
                                     int fw, int H, int Dh, int Nq, int P, int Z, int qgrid_w,
                                     int qgrid_h, int ref_is_grid, int dtype,
                                     const int32_t* visible_lists, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  UBV_CHECK_ARG(value && offsets && logits && ref && grad_out && grad_value && grad_offsets &&
                    grad_logits, "bev_lift_backward: null pointer");
  UBV_CHECK_ARG(offlog_dtype == UBV_F32 || offlog_dtype == dtype,
                "bev_lift: offlog_dtype %d must be f32 or equal dtype %d", offlog_dtype, dtype);
  ubv::LiftArgs a{};
  a.ol16 = (offlog_dtype != UBV_F32) ? 1 : 0;
  a.value = value; a.offsets = offsets; a.off_stride = off_stride; a.logits = logits;
  a.log_stride = log_stride; a.ref = ref; a.vis0 = vis0; a.count = count; a.gout = grad_out;
  (void)slot_center;          // accepted for ABI stability; the bins plan does not need the hint
  a.gvalue = grad_value; a.gvalue_lp = grad_value_lowp; a.goff = grad_offsets; a.goff_stride = goff_stride; a.glog = grad_logits;
  a.glog_stride = glog_stride;
  a.B = B; a.Nc = Nc; a.fh = fh; a.fw = fw; a.H = H; a.Nq = Nq; a.Z = Z; a.qw = qgrid_w;
  a.qh = qgrid_h;
  if (visible_lists != nullptr) {
    a.ext_list = 1;
    a.cam_list = const_cast<int*>(visible_lists);
    a.cam_n = a.cam_list + (size_t)Nc * Nq;
  }
  return ubv::lift_run(a, Dh, P, dtype, true, ref_is_grid, workspace, workspace_bytes, stream);
}
