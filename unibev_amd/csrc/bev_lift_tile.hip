// TILE plan: f32 GRID lifting with ONE LANE PER SAMPLING POINT (its own translation unit; the argument blocks and
// the query / tile decoding come from bev_lift_core.h).
//
// The shared-footprint and LDS-window kernels put the Dh channels of a (query, head) on LP = 8 adjacent lanes: every
// corner is one coalesced 128-byte gather, but each of its 4 FMAs per lane is paid with a broadcast of the corner's
// index and coefficient and (backward) a 3-step lane reduction of the dot product — ~1.5 wave instructions per
// (query, head, corner), and PMC shows these kernels bound by VALU issue (a wave64 VALU instruction holds its SIMD for a
// quad-cycle).  Once the corners come from LDS there is nothing left to coalesce, so here
//   block  = (8x8 query tile, one head), 4 waves of 16 queries;
//   lane   = (query, slot pp of 4): it owns the points pp, pp + 4 of its query — loads only their offsets / logit,
//            builds only their footprints; softmax over the quad with DPP;
//   window = the pixel box of the block's footprints (at most 16x16) of the head's value slice, copied into LDS with
//            whole 128-byte lines; only the rows / columns of the box are fetched (a 16x16 window per 8x8 tile would
//            move 4x the map);
//   corner = 8 ds_read_b128 + 32 v_fma_f32 per lane (no packed f32 instructions: Makefile), all arithmetic plain f32 (no operand splitting), no cross-lane
//            traffic; a lane whose corner falls outside the window fetches that row from global memory itself (the
//            window is a cache, never an approximation);
//   output = forward: the 4 lanes of a query add their rows with DPP quad permutes and store 32 bytes each;
//            backward: every lane has its own points' 4 dot products — d(offset), d(logit) need one quad sum.
// The backward kernel also BINS its points by owner tile (the records lift_bwd_value_kernel reads): the separate
// lift_bin_kernel recomputed every softmax and footprint for that.
// Measured at bs = 2 on the 200x200 (P = 4) / 180x180 (P = 8) maps, us: forward 81 -> 71 / 115 -> 92; query gradient
// (+ bins) 108 + 37 -> 78 (+ bins) / 157 + 65 -> 112 (+ bins).  profiles/r04_tile_*.txt.

#include <string.h>

#include <type_traits>

#include "bev_lift_core.h"

namespace ubv {

// LDS window of the TILE kernels: pixel (dx, dy) at dy * kTWinRow + dx * kWinRowB.  144-byte pixels put the 16 columns
// of one row on 16 distinct bank quads (9 dx mod 16); the 64 bytes of padding per window row add 4 dy, so that points
// stepping along either axis or a diagonal — the lanes of a ds_read_b128 group are 4 queries x 4 points — stay on
// distinct quads: quad = (9 dx + 4 dy) mod 16.
constexpr int kTWinRow = kWin * kWinRowB + 64;
constexpr int kTWinLds = kWin * kTWinRow;

// (tile, 128-byte line of heads) of this block with H = 8 heads (tile_ok): HPB heads share a line (Dh = 32: one head,
// Dh = 16: two); line fastest, XCD x owns a contiguous range of units
template <int HPB>
__device__ __forceinline__ bool tile_decode8(const LiftArgs& a, int chunk, WinGeom& g) {
  constexpr int UPT = 8 / HPB;                            // units per tile
  const int v = xcd_remap(blockIdx.x, chunk);
  const int item = v / UPT;
  if (item >= a.total_tiles) return false;
  g.hg = v % UPT;
  g.b = div_mg(item, a.tiles_per_sample, a.mg_tps);
  g.tile = item;
  return true;
}

// min over the wave: DPP inside the rows of 16 lanes (quad permutes, row_half_mirror, row_mirror), the four rows
// through scalar reads
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

// The sampling points of this lane — lane = (query, slot pp of 4), points pp + 4 j — as pixel coordinates and softmax
// weights (0 for a query outside the grid), and the pixel box of the block's corners: those with a non-zero bilinear
// weight (forward), every corner inside the map (BWD: a corner of weight 0 still has a derivative).  One barrier.
// K1: the mmcv operator's inputs — a.offsets holds the sampling LOCATIONS [B, Nq, H, P, 2] (normalised), a.logits the
// attention WEIGHTS [B, Nq, H, P] (already normalised: no softmax), no reference points.
// TO: the element type of offsets / logits — float, or the value's 16-bit type (LiftArgs.ol16)
template <typename TO> __device__ __forceinline__ float2 tile_load2(const TO* p) {
  if constexpr (sizeof(TO) == 4) return *reinterpret_cast<const float2*>(p);
  else {
    TO v[2];
    *reinterpret_cast<uint32_t*>(v) = *reinterpret_cast<const uint32_t*>(p);
    return make_float2(elem<TO>::to_float(v[0]), elem<TO>::to_float(v[1]));
  }
}
template <typename TO> __device__ __forceinline__ void tile_store2(TO* p, float x, float y) {
  if constexpr (sizeof(TO) == 4) *reinterpret_cast<float2*>(p) = make_float2(x, y);
  else {
    TO v[2] = {elem<TO>::from_float(x), elem<TO>::from_float(y)};
    *reinterpret_cast<uint32_t*>(p) = *reinterpret_cast<const uint32_t*>(v);
  }
}

template <int P, bool BWD, bool K1 = false, typename TO = float>
__device__ __forceinline__ int4 tile_points(const LiftArgs& a, long bq, bool valid, int h, int pp, int wv, int lane,
                                            float (&rx)[P / 4], float (&ry)[P / 4], float (&rw)[P / 4], int4* wbox) {
  constexpr int PW = P / 4;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const TO* __restrict__ offp = (const TO*)a.offsets + bq * a.off_stride + h * 2 * P;
  const TO* __restrict__ lgp = (const TO*)a.logits + bq * a.log_stride + h * P;
  const float* __restrict__ rp = K1 ? nullptr : a.ref + bq * a.Z * 2;
  float2 off[PW], ref[PW];
  float lg[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = pp + 4 * j;
    off[j] = tile_load2<TO>(offp + 2 * p);
    if constexpr (!K1) ref[j] = *reinterpret_cast<const float2*>(rp + (p % a.Z) * 2);
    else ref[j] = make_float2(0.0f, 0.0f);
    lg[j] = elem<TO>::to_float(lgp[p]);
  }
  int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const float lx = K1 ? off[j].x : ref[j].x + off[j].x / fwf, ly = K1 ? off[j].y : ref[j].y + off[j].y / fhf;
    rx[j] = lx * fwf - 0.5f; ry[j] = ly * fhf - 0.5f;
    const Footprint f = footprint_px(rx[j], ry[j], a.fh, a.fw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (valid && (BWD ? f.m[k] : f.w[k]) != 0.0f) {
        x0 = min(x0, f.xc[k & 1]); x1 = max(x1, f.xc[k & 1]);
        y0 = min(y0, f.yc[k >> 1]); y1 = max(y1, f.yc[k >> 1]);
      }
    }
  }
  x0 = wave_min_i32(x0); y0 = wave_min_i32(y0); x1 = wave_max_i32(x1); y1 = wave_max_i32(y1);
  if (lane == 0) wbox[wv] = make_int4(x0, y0, x1, y1);
  if constexpr (K1) {                                     // weights as given
#pragma unroll
    for (int j = 0; j < PW; ++j) rw[j] = valid ? lg[j] : 0.0f;
    __syncthreads();
    const int4 c0 = wbox[0], c1 = wbox[1], c2 = wbox[2], c3 = wbox[3];
    return make_int4(min(min(c0.x, c1.x), min(c2.x, c3.x)), min(min(c0.y, c1.y), min(c2.y, c3.y)),
                     max(max(c0.z, c1.z), max(c2.z, c3.z)), max(max(c0.w, c1.w), max(c2.w, c3.w)));
  }
  // softmax of the query's P logits: the quad holds them
  float m = lg[0];
#pragma unroll
  for (int j = 1; j < PW; ++j) m = fmaxf(m, lg[j]);
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, true)));
  float ssum = 0.0f;
#pragma unroll
  for (int j = 0; j < PW; ++j) { rw[j] = expf(lg[j] - m); ssum += rw[j]; }
  ssum = add_xor<2>(add_xor<1>(ssum));
#pragma unroll
  for (int j = 0; j < PW; ++j) rw[j] = valid ? rw[j] / ssum : 0.0f;
  __syncthreads();
  const int4 b0 = wbox[0], b1 = wbox[1], b2 = wbox[2], b3 = wbox[3];
  return make_int4(min(min(b0.x, b1.x), min(b2.x, b3.x)), min(min(b0.y, b1.y), min(b2.y, b3.y)),
                   max(max(b0.z, b1.z), max(b2.z, b3.z)), max(max(b0.w, b1.w), max(b2.w, b3.w)));
}

// acc[0..DH) += c * (DH consecutive elements at p): LDS or global, 16-byte pieces
template <typename T, int DH>
__device__ __forceinline__ void tile_axpy(const T* __restrict__ p, float c, float (&acc)[DH]) {
  constexpr int V = 16 / (int)sizeof(T);                  // elements per 16-byte piece
#pragma unroll
  for (int i = 0; i < DH / V; ++i) {
    float v[V];
    vec_io<T, V>::load(p + V * i, v);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[V * i + e] = fmaf(c, v[e], acc[V * i + e]);
  }
}
template <typename T, int DH>
__device__ __forceinline__ float tile_dot(const T* __restrict__ p, const float (&g)[DH]) {
  constexpr int V = 16 / (int)sizeof(T);
  float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
  for (int i = 0; i < DH / V; ++i) {
    float v[V];
    vec_io<T, V>::load(p + V * i, v);
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      d0 = fmaf(g[V * i + e], v[e], d0);
      d1 = fmaf(g[V * i + e + 1], v[e + 1], d1);
      d0 = fmaf(g[V * i + e + 2], v[e + 2], d0);
      d1 = fmaf(g[V * i + e + 3], v[e + 3], d1);
    }
  }
  return d0 + d1;
}

__device__ __forceinline__ int4 box_union(const int4 a, const int4 b) {
  return make_int4(min(a.x, b.x), min(a.y, b.y), max(a.z, b.z), max(a.w, b.w));
}

// A box wider (taller) than the window — offsets scattered by several pixels per query, a trained layer's — cannot be
// copied whole.  Anchoring the window at the box's smallest corner (round 4) then leaves the far side of the tile without
// any margin: most waves had a corner outside and took the slow path.  The window is centred on the block's MEAN sampling
// position instead (the sampling pattern of a head leans one way; the mean follows it), clamped into the box.  Only in
// that case (block-uniform): one more barrier, three wave sums.  UBV_TILE_CENTER=0: the round-4 anchor (A/B runs).
template <int HPB, int PW>
__device__ __forceinline__ void tile_recentre(const LiftArgs& a, int4& bb, const float (&rx)[HPB][PW], const float (&ry)[HPB][PW],
                                              bool valid, int wv, int lane, float (*cred)[4], int centre) {
  const bool wide = bb.z - bb.x >= kWin, tall = bb.w - bb.y >= kWin;
  if (!centre || !(wide || tall)) return;                 // block-uniform
  float sx = 0.0f, sy = 0.0f, n = 0.0f;
  const float xm = (float)(a.fw - 1), ym = (float)(a.fh - 1);
#pragma unroll
  for (int hh = 0; hh < HPB; ++hh)
#pragma unroll
    for (int j = 0; j < PW; ++j)
      if (valid) { sx += fminf(fmaxf(rx[hh][j], 0.0f), xm); sy += fminf(fmaxf(ry[hh][j], 0.0f), ym); n += 1.0f; }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { sx += __shfl_xor(sx, m, 64); sy += __shfl_xor(sy, m, 64); n += __shfl_xor(n, m, 64); }
  if (lane == 0) { cred[wv][0] = sx; cred[wv][1] = sy; cred[wv][2] = n; }
  __syncthreads();
  const float tn = fmaxf(cred[0][2] + cred[1][2] + cred[2][2] + cred[3][2], 1.0f);
  const int cx = (int)floorf((cred[0][0] + cred[1][0] + cred[2][0] + cred[3][0]) / tn) - (kWin / 2 - 1);
  const int cy = (int)floorf((cred[0][1] + cred[1][1] + cred[2][1] + cred[3][1]) / tn) - (kWin / 2 - 1);
  if (wide) bb.x = min(max(cx, bb.x), bb.z - (kWin - 1));
  if (tall) bb.y = min(max(cy, bb.y), bb.w - (kWin - 1));
}

// Window of a block from its box: origin (clamped into the map like win_origin) and the rows / columns worth loading.
struct TileWin { int rows, cols; };
__device__ __forceinline__ TileWin tile_window(const LiftArgs& a, const int4 bb, WinGeom& g, int max_box) {
  g.wx0 = min(max(bb.x, 0), max(a.fw - kWin, 0));
  g.wy0 = min(max(bb.y, 0), max(a.fh - kWin, 0));
  TileWin t;
  t.cols = min(max(bb.z - g.wx0 + 1, 0), min(kWin, a.fw));
  t.rows = min(max(bb.w - g.wy0 + 1, 0), min(kWin, a.fh));
  // a box beyond `max_box` pixels is not copied: every lane then fetches its corners from global memory itself (the
  // window is a cache) — with offsets scattered by several pixels per query the copy of a 16 x 16 window costs a
  // 4-point block more than the 1 024 corner rows it serves (block-uniform)
  if (t.cols * t.rows > max_box) { t.cols = 0; t.rows = 0; }
  return t;
}

// Fills rows [0, rows) x columns [0, cols) of the window: 8 lanes per pixel (one 128-byte line per 8 lanes), 32
// pixels = 2 window rows per pass, all passes' loads in flight before the first LDS store.  Ends with a barrier.
__device__ __forceinline__ void tile_fill(const LiftArgs& a, const WinGeom& g, const TileWin t, int line, int rowi,
                                          unsigned char* __restrict__ win) {
  const int tid = threadIdx.x, piece = tid & 7, pxl = tid >> 3;
  const int dx = pxl & 15, dyl = pxl >> 4;
  const int cdx = min(dx, max(t.cols - 1, 0));            // columns past the box re-read its last one (same line: free)
  // (rowi = DWORDS per pixel over all heads — 16-bit maps: half their element count; `line` = which 128-byte line of
  //  the pixel: HPB heads)
  const float* vb = (const float*)a.value + (long)g.b * a.fh * a.fw * rowi + line * 32 + piece * 4;
  // element offset of this thread's pixel in pass 0, and the (uniform) step of a pass = 2 map rows; a pass whose second
  // row lies past the box re-reads its first one (dyl = 1 lanes step back one row)
  const unsigned off0 = (unsigned)(((g.wy0 + dyl) * a.fw + g.wx0 + cdx) * rowi);
  const unsigned step = (unsigned)(2 * a.fw * rowi), back = (unsigned)(dyl * a.fw * rowi);
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (2 * i < t.rows) {                                  // block-uniform
      const unsigned o = off0 + (unsigned)i * step - ((2 * i + 1 < t.rows) ? 0u : back);
      x = *reinterpret_cast<const uint4*>(gather_ptr(vb, o));
    }
    v[i] = x;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (2 * i < t.rows)
      *reinterpret_cast<uint4*>(win + (2 * i + dyl) * kTWinRow + dx * kWinRowB + piece * 16) = v[i];
  // an EMPTY box (every point of the tile off the map) loads nothing, and the no-miss paths read a weightless corner
  // from pixel (0, 0) and multiply by 0: stale LDS bits there may be NaN / Inf — make the pixel a finite one
  if (t.rows == 0 && tid < 8) *reinterpret_cast<uint4*>(win + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
}

// LDS byte offset of a corner's pixel, or -1 when it lies outside the loaded part of the window.
__device__ __forceinline__ int tile_row(int xc, int yc, const WinGeom& g, const TileWin t) {
  const int dx = xc - g.wx0, dy = yc - g.wy0;
  return ((unsigned)dx < (unsigned)t.cols && (unsigned)dy < (unsigned)t.rows) ? dy * kTWinRow + dx * kWinRowB : -1;
}

// ------------------------------------------------------------------------------------------------
// Forward.  Two barriers per head of the line (block box) + one (window fill).  DH = 16: the two heads of a 128-byte
// line are one block's work — both boxes first, ONE window for their union, then the corners head by head.
template <int P, int DH, bool K1 = false, typename T = float, bool OL16 = false>
__global__ __launch_bounds__(256) void lift_tile_fwd_kernel(const LiftArgs a, int chunk, int max_box, int centre) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int ES = (int)sizeof(T);
  constexpr int PW = P / 4, HPB = 128 / (DH * ES);        // heads per 128-byte line
  __shared__ int4 wbox[HPB][4];
  __shared__ float cred[4][4];
  using TO = std::conditional_t<OL16, T, float>;
  WinGeom g;
  if (!tile_decode8<HPB>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  constexpr int rowi = 8 * DH;                            // elements per pixel / per query row (H = 8)
  const int li = wv * 16 + (lane >> 2), pp = lane & 3;
  float rx[HPB][PW], ry[HPB][PW], rw[HPB][PW];
  int b, q;
  const bool valid = lift_query(a, g.tile, li, b, q);
  if (!valid) q = 0;
  const long bq = (long)b * a.Nq + q;
  int4 bb = make_int4(INT_MAX, INT_MAX, -1, -1);
#pragma unroll
  for (int hh = 0; hh < HPB; ++hh)
    bb = box_union(bb, tile_points<P, false, K1, TO>(a, bq, valid, g.hg * HPB + hh, pp, wv, lane, rx[hh], ry[hh], rw[hh], wbox[hh]));
  tile_recentre<HPB, PW>(a, bb, rx, ry, valid, wv, lane, cred, centre);
  const TileWin tw = tile_window(a, bb, g, max_box);
  tile_fill(a, g, tw, g.hg, rowi * ES / 4, win);
#pragma unroll
  for (int hh = 0; hh < HPB; ++hh) {
    const int h = g.hg * HPB + hh;
    const T* vb = (const T*)a.value + (long)g.b * a.fh * a.fw * rowi + h * DH;      // wave-uniform
    const unsigned char* wh = win + hh * (DH * ES);       // this head's part of a window pixel
    float acc[DH];
#pragma unroll
    for (int i = 0; i < DH; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const Footprint f = footprint_px(rx[hh][j], ry[hh][j], a.fh, a.fw);
      float c[4];
      int wr[4];
      bool miss = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c[k] = rw[hh][j] * f.w[k];
        wr[k] = tile_row(f.xc[k & 1], f.yc[k >> 1], g, tw);
        miss = miss || (wr[k] < 0 && c[k] != 0.0f);
      }
      if (__ballot(miss) == 0ull) {
        // the wave's corners are all in the window (or weightless: any row will do): straight-line code, the LDS reads
        // of the point in flight together
#pragma unroll
        for (int k = 0; k < 4; ++k) tile_axpy<T, DH>(reinterpret_cast<const T*>(wh + (unsigned)max(wr[k], 0)), c[k], acc);
      } else {
        // some lane has a corner outside the window: the window's corners first, still straight-line (a corner outside
        // contributes with weight 0), then each corner SLOT that any lane missed as one more pass in which only the
        // missing lanes fetch their row from global memory (round 4 branched per lane and corner: 8 divergent bodies)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tile_axpy<T, DH>(reinterpret_cast<const T*>(wh + (unsigned)max(wr[k], 0)), wr[k] >= 0 ? c[k] : 0.0f, acc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ms = wr[k] < 0 && c[k] != 0.0f;
          if (__ballot(ms) != 0ull) {
            if (ms) tile_axpy<T, DH>(gather_ptr(vb, (unsigned)(f.idx[k] * rowi)), c[k], acc);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < DH; ++i) acc[i] = add_xor<2>(add_xor<1>(acc[i]));
    if (valid) {
      constexpr int Q = DH / 4;                           // elements per lane of the quad
      float o[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) o[i] = pp == 0 ? acc[i] : pp == 1 ? acc[Q + i] : pp == 2 ? acc[2 * Q + i] : acc[3 * Q + i];
      vec_io<T, Q>::store((T*)a.out + bq * rowi + h * DH + pp * Q, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward, PERSISTENT and software-pipelined (round 6; f32 maps, Dh = 32, module form).  lift_tile_fwd_kernel's block is a
// chain of two global round trips (offsets / logits / anchors, then the window) in front of its arithmetic, and the four
// blocks of a CU pass through their phases more or less together (the phases add up).  Here a block stays
// resident and walks units k = (tile, head) of its XCD's range with three units in flight:
//     D(k)    corners of unit k from the LDS window            <- arithmetic
//     B(k+1)  points / box / window geometry of unit k + 1     (its raw operands were requested one iteration ago)
//     S(k+1)  window of unit k + 1: registers -> LDS           (its loads were requested one iteration ago)
//     F(k+2)  window loads of unit k + 2 -> registers          (requested now, consumed one iteration later)
//     R(k+3)  raw operand loads of unit k + 3 -> registers
// so that both round trips of a unit overlap another unit's corners.  Two barriers per unit, as before: the one inside
// B (box merge; it also certifies that every wave has finished D(k) before S(k+1) overwrites the window) and the one
// after S.  UBV_TILE_PIPE=0: lift_tile_fwd_kernel for these instances too (A/B runs).
template <int PW> struct TileRaw { float2 off[PW], ref[PW]; float lg[PW]; long bq; int b, tile, head; bool valid, live; };
template <int PW> struct TilePts { float rx[PW], ry[PW], rw[PW]; long bq; int head; bool valid, live; WinGeom g; TileWin tw; };

template <int P>
__device__ __forceinline__ void pipe_request(const LiftArgs& a, long unit, long end, int li, int pp, TileRaw<P / 4>& r) {
  constexpr int PW = P / 4;
  r.live = unit < end;                                    // (block-uniform)
  const long u = r.live ? unit : end - 1;                 // past the end: the last unit again, never used
  const int item = (int)(u >> 3);
  r.head = (int)(u & 7);
  r.tile = item;
  int q;
  r.valid = lift_query(a, item, li, r.b, q);
  if (!r.valid) q = 0;
  r.bq = (long)r.b * a.Nq + q;
  const float* __restrict__ offp = (const float*)a.offsets + r.bq * a.off_stride + r.head * 2 * P;
  const float* __restrict__ lgp = (const float*)a.logits + r.bq * a.log_stride + r.head * P;
  const float* __restrict__ rp = a.ref + r.bq * a.Z * 2;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = pp + 4 * j;
    r.off[j] = *reinterpret_cast<const float2*>(offp + 2 * p);
    r.ref[j] = *reinterpret_cast<const float2*>(rp + (p % a.Z) * 2);
    r.lg[j] = lgp[p];
  }
}

// B: tile_points' arithmetic on operands already in registers + the window geometry.  One barrier (two when the box is
// over-wide: tile_recentre).
template <int P>
__device__ __forceinline__ void pipe_points(const LiftArgs& a, const TileRaw<P / 4>& r, int wv, int lane, int4* wbox,
                                            float (*cred)[4], int max_box, int centre, TilePts<P / 4>& t) {
  constexpr int PW = P / 4;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  t.bq = r.bq; t.head = r.head; t.valid = r.valid; t.live = r.live;
  t.g.b = r.b; t.g.tile = r.tile; t.g.hg = r.head;
  int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const float lx = r.ref[j].x + r.off[j].x / fwf, ly = r.ref[j].y + r.off[j].y / fhf;
    t.rx[j] = lx * fwf - 0.5f; t.ry[j] = ly * fhf - 0.5f;
    const Footprint f = footprint_px(t.rx[j], t.ry[j], a.fh, a.fw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r.valid && f.w[k] != 0.0f) {
        x0 = min(x0, f.xc[k & 1]); x1 = max(x1, f.xc[k & 1]);
        y0 = min(y0, f.yc[k >> 1]); y1 = max(y1, f.yc[k >> 1]);
      }
    }
  }
  x0 = wave_min_i32(x0); y0 = wave_min_i32(y0); x1 = wave_max_i32(x1); y1 = wave_max_i32(y1);
  if (lane == 0) wbox[wv] = make_int4(x0, y0, x1, y1);
  float m = r.lg[0];
#pragma unroll
  for (int j = 1; j < PW; ++j) m = fmaxf(m, r.lg[j]);
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, true)));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xf, 0xf, true)));
  float ssum = 0.0f;
#pragma unroll
  for (int j = 0; j < PW; ++j) { t.rw[j] = expf(r.lg[j] - m); ssum += t.rw[j]; }
  ssum = add_xor<2>(add_xor<1>(ssum));
#pragma unroll
  for (int j = 0; j < PW; ++j) t.rw[j] = r.valid ? t.rw[j] / ssum : 0.0f;
  __syncthreads();
  int4 bb = box_union(box_union(wbox[0], wbox[1]), box_union(wbox[2], wbox[3]));
  {
    float rx1[1][PW], ry1[1][PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) { rx1[0][j] = t.rx[j]; ry1[0][j] = t.ry[j]; }
    tile_recentre<1, PW>(a, bb, rx1, ry1, r.valid, wv, lane, cred, centre);
  }
  t.tw = tile_window(a, bb, t.g, max_box);
}

// F: the window's loads of tile_fill, into registers
__device__ __forceinline__ void pipe_fill_request(const LiftArgs& a, const WinGeom& g, const TileWin t, uint4 (&v)[8]) {
  constexpr int rowi = 256;
  const int tid = threadIdx.x, piece = tid & 7, pxl = tid >> 3;
  const int dx = pxl & 15, dyl = pxl >> 4;
  const int cdx = min(dx, max(t.cols - 1, 0));
  const float* vb = (const float*)a.value + (long)g.b * a.fh * a.fw * rowi + g.hg * 32 + piece * 4;
  const unsigned off0 = (unsigned)(((g.wy0 + dyl) * a.fw + g.wx0 + cdx) * rowi);
  const unsigned step = (unsigned)(2 * a.fw * rowi), back = (unsigned)(dyl * a.fw * rowi);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (2 * i < t.rows) {
      const unsigned o = off0 + (unsigned)i * step - ((2 * i + 1 < t.rows) ? 0u : back);
      x = *reinterpret_cast<const uint4*>(gather_ptr(vb, o));
    }
    v[i] = x;
  }
}

// S: registers -> LDS.  Ends with a barrier.
__device__ __forceinline__ void pipe_fill_store(const TileWin t, const uint4 (&v)[8], unsigned char* __restrict__ win) {
  const int tid = threadIdx.x, piece = tid & 7, pxl = tid >> 3;
  const int dx = pxl & 15, dyl = pxl >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (2 * i < t.rows)
      *reinterpret_cast<uint4*>(win + (2 * i + dyl) * kTWinRow + dx * kWinRowB + piece * 16) = v[i];
  if (t.rows == 0 && tid < 8) *reinterpret_cast<uint4*>(win + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
}

// D: the corners of one unit from the LDS window -> its rows of the output
template <int P>
__device__ __forceinline__ void pipe_corners(const LiftArgs& a, const TilePts<P / 4>& cur, const unsigned char* __restrict__ win, int pp) {
  constexpr int PW = P / 4, DH = 32, rowi = 256;
  const float* vb = (const float*)a.value + (long)cur.g.b * a.fh * a.fw * rowi + cur.head * DH;
  float acc[DH];
#pragma unroll
  for (int i = 0; i < DH; ++i) acc[i] = 0.0f;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const Footprint f = footprint_px(cur.rx[j], cur.ry[j], a.fh, a.fw);
    float c[4];
    int wr[4];
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[k] = cur.rw[j] * f.w[k];
      wr[k] = tile_row(f.xc[k & 1], f.yc[k >> 1], cur.g, cur.tw);
      miss = miss || (wr[k] < 0 && c[k] != 0.0f);
    }
    if (__ballot(miss) == 0ull) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tile_axpy<float, DH>(reinterpret_cast<const float*>(win + (unsigned)max(wr[k], 0)), c[k], acc);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        tile_axpy<float, DH>(reinterpret_cast<const float*>(win + (unsigned)max(wr[k], 0)), wr[k] >= 0 ? c[k] : 0.0f, acc);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ms = wr[k] < 0 && c[k] != 0.0f;
        if (__ballot(ms) != 0ull) {
          if (ms) tile_axpy<float, DH>(gather_ptr(vb, (unsigned)(f.idx[k] * rowi)), c[k], acc);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < DH; ++i) acc[i] = add_xor<2>(add_xor<1>(acc[i]));
  if (cur.valid) {
    constexpr int Q = DH / 4;
    float o[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) o[i] = pp == 0 ? acc[i] : pp == 1 ? acc[Q + i] : pp == 2 ? acc[2 * Q + i] : acc[3 * Q + i];
    vec_io<float, Q>::store((float*)a.out + cur.bq * rowi + cur.head * DH + pp * Q, o);
  }
}

template <int P, bool LIGHT>
__global__ __launch_bounds__(256) void lift_tile_fwd_pipe_kernel(const LiftArgs a, int chunk, long units, int max_box, int centre) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int PW = P / 4;
  __shared__ int4 wbox[4];
  __shared__ float cred[4][4];
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int li = wv * 16 + (lane >> 2), pp = lane & 3;
  // XCD x = blockIdx.x % 8 owns units [x chunk, (x + 1) chunk); its blocks take them with stride gridDim.x / 8
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long stride = gridDim.x >> 3;
  const long end = min((long)(xcd + 1) * chunk, units);
  long u = (long)xcd * chunk + slot;
  if (u >= end) return;
  TileRaw<PW> raw;
  TilePts<PW> cur, nxt;
  uint4 vf[8];
  if constexpr (LIGHT) {
    // LIGHT: only the raw operands of the next unit are in flight during the corners (the window's registers do not
    // live across them: ~100 VGPRs instead of 158 / 199, four blocks per CU stay resident)
    pipe_request<P>(a, u, end, li, pp, raw);
    for (;;) {
      pipe_points<P>(a, raw, wv, lane, wbox, cred, max_box, centre, cur);      // (its barrier: every wave is done with the previous window)
      pipe_fill_request(a, cur.g, cur.tw, vf);
      pipe_request<P>(a, u + stride, end, li, pp, raw);
      pipe_fill_store(cur.tw, vf, win);
      pipe_corners<P>(a, cur, win, pp);
      u += stride;
      if (u >= end) break;                                  // (block-uniform)
    }
    return;
  }
  // prologue: unit 0 up to its window in LDS, unit 1 up to its window loads, unit 2's raw operands
  pipe_request<P>(a, u, end, li, pp, raw);
  pipe_points<P>(a, raw, wv, lane, wbox, cred, max_box, centre, cur);
  pipe_fill_request(a, cur.g, cur.tw, vf);
  pipe_request<P>(a, u + stride, end, li, pp, raw);
  pipe_fill_store(cur.tw, vf, win);
  pipe_points<P>(a, raw, wv, lane, wbox, cred, max_box, centre, nxt);
  if (nxt.live) pipe_fill_request(a, nxt.g, nxt.tw, vf);
  pipe_request<P>(a, u + 2 * stride, end, li, pp, raw);
  for (;;) {
    pipe_corners<P>(a, cur, win, pp);
    if (!nxt.live) break;                                 // (block-uniform)
    u += stride;
    // ---- B(k+2) first: its barrier certifies that every wave is done with the window of unit k
    TilePts<PW> nn;
    pipe_points<P>(a, raw, wv, lane, wbox, cred, max_box, centre, nn);
    // ---- S(k+1): the window of `nxt` (loads requested one iteration ago)
    pipe_fill_store(nxt.tw, vf, win);
    cur = nxt;
    nxt = nn;
    // ---- F(k+2), R(k+3)
    if (nxt.live) pipe_fill_request(a, nxt.g, nxt.tw, vf);
    pipe_request<P>(a, u + 2 * stride, end, li, pp, raw);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, query side: d(offsets), d(logits) — the forward's block with the query's grad_out row in registers and a
// dot product per corner instead of an axpy.  BINS: the kernel also appends every point to the bucket of each owner
// tile that holds one of its corners of non-zero coefficient (what lift_bin_kernel<MODE 0> does, same record format,
// same fixed-capacity buckets + overflow list; see there): ranks inside the wave through LDS counters on an 8x8 torus
// of tile slots, one returning global atomic per occupied slot.  The caller zeroes the counters.
template <int P, int DH, int BINS, bool K1 = false, typename T = float, bool OL16 = false>
__global__ __launch_bounds__(256) void lift_tile_bwd_query_kernel(const LiftArgs a, int chunk, int tiles_x, int tiles, int max_box, int centre) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int ES = (int)sizeof(T);
  __shared__ int4 wbox[128 / (DH * ES)][4];
  __shared__ float cred[4][4];
  using TO = std::conditional_t<OL16, T, float>;
  // per wave: a 4x4 torus of tile slots — occupant tile, local count, global base (a wave's 16 queries x 4 points
  // reach a handful of tiles; two tiles that collide on the torus take the direct global path)
  __shared__ volatile int slot_tile[4][16];
  __shared__ int slot_cnt[4][16], slot_base[4][16];
  constexpr int PW = P / 4, HPB = 128 / (DH * ES);
  WinGeom g;
  if (!tile_decode8<HPB>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  constexpr int rowi = 8 * DH;
  const int li = wv * 16 + (lane >> 2), pp = lane & 3;
  int b, q;
  const bool valid = lift_query(a, g.tile, li, b, q);
  if (!valid) q = 0;
  const long bq = (long)b * a.Nq + q;
  if (BINS != 0 && lane < 16) { slot_tile[wv][lane] = -1; slot_cnt[wv][lane] = 0; }
  float rx[HPB][PW], ry[HPB][PW], rw[HPB][PW];
  int4 bb = make_int4(INT_MAX, INT_MAX, -1, -1);
#pragma unroll
  for (int hh = 0; hh < HPB; ++hh)
    bb = box_union(bb, tile_points<P, true, K1, TO>(a, bq, valid, g.hg * HPB + hh, pp, wv, lane, rx[hh], ry[hh], rw[hh], wbox[hh]));
  const int qbx0 = max(bb.x, 0) >> 3, qby0 = max(bb.y, 0) >> 3;       // BINS = 2: origin of the block's 8 x 8 neighbourhood of owner tiles
  tile_recentre<HPB, PW>(a, bb, rx, ry, valid, wv, lane, cred, centre);
  const TileWin tw = tile_window(a, bb, g, max_box);
  tile_fill(a, g, tw, g.hg, rowi * ES / 4, win);

#pragma unroll
  for (int hh = 0; hh < HPB; ++hh) {
  const int h = g.hg * HPB + hh;
  const unsigned char* wh = win + hh * (DH * ES);
  float go[DH];
  {
    constexpr int V = 16 / ES;
    const T* gp = (const T*)a.gout + bq * rowi + h * DH;
#pragma unroll
    for (int i = 0; i < DH / V; ++i) {
      float v[V];
      vec_io<T, V>::load(gp + V * i, v);
#pragma unroll
      for (int e = 0; e < V; ++e) go[V * i + e] = v[e];
    }
  }
  const T* vb = (const T*)a.value + (long)g.b * a.fh * a.fw * rowi + h * DH;      // wave-uniform

  float gw[PW], gx[PW], gy[PW];
  float sp = 0.0f;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const Footprint f = footprint_px(rx[hh][j], ry[hh][j], a.fh, a.fw);
    float d[4];
    int wr[4];
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wr[k] = tile_row(f.xc[k & 1], f.yc[k >> 1], g, tw);
      miss = miss || (wr[k] < 0 && valid && f.m[k] != 0.0f);
    }
    if (__ballot(miss) == 0ull) {          // all in the window (or masked out): straight-line code
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = tile_dot<T, DH>(reinterpret_cast<const T*>(wh + (unsigned)max(wr[k], 0)), go) * f.m[k];
    } else {
      // (as in the forward: the window's corners straight-line, then one pass per corner slot that some lane missed)
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = tile_dot<T, DH>(reinterpret_cast<const T*>(wh + (unsigned)max(wr[k], 0)), go);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ms = wr[k] < 0 && valid && f.m[k] != 0.0f;
        if (__ballot(ms) != 0ull) {
          if (ms) d[k] = tile_dot<T, DH>(gather_ptr(vb, (unsigned)(f.idx[k] * rowi)), go);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = (wr[k] >= 0 || (valid && f.m[k] != 0.0f)) ? d[k] * f.m[k] : 0.0f;
    }
    const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
    gw[j] = hy * hx * d[0] + hy * f.lx * d[1] + f.ly * hx * d[2] + f.ly * f.lx * d[3];
    gx[j] = (d[1] - d[0]) * hy + (d[3] - d[2]) * f.ly;
    gy[j] = (d[2] - d[0]) * hx + (d[3] - d[1]) * f.lx;
    sp = fmaf(rw[hh][j], gw[j], sp);

    if constexpr (BINS == 1) {
      // one lane = one point: append it to the bucket of every tile that holds a corner of non-zero coefficient
      const int tile_base = (g.b * a.H + h) * tiles;
      int* __restrict__ cntp = a.bin_cnt + tile_base;
      float4* __restrict__ binp = a.bins + (long)tile_base * a.cap;
      const float4 rec = make_float4(rx[hh][j], ry[hh][j], rw[hh][j], __int_as_float(q));
      auto put = [&](int tile, int idx) {
        if (idx < a.cap) {
          binp[(long)tile * a.cap + idx] = rec;
        } else {
          const int o = atomicAdd(a.ovf_n, 1);
          if (o < a.ovf_cap) { a.ovf_rec[o] = rec; a.ovf_tile[o] = tile_base + tile; }
        }
      };
      int tk[4], hs[4], rank[4];
      bool nz[4], lead[4], local[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        nz[k] = rw[hh][j] != 0.0f && f.w[k] != 0.0f;
        const int tx = f.xc[k & 1] >> 3, ty = f.yc[k >> 1] >> 3;
        tk[k] = ty * tiles_x + tx;
        hs[k] = ((ty & 3) << 2) | (tx & 3);
      }
      // a tile receives the record once: through its first corner with non-zero weight
      lead[0] = nz[0];
      lead[1] = nz[1] && !(nz[0] && tk[1] == tk[0]);
      lead[2] = nz[2] && !(nz[0] && tk[2] == tk[0]) && !(nz[1] && tk[2] == tk[1]);
      lead[3] = nz[3] && !(nz[0] && tk[3] == tk[0]) && !(nz[1] && tk[3] == tk[1]) && !(nz[2] && tk[3] == tk[2]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        local[k] = false;
        rank[k] = 0;
        if (lead[k]) {
          volatile int* st = &slot_tile[wv][hs[k]];
          if (*st == -1) *st = tk[k];
          local[k] = *st == tk[k];
          if (local[k]) rank[k] = atomicAdd(&slot_cnt[wv][hs[k]], 1);
          else put(tk[k], atomicAdd(cntp + tk[k], 1));
        }
      }
      if (lane < 16) {
        const int c = slot_cnt[wv][lane];
        if (c > 0) slot_base[wv][lane] = atomicAdd(cntp + slot_tile[wv][lane], c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lead[k] && local[k]) put(tk[k], slot_base[wv][hs[k]] + rank[k]);
      if (lane < 16) { slot_tile[wv][lane] = -1; slot_cnt[wv][lane] = 0; }
    }
  }
  if constexpr (BINS == 2) {
    // QUERY records (lift_bwd_value_q_kernel): the points of this (query, head) stored once, the query index appended
    // to the bucket of every owner tile one of its points touches.  Tiles are addressed relative to the block's box
    // (every corner of non-zero coefficient lies inside it) on an 8 x 8 grid; a corner beyond it — a box wider than 64
    // pixels — and the points of an entry that does not fit its bucket go to the overflow list as point records.
    const int tile_base = (g.b * a.H + h) * tiles;
    int* __restrict__ cntp = a.bin_cnt + tile_base;
    int* __restrict__ qbin = reinterpret_cast<int*>(a.bins) + (long)tile_base * a.cap;
    auto overflow = [&](int j, int tile) {
      const int o = atomicAdd(a.ovf_n, 1);
      if (o < a.ovf_cap) { a.ovf_rec[o] = make_float4(rx[hh][j], ry[hh][j], rw[hh][j], __int_as_float(q)); a.ovf_tile[o] = tile_base + tile; }
    };
    uint32_t mlo = 0u, mhi = 0u;                           // bit (ty - qby0) * 8 + (tx - qbx0)
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      if (valid) {
        float* qp = a.qpts + ((bq * a.H + h) * P + (pp + 4 * j)) * 3;
        qp[0] = rx[hh][j]; qp[1] = ry[hh][j]; qp[2] = rw[hh][j];
      }
      const Footprint f = footprint_px(rx[hh][j], ry[hh][j], a.fh, a.fw);
      int far_t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        far_t[k] = -1;
        if (rw[hh][j] != 0.0f && f.w[k] != 0.0f) {
          const int tx = f.xc[k & 1] >> 3, ty = f.yc[k >> 1] >> 3;
          const int dx = tx - qbx0, dy = ty - qby0;
          if ((unsigned)dx < 8u && (unsigned)dy < 8u) {
            const int bit = dy * 8 + dx;
            if (bit < 32) mlo |= 1u << bit; else mhi |= 1u << (bit - 32);
          } else {
            far_t[k] = ty * tiles_x + tx;
          }
        }
      }
      // (far tiles: once per point and tile)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool first = far_t[k] >= 0;
#pragma unroll
        for (int k2 = 0; k2 < k; ++k2) first = first && far_t[k2] != far_t[k];
        if (first) overflow(j, far_t[k]);
      }
    }
    // the quad's tiles: OR over its four lanes, kept by lane pp == 0
    mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xf, 0xf, true);
    mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xf, 0xf, true);
    mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0x4E, 0xf, 0xf, true);
    mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0x4E, 0xf, 0xf, true);
    uint64_t m = pp == 0 ? ((uint64_t)mhi << 32) | mlo : 0ull;
    // rounds: every query's next tile.  Ranks inside the wave through the LDS slot counters (4 x 4 torus of tiles, as
    // the point records did), ONE returning global atomic per occupied slot, all of them in flight together.
    while (__ballot(m != 0ull) != 0ull) {
      const bool act = m != 0ull;
      const int bit = act ? __ffsll((unsigned long long)m) - 1 : 0;
      m &= m - 1ull;
      const int tx = qbx0 + (bit & 7), ty = qby0 + (bit >> 3);
      const int tk = ty * tiles_x + tx, hs = ((ty & 3) << 2) | (tx & 3);
      bool local = false;
      int rank = 0, idx = 0;
      if (act) {
        volatile int* stp = &slot_tile[wv][hs];
        if (*stp == -1) *stp = tk;
        local = *stp == tk;
        if (local) rank = atomicAdd(&slot_cnt[wv][hs], 1);
        else idx = atomicAdd(cntp + tk, 1);
      }
      if (lane < 16) {
        const int c = slot_cnt[wv][lane];
        if (c > 0) slot_base[wv][lane] = atomicAdd(cntp + slot_tile[wv][lane], c);
      }
      if (act && local) idx = slot_base[wv][hs] + rank;
      const bool fits = act && idx < a.cap;
      if (fits) qbin[(long)tk * a.cap + idx] = q;
      // an entry that did not fit: the quad's points that touch the tile become point records
      int ovt = (act && !fits) ? tk : -1;
      ovt = __builtin_amdgcn_update_dpp(0, ovt, 0x00, 0xf, 0xf, true);       // quad_perm [0, 0, 0, 0]: lane pp == 0's value
      if (__ballot(ovt >= 0) != 0ull) {
        if (ovt >= 0) {
#pragma unroll
          for (int j = 0; j < PW; ++j) {
            const Footprint f = footprint_px(rx[hh][j], ry[hh][j], a.fh, a.fw);
            bool hit = false;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              hit = hit || (rw[hh][j] != 0.0f && f.w[k] != 0.0f && (f.yc[k >> 1] >> 3) * tiles_x + (f.xc[k & 1] >> 3) == ovt);
            if (hit) overflow(j, ovt);
          }
        }
      }
      if (lane < 16) { slot_tile[wv][lane] = -1; slot_cnt[wv][lane] = 0; }
    }
  }
  sp = add_xor<2>(add_xor<1>(sp));
  if (valid) {
    const float fwf = (float)a.fw, fhf = (float)a.fh;
    TO* __restrict__ glog = (TO*)a.glog + bq * a.glog_stride + h * P;
    TO* __restrict__ goff = (TO*)a.goff + bq * a.goff_stride + h * 2 * P;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int p = pp + 4 * j;
      if constexpr (K1) {
        // the operator's gradients: d(weight) = the interpolated dot, d(location) = w * g * (W, H)
        glog[p] = elem<TO>::from_float(gw[j]);
        tile_store2<TO>(goff + 2 * p, rw[hh][j] * gx[j] * fwf, rw[hh][j] * gy[j] * fhf);
      } else {
        glog[p] = elem<TO>::from_float(rw[hh][j] * (gw[j] - sp));
        // d loc = w * g * W; d off = d loc / W (the reference's rounding)
        tile_store2<TO>(goff + 2 * p, (rw[hh][j] * gx[j] * fwf) / fwf, (rw[hh][j] * gy[j] * fhf) / fhf);
      }
    }
  }
  }  // heads of the line
}

// Dh = 32 or 16 (H = 8), one map per sample, grid-tiled queries, no visibility / count: the BEV self-attention and
// SCA-pts instances.  f32 maps; since round 5 also 16-bit maps (a 128-byte line then holds two heads of 32 channels or
// four of 16; offsets / logits f32 or in the map's type) — for the BACKWARD only by default: measured at bs = 2 (job
// r5t1, bf16 / fp16, us) the query-gradient + binning kernel takes the self-attention backward 159 -> 145 / 138 and
// SCA-pts' 265 -> 260 / 250, but the forward loses to the window / shared-footprint kernels (43 -> 56 / 49, 76 -> 95 /
// 80: every element of a corner costs a lane an unpack instruction on top of its FMA, where those kernels spread a
// corner over 8 lanes).  UBV_LIFT_TILE=0 switches the plan off (A/B runs against the window / gather kernels),
// UBV_LIFT_TILE16=0 keeps Dh = 16 off it, UBV_LIFT_TILE_LP=0 / 1 / 2: 16-bit maps never / backward only / both ways.
bool tile_ok(const LiftArgs& a, int Dh, int P, int dtype, bool bwd) {
  static const int env = getenv("UBV_LIFT_TILE") ? atoi(getenv("UBV_LIFT_TILE")) : 1;
  static const int env16 = getenv("UBV_LIFT_TILE16") ? atoi(getenv("UBV_LIFT_TILE16")) : 1;
  static const int envlp = getenv("UBV_LIFT_TILE_LP") ? atoi(getenv("UBV_LIFT_TILE_LP")) : 1;
  const bool type_ok = dtype == UBV_F32 ? a.ol16 == 0 : (envlp >= 2 || (envlp == 1 && bwd));
  return env != 0 && type_ok && (Dh == 32 || (Dh == 16 && env16 != 0)) && a.H == 8 && (P == 4 || P == 8) &&
         a.Nc == 1 && a.qw > 0 && a.vis0 == nullptr && a.count == nullptr && a.fh >= 1 && a.fw >= 1;
}

// largest pixel box a block copies into LDS (UBV_TILE_MAXBOX_FWD / _BWD, one value or "P4,P8"; 256 = always)
static int tile_max_box(const char* env, int P, int dflt4, int dflt8) {
  const char* e = getenv(env);
  int v4 = dflt4, v8 = dflt8;
  if (e != nullptr) {
    v4 = v8 = atoi(e);
    const char* c = strchr(e, ',');
    if (c != nullptr) v8 = atoi(c + 1);
  }
  return P == 4 ? v4 : v8;
}

// UBV_TILE_PIPE = blocks per CU of the persistent forward kernel (0: the one-unit-per-block kernel)
static int tile_pipe() {
  static const int v = getenv("UBV_TILE_PIPE") ? atoi(getenv("UBV_TILE_PIPE")) : 0;
  return v;
}

static int tile_centre() {
  static const int v = getenv("UBV_TILE_CENTER") ? atoi(getenv("UBV_TILE_CENTER")) : 1;
  return v;
}

// (tile, 128-byte line of heads) units of a launch: 8 heads of Dh elements of `es` bytes
static long tile_units(const LiftArgs& a, int Dh, int es) { return (long)a.total_tiles * (8 * Dh * es / 128); }

template <typename T, bool OL16>
static void tile_fwd_launch_t(const LiftArgs& a, int P, hipStream_t st, int Dh, int chunk, int mb4, int mb8) {
  const dim3 grid(8 * chunk), blk(256);
  if (Dh == 16) {
    if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_kernel<4, 16, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, mb4, tile_centre());
    else hipLaunchKernelGGL((lift_tile_fwd_kernel<8, 16, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, mb8, tile_centre());
  } else {
    if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_kernel<4, 32, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, mb4, tile_centre());
    else hipLaunchKernelGGL((lift_tile_fwd_kernel<8, 32, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, mb8, tile_centre());
  }
}

void tile_fwd_launch(const LiftArgs& a, int P, hipStream_t st, bool k1, int Dh, int dtype) {
  const long units = tile_units(a, Dh, dtype == UBV_F32 ? 4 : 2);
  const int chunk = (int)((units + 7) / 8);
  static const int mb4 = tile_max_box("UBV_TILE_MAXBOX_FWD", 4, 256, 256), mb8 = tile_max_box("UBV_TILE_MAXBOX_FWD", 8, 256, 256);
  const dim3 grid(8 * chunk), blk(256);
  if (k1) {                                               // (the operator's form: f32, Dh = 32 only)
    if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_kernel<4, 32, true>), grid, blk, kTWinLds, st, a, chunk, mb4, tile_centre());
    else hipLaunchKernelGGL((lift_tile_fwd_kernel<8, 32, true>), grid, blk, kTWinLds, st, a, chunk, mb8, tile_centre());
  } else if (dtype == UBV_F32 && Dh == 32 && tile_pipe() && a.Z >= 1) {
    // persistent, software-pipelined blocks: `per_cu` per CU (the window's LDS allows 4)
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t pr;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    }
    const int per_cu = tile_pipe();
    long nb = (long)cus * per_cu / 8 * 8;
    const long need = ((long)chunk) * 8;                  // one block per unit at most
    if (nb > need) nb = need;
    static const int light = getenv("UBV_TILE_PIPE_LIGHT") ? atoi(getenv("UBV_TILE_PIPE_LIGHT")) : 1;
    if (light) {
      if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_pipe_kernel<4, true>), dim3((unsigned)nb), blk, kTWinLds, st, a, chunk, units, mb4, tile_centre());
      else hipLaunchKernelGGL((lift_tile_fwd_pipe_kernel<8, true>), dim3((unsigned)nb), blk, kTWinLds, st, a, chunk, units, mb8, tile_centre());
    } else {
      if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_pipe_kernel<4, false>), dim3((unsigned)nb), blk, kTWinLds, st, a, chunk, units, mb4, tile_centre());
      else hipLaunchKernelGGL((lift_tile_fwd_pipe_kernel<8, false>), dim3((unsigned)nb), blk, kTWinLds, st, a, chunk, units, mb8, tile_centre());
    }
  } else if (dtype == UBV_F32) {
    tile_fwd_launch_t<float, false>(a, P, st, Dh, chunk, mb4, mb8);
  } else if (dtype == UBV_F16) {
    if (a.ol16) tile_fwd_launch_t<f16_t, true>(a, P, st, Dh, chunk, mb4, mb8);
    else tile_fwd_launch_t<f16_t, false>(a, P, st, Dh, chunk, mb4, mb8);
  } else {
    if (a.ol16) tile_fwd_launch_t<bf16_t, true>(a, P, st, Dh, chunk, mb4, mb8);
    else tile_fwd_launch_t<bf16_t, false>(a, P, st, Dh, chunk, mb4, mb8);
  }
}

template <typename T, bool OL16>
static void tile_bwd_launch_t(const LiftArgs& a, int P, bool bins, int tiles_x, int tiles, hipStream_t st, int Dh, int chunk,
                              int mb4, int mb8) {
  const dim3 grid(8 * chunk), blk(256);
#define UBV_TILE_BWD(PV, DHV, MB)                                                                                           \
  do {                                                                                                                      \
    if (bins) hipLaunchKernelGGL((lift_tile_bwd_query_kernel<PV, DHV, 1, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, MB, tile_centre()); \
    else hipLaunchKernelGGL((lift_tile_bwd_query_kernel<PV, DHV, 0, false, T, OL16>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, MB, tile_centre());     \
  } while (0)
  if (Dh == 16) { if (P == 4) UBV_TILE_BWD(4, 16, mb4); else UBV_TILE_BWD(8, 16, mb8); }
  else { if (P == 4) UBV_TILE_BWD(4, 32, mb4); else UBV_TILE_BWD(8, 32, mb8); }
#undef UBV_TILE_BWD
}

// bins: the points are binned here (the caller zeroed a.bin_cnt / a.ovf_n and launches no lift_bin_kernel)
void tile_bwd_query_launch(const LiftArgs& a, int P, bool bins, int tiles_x, int tiles, hipStream_t st, bool k1, int Dh,
                           int dtype) {
  const long units = tile_units(a, Dh, dtype == UBV_F32 ? 4 : 2);
  const int chunk = (int)((units + 7) / 8);
  const dim3 grid(8 * chunk), blk(256);
  static const int mb4 = tile_max_box("UBV_TILE_MAXBOX_BWD", 4, 256, 256), mb8 = tile_max_box("UBV_TILE_MAXBOX_BWD", 8, 256, 256);
  if (k1) {                                               // (the operator's backward always bins; f32, Dh = 32 only)
    if (P == 4) hipLaunchKernelGGL((lift_tile_bwd_query_kernel<4, 32, 1, true>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, mb4, tile_centre());
    else hipLaunchKernelGGL((lift_tile_bwd_query_kernel<8, 32, 1, true>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, mb8, tile_centre());
    return;
  }
  if (dtype == UBV_F32 && a.qrec && bins && Dh == 32) {     // query records (lift_bwd_value_q_kernel)
    if (P == 4) hipLaunchKernelGGL((lift_tile_bwd_query_kernel<4, 32, 2, false, float, false>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, mb4, tile_centre());
    else hipLaunchKernelGGL((lift_tile_bwd_query_kernel<8, 32, 2, false, float, false>), grid, blk, kTWinLds, st, a, chunk, tiles_x, tiles, mb8, tile_centre());
    return;
  }
  if (dtype == UBV_F32) tile_bwd_launch_t<float, false>(a, P, bins, tiles_x, tiles, st, Dh, chunk, mb4, mb8);
  else if (dtype == UBV_F16) {
    if (a.ol16) tile_bwd_launch_t<f16_t, true>(a, P, bins, tiles_x, tiles, st, Dh, chunk, mb4, mb8);
    else tile_bwd_launch_t<f16_t, false>(a, P, bins, tiles_x, tiles, st, Dh, chunk, mb4, mb8);
  } else {
    if (a.ol16) tile_bwd_launch_t<bf16_t, true>(a, P, bins, tiles_x, tiles, st, Dh, chunk, mb4, mb8);
    else tile_bwd_launch_t<bf16_t, false>(a, P, bins, tiles_x, tiles, st, Dh, chunk, mb4, mb8);
  }
}

}  // namespace ubv
