// TILE plan: f32 GRID lifting with ONE LANE PER QUERY (its own translation unit; the argument blocks and the LDS-window
// helpers win_decode / win_origin / win_load / win_row come from bev_lift_core.h).
//
// The shared-footprint and LDS-window kernels put the Dh channels of a (query, head) on LP = 8 adjacent lanes: every
// corner is one coalesced 128-byte gather, but each of its 4 FMAs per lane is paid with a broadcast of the corner's
// index and coefficient and (backward) a 3-step lane reduction of the dot product — ~1.5 wave instructions per
// (query, head, corner), and a wave64 VALU instruction occupies its SIMD for 4 cycles.  Once the corners come from an
// LDS window there is nothing left to coalesce, so here a lane IS a query:
//   block  = (8x8 query tile, one head): 4 waves, each takes P / 4 of the head's sampling points for all 64 queries;
//   window = 16x16 pixels of the head's value slice (128 B per pixel) copied into LDS once per block;
//   corner = 8 ds_read_b128 + 32 FMAs per lane, no cross-lane traffic — 0.63 wave instructions per corner — and all
//            arithmetic stays plain f32 (no operand splitting);
//   a lane whose corner falls outside the window fetches that corner's row from global memory itself (the window is
//   a cache, never an approximation), the other lanes of the wave keep the LDS path.
// The four waves' partial sums meet in LDS (the window's bytes, after a barrier).
//
// Backward (lift_tile_bwd_query_kernel, lift_tile_bwd_value_kernel): see below.

#include "bev_lift_core.h"

namespace ubv {

// LDS window of the TILE kernels: pixel (dx, dy) at dy * kTWinRow + dx * kWinRowB.  144-byte pixels put the 16 columns
// of one row on 16 distinct bank quads (9 dx mod 16); the 64 bytes of padding per window row add 4 dy, so that points
// stepping along either axis or a diagonal — the lanes of a ds_read_b128 group are 4 queries x 4 points — stay on
// distinct quads: quad = (9 dx + 4 dy) mod 16.
constexpr int kTWinRow = kWin * kWinRowB + 64;
constexpr int kTWinLds = kWin * kTWinRow;
constexpr int kRecUnit(int P) { return P * 3 * 64; }          // floats per (sample, tile, head) unit

// min / max over the wave: DPP inside the rows of 16 lanes (quad permutes, row_half_mirror, row_mirror), the four
// rows through scalar reads
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

// The sampling points of this lane — lane = (query li of the tile, slot pp of 4), points pp + 4 j — as pixel
// coordinates and softmax weights (0 for a query outside the grid), and the pixel box of the block's live corners
// (a corner is live when its bilinear weight is non-zero).  Softmax over the 4 lanes of the query with DPP quad
// permutes.  One barrier.
// per-slot reductions over the lanes of a row of 16 that share lane & 3 (DPP row rotations by 4 and 8)
__device__ __forceinline__ int row4_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
  return min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
}

// Box of the corners a set of points touches
struct TileBox {
  int x0, y0, x1, y1;
  __device__ __forceinline__ void init() { x0 = INT_MAX; y0 = INT_MAX; x1 = -1; y1 = -1; }
  __device__ __forceinline__ void add(int x, int y) { x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y); }
};

// The sampling points of this lane — lane = (query li of the tile, slot pp of 4), points pp + 4 j — as pixel
// coordinates and softmax weights (0 for a query outside the grid), and the pixel box of the block's corners: the
// corners with a non-zero bilinear weight (forward), every corner inside the map (BWD: a corner of weight 0 still has
// a derivative).  Softmax over the 4 lanes of the query with DPP quad permutes.  One barrier.
// BWD also reduces, per point p of the head, the box of the corners with a non-zero COEFFICIENT into pbox[p] (LDS,
// initialised here): what the owner tiles of the value gradient look at.
template <int P, bool BWD>
__device__ __forceinline__ int4 tile_points(const LiftArgs& a, long bq, bool valid, int h, int pp, int wv, int lane,
                                            float (&rx)[P / 4], float (&ry)[P / 4], float (&rw)[P / 4],
                                            int (*pbox)[4]) {
  constexpr int PW = P / 4;
  __shared__ int4 wbox[4];
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float* __restrict__ offp = (const float*)a.offsets + bq * a.off_stride + h * 2 * P;
  const float* __restrict__ lgp = (const float*)a.logits + bq * a.log_stride + h * P;
  const float* __restrict__ rp = a.ref + bq * a.Z * 2;
  float2 off[PW], ref[PW];
  float lg[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = pp + 4 * j;
    off[j] = *reinterpret_cast<const float2*>(offp + 2 * p);
    ref[j] = *reinterpret_cast<const float2*>(rp + (p % a.Z) * 2);
    lg[j] = lgp[p];
  }
  if (BWD && threadIdx.x < P) { pbox[threadIdx.x][0] = INT_MAX; pbox[threadIdx.x][1] = INT_MAX; pbox[threadIdx.x][2] = -1; pbox[threadIdx.x][3] = -1; }
  TileBox wb, cb[PW];
  wb.init();
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const float lx = ref[j].x + off[j].x / fwf, ly = ref[j].y + off[j].y / fhf;
    rx[j] = lx * fwf - 0.5f; ry[j] = ly * fhf - 0.5f;
    const Footprint f = footprint_px(rx[j], ry[j], a.fh, a.fw);
    cb[j].init();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (valid && (BWD ? f.m[k] : f.w[k]) != 0.0f) wb.add(f.xc[k & 1], f.yc[k >> 1]);
      if (BWD && valid && f.w[k] != 0.0f) cb[j].add(f.xc[k & 1], f.yc[k >> 1]);
    }
  }
  const int minx = wave_min_i32(wb.x0), miny = wave_min_i32(wb.y0), maxx = wave_max_i32(wb.x1), maxy = wave_max_i32(wb.y1);
  if (lane == 0) wbox[wv] = make_int4(minx, miny, maxx, maxy);
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
  if constexpr (BWD) {
    // (a weight that underflowed to 0 touches nothing)
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const bool on = rw[j] != 0.0f;
      const int x0 = row4_min_i32(on ? cb[j].x0 : INT_MAX), y0 = row4_min_i32(on ? cb[j].y0 : INT_MAX);
      const int x1 = -row4_min_i32(on ? -cb[j].x1 : 1), y1 = -row4_min_i32(on ? -cb[j].y1 : 1);
      if ((lane & 12) == 0) {                       // lanes 0..3 of each row: one per slot
        int* pb = pbox[pp + 4 * j];
        atomicMin(pb, x0); atomicMin(pb + 1, y0); atomicMax(pb + 2, x1); atomicMax(pb + 3, y1);
      }
    }
  }
  const int4 b0 = wbox[0], b1 = wbox[1], b2 = wbox[2], b3 = wbox[3];
  return make_int4(min(min(b0.x, b1.x), min(b2.x, b3.x)), min(min(b0.y, b1.y), min(b2.y, b3.y)),
                   max(max(b0.z, b1.z), max(b2.z, b3.z)), max(max(b0.w, b1.w), max(b2.w, b3.w)));
}

// acc[0..32) += c * (32 consecutive floats at p): LDS or global, 16-byte pieces
__device__ __forceinline__ void tile_axpy32(const float* __restrict__ p, float c, float (&acc)[32]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    acc[4 * i] = fmaf(c, v.x, acc[4 * i]);
    acc[4 * i + 1] = fmaf(c, v.y, acc[4 * i + 1]);
    acc[4 * i + 2] = fmaf(c, v.z, acc[4 * i + 2]);
    acc[4 * i + 3] = fmaf(c, v.w, acc[4 * i + 3]);
  }
}
__device__ __forceinline__ float tile_dot32(const float* __restrict__ p, const float (&g)[32]) {
  float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 v = reinterpret_cast<const float4*>(p)[i];
    d0 = fmaf(g[4 * i], v.x, d0);
    d1 = fmaf(g[4 * i + 1], v.y, d1);
    d0 = fmaf(g[4 * i + 2], v.z, d0);
    d1 = fmaf(g[4 * i + 3], v.w, d1);
  }
  return d0 + d1;
}

// Window of a unit from its box: origin (clamped into the map like win_origin) and the rows / columns worth loading.
struct TileWin { int rows, cols; };
__device__ __forceinline__ TileWin tile_window(const LiftArgs& a, const int4 bb, WinGeom& g) {
  g.wx0 = min(max(bb.x, 0), max(a.fw - kWin, 0));
  g.wy0 = min(max(bb.y, 0), max(a.fh - kWin, 0));
  TileWin t;
  t.cols = min(max(bb.z - g.wx0 + 1, 0), min(kWin, a.fw));
  t.rows = min(max(bb.w - g.wy0 + 1, 0), min(kWin, a.fh));
  return t;
}

// Fills rows [0, rows) x columns [0, cols) of the window: 8 lanes per pixel (one 128-byte line per 8 lanes), 32
// pixels = 2 window rows per pass, all passes' loads in flight before the first LDS store.  Ends with a barrier.
__device__ __forceinline__ void tile_fill(const LiftArgs& a, const WinGeom& g, const TileWin t, int h,
                                          unsigned char* __restrict__ win) {
  const int tid = threadIdx.x, piece = tid & 7, pxl = tid >> 3;
  const int dx = pxl & 15, dyl = pxl >> 4;
  const int cdx = min(dx, max(t.cols - 1, 0));            // columns past the box re-read its last one (same line: free)
  const int rowi = a.H * 32;
  const float* vb = (const float*)a.value + (long)g.b * a.fh * a.fw * rowi + h * 32 + piece * 4;
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (2 * i < t.rows) {                                  // block-uniform
      const int dy = min(2 * i + dyl, t.rows - 1);
      x = *reinterpret_cast<const uint4*>(gather_ptr(vb, (unsigned)(((g.wy0 + dy) * a.fw + g.wx0 + cdx) * rowi)));
    }
    v[i] = x;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (2 * i < t.rows)
      *reinterpret_cast<uint4*>(win + (2 * i + dyl) * kTWinRow + dx * kWinRowB + piece * 16) = v[i];
  __syncthreads();
}

// LDS byte offset of a corner's pixel, or -1 when it lies outside the loaded part of the window.
__device__ __forceinline__ int tile_row(int xc, int yc, const WinGeom& g, const TileWin t) {
  const int dx = xc - g.wx0, dy = yc - g.wy0;
  return ((unsigned)dx < (unsigned)t.cols && (unsigned)dy < (unsigned)t.rows) ? dy * kTWinRow + dx * kWinRowB : -1;
}

// ------------------------------------------------------------------------------------------------
// Forward.  Block = unit (8x8 query tile, head), wave = 16 queries, lane = (query, slot pp of 4): the lane takes
// points pp, pp + 4 (P = 8) of its query; the 4 lanes of a query add their rows with DPP quad permutes and store 32
// bytes each.  Two barriers (block box, window fill).
template <int P>
__global__ __launch_bounds__(256) void lift_tile_fwd_kernel(const LiftArgs a, int chunk, int abl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int PW = P / 4;
  WinGeom g;
  if (!win_decode<1>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int h = g.hg;
  const int rowi = a.H * 32;
  const int li = wv * 16 + (lane >> 2), pp = lane & 3;
  float rx[PW], ry[PW], rw[PW];
  int b, q;
  const bool valid = lift_query(a, g.tile, li, b, q);
  if (!valid) q = 0;
  const long bq = (long)b * a.Nq + q;
  const int4 bb = tile_points<P, false>(a, bq, valid, h, pp, wv, lane, rx, ry, rw, nullptr);
  const TileWin tw = tile_window(a, bb, g);
  if (!(abl & 1)) tile_fill(a, g, tw, h, win); else __syncthreads();
  const float* vb = (const float*)a.value + (long)g.b * a.fh * a.fw * rowi + h * 32;      // wave-uniform

  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
  if (!(abl & 2))
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const Footprint f = footprint_px(rx[j], ry[j], a.fh, a.fw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = rw[j] * f.w[k];
      const int wr = tile_row(f.xc[k & 1], f.yc[k >> 1], g, tw);
      if (wr >= 0) tile_axpy32(reinterpret_cast<const float*>(win + (unsigned)wr), c, acc);
      else if (c != 0.0f) tile_axpy32(gather_ptr(vb, (unsigned)(f.idx[k] * rowi)), c, acc);
    }
  }
  if (abl & 4) { if (acc[0] == 123.456f) ((float*)a.out)[0] = acc[1]; return; }
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = add_xor<2>(add_xor<1>(acc[i]));
  if (valid) {
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = pp == 0 ? acc[i] : pp == 1 ? acc[8 + i] : pp == 2 ? acc[16 + i] : acc[24 + i];
    float4* dst = reinterpret_cast<float4*>((float*)a.out + bq * rowi + h * 32 + pp * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, query side: d(offsets), d(logits) — the forward's block with the query's grad_out row in registers and a
// dot product per corner instead of an axpy; no cross-lane traffic except the softmax sum over the quad.  The kernel
// also leaves what the value side (lift_tile_bwd_value_kernel) needs: one (x_pix, y_pix, weight) record per sampling
// point, [unit][point][x | y | w][64 queries], and per (unit, point) the box of the pixels that receive a non-zero
// coefficient.
template <int P>
__global__ __launch_bounds__(256) void lift_tile_bwd_query_kernel(const LiftArgs a, int chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  __shared__ int pbox[P][4];
  constexpr int PW = P / 4;
  WinGeom g;
  if (!win_decode<1>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int h = g.hg;
  const int rowi = a.H * 32;
  const int li = wv * 16 + (lane >> 2), pp = lane & 3;
  int b, q;
  const bool valid = lift_query(a, g.tile, li, b, q);
  if (!valid) q = 0;
  const long bq = (long)b * a.Nq + q;
  float go[32];
  {
    const float4* gp = reinterpret_cast<const float4*>((const float*)a.gout + bq * rowi + h * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = gp[i];
      go[4 * i] = v.x; go[4 * i + 1] = v.y; go[4 * i + 2] = v.z; go[4 * i + 3] = v.w;
    }
  }
  float rx[PW], ry[PW], rw[PW];
  const int4 bb = tile_points<P, true>(a, bq, valid, h, pp, wv, lane, rx, ry, rw, pbox);
  const TileWin tw = tile_window(a, bb, g);
  tile_fill(a, g, tw, h, win);
  const float* vb = (const float*)a.value + (long)g.b * a.fh * a.fw * rowi + h * 32;      // wave-uniform
  const long unit = (long)g.tile * a.H + h;
  if (threadIdx.x < P)                                      // (the fill's barrier ordered the atomics before this read)
    a.tbox[unit * P + threadIdx.x] = make_int4(pbox[threadIdx.x][0], pbox[threadIdx.x][1], pbox[threadIdx.x][2], pbox[threadIdx.x][3]);
  float* __restrict__ rec = a.trec + unit * kRecUnit(P) + li;

  float gw[PW], gx[PW], gy[PW];
  float sp = 0.0f;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int p = pp + 4 * j;
    rec[p * 192] = rx[j]; rec[p * 192 + 64] = ry[j]; rec[p * 192 + 128] = rw[j];
    const Footprint f = footprint_px(rx[j], ry[j], a.fh, a.fw);
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int wr = tile_row(f.xc[k & 1], f.yc[k >> 1], g, tw);
      d[k] = 0.0f;
      if (wr >= 0) d[k] = tile_dot32(reinterpret_cast<const float*>(win + (unsigned)wr), go);
      else if (valid && f.m[k] != 0.0f) d[k] = tile_dot32(gather_ptr(vb, (unsigned)(f.idx[k] * rowi)), go);
      d[k] *= f.m[k];
    }
    const float hx = 1.0f - f.lx, hy = 1.0f - f.ly;
    gw[j] = hy * hx * d[0] + hy * f.lx * d[1] + f.ly * hx * d[2] + f.ly * f.lx * d[3];
    gx[j] = (d[1] - d[0]) * hy + (d[3] - d[2]) * f.ly;
    gy[j] = (d[2] - d[0]) * hx + (d[3] - d[1]) * f.lx;
    sp = fmaf(rw[j], gw[j], sp);
  }
  sp = add_xor<2>(add_xor<1>(sp));
  if (valid) {
    const float fwf = (float)a.fw, fhf = (float)a.fh;
    float* __restrict__ glog = (float*)a.glog + bq * a.glog_stride + h * P;
    float* __restrict__ goff = (float*)a.goff + bq * a.goff_stride + h * 2 * P;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int p = pp + 4 * j;
      glog[p] = rw[j] * (gw[j] - sp);
      // d loc = w * g * W; d off = d loc / W (the reference's rounding)
      *reinterpret_cast<float2*>(goff + 2 * p) = make_float2((rw[j] * gx[j] * fwf) / fwf, (rw[j] * gy[j] * fhf) / fhf);
    }
  }
}

// f32 data, Dh = 32, one map per sample, grid-tiled queries, no visibility / count: the BEV self-attention and
// SCA-pts instances.  UBV_LIFT_TILE=0 switches the plan off (A/B runs against the window / gather kernels).
bool tile_ok(const LiftArgs& a, int Dh, int P, int dtype) {
  static const int env = getenv("UBV_LIFT_TILE") ? atoi(getenv("UBV_LIFT_TILE")) : 1;
  return env != 0 && dtype == UBV_F32 && Dh == 32 && (P == 4 || P == 8) && a.ol16 == 0 && a.Nc == 1 && a.qw > 0 &&
         a.vis0 == nullptr && a.count == nullptr && a.fh >= 1 && a.fw >= 1;
}

// records + boxes of the backward: [units][P][3][64] floats, [units][P] int4
static size_t tile_units(const LiftArgs& a) { return (size_t)a.B * (size_t)(((a.qw + 7) / 8) * ((a.qh + 7) / 8)) * a.H; }
static size_t tile_rec_bytes(const LiftArgs& a, int P) { return ((tile_units(a) * kRecUnit(P) * sizeof(float)) + 255) & ~(size_t)255; }
size_t tile_bwd_ws_bytes(const LiftArgs& a, int P) {
  return tile_rec_bytes(a, P) + ((tile_units(a) * P * sizeof(int4) + 255) & ~(size_t)255);
}
void tile_bwd_query_launch(LiftArgs a, int P, void* ws, hipStream_t st) {
  a.trec = (float*)ws;
  a.tbox = (int4*)((char*)ws + tile_rec_bytes(a, P));
  const long units = (long)a.total_tiles * a.H;
  const int chunk = (int)((units + 7) / 8);
  if (P == 4) hipLaunchKernelGGL((lift_tile_bwd_query_kernel<4>), dim3(8 * chunk), dim3(256), kTWinLds, st, a, chunk);
  else hipLaunchKernelGGL((lift_tile_bwd_query_kernel<8>), dim3(8 * chunk), dim3(256), kTWinLds, st, a, chunk);
}

void tile_fwd_launch(const LiftArgs& a, int P, hipStream_t st) {
  const long units = (long)a.total_tiles * a.H;
  const int chunk = (int)((units + 7) / 8);
  const int abl = getenv("UBV_TILE_ABL") ? atoi(getenv("UBV_TILE_ABL")) : 0;
  if (P == 4) hipLaunchKernelGGL((lift_tile_fwd_kernel<4>), dim3(8 * chunk), dim3(256), kTWinLds, st, a, chunk, abl);
  else hipLaunchKernelGGL((lift_tile_fwd_kernel<8>), dim3(8 * chunk), dim3(256), kTWinLds, st, a, chunk, abl);
}

}  // namespace ubv
