// Definitions shared by the translation units of the lifting kernels (bev_lift.hip, bev_lift_tile.hip): the
// argument blocks, query / tile decoding, the LDS-window helpers and the split-bf16 matrix-core helpers.
#pragma once
#include <limits.h>
#include <stdlib.h>

#include "ubv_common.h"

namespace ubv {

struct LiftArgs {
  const void* value; const void* offsets; long off_stride; const void* logits; long log_stride;
  int ol16;                                      // offsets / logits (and their gradients) are f32 (0)
                                                 // or the value's own 16-bit type (1)
  const float* ref; const uint8_t* vis0; const float* count;
  void* out;                                     // fwd
  const void* gout; float* gvalue; void* goff; long goff_stride; void* glog; long glog_stride;
  void* gvalue_lp;                               // final grad_value in the value's 16-bit type or null
  int B, Nc, fh, fw, H, Nq, Z, qw, qh, tiles_x, tiles_per_sample, total_tiles, chunk;
  unsigned mg_tps, mg_tx;                        // multiply-high reciprocals of tiles_per_sample / tiles_x (0: divide)
  // GRID backward: sampling points binned by owner tile
  int* bin_cnt;                                  // [B,H,tiles] points appended per tile (may exceed cap)
  float4* bins;                                  // [B,H,tiles,cap] (x_pix, y_pix, w/count, query index)
  int cap;                                       // bucket capacity
  int* ovf_n; float4* ovf_rec; int* ovf_tile; int ovf_cap;   // the appends that did not fit
  // QREC (round 6, f32 TILE instances): the query kernel stores every (query, head)'s points once — qpts [B, Nq, H, P, 3]
  // (x_pix, y_pix, softmax weight) — and appends the QUERY INDEX to the bucket of each owner tile one of its points
  // touches (`bins` then holds int32 entries, `cap` of them per tile); lift_bwd_value_q_kernel expands them
  float* qpts; int qrec;
  int cnt_words;                                 // > 0: the query-gradient kernel (first of the op) zeroes bin_cnt[0 .. cnt_words)
  int ovf_after;                                 // the overflow list is scattered AFTER the owner tiles stored (f32 grad_value)
  int* cam_list; int* cam_n;                     // per-camera compacted visible queries or null
  int ext_list;                                  // lists supplied by the caller (ubv_compact_visible)
  float* slab;                                   // CAMERA: per-chunk partial maps or null
  const void* vnat_hi; const void* vnat_lo;      // f32 matrix-core CAMERA plan: bf16 hi / lo copies of value (its layout)
  // MAPS backward (large per-camera maps): exact CSR buckets + work items (bev_lift_maps.inl)
  int* bin_cur;                                  // fill cursors per bucket
  int* bin_start;                                // [buckets + 1] first record of each bucket
  int* item_first;                               // [buckets + 1] first work item of each bucket
  int* item_bucket;                              // [items] bucket of each work item
  int* n_items;                                  // device scalar
  int max_items;
};

// The wave's index inside its block as a SCALAR: threadIdx.x >> 6 is wave-uniform, but the compiler cannot know, and
// everything decoded from it (tile, head, bucket, base addresses) would otherwise live in vector registers and be
// computed with vector instructions.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// base (wave-uniform) + element offset as a 32-bit BYTE offset (the host checks that one map / one sample's rows
// stay below 4 GiB): lets the backend emit global_load with a scalar base and a 32-bit vector offset
template <typename T> __device__ __forceinline__ const T* gather_ptr(const T* base, unsigned elem_off) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (unsigned)(elem_off * (unsigned)sizeof(T)));
}

// Decode (b, q, valid) of the query this lane works on in iteration `it`.
// n / d through the host-made reciprocal mg = floor(2^32 / d) + 1 (exact while n * d < 2^32; the host
// passes 0 otherwise): an integer division is ~40 instructions, this is one.
__device__ __forceinline__ int div_mg(int n, int d, unsigned mg) {
  return mg != 0u ? (int)__umulhi((unsigned)n, mg) : n / d;
}

// GRID backward, first kernel of the op: zero the tile counters + the overflow counter the bin kernel (next launch on
// the stream) appends through — a memset node less per op; every block takes a slice BEFORE any early exit.
__device__ __forceinline__ void lift_zero_counters(const LiftArgs& a) {
  if (a.cnt_words > 0)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.cnt_words; i += gridDim.x * blockDim.x) a.bin_cnt[i] = 0;
}

__device__ __forceinline__ bool lift_query(const LiftArgs& a, int item, int li, int& b, int& q) {
  b = div_mg(item, a.tiles_per_sample, a.mg_tps);
  const int tile = item - b * a.tiles_per_sample;
  if (a.qw > 0) {
    const int ty = div_mg(tile, a.tiles_x, a.mg_tx), tx = tile - ty * a.tiles_x;
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


// v + (v of the lane whose index differs in bit log2(M)): DPP quad permutes inside a quad (the
// compiler folds them into the add), a wave shuffle beyond.
template <int M>
__device__ __forceinline__ float add_xor(float v) {
  if constexpr (M == 1)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
  else if constexpr (M == 2)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
  else if constexpr (M == 4)
    // third step of a 1-2-4 butterfly: the lanes of a quad already agree, so "the other quad of my 8 lanes" is
    // as good as "lane ^ 4" — DPP row_half_mirror (lane i <- lane 7 - i) instead of a ds_bpermute through LDS
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
  else
    return v + __shfl_xor(v, M, 64);
}


struct TileArgs {
  int mode;            // 1 = GRID, 2 = CAMERA
  int tile_w, tile_h;  // pixels
  int tiles_x, tiles_y;
  int chunks, chunk_q; // CAMERA: query chunks per tile
  int total, chunk;    // tiles, blocks per XCD
  int waves;           // waves (= tiles) per block
  int cap;             // GRID: bucket capacity (records per tile)
  int balanced;        // CAMERA, matrix-core plans: the waves of a (sample, head) share ALL cameras' lists evenly (cam_share)
};


// ---- shared pieces of the owner-tile kernel -------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// 16-bit MFMA operand type used for pipeline element type T
template <typename T> struct mma_traits;
template <> struct mma_traits<bf16_t> {
  static constexpr bool kSplit = false;
  static __device__ __forceinline__ uint16_t enc(float v) { return (uint16_t)float_to_bf16_bits(v); }
  static __device__ __forceinline__ float dec(uint16_t b) { return bf16_bits_to_float(b); }
  static __device__ __forceinline__ f32x16_t mma(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct mma_traits<f16_t> {
  static constexpr bool kSplit = false;
  static __device__ __forceinline__ uint16_t enc(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
  static __device__ __forceinline__ float dec(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ f32x16_t mma(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct mma_traits<float> : mma_traits<bf16_t> {     // f32 data: bf16 hi + lo parts
  static constexpr bool kSplit = true;
};


// 8 f32 -> 4 dwords of bf16 hi pairs + 4 dwords of bf16 lo pairs
__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = cvt_pk_bf16(f[2 * k], f[2 * k + 1]);
    l[k] = cvt_pk_bf16(f[2 * k] - __uint_as_float(h[k] << 16), f[2 * k + 1] - __uint_as_float(h[k] & 0xffff0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// packed coefficient (hi | lo << 16) += c
__device__ __forceinline__ uint32_t coef_add(uint32_t e, float c) {
  const float s = (__uint_as_float(e << 16) + c) + __uint_as_float(e & 0xffff0000u);
  const uint32_t r = cvt_pk_bf16(s, s);
  const float nl = s - __uint_as_float(r << 16);
  const uint32_t r2 = cvt_pk_bf16(nl, nl);
  return (r & 0xffffu) | (r2 << 16);
}

// 8 consecutive packed coefficients -> MFMA fragment of the hi halves and of the lo halves
__device__ __forceinline__ void coef_frag(const uint32_t* p, uint4& hi, uint4& lo) {
  const uint4 e0 = *reinterpret_cast<const uint4*>(p), e1 = *reinterpret_cast<const uint4*>(p + 4);
  hi = make_uint4((e0.x & 0xffffu) | (e0.y << 16), (e0.z & 0xffffu) | (e0.w << 16),
                  (e1.x & 0xffffu) | (e1.y << 16), (e1.z & 0xffffu) | (e1.w << 16));
  lo = make_uint4((e0.x >> 16) | (e0.y & 0xffff0000u), (e0.z >> 16) | (e0.w & 0xffff0000u),
                  (e1.x >> 16) | (e1.y & 0xffff0000u), (e1.z >> 16) | (e1.w & 0xffff0000u));
}


typedef short i16x4_t __attribute__((ext_vector_type(4)));
// 8 rows (p, p + stride, ...) of the lane's column through two transposing reads of 4 rows each
__device__ __forceinline__ uint4 tr16_frag(const uint16_t* p, int stride) {
  typedef __attribute__((address_space(3))) i16x4_t lds_v4;
  const i16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const i16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 4 * stride));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return make_uint4(ua.x, ua.y, ub.x, ub.y);
}


// ---- LDS window of the value map (bev_lift_win.inl, bev_lift_tile.hip) ----
constexpr int kWin = 16;                                  // window side, pixels
constexpr int kWinRowB = 128 + 16;                        // bytes per window row
constexpr int kWinLds = kWin * kWin * kWinRowB;

struct WinGeom { int b, tile, hg, wx0, wy0; };

// (item, head group) of this block.
template <int HG>
__device__ __forceinline__ bool win_decode(const LiftArgs& a, int chunk, WinGeom& g) {
  const int NG = a.H / HG;
  const int v = xcd_remap(blockIdx.x, chunk);
  const int item = v / NG;
  if (item >= a.total_tiles) return false;
  g.hg = v - item * NG;
  g.b = div_mg(item, a.tiles_per_sample, a.mg_tps);
  g.tile = item;
  return true;
}

// Window origin = the block's smallest corner column / row (the sampling pattern of a head leans one way: the
// window follows it instead of sitting centred on the tile), clamped into the map.  Ends with a barrier.
__device__ __forceinline__ void win_origin(const LiftArgs& a, int minx, int miny, WinGeom& g) {
  __shared__ int red[4][2];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    minx = min(minx, __shfl_xor(minx, m, 64));
    miny = min(miny, __shfl_xor(miny, m, 64));
  }
  const int wv = wave_in_block();
  if ((threadIdx.x & 63) == 0) { red[wv][0] = minx; red[wv][1] = miny; }
  __syncthreads();
  const int bx = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
  const int by = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
  g.wx0 = min(max(bx, 0), max(a.fw - kWin, 0));           // (no valid point at all: INT_MAX -> the last window)
  g.wy0 = min(max(by, 0), max(a.fh - kWin, 0));
}

// Copies the window into LDS: one row (pixel) per thread, 128 bytes = the HG heads' channels.  Ends with a barrier.
template <typename T, int DH, int HG>
__device__ __forceinline__ void win_load(const LiftArgs& a, const WinGeom& g, unsigned char* __restrict__ win) {
  static_assert(HG * DH * sizeof(T) == 128, "a window row is one 128-byte line");
  const int t = threadIdx.x, wy = t >> 4, wx = t & 15;
  const int py = min(g.wy0 + wy, a.fh - 1), px = min(g.wx0 + wx, a.fw - 1);
  const long row = (long)a.H * DH;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(
      (const T*)a.value + ((long)g.b * a.fh * a.fw + (long)py * a.fw + px) * row + g.hg * HG * DH);
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = src[i];
  uint4* dst = reinterpret_cast<uint4*>(win + t * kWinRowB);
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = v[i];
  __syncthreads();
}

// Window row of the corner at (column xc, row yc), or -1 when it lies outside the window.
__device__ __forceinline__ int win_row(int xc, int yc, const WinGeom& g) {
  const int dx = xc - g.wx0, dy = yc - g.wy0;
  return ((unsigned)dx < (unsigned)kWin && (unsigned)dy < (unsigned)kWin) ? dy * kWin + dx : -1;
}


// ---- TILE plan (bev_lift_tile.hip): f32 GRID instances, one lane per query ----
bool tile_ok(const LiftArgs& a, int Dh, int P, int dtype, bool bwd = false);
void tile_fwd_launch(const LiftArgs& a, int P, hipStream_t st, bool k1 = false, int Dh = 32, int dtype = 0);
void tile_bwd_query_launch(const LiftArgs& a, int P, bool bins, int tiles_x, int tiles, hipStream_t st, bool k1 = false, int Dh = 32,
                           int dtype = 0);

}  // namespace ubv
