// CAMERA plan on the matrix cores — included by bev_lift.hip inside namespace ubv.
//
// For small per-camera maps (the 8x22 maps of 256x704 inputs) the whole value slice of a (sample,
// camera, head) is a ~200 x 32 matrix, so the bilinear gather of 64 queries
//        out[q, :] = sum_p w_p sum_k c_pk V[pix_pk, :]
// is the dense product  out^T[32 ch, 64 q] = V^T[32, K] . A^T[K, 64]  with a sparse coefficient
// matrix A[q][pix] (<= 32 non-zeros per row).  One lane per query builds its own row of A in LDS
// (read-modify-write: points of one query may share a pixel; rows are lane-private and the LDS
// operations of a wave execute in order), then v_mfma_f32_32x32x16 runs over the 16-pixel K-blocks
// the wave touched.  Softmax, sampling locations and footprints are computed once per (query, head,
// camera) instead of once per channel lane, and no gather goes through the L1 / texture path: the
// gather kernel issued 4*P*Nc_visible 16-byte loads per lane and sat at 7-8 % of the HBM roofline,
// bound by the texture addresser and its 4x redundant footprint arithmetic.
//
// K index of a pixel: the map is padded by a one-pixel ZERO border and stored column-major,
//        k(x, y) = (x + 1) * (fh + 1) + (y + 1),     x in [-1, fw], y in [-1, fh]
// (the bottom border of column x shares its slot with the top border of column x + 1).  With the
// border every corner of every in-range point is an ordinary entry: no clamping, no per-corner
// validity masks (the reference's "zero padding per corner" is the zero rows of V), and a point
// outside the map is moved onto the border, where all of its weight meets zeros.  Column-major
// because an 8x8 tile of BEV queries projects onto a few image columns but — through the 4 pillar
// heights — onto every row: column-major K-blocks are the ones a wave can skip.
//
// V^T fragments (MFMA A operand: row = channel, 8 consecutive k per lane) come from a
// fragment-ordered, padded copy of `value` written by value_frags_kernel (1.4 MB at bs = 2): one
// coalesced 16-byte load per lane per K-block.

constexpr int kCamKbMax = 15;        // K-blocks of 16 padded pixels: (fw + 2) * (fh + 1) + 1 <= 240

struct CamArgs {
  const void* vfrag;    // [B*Nc][H][KB][64 lanes][8] value fragments (f32 data: the bf16 hi halves)
  const void* vfrag_lo; // f32 data: the bf16 lo halves, same layout
  int KB;               // K-blocks of the fragment buffer (14 or 15)
  int witems;           // (sample, tile, head) wave items
  int chunk;            // blocks per XCD
  int fh1;              // fh + 1: column stride of the padded map
  unsigned mg;          // ceil(2^16 / fh1): kk / fh1 == (kk * mg) >> 16 for kk < 256, fh1 <= 14 (cam_pixel_mg)
};

__host__ __device__ inline int cam_kpad(int fh, int fw) { return (fw + 2) * (fh + 1) + 1; }

// padded K index -> pixel of the map, or false on the border
__device__ __forceinline__ bool cam_pixel(int kk, int fh, int fw, int& pix) {
  const int c = kk / (fh + 1), r = kk - c * (fh + 1);
  pix = (r - 1) * fw + (c - 1);
  return r >= 1 && c >= 1 && c <= fw;
}

// the same with the division by a host-made reciprocal (kk < 256, fh + 1 <= 14: exact): the integer division was
// ~25 instructions per call, 112 calls in the epilogue of the value-gradient kernels and 4 per pass of the
// query-gradient kernels
__device__ __forceinline__ bool cam_pixel_mg(int kk, int fh1, unsigned mg, int fw, int& pix) {
  const int c = (int)(((unsigned)kk * mg) >> 16), r = kk - c * fh1;
  pix = (r - 1) * fw + (c - 1);
  return r >= 1 && c >= 1 && c <= fw;
}

template <typename T>
__global__ __launch_bounds__(256) void value_frags_kernel(const T* __restrict__ value, T* __restrict__ vf,
                                                          int BNc, int S, int H, int fh, int fw, int KB) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)BNc * H * KB * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long r = t >> 6;
  const int kb = (int)(r % KB); r /= KB;
  const int h = (int)(r % H);
  const long bnc = r / H;
  const int m = lane & 31, kg = lane >> 5;
  T o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int pix;
    const bool real = cam_pixel(kb * 16 + kg * 8 + j, fh, fw, pix);
    o[j] = real ? value[((bnc * S + pix) * H + h) * 32 + m] : elem<T>::from_float(0.0f);
  }
  *reinterpret_cast<uint4*>(vf + t * 8) = *reinterpret_cast<const uint4*>(o);
}

// One rounding instruction per coefficient (the software round-to-nearest-even of
// float_to_bf16_bits is five).
template <typename T> struct cam_cvt;
template <> struct cam_cvt<bf16_t> {
  static __device__ __forceinline__ float dec(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
  static __device__ __forceinline__ uint16_t enc(float v) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(v));
    return (uint16_t)r;
  }
};
template <> struct cam_cvt<f16_t> {
  static __device__ __forceinline__ float dec(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ uint16_t enc(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }
};

// Footprint of one sampling point in the padded map: K index of its top-left corner (the others sit
// at +1, +fh1, +fh1+1) and the fractional parts.  A point outside the map goes to (-1, -1): K index
// 0, lx = ly = 0, i.e. its whole weight on a border slot.
struct PadFoot { int k0; float lx, ly; };
__device__ __forceinline__ PadFoot pad_foot(float loc_x, float loc_y, float fwf, float fhf, int fw, int fh,
                                            int fh1) {
  const float x = loc_x * fwf - 0.5f, y = loc_y * fhf - 0.5f;
  const float xf = floorf(x), yf = floorf(y);
  const int xi = (int)xf, yi = (int)yf;              // saturating; NaN -> 0
  // -1 < x < fw (x = -1 itself would put all weight on the border: same result); the float compares
  // reject NaN and the low side, the integer ones the high side
  const bool inside = (x > -1.0f) & (y > -1.0f) & (xi < fw) & (yi < fh);
  PadFoot f;
  f.k0 = inside ? __mul24(xi + 1, fh1) + yi + 1 : 0;
  f.lx = inside ? x - xf : 0.0f;
  f.ly = inside ? y - yf : 0.0f;
  return f;
}

// Work decomposition: a wave is HALF an 8x8 tile of BEV queries (4 rows = 32 queries) for one head;
// lanes l and l + 32 share a query and split its P = 8 points (the MFMA's N is 32 queries).  What
// that buys is LDS: a 32-row coefficient matrix is 14.6 KB, so 10 waves fit a CU — with 64 rows it
// was 5, and a lone wave per SIMD issues one dependent VALU instruction per ~4 cycles and hides none
// of its LDS / L2 latency (measured: 48 % of wave cycles waiting).  The two point groups update the
// shared row one after the other (two exec-masked phases of 4 read-modify-write rounds).
//
// KBT: K-blocks of the fragment buffer (a compile-time count keeps the fragment loads and the MFMA
// chain free of branches).  Every K-block is multiplied: the MFMA pipe has slack here, skipping
// untouched blocks would cost more VALU / SALU work in masks and ballots than the MFMAs it saves.
__device__ __forceinline__ bool cam_item(const LiftArgs& a, int witem, int lane, int& h, int& item,
                                         int& j, int& b, int& q) {
  h = witem % a.H;
  const int r = witem / a.H;
  item = r >> 1;
  j = (r & 1) * 32 + (lane & 31);                      // tile-local query
  return lift_query(a, item, j, b, q);
}

template <typename T, int P, bool OL16, int KBT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void lift_cam_fwd_kernel(const LiftArgs a, const CamArgs c) {
  static_assert(P == 8, "two groups of 4 points");
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  using M = mma_traits<T>;
  using CV = cam_cvt<T>;
  constexpr int ASTR = KBT * 16 + 4;
  constexpr int PG = P / 2;
  const int lane = threadIdx.x, g = lane >> 5;
  const int witem = xcd_remap(blockIdx.x, c.chunk);     // (sample, tile, half tile, head), head fastest
  if (witem >= c.witems) return;
  uint16_t* A = lds_all;
  for (int i = lane; i < 32 * ASTR / 4; i += 64) reinterpret_cast<uint2*>(A)[i] = make_uint2(0u, 0u);
  int h, item, j, b, q;
  const bool valid = cam_item(a, witem, lane, h, item, j, b, q);
  if (!valid) q = 0;
  uint16_t* arow = A + (lane & 31) * ASTR;
  const long bq = (long)b * a.Nq + q;
  float lg[P], w[P], off[2 * PG];
  load_ol<T, P>(a.logits, bq * a.log_stride + h * P, OL16, lg);
  load_ol<T, 2 * PG>(a.offsets, bq * a.off_stride + h * 2 * P + g * 2 * PG, OL16, off);
  // visibility of this query in every camera, all loads in flight together
  unsigned vismask = 0u;
  if (a.vis0 == nullptr) vismask = valid ? ~0u : 0u;
  else
    for (int cam = 0; cam < a.Nc; ++cam)
      vismask |= (valid && a.vis0[(long)cam * a.Nq + q] != 0) ? (1u << cam) : 0u;
  softmax_row<P, true>(lg, w);
  float wg[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) wg[i] = g ? w[PG + i] : w[i];
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  {
    const float inv_fw = __builtin_amdgcn_rcpf(fwf), inv_fh = __builtin_amdgcn_rcpf(fhf);
#pragma unroll
    for (int i = 0; i < PG; ++i) { off[2 * i] *= inv_fw; off[2 * i + 1] *= inv_fh; }
  }
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const uint4* __restrict__ vfb = reinterpret_cast<const uint4*>(c.vfrag);
  const uint16_t* brow = arow + g * 8;
  const int fh1 = c.fh1;
  // anchor of flat point p is p % Z; this lane's points are g*4 + i
  int za[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) za[i] = (g * PG + i) % a.Z;

  for (int cam = 0; cam < a.Nc; ++cam) {
    const bool v = (vismask >> cam) & 1u;
    if (__ballot(v) == 0ull) continue;
    // V^T fragments of (b, cam, h): every K-block in flight before the footprint arithmetic
    const uint4* __restrict__ vf = vfb + ((((long)b * a.Nc + cam) * a.H + h) * KBT) * 64 + lane;
    uint4 av[KBT];
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) av[kb] = vf[(long)kb * 64];
    const float* __restrict__ rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
    int k0s[PG];
    float cf[PG][4];
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const float2 r = *reinterpret_cast<const float2*>(rp + za[i] * 2);
      const PadFoot f = pad_foot(r.x + off[2 * i], r.y + off[2 * i + 1], fwf, fhf, a.fw, a.fh, fh1);
      const float wp = v ? wg[i] : 0.0f;                    // invisible here: adds zeros
      const float wl = wp * f.lx, wh = wp - wl;             // w*lx, w*(1 - lx)
      cf[i][3] = wl * f.ly; cf[i][2] = wl - cf[i][3];       // column x+1: ly, 1 - ly
      cf[i][1] = wh * f.ly; cf[i][0] = wh - cf[i][1];       // column x
      k0s[i] = f.k0;
    }
    // the two point groups of a query share its row: one group at a time.  Within a point the 4
    // corners are 4 different slots (fh1 >= 2): read them all, then write them all; an earlier
    // point's writes to the same slots are seen (one wave's LDS operations run in order).
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (g == ph) {
#pragma unroll
        for (int i = 0; i < PG; ++i) {
          uint16_t* e = arow + k0s[i];
          const uint16_t u00 = e[0], u10 = e[1], u01 = e[fh1], u11 = e[fh1 + 1];
          e[0] = CV::enc(CV::dec(u00) + cf[i][0]);
          e[1] = CV::enc(CV::dec(u10) + cf[i][1]);
          e[fh1] = CV::enc(CV::dec(u01) + cf[i][2]);
          e[fh1 + 1] = CV::enc(CV::dec(u11) + cf[i][3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
      const uint2 bl = *reinterpret_cast<const uint2*>(brow + kb * 16);
      const uint2 bh = *reinterpret_cast<const uint2*>(brow + kb * 16 + 4);
      acc = M::mma(av[kb], make_uint4(bl.x, bl.y, bh.x, bh.y), acc);
    }
    __builtin_amdgcn_wave_barrier();
    // every lane zeroes the slots it wrote
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      uint16_t* e = arow + k0s[i];
      e[0] = 0; e[1] = 0; e[fh1] = 0; e[fh1 + 1] = 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // D^T: row = channel (r & 3) + 8 (r >> 2) + 4 g, column = this lane's query
  if (valid) {
    const float inv = (a.count != nullptr) ? __builtin_amdgcn_rcpf(a.count[bq]) : 1.0f;
    T* o = (T*)a.out + bq * ((long)a.H * 32) + h * 32 + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float vv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vv[i] = acc[4 * k + i] * inv;
      vec_io<T, 4>::store(o + 8 * k, vv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, query side, same plan and the same half-tile waves: the dot products
// grad_out[q, :] . V[pix, :] for EVERY slot of the padded map are one product
//        dA[K slots, 32 q] = V[K, 32 ch] . G^T[32, 32]
// (both operands straight from their natural layouts: 16 contiguous bytes per lane; border rows of V
// are zeros, so the dots of border corners vanish like the reference's per-corner masks), written
// through LDS so that each lane can pick the 16 slots its 4 points touch.  dA stays f32
// (differences of neighbouring dots make the offset gradient; rounding them to 16 bits would cost
// 4x the noise the 16-bit values already carry), so the map is walked in passes of 4 row blocks
// (128 slots, 16.6 KB per wave) that start every 96 slots: a point belongs to the pass its first
// corner lies in, and its last corner (<= fh + 2 <= 15 slots further) is still inside that pass.
// The gradients are linear in the dots:
//   gw += bw_k d, gx += sx_k d, gy += sy_k d   with (sx, sy) = (-hy,-hx), (-ly,hx), (hy,-lx), (ly,lx)
// for the corners (x,y), (x,y+1), (x+1,y), (x+1,y+1).
constexpr int kCamDStr = 130;        // f32 per dA row: 128 slots + 2 (8-byte aligned rows, 2 (mod 32) banks)
constexpr int kCamPasses = (kCamKbMax * 16 + 95) / 96;

template <typename T, int P, bool OL16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4))) void lift_cam_bwd_query_kernel(const LiftArgs a, const CamArgs c) {
  static_assert(P == 8, "two groups of 4 points");
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  using M = mma_traits<T>;
  constexpr int PG = P / 2;
  const int lane = threadIdx.x, n = lane & 31, g = lane >> 5;
  const int witem = xcd_remap(blockIdx.x, c.chunk);
  if (witem >= c.witems) return;
  float* D = lds_f;
  const float* drow = D + n * kCamDStr;
  int h, item, j, b, q;
  const bool valid = cam_item(a, witem, lane, h, item, j, b, q);
  if (!valid) q = 0;
  const long bq = (long)b * a.Nq + q;
  const long row = (long)a.H * 32;
  const int S = a.fh * a.fw;
  float lg[P], w[P], off[2 * PG];
  load_ol<T, P>(a.logits, bq * a.log_stride + h * P, OL16, lg);
  load_ol<T, 2 * PG>(a.offsets, bq * a.off_stride + h * 2 * P + g * 2 * PG, OL16, off);
  unsigned vismask = 0u;
  if (a.vis0 == nullptr) vismask = valid ? ~0u : 0u;
  else
    for (int cam = 0; cam < a.Nc; ++cam)
      vismask |= (valid && a.vis0[(long)cam * a.Nq + q] != 0) ? (1u << cam) : 0u;
  // grad_out fragments (MFMA B operand: column = this lane's query, channels kb*16 + g*8 .. +8)
  uint4 gf[2];
  {
    const T* gp = (const T*)a.gout + bq * row + h * 32 + g * 8;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
      gf[kb] = valid ? *reinterpret_cast<const uint4*>(gp + kb * 16) : make_uint4(0u, 0u, 0u, 0u);
  }
  softmax_row<P, true>(lg, w);
  float wg[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) wg[i] = g ? w[PG + i] : w[i];
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  {
    const float inv_fw = __builtin_amdgcn_rcpf(fwf), inv_fh = __builtin_amdgcn_rcpf(fhf);
#pragma unroll
    for (int i = 0; i < PG; ++i) { off[2 * i] *= inv_fw; off[2 * i + 1] *= inv_fh; }
  }
  float gw[PG], gx[PG], gy[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) { gw[i] = 0.0f; gx[i] = 0.0f; gy[i] = 0.0f; }
  const int MB = (c.KB + 1) >> 1;                       // 32-slot row blocks of the padded map
  const T* __restrict__ value = (const T*)a.value;
  const int fh1 = c.fh1;
  int za[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) za[i] = (g * PG + i) % a.Z;

  for (int cam = 0; cam < a.Nc; ++cam) {
    const bool v = (vismask >> cam) & 1u;
    if (__ballot(v) == 0ull) continue;
    const T* __restrict__ vb = value + ((long)b * a.Nc + cam) * S * row + h * 32 + g * 8;
    const float* __restrict__ rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
    float flx[PG], fly[PG];
    int fk0[PG];
    unsigned pmask = 0u;                                  // passes this lane's points fall in
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const float2 r = *reinterpret_cast<const float2*>(rp + za[i] * 2);
      const PadFoot f = pad_foot(r.x + off[2 * i], r.y + off[2 * i + 1], fwf, fhf, a.fw, a.fh, fh1);
      flx[i] = f.lx; fly[i] = f.ly;
      fk0[i] = v ? f.k0 : -4096;                           // invisible here: in no pass
      const int ps = (f.k0 >= 96) + (f.k0 >= 192);
      if (v) pmask |= 1u << ps;
    }
#pragma unroll
    for (int s = 0; s < kCamPasses; ++s) {
      if (s * 3 >= MB || __ballot((pmask >> s) & 1u) == 0ull) continue;       // wave-uniform
      // V fragments (MFMA A operand: row = padded slot (3s + jb)*32 + n, channels kb*16 + g*8 .. +8)
      uint4 va[4][2];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const int mb = s * 3 + jb;
        int pix;
        const bool in = cam_pixel_mg(mb * 32 + n, fh1, c.mg, a.fw, pix) && mb < MB;
        const T* vp = vb + (long)(in ? pix : 0) * row;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          va[jb][kb] = in ? *reinterpret_cast<const uint4*>(vp + kb * 16) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        f32x16_t d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.0f;
        d = M::mma(va[jb][0], gf[0], d);
        d = M::mma(va[jb][1], gf[1], d);
        // D: row = slot (r & 3) + 8 (r >> 2) + 4 g within the block, column = this lane's query
        float* w0 = D + n * kCamDStr + jb * 32 + 4 * g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          *reinterpret_cast<float2*>(w0 + 8 * k) = make_float2(d[4 * k], d[4 * k + 1]);
          *reinterpret_cast<float2*>(w0 + 8 * k + 2) = make_float2(d[4 * k + 2], d[4 * k + 3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int lo = s * 96;
#pragma unroll
      for (int i = 0; i < PG; ++i) {
        const int k0 = fk0[i] - lo;
        const bool here = (unsigned)k0 < 96u;                // the pass of this point
        const float* e = drow + (here ? k0 : 0);
        const float sel = here ? 1.0f : 0.0f;
        const float d00 = e[0] * sel, d10 = e[1] * sel, d01 = e[fh1] * sel, d11 = e[fh1 + 1] * sel;
        const float lx = flx[i], ly = fly[i], hx = 1.0f - lx, hy = 1.0f - ly;
        gw[i] += hx * (hy * d00 + ly * d10) + lx * (hy * d01 + ly * d11);
        gx[i] += hy * (d01 - d00) + ly * (d11 - d10);
        gy[i] += hx * (d10 - d00) + lx * (d11 - d01);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // softmax backward over all 8 points: the other 4 live in the partner lane
  const float inv_cnt = (a.count != nullptr) ? __builtin_amdgcn_rcpf(a.count[bq]) : 1.0f;
  float sp = 0.0f;
#pragma unroll
  for (int i = 0; i < PG; ++i) { gw[i] *= inv_cnt; sp = fmaf(wg[i], gw[i], sp); }
  const float s = sp + __shfl_xor(sp, 32, 64);
  if (valid) {
    float gl[PG], gofs[2 * PG];
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      gl[i] = wg[i] * (gw[i] - s);
      gofs[2 * i] = wg[i] * gx[i] * inv_cnt;
      gofs[2 * i + 1] = wg[i] * gy[i] * inv_cnt;
    }
    store_ol<T, PG>(a.glog, bq * a.glog_stride + h * P + g * PG, OL16, gl);
    store_ol<T, 2 * PG>(a.goff, bq * a.goff_stride + h * 2 * P + g * 2 * PG, OL16, gofs);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, value side, same padded-map coefficients: grad_value[pix, :] = sum_q A[q][pix] G[q, :]
// over a camera's visible queries (G = grad_out / count).  As in lift_bwd_value_camera_kernel a wave
// walks its share of the camera's compacted list 64 queries at a time, keeps the partial map in
// registers and writes one slab; but all 8 points of a query go into ONE coefficient matrix
// A^T[slot][query] (read-modify-write in the lane's own column, no per-corner masks thanks to the zero
// border), so a batch is one MFMA round over the touched 32-slot row blocks instead of eight rounds
// of scatter / ballot / multiply / clear.
constexpr int kCamVStride = 72;      // u16 per A^T row: 64 query columns + 8 (144 B: conflict-free 16-byte reads)

template <typename T, int P, int MBT>
__global__ __launch_bounds__(256) void lift_cam_bwd_value_kernel(const LiftArgs a, const TileArgs t, const CamArgs c) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  using M = mma_traits<T>;
  using CV = cam_cvt<T>;
  constexpr int DH = 32;
  constexpr int NV = DH * elem<T>::kBytes / 16;
  constexpr int kA = MBT * 32 * kCamVStride, kG = 64 * DH;
  TileGeom g;
  if (!tile_decode(a, t, g)) return;
  const int lane = threadIdx.x & 63;
  int l0, ncand;
  long slab_idx;
  if (t.balanced) {
    const int W = a.Nc * t.chunks, k = g.cam * t.chunks + g.ck;
    CamShare sh;
    if (!cam_share(a, W, k, sh)) return;
    g.cam = sh.cam; l0 = sh.l0; ncand = sh.ncand;
    slab_idx = ((long)g.b * a.H + g.h) * W + k;
  } else {
    const int cq = cam_chunk_len(a.cam_n[g.cam], t.chunks);
    l0 = g.ck * cq;
    ncand = min(cq, a.cam_n[g.cam] - l0);
    slab_idx = (((long)g.b * a.Nc + g.cam) * a.H + g.h) * t.chunks + g.ck;
  }
  if (ncand <= 0) return;
  uint16_t* A = lds_all + (threadIdx.x >> 6) * (kA + kG);
  uint16_t* G = A + kA;
  for (int i = lane; i < kA / 8; i += 64) reinterpret_cast<uint4*>(A)[i] = make_uint4(0u, 0u, 0u, 0u);
  f32x16_t acc[MBT];
#pragma unroll
  for (int mb = 0; mb < MBT; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
  const long row = (long)a.H * DH;
  const T* __restrict__ gout = (const T*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = __builtin_amdgcn_rcpf(fwf), inv_fh = __builtin_amdgcn_rcpf(fhf);
  const int n = lane & 31, kg = lane >> 5;
  const int fh1 = c.fh1;
  uint16_t* acol = A + lane;

  struct Raw {
    float off[2 * P], lg[P];
    float2 ref[P];
    float cnt;
    uint4 grow[NV];
    bool valid;
  };
  auto fetch_q = [&](int c0, bool& valid) -> int {
    const int cc = c0 + lane;
    valid = cc < ncand;
    return a.cam_list[(long)g.cam * a.Nq + l0 + (valid ? cc : 0)];
  };
  auto fetch_raw = [&](int q, bool valid, Raw& rw) {
    rw.valid = valid;
    const long bq = (long)g.b * a.Nq + q;
    load_ol<T, P>(a.logits, bq * a.log_stride + g.h * P, a.ol16, rw.lg);
    load_ol<T, 2 * P>(a.offsets, bq * a.off_stride + g.h * 2 * P, a.ol16, rw.off);
    const float* rp = a.ref + (((long)g.cam * a.B + g.b) * a.Nq + q) * a.Z * 2;
    int zi = 0;
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
    // grad_out rows of the 64 queries -> LDS, then this lane's B fragments (8 queries x its channel)
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<uint4*>(G + lane * DH)[i] = cur.grow[i];
    uint4 bfr[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      uint16_t bh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bh[j] = G[(kb * 16 + kg * 8 + j) * DH + n];
      bfr[kb] = make_uint4(bh[0] | ((uint32_t)bh[1] << 16), bh[2] | ((uint32_t)bh[3] << 16),
                           bh[4] | ((uint32_t)bh[5] << 16), bh[6] | ((uint32_t)bh[7] << 16));
    }
    float w[P];
    softmax_row<P, true>(cur.lg, w);
    const float sc = cur.valid ? __builtin_amdgcn_rcpf(cur.cnt) : 0.0f;     // 1 / count, 0 for the list's tail
    unsigned mbmask = 0u;
    int k0s[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PadFoot f = pad_foot(cur.ref[p].x + cur.off[2 * p] * inv_fw, cur.ref[p].y + cur.off[2 * p + 1] * inv_fh,
                                 fwf, fhf, a.fw, a.fh, fh1);
      const float wp = w[p] * sc;
      const float wl = wp * f.lx, wh = wp - wl;
      const float c11 = wl * f.ly, c01 = wl - c11, c10 = wh * f.ly, c00 = wh - c10;
      uint16_t* e = acol + f.k0 * kCamVStride;
      const uint16_t u00 = e[0], u10 = e[kCamVStride], u01 = e[fh1 * kCamVStride], u11 = e[(fh1 + 1) * kCamVStride];
      e[0] = CV::enc(CV::dec(u00) + c00);
      e[kCamVStride] = CV::enc(CV::dec(u10) + c10);
      e[fh1 * kCamVStride] = CV::enc(CV::dec(u01) + c01);
      e[(fh1 + 1) * kCamVStride] = CV::enc(CV::dec(u11) + c11);
      k0s[p] = f.k0;
      mbmask |= (1u << (f.k0 >> 5)) | (1u << ((f.k0 + fh1 + 1) >> 5));
    }
    __builtin_amdgcn_wave_barrier();
    // (written out per row block: a loop with the wave-uniform skip inside is not unrolled and would
    // index acc[] dynamically)
#define UBV_CAM_MB_STEP(mb)                                                                               \
    if ((mb) < MBT && __ballot((mbmask >> (mb)) & 1u) != 0ull) {                                          \
      _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                                  \
        const uint4 af = *reinterpret_cast<const uint4*>(A + ((mb) * 32 + n) * kCamVStride + kb * 16 + kg * 8); \
        acc[(mb) < MBT ? (mb) : 0] = M::mma(af, bfr[kb], acc[(mb) < MBT ? (mb) : 0]);                     \
      }                                                                                                   \
    }
    UBV_CAM_MB_STEP(0) UBV_CAM_MB_STEP(1) UBV_CAM_MB_STEP(2) UBV_CAM_MB_STEP(3)
    UBV_CAM_MB_STEP(4) UBV_CAM_MB_STEP(5) UBV_CAM_MB_STEP(6) UBV_CAM_MB_STEP(7)
#undef UBV_CAM_MB_STEP
    static_assert(MBT <= 8, "row-block steps are written out for 8 blocks");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < P; ++p) {
      uint16_t* e = acol + k0s[p] * kCamVStride;
      e[0] = 0; e[kCamVStride] = 0; e[fh1 * kCamVStride] = 0; e[(fh1 + 1) * kCamVStride] = 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- this share's partial map -> its slab ([b][cam][h][chunk][S][Dh], plain stores).  D: column =
  // channel lane & 31, row = padded slot (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of block mb
  float* __restrict__ slab = a.slab + slab_idx * ((long)a.fh * a.fw * DH);
#pragma unroll
  for (int mb = 0; mb < MBT; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int pix;
      if (cam_pixel_mg(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, fh1, c.mg, a.fw, pix)) slab[(long)pix * DH + n] = acc[mb][r];
    }
  }
}
