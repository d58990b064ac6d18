// DCNv2 d(input) by OWNER TILES — included by bev_lift.hip inside namespace ubv (it reuses the lifting backward's
// TileAcc: "scatter as a matmul", grad_x[tile] = A . dCol with a sparse coefficient matrix in LDS and MFMA).
//
// The generic col2im kernel (deform_conv.hip) adds every (pixel, tap) sample into its 4 corners with f32 atomics:
// 36 atomics per input element, bound by the atomic rate (890 us at 12 x 256 x 16x44).  Learned offsets are small
// against the regular tap grid, so a wave that owns an 8x8 tile of INPUT pixels and 32 channels can find its own
// samples without binning: per tap, only the output pixels whose unshifted tap position lies within R pixels of
// the tile can reach it with |offset| <= R.  It walks those candidates (64 per batch), keeps the samples whose
// corners fall inside the tile and accumulates them on the matrix cores; every pixel of grad_x is then written
// exactly once with a plain store.  Samples with an offset component beyond R ("far") are skipped here and added
// afterwards by the generic kernel with atomics — the result is exact for any offsets, fast for ordinary ones.
constexpr int kDcnReach = 3;

struct DcnOwnArgs {
  const void* gcol; const void* offset; const void* mask; float* gx;
  int N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg, tiles_x, tiles_y;
};

__device__ __forceinline__ int dcn_ceil_div(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
__device__ __forceinline__ int dcn_floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <typename T>
__global__ __launch_bounds__(256, 3) void dcn_owner_kernel(const DcnOwnArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  constexpr int DH = 32, RB = 2;
  using L = TileLds<T, DH, RB>;
  const int heads = a.C / DH, K = a.kh * a.kw;
  const long item = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
  if (item >= (long)a.N * a.tiles_y * a.tiles_x * heads) return;
  const int lane = threadIdx.x & 63;
  TileGeom g;
  {
    long r = item;
    g.h = (int)(r % heads); r /= heads;                   // head fastest: neighbours share the dCol rows
    const int tx = (int)(r % a.tiles_x); r /= a.tiles_x;
    const int ty = (int)(r % a.tiles_y);
    g.b = (int)(r / a.tiles_y);
    g.cam = 0; g.ck = 0;
    g.x0 = tx * 8; g.y0 = ty * 8;
    g.tw = min(8, a.W - g.x0); g.th = min(8, a.H - g.y0);
    g.npx = 64;
  }
  uint16_t* __restrict__ lds = lds_all + wave_in_block() * L::kWords;
  TileAcc<T, DH, RB> ta;
  ta.init(lds, lane);
  const T* __restrict__ offset = (const T*)a.offset;
  const T* __restrict__ mask = (const T*)a.mask;
  const int grp = (g.h * DH) / (a.C / a.dg);
  const long plane = (long)a.Ho * a.Wo;
  const T* __restrict__ rows = (const T*)a.gcol + (long)g.b * plane * K * a.C + g.h * DH;   // rows (pixel, tap) of C
  const float reach = (float)kDcnReach;
  for (int k = 0; k < K; ++k) {
    const int i = k / a.kw, j = k - i * a.kw;
    // output pixels whose tap k, unshifted, sits within the reach of the tile
    const int ho_lo = max(dcn_ceil_div(g.y0 - kDcnReach - 1 + a.ph - i * a.dh, a.sh), 0);
    const int ho_hi = min(dcn_floor_div(g.y0 + 7 + kDcnReach + a.ph - i * a.dh, a.sh), a.Ho - 1);
    const int wo_lo = max(dcn_ceil_div(g.x0 - kDcnReach - 1 + a.pw - j * a.dw, a.sw), 0);
    const int wo_hi = min(dcn_floor_div(g.x0 + 7 + kDcnReach + a.pw - j * a.dw, a.sw), a.Wo - 1);
    const int nh = ho_hi - ho_lo + 1, nw = wo_hi - wo_lo + 1;
    if (nh <= 0 || nw <= 0) continue;
    const int total = nh * nw;
    const long ob = (((long)g.b * a.dg + grp) * 2 * K + 2 * k) * plane;
    const long mb = (((long)g.b * a.dg + grp) * K + k) * plane;
    for (int c0 = 0; c0 < total; c0 += 64) {
      const int cand = min(c0 + lane, total - 1);
      const bool valid = c0 + lane < total;
      const int ho = ho_lo + cand / nw, wo = wo_lo + (cand - (cand / nw) * nw);
      const long pix = (long)ho * a.Wo + wo;
      const float dy = elem<T>::to_float(offset[ob + pix]), dx = elem<T>::to_float(offset[ob + plane + pix]);
      const float mk = elem<T>::to_float(mask[mb + pix]);
      const bool near = fabsf(dy) <= reach && fabsf(dx) <= reach;       // (a NaN offset is "far")
      const float hp = (float)(ho * a.sh - a.ph + i * a.dh) + dy, wp = (float)(wo * a.sw - a.pw + j * a.dw) + dx;
      const Footprint f = footprint_px(wp, hp, a.H, a.W);
      int lp[4];
      float cwt[4];
      const bool any = tile_own(f, mk, valid && near, g, 8, lp, cwt);
      ta.add(lp, cwt, any, rows, (unsigned)(pix * K + k) * (unsigned)a.C, lane);
    }
  }
  if (ta.fill > 0) ta.flush(lane);
  // D layout: col = lane & 31, pixel (rb, r): lx = (r & 3) + 4 (lane >> 5), ly = 4 rb + (r >> 2)
  const int col = lane & 31, lxh = 4 * (lane >> 5);
  float* __restrict__ out = a.gx + (((long)g.b * a.H + g.y0) * a.W + g.x0) * a.C + g.h * DH + col;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lx = (r & 3) + lxh, ly = 4 * rb + (r >> 2);
      if (lx < g.tw && ly < g.th) out[((long)ly * a.W + lx) * a.C] = ta.acc[rb][r];
    }
}

// Host side of the owner-tile pass; false when the shape is outside its reach (the caller scatters everything).
bool dcn_owner_launch(const void* gcol, const void* offset, const void* mask, float* gx, int N, int H, int W, int C,
                      int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                      int dtype, hipStream_t st) {
  static const int env = getenv("UBV_DCN_OWNER") ? atoi(getenv("UBV_DCN_OWNER")) : 1;
  if (!env || C % 32 != 0 || (C / dg) % 32 != 0 || (long)Ho * Wo * kh * kw * C >= (1L << 30)) return false;   // 32-bit byte offsets into one image's dCol rows
  DcnOwnArgs a{gcol, offset, mask, gx, N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg, (W + 7) / 8, (H + 7) / 8};
  const long items = (long)N * a.tiles_y * a.tiles_x * (C / 32);
  const int waves = dtype == UBV_F32 ? 2 : 4;             // LDS per wave: see plan_backward's GRID settings
  const dim3 grid((unsigned)((items + waves - 1) / waves)), blk(64 * waves);
  using L32 = TileLds<float, 32, 2>;
  using L16 = TileLds<bf16_t, 32, 2>;
  const size_t lds = (size_t)waves * (dtype == UBV_F32 ? L32::kWords : L16::kWords) * sizeof(uint16_t);
  switch (dtype) {
    case UBV_F32: hipLaunchKernelGGL(dcn_owner_kernel<float>, grid, blk, lds, st, a); break;
    case UBV_F16: hipLaunchKernelGGL(dcn_owner_kernel<f16_t>, grid, blk, lds, st, a); break;
    default: hipLaunchKernelGGL(dcn_owner_kernel<bf16_t>, grid, blk, lds, st, a); break;
  }
  return hipGetLastError() == hipSuccess;
}
