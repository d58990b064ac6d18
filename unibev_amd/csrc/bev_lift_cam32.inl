// CAMERA plan on the matrix cores for f32 data — included by bev_lift.hip inside namespace ubv, after
// bev_lift_cam.inl whose padded-map geometry (cam_kpad, cam_pixel, pad_foot, cam_item) it shares.
//
// The 16-bit plan multiplies V^T[32 ch, K] . A^T[K, 32 q] with one v_mfma_f32_32x32x16 per K-block.  f32 data
// keeps f32 accuracy on the same matrix cores by splitting BOTH operands into bf16 halves,
//        x = x_hi + x_lo,   x_hi = bf16(x),   x_lo = bf16(x - x_hi)        (both round-to-nearest-even)
// and taking  V.A ~= V_hi.A_hi + V_lo.A_hi + V_hi.A_lo  with f32 accumulation: the dropped V_lo.A_lo and the
// rounding of the lo halves are each <= 2^-18 of a term (bf16 x bf16 products are exact in the f32 accumulator).
// Three MFMAs cost 3/16 of ONE f32-input MFMA pass, so the kernels stay bound by the footprint arithmetic,
// not by the matrix cores, like their 16-bit twins.
//
// What differs from the 16-bit kernels:
//  * a coefficient is ONE dword in LDS, hi | lo << 16, updated by read-modify-write as before (decode both
//    halves, add, re-split: 9 VALU per corner).  Keeping plain f32 in LDS and splitting at fragment-read time
//    would cost 24 VALU per 8-slot fragment, 15 fragments per camera pass against 16 corners.
//  * the MFMA operand fragments de-interleave 8 consecutive dwords into 4 dwords of hi pairs + 4 of lo pairs.
//  * the forward walks the padded map in 128-slot passes (a 17 KB coefficient window per wave) and the value-gradient
//    kernel batches 32 queries (lanes l / l + 32 share a query, as in the forward) so that its A^T stays at
//    the 16-bit kernel's 36 KB per wave.
//  * V fragments come from pre-split copies written once per call: fragment-ordered (forward), natural
//    layout (query gradient).  The forward keeps IEEE softmax / divisions like every other f32 kernel here; the two
//    backward kernels recompute the weights and locations with v_exp / v_rcp (~2 ulp: 1e-7 of a gradient, three
//    orders below its test bar) — exp and the divisions were a fifth of their VALU instructions.

// value (f32, [B*Nc][S][H][32]) -> fragment-ordered padded copies of its hi and lo halves (see value_frags_kernel)
__global__ __launch_bounds__(256) void value_frags32_kernel(const float* __restrict__ value, uint16_t* __restrict__ vf_hi,
                                                            uint16_t* __restrict__ vf_lo, int BNc, int S, int H,
                                                            int fh, int fw, int KB) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)BNc * H * KB * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long r = t >> 6;
  const int kb = (int)(r % KB); r /= KB;
  const int h = (int)(r % H);
  const long bnc = r / H;
  const int m = lane & 31, kg = lane >> 5;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int pix;
    const bool real = cam_pixel(kb * 16 + kg * 8 + j, fh, fw, pix);
    o[j] = real ? value[((bnc * S + pix) * H + h) * 32 + m] : 0.0f;
  }
  uint4 hi, lo;
  split8(o, hi, lo);
  *reinterpret_cast<uint4*>(vf_hi + t * 8) = hi;
  *reinterpret_cast<uint4*>(vf_lo + t * 8) = lo;
}

// value (f32) -> bf16 hi / lo copies in its own layout, 8 elements per thread
__global__ __launch_bounds__(256) void value_split32_kernel(const float* __restrict__ value, uint16_t* __restrict__ hi,
                                                            uint16_t* __restrict__ lo, long n8) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(value)[2 * t], b = reinterpret_cast<const float4*>(value)[2 * t + 1];
  const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint4 h, l;
  split8(f, h, l);
  reinterpret_cast<uint4*>(hi)[t] = h;
  reinterpret_cast<uint4*>(lo)[t] = l;
}

// ------------------------------------------------------------------------------------------------
// Forward.  Same work decomposition as lift_cam_fwd_kernel — a wave is half an 8x8 query tile for one head, lanes
// l and l + 32 share a query and split its 8 points — but the padded map is walked in PASSES of 128 slots that start
// every 96 (as the query-gradient kernels do): a point belongs to the pass its first corner lies in, its last corner
// (<= fh + 2 <= 15 slots further) is still inside it.  The coefficient window is then 32 x 132 dwords = 17 KB instead
// of 31 (9 waves per CU instead of 5), a pass holds 8 hi + 8 lo V fragments instead of 30 in registers, and a half
// tile of BEV queries — a few image columns wide — usually touches one or two of the three passes.
constexpr int kCam32Win = 128;                 // slots per pass
constexpr int kCam32AStr = kCam32Win + 4;      // dwords per row; / 4 is odd: conflict-free 16-byte reads

template <int P, int KBT>
__global__ __launch_bounds__(64) void lift_cam32_fwd_kernel(const LiftArgs a, const CamArgs c) {
  static_assert(P == 8, "two groups of 4 points");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  using M = mma_traits<bf16_t>;
  constexpr int ASTR = kCam32AStr;
  constexpr int PG = P / 2;
  constexpr int NPASS = (KBT * 16 + 95) / 96;
  const int lane = threadIdx.x, g = lane >> 5;
  const int witem = xcd_remap(blockIdx.x, c.chunk);
  if (witem >= c.witems) return;
  uint32_t* A = lds32;
  for (int i = lane; i < 32 * ASTR / 4; i += 64) reinterpret_cast<uint4*>(A)[i] = make_uint4(0u, 0u, 0u, 0u);
  int h, item, j, b, q;
  const bool valid = cam_item(a, witem, lane, h, item, j, b, q);
  if (!valid) q = 0;
  uint32_t* arow = A + (lane & 31) * ASTR;
  const long bq = (long)b * a.Nq + q;
  float lg[P], w[P], off[2 * PG];
  load_ol<float, P>(a.logits, bq * a.log_stride + h * P, false, lg);
  load_ol<float, 2 * PG>(a.offsets, bq * a.off_stride + h * 2 * P + g * 2 * PG, false, off);
  unsigned vismask = 0u;
  if (a.vis0 == nullptr) vismask = valid ? ~0u : 0u;
  else
    for (int cam = 0; cam < a.Nc; ++cam)
      vismask |= (valid && a.vis0[(long)cam * a.Nq + q] != 0) ? (1u << cam) : 0u;
  softmax_row<P, false>(lg, w);
  float wg[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) wg[i] = g ? w[PG + i] : w[i];
  const float fwf = (float)a.fw, fhf = (float)a.fh;
#pragma unroll
  for (int i = 0; i < PG; ++i) { off[2 * i] /= fwf; off[2 * i + 1] /= fhf; }
  f32x16_t acc0, acc1, acc2;                     // V_hi.A_hi, V_lo.A_hi, V_hi.A_lo: three independent MFMA chains
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; acc2[r] = 0.0f; }
  const uint4* __restrict__ vfh = reinterpret_cast<const uint4*>(c.vfrag);
  const uint4* __restrict__ vfl = reinterpret_cast<const uint4*>(c.vfrag_lo);
  const uint32_t* brow = arow + g * 8;
  const int fh1 = c.fh1;
  int za[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) za[i] = (g * PG + i) % a.Z;

  for (int cam = 0; cam < a.Nc; ++cam) {
    const bool v = (vismask >> cam) & 1u;
    if (__ballot(v) == 0ull) continue;
    const long fo = ((((long)b * a.Nc + cam) * a.H + h) * KBT) * 64 + lane;
    const float* __restrict__ rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
    int k0s[PG], ps[PG];
    float cf[PG][4];
    unsigned pmask = 0u;
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const float2 r = *reinterpret_cast<const float2*>(rp + za[i] * 2);
      const PadFoot f = pad_foot(r.x + off[2 * i], r.y + off[2 * i + 1], fwf, fhf, a.fw, a.fh, fh1);
      const float wp = v ? wg[i] : 0.0f;
      const float wl = wp * f.lx, wh = wp - wl;
      cf[i][3] = wl * f.ly; cf[i][2] = wl - cf[i][3];
      cf[i][1] = wh * f.ly; cf[i][0] = wh - cf[i][1];
      ps[i] = v ? (f.k0 >= 96) + (f.k0 >= 192) : -1;      // invisible here: in no pass
      k0s[i] = f.k0 - 96 * (ps[i] < 0 ? 0 : ps[i]);        // window-local slot
      if (v) pmask |= 1u << ps[i];
    }
#pragma unroll
    for (int s = 0; s < NPASS; ++s) {
      if (__ballot((pmask >> s) & 1u) == 0ull) continue;     // wave-uniform
      constexpr int kFull = 8;
      const int nkb = KBT - 6 * s < kFull ? KBT - 6 * s : kFull;       // K-blocks of this pass (compile time per s)
      uint4 avh[kFull], avl[kFull];
#pragma unroll
      for (int kb = 0; kb < kFull; ++kb)
        if (kb < nkb) { avh[kb] = vfh[fo + (long)(6 * s + kb) * 64]; avl[kb] = vfl[fo + (long)(6 * s + kb) * 64]; }
      // the two point groups of a query share its row: one group at a time
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if (g == ph) {
#pragma unroll
          for (int i = 0; i < PG; ++i) {
            if (ps[i] == s) {
              uint32_t* e = arow + k0s[i];
              const uint32_t u00 = e[0], u10 = e[1], u01 = e[fh1], u11 = e[fh1 + 1];
              e[0] = coef_add(u00, cf[i][0]);
              e[1] = coef_add(u10, cf[i][1]);
              e[fh1] = coef_add(u01, cf[i][2]);
              e[fh1 + 1] = coef_add(u11, cf[i][3]);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int kb = 0; kb < kFull; ++kb) {
        if (kb < nkb) {
          uint4 bh, bl;
          coef_frag(brow + kb * 16, bh, bl);
          acc0 = M::mma(avh[kb], bh, acc0);
          acc1 = M::mma(avl[kb], bh, acc1);
          acc2 = M::mma(avh[kb], bl, acc2);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < PG; ++i) {
        if (ps[i] == s) {
          uint32_t* e = arow + k0s[i];
          e[0] = 0u; e[1] = 0u; e[fh1] = 0u; e[fh1 + 1] = 0u;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // D^T: row = channel (r & 3) + 8 (r >> 2) + 4 g, column = this lane's query
  if (valid) {
    const float cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
    float* o = (float*)a.out + bq * ((long)a.H * 32) + h * 32 + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float vv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vv[i] = (acc0[4 * k + i] + (acc1[4 * k + i] + acc2[4 * k + i])) / cnt;
      vec_io<float, 4>::store(o + 8 * k, vv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, query side: dA[K slots, 32 q] = V[K, 32 ch] . G^T[32, 32] with both operands split
// (V from the natural-layout hi / lo copies, G^T split once per wave); everything after the product
// is lift_cam_bwd_query_kernel's.
template <int P>
__global__ __launch_bounds__(64) void lift_cam32_bwd_query_kernel(const LiftArgs a, const CamArgs c) {
  static_assert(P == 8, "two groups of 4 points");
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  using M = mma_traits<bf16_t>;
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
  load_ol<float, P>(a.logits, bq * a.log_stride + h * P, false, lg);
  load_ol<float, 2 * PG>(a.offsets, bq * a.off_stride + h * 2 * P + g * 2 * PG, false, off);
  unsigned vismask = 0u;
  if (a.vis0 == nullptr) vismask = valid ? ~0u : 0u;
  else
    for (int cam = 0; cam < a.Nc; ++cam)
      vismask |= (valid && a.vis0[(long)cam * a.Nq + q] != 0) ? (1u << cam) : 0u;
  // grad_out fragments (MFMA B operand: column = this lane's query, channels kb*16 + g*8 .. +8), split
  uint4 gfh[2], gfl[2];
  {
    const float* gp = (const float*)a.gout + bq * row + h * 32 + g * 8;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float f[8];
      if (valid) {
        const float4 x = *reinterpret_cast<const float4*>(gp + kb * 16), y = *reinterpret_cast<const float4*>(gp + kb * 16 + 4);
        f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w; f[4] = y.x; f[5] = y.y; f[6] = y.z; f[7] = y.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = 0.0f;
      }
      split8(f, gfh[kb], gfl[kb]);
    }
  }
  softmax_row<P, true>(lg, w);                  // (backward: hardware exp / reciprocal, see the header)
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
  const int MB = (c.KB + 1) >> 1;
  const uint16_t* __restrict__ vhi = (const uint16_t*)a.vnat_hi;
  const uint16_t* __restrict__ vlo = (const uint16_t*)a.vnat_lo;
  const int fh1 = c.fh1;
  int za[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) za[i] = (g * PG + i) % a.Z;

  for (int cam = 0; cam < a.Nc; ++cam) {
    const bool v = (vismask >> cam) & 1u;
    if (__ballot(v) == 0ull) continue;
    const long vo = ((long)b * a.Nc + cam) * S * row + h * 32 + g * 8;
    const float* __restrict__ rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
    float flx[PG], fly[PG];
    int fk0[PG];
    unsigned pmask = 0u;
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const float2 r = *reinterpret_cast<const float2*>(rp + za[i] * 2);
      const PadFoot f = pad_foot(r.x + off[2 * i], r.y + off[2 * i + 1], fwf, fhf, a.fw, a.fh, fh1);
      flx[i] = f.lx; fly[i] = f.ly;
      fk0[i] = v ? f.k0 : -4096;
      const int ps = (f.k0 >= 96) + (f.k0 >= 192);
      if (v) pmask |= 1u << ps;
    }
#pragma unroll
    for (int s = 0; s < kCamPasses; ++s) {
      if (s * 3 >= MB || __ballot((pmask >> s) & 1u) == 0ull) continue;
      uint4 vah[4][2], val[4][2];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const int mb = s * 3 + jb;
        int pix;
        const bool in = cam_pixel_mg(mb * 32 + n, fh1, c.mg, a.fw, pix) && mb < MB;
        const long po = vo + (long)(in ? pix : 0) * row;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          vah[jb][kb] = in ? *reinterpret_cast<const uint4*>(vhi + po + kb * 16) : make_uint4(0u, 0u, 0u, 0u);
          val[jb][kb] = in ? *reinterpret_cast<const uint4*>(vlo + po + kb * 16) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        f32x16_t d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          d = M::mma(vah[jb][kb], gfh[kb], d);
          d = M::mma(val[jb][kb], gfh[kb], d);
          d = M::mma(vah[jb][kb], gfl[kb], d);
        }
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
        const bool here = (unsigned)k0 < 96u;
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
      // d(loc) -> d(offset): loc = ref + off / (fw, fh) and x = loc * fw - 0.5: the map sizes cancel
      gofs[2 * i] = wg[i] * gx[i] * inv_cnt;
      gofs[2 * i + 1] = wg[i] * gy[i] * inv_cnt;
    }
    store_ol<float, PG>(a.glog, bq * a.glog_stride + h * P + g * PG, false, gl);
    store_ol<float, 2 * PG>(a.goff, bq * a.goff_stride + h * 2 * P + g * 2 * PG, false, gofs);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, value side: grad_value[pix, :] = sum_q A[q][pix] G[q, :] over a camera's visible queries, 32 per batch.
// Lanes l and l + 32 share query l of the batch: each builds the footprints of 4 of its 8 points and loads half of
// its grad_out row; A^T[slot][query] holds packed (hi | lo << 16) coefficients, G sits in LDS as f32 and is split
// when the B fragments are read.  Like the forward, the padded map is walked in passes of 128 slots that start every
// 96 (row blocks 3s .. 3s + 3 of the accumulators): A^T is a 128-row window, 18 KB + 4 KB of grad_out rows per wave
// instead of 37 + 4, so seven one-wave blocks fit a CU where one four-wave block did — the kernel spent 43 % of its
// cycles waiting with one wave per SIMD (profiles/r03_pmc_sca_img_fp32_sq.txt).
constexpr int kCam32VStride = 36;    // dwords per A^T row: 32 query columns + 4 (144 B: conflict-free 16-byte reads)

template <int P, int MBT>
__global__ __launch_bounds__(256) void lift_cam32_bwd_value_kernel(const LiftArgs a, const TileArgs t, const CamArgs c) {
  static_assert(P == 8, "two groups of 4 points");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
  using M = mma_traits<bf16_t>;
  constexpr int DH = 32, PG = P / 2;
  constexpr int kA = kCam32Win * kCam32VStride, kG = 32 * DH;    // dwords: one pass window of A^T, grad_out rows
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
  uint32_t* A = lds32 + (threadIdx.x >> 6) * (kA + kG);
  float* G = reinterpret_cast<float*>(A + kA);
  for (int i = lane; i < kA / 4; i += 64) reinterpret_cast<uint4*>(A)[i] = make_uint4(0u, 0u, 0u, 0u);
  f32x16_t acc[MBT];
#pragma unroll
  for (int mb = 0; mb < MBT; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
  const long row = (long)a.H * DH;
  const float* __restrict__ gout = (const float*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = __builtin_amdgcn_rcpf(fwf), inv_fh = __builtin_amdgcn_rcpf(fhf);
  const int n = lane & 31, kg = lane >> 5;
  const int fh1 = c.fh1;
  uint32_t* acol = A + n;
  int za[PG];
#pragma unroll
  for (int i = 0; i < PG; ++i) za[i] = (kg * PG + i) % a.Z;

  struct Raw {
    float off[2 * PG], lg[P];
    float2 ref[PG];
    float cnt;
    uint4 grow[4];                      // channels kg*16 .. +16 of the query's grad_out row
    bool valid;
  };
  auto fetch_q = [&](int c0, bool& valid) -> int {
    const int cc = c0 + n;
    valid = cc < ncand;
    return a.cam_list[(long)g.cam * a.Nq + l0 + (valid ? cc : 0)];
  };
  auto fetch_raw = [&](int q, bool valid, Raw& rw) {
    rw.valid = valid;
    const long bq = (long)g.b * a.Nq + q;
    load_ol<float, P>(a.logits, bq * a.log_stride + g.h * P, false, rw.lg);
    load_ol<float, 2 * PG>(a.offsets, bq * a.off_stride + g.h * 2 * P + kg * 2 * PG, false, rw.off);
    const float* rp = a.ref + (((long)g.cam * a.B + g.b) * a.Nq + q) * a.Z * 2;
#pragma unroll
    for (int i = 0; i < PG; ++i) rw.ref[i] = *reinterpret_cast<const float2*>(rp + za[i] * 2);
    rw.cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
    const uint4* gp = reinterpret_cast<const uint4*>(gout + bq * row + g.h * DH + kg * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw.grow[i] = gp[i];
  };

  bool v1, v2;
  Raw cur, nxt;
  const int q1 = fetch_q(0, v1);
  fetch_raw(q1, v1, nxt);
  int q2 = fetch_q(32, v2);
  for (int c0 = 0; c0 < ncand; c0 += 32) {
    cur = nxt;
    if (c0 + 32 < ncand) {
      fetch_raw(q2, v2, nxt);
      q2 = fetch_q(c0 + 64, v2);
    }
    // grad_out rows of the 32 queries -> LDS (f32), then this lane's B fragments: 8 queries x its channel, split
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<uint4*>(G + n * DH + kg * 16)[i] = cur.grow[i];
    __builtin_amdgcn_wave_barrier();
    uint4 bh[2], bl[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float f[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) f[jj] = G[(kb * 16 + kg * 8 + jj) * DH + n];
      split8(f, bh[kb], bl[kb]);
    }
    float w[P];
    softmax_row<P, true>(cur.lg, w);
    const float sc = cur.valid ? __builtin_amdgcn_rcpf(cur.cnt) : 0.0f;           // 1 / count, 0 for the list's tail
    unsigned pmask = 0u;
    int k0s[PG], ps[PG];
    float cf[PG][4];
#pragma unroll
    for (int i = 0; i < PG; ++i) {
      const PadFoot f = pad_foot(cur.ref[i].x + cur.off[2 * i] * inv_fw, cur.ref[i].y + cur.off[2 * i + 1] * inv_fh,
                                 fwf, fhf, a.fw, a.fh, fh1);
      const float wp = (kg ? w[PG + i] : w[i]) * sc;
      const float wl = wp * f.lx, wh = wp - wl;
      cf[i][3] = wl * f.ly; cf[i][2] = wl - cf[i][3];
      cf[i][1] = wh * f.ly; cf[i][0] = wh - cf[i][1];
      ps[i] = (f.k0 >= 96) + (f.k0 >= 192);
      k0s[i] = f.k0 - 96 * ps[i];                          // window-local slot
      pmask |= 1u << ps[i];
    }
    // one pass: the points whose first corner lies in slots [96 s, 96 s + 96) -> coefficients (the two point groups
    // of a query share its column: one group at a time) -> the touched row blocks 3 s + jb -> clear
#define UBV_CAM32_MB_STEP(s, jb)                                                                          \
    if (3 * (s) + (jb) < MBT && __ballot((mbm >> (jb)) & 1u) != 0ull) {                                   \
      constexpr int mbi = 3 * (s) + (jb) < MBT ? 3 * (s) + (jb) : 0;                                      \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                  \
        uint4 ah, al;                                                                                     \
        coef_frag(A + ((jb) * 32 + n) * kCam32VStride + kb * 16 + kg * 8, ah, al);                         \
        acc[mbi] = M::mma(ah, bh[kb], acc[mbi]);                                                          \
        acc[mbi] = M::mma(al, bh[kb], acc[mbi]);                                                          \
        acc[mbi] = M::mma(ah, bl[kb], acc[mbi]);                                                          \
      }                                                                                                   \
    }
#define UBV_CAM32_PASS(s)                                                                                 \
    if (3 * (s) < MBT && __ballot((pmask >> (s)) & 1u) != 0ull) {                                         \
      unsigned mbm = 0u;                                                                                  \
      _Pragma("unroll") for (int ph = 0; ph < 2; ++ph) {                                                  \
        if (kg == ph) {                                                                                   \
          _Pragma("unroll") for (int i = 0; i < PG; ++i) {                                                \
            if (ps[i] == (s)) {                                                                           \
              uint32_t* e = acol + k0s[i] * kCam32VStride;                                                \
              const uint32_t u00 = e[0], u10 = e[kCam32VStride], u01 = e[fh1 * kCam32VStride],            \
                             u11 = e[(fh1 + 1) * kCam32VStride];                                          \
              e[0] = coef_add(u00, cf[i][0]);                                                             \
              e[kCam32VStride] = coef_add(u10, cf[i][1]);                                                 \
              e[fh1 * kCam32VStride] = coef_add(u01, cf[i][2]);                                           \
              e[(fh1 + 1) * kCam32VStride] = coef_add(u11, cf[i][3]);                                     \
            }                                                                                             \
          }                                                                                               \
        }                                                                                                 \
        __builtin_amdgcn_wave_barrier();                                                                  \
      }                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < PG; ++i)                                                      \
        if (ps[i] == (s)) mbm |= (1u << (k0s[i] >> 5)) | (1u << ((k0s[i] + fh1 + 1) >> 5));               \
      UBV_CAM32_MB_STEP(s, 0) UBV_CAM32_MB_STEP(s, 1) UBV_CAM32_MB_STEP(s, 2) UBV_CAM32_MB_STEP(s, 3)     \
      __builtin_amdgcn_wave_barrier();                                                                    \
      _Pragma("unroll") for (int i = 0; i < PG; ++i) {                                                    \
        if (ps[i] == (s)) {                                                                               \
          uint32_t* e = acol + k0s[i] * kCam32VStride;                                                    \
          e[0] = 0u; e[kCam32VStride] = 0u; e[fh1 * kCam32VStride] = 0u; e[(fh1 + 1) * kCam32VStride] = 0u; \
        }                                                                                                 \
      }                                                                                                   \
      __builtin_amdgcn_wave_barrier();                                                                    \
    }
    UBV_CAM32_PASS(0) UBV_CAM32_PASS(1) UBV_CAM32_PASS(2)
#undef UBV_CAM32_PASS
#undef UBV_CAM32_MB_STEP
    static_assert(MBT <= 8, "three passes cover row blocks 0 .. 8");
  }
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
