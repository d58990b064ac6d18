// GRID lifting through an LDS WINDOW of the value map — included by bev_lift.hip inside namespace ubv.
//
// The shared-footprint kernels (bev_lift_shared.inl) gather every corner through the L1 / texture path:
// 4 P 16-byte loads per lane, 13-17 TB/s aggregate, a third of that path's peak — that gather, not HBM, is
// what bounds them.  For BEV-grid queries (self-attention, SCA-pts) the 64 queries of an 8x8 tile sample
// around their own cell, so one block = (tile, group of HG heads) first copies a 16x16-pixel window of the
// map (its slice of HG heads: whole 128-byte lines) into LDS with coalesced loads and then serves the
// corners from there: ds_read_b128 at 128 B / clk / CU instead of the texture addresser.  The window is placed
// at the block's smallest corner coordinates (every footprint of the block is computed first and kept in
// registers).  A wave whose pass has a corner outside the window gathers that pass from global memory as
// before, so the result does not depend on the window: it is a cache, never an approximation.
//
// HG * Dh * sizeof(T) = 128 bytes per window row (bf16 / f16: 2 heads, f32: 1 head) + 16 bytes of padding,
// 256 rows = 36 KB per block, 4 blocks per CU.  Lanes: LP = Dh / VEC per (query, head), HG * LP = 8 per
// query, 8 queries per wave, 32 per pass, 2 passes.

template <typename T, int DH, int VEC, int P, bool OL16, int HG>
__global__ __launch_bounds__(256) void lift_fwd_win_kernel(const LiftArgs a, int chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int LP = DH / VEC;
  constexpr int NOWN = (P + LP - 1) / LP;
  constexpr bool FAST = sizeof(T) == 2;
  constexpr int LQ = HG * LP, QW = kWave / LQ, NPASS = 64 / (4 * QW);
  WinGeom g;
  if (!win_decode<HG>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int cg = lane % LP, hl = (lane / LP) % HG, sub = lane / LQ;
  const int h = g.hg * HG + hl;
  const long row = (long)a.H * DH;
  const int rowi = (int)row;
  const T* __restrict__ value = (const T*)a.value;
  T* __restrict__ outp = (T*)a.out;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;

  // ---- every footprint of the block first (kept in registers): they decide where the window goes
  long bqs[NPASS];
  bool valids[NPASS];
  int oi[NPASS][NOWN][4], cx[NPASS][NOWN][2], cy[NPASS][NOWN][2];
  float oc[NPASS][NOWN][4];
  int minx = 0x7fffffff, miny = 0x7fffffff;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    int b, q;
    valids[ps] = lift_query(a, g.tile, ps * 4 * QW + wv * QW + sub, b, q);
    if (!valids[ps]) q = 0;
    const long bq = (long)g.b * a.Nq + q;
    bqs[ps] = bq;
    float ox[NOWN], oy[NOWN], w_own[NOWN];
    own_points<T, P, LP, OL16>(a, bq, h, cg, ox, oy, w_own);
    const float* rp = a.ref + bq * a.Z * 2;
#pragma unroll
    for (int s = 0; s < NOWN; ++s) {
      const float2 r = *reinterpret_cast<const float2*>(rp + ((s * LP + cg) % a.Z) * 2);
      const float lx = r.x + div_or_mul<FAST>(ox[s], fwf, inv_fw);
      const float ly = r.y + div_or_mul<FAST>(oy[s], fhf, inv_fh);
      const Footprint f = make_footprint(lx, ly, a.fh, a.fw);
      cx[ps][s][0] = f.xc[0]; cx[ps][s][1] = f.xc[1]; cy[ps][s][0] = f.yc[0]; cy[ps][s][1] = f.yc[1];
      bool live = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        oi[ps][s][k] = f.idx[k] * rowi;
        oc[ps][s][k] = w_own[s] * f.w[k];
        live = live || oc[ps][s][k] != 0.0f;
      }
      if (live && valids[ps]) { minx = min(minx, f.xc[0]); miny = min(miny, f.yc[0]); }
    }
  }
  win_origin(a, minx, miny, g);
  win_load<T, DH, HG>(a, g, win);
  const T* vb = value + (long)g.b * a.fh * a.fw * row;               // wave-uniform
  const unsigned lane_off = (unsigned)(h * DH + cg * VEC);
  const unsigned lane_lds = (unsigned)((hl * DH + cg * VEC) * sizeof(T));

#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    int ow[NOWN][4];
    bool miss = false;
#pragma unroll
    for (int s = 0; s < NOWN; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ow[s][k] = win_row(cx[ps][s][k & 1], cy[ps][s][k >> 1], g);
        miss = miss || (ow[s][k] < 0 && oc[ps][s][k] != 0.0f);
      }
    // one decision per wave and pass: either path is straight-line code with all of its loads in flight
    if (__ballot(miss) == 0ull) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int s = p / LP, owner = p % LP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int wr = bcast_i<LP>(ow[s][k], owner);
          const float c = bcast_f<LP>(oc[ps][s][k], owner);
          float v[VEC];
          vec_io<T, VEC>::load(reinterpret_cast<const T*>(win + (unsigned)max(wr, 0) * kWinRowB + lane_lds), v);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int s = p / LP, owner = p % LP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = bcast_i<LP>(oi[ps][s][k], owner);
          const float c = bcast_f<LP>(oc[ps][s][k], owner);
          float v[VEC];
          vec_io<T, VEC>::load(gather_ptr(vb, (unsigned)idx + lane_off), v);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
        }
      }
    }
    if (valids[ps]) {
      if (a.count != nullptr) {
        const float cnt = a.count[bqs[ps]], inv_cnt = 1.0f / cnt;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = div_or_mul<FAST>(acc[i], cnt, inv_cnt);
      }
      vec_io<T, VEC>::store(outp + bqs[ps] * row + h * DH + cg * VEC, acc);
    }
  }
}

template <typename T, int DH, int VEC, int P, bool OL16, int HG>
__global__ __launch_bounds__(256) void lift_bwd_query_win_kernel(const LiftArgs a, int chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char win[];
  constexpr int LP = DH / VEC;
  constexpr int NOWN = (P + LP - 1) / LP;
  constexpr bool FAST = sizeof(T) == 2;
  constexpr int LQ = HG * LP, QW = kWave / LQ, NPASS = 64 / (4 * QW);
  WinGeom g;
  if (!win_decode<HG>(a, chunk, g)) return;
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int cg = lane % LP, hl = (lane / LP) % HG, sub = lane / LQ;
  const int h = g.hg * HG + hl;
  const long row = (long)a.H * DH;
  const int rowi = (int)row;
  const T* __restrict__ value = (const T*)a.value;
  const T* __restrict__ gout = (const T*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;

  long bqs[NPASS];
  bool valids[NPASS];
  int oi[NPASS][NOWN][4], cx[NPASS][NOWN][2], cy[NPASS][NOWN][2];
  float om[NPASS][NOWN][4], olx[NPASS][NOWN], oly[NPASS][NOWN], wo[NPASS][NOWN];
  int minx = 0x7fffffff, miny = 0x7fffffff;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    int b, q;
    valids[ps] = lift_query(a, g.tile, ps * 4 * QW + wv * QW + sub, b, q);
    if (!valids[ps]) q = 0;
    const long bq = (long)g.b * a.Nq + q;
    bqs[ps] = bq;
    float ox[NOWN], oy[NOWN];
    own_points<T, P, LP, OL16>(a, bq, h, cg, ox, oy, wo[ps]);
    const float* rp = a.ref + bq * a.Z * 2;
#pragma unroll
    for (int s = 0; s < NOWN; ++s) {
      const float2 r = *reinterpret_cast<const float2*>(rp + ((s * LP + cg) % a.Z) * 2);
      const float lx = r.x + div_or_mul<FAST>(ox[s], fwf, inv_fw);
      const float ly = r.y + div_or_mul<FAST>(oy[s], fhf, inv_fh);
      const Footprint f = footprint_px(lx * fwf - 0.5f, ly * fhf - 0.5f, a.fh, a.fw);
      cx[ps][s][0] = f.xc[0]; cx[ps][s][1] = f.xc[1]; cy[ps][s][0] = f.yc[0]; cy[ps][s][1] = f.yc[1];
      olx[ps][s] = f.lx; oly[ps][s] = f.ly;
      bool live = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        oi[ps][s][k] = f.idx[k] * rowi;
        om[ps][s][k] = f.m[k];
        live = live || f.m[k] != 0.0f;
      }
      if (live && valids[ps]) { minx = min(minx, f.xc[0]); miny = min(miny, f.yc[0]); }
    }
  }
  win_origin(a, minx, miny, g);
  win_load<T, DH, HG>(a, g, win);
  const T* vb = value + (long)g.b * a.fh * a.fw * row;
  const unsigned lane_off = (unsigned)(h * DH + cg * VEC);
  const unsigned lane_lds = (unsigned)((hl * DH + cg * VEC) * sizeof(T));

#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const long bq = bqs[ps];
    const bool valid = valids[ps];
    // 16-bit data with 8 channels per lane: grad_out stays packed and the dots are v_dot2c; the 1 / count and
    // validity factors are applied to the (group-uniform) dot afterwards
    constexpr bool PK = sizeof(T) == 2 && VEC == 8;
    float go[VEC];
    uint4 gop = make_uint4(0u, 0u, 0u, 0u);
    const float inv = valid ? 1.0f : 0.0f;
    const float cnt = (a.count != nullptr) ? a.count[bq] : 1.0f;
    const float inv_cnt = 1.0f / cnt;
    const float gscale = (a.count != nullptr ? inv_cnt : 1.0f) * inv;
    if constexpr (PK) {
      gop = *reinterpret_cast<const uint4*>(gout + bq * row + h * DH + cg * VEC);
    } else {
      vec_io<T, VEC>::load(gout + bq * row + h * DH + cg * VEC, go);
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        go[i] = (a.count != nullptr ? div_or_mul<FAST>(go[i], cnt, inv_cnt) : go[i]) * inv;
    }
    float gw[NOWN], gx[NOWN], gy[NOWN];
    int ow[NOWN][4];
    bool miss = false;
#pragma unroll
    for (int s = 0; s < NOWN; ++s) {
      gw[s] = 0.0f; gx[s] = 0.0f; gy[s] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ow[s][k] = win_row(cx[ps][s][k & 1], cy[ps][s][k >> 1], g);
        miss = miss || (ow[s][k] < 0 && om[ps][s][k] != 0.0f);
      }
    }
    auto corner_dots = [&](auto from_lds) {
      constexpr bool LDS = decltype(from_lds)::value;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int s = p / LP, owner = p % LP;
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = bcast_i<LP>(LDS ? ow[s][k] : oi[ps][s][k], owner);
          float d = 0.0f;
          if constexpr (PK) {
            uint4 vp;
            if constexpr (LDS) vp = *reinterpret_cast<const uint4*>(win + (unsigned)max(idx, 0) * kWinRowB + lane_lds);
            else vp = *reinterpret_cast<const uint4*>(gather_ptr(vb, (unsigned)idx + lane_off));
            d = dot2_pk<T>(gop.x, vp.x, d); d = dot2_pk<T>(gop.y, vp.y, d);
            d = dot2_pk<T>(gop.z, vp.z, d); d = dot2_pk<T>(gop.w, vp.w, d);
            dot[k] = group_sum<LP>(d) * gscale;
          } else {
            float v[VEC];
            if constexpr (LDS) vec_io<T, VEC>::load(reinterpret_cast<const T*>(win + (unsigned)max(idx, 0) * kWinRowB + lane_lds), v);
            else vec_io<T, VEC>::load(gather_ptr(vb, (unsigned)idx + lane_off), v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d = fmaf(go[i], v[i], d);
            dot[k] = group_sum<LP>(d);
          }
        }
        if (cg == owner) {
          const float d0 = dot[0] * om[ps][s][0], d1 = dot[1] * om[ps][s][1], d2 = dot[2] * om[ps][s][2], d3 = dot[3] * om[ps][s][3];
          const float lxs = olx[ps][s], lys = oly[ps][s];
          const float hx_ = 1.0f - lxs, hy_ = 1.0f - lys;
          gw[s] += hy_ * hx_ * d0 + hy_ * lxs * d1 + lys * hx_ * d2 + lys * lxs * d3;
          gx[s] += (d1 - d0) * hy_ + (d3 - d2) * lys;
          gy[s] += (d2 - d0) * hx_ + (d3 - d1) * lxs;
        }
      }
    };
    if (__ballot(miss) == 0ull) corner_dots(std::true_type{});
    else corner_dots(std::false_type{});
    float sp = 0.0f;
#pragma unroll
    for (int s = 0; s < NOWN; ++s) sp = fmaf(wo[ps][s], gw[s], sp);
    sp = group_sum<LP>(sp);
    if (valid) {
#pragma unroll
      for (int s = 0; s < NOWN; ++s) {
        const int p = s * LP + cg;
        if (p >= P) continue;
        const float w = wo[ps][s];
        const float gl = w * (gw[s] - sp);
        const float gofx = FAST ? w * gx[s] : (w * gx[s] * fwf) / fwf;
        const float gofy = FAST ? w * gy[s] : (w * gy[s] * fhf) / fhf;
        const long li = bq * a.glog_stride + h * P + p, oi2 = bq * a.goff_stride + h * 2 * P + 2 * p;
        if constexpr (sizeof(T) == 2) {
          if (OL16) {
            ((T*)a.glog)[li] = elem<T>::from_float(gl);
            ((T*)a.goff)[oi2] = elem<T>::from_float(gofx);
            ((T*)a.goff)[oi2 + 1] = elem<T>::from_float(gofy);
            continue;
          }
        }
        ((float*)a.glog)[li] = gl;
        *reinterpret_cast<float2*>((float*)a.goff + oi2) = make_float2(gofx, gofy);
      }
    }
  }
}

// Window kernels apply to one map per sample with grid-tiled queries and whole head groups.  Measured against the
// shared-footprint gather kernels (bs = 2, init-like offsets, us forward / query-gradient):
//            f32 P=8 (SCA-pts)   f32 P=4 (self)   bf16 P=8        bf16 P=4
//   gather   151 / 171           96 / 108         85 / 96         50 / 52
//   window   114 / 156           97 / 123         105 / 94        56 / 54
// so they are the default only for f32 data with 8 points (the case with the most gather traffic per footprint);
// UBV_LIFT_WIN=2 takes them wherever they apply, 0 never.
template <typename T, int DH, int P>
static bool win_ok(const LiftArgs& a) {
  static const int env = getenv("UBV_LIFT_WIN") ? atoi(getenv("UBV_LIFT_WIN")) : 1;
  constexpr int HG = 128 / (DH * (int)sizeof(T));
  const bool pays = env == 2 || (sizeof(T) == 4 && P == 8);
  return env != 0 && pays && HG >= 1 && HG * DH * (int)sizeof(T) == 128 && a.Nc == 1 && a.qw > 0 && a.vis0 == nullptr &&
         a.H % HG == 0 && a.fh >= 1 && a.fw >= 1;
}
