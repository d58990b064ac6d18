// GRID / generic lifting with the per-point arithmetic SHARED inside a lane group — included by
// bev_lift.hip inside namespace ubv.
//
// In lift_fwd_kernel / lift_bwd_query_kernel the LP = Dh/VEC lanes of a (query, head) each recompute
// the softmax and every footprint (8x redundant for f32 data, 4x for 16-bit).  Here lane cg of the
// group OWNS the points p = s*LP + cg: it loads only their offsets / logits, the softmax is one
// max / sum butterfly over the group, it builds only their footprints, and each point's 4 corner
// offsets (+ coefficients) reach the other lanes with DPP quad permutes (LP <= 4) or ds_swizzle
// (LP = 8): a broadcast costs one instruction, a footprint ~45.
template <int LP>
__device__ __forceinline__ int bcast_i(int v, int owner);
template <> __device__ __forceinline__ int bcast_i<4>(int v, int owner) {
  switch (owner) {
    case 0: return __builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, true);
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, true);
  }
}
template <> __device__ __forceinline__ int bcast_i<2>(int v, int owner) {
  // lane pairs inside a quad: (0,1) <- lane owner, (2,3) <- lane 2 + owner
  return owner == 0 ? __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xf, 0xf, true)     // [0,0,2,2]
                    : __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xf, 0xf, true);    // [1,1,3,3]
}
template <> __device__ __forceinline__ int bcast_i<8>(int v, int owner) {
  // ds_swizzle, bit mode: lane' = (lane & 0x18) | owner inside each 32-lane half
  switch (owner) {
    case 0: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (0 << 5));
    case 1: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (1 << 5));
    case 2: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (2 << 5));
    case 3: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (3 << 5));
    case 4: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (4 << 5));
    case 5: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (5 << 5));
    case 6: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (6 << 5));
    default: return __builtin_amdgcn_ds_swizzle(v, 0x18 | (7 << 5));
  }
}
template <int LP> __device__ __forceinline__ float bcast_f(float v, int owner) {
  return __int_as_float(bcast_i<LP>(__float_as_int(v), owner));
}

// c + a.lo * b.lo + a.hi * b.hi on packed 16-bit pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): the dot product of a
// gathered value row with the grad_out slice without unpacking either (8 shifts / masks + 8 FMAs -> 4 dot2 per 8
// channels).
template <typename T> __device__ __forceinline__ float dot2_pk(uint32_t a, uint32_t b, float c);
template <> __device__ __forceinline__ float dot2_pk<bf16_t>(uint32_t a, uint32_t b, float c) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a), __builtin_bit_cast(bf2, b), c, false);
}
template <> __device__ __forceinline__ float dot2_pk<f16_t>(uint32_t a, uint32_t b, float c) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}
template <> __device__ __forceinline__ float dot2_pk<float>(uint32_t, uint32_t, float c) { return c; }   // unused

template <int M> __device__ __forceinline__ float max_xor(float v) {
  if constexpr (M == 1)
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)));
  else if constexpr (M == 2)
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)));
  else if constexpr (M == 4)       // after the 1- and 2-steps: row_half_mirror (see add_xor)
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));
  else
    return fmaxf(v, __shfl_xor(v, M, 64));
}
template <int LP> __device__ __forceinline__ float group_sum(float v) {
  if (LP > 1) v = add_xor<1>(v);
  if (LP > 2) v = add_xor<2>(v);
  if (LP > 4) v = add_xor<4>(v);
  return v;
}
template <int LP> __device__ __forceinline__ float group_max(float v) {
  if (LP > 1) v = max_xor<1>(v);
  if (LP > 2) v = max_xor<2>(v);
  if (LP > 4) v = max_xor<4>(v);
  return v;
}

// Offsets / logit of the points this lane owns + the group's softmax.  w_own[s] = 0 for slots past P.
template <typename T, int P, int LP, bool OL16>
__device__ __forceinline__ void own_points(const LiftArgs& a, long bq, int h, int cg,
                                           float (&ox)[(P + LP - 1) / LP], float (&oy)[(P + LP - 1) / LP],
                                           float (&w_own)[(P + LP - 1) / LP]) {
  constexpr int NOWN = (P + LP - 1) / LP;
  constexpr bool FAST = sizeof(T) == 2;
  float lg[NOWN];
  float m = -3.0e38f;
#pragma unroll
  for (int s = 0; s < NOWN; ++s) {
    const int p = s * LP + cg;
    const bool pv = p < P;
    const int pc = pv ? p : 0;
    lg[s] = pv ? ld_ol<T>(a.logits, bq * a.log_stride + h * P + pc, OL16) : -3.0e38f;
    ox[s] = ld_ol<T>(a.offsets, bq * a.off_stride + h * 2 * P + 2 * pc, OL16);
    oy[s] = ld_ol<T>(a.offsets, bq * a.off_stride + h * 2 * P + 2 * pc + 1, OL16);
    m = fmaxf(m, lg[s]);
  }
  m = group_max<LP>(m);
  float sum = 0.0f;
#pragma unroll
  for (int s = 0; s < NOWN; ++s) {
    const bool pv = s * LP + cg < P;
    w_own[s] = pv ? (FAST ? __expf(lg[s] - m) : expf(lg[s] - m)) : 0.0f;
    sum += w_own[s];
  }
  sum = group_sum<LP>(sum);
  const float inv = FAST ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;
#pragma unroll
  for (int s = 0; s < NOWN; ++s) w_own[s] = FAST ? w_own[s] * inv : w_own[s] / sum;
}

// Work item of a block: (8x8 query tile, group of HB heads).  HB = 0: all heads, items dealt to the XCDs in contiguous
// bands (xcd_remap).  HB > 0: an XCD still owns a contiguous band of `chunk` tiles, but walks it once PER HEAD GROUP,
// group slowest — the blocks resident on an XCD at any time then gather from 1/NG of the bytes of the band's value
// rows (f32, HB = 1: 6-8 tile rows x 200 px x 128 B = 1.7 MB instead of 13 MB), which fits its 4 MiB L2.  With all
// heads in one block the band's rows fell out of L2 between neighbouring tile rows and every corner was re-fetched
// over the fabric (PMC: 463 MB fetched for 113 MB of operands, 5.8 TB/s — the kernel was bound by that).
template <int HB>
__device__ __forceinline__ bool shared_item(const LiftArgs& a, int& item, int& h0) {
  if constexpr (HB == 0) {
    item = xcd_remap(blockIdx.x, a.chunk);
    h0 = 0;
  } else {
    const int v = (int)(blockIdx.x >> 3);
    const int hg = v / a.chunk;
    item = (int)(blockIdx.x & 7) * a.chunk + (v - hg * a.chunk);
    h0 = hg * HB;
  }
  return item < a.total_tiles;
}

template <typename T, int DH, int VEC, int P, bool OL16, int HB = 0>
__global__ __launch_bounds__(256) void lift_fwd_shared_kernel(const LiftArgs a) {
  constexpr int LP = DH / VEC;
  constexpr int NOWN = (P + LP - 1) / LP;
  constexpr bool FAST = sizeof(T) == 2;
  int item, h0;
  if (!shared_item<HB>(a, item, h0)) return;
  const int HL = HB == 0 ? a.H : HB;                // heads of this block
  const int LQ = HL * LP, QW = kWave / LQ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cg = lane % LP, h = h0 + (lane / LP) % HL, sub = lane / LQ;
  const int S = a.fh * a.fw;
  const long row = (long)a.H * DH;
  const T* __restrict__ value = (const T*)a.value;
  T* __restrict__ out = (T*)a.out;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;
  const int rowi = (int)row;
  int zown[NOWN];                                   // anchor of flat point p is p % Z (quirk q3)
#pragma unroll
  for (int s = 0; s < NOWN; ++s) zown[s] = (s * LP + cg) % a.Z;

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    const bool valid = lift_query(a, item, li0 + wv * QW + sub, b, q);
    if (!valid) q = 0;                              // keep every lane in the group operations
    b = __builtin_amdgcn_readfirstlane(b);          // a wave's queries share the sample: scalar base addresses
    const long bq = (long)b * a.Nq + q;
    float ox[NOWN], oy[NOWN], w_own[NOWN];
    own_points<T, P, LP, OL16>(a, bq, h, cg, ox, oy, w_own);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;

    for (int cam = 0; cam < a.Nc; ++cam) {
      // (group-uniform: the LP lanes of a group share q)
      if (a.vis0 != nullptr && a.vis0[(long)cam * a.Nq + q] == 0) continue;
      const float* rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
      int oi[NOWN][4];
      float oc[NOWN][4];
#pragma unroll
      for (int s = 0; s < NOWN; ++s) {
        const float2 r = *reinterpret_cast<const float2*>(rp + zown[s] * 2);
        const float lx = r.x + div_or_mul<FAST>(ox[s], fwf, inv_fw);
        const float ly = r.y + div_or_mul<FAST>(oy[s], fhf, inv_fh);
        const Footprint f = make_footprint(lx, ly, a.fh, a.fw);
#pragma unroll
        for (int k = 0; k < 4; ++k) { oi[s][k] = f.idx[k] * rowi; oc[s][k] = w_own[s] * f.w[k]; }
      }
      // wave-uniform base (scalar registers) + 32-bit per-lane element offset: global_load with an SGPR base,
      // no 64-bit address arithmetic per gather
      const T* vb = value + ((long)b * a.Nc + cam) * S * row;
      const unsigned lane_off = (unsigned)(h * DH + cg * VEC);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        constexpr int dummy = 0; (void)dummy;
        const int s = p / LP, owner = p % LP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = bcast_i<LP>(oi[s][k], owner);
          const float c = bcast_f<LP>(oc[s][k], owner);
          float v[VEC];
          vec_io<T, VEC>::load(gather_ptr(vb, (unsigned)idx + lane_off), v);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, v[i], acc[i]);
        }
      }
    }
    if (valid) {
      if (a.count != nullptr) {
        const float cnt = a.count[bq], inv_cnt = 1.0f / cnt;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = div_or_mul<FAST>(acc[i], cnt, inv_cnt);
      }
      vec_io<T, VEC>::store(out + bq * row + h * DH + cg * VEC, acc);
    }
  }
}

template <typename T, int DH, int VEC, int P, bool OL16, int HB = 0>
__global__ __launch_bounds__(256) void lift_bwd_query_shared_kernel(const LiftArgs a) {
  lift_zero_counters(a);
  constexpr int LP = DH / VEC;
  constexpr int NOWN = (P + LP - 1) / LP;
  constexpr bool FAST = sizeof(T) == 2;
  int item, h0;
  if (!shared_item<HB>(a, item, h0)) return;
  const int HL = HB == 0 ? a.H : HB;
  const int LQ = HL * LP, QW = kWave / LQ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cg = lane % LP, h = h0 + (lane / LP) % HL, sub = lane / LQ;
  const int S = a.fh * a.fw;
  const long row = (long)a.H * DH;
  const T* __restrict__ value = (const T*)a.value;
  const T* __restrict__ gout = (const T*)a.gout;
  const float fwf = (float)a.fw, fhf = (float)a.fh;
  const float inv_fw = 1.0f / fwf, inv_fh = 1.0f / fhf;
  const int rowi = (int)row;
  int zown[NOWN];
#pragma unroll
  for (int s = 0; s < NOWN; ++s) zown[s] = (s * LP + cg) % a.Z;

  for (int li0 = 0; li0 < 64; li0 += 4 * QW) {
    int b, q;
    const bool valid = lift_query(a, item, li0 + wv * QW + sub, b, q);
    if (!valid) q = 0;
    b = __builtin_amdgcn_readfirstlane(b);
    const long bq = (long)b * a.Nq + q;
    float ox[NOWN], oy[NOWN], w_own[NOWN];
    own_points<T, P, LP, OL16>(a, bq, h, cg, ox, oy, w_own);
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
#pragma unroll
    for (int s = 0; s < NOWN; ++s) { gw[s] = 0.0f; gx[s] = 0.0f; gy[s] = 0.0f; }

    for (int cam = 0; cam < a.Nc; ++cam) {
      if (a.vis0 != nullptr && a.vis0[(long)cam * a.Nq + q] == 0) continue;
      const float* rp = a.ref + (((long)cam * a.B + b) * a.Nq + q) * a.Z * 2;
      int oi[NOWN][4];
      float om[NOWN][4], olx[NOWN], oly[NOWN];
#pragma unroll
      for (int s = 0; s < NOWN; ++s) {
        const float2 r = *reinterpret_cast<const float2*>(rp + zown[s] * 2);
        const float lx = r.x + div_or_mul<FAST>(ox[s], fwf, inv_fw);
        const float ly = r.y + div_or_mul<FAST>(oy[s], fhf, inv_fh);
        const Footprint f = footprint_px(lx * fwf - 0.5f, ly * fhf - 0.5f, a.fh, a.fw);
        olx[s] = f.lx; oly[s] = f.ly;
#pragma unroll
        for (int k = 0; k < 4; ++k) { oi[s][k] = f.idx[k] * rowi; om[s][k] = f.m[k]; }
      }
      // wave-uniform base (scalar registers) + 32-bit per-lane element offset: global_load with an SGPR base,
      // no 64-bit address arithmetic per gather
      const T* vb = value + ((long)b * a.Nc + cam) * S * row;
      const unsigned lane_off = (unsigned)(h * DH + cg * VEC);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int s = p / LP, owner = p % LP;
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = bcast_i<LP>(oi[s][k], owner);
          float d = 0.0f;
          if constexpr (PK) {
            const uint4 vp = *reinterpret_cast<const uint4*>(gather_ptr(vb, (unsigned)idx + lane_off));
            d = dot2_pk<T>(gop.x, vp.x, d); d = dot2_pk<T>(gop.y, vp.y, d);
            d = dot2_pk<T>(gop.z, vp.z, d); d = dot2_pk<T>(gop.w, vp.w, d);
            dot[k] = group_sum<LP>(d) * gscale;             // every lane of the group holds the full dot
          } else {
            float v[VEC];
            vec_io<T, VEC>::load(gather_ptr(vb, (unsigned)idx + lane_off), v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d = fmaf(go[i], v[i], d);
            dot[k] = group_sum<LP>(d);
          }
        }
        // (measured: reducing the 4 dots per point and letting the owner combine them beats keeping 3P
        //  partial sums per lane and reducing those at the end — 61 vs 79 us on the self-attention shape)
        if (cg == owner) {                                 // the owner combines with ITS footprint of slot s
          const float d0 = dot[0] * om[s][0], d1 = dot[1] * om[s][1], d2 = dot[2] * om[s][2], d3 = dot[3] * om[s][3];
          const float hx_ = 1.0f - olx[s], hy_ = 1.0f - oly[s];
          gw[s] += hy_ * hx_ * d0 + hy_ * olx[s] * d1 + oly[s] * hx_ * d2 + oly[s] * olx[s] * d3;
          gx[s] += (d1 - d0) * hy_ + (d3 - d2) * oly[s];
          gy[s] += (d2 - d0) * hx_ + (d3 - d1) * olx[s];
        }
      }
    }
    // softmax backward over all P points of the group: dlogit_p = w_p (gw_p - sum_k w_k gw_k)
    float sp = 0.0f;
#pragma unroll
    for (int s = 0; s < NOWN; ++s) sp = fmaf(w_own[s], gw[s], sp);      // w_own = 0 past P
    sp = group_sum<LP>(sp);
    if (valid) {
#pragma unroll
      for (int s = 0; s < NOWN; ++s) {
        const int p = s * LP + cg;
        if (p >= P) continue;
        const float gl = w_own[s] * (gw[s] - sp);
        const float gofx = FAST ? w_own[s] * gx[s] : (w_own[s] * gx[s] * fwf) / fwf;
        const float gofy = FAST ? w_own[s] * gy[s] : (w_own[s] * gy[s] * fhf) / fhf;
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
