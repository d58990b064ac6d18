// Wave-specialised weight gradient (round 5) — included by gemm_mfma.hip inside namespace ubv, after gemm_wgrad_kernel,
// whose tile, LDS layout, fragment reads, pair lists / packing and epilogue it keeps.
//
// What the 4-wave kernel does per 64-row chunk: [wait for the chunk's loads] barrier [split f32 -> bf16 hi + lo, store
// to LDS: ~16 VALU instructions per 16 bytes] barrier [issue the next chunk's loads; transposing LDS reads + 48 MFMAs].
// The four pipes a chunk needs — vector memory 64 KB, VALU ~1 400 cycles per wave, LDS ~1 500 cycles per block, matrix
// ~1 536 cycles per wave — run one after the other inside a block, and two co-resident blocks overlap them only by
// chance: 6.5 us per chunk against 0.7 for any one of the pipes (matrix pipes busy 18 - 20 %, rocprof PMC).
//
// Here a block is 8 waves, ONE block per CU, two waves per SIMD:
//   waves 4 - 7, PRODUCERS: global loads two chunks ahead in two register sets (the gathered variant: pair indices
//       three chunks ahead in two index sets), split, store into operand buffer c & 1;
//   waves 0 - 3, CONSUMERS: transposing reads + MFMAs on buffer c & 1, the 128 x 128 tile in registers as before.
// The operand planes are double-buffered (2 x 66.5 KB) and ONE barrier per chunk hands buffer c & 1 over: the producer's
// VALU / LDS stores / loads for chunk c + 1 run beside the consumer's MFMAs for chunk c on the same SIMD — overlap by
// construction, not by the scheduler's luck.  All loads are unconditional (row indices clamped, rows past the slab
// zeroed by a select when the chunk is stored): a load under a condition makes hipcc drain the whole prefetch at the
// merge (see gemm_wgrad_kernel, gemm_ws.hip).
// ABL (timing study only, results wrong): 1 consumers skip reads + MFMAs, 2 producers skip the global loads, 3 producers
// skip split + LDS stores, 6 barriers and the epilogue only, 7 loads only.
// 4 sets the producers' priority to 1, 5 the consumers'.  PW = producer waves (4, or 8: a 12-wave block).
#ifndef UBV_WGRAD_NSET
#define UBV_WGRAD_NSET 2
#endif
template <int N, typename F> __device__ __forceinline__ void wg_static_for(F&& f) {
  if constexpr (N > 0) {
    wg_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <bool SPLIT, bool F16, bool GATHER, int ABL = 0, int PW = 4>
__global__ __launch_bounds__(256 + 64 * PW, 1) void gemm_wgrad_ws_kernel(const void* __restrict__ dYv, const void* __restrict__ Xv,
                                                               float* __restrict__ partials, long M, int N, int K,
                                                               int tiles_k, int tiles, int splits, int rows_per_split,
                                                               const int32_t* __restrict__ xidx, long xld,
                                                               const int32_t* __restrict__ yidx,
                                                               const int32_t* __restrict__ cnt,
                                                               const void* __restrict__ dY2v, int n_split,
                                                               int kvol, int cwsh) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int BUF = (SPLIT ? 4 : 2) * kWgPlane;         // halves per operand buffer: [ty_h][tx_h]([ty_l][tx_l])
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
  if (split >= splits) return;
  const int tn = GATHER ? 0 : tile / tiles_k, tk = GATHER ? 0 : tile - tn * tiles_k;
  const int n0 = tn * kWgTile, k0 = tk * kWgTile;
  const int pack = GATHER ? (128 >> cwsh) : 1, cmask = (1 << cwsh) - 1;
  const long mbeg = (long)split * rows_per_split;
  const long slab_end = mbeg + rows_per_split < M ? mbeg + rows_per_split : M;
  long mend = slab_end;
  if (GATHER && cnt != nullptr) {
    long most = 0;
    for (int q = 0; q < pack; ++q) {
      const int kq = tile * pack + q;
      if (kq < kvol) most = most > (long)cnt[kq] ? most : (long)cnt[kq];
    }
    mend = mend < most ? mend : most;
  }
  const long n = mbeg < mend ? (mend - mbeg + kWgMC - 1) / kWgMC : 0;    // chunks of this block
  constexpr int NSET = (GATHER && PW == 4) ? 2 : UBV_WGRAD_NSET;   // producer register sets = chunks in flight (registers: 4 gathering waves hold two)
  const long nN = (n + NSET - 1) / NSET * NSET;           // both roles walk whole rounds of the sets (all-zero chunks at the end)

  if (wv >= 4) {
    // ------------------------------------------------------------------------------------------ producers
    const int pt = tid - 256;
    constexpr int NI = (SPLIT ? 8 : 4) * 4 / PW;          // loads per operand per thread per chunk
    if constexpr (ABL == 4) __builtin_amdgcn_s_setprio(1);
    constexpr int CW = SPLIT ? 4 : 8;                     // columns per load
    constexpr int TPR = 128 / CW;                         // threads per row
    constexpr int RS = 64 * PW / TPR;                     // row step between a thread's loads
    using Piece = std::conditional_t<SPLIT, gf32x4_t, gu32x4_t>;
    const int sc = (pt % TPR) * CW, sr = pt / TPR;
    const int mykk = GATHER ? tile * pack + (sc >> cwsh) : 0;
    const bool kk_ok = !GATHER || mykk < kvol;
    const int cl = GATHER ? (sc & cmask) : sc;
    if (GATHER) {
      xidx += (long)(kk_ok ? mykk : 0) * xld;
      if (yidx != nullptr) yidx += (long)(kk_ok ? mykk : 0) * xld;
    }
    long my_end = mend;
    if (GATHER && cnt != nullptr) {
      const long c = kk_ok ? (long)cnt[kk_ok ? mykk : 0] : 0;
      my_end = slab_end < c ? slab_end : c;
    }
    const bool yok = kk_ok && n0 + cl < N, xok = kk_ok && k0 + cl < K;
    const long ycol = yok ? n0 + cl : 0, xcol = xok ? k0 + cl : 0;
    const int soff = (sc >> 4) * kWgCS + (sc & 15);
    // dual dY (f32): block-uniform choice of the matrix, its row pitch and the column inside it
    const bool second = SPLIT && n_split > 0 && n0 >= n_split;
    const char* ybase = (const char*)(second ? dY2v : dYv);
    const long yld = SPLIT && n_split > 0 ? (second ? N - n_split : n_split) : N;
    const long ycol2 = (second && yok) ? ycol - n_split : ycol;
    constexpr int ES = SPLIT ? 4 : 2;                     // bytes per element
    // Rows are counted from the slab's first row in 32 bits and an address is ONE v_mad_u64_u32 (row x pitch + base):
    // the 64-bit compares / selects / multiplies of gemm_wgrad_kernel's address arithmetic were 22 instructions per
    // pair of loads, four of them quarter-rate multiplies — as much producer time as the bf16 split itself.
    const long own_end = my_end > mbeg ? my_end : mbeg;
    const int lim = (int)(own_end - mbeg) - 1;            // this thread's last row, relative (-1: none)
    const char* ysl = ybase + ((GATHER ? 0L : mbeg * yld) + ycol2) * ES;
    const char* xsl = (const char*)Xv + ((GATHER ? 0L : mbeg * (long)K) + xcol) * ES;
    const uint32_t ypitch = (uint32_t)(yld * ES), xpitch = (uint32_t)((long)K * ES);
    if (GATHER) {
      xidx += mbeg;
      yidx += mbeg;
    }
    constexpr uint32_t ALL = (1u << NI) - 1u;

    auto load_idx = [&](int (&pix)[NI], int (&piy)[NI], int crel) __attribute__((always_inline)) {
      if constexpr (GATHER) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          int r = crel + sr + RS * i;
          r = r < lim ? r : lim;
          r = r > 0 ? r : 0;                              // (no row of its own: entry 0 of the slab, never an address)
          pix[i] = xidx[r];
          piy[i] = yidx[r];                               // (pair lists only: the host sends the dense-map form, whose
                                                          //  `yidx != nullptr ? load : row` is a branch and a vmcnt(0)
                                                          //  per load, to the 4-wave kernel)
        }
      }
    };
    auto load_chunk = [&](Piece (&py)[NI], Piece (&px)[NI], uint32_t& ymask, uint32_t& xmask, const int (&pix)[NI],
                          const int (&piy)[NI], int crel) __attribute__((always_inline)) {
      ymask = 0u; xmask = 0u;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rel = crel + sr + RS * i;
        const bool ok = rel <= lim;
        int r = rel < lim ? rel : lim;
        r = r > 0 ? r : 0;
        int xr = r, yr = r;
        bool xrow_ok = ok;
        if constexpr (GATHER) {
          xrow_ok = ok && pix[i] >= 0;
          xr = xrow_ok ? pix[i] : 0;
          yr = (ok && piy[i] >= 0) ? piy[i] : 0;
        }
        py[i] = *reinterpret_cast<const Piece*>(ysl + (uint64_t)(uint32_t)yr * ypitch);
        px[i] = *reinterpret_cast<const Piece*>(xsl + (uint64_t)(uint32_t)xr * xpitch);
        ymask |= (ok && yok) ? (1u << i) : 0u;
        xmask |= (xrow_ok && xok) ? (1u << i) : 0u;
      }
    };
    auto split_store = [&](uint16_t* th, uint16_t* tl, int o, const gf32x4_t v) __attribute__((always_inline)) {
      const uint32_t h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
      const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
      const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
      *reinterpret_cast<uint2*>(th + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(tl + o) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
    };
    // MASKED = false: every piece of every lane of the wave is data (all chunks but a slab's last, whole tiles) — the
    // four selects per piece (a fifth of the split's instructions) are skipped under a wave-uniform branch
    auto store_pieces = [&](const Piece (&py)[NI], const Piece (&px)[NI], uint32_t ymask, uint32_t xmask,
                            uint16_t* buf, auto masked) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked)::value;
      uint16_t* ty_h = buf;
      uint16_t* tx_h = buf + kWgPlane;
      uint16_t* ty_l = buf + 2 * kWgPlane;
      uint16_t* tx_l = buf + 3 * kWgPlane;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int o = soff + (sr + RS * i) * 16;
        const bool ky = !MASKED || ((ymask >> i) & 1u), kx = !MASKED || ((xmask >> i) & 1u);
        if constexpr (SPLIT) {
          split_store(ty_h, ty_l, o, ky ? py[i] : gf32x4_t{0.f, 0.f, 0.f, 0.f});
          split_store(tx_h, tx_l, o, kx ? px[i] : gf32x4_t{0.f, 0.f, 0.f, 0.f});
        } else {
          *reinterpret_cast<gu32x4_t*>(ty_h + o) = ky ? py[i] : gu32x4_t{0u, 0u, 0u, 0u};
          *reinterpret_cast<gu32x4_t*>(tx_h + o) = kx ? px[i] : gu32x4_t{0u, 0u, 0u, 0u};
        }
      }
    };
    auto store_chunk = [&](const Piece (&py)[NI], const Piece (&px)[NI], uint32_t ymask, uint32_t xmask,
                           uint16_t* buf) __attribute__((always_inline)) {
      if constexpr (ABL == 3 || ABL == 6 || ABL == 7) {
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" :: "v"(py[i]), "v"(px[i]));
      } else
      if (__all(ymask == ALL && xmask == ALL)) store_pieces(py, px, ymask, xmask, buf, std::false_type{});
      else store_pieces(py, px, ymask, xmask, buf, std::true_type{});
    };

    // NSET register sets: chunk c lives in set c % NSET from the step of chunk c - NSET to its own; its pair indices in
    // index set c % NSET from the step of chunk c - 2 NSET
    Piece ys[NSET][NI], xs[NSET][NI];
    uint32_t yms[NSET], xms[NSET];
    int ixs[NSET][GATHER ? NI : 1], iys[NSET][GATHER ? NI : 1];
    auto idx = [&](int set, int crel) __attribute__((always_inline)) {
      if constexpr (GATHER) load_idx(ixs[set], iys[set], crel);
    };
    auto chunk = [&](int set, int crel) __attribute__((always_inline)) {
      if constexpr (ABL == 2 || ABL == 6) { yms[set] = ALL; xms[set] = ALL; }
      else if constexpr (GATHER) load_chunk(ys[set], xs[set], yms[set], xms[set], ixs[set], iys[set], crel);
      else {
        const int dummy[NI] = {};
        load_chunk(ys[set], xs[set], yms[set], xms[set], dummy, dummy, crel);
      }
    };
    // prologue: chunks 0 .. NSET - 1 on their way, the indices of chunks NSET .. 2 NSET - 1 behind them.
    // (The sets' loads must stay in the loop's order — set 0, 1, ..: hipcc clusters them otherwise, and the merge of that
    //  order with the back edge's makes every wait in the loop a wait for ALL sets, vmcnt(7) .. (0).)
    wg_static_for<NSET>([&](auto S) __attribute__((always_inline)) { idx(decltype(S)::value, decltype(S)::value * kWgMC); });
    wg_static_for<NSET>([&](auto S) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value;
      chunk(s, s * kWgMC);
      idx(s, (s + NSET) * kWgMC);
      __builtin_amdgcn_sched_barrier(0);
    });
    for (int c = 0; c < (int)nN; c += NSET) {
      wg_static_for<NSET>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const int crel = (c + s) * kWgMC;
        store_chunk(ys[s], xs[s], yms[s], xms[s], lds + ((c + s) & 1) * BUF);   // chunk c + s (its buffer's readers passed barrier c + s - 1)
        chunk(s, crel + NSET * kWgMC);
        idx(s, crel + 2 * NSET * kWgMC);
        __syncthreads();                                  // barrier c + s: buffer (c + s) & 1 complete
      });
    }
    return;
  }

  // -------------------------------------------------------------------------------------------- consumers
  const int wk = wv >> 1, wn = wv & 1;
  const bool want_bias = !GATHER && tk == 0 && wk == 0;
  if constexpr (ABL == 5) __builtin_amdgcn_s_setprio(1);
  gf32x16_t acc[2][2], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }
  const int fbase = ((lane >> 4) & 1) * 4 * kWgCS + (lane >> 5) * 8 * 16 + (lane & 15) * 4;
  const uint32_t one2 = F16 ? 0x3C003C00u : 0x3F803F80u;
  const uint4 ones = make_uint4(one2, one2, one2, one2);
  // Fragments of k step s + 1 are read (into the other register set) before the MFMAs of step s; the bias product is a
  // compile-time variant (a branch inside the chunk — round 5's first version — kept hipcc from moving any read across it)
  auto mma_chunk = [&](const uint16_t* buf, auto with_bias) __attribute__((always_inline)) {
    constexpr bool BIAS = decltype(with_bias)::value;
    const uint16_t* ty_h = buf;
    const uint16_t* tx_h = buf + kWgPlane;
    const uint16_t* ty_l = buf + 2 * kWgPlane;
    const uint16_t* tx_l = buf + 3 * kWgPlane;
    uint4 ah[2][2], al[2][2], bh[2][2], bl[2][2];
    auto frags = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = (wk + 2 * i) * kWgCS + fbase + ks * 16;
        ah[set][i] = wg_frag(tx_h + o);
        if constexpr (SPLIT) al[set][i] = wg_frag(tx_l + o);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = (wn + 2 * j) * kWgCS + fbase + ks * 16;
        bh[set][j] = wg_frag(ty_h + o);
        if constexpr (SPLIT) bl[set][j] = wg_frag(ty_l + o);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < kWgMC / 16; ++s) {
      const int cur = s & 1;
      if (s + 1 < kWgMC / 16) frags(cur ^ 1, (s + 1) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = gemm_mma<F16>(ah[cur][i], bh[cur][j], acc[i][j]);
          if constexpr (SPLIT) {
            acc[i][j] = gemm_mma<F16>(ah[cur][i], bl[cur][j], acc[i][j]);
            acc[i][j] = gemm_mma<F16>(al[cur][i], bh[cur][j], acc[i][j]);
          }
        }
      if constexpr (BIAS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accb[j] = gemm_mma<F16>(ones, bh[cur][j], accb[j]);
          if constexpr (SPLIT) accb[j] = gemm_mma<F16>(ones, bl[cur][j], accb[j]);
        }
      }
    }
  };
  auto walk = [&](auto with_bias) __attribute__((always_inline)) {
    for (long c = 0; c < nN; ++c) {
      __syncthreads();                                    // barrier c
      if constexpr (ABL != 1 && ABL != 6 && ABL != 7) if (c < n) mma_chunk(lds + (c & 1) * BUF, with_bias);
    }
  };
  if (want_bias) walk(std::true_type{});
  else walk(std::false_type{});

  // ---- partial tile (gemm_wgrad_kernel's epilogue)
  float* part = partials + ((long)split * (GATHER ? kvol : 1)) * ((long)N * K + N);
  const int half = lane >> 5, u = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nt = 16 * (wn + 2 * j + 4 * (u >> 4)) + (u & 15);
    const int nn = GATHER ? (nt & cmask) : n0 + nt;
    if (nn >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int uk = 8 * g + 4 * half;
        const int kt = 16 * (wk + 2 * i + 4 * (uk >> 4)) + (uk & 15);
        const int k = GATHER ? (kt & cmask) : k0 + kt;
        if (k >= K) continue;
        if constexpr (GATHER) {
          const int pn = nt >> cwsh;
          if (pn != (kt >> cwsh) || tile * pack + pn >= kvol) continue;
          *reinterpret_cast<float4*>(part + (long)(tile * pack + pn) * ((long)N * K + N) + (long)nn * K + k) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          continue;
        }
        *reinterpret_cast<float4*>(part + (long)nn * K + k) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
    }
    if (want_bias && half == 0) part[(long)N * K + nn] = accb[j][0];
  }
}
