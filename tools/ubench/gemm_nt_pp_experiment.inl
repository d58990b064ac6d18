// EXPERIMENT RECORD — not built, not part of libunibev_hip.so.
//
// The ping-pong persistent form of the f32 gemm_nt kernel that round 3 (session 5) built, measured and did not keep
// (profiles/r03_gemm_pingpong_experiment.txt): 65 - 67 us at M = 80 000, N = K = 256 against
// 52 - 54 for the kernel in the tree at the time (K loop alone 44 against ~50).  It was compiled inside
// unibev_amd/csrc/gemm_mfma.hip (it uses that file's GemmAct, gemm_mma, cvt_pk_bf16, drop_* helpers, with two study
// fields added to GemmAct: `int w_tiled; int dbg;`), passed tests/test_gemm_gpu.py, and is kept here as the starting
// point for the hand-scheduled version the notes describe.  Kernel first, then the hook that stood in
// gemm_nt_launch() ahead of the regular dispatch.

// ------------------------------------------------------------------------------------------------
// PING-PONG form of the f32 (split-bf16) kernel above for the tall GEMMs of the step (round 3).
//
// Why: with 2 - 3 blocks of gemm_nt_kernel per CU every block is a serial chain of K / 32 memory round trips, and
// the blocks of a CU fall into LOCK-STEP — they start together, wait together, stage together and then queue for
// the matrix pipe together — so a CU's time per block is the SUM of its load, staging, MFMA and store phases
// (10.5 us per block per CU at 3 blocks / CU against 14 us for a lone block: tools/ab/gemm_msweep.py; every single
// resource is ~25 % busy: profiles/r03_pmc_gemm.txt).  More waves (8-wave blocks) or fewer bytes (256-column
// tiles) change nothing; what has to change is WHO is in which phase WHEN.
//
// Here ONE persistent block per CU holds two groups of 4 waves, A and B, each working through its own sequence of
// 128 x 128 tiles with its own LDS operand region, and the block's barriers hold them half a step apart:
//
//      barrier    A: S(c)   B: M(c-1)      S(c): chunk c registers -> bf16 hi / lo -> LDS, then issue the loads of
//      barrier    A: M(c)   B: S(c)              chunk c + XD (X) and c + 1 (W)
//      barrier    A: S(c+1) B: M(c)        M(c): fragment reads + 24 MFMAs per wave
//
// so each SIMD always has one wave on the matrix pipe and its partner on the memory / LDS side (the pairing
// MI355X_MICROARCH.md describes for 8-wave attention loops).  X runs XD = 2 chunks ahead in registers (one block
// per CU leaves 256 VGPRs per wave).  The epilogue (tile -> LDS -> whole rows, as above) is four more phases of a
// group during which its partner keeps the matrix pipe busy.  A and B take the two column tiles of one row tile,
// so B's X loads hit what A fetched.
constexpr int kPPRegion = 2 * (kGemmBM + 128) * 40;       // halves per group: X hi, X lo, W hi, W lo of a 32-deep chunk

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's global STORES (its release
// fence covers global memory), which would end every epilogue phase with a full write round trip
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int XD, bool DUAL>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(const float* __restrict__ X, long ldx,
                                                             const uint16_t* __restrict__ Wh,
                                                             const uint16_t* __restrict__ Wl, long ldw,
                                                             const float* __restrict__ bias, const float* __restrict__ Rv,
                                                             float* __restrict__ Yv, long ldy, long M, int N, int K,
                                                             const GemmAct act, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int NT = 128, KC = 32, LD = KC + 8, XI = 4, WI = 2;
  const int g = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  uint16_t* xh = lds + g * kPPRegion;
  uint16_t* xl = xh + kGemmBM * LD;
  uint16_t* wh = xl + kGemmBM * LD;
  uint16_t* wl = wh + NT * LD;
  const int ny = (N + NT - 1) / NT, nch = K / KC;
  const int xr = tid >> 3, xc = (tid & 7) * 4;            // X chunk [128 x 32] f32: 8 threads per row, rows xr + 32 i
  const int wr = tid >> 2, wc = (tid & 3) * 8;            // W chunk [128 x 32] 16-bit: 4 threads per row, rows wr + 64 i
  // tile of (iteration, group).  ny even: the groups take the two column tiles of a pair; pairs are dealt like the
  // blocks of gemm_nt_kernel (round-robin over the XCDs, the pairs of one row tile in consecutive slots of one XCD);
  // otherwise tiles in linear order
  const bool paired = (ny & 1) == 0;
  const int ppr = ny >> 1;
  auto tile_of = [&](int it, long& m0, int& n0) {
    const unsigned q = (unsigned)it * gridDim.x + blockIdx.x;
    if (paired) {
      const unsigned slot = q >> 3;
      m0 = (long)((slot / (unsigned)ppr) * 8u + (q & 7u)) * kGemmBM;
      n0 = (int)((slot % (unsigned)ppr) * 2u + (unsigned)g) * NT;
    } else {
      const unsigned u = q * 2u + (unsigned)g;
      m0 = (long)(u / (unsigned)ny) * kGemmBM;
      n0 = (int)(u % (unsigned)ny) * NT;
    }
  };
  gf32x4_t xf[XD][XI];
  gu32x4_t wqh[WI], wql[WI];
  // ---- the load streams run over the group's chunks in order, across tile boundaries (and through the
  // epilogue phases); past the last iteration they re-read the last tile (never used).  Row addresses are
  // rebuilt when a stream enters a new tile, a chunk's loads only add its k offset.
  int lx_it = 0, lx_c = 0, lw_it = 0, lw_c = 0;
  // BYTE offsets from the matrix bases, 32-bit (host check: every operand below 4 GB): a load is scalar base +
  // vector offset, no 64-bit vector arithmetic in the staging phase
  uint32_t xo1[XI], xo2[DUAL ? XI : 1], wo[WI];
  auto x_tile = [&]() {
    long m0; int n0;
    tile_of(lx_it < iters ? lx_it : iters - 1, m0, n0);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      long r = m0 + xr + 32 * i;
      r = r < M ? r : M - 1;
      xo1[i] = (uint32_t)((r * ldx + xc) * 4);
      if constexpr (DUAL) xo2[i] = (uint32_t)((r * act.ldx2 + xc) * 4);
    }
  };
  auto w_tile = [&]() {
    long m0; int n0;
    tile_of(lw_it < iters ? lw_it : iters - 1, m0, n0);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int n = n0 + wr + 64 * i;
      n = n < N ? n : N - 1;
      wo[i] = (uint32_t)(((long)n * ldw + wc) * 2);
    }
  };
  x_tile();
  w_tile();
  auto load_x = [&](auto setc) {
    constexpr int set = decltype(setc)::value;
    const int k0 = lx_c * KC;
    bool second = false;
    if constexpr (DUAL) second = k0 >= act.k_split;
    const char* xb = reinterpret_cast<const char*>(second ? (const float*)act.x2 + (k0 - act.k_split) : X + k0);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      uint32_t o = xo1[i];
      if constexpr (DUAL) o = second ? xo2[i] : o;
      xf[set][i] = *reinterpret_cast<const gf32x4_t*>(xb + o);
    }
    if (++lx_c == nch) { lx_c = 0; ++lx_it; x_tile(); }
  };
  auto load_w = [&]() {
    const char* hb = reinterpret_cast<const char*>(Wh + lw_c * KC);
    const char* lb = reinterpret_cast<const char*>(Wl + lw_c * KC);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      wqh[i] = *reinterpret_cast<const gu32x4_t*>(hb + wo[i]);
      wql[i] = *reinterpret_cast<const gu32x4_t*>(lb + wo[i]);
    }
    if (++lw_c == nch) { lw_c = 0; ++lw_it; w_tile(); }
  };
  auto store_chunk = [&](auto setc) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const gf32x4_t v = xf[set][i];
      const uint32_t h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
      const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
      const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
      const int o = (xr + 32 * i) * LD + xc;
      *reinterpret_cast<uint2*>(xh + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(xl + o) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      *reinterpret_cast<gu32x4_t*>(wh + (wr + 64 * i) * LD + wc) = wqh[i];
      *reinterpret_cast<gu32x4_t*>(wl + (wr + 64 * i) * LD + wc) = wql[i];
    }
  };
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  gf32x16_t acc[2][2];
  auto mfma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
      uint4 bh[2], bl[2], ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int xo = ((wm * 2 + i) * 32 + fr) * LD + ks + fk;
        bh[i] = *reinterpret_cast<const uint4*>(xh + xo);
        bl[i] = *reinterpret_cast<const uint4*>(xl + xo);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int wo = ((wn * 2 + j) * 32 + fr) * LD + ks + fk;
        ah[j] = *reinterpret_cast<const uint4*>(wh + wo);
        al[j] = *reinterpret_cast<const uint4*>(wl + wo);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j] = gemm_mma<false>(ah[j], bh[i], acc[i][j]);
          acc[i][j] = gemm_mma<false>(ah[j], bl[i], acc[i][j]);
          acc[i][j] = gemm_mma<false>(al[j], bh[i], acc[i][j]);
        }
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, XD - 1>;
  const int half = lane >> 5;
  const uint64_t act_seed = act.seed + ((act.mode == 1 && act.seed_dev != nullptr) ? *act.seed_dev : 0ull);
  constexpr int TLD = NT + 4;
  float* tile = reinterpret_cast<float*>(xh);              // the group's operand region, free after the K loop

  load_w();
  load_x(Set0{});
  if constexpr (XD == 2) load_x(Set1{});
  if (g == 1) lds_barrier();                             // B runs one phase behind A
  for (int it = 0; it < iters; ++it) {
    long m0; int n0;
    tile_of(it, m0, n0);
    const bool valid = m0 < M && n0 < N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool do_m = valid && !(act.dbg & 1), do_s = !(act.dbg & 2), do_l = !(act.dbg & 4);
    for (int c = 0; c < nch; c += XD) {
      if (do_s) store_chunk(Set0{});                       // S(c)
      if (do_l) { load_w(); __builtin_amdgcn_sched_barrier(0); load_x(Set0{}); }
      lds_barrier();
      if (do_m) mfma_chunk();                              // M(c)
      lds_barrier();
      if constexpr (XD == 2) {
        if (do_s) store_chunk(Set1{});                     // S(c + 1)
        if (do_l) { load_w(); __builtin_amdgcn_sched_barrier(0); load_x(Set1{}); }
        lds_barrier();
        if (do_m) mfma_chunk();                            // M(c + 1)
        lds_barrier();
      }
    }
    // ---- epilogue: two halves of 64 rows, each a write phase (the two waves that own the half: accumulators -> LDS,
    // no global access) and a row phase (all 256 threads: whole rows + bias / activation / residual -> global).  In
    // the row phase a thread always handles the same 4 columns, so it needs ONE bias vector per tile; the operand
    // loads of a half (mask, residual) are issued together, ahead of their uses.
    constexpr int CPR = NT / 4, RPT = 64 * CPR / 256;      // 16-byte pieces per row; rows per thread per half (8)
    const int c4 = (tid % CPR) * 4, rl0 = tid / CPR;       // this thread's columns, first row (rows rl0 + 8 k)
    const bool col_ok = n0 + c4 < N;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && bias != nullptr && col_ok) b4 = *reinterpret_cast<const float4*>(bias + n0 + c4);
    // the next tile's first chunks (issued two and more phases ago) land HERE, before the epilogue's stores are in
    // flight: the staging phases that follow the epilogue then wait for nothing that is queued behind a store
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    const bool second = act.n_split > 0 && n0 >= act.n_split;
    float* const Yo = second ? (float*)act.y2 : Yv;
    const long ldo = second ? act.ldy2 : ldy;
    const int nb = second ? n0 - act.n_split : n0;
    const bool with_r = Rv != nullptr && (act.n_split == 0 || second);
    const bool do_e = valid && !(act.dbg & 8);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (do_e && wm == hh) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mrow = (wm * 2 + i) * 32 + fr;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int nl = (wn * 2 + j) * 32 + 8 * g4 + 4 * half;
              *reinterpret_cast<float4*>(tile + (mrow & 63) * TLD + nl) =
                  make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
            }
        }
      }
      lds_barrier();
      if (do_e) {
        // 4 rows at a time; ONE auxiliary operand per row is fetched ahead of its use — the mask (mode 2) or else the
        // residual; a GEMM with both reads its residual in line.  32-bit byte offsets (host check) keep this phase
        // inside the register budget next to the accumulators and the two chunk sets in flight.
        const bool aux_mask = act.mode == 2, aux_res = with_r && !aux_mask;
        const char* const auxb = aux_mask ? (const char*)act.mask : (const char*)Rv;
#pragma unroll 1
        for (int kb = 0; kb < RPT; kb += 4) {
          uint32_t o[4];
          gf32x4_t a4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const long m = m0 + hh * 64 + rl0 + (256 / CPR) * (kb + k);
            const bool ok = m < M && col_ok;
            const long mc = ok ? m : m0;                   // (clamped: straight-line loads, the result is dropped)
            o[k] = ok ? (uint32_t)((mc * ldo + nb + c4) * 4) : 0xffffffffu;
            if (aux_mask || aux_res) {
              const long ao = (aux_res && act.res_period > 0)
                                  ? ((long)((unsigned)mc % (unsigned)act.res_period) * act.res_ld + nb + (col_ok ? c4 : 0)) * 4
                                  : (mc * ldo + nb + (col_ok ? c4 : 0)) * 4;
              a4[k] = *reinterpret_cast<const gf32x4_t*>(auxb + ao);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rl = rl0 + (256 / CPR) * (kb + k);
            const float4 t4 = *reinterpret_cast<const float4*>(tile + rl * TLD + c4);
            float v[4] = {t4.x + b4.x, t4.y + b4.y, t4.z + b4.z, t4.w + b4.w};
            if (act.mode == 1) {
              const long m = m0 + hh * 64 + rl;
              const uint64_t mix = act.thresh != 0u ? drop_mix64(act_seed, (uint64_t)(m * ldy + n0 + c4) >> 2) : 0ull;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float r = fmaxf(v[e], 0.0f);
                if (act.thresh != 0u) r = drop_keep16(mix, e, act.thresh) ? r * act.scale : 0.0f;
                v[e] = r;
              }
            }
            if (aux_mask) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (a4[k][e] != 0.0f) ? v[e] * act.scale : 0.0f;
            }
            if (aux_res) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += a4[k][e];
            }
            if (o[k] != 0xffffffffu) {
              if (with_r && aux_mask) {                    // both: the residual in line
                const long m = m0 + hh * 64 + rl;
                const long ro = act.res_period > 0 ? (long)((unsigned)m % (unsigned)act.res_period) * act.res_ld + nb + c4
                                                   : m * ldo + nb + c4;
                const float4 r = *reinterpret_cast<const float4*>(Rv + ro);
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
              }
              *reinterpret_cast<float4*>(reinterpret_cast<char*>(Yo) + o[k]) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      }
      lds_barrier();
    }
  }
  if (g == 0) lds_barrier();                             // A's last barrier pairs with B's last phase
}


// ---- launcher hook (inside gemm_nt_launch<SPLIT, F16, OUT16>, after `row_tiles` is known) ----
#if 0
  if constexpr (SPLIT) {
    // ping-pong persistent form (gemm_nt_pp_kernel): tall f32 GEMMs with 128-column tiles.  UBV_GEMM_PP=0: off
    static const int pp_env = getenv("UBV_GEMM_PP") ? atoi(getenv("UBV_GEMM_PP")) : 1;
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kPPRegion * (int)sizeof(uint16_t));
      (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * kPPRegion * (int)sizeof(uint16_t));
    }
    const int ny = (N + 127) / 128;
    const long pairs = (ny & 1) == 0 ? (row_tiles + 7) / 8 * 8 * (ny / 2) : (row_tiles * ny + 1) / 2;
    const bool small = M * ldx < (1L << 30) && (act.x2 == nullptr || M * act.ldx2 < (1L << 30)) && (long)N * ldw < (1L << 31) &&
                       M * ldy < (1L << 30) - 1 && (act.y2 == nullptr || M * act.ldy2 < (1L << 30) - 1);
    if (pp_env != 0 && nt == 128 && K % 64 == 0 && pairs >= 2L * cus && !act.w_tiled && small) {
      const int iters = (int)((pairs + cus - 1) / cus);
      if (act.x2 != nullptr)
        hipLaunchKernelGGL((gemm_nt_pp_kernel<2, true>), dim3((unsigned)cus), dim3(512), 2 * kPPRegion * sizeof(uint16_t), st,
                           (const float*)X, ldx, (const uint16_t*)Wh, (const uint16_t*)Wl, ldw, bias, (const float*)R,
                           (float*)Y, ldy, M, N, K, act, iters);
      else
        hipLaunchKernelGGL((gemm_nt_pp_kernel<2, false>), dim3((unsigned)cus), dim3(512), 2 * kPPRegion * sizeof(uint16_t), st,
                           (const float*)X, ldx, (const uint16_t*)Wh, (const uint16_t*)Wl, ldw, bias, (const float*)R,
                           (float*)Y, ldy, M, N, K, act, iters);
      return UBV_OK;
    }
  }
#endif
