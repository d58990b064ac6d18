// Hand-written MFMA GEMMs of the encoder's Linear layers (gfx950).
//
//   Y[M, N] = X[M, K] . W[N, K]^T (+ bias[N]) (+ R[M, N])        "NT": both operands K-contiguous
//
// is the forward of every Linear (value_proj, sampling_offsets | attention_weights, output_proj, FFN)
// and, with W := the transposed weight, its input gradient (dX = dY . W, the residual branch's
// gradient R added in the epilogue).  M = bs x 40 000 rows, N, K in {96 .. 512}: tall and skinny,
// bound by streaming X in and Y out.
//
// f32 data runs on the matrix cores as a SPLIT-bf16 product: x = x_hi + x_lo, w = w_hi + w_lo with
// bf16 halves, y = x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in f32.  The dropped x_lo w_lo term
// and the roundings of the lo halves leave ~2^-17 per product — the BEV features of the full-size
// fixture move from 6.0e-5 (IEEE f32 GEMMs) to 3.2e-4 of the reference's, inside the 1e-3 bar —
// while three bf16 MFMAs cost 3/16 of one f32 MFMA pass: the kernel is HBM-bound (164 MB per
// 80 000 x 256 x 256 GEMM) where the f32 library GEMM is MFMA-bound (88 - 180 us measured).
// Weights arrive pre-split (ubv_split_weight: once per step per Linear); activations are split in
// registers on their way into LDS.  16-bit data takes the same kernel with one product.
//
// Block = 4 waves = 128 rows x NT columns (NT = 32 NB <= 256), K in chunks of 32 through LDS
// ([rows][32 + 8] bf16: 80-byte rows keep the 16-byte fragment reads conflict-free; 60 KB per block,
// two blocks per CU); the next chunk's global loads are issued before the MFMAs of the current one.  Operand roles are swapped
// (A = W fragment, B = X fragment) so that a lane ends up with 4 CONSECUTIVE columns of one output
// row: bias / residual / store are 16-byte (8-byte for 16-bit outputs) vectors.
#include <type_traits>

#include "gemm_act.h"
#include "ubv_common.h"

namespace ubv {

typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 gf16x8_t;
typedef __attribute__((ext_vector_type(16))) float gf32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t gu32x4_t;   // staging registers: native vectors (a HIP
typedef __attribute__((ext_vector_type(4))) float gf32x4_t;      // uint4 / float4 struct copy ended up in scratch)

constexpr int kGemmBM = 128;
#ifndef UBV_GEMM_XD
#define UBV_GEMM_XD 1
#endif
// K is walked in chunks of KC = 32 (f32 data: the hi + lo images double the LDS) or 64 (16-bit data, where
// a 32-wide chunk is 8 MFMAs per wave between two barriers); LDS rows of KC + 8 halves (80 / 144 bytes)
// keep the 16-byte fragment reads conflict-free.

template <bool F16> __device__ __forceinline__ gf32x16_t gemm_mma(uint4 a, uint4 b, gf32x16_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gf16x8_t, a), __builtin_bit_cast(gf16x8_t, b),
                                                  c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8_t, a),
                                                   __builtin_bit_cast(gbf16x8_t, b), c, 0, 0, 0);
}

// SPLIT: X is f32, W is given as two bf16 matrices (hi, lo); else X, W are 16-bit (F16: f16, else bf16).
// OUT16: Y (and R) in the 16-bit type, else f32.
// Wave tiling: NB even -> 2 x 2 waves of (2 row blocks) x (NB / 2 column blocks): per 16-wide k step a
// wave reads 2 + NB/2 fragment pairs from LDS for 2 * NB/2 MFMA groups (a 1-D split read 1 + NB for
// NB groups and ran at a third of the speed); NB odd -> 4 x 1 waves of 1 x NB.
template <int NB, bool SPLIT, bool F16, bool OUT16, int KC>
__global__ __launch_bounds__(256, (NB <= 4 ? 3 : 2)) void gemm_nt_kernel(const void* __restrict__ Xv, long ldx,
                                                      const uint16_t* __restrict__ Wh,
                                                      const uint16_t* __restrict__ Wl, long ldw,
                                                      const float* __restrict__ bias, const void* __restrict__ Rv,
                                                      void* __restrict__ Yv, long ldy, long M, int N, int K,
                                                      const GemmAct act) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  constexpr int NT = 32 * NB;
  constexpr int XD = UBV_GEMM_XD;                         // register sets of X in flight
  constexpr int kGemmKC = KC, kGemmLd = KC + 8;
  constexpr bool TWO_D = (NB % 2) == 0;
  constexpr int WMB = TWO_D ? 2 : 1;                     // row blocks per wave
  constexpr int WNB = TWO_D ? NB / 2 : NB;               // column blocks per wave
  uint16_t* xh = lds;                                    // [128][40]
  uint16_t* xl = xh + kGemmBM * kGemmLd;                 // (SPLIT only)
  uint16_t* wh = xh + (SPLIT ? 2 : 1) * kGemmBM * kGemmLd;   // [NT][40]
  uint16_t* wl = wh + NT * kGemmLd;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = TWO_D ? (wv >> 1) : wv, wn = TWO_D ? (wv & 1) : 0;
  // 1-D grid, XCD-aware: block ids go round-robin over the 8 XCDs, so the column tiles of one row tile
  // are given consecutive slots of ONE XCD — they run together and share the X rows through its L2
  const int ny = (N + NT - 1) / NT, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long m0 = ((long)(slot / ny) * 8 + xcd) * kGemmBM;
  const int n0 = (slot % ny) * NT;
  if (m0 >= M) return;
  // ---- staging maps.  X chunk [128 x KC]: a row is KC/4 (f32) or KC/8 (16-bit) threads of 16 bytes;
  // W chunk [NT x KC] 16-bit: KC/8 threads per row.
  constexpr int TPRX = SPLIT ? KC / 4 : KC / 8, XSTEP = 256 / TPRX, XI = kGemmBM / XSTEP;
  constexpr int TPRW = KC / 8, WSTEP = 256 / TPRW, WI = (NT + WSTEP - 1) / WSTEP;
  const int xr = tid / TPRX, xc = (tid % TPRX) * (SPLIT ? 4 : 8);
  const int wr = tid / TPRW, wc = (tid % TPRW) * 8;
  // X chunks are fetched XD chunks ahead (register sets xf[0..XD)), W chunks — L2 hits — one ahead:
  // the bytes in flight per CU, not the MFMA or LDS rate, set the speed of this kernel
  gf32x4_t xf[XD][SPLIT ? XI : 1];
  gu32x4_t xq[XD][SPLIT ? 1 : XI];
  gu32x4_t wqh[WI], wql[SPLIT ? WI : 1];

  // (rows past M are clamped, not predicated — their results are never stored — so that the loads are
  //  straight-line code and the compiler can count them: a predicated load made it wait for vmcnt(0) at
  //  the top of every chunk, which put the whole prefetch back in series with the MFMAs)
  // 32-bit BYTE offsets from the matrix bases (host check: operands below 4 GB): a load is scalar base + vector
  // offset, and the address registers of a thread shrink from 20 to 10 — the room the second X set needs
  constexpr int XB = SPLIT ? 4 : 2;                        // bytes per X element
  uint32_t xrow[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const long r = m0 + xr + XSTEP * i;
    xrow[i] = (uint32_t)(((r < M ? r : M - 1) * ldx + xc) * XB);
  }
  uint32_t xrow2[SPLIT ? XI : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const long r = m0 + xr + XSTEP * i;
      xrow2[i] = (uint32_t)(((r < M ? r : M - 1) * act.ldx2 + xc) * XB);
    }
  }
  uint32_t wrow[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int r = wr + WSTEP * i;
    const int n = n0 + (r < NT ? r : NT - 1);
    wrow[i] = (uint32_t)(((long)(n < N ? n : N - 1) * ldw + wc) * 2);  // (a ragged last tile re-reads row N - 1: never stored)
  }
  auto load_x = [&](int k0, auto setc) {
    constexpr int set = decltype(setc)::value;
    if constexpr (SPLIT) {
      const bool second = act.x2 != nullptr && k0 >= act.k_split;            // chunk-uniform (k_split % KC == 0)
      const char* xb = reinterpret_cast<const char*>(second ? (const float*)act.x2 + (k0 - act.k_split) : (const float*)Xv + k0);
#pragma unroll
      for (int i = 0; i < XI; ++i) xf[set][i] = *reinterpret_cast<const gf32x4_t*>(xb + (second ? xrow2[i] : xrow[i]));
    } else {
      const char* xb = reinterpret_cast<const char*>((const uint16_t*)Xv + k0);
#pragma unroll
      for (int i = 0; i < XI; ++i) xq[set][i] = *reinterpret_cast<const gu32x4_t*>(xb + xrow[i]);
    }
  };
  auto load_w = [&](int k0) {
    const char* hb = reinterpret_cast<const char*>(Wh + k0);
    const char* lb = reinterpret_cast<const char*>(Wl + k0);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      wqh[i] = *reinterpret_cast<const gu32x4_t*>(hb + wrow[i]);
      if constexpr (SPLIT) wql[i] = *reinterpret_cast<const gu32x4_t*>(lb + wrow[i]);
    }
  };
  auto store_chunk = [&](auto setc) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if constexpr (SPLIT) {
        const gf32x4_t v = xf[set][i];
        const uint32_t h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
        const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
        const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
        const int o = (xr + XSTEP * i) * kGemmLd + xc;
        *reinterpret_cast<uint2*>(xh + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(xl + o) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
      } else {
        *reinterpret_cast<gu32x4_t*>(xh + (xr + XSTEP * i) * kGemmLd + xc) = xq[set][i];
      }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      if (wr + WSTEP * i < NT) {
        *reinterpret_cast<gu32x4_t*>(wh + (wr + WSTEP * i) * kGemmLd + wc) = wqh[i];
        if constexpr (SPLIT) *reinterpret_cast<gu32x4_t*>(wl + (wr + WSTEP * i) * kGemmLd + wc) = wql[i];
      }
    }
  };

  const int fr = lane & 31, fk = (lane >> 5) * 8;         // fragment row within a 32-block, k offset
  // staged epilogue (NT <= 128): a thread of the row phase always handles the same 4 columns, so bias is ONE vector
  // per thread, fetched here — in the epilogue's write phase it was a load + full wait per 4 columns, 4 NB dependent
  // L2 round trips at the end of every block's serial chain
  constexpr bool STAGE_B = NT <= 128 && (256 % (NT / 4)) == 0;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (STAGE_B) {
    const int bc = n0 + (tid % (NT / 4)) * 4;
    if (bias != nullptr && bc < N) b4 = *reinterpret_cast<const float4*>(bias + bc);
  }
  gf32x16_t acc[WMB][WNB];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto mfma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < kGemmKC; ks += 16) {
      uint4 bh[WMB], bl[WMB], ah[WNB], al[WNB];
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        const int xo = ((wm * WMB + i) * 32 + fr) * kGemmLd + ks + fk;
        bh[i] = *reinterpret_cast<const uint4*>(xh + xo);
        if constexpr (SPLIT) bl[i] = *reinterpret_cast<const uint4*>(xl + xo);
      }
#pragma unroll
      for (int j = 0; j < WNB; ++j) {
        const int wo = ((wn * WNB + j) * 32 + fr) * kGemmLd + ks + fk;
        ah[j] = *reinterpret_cast<const uint4*>(wh + wo);
        if constexpr (SPLIT) al[j] = *reinterpret_cast<const uint4*>(wl + wo);
      }
#pragma unroll
      for (int j = 0; j < WNB; ++j)
#pragma unroll
        for (int i = 0; i < WMB; ++i) {
          acc[i][j] = gemm_mma<F16>(ah[j], bh[i], acc[i][j]);
          if constexpr (SPLIT) {
            acc[i][j] = gemm_mma<F16>(ah[j], bl[i], acc[i][j]);
            acc[i][j] = gemm_mma<F16>(al[j], bh[i], acc[i][j]);
          }
        }
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, XD - 1>;
  // one pipeline step: chunk c (held in register set `setc`) -> LDS -> MFMAs, with the W chunk one ahead
  // and the X chunk XD ahead issued in between.  W first: loads return in order, so the next step's wait
  // for W leaves the later X loads in flight.  The prefetches are compile-time flags (main loop / tail).
  auto step = [&](int c, auto setc, auto lw, auto lx) {
    __syncthreads();                                      // previous chunk's fragments consumed
    store_chunk(setc);
    __syncthreads();
    if constexpr (decltype(lw)::value) load_w((c + 1) * kGemmKC);
    if constexpr (decltype(lx)::value) load_x((c + XD) * kGemmKC, setc);
    mfma_chunk();
  };
  using Yes = std::true_type;
  using No = std::false_type;
  const int nch = K / kGemmKC;
  load_w(0);
  load_x(0, Set0{});
  int c = 0;
  if constexpr (XD == 2) {
    if (nch > 1) load_x(kGemmKC, Set1{});
    for (; c + 3 < nch; c += 2) { step(c, Set0{}, Yes{}, Yes{}); step(c + 1, Set1{}, Yes{}, Yes{}); }
    const int rem = nch - c;
    if (rem == 3) { step(c, Set0{}, Yes{}, Yes{}); step(c + 1, Set1{}, Yes{}, No{}); step(c + 2, Set0{}, No{}, No{}); }
    else if (rem == 2) { step(c, Set0{}, Yes{}, No{}); step(c + 1, Set1{}, No{}, No{}); }
    else step(c, Set0{}, No{}, No{});
  } else {
    for (; c + 1 < nch; ++c) step(c, Set0{}, Yes{}, Yes{});
    step(c, Set0{}, No{}, No{});
  }
  // ---- epilogue.  D[n][m]: column = this lane's row m, rows n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // within a column block: 4 consecutive output columns per (block, r >> 2).
  // Stored directly, that is 16 bytes per lane on 32 different rows per instruction: 32-byte pieces that L2 has to
  // merge into lines (skipping the stores took the 256 x 256 f32 GEMM from 51 to 35 us; non-temporal stores: 107).
  // So the tile goes through LDS (free after the K loop) in two halves of 64 rows and leaves as whole rows: a wave
  // store is two full rows of the tile; the residual / mask operands are read with the same pattern.
  const int half = lane >> 5;
  const uint64_t act_seed = act.seed + ((act.mode == 1 && act.seed_dev != nullptr) ? *act.seed_dev : 0ull);
  using TO = typename std::conditional<OUT16, typename std::conditional<F16, f16_t, bf16_t>::type, float>::type;
  constexpr bool STAGE = NT <= 128;
  constexpr int TLD = NT + 4;                              // f32 per staged row: 16-byte shifts per row
  float* tile = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int hh = 0; hh < (STAGE ? 2 : 1); ++hh) {
    if constexpr (STAGE) __syncthreads();                 // fragments consumed / previous half stored
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int mrow = (wm * WMB + i) * 32 + fr;          // row inside the block tile
      if (STAGE && (mrow >> 6) != hh) continue;           // (wave-uniform: a wave's rows lie in one half)
      const long m = m0 + mrow;
#pragma unroll
      for (int j = 0; j < WNB; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = (wn * WNB + j) * 32 + 8 * g + 4 * half, n = n0 + nl;
          float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (!STAGE && bias != nullptr && n < N) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
          if (!STAGE && act.mode == 1) {
            const uint64_t mix = act.thresh != 0u ? drop_mix64(act_seed, (uint64_t)(m * ldy + n) >> 2) : 0ull;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float r = fmaxf(v[e], 0.0f);
              if (act.thresh != 0u) r = drop_keep16(mix, e, act.thresh) ? r * act.scale : 0.0f;
              v[e] = r;
            }
          }
          if constexpr (STAGE) {
            *reinterpret_cast<float4*>(tile + (mrow & 63) * TLD + nl) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            if (m >= M || n >= N) continue;
            if (act.mode == 2) {
              float k4[4];
              vec_io<TO, 4>::load((const TO*)act.mask + m * ldy + n, k4);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (k4[e] != 0.0f) ? v[e] * act.scale : 0.0f;
            }
            if (Rv != nullptr) {
              float r4[4];
              const long ro = act.res_period > 0 ? (m % act.res_period) * act.res_ld + n : m * ldy + n;
              vec_io<TO, 4>::load((const TO*)Rv + ro, r4);
              v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
            }
            vec_io<TO, 4>::store((TO*)Yv + m * ldy + n, v);
          }
        }
      }
    }
    if constexpr (STAGE) {
      __syncthreads();
      constexpr int CPR = NT / 4;                          // 16-byte chunks per row
      const bool second = act.n_split > 0 && n0 >= act.n_split;           // (block-uniform) this tile belongs to Y2
      TO* const Yo = second ? (TO*)act.y2 : (TO*)Yv;
      const long ldo = second ? act.ldy2 : ldy;
      const int nb = second ? n0 - act.n_split : n0;
      const bool with_r = Rv != nullptr && (act.n_split == 0 || second);
      // 4 pieces at a time: the mask / residual operands of a batch are fetched together, ahead of their uses (one
      // piece at a time, each was a load + full wait — 8 .. 16 dependent round trips at the end of the block's chain)
      constexpr int PIECES = 64 * CPR, BATCH = 4;
      for (int e0 = tid; e0 < PIECES; e0 += 256 * BATCH) {
        long o[BATCH];
        bool ok[BATCH];
        float k4[BATCH][4], r4[BATCH][4], bq[STAGE_B ? 1 : BATCH][4];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int e = e0 + 256 * u;
          const int rl = e / CPR, c4 = (e - rl * CPR) * 4;
          const long m = m0 + hh * 64 + rl;
          ok[u] = e < PIECES && m < M && n0 + c4 < N;
          const long mc = ok[u] ? m : m0;                  // (clamped: straight-line loads, the result is dropped)
          const int cc = ok[u] ? c4 : 0;
          o[u] = mc * ldo + nb + cc;
          if constexpr (!STAGE_B) {                        // (NT = 96: the thread's columns change from piece to piece)
            if (bias != nullptr) vec_io<float, 4>::load(bias + n0 + cc, bq[u]);
          }
          if (act.mode == 2) vec_io<TO, 4>::load((const TO*)act.mask + o[u], k4[u]);
          if (with_r) {
            const long ro = act.res_period > 0 ? (long)((unsigned)mc % (unsigned)act.res_period) * act.res_ld + nb + cc : o[u];
            vec_io<TO, 4>::load((const TO*)Rv + ro, r4[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int e = e0 + 256 * u;
          const int rl = (e < PIECES ? e : tid) / CPR, c4 = (e - (e / CPR) * CPR) * 4;
          const long m = m0 + hh * 64 + rl;
          const float4 t4 = *reinterpret_cast<const float4*>(tile + rl * TLD + c4);
          float v[4] = {t4.x + b4.x, t4.y + b4.y, t4.z + b4.z, t4.w + b4.w};
          if constexpr (!STAGE_B) {
            if (bias != nullptr) { v[0] += bq[u][0]; v[1] += bq[u][1]; v[2] += bq[u][2]; v[3] += bq[u][3]; }
          }
          if (act.mode == 1) {
            const uint64_t mix = act.thresh != 0u ? drop_mix64(act_seed, (uint64_t)(m * ldy + n0 + c4) >> 2) : 0ull;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float r = fmaxf(v[q], 0.0f);
              if (act.thresh != 0u) r = drop_keep16(mix, q, act.thresh) ? r * act.scale : 0.0f;
              v[q] = r;
            }
          }
          if (act.mode == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (k4[u][q] != 0.0f) ? v[q] * act.scale : 0.0f;
          }
          if (with_r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += r4[u][q];
          }
          if (ok[u]) vec_io<TO, 4>::store(Yo + o[u], v);
        }
      }
    }
  }
}

// f32 weight [N, K] -> bf16 hi / lo halves in both orientations: wh, wl [N, K] (forward) and
// wth, wtl [K, N] (input gradient).  One launch per Linear per step.
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ w, int N, int K,
                                                           uint16_t* __restrict__ wh, uint16_t* __restrict__ wl,
                                                           uint16_t* __restrict__ wth, uint16_t* __restrict__ wtl) {
  __shared__ uint16_t th[32][33], tl[32][33];
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;
    uint16_t h = 0, l = 0;
    if (n < N && k < K) {
      const float v = w[(long)n * K + k];
      h = (uint16_t)float_to_bf16_bits(v);
      l = (uint16_t)float_to_bf16_bits(v - bf16_bits_to_float(h));
      wh[(long)n * K + k] = h;
      wl[(long)n * K + k] = l;
    }
    th[ty + 8 * i][tx] = h;
    tl[ty + 8 * i][tx] = l;
  }
  __syncthreads();
  if (wth == nullptr) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, n = n0 + tx;
    if (n < N && k < K) {
      wth[(long)k * N + n] = th[tx][ty + 8 * i];
      wtl[(long)k * N + n] = tl[tx][ty + 8 * i];
    }
  }
}

// All the Linear weights of a pass in ONE launch (the per-weight kernel above is 46 launches of ~5 us per f32
// step, each far too small to fill the chip).  An entry is one parameter [rows, cols]; several entries may be the
// row blocks of one concatenated weight: wh / wl point at the entry's first row of the [N_total, cols] halves,
// wth / wtl at column col_t of the [cols, N_total] transposed halves (leading dimension ld_t).
constexpr int kSplitBatch = 48;
struct SplitEntry {
  const float* w;
  uint16_t *wh, *wl, *wth, *wtl;
  int rows, cols, ld_t, tile_end;                       // tile_end: exclusive prefix of 32 x 32 tiles
};
struct SplitBatch { SplitEntry e[kSplitBatch]; int n; };

__global__ __launch_bounds__(256) void split_weights_batched_kernel(const SplitBatch b) {
  __shared__ uint16_t th[32][33], tl[32][33];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.e[i].tile_end) ++i;
  const SplitEntry& e = b.e[i];
  const int t = blockIdx.x - (i ? b.e[i - 1].tile_end : 0);
  const int tk = (e.cols + 31) / 32;
  const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 8 * r, k = k0 + tx;
    uint16_t h = 0, l = 0;
    if (n < e.rows && k < e.cols) {
      const float v = e.w[(long)n * e.cols + k];
      h = (uint16_t)float_to_bf16_bits(v);
      l = (uint16_t)float_to_bf16_bits(v - bf16_bits_to_float(h));
      e.wh[(long)n * e.cols + k] = h;
      e.wl[(long)n * e.cols + k] = l;
    }
    th[ty + 8 * r][tx] = h;
    tl[ty + 8 * r][tx] = l;
  }
  __syncthreads();
  if (e.wth == nullptr) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + ty + 8 * r, n = n0 + tx;
    if (n < e.rows && k < e.cols) {
      e.wth[(long)k * e.ld_t + n] = th[tx][ty + 8 * r];
      e.wtl[(long)k * e.ld_t + n] = tl[tx][ty + 8 * r];
    }
  }
}

template <bool SPLIT, bool F16, bool OUT16>
static int gemm_nt_launch(const void* X, long ldx, const void* Wh, const void* Wl, long ldw, const float* bias,
                          const void* R, void* Y, long ldy, long M, int N, int K, const GemmAct& act, hipStream_t st) {
  // column tile: the largest of 256 / 192 / 128 / 96 / 64 / 32 that N fills evenly
  // 128-column tiles: 64 accumulator registers per lane -> 3 blocks per CU (49 vs 62 us at N = K = 256, f32); the
  // column tiles of a row tile share X through L2 (block order above).  UBV_GEMM_NT: study knob.
  static const int nt_cap = getenv("UBV_GEMM_NT") ? atoi(getenv("UBV_GEMM_NT")) : 128;
  static const int kc_cap = getenv("UBV_GEMM_KC") ? atoi(getenv("UBV_GEMM_KC")) : 64;
  const int kc = (!SPLIT && K % 64 == 0 && kc_cap >= 64) ? 64 : 32;
  int nt = 0;
  for (int c : {256, 192, 128, 96, 64, 32})
    if (c <= nt_cap && N % c == 0) { nt = c; break; }
  if (act.n_split > 0) nt = 128;                           // dual outputs: 128-column tiles, the last one may be ragged
  if (nt == 0) return UBV_ERR_UNSUPPORTED;
  const long row_tiles = (M + kGemmBM - 1) / kGemmBM;
  const dim3 grid((unsigned)((row_tiles + 7) / 8 * 8 * ((N + nt - 1) / nt))), blk(256);
  // operand tiles, or the epilogue's staged half tile (64 rows of nt + 4 floats) where that is larger: 16-bit data
  // with 32-deep chunks (K % 64 != 0)
  size_t lds = (size_t)((SPLIT ? 2 : 1) * (kGemmBM + nt) * (kc + 8)) * sizeof(uint16_t);
  if (nt <= 128 && lds < (size_t)64 * (nt + 4) * sizeof(float)) lds = (size_t)64 * (nt + 4) * sizeof(float);
#define UBV_GEMM_NB(NBV)                                                                                     \
  case NBV:                                                                                                  \
    if constexpr (!SPLIT) {                                                                                  \
      if (kc == 64) {                                                                                        \
        hipLaunchKernelGGL((gemm_nt_kernel<NBV, SPLIT, F16, OUT16, 64>), grid, blk, lds, st, X, ldx,         \
                           (const uint16_t*)Wh, (const uint16_t*)Wl, ldw, bias, R, Y, ldy, M, N, K, act);    \
        break;                                                                                               \
      }                                                                                                      \
    }                                                                                                        \
    hipLaunchKernelGGL((gemm_nt_kernel<NBV, SPLIT, F16, OUT16, 32>), grid, blk, lds, st, X, ldx,             \
                       (const uint16_t*)Wh, (const uint16_t*)Wl, ldw, bias, R, Y, ldy, M, N, K, act);        \
    break;
  switch (nt / 32) {
    UBV_GEMM_NB(8) UBV_GEMM_NB(6) UBV_GEMM_NB(4) UBV_GEMM_NB(3) UBV_GEMM_NB(2) UBV_GEMM_NB(1)
    default: return UBV_ERR_UNSUPPORTED;
  }
#undef UBV_GEMM_NB
  return UBV_OK;
}

}  // namespace ubv

extern "C" int ubv_split_weights_batched(int n, const float* const* w, const int* rows, const int* cols,
                                         void* const* wh, void* const* wl, void* const* wth, void* const* wtl,
                                         const int* ld_t, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(n >= 0 && (n == 0 || (w && rows && cols && wh && wl && wth && wtl && ld_t)), "split_weights_batched: bad arguments");
  for (int base = 0; base < n; base += kSplitBatch) {
    SplitBatch b{};
    b.n = n - base < kSplitBatch ? n - base : kSplitBatch;
    int tiles = 0;
    for (int i = 0; i < b.n; ++i) {
      const int j = base + i;
      UBV_CHECK_ARG(w[j] && wh[j] && wl[j] && rows[j] > 0 && cols[j] > 0 && (wth[j] == nullptr) == (wtl[j] == nullptr),
                    "split_weights_batched: entry %d", j);
      tiles += ((rows[j] + 31) / 32) * ((cols[j] + 31) / 32);
      b.e[i] = SplitEntry{w[j], (uint16_t*)wh[j], (uint16_t*)wl[j], (uint16_t*)wth[j], (uint16_t*)wtl[j],
                          rows[j], cols[j], ld_t[j], tiles};
    }
    hipLaunchKernelGGL(split_weights_batched_kernel, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), b);
  }
  UBV_CHECK_LAUNCH("split_weights_batched");
  return UBV_OK;
}

extern "C" int ubv_split_weight(const float* w, int N, int K, void* wh, void* wl, void* wth, void* wtl,
                                void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(w && wh && wl && N > 0 && K > 0, "split_weight: bad arguments");
  UBV_CHECK_ARG((wth == nullptr) == (wtl == nullptr), "split_weight: give both transposed halves or neither");
  hipLaunchKernelGGL(split_weight_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, as_stream(stream), w,
                     N, K, (uint16_t*)wh, (uint16_t*)wl, (uint16_t*)wth, (uint16_t*)wtl);
  UBV_CHECK_LAUNCH("split_weight");
  return UBV_OK;
}

static int gemm_nt_run(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                       const float* bias, const void* residual, void* y, int64_t ldy, int64_t M, int N, int K,
                       int dtype, const ubv::GemmAct& act, void* stream, const char* who) {
  using namespace ubv;
  UBV_CHECK_ARG(x && w_hi && y && M >= 0 && N > 0 && K > 0, "%s: bad arguments", who);
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "%s: unknown dtype %d", who, dtype);
  UBV_CHECK_ARG((dtype == UBV_F32) == (w_lo != nullptr), "%s: f32 data takes split weights (hi, lo), 16-bit data one", who);
  if (M == 0) return UBV_OK;
  const int al = dtype == UBV_F32 ? 4 : 8;
  if (K % 32 != 0 || N % 32 != 0 || ldx % al != 0 || ldw % 8 != 0 || ldy % 4 != 0 ||
      ((uintptr_t)x % 16) != 0 || ((uintptr_t)w_hi % 16) != 0 || ((uintptr_t)y % 16) != 0 ||
      (residual != nullptr && ((uintptr_t)residual % 16) != 0) || (bias != nullptr && ((uintptr_t)bias % 16) != 0) ||
      (act.mode == 2 && ((uintptr_t)act.mask % 8) != 0)) {
    set_error("%s: shape M=%lld N=%d K=%d needs K %% 32 == 0, N %% 32 == 0 and 16-byte aligned rows", who,
              (long long)M, N, K);
    return UBV_ERR_UNSUPPORTED;
  }
  // the kernel addresses X and W with 32-bit byte offsets
  if (M * ldx * 4 >= (1LL << 32) || (act.x2 != nullptr && M * act.ldx2 * 4 >= (1LL << 32)) || (int64_t)N * ldw * 2 >= (1LL << 32)) {
    set_error("%s: operand of M=%lld rows x ld=%lld is 4 GB or more", who, (long long)M, (long long)ldx);
    return UBV_ERR_UNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  int rc;
  if (dtype == UBV_F32 && gemm_ws_try(x, ldx, w_hi, w_lo, ldw, bias, residual, y, ldy, M, N, K, act, st)) {
    UBV_CHECK_LAUNCH(who);
    return UBV_OK;
  }
  if (dtype == UBV_F32) rc = gemm_nt_launch<true, false, false>(x, ldx, w_hi, w_lo, ldw, bias, residual, y, ldy, M, N, K, act, st);
  else if (dtype == UBV_F16) rc = gemm_nt_launch<false, true, true>(x, ldx, w_hi, nullptr, ldw, bias, residual, y, ldy, M, N, K, act, st);
  else rc = gemm_nt_launch<false, false, true>(x, ldx, w_hi, nullptr, ldw, bias, residual, y, ldy, M, N, K, act, st);
  if (rc != UBV_OK) { set_error("%s: no kernel for N=%d", who, N); return rc; }
  UBV_CHECK_LAUNCH(who);
  return UBV_OK;
}

extern "C" int ubv_gemm_nt(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                           const float* bias, const void* residual, void* y, int64_t ldy, int64_t M, int N,
                           int K, int dtype, void* stream) {
  return gemm_nt_run(x, ldx, w_hi, w_lo, ldw, bias, residual, y, ldy, M, N, K, dtype, ubv::GemmAct{}, stream, "gemm_nt");
}

extern "C" int ubv_gemm_nt_rowbias(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                                   const float* bias, const void* row_bias, int64_t row_period, int64_t row_ld, void* y,
                                   int64_t ldy, int64_t M, int N, int K, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(row_bias != nullptr && row_period > 0 && row_period < (1LL << 31) && M < (1LL << 31) && row_ld >= N &&
                    row_ld % 4 == 0 && ((uintptr_t)row_bias % 16) == 0,
                "gemm_nt_rowbias: row_bias [row_period, N] with a 16-byte aligned leading dimension >= N");
  GemmAct a{};
  a.res_period = row_period; a.res_ld = row_ld;
  return gemm_nt_run(x, ldx, w_hi, w_lo, ldw, bias, row_bias, y, ldy, M, N, K, dtype, a, stream, "gemm_nt_rowbias");
}

extern "C" int ubv_gemm_nt_dual(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k_split, const void* w_hi,
                                const void* w_lo, int64_t ldw, const float* bias, const void* residual,
                                const void* row_bias, int64_t row_period, int64_t row_ld, void* y, int64_t ldy, void* y2,
                                int64_t ldy2, int n_split, int64_t M, int N, int K, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG((x2 == nullptr) == (k_split == 0) && (y2 == nullptr) == (n_split == 0), "gemm_nt_dual: x2 / k_split and y2 / n_split come in pairs");
  UBV_CHECK_ARG(x2 == nullptr || (k_split > 0 && k_split < K && k_split % 32 == 0 && ldx2 % 4 == 0 && ((uintptr_t)x2 % 16) == 0),
                "gemm_nt_dual: k_split must be a multiple of 32 inside K, x2 16-byte aligned rows");
  UBV_CHECK_ARG(y2 == nullptr || (n_split > 0 && n_split < N && n_split % 128 == 0 && ldy2 % 4 == 0 && ((uintptr_t)y2 % 16) == 0),
                "gemm_nt_dual: n_split must be a multiple of 128 inside N, y2 16-byte aligned rows");
  UBV_CHECK_ARG(residual == nullptr || row_bias == nullptr, "gemm_nt_dual: residual or row_bias, not both");
  UBV_CHECK_ARG(row_bias == nullptr || (row_period > 0 && row_period < (1LL << 31) && M < (1LL << 31) && row_ld % 4 == 0 &&
                                        ((uintptr_t)row_bias % 16) == 0), "gemm_nt_dual: bad row_bias");
  UBV_CHECK_ARG(residual == nullptr || y2 == nullptr, "gemm_nt_dual: a full residual needs a single output");
  GemmAct a{};
  a.x2 = x2; a.ldx2 = ldx2; a.k_split = k_split; a.y2 = y2; a.ldy2 = ldy2; a.n_split = n_split;
  if (row_bias != nullptr) { a.res_period = row_period; a.res_ld = row_ld; }
  return gemm_nt_run(x, ldx, w_hi, w_lo, ldw, bias, row_bias != nullptr ? row_bias : residual, y, ldy, M, N, K, UBV_F32,
                     a, stream, "gemm_nt_dual");
}

extern "C" int ubv_gemm_nt_act(const void* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw,
                               const float* bias, void* y, int64_t ldy, int64_t M, int N, int K, int dtype,
                               int act, const void* mask, float p, uint64_t seed, const uint64_t* seed_dev,
                               void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(act == 1 || act == 2, "gemm_nt_act: act must be 1 (relu + dropout) or 2 (masked gradient)");
  UBV_CHECK_ARG(act != 2 || mask != nullptr, "gemm_nt_act: act 2 needs the activation's output as mask");
  UBV_CHECK_ARG(p >= 0.0f && p < 1.0f, "gemm_nt_act: dropout probability %f outside [0, 1)", (double)p);
  GemmAct a{};
  a.mode = act; a.mask = mask; a.seed = seed; a.seed_dev = seed_dev;
  drop_params(p, a.thresh, a.scale);
  return gemm_nt_run(x, ldx, w_hi, w_lo, ldw, bias, nullptr, y, ldy, M, N, K, dtype, a, stream, "gemm_nt_act");
}

namespace ubv {

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[N, K] = sum_m dY[m, n] X[m, k],   db[n] = sum_m dY[m, n]
// — a product that REDUCES over the M = bs x 40 000 rows into a 256 x 256 .. 512 x 256 output.  Both
// operands are row-major with the reduction index outermost, the opposite of what MFMA fragments want
// (8 consecutive reduction steps per lane).  The transposition is done by the LDS read:
// ds_read_b64_tr_b16 hands lane l of a 16-lane group column l of a [4 rows][16 columns] block whose
// 16 pieces of 4 columns the lanes address (probed in tools/ubench/tr16_probe.hip).  So a chunk of
// 64 rows is staged row-major, as it arrives — 16-byte loads, 16-byte LDS stores, no shuffling — in
// column groups: T[column / 16][row][16] (+ 32 bytes per group, which puts the 8 groups of a row on 8
// different bank octets for the stores and groups c / c + 4 on opposite bank halves for the reads),
// and a fragment is two transposing reads.  An MFMA block takes its 32 rows / columns from groups
// (c, c + 4): the tile's rows are permuted, which only the epilogue has to know.
// f32 data is split into bf16 halves on the way in (dY_hi X_hi + dY_hi X_lo + dY_lo X_hi, f32
// accumulation), 16-bit data takes one product.  db comes from the same dY fragments and an all-ones
// A operand (one more MFMA per column block in the blocks of k tile 0): no VALU work at all.
//
// A block owns a 128 x 128 tile of dW for one slab of rows (split-K over M); its 4 waves hold 64 x 64
// each in registers.  Roles are swapped (A = X^T fragment, B = dY^T fragment) so that a lane ends up
// with 4 consecutive k of one output row n: 16-byte stores.  The slabs' partial tiles go to
// `partials` [S][N*K + N] (f32; the last N entries of a slab are its bias sums), summed by
// slab_sum_kernel — one read of dY and X instead of the strided-batched library GEMM + a separate
// column-sum pass.
constexpr int kWgTile = 128, kWgMC = 64;
#ifndef UBV_WGRAD_SPREAD
#define UBV_WGRAD_SPREAD 0
#endif
constexpr int kWgCS = kWgMC * 16 + 16;                    // halves per column group (64 rows x 16 + 32 B)
constexpr int kWgPlane = 8 * kWgCS;                       // halves per operand plane (128 columns)

typedef short gi16x4_t __attribute__((ext_vector_type(4)));

// 8 reduction steps (rows m0 .. m0 + 7 of this lane's half) of one column: two transposing reads
__device__ __forceinline__ uint4 wg_frag(const uint16_t* p) {
  typedef __attribute__((address_space(3))) gi16x4_t lds_v4;
  const gi16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)p);
  const gi16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 4 * 16));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return make_uint4(ua.x, ua.y, ub.x, ub.y);
}

// GATHER (the sparse convolutions' weight gradient, sparse_conv.hip): X rows are taken through an index map,
// x_row(m) = xidx[kk * xld + m] (-1: a zero row), one 128 x 128 tile per kernel offset kk — the "tiles" of a
// slab are then the kvol offsets, which share the slab's rows of dY through one XCD's L2.
template <bool SPLIT, bool F16, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_kernel(const void* __restrict__ dYv, const void* __restrict__ Xv,
                                                            float* __restrict__ partials, long M, int N, int K,
                                                            int tiles_k, int tiles, int splits, int rows_per_split,
                                                            const int32_t* __restrict__ xidx, long xld,
                                                            const int32_t* __restrict__ yidx,
                                                            const int32_t* __restrict__ cnt,
                                                            const void* __restrict__ dY2v, int n_split,
                                                            int kvol, int cwsh) {
  // kvol / cwsh (GATHER with pairs): narrow layers PACK several kernel offsets into one 128-wide tile — offset p of
  // the block's 128 >> cwsh takes columns [p << cwsh, (p + 1) << cwsh) of BOTH operand planes, each thread gathers the
  // rows of ITS offset's pair list, and of the 128 x 128 product only the diagonal blocks (same offset on both sides)
  // are stored.  A 64-channel layer then runs 14 tiles instead of 27 with every load used (a quarter of a one-offset
  // tile was data, the rest clamped zero loads), a 32-channel layer 7, a 16-channel layer 4.  cwsh = 7: one offset.
  // dY2v / n_split (f32 only, ubv_gemm_wgrad_dual): columns n >= n_split of dY live in a second matrix [M, N - n_split]
  // and the first one is [M, n_split] — the fused value_proj | offsets | logits weight gradient of the BEV self-attention
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  uint16_t* ty_h = lds;
  uint16_t* tx_h = ty_h + kWgPlane;
  uint16_t* ty_l = tx_h + kWgPlane;                       // (SPLIT only)
  uint16_t* tx_l = ty_l + kWgPlane;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // 1-D grid, XCD-aware: block ids go round-robin over the 8 XCDs; the tiles of one slab take consecutive
  // slots of ONE XCD, so the slab's rows of dY and X (each read by 2 - 4 tiles) come from HBM once
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
  if (split >= splits) return;
  const int tn = GATHER ? 0 : tile / tiles_k, tk = GATHER ? 0 : tile - tn * tiles_k;
  const int n0 = tn * kWgTile, k0 = tk * kWgTile;
  const int pack = GATHER ? (128 >> cwsh) : 1, cmask = (1 << cwsh) - 1;
  // GATHER with compacted pairs (ubv_spconv_wgrad_pairs): row i < cnt[kk] of offset kk multiplies
  // dY[yidx[i]] with X[xidx[i]]; rows past the count hold no pair — slabs beyond it store a zero tile at once
  const long mbeg = (long)split * rows_per_split;
  const long slab_end = mbeg + rows_per_split < M ? mbeg + rows_per_split : M;
  long mend = slab_end;
  // (slabs keep their fixed row ranges: dealing each offset's pairs evenly to the slabs measured slower, 8.5 vs 7.5 ms
  //  per encoder backward — the offsets of one slab then walk different rows of dY and stop sharing them through L2)
  if (GATHER && cnt != nullptr) {
    long most = 0;
    for (int q = 0; q < pack; ++q) {
      const int kq = tile * pack + q;
      if (kq < kvol) most = most > (long)cnt[kq] ? most : (long)cnt[kq];
    }
    mend = mend < most ? mend : most;
  }
  const int wk = wv >> 1, wn = wv & 1;                    // wave: k groups {wk, wk+2} (+4), n groups {wn, wn+2} (+4)
  const bool want_bias = !GATHER && tk == 0 && wk == 0;

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

  // ---- staging maps.  16-bit: thread t <-> columns 8 (t % 16) .. + 7 of rows t / 16 + 16 i (i < 4);
  // f32: columns 4 (t % 32) .. + 3 of rows t / 32 + 8 i (i < 8).  Rows past the slab and columns past
  // N / K are clamped for the load and zeroed by a select (straight-line loads, see gemm_nt_kernel).
  constexpr int NI = SPLIT ? 8 : 4;                       // loads per operand per thread per chunk
  constexpr int CW = SPLIT ? 4 : 8;                       // columns per load
  constexpr int TPR = 128 / CW;                           // threads per row
  const int sc = (tid % TPR) * CW, sr = tid / TPR;        // column within the tile, first row
  // this thread's kernel offset (GATHER): its pair lists, its own row count
  const int mykk = GATHER ? tile * pack + (sc >> cwsh) : 0;
  const bool kk_ok = !GATHER || mykk < kvol;
  const int cl = GATHER ? (sc & cmask) : sc;              // column inside the offset's block
  if (GATHER) {
    xidx += (long)(kk_ok ? mykk : 0) * xld;
    if (yidx != nullptr) yidx += (long)(kk_ok ? mykk : 0) * xld;
  }
  long my_end = mend;                                     // rows of THIS thread's operand columns
  if (GATHER && cnt != nullptr) {
    const long c = kk_ok ? (long)cnt[kk_ok ? mykk : 0] : 0;
    my_end = slab_end < c ? slab_end : c;
  }
  const bool yok = kk_ok && n0 + cl < N, xok = kk_ok && k0 + cl < K;        // N, K multiples of CW (host check)
  const long ycol = yok ? n0 + cl : 0, xcol = xok ? k0 + cl : 0;
  const int soff = (sc >> 4) * kWgCS + (sc & 15);         // LDS offset of (row 0, column sc)
  gf32x4_t fy[SPLIT ? NI : 1], fx[SPLIT ? NI : 1];
  gu32x4_t hy[SPLIT ? 1 : NI], hx[SPLIT ? 1 : NI];
  // GATHER: the pair indices of a chunk are fetched ONE CHUNK BEFORE its rows (pidx holds the next chunk's), so a
  // chunk's rows go out without waiting for an index load — index -> row was two round trips in series per 64-row
  // chunk, with 48 MFMAs per wave to hide them behind (MFMA busy 18 % at 128 channels)
  int pix[GATHER ? NI : 1], piy[GATHER ? NI : 1];
  // Round 5 (from csrc/gemm_wgrad_ws.inl): rows are counted from the slab's first row in 32 bits and an address is ONE
  // v_mad_u64_u32 (row x pitch + base) — the 64-bit compares / selects / multiplies written here before were 22
  // instructions per pair of loads, four of them quarter-rate multiplies.
  constexpr int ES = SPLIT ? 4 : 2;                       // bytes per element
  const long own_end = my_end > mbeg ? my_end : mbeg;
  const int lim = (int)(own_end - mbeg) - 1;              // this thread's last row, relative (-1: none)
  const bool second = SPLIT && n_split > 0 && n0 >= n_split;       // dual dY: block-uniform (n_split % 128 == 0)
  const char* ybase = (const char*)(second ? dY2v : dYv);
  const long yld = SPLIT && n_split > 0 ? (second ? N - n_split : n_split) : N;
  const long ycol2 = (second && yok) ? ycol - n_split : ycol;
  const char* ysl = ybase + ((GATHER ? 0L : mbeg * yld) + ycol2) * ES;
  const char* xsl = (const char*)Xv + ((GATHER ? 0L : mbeg * (long)K) + xcol) * ES;
  const uint32_t ypitch = (uint32_t)(yld * ES), xpitch = (uint32_t)((long)K * ES);
  const int32_t* xlist = GATHER ? xidx + mbeg : nullptr;
  const int32_t* ylist = (GATHER && yidx != nullptr) ? yidx + mbeg : nullptr;
  auto load_idx = [&](long mc) {
    if constexpr (GATHER) {
      const int crel = (int)(mc - mbeg);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int r = crel + sr + (256 / TPR) * i;
        r = r < lim ? r : lim;
        r = r > 0 ? r : 0;
        // (unconditional loads — a conditional one put a branch and a wait in front of the prefetch: 148 -> 221 us per
        //  launch.  An offset with no pair in this slab reads the slab's first entry of its list, which ubv_spconv_pairs
        //  leaves unwritten past the count: load_chunk never turns an index of a row past my_end into an address)
        pix[i] = xlist[r];
        piy[i] = ylist != nullptr ? ylist[r] : (int)mbeg + r;
      }
    }
  };
  // Rows past the slab and columns past N / K are clamped for the load and ZEROED WHEN THE CHUNK IS STORED TO LDS
  // (ymask / xmask: one bit per load).  Zeroing them right behind the load — rounds 2 - 4 — made the compiler wait for
  // every load of the NEXT chunk (vmcnt 14 .. 0) before the first MFMA of the current one: the prefetch ran in series
  // with the arithmetic.
  uint32_t ymask = 0u, xmask = 0u;
  auto load_piece = [&](long mc, int i) __attribute__((always_inline)) {
    {
      const int rel = (int)(mc - mbeg) + sr + (256 / TPR) * i;
      const bool ok = rel <= lim;
      int r = rel < lim ? rel : lim;
      r = r > 0 ? r : 0;                                   // (an offset with no pair in this slab: the slab's row 0, dropped)
      int xr = r, yr = r;
      bool xrow_ok = ok;
      if constexpr (GATHER) {
        xrow_ok = ok && pix[i] >= 0;                       // rel <= lim: only then are the indices defined
        xr = xrow_ok ? pix[i] : 0;
        yr = (ok && piy[i] >= 0) ? piy[i] : 0;
      }
      if constexpr (SPLIT) {
        fy[i] = *reinterpret_cast<const gf32x4_t*>(ysl + (uint64_t)(uint32_t)yr * ypitch);
        fx[i] = *reinterpret_cast<const gf32x4_t*>(xsl + (uint64_t)(uint32_t)xr * xpitch);
      } else {
        hy[i] = *reinterpret_cast<const gu32x4_t*>(ysl + (uint64_t)(uint32_t)yr * ypitch);
        hx[i] = *reinterpret_cast<const gu32x4_t*>(xsl + (uint64_t)(uint32_t)xr * xpitch);
      }
      ymask |= (ok && yok) ? (1u << i) : 0u;
      xmask |= (xrow_ok && xok) ? (1u << i) : 0u;
    }
  };
  auto load_chunk = [&](long mc) __attribute__((always_inline)) {
    ymask = 0u; xmask = 0u;
#pragma unroll
    for (int i = 0; i < NI; ++i) load_piece(mc, i);
  };
  auto split_store = [&](uint16_t* th, uint16_t* tl, int o, const gf32x4_t v) {
    const uint32_t h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - __uint_as_float(h0 << 16), r1 = v.y - __uint_as_float(h0 & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(h1 << 16), r3 = v.w - __uint_as_float(h1 & 0xffff0000u);
    *reinterpret_cast<uint2*>(th + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(tl + o) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
  };
  // masked = false: every piece of every lane of the wave is data (all chunks but a slab's last, whole tiles) — the
  // four selects per piece, a fifth of the split's instructions, are skipped under a wave-uniform branch
  // A thread whose columns lie past N / K (the 96 columns of a 128-wide tile: sampling_offsets | attention_weights, the last
  // tile of the fused 352-wide product) holds zeros in its LDS slots from the start and skips its stores on the unmasked
  // path — its column tail alone used to send the whole wave down the masked one in every chunk.
  auto store_pieces = [&](auto masked) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int o = soff + (sr + (256 / TPR) * i) * 16;
      const bool ky = !MASKED || ((ymask >> i) & 1u), kx = !MASKED || ((xmask >> i) & 1u);
      if constexpr (SPLIT) {
        if (MASKED || yok) split_store(ty_h, ty_l, o, ky ? fy[i] : gf32x4_t{0.f, 0.f, 0.f, 0.f});
        if (MASKED || xok) split_store(tx_h, tx_l, o, kx ? fx[i] : gf32x4_t{0.f, 0.f, 0.f, 0.f});
      } else {
        if (MASKED || yok) *reinterpret_cast<gu32x4_t*>(ty_h + o) = ky ? hy[i] : gu32x4_t{0u, 0u, 0u, 0u};
        if (MASKED || xok) *reinterpret_cast<gu32x4_t*>(tx_h + o) = kx ? hx[i] : gu32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };
  constexpr uint32_t ALLP = (1u << NI) - 1u;
  auto store_chunk = [&]() {
    if (__all((ymask == ALLP || !yok) && (xmask == ALLP || !xok))) store_pieces(std::false_type{});
    else store_pieces(std::true_type{});
  };
  if (!yok || !xok) {                                     // the column tail's slots: zero, once (ordered by the loop's first barrier)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int o = soff + (sr + (256 / TPR) * i) * 16;
      if constexpr (SPLIT) {
        if (!yok) { *reinterpret_cast<uint2*>(ty_h + o) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(ty_l + o) = make_uint2(0u, 0u); }
        if (!xok) { *reinterpret_cast<uint2*>(tx_h + o) = make_uint2(0u, 0u); *reinterpret_cast<uint2*>(tx_l + o) = make_uint2(0u, 0u); }
      } else {
        if (!yok) *reinterpret_cast<gu32x4_t*>(ty_h + o) = gu32x4_t{0u, 0u, 0u, 0u};
        if (!xok) *reinterpret_cast<gu32x4_t*>(tx_h + o) = gu32x4_t{0u, 0u, 0u, 0u};
      }
    }
  }

  // fragment base of this lane inside an MFMA block whose first column group is c: group (lane >> 4) & 1
  // selects c / c + 4, lane >> 5 the reduction half (rows 8 .. 15), lane & 15 the piece
  const int fbase = ((lane >> 4) & 1) * 4 * kWgCS + (lane >> 5) * 8 * 16 + (lane & 15) * 4;
  const uint32_t one2 = F16 ? 0x3C003C00u : 0x3F803F80u;
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  if (mbeg < mend) {
    load_idx(mbeg);
    load_chunk(mbeg);
    if (mbeg + kWgMC < mend) load_idx(mbeg + kWgMC);
  }
  for (long mc = mbeg; mc < mend; mc += kWgMC) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    // UBV_WGRAD_SPREAD=1 (compile time; round-5 experiment, measured neutral: 70.4 vs 71.2 us hot, 66.3 vs 66.2 cold at
    // 256 x 256, job r5s): the next chunk's 2 NI loads of dense operands issued BETWEEN the MFMA groups of this chunk, NI / 2
    // per K step, instead of one burst in front of them — what took the weight-stationary forward GEMM's tile from 5 700 to
    // 4 400 cycles (csrc/gemm_ws.hip) does nothing here: this kernel's time is its transposing LDS reads and barriers.
    // Also measured and dropped (job r5a1, second run): the wave-specialised kernel's consumer side — fragments of k step
    // s + 1 read before the MFMAs of step s, the bias product as a compile-time variant — 210 -> 234 registers, two-stream
    // step 151.4 -> 150.7 samples/s.
    constexpr bool SPREAD = !GATHER && UBV_WGRAD_SPREAD;
    const bool more = mc + kWgMC < mend;
    if (more) {
      if constexpr (SPREAD) { ymask = 0u; xmask = 0u; }
      else load_chunk(mc + kWgMC);
      if (mc + 2 * kWgMC < mend) load_idx(mc + 2 * kWgMC);
    }
#pragma unroll
    for (int ks = 0; ks < kWgMC; ks += 16) {
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = (wk + 2 * i) * kWgCS + fbase + ks * 16;
        ah[i] = wg_frag(tx_h + o);
        if constexpr (SPLIT) al[i] = wg_frag(tx_l + o);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = (wn + 2 * j) * kWgCS + fbase + ks * 16;
        bh[j] = wg_frag(ty_h + o);
        if constexpr (SPLIT) bl[j] = wg_frag(ty_l + o);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = gemm_mma<F16>(ah[i], bh[j], acc[i][j]);
          if constexpr (SPLIT) {
            acc[i][j] = gemm_mma<F16>(ah[i], bl[j], acc[i][j]);
            acc[i][j] = gemm_mma<F16>(al[i], bh[j], acc[i][j]);
          }
        }
      if (want_bias) {                                    // column sums of dY: an all-ones A operand
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accb[j] = gemm_mma<F16>(ones, bh[j], accb[j]);
          if constexpr (SPLIT) accb[j] = gemm_mma<F16>(ones, bl[j], accb[j]);
        }
      }
      if constexpr (SPREAD) {
        constexpr int PER = NI / (kWgMC / 16);            // pieces per K step
        __builtin_amdgcn_sched_barrier(0x7);              // (ALU may cross; MFMA, LDS and vector-memory instructions not)
        if (more) {
#pragma unroll
          for (int q = 0; q < PER; ++q) load_piece(mc + kWgMC, (ks / 16) * PER + q);
        }
        __builtin_amdgcn_sched_barrier(0x7);
      }
    }
  }
  // ---- partial tile.  D[k][n]: column u = lane & 31 of block j, rows u' = (r & 3) + 8 (r >> 2) + 4 half of
  // block i; block-local index u -> tile index 16 (c + 4 (u >> 4)) + (u & 15) with c the block's first group
  float* part = partials + ((long)split * (GATHER ? kvol : 1)) * ((long)N * K + N);
  const int half = lane >> 5, u = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nt = 16 * (wn + 2 * j + 4 * (u >> 4)) + (u & 15);             // column index inside the tile
    const int n = GATHER ? (nt & cmask) : n0 + nt;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int uk = 8 * g + 4 * half;                  // 4 consecutive rows, inside one 16-group
        const int kt = 16 * (wk + 2 * i + 4 * (uk >> 4)) + (uk & 15);
        const int k = GATHER ? (kt & cmask) : k0 + kt;
        if (k >= K) continue;                             // K is a multiple of 4 (host check)
        if constexpr (GATHER) {                           // diagonal blocks only: the same offset on both sides
          const int pn = nt >> cwsh;
          if (pn != (kt >> cwsh) || tile * pack + pn >= kvol) continue;
          *reinterpret_cast<float4*>(part + (long)(tile * pack + pn) * ((long)N * K + N) + (long)n * K + k) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          continue;
        }
        *reinterpret_cast<float4*>(part + (long)n * K + k) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
    }
    if (want_bias && half == 0) part[(long)N * K + n] = accb[j][0];   // every row of the ones product is the sum
  }
}

// out[i] = sum over the S slabs of partials[s][i] (i < len, len % 4 == 0).  A block is 32 columns of 4
// elements x 8 slab groups: every thread sums S / 8 slabs with independent 16-byte loads, the groups
// meet in LDS.  (One thread walking all S slabs — the shape of ubv_linear_grad_reduce, built for 32
// slabs — is a 128-deep chain of dependent loads on 16 blocks.)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, int S, long len,
                                                       float* __restrict__ out) {
  __shared__ float4 red[8][32];
  const int c = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const long i = ((long)blockIdx.x * 32 + c) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < len) {
    for (int s0 = gq; s0 < S; s0 += 32) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int sl = s0 + 8 * u;
        v[u] = sl < S ? *reinterpret_cast<const float4*>(part + (long)sl * len + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
  }
  red[gq][c] = a;
  __syncthreads();
  if (gq == 0 && i < len) {
#pragma unroll
    for (int g = 1; g < 8; ++g) { const float4 b = red[g][c]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *reinterpret_cast<float4*>(out + i) = a;
  }
}

#include "gemm_wgrad_ws.inl"

// Which weight-gradient kernel runs: 0 (default) the 4-wave kernel, two blocks per CU; 4 / 8 the wave-specialised one with
// that many producer waves, one block per CU.  Measured (profiles/r05_wgrad_ws.txt): alone the wave-specialised kernel was
// 8 - 10 % faster than round 4's 4-wave kernel (80 000 x 256 x 256 cold: 65.7 -> 58.8 us), in the one-stream step + 1.6 %,
// in the two-stream step — the default — - 0.6 %: a 133 KB block owns its CU and the other stream's kernels lose the
// co-residency they had.  Its producer-side savings (32-bit row arithmetic, unmasked store path) then went into the
// 4-wave kernel: 65 - 67 -> 62.4 us cold, 512 x 256 106 - 110 -> 97, two-stream step + 0.9 % (job r5a1).
// UBV_WGRAD_WS=0 / 1 and UBV_WGRAD_PW=4 / 8 set the start value, ubv_debug_set_wgrad_ws changes it (tests).
static int g_wgrad_ws = -1;
static int wgrad_ws_mode() {
  if (g_wgrad_ws < 0) {
    const int on = getenv("UBV_WGRAD_WS") ? atoi(getenv("UBV_WGRAD_WS")) : 0;
    const int pw = getenv("UBV_WGRAD_PW") ? atoi(getenv("UBV_WGRAD_PW")) : 8;
    g_wgrad_ws = on == 0 ? 0 : (pw == 4 ? 4 : 8);
  }
  return g_wgrad_ws;
}
static bool wgrad_ws_on() { return wgrad_ws_mode() != 0; }
// The sparse convolutions' weight gradient (ubv_spconv_wgrad_pairs): same choice, its own switch
// (UBV_SPCONV_WGRAD_WS=0 / 4 / 8, default 0).  It ran on the wave-specialised kernel for a few sessions (the LiDAR front
// end is outside the two-stream region: middle encoder 10.25 -> 9.96 ms); since the 4-wave kernel took over that
// kernel's row arithmetic and unmasked store path the two are level there too (10.01 - 10.10 vs 10.06 - 10.30 ms, job r5a2).
static int g_spwg_ws = -1;
static int spwg_ws_mode() {
  if (g_spwg_ws < 0) {
    const int v = getenv("UBV_SPCONV_WGRAD_WS") ? atoi(getenv("UBV_SPCONV_WGRAD_WS")) : 0;
    g_spwg_ws = v == 0 ? 0 : (v == 4 ? 4 : 8);
  }
  return g_spwg_ws;
}

template <typename Kern> static void wgrad_ws_lds(Kern kern, size_t lds) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace ubv

extern "C" int ubv_debug_set_wgrad_ws(int dense, int sparse) {
  using namespace ubv;
  auto ok = [](int v) { return v == -1 || v == 0 || v == 4 || v == 8; };
  UBV_CHECK_ARG(ok(dense) && ok(sparse), "set_wgrad_ws: -1 (keep), 0 (4-wave kernel), 4 or 8 producer waves, got %d, %d",
                dense, sparse);
  if (dense >= 0) g_wgrad_ws = dense;
  if (sparse >= 0) g_spwg_ws = sparse;
  return UBV_OK;
}

extern "C" int ubv_gemm_wgrad_splits(int64_t M, int N, int K) {
  const int tiles = ((N + ubv::kWgTile - 1) / ubv::kWgTile) * ((K + ubv::kWgTile - 1) / ubv::kWgTile);
  static const int blocks_env = getenv("UBV_WGRAD_BLOCKS") ? atoi(getenv("UBV_WGRAD_BLOCKS")) : 0;   // study knob
  // one-tile products (C = 128 configurations: 128 x 128) take one block per CU: with two, a block's five chunks are
  // as much prologue + 64 KB partial tile as product (cfg5 f32: 245.3 -> 247.8 samples/s, job r5c6)
  const int blocks = blocks_env > 0 ? blocks_env : ((ubv::wgrad_ws_on() || tiles == 1) ? 256 : 512);
  long s = (blocks + tiles - 1) / tiles;                  // one 8-wave / two 4-wave blocks per CU
  const long max_s = (M + 255) / 256;                     // at least 256 rows per split
  if (s > max_s) s = max_s;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

static int gemm_wgrad_run(const void* grad_out, const void* grad_out2, int n_split, const void* x, float* partials,
                          float* grad_wb, int64_t M, int N, int K, int splits, int dtype, void* stream);

extern "C" int ubv_gemm_wgrad(const void* grad_out, const void* x, float* partials, float* grad_wb, int64_t M,
                              int N, int K, int splits, int dtype, void* stream) {
  return gemm_wgrad_run(grad_out, nullptr, 0, x, partials, grad_wb, M, N, K, splits, dtype, stream);
}

extern "C" int ubv_gemm_wgrad_dual(const void* grad_out, const void* grad_out2, int n_split, const void* x,
                                   float* partials, float* grad_wb, int64_t M, int N, int K, int splits, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(grad_out2 != nullptr && n_split > 0 && n_split < N && n_split % 128 == 0 && (N - n_split) % 4 == 0 &&
                    ((uintptr_t)grad_out2 % 16) == 0, "gemm_wgrad_dual: n_split must be a multiple of 128 inside N");
  return gemm_wgrad_run(grad_out, grad_out2, n_split, x, partials, grad_wb, M, N, K, splits, UBV_F32, stream);
}

static int gemm_wgrad_run(const void* grad_out, const void* grad_out2, int n_split, const void* x, float* partials,
                          float* grad_wb, int64_t M, int N, int K, int splits, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(grad_out && x && partials && grad_wb && M > 0 && N > 0 && K > 0 && splits > 0,
                "gemm_wgrad: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "gemm_wgrad: unknown dtype %d", dtype);
  const int cw = dtype == UBV_F32 ? 4 : 8;                // columns per 16-byte load
  if (N % cw != 0 || K % cw != 0 || ((uintptr_t)grad_out % 16) != 0 || ((uintptr_t)x % 16) != 0 ||
      ((uintptr_t)partials % 16) != 0 || ((uintptr_t)grad_wb % 16) != 0) {
    set_error("gemm_wgrad: N=%d, K=%d must be multiples of %d with 16-byte aligned buffers", N, K, cw);
    return UBV_ERR_UNSUPPORTED;
  }
  const int tiles_k = (K + kWgTile - 1) / kWgTile, tiles_n = (N + kWgTile - 1) / kWgTile;
  long rps = (M + splits - 1) / splits;
  rps = (rps + kWgMC - 1) / kWgMC * kWgMC;                // whole chunks (and even row pairs)
  const int tiles = tiles_n * tiles_k;
  const dim3 grid((unsigned)((splits + 7) / 8 * 8 * tiles)), blk(256);
  const size_t lds = (size_t)(dtype == UBV_F32 ? 4 : 2) * kWgPlane * sizeof(uint16_t);
  hipStream_t st = as_stream(stream);
  if (wgrad_ws_on()) {
    rps = (rps + 2 * kWgMC - 1) / (2 * kWgMC) * (2 * kWgMC);   // an even number of chunks per slab
    static const int abl = getenv("UBV_WGRAD_ABL") ? atoi(getenv("UBV_WGRAD_ABL")) : 0;   // timing study (wrong results)
    if (dtype == UBV_F32 && abl != 0) {
#define UBV_WG_ABL(A) { wgrad_ws_lds(gemm_wgrad_ws_kernel<true, false, false, A>, 2 * lds); \
      hipLaunchKernelGGL((gemm_wgrad_ws_kernel<true, false, false, A>), grid, dim3(512), 2 * lds, st, grad_out, x, partials, (long)M, N, K, tiles_k, tiles, splits, (int)rps, (const int32_t*)nullptr, 0L, (const int32_t*)nullptr, (const int32_t*)nullptr, grad_out2, n_split, 1, 7); }
      if (abl == 1) UBV_WG_ABL(1) else if (abl == 2) UBV_WG_ABL(2) else if (abl == 3) UBV_WG_ABL(3)
      else if (abl == 4) UBV_WG_ABL(4) else if (abl == 5) UBV_WG_ABL(5) else if (abl == 6) UBV_WG_ABL(6) else UBV_WG_ABL(7)
#undef UBV_WG_ABL
    } else {
    const int pw = wgrad_ws_mode();                     // producer waves: 4 or 8
#define UBV_WG_GO(S, H, P) { wgrad_ws_lds(gemm_wgrad_ws_kernel<S, H, false, 0, P>, 2 * lds); \
      hipLaunchKernelGGL((gemm_wgrad_ws_kernel<S, H, false, 0, P>), grid, dim3(256 + 64 * P), 2 * lds, st, grad_out, x, partials, (long)M, N, K, tiles_k, tiles, splits, (int)rps, (const int32_t*)nullptr, 0L, (const int32_t*)nullptr, (const int32_t*)nullptr, grad_out2, n_split, 1, 7); }
    if (dtype == UBV_F32) { if (pw == 8) UBV_WG_GO(true, false, 8) else UBV_WG_GO(true, false, 4) }
    else if (dtype == UBV_F16) { if (pw == 8) UBV_WG_GO(false, true, 8) else UBV_WG_GO(false, true, 4) }
    else { if (pw == 8) UBV_WG_GO(false, false, 8) else UBV_WG_GO(false, false, 4) }
#undef UBV_WG_GO
    }
  } else
  if (dtype == UBV_F32)
    hipLaunchKernelGGL((gemm_wgrad_kernel<true, false, false>), grid, blk, lds, st, grad_out, x, partials, (long)M, N, K, tiles_k, tiles, splits, (int)rps, (const int32_t*)nullptr, 0L, (const int32_t*)nullptr, (const int32_t*)nullptr, grad_out2, n_split, 1, 7);
  else if (dtype == UBV_F16)
    hipLaunchKernelGGL((gemm_wgrad_kernel<false, true, false>), grid, blk, lds, st, grad_out, x, partials, (long)M, N, K, tiles_k, tiles, splits, (int)rps, (const int32_t*)nullptr, 0L, (const int32_t*)nullptr, (const int32_t*)nullptr, grad_out2, n_split, 1, 7);
  else
    hipLaunchKernelGGL((gemm_wgrad_kernel<false, false, false>), grid, blk, lds, st, grad_out, x, partials, (long)M, N, K, tiles_k, tiles, splits, (int)rps, (const int32_t*)nullptr, 0L, (const int32_t*)nullptr, (const int32_t*)nullptr, grad_out2, n_split, 1, 7);
  const long len = (long)N * K + N;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((len / 4 + 31) / 32)), dim3(256), 0, st, partials, splits, len,
                     grad_wb);
  UBV_CHECK_LAUNCH("gemm_wgrad");
  return UBV_OK;
}

// Weight gradient of a sparse convolution: for every kernel offset k, dW_k[Cout, Cin] = sum_rows grad_out[row, :]^T .
// feats[nbr[k][row], :] (rows without a neighbour contribute nothing) — gemm_wgrad_kernel with gathered X rows, one
// 128 x 128 tile per offset, split-K over the rows.  partials [splits, kvol, Cout*Cin + Cout] f32 scratch;
// grad_w [kvol, Cout*Cin + Cout] f32, WRITTEN (the last Cout entries of each block are unused).
extern "C" int ubv_spconv_wgrad_splits(int64_t rows, int kvol) {
  static const int blocks_env = getenv("UBV_SPCONV_WGRAD_BLOCKS") ? atoi(getenv("UBV_SPCONV_WGRAD_BLOCKS")) : 768;   // study knob
  long s = (blocks_env + kvol - 1) / kvol;
  const long max_s = (rows + 255) / 256;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

static int spconv_wgrad_run(const void* grad_out, const void* feats, const int32_t* nbr, const int32_t* out_rows,
                            const int32_t* counts, int64_t ld, int64_t rows, float* partials, float* grad_w, int Cout,
                            int Cin, int kvol, int splits, int dtype, void* stream);

extern "C" int ubv_spconv_wgrad(const void* grad_out, const void* feats, const int32_t* nbr, int64_t ld, int64_t rows,
                                float* partials, float* grad_w, int Cout, int Cin, int kvol, int splits, int dtype,
                                void* stream) {
  return spconv_wgrad_run(grad_out, feats, nbr, nullptr, nullptr, ld, rows, partials, grad_w, Cout, Cin, kvol, splits,
                          dtype, stream);
}

extern "C" int ubv_spconv_wgrad_pairs(const void* grad_out, const void* feats, const int32_t* in_rows,
                                      const int32_t* out_rows, const int32_t* counts, int64_t ld, int64_t rows,
                                      float* partials, float* grad_w, int Cout, int Cin, int kvol, int splits,
                                      int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(out_rows && counts, "spconv_wgrad_pairs: bad arguments");
  return spconv_wgrad_run(grad_out, feats, in_rows, out_rows, counts, ld, rows, partials, grad_w, Cout, Cin, kvol,
                          splits, dtype, stream);
}

static int spconv_wgrad_run(const void* grad_out, const void* feats, const int32_t* nbr, const int32_t* out_rows,
                            const int32_t* counts, int64_t ld, int64_t rows, float* partials, float* grad_w, int Cout,
                            int Cin, int kvol, int splits, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(grad_out && feats && nbr && partials && grad_w && rows > 0 && ld >= rows && kvol > 0 && splits > 0,
                "spconv_wgrad: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "spconv_wgrad: unknown dtype %d", dtype);
  const int cw = dtype == UBV_F32 ? 4 : 8;
  if (Cout % cw != 0 || Cin % cw != 0 || Cout > kWgTile || Cin > kWgTile || ((uintptr_t)grad_out % 16) != 0 ||
      ((uintptr_t)feats % 16) != 0) {
    set_error("spconv_wgrad: Cout=%d, Cin=%d must be multiples of %d and at most %d", Cout, Cin, cw, kWgTile);
    return UBV_ERR_UNSUPPORTED;
  }
  long rps = (rows + splits - 1) / splits;
  rps = (rps + kWgMC - 1) / kWgMC * kWgMC;
  // narrow layers with compacted pairs: several offsets per 128-wide tile (see the kernel).  UBV_SPCONV_PACK=0: off
  static const int pack_env = getenv("UBV_SPCONV_PACK") ? atoi(getenv("UBV_SPCONV_PACK")) : 1;
  int cwsh = 7;
  if (out_rows != nullptr && counts != nullptr && pack_env != 0) {
    const int cmax = Cout > Cin ? Cout : Cin;
    cwsh = cmax <= 16 ? 4 : cmax <= 32 ? 5 : cmax <= 64 ? 6 : 7;
  }
  const int pack = 128 >> cwsh, ptiles = (kvol + pack - 1) / pack;
  const dim3 grid((unsigned)((splits + 7) / 8 * 8 * ptiles)), blk(256);
  const size_t lds = (size_t)(dtype == UBV_F32 ? 4 : 2) * kWgPlane * sizeof(uint16_t);
  hipStream_t st = as_stream(stream);
  if (spwg_ws_mode() != 0 && out_rows != nullptr && counts != nullptr) {   // (pair lists: see load_idx)
    rps = (rps + 2 * kWgMC - 1) / (2 * kWgMC) * (2 * kWgMC);
    const int pw = spwg_ws_mode();
#define UBV_WG_GO(S, H, P) { wgrad_ws_lds(gemm_wgrad_ws_kernel<S, H, true, 0, P>, 2 * lds); \
      hipLaunchKernelGGL((gemm_wgrad_ws_kernel<S, H, true, 0, P>), grid, dim3(256 + 64 * P), 2 * lds, st, grad_out, feats, partials, (long)rows, Cout, Cin, 1, ptiles, splits, (int)rps, nbr, (long)ld, out_rows, counts, (const void*)nullptr, 0, kvol, cwsh); }
    if (dtype == UBV_F32) { if (pw == 8) UBV_WG_GO(true, false, 8) else UBV_WG_GO(true, false, 4) }
    else if (dtype == UBV_F16) { if (pw == 8) UBV_WG_GO(false, true, 8) else UBV_WG_GO(false, true, 4) }
    else { if (pw == 8) UBV_WG_GO(false, false, 8) else UBV_WG_GO(false, false, 4) }
#undef UBV_WG_GO
  } else
  if (dtype == UBV_F32)
    hipLaunchKernelGGL((gemm_wgrad_kernel<true, false, true>), grid, blk, lds, st, grad_out, feats, partials, (long)rows, Cout, Cin, 1, ptiles, splits, (int)rps, nbr, (long)ld, out_rows, counts, (const void*)nullptr, 0, kvol, cwsh);
  else if (dtype == UBV_F16)
    hipLaunchKernelGGL((gemm_wgrad_kernel<false, true, true>), grid, blk, lds, st, grad_out, feats, partials, (long)rows, Cout, Cin, 1, ptiles, splits, (int)rps, nbr, (long)ld, out_rows, counts, (const void*)nullptr, 0, kvol, cwsh);
  else
    hipLaunchKernelGGL((gemm_wgrad_kernel<false, false, true>), grid, blk, lds, st, grad_out, feats, partials, (long)rows, Cout, Cin, 1, ptiles, splits, (int)rps, nbr, (long)ld, out_rows, counts, (const void*)nullptr, 0, kvol, cwsh);
  const long len = (long)kvol * ((long)Cout * Cin + Cout);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((len / 4 + 31) / 32)), dim3(256), 0, st, partials, splits, len, grad_w);
  UBV_CHECK_LAUNCH("spconv_wgrad");
  return UBV_OK;
}
