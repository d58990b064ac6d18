// Weight-stationary GEMM for the encoder's f32 Linear layers (round 5):
//
//   Y[M, N] = X[M, K] . W[N, K]^T (+ bias[N]) (+ R[M, N] | row-periodic R) (x mask)      f32 data, split-bf16 products
//
// M = bs x 40 000 rows against N, K <= 256..512: the weight is tiny (256 x 256 bf16 hi + lo = 256 KB), the activations
// are the traffic.  gemm_mfma.hip walks K in 32-wide chunks through LDS for BOTH operands and gives every 128 x 128 output
// tile its own block: a block's life is a serial chain of 8 x (load, stage, barrier, MFMA, barrier) + epilogue, the weight
// is re-read from L2 by every block (160 MB of L2 traffic per GEMM) and co-resident blocks move through their phases in
// step (the phases ADD).  Here the weight never moves:
//
//   block  = 8 waves, one per CU, PERSISTENT: grid = #CUs (x column groups), each block streams row tiles of 32 rows;
//   wave w = the 32 output columns [32 w, 32 w + 32) of the block's column group: its slice of W — all K steps, hi and lo
//            halves — sits in 8 K/16 registers as MFMA A-fragments for the whole kernel (128 VGPRs at K = 256); waves
//            beyond N (N = 192, 128) only help to stage X;
//   X tile = 32 rows x K f32.  It travels global -> LDS by LDS-DMA (global_load_lds_dwordx4, issued from inline asm so
//            that hipcc neither drains it at barriers nor needs registers for it) into a ring of TWO raw tiles, two tiles
//            ahead of its use; every thread then reads back the 16-byte pieces IT requested (so a counted
//            `s_waitcnt vmcnt` of its own is all the synchronisation the ring needs), splits them into bf16 hi / lo
//            planes and writes those to a double-buffered MFMA operand tile.  ONE barrier per tile;
//   MFMA   = 3 K/16 v_mfma_f32_32x32x16_bf16 per wave and tile (x_hi w_hi + x_lo w_hi + x_hi w_lo), B-fragments from
//            LDS with two ds_read_b128 per K step (528-byte rows at K = 256: conflict-free);
//   VMEM   = every vector-memory instruction of a tile is issued INSIDE the MFMA loop, one every few MFMAs: the 4 stores
//            of the PREVIOUS tile's results, then the DMA requests of the tile two ahead.  (First version: loads in a
//            burst before the barrier, stores in a burst after the MFMAs.  Cycle stamps — profiles/r05_gemm_ws.txt —
//            showed a wave blocked ~2 500 cycles per tile on the issue of those instructions, the CU's memory pipeline
//            accepts them at its own pace, then 1 950 cycles in its MFMAs while that pipeline drained and idled: the two
//            phases added up, exactly what section 3.6b of DESIGN.md found for the tile-per-block kernel.)
//   output = operand roles swapped (A = W fragment, B = X fragment): a lane ends with ONE row and 4 x 4 consecutive
//            columns, i.e. four 16-byte stores per tile (one column per lane = 16 dword stores per wave and tile cost
//            ~100 cycles of issue each); residual / mask operands are read with the same pattern before the MFMAs, the
//            bias sits in LDS.
// N > 256 runs as column groups of 256 (each group its own persistent blocks, interleaved so that the groups of one row
// tile run on the same XCD at about the same time and share X through its L2).
//
// Wait counts are HAND-COUNTED (the DMA requests are invisible to hipcc): a stage issues exactly E + 4 + PPT vector-memory
// instructions per wave (E = 4 epilogue-operand loads or 0, 4 stores, PPT DMA requests), all of them unconditional, and
// the kernels must not spill (a scratch access is a vector-memory instruction too): tests/test_build_isa.py checks
// `.vgpr_spill_count: 0` for every instantiation.
//
// Scope: f32 data (SPLIT), K in {64, 128, 192, 256}, N % 32 == 0 with N <= 256 or N % 256 == 0, every activation mode of
// GemmAct, the row-periodic and the plain residual.  Everything else (K = 512, the fused self-attention GEMM with its two
// sources / outputs, 16-bit data) stays on gemm_nt_kernel (gemm_mfma.hip).
// UBV_GEMM_WS=0 switches the kernel off (A/B runs).
#include <type_traits>
#include <utility>

#include "gemm_act.h"
#include "ubv_common.h"

namespace ubv {

typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8_t;
typedef __attribute__((ext_vector_type(16))) float wf32x16_t;
typedef __attribute__((ext_vector_type(4))) float wf32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t wu32x4_t;

struct WsArgs {
  const float* X; const uint16_t* Wh; const uint16_t* Wl; const float* bias; const float* R; const float* mask; float* Y;
  long M; uint32_t ldx, ldw, ldy, ldr;
  int N, K, res_period, ncg, tiles;
  float scale; uint32_t thresh; uint64_t seed; const uint64_t* seed_dev;
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <typename F, int... I>
__device__ __forceinline__ void ws_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ws_static_for(F&& f) { ws_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// timing study (UBV_WS_ABL=16): shader-clock stamps of wave 0 of blocks 0 and 100, 8 per tile
__device__ uint64_t ws_dbg[2 * 16 * 8];

// LDS-DMA: 16 bytes per lane from `gsrc` (per lane) to LDS byte address `lds_dst` + 16 lane (`lds_dst` wave-uniform).
// M0 carries the LDS base and is compiler-reserved: written and restored in the same statement.
__device__ __forceinline__ void ws_dma16(const float* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ wf32x16_t ws_mma(wu32x4_t a, wu32x4_t b, wf32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8_t, a), __builtin_bit_cast(wbf16x8_t, b), c, 0, 0, 0);
}

// KS = K / 16; EXTRA: 0 nothing, 1 residual (plain or row-periodic), 2 mask (GemmAct mode 2), 3 ReLU + dropout (mode 1:
// the keep mask of ubv_relu_dropout_forward, one 64-bit mix per 4 consecutive columns); ABL: timing-study switches
// (1 no stores, 2 no MFMAs, 16 cycle stamps).
template <int KS, int EXTRA, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];
  constexpr int K = KS * 16, NT = 512;
  constexpr int PITCH = K * 2 + 16;                       // bytes per row of one operand plane: odd multiple of 16 -> b128 reads conflict-free
  constexpr int PLANE = 32 * PITCH, BUF = 2 * PLANE;      // operand tile = hi plane + lo plane
  constexpr int RAW = 32 * K * 4;                         // raw f32 tile
  constexpr int PIECES = 8 * K;                           // 16-byte pieces of a tile
  constexpr int PPT = PIECES / NT;                        // pieces (= DMA requests) per thread and tile
  static_assert(PIECES % NT == 0, "K must be a multiple of 64");
  constexpr int RP = K / 4;                               // pieces per row
  // LDS map: [2 operand tiles][2 raw tiles][bias, 1 KB]
  constexpr int OFF_RAW = 2 * BUF, OFF_BIAS = OFF_RAW + 2 * RAW;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, fr = lane & 31;
  // block -> (column group, position in the group's tile sequence).  Blocks b, b + 8, ... sit on one XCD: the ncg groups
  // of a tile sequence take consecutive slots there.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cg = slot % a.ncg, seq = (slot / a.ncg) * 8 + xcd;
  const int nseq = gridDim.x / a.ncg;                     // blocks per column group (host: gridDim.x = 8 k ncg)
  const int ncols = min(a.N - cg * 256, 256);
  const bool active = wv * 32 < ncols;                    // (wave-uniform) this wave has columns
  const int nc = cg * 256 + wv * 32 + 4 * half;           // first of this lane's output columns: nc + 8 g + (0..3)
  const int my_tiles = (a.tiles - seq + nseq - 1) / nseq; // tiles seq, seq + nseq, ...
  if (my_tiles <= 0) return;

  // ---- DMA requests of this thread: piece p = tid + 512 j of a tile -> row p / RP, 16-byte column p % RP; it lands at
  // byte 16 p of the raw tile (a wave instruction = 1 KB of consecutive pieces)
  uint32_t poff[PPT];                                     // f32 offset inside a full tile: row * ldx + 4 col
  int prow[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + NT * j;
    prow[j] = p / RP;
    poff[j] = (uint32_t)prow[j] * a.ldx + (uint32_t)(p % RP) * 4u;
  }
  // LDS byte address of the dynamic segment (the local address space's own 32-bit pointer)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ws_lds;
  auto request = [&](int i) __attribute__((always_inline)) {          // all PPT requests of tile i (the prologue's form)
    const long m0 = (seq + (long)min(i, my_tiles - 1) * nseq) * 32;   // (past the end: the last tile again, never used)
    const int rv = (int)min(32L, a.M - m0) - 1;                       // rows past M re-read the last valid one
    const float* xb = a.X + m0 * a.ldx;
    const uint32_t dst = lds0 + OFF_RAW + (i & 1) * RAW + wv * 1024;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const uint32_t o = prow[j] <= rv ? poff[j] : poff[j] - (uint32_t)(prow[j] - rv) * a.ldx;
      ws_dma16(xb + o, dst + j * (NT * 16));
    }
  };

  // ---- the wave's slice of W as MFMA A-fragments, for the whole kernel.  A fragment is 16 bytes of one row per lane: read
  // straight from global memory that is 32 rows x 32 bytes per wave instruction — 64 KB of 128-byte lines for every 16 KB
  // of fragments, 8 waves of it thrashing the L1 — and it made the first tile of every block ~4x as long as a later one
  // (cycle stamps: 17 000 of a call's 57 000 cycles).  So each wave copies its slice (32 rows = 64 K contiguous bytes per
  // half) into LDS by DMA in whole rows — 16-byte chunk c of row r goes to position c ^ (r & SWZ) of that row, the
  // swizzle is applied on the GLOBAL side, LDS-DMA writes linearly — and reads its fragments back conflict-free.  The
  // region is private to the wave (no barrier), the tile buffers are not in use yet.
  wu32x4_t wh[KS], wl[KS];
  if (active) {
    constexpr int CPR = K / 8, SWZ = CPR % 16 == 0 ? 15 : 7;   // 16-byte chunks per row of a half; the XOR must stay inside the row
    const uint32_t wreg = lds0 + wv * (64 * K);
    const unsigned char* const wsrc = ws_lds + wv * (64 * K);
    auto fetch = [&](const uint16_t* Wp, wu32x4_t (&dst)[KS]) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(Wp) + (size_t)(cg * 256 + wv * 32) * a.ldw * 2;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const int q = c * 64 + lane, r = q / CPR, pos = q % CPR;
        ws_dma16(reinterpret_cast<const float*>(base + (size_t)r * a.ldw * 2 + ((pos ^ (r & SWZ)) * 16)), wreg + c * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        dst[ks] = *reinterpret_cast<const wu32x4_t*>(wsrc + (fr * CPR + ((2 * ks + half) ^ (fr & SWZ))) * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the reads are done before the region is written again)
    };
    fetch(a.Wh, wh);
    fetch(a.Wl, wl);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { wh[ks] = wu32x4_t{0u, 0u, 0u, 0u}; wl[ks] = wh[ks]; }
  }
  __syncthreads();                                        // every wave is done with its staging region: the tile buffers are free
  const uint64_t act_seed = a.seed + ((EXTRA == 3 && a.seed_dev != nullptr) ? *a.seed_dev : 0ull);
  // bias of the block's columns in LDS: read back as 4 x float4 per tile
  float* const bias_lds = reinterpret_cast<float*>(ws_lds + OFF_BIAS);
  if (tid < 256) bias_lds[tid] = (a.bias != nullptr && tid < ncols) ? a.bias[cg * 256 + tid] : 0.0f;
  // (the W fragments are in registers before the first DMA request is issued: from here on this wave's vector-memory
  //  instructions are exactly the counted ones)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  request(0);
  request(1);

  // One tile.  ACT: the wave has columns (else it only stages X).  RAG: the tile has rows past M.  FIRST: stage 0 of the
  // block (its wait count differs: only tile 1's requests are newer than tile 0's).
  wf32x4_t pv[4];                                         // the previous tile's results and where they go
  float* pyp;
  {
    // stage 0 has no previous tile: it stores zeros over ITS OWN output rows, which its real results overwrite one
    // stage later (same wave, same addresses, program order).  Rows that exist: min(., M - 1).
    pyp = a.Y + min((long)seq * 32 + fr, a.M - 1) * (long)a.ldy + nc;
#pragma unroll
    for (int g = 0; g < 4; ++g) pv[g] = wf32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
  }
  auto stage = [&](int i, auto actc, auto ragc, auto firstc) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(actc)::value, RAG = decltype(ragc)::value, FIRST = decltype(firstc)::value;
    // vector-memory instructions of this wave issued after tile i's requests: one whole stage (tile i - 1's) — or, for
    // stage 0, tile 1's requests
    constexpr int E = (ACT && (EXTRA == 1 || EXTRA == 2)) ? 4 : 0, PER = E + (ACT ? 4 : 0) + PPT;
    const int buf = i & 1;
    const long m0 = (seq + (long)i * nseq) * 32;
    const bool stamp = (ABL & 16) != 0 && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && i < 16;
    uint64_t* const dbg = ws_dbg + ((blockIdx.x == 0 ? 0 : 1) * 16 + (i & 15)) * 8;
    if (stamp) dbg[0] = __builtin_readcyclecounter();
    ws_wait_vm<FIRST ? PPT : PER>();
    if (stamp) dbg[1] = __builtin_readcyclecounter();
    // this thread's pieces: raw f32 -> bf16 hi / lo planes of operand tile `buf`
    {
      const unsigned char* raw = ws_lds + OFF_RAW + buf * RAW + tid * 16;
      unsigned char* opd = ws_lds + buf * BUF;
      wf32x4_t v[PPT];
#pragma unroll
      for (int j = 0; j < PPT; ++j) v[j] = *reinterpret_cast<const wf32x4_t*>(raw + j * (NT * 16));
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int p = tid + NT * j;
        const uint32_t h0 = cvt_pk_bf16(v[j][0], v[j][1]), h1 = cvt_pk_bf16(v[j][2], v[j][3]);
        const float r0 = v[j][0] - __uint_as_float(h0 << 16), r1 = v[j][1] - __uint_as_float(h0 & 0xffff0000u);
        const float r2 = v[j][2] - __uint_as_float(h1 << 16), r3 = v[j][3] - __uint_as_float(h1 & 0xffff0000u);
        const int o = (p / RP) * PITCH + (p % RP) * 8;
        *reinterpret_cast<uint2*>(opd + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(opd + PLANE + o) = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
      }
    }
    if (stamp) dbg[2] = __builtin_readcyclecounter();
    wf32x4_t ex[4];
    const int rv = RAG ? (int)(a.M - m0) - 1 : 31;         // last valid row of the tile
    const long mrow = m0 + (RAG ? min(fr, rv) : fr);       // this lane's row (rows past M: the last valid one again)
    if constexpr (ACT && (EXTRA == 1 || EXTRA == 2)) {    // epilogue operands of THIS tile, requested before the MFMAs
      const float* src = EXTRA == 2 ? a.mask : a.R;
      const long rr = (EXTRA == 1 && a.res_period > 0) ? (long)((uint32_t)mrow % (uint32_t)a.res_period) : mrow;
      const float* rowp = src + rr * (long)a.ldr + nc;
#pragma unroll
      for (int g = 0; g < 4; ++g) ex[g] = *reinterpret_cast<const wf32x4_t*>(rowp + 8 * g);
    }
    // the requests of the tile two ahead (into the raw tile this thread has just read): addresses ahead of the loop
    const long m0n = (seq + (long)min(i + 2, my_tiles - 1) * nseq) * 32;
    const int rvn = (int)min(32L, a.M - m0n) - 1;
    const float* const xbn = a.X + m0n * a.ldx;
    const uint32_t dstn = lds0 + OFF_RAW + buf * RAW + wv * 1024;
    if (stamp) dbg[3] = __builtin_readcyclecounter();
    __syncthreads();                                      // operand tile `buf` complete; every wave is done with it as of tile i - 2
    if (stamp) dbg[4] = __builtin_readcyclecounter();
    auto dma_slot = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const uint32_t o = prow[j] <= rvn ? poff[j] : poff[j] - (uint32_t)(prow[j] - rvn) * a.ldx;
      ws_dma16(xbn + o, dstn + j * (NT * 16));
    };
    if constexpr (!ACT) {
      ws_static_for<PPT>([&](auto jc) __attribute__((always_inline)) { dma_slot(jc); });
    } else {
      wf32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const unsigned char* ab = ws_lds + buf * BUF + fr * PITCH + half * 16;
      // VMEM slots of the K loop: the 4 stores first, then the PPT requests, spread evenly over the K steps.  A
      // scheduling fence around each slot: ALU instructions may cross it, MFMAs, LDS and vector-memory instructions not.
      constexpr int NSLOT = 4 + PPT, kPin = 0x1 | 0x2 | 0x4;
      ws_static_for<KS>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        const wu32x4_t bh = *reinterpret_cast<const wu32x4_t*>(ab + ks * 32);
        const wu32x4_t bl = *reinterpret_cast<const wu32x4_t*>(ab + PLANE + ks * 32);
        if constexpr (ABL & 2) {
          acc[ks & 15] += __uint_as_float(bh[0] ^ bl[1]);
        } else {
          acc = ws_mma(wh[ks], bh, acc);
          acc = ws_mma(wh[ks], bl, acc);
          acc = ws_mma(wl[ks], bh, acc);
        }
        constexpr int first = ks * NSLOT / KS, last = (ks + 1) * NSLOT / KS;
        ws_static_for<last - first>([&](auto sc) __attribute__((always_inline)) {
          constexpr int s = first + decltype(sc)::value;
          __builtin_amdgcn_sched_barrier(kPin);
          if constexpr (s < 4) {
            if constexpr (ABL & 1) { if (pv[s][0] == 123.456f) *reinterpret_cast<wf32x4_t*>(pyp + 8 * s) = pv[s]; }
            else *reinterpret_cast<wf32x4_t*>(pyp + 8 * s) = pv[s];
          } else {
            dma_slot(std::integral_constant<int, s - 4>{});
          }
          __builtin_amdgcn_sched_barrier(kPin);
        });
      });
      if (stamp) dbg[5] = __builtin_readcyclecounter();
      // D[n][m]: lane = row m = fr of the tile, columns n = nc + 8 g + e for acc[4 g + e].  The values wait in pv for the
      // next stage's store slots.
      const float* const bl4 = bias_lds + wv * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const wf32x4_t b4 = *reinterpret_cast<const wf32x4_t*>(bl4 + 8 * g);
        wf32x4_t v = {acc[4 * g] + b4[0], acc[4 * g + 1] + b4[1], acc[4 * g + 2] + b4[2], acc[4 * g + 3] + b4[3]};
        if constexpr (EXTRA == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ex[g][e] != 0.0f ? v[e] * a.scale : 0.0f;
        }
        if constexpr (EXTRA == 3) {
          // dropout(relu(.)): the group of 4 elements m * ldy + n .. + 3 shares one mix (n % 4 == 0, ldy % 4 == 0)
          const uint64_t mix = a.thresh != 0u ? drop_mix64(act_seed, (uint64_t)(mrow * (long)a.ldy + nc + 8 * g) >> 2) : 0ull;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r = fmaxf(v[e], 0.0f);
            if (a.thresh != 0u) r = drop_keep16(mix, e, a.thresh) ? r * a.scale : 0.0f;
            v[e] = r;
          }
        }
        if constexpr (EXTRA == 1) {                       // (element by element: a vector add is a packed f32 instruction)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ex[g][e];
        }
        pv[g] = v;
      }
      pyp = a.Y + mrow * (long)a.ldy + nc;
    }
    if (stamp) dbg[6] = __builtin_readcyclecounter();
  };

  using Yes = std::true_type;
  using No = std::false_type;
  // the tile with rows past M (at most one in the problem) is this block's LAST tile, if it is this block's at all: it
  // runs after the loop
  const bool ragged = (a.M & 31) != 0 && (a.tiles - 1 - seq) % nseq == 0;
  const int n_loop = my_tiles - (ragged ? 1 : 0);
  auto run = [&](auto actc) __attribute__((always_inline)) {
    if (n_loop > 0) {
      stage(0, actc, No{}, Yes{});
      for (int i = 1; i < n_loop; ++i) stage(i, actc, No{}, No{});
      if (ragged) stage(n_loop, actc, Yes{}, No{});
    } else {
      stage(0, actc, Yes{}, Yes{});                       // (the block's only tile is the ragged one)
    }
    if constexpr (decltype(actc)::value) {                // the last tile's results
      const int rv_last = ragged ? (int)(a.M - (seq + (long)(my_tiles - 1) * nseq) * 32) - 1 : 31;
      if (fr <= rv_last) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<wf32x4_t*>(pyp + 8 * g) = pv[g];
      }
    }
  };
  // (wave-uniform) one loop per kind of wave: their vector-memory instruction counts differ
  if (active) run(Yes{});
  else run(No{});
}

template <int KS>
constexpr int ws_lds_bytes() { return 2 * 2 * 32 * (KS * 32 + 16) + 2 * 32 * KS * 64 + 1024; }

template <int KS, int EXTRA>
static void ws_launch(const WsArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((gemm_ws_kernel<KS, EXTRA>), dim3(grid), dim3(512), ws_lds_bytes<KS>(), st, a);
}

template <int KS>
static void ws_launch_extra(const WsArgs& a, int extra, int grid, hipStream_t st) {
  if (extra == 3) ws_launch<KS, 3>(a, grid, st);
  else if (extra == 2) ws_launch<KS, 2>(a, grid, st);
  else if (extra == 1) ws_launch<KS, 1>(a, grid, st);
  else {
    // timing studies of the main variant (UBV_WS_ABL: 1 no stores, 2 no MFMAs, 16 cycle stamps)
    static const int abl = getenv("UBV_WS_ABL") ? atoi(getenv("UBV_WS_ABL")) : 0;
    if constexpr (KS == 16) {
      if (abl == 1) { hipLaunchKernelGGL((gemm_ws_kernel<KS, 0, 1>), dim3(grid), dim3(512), ws_lds_bytes<KS>(), st, a); return; }
      if (abl == 2) { hipLaunchKernelGGL((gemm_ws_kernel<KS, 0, 2>), dim3(grid), dim3(512), ws_lds_bytes<KS>(), st, a); return; }
      if (abl == 16) { hipLaunchKernelGGL((gemm_ws_kernel<KS, 0, 16>), dim3(grid), dim3(512), ws_lds_bytes<KS>(), st, a); return; }
    }
    ws_launch<KS, 0>(a, grid, st);
  }
}

// true: launched.  false: the shape / epilogue is outside this kernel's scope (the caller takes gemm_nt_kernel).
bool gemm_ws_try(const void* X, long ldx, const void* Wh, const void* Wl, long ldw, const float* bias, const void* R, void* Y,
                 long ldy, long M, int N, int K, const GemmAct& act, hipStream_t st) {
  static const int env = getenv("UBV_GEMM_WS") ? atoi(getenv("UBV_GEMM_WS")) : 1;
  if (env == 0) return false;
  if (M < 1 || act.mode < 0 || act.mode > 2 || (act.mode != 0 && R != nullptr)) return false;
  if (act.x2 != nullptr || act.y2 != nullptr || act.n_split != 0) return false;
  if (K != 256 && K != 192 && K != 128 && K != 64) return false;
  if (N % 32 != 0 || (N > 256 && N % 256 != 0)) return false;
  if (ldx >= (1L << 26) || ldy >= (1L << 26) || ldw >= (1L << 31) || (act.res_period > 0 && act.res_ld >= (1L << 31))) return false;
  // stricter than gemm_nt_run's argument checks: this kernel reads the residual / mask rows and both weight halves as
  // 16-byte vectors and moves X by 16-byte DMA — a sliced or offset operand stays on gemm_nt_kernel
  {
    const long ldr_eff = act.res_period > 0 ? act.res_ld : ldy;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (!al16(X) || !al16(Wh) || !al16(Wl) || !al16(Y) || (R != nullptr && !al16(R)) || (act.mode == 2 && !al16(act.mask))) return false;
    if (ldx % 4 != 0 || ldy % 4 != 0 || ldw % 8 != 0 || ((R != nullptr || act.mode == 2) && ldr_eff % 4 != 0)) return false;
  }
  WsArgs a{};
  a.X = (const float*)X; a.Wh = (const uint16_t*)Wh; a.Wl = (const uint16_t*)Wl;
  a.bias = bias; a.R = (const float*)R; a.mask = (const float*)act.mask; a.Y = (float*)Y;
  a.M = M; a.ldx = (uint32_t)ldx; a.ldw = (uint32_t)ldw; a.ldy = (uint32_t)ldy;
  a.ldr = (uint32_t)(act.res_period > 0 ? act.res_ld : ldy);
  a.N = N; a.K = K; a.res_period = (int)act.res_period;
  a.ncg = (N + 255) / 256; a.tiles = (int)((M + 31) / 32); a.scale = act.scale;
  a.thresh = act.thresh; a.seed = act.seed; a.seed_dev = act.seed_dev;
  const int extra = act.mode == 1 ? 3 : (act.mode == 2 ? 2 : (R != nullptr ? 1 : 0));
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
    cus = p.multiProcessorCount;
  }
  // UBV_WS_CUS: CUs a launch may occupy (default all: a persistent block holds its CU's registers and LDS until the
  // kernel ends, so a kernel of another stream runs beside it only on CUs left out here)
  static const int cu_cap = getenv("UBV_WS_CUS") ? atoi(getenv("UBV_WS_CUS")) : 0;
  const int use_cus = cu_cap > 0 && cu_cap < cus ? cu_cap : cus;
  int per_group = use_cus / a.ncg;                                    // blocks per column group
  per_group = per_group / 8 * 8;                                      // ... a multiple of 8: one XCD's slots per sequence step
  if (per_group < 8) per_group = 8;
  while (per_group > 8 && per_group - 8 >= a.tiles) per_group -= 8;   // (tiny M: no idle blocks)
  const int grid = per_group * a.ncg;
  switch (K / 16) {
    case 16: ws_launch_extra<16>(a, extra, grid, st); break;
    case 12: ws_launch_extra<12>(a, extra, grid, st); break;
    case 8: ws_launch_extra<8>(a, extra, grid, st); break;
    case 4: ws_launch_extra<4>(a, extra, grid, st); break;
    default: return false;
  }
  return true;
}

}  // namespace ubv

// timing study: the stamps of the last UBV_WS_ABL=16 launch (2 blocks x 16 tiles x 8 stamps, shader clock)
extern "C" int ubv_debug_ws_timing(uint64_t* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ubv::ws_dbg), sizeof(uint64_t) * 2 * 16 * 8) == hipSuccess ? 0 : -2;
}
