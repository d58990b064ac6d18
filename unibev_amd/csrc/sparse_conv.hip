// Sparse 3-D convolutions of the LiDAR middle encoder for gfx950.
//
// [ext] mmdet3d 0.18.1 `SparseEncoder` (configured at projects/UniBEV/configs/unibev/
// unibev_nus_LC_cnw_256_modality_dropout.py:194-208) is built from spconv's SubMConv3d / SparseConv3d:
//        out[o, :] = sum_k  in[ nbr(k, o), : ] . W_k          (k: the kz x ky x kx kernel offsets)
// over the ACTIVE voxels only.  spconv materialises, per offset, (input, output) index pairs with atomic
// counters, gathers both sides into dense buffers, multiplies and scatter-adds.  Here:
//
//   * the rulebook is a dense NEIGHBOUR MAP nbr[k][row] (-1 = no active voxel there), built by one lookup
//     per (row, offset) in an open-addressing hash of the voxel keys — no atomics beyond the key CAS, no
//     compaction, nothing order-dependent.  Forward maps output rows to inputs (in = out*stride - pad + k),
//     the input gradient uses the transposed map (out = (in + pad - k) / stride where divisible): BOTH passes
//     are gathers, so neither needs float atomics and both are bit-reproducible.
//   * the product is ONE kernel (spconv_gather_mma_kernel): a wave owns 64 output rows x all output
//     channels; for every offset with at least one neighbour in the wave it loads its MFMA A fragment
//     straight from the gathered rows (a lane = one row x 8 consecutive channels: exactly the operand
//     layout of v_mfma_f32_32x32x16, no LDS), the B fragments from the weight block W_k (K-contiguous copy
//     made once per step), and accumulates in registers over all 27 offsets.  f32 features run as split-bf16
//     products (hi.hi + hi.lo + lo.hi, f32 accumulation) like the Linear layers (gemm_mfma.hip).
//
// Keys: ((b * D + z) * H + y) * W + x as int64.  Coordinates are (batch, z, y, x) int32 rows, the layout of
// mmdet3d's voxel `coors`.
#include "ubv_common.h"

namespace ubv {

typedef __attribute__((ext_vector_type(8))) __bf16 sbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 sf16x8_t;
typedef __attribute__((ext_vector_type(16))) float sf32x16_t;

struct SpGeom {
  int B;
  int in_dims[3];       // D, H, W of the map the QUERY rows live in
  int tgt_dims[3];      // D, H, W of the map that is looked up
  int ksize[3], stride[3], pad[3];
  int mode;             // 0: target = row * stride - pad + k   (outputs -> inputs; SubM: stride 1)
                        // 1: target = (row + pad - k) / stride where divisible (inputs -> outputs)
};

__device__ __forceinline__ uint64_t spc_hash(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

// table: tkeys[cap] (int64, -1 = empty), tvals[cap]; cap a power of two >= 2 n.  Keys are unique.
__global__ __launch_bounds__(256) void spc_insert_kernel(const int32_t* __restrict__ coors, long n, int D, int H,
                                                         int W, long long* __restrict__ tkeys,
                                                         int32_t* __restrict__ tvals, long cap_mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* c = coors + i * 4;
  const long long key = (((long long)c[0] * D + c[1]) * H + c[2]) * W + c[3];
  long slot = (long)(spc_hash((uint64_t)key) & (uint64_t)cap_mask);
  for (;;) {
    const unsigned long long prev = atomicCAS((unsigned long long*)(tkeys + slot), ~0ull, (unsigned long long)key);
    if (prev == ~0ull || prev == (unsigned long long)key) { tvals[slot] = (int32_t)i; return; }
    slot = (slot + 1) & cap_mask;
  }
}

__device__ __forceinline__ int spc_lookup(const long long* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                          long cap_mask, long long key) {
  long slot = (long)(spc_hash((uint64_t)key) & (uint64_t)cap_mask);
  for (;;) {
    const long long k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == -1) return -1;
    slot = (slot + 1) & cap_mask;
  }
}

// target coordinate of (row coordinate c, kernel offset kk) or false
__device__ __forceinline__ bool spc_target(const SpGeom& g, const int32_t* c, int kz, int ky, int kx, int (&t)[3]) {
  const int kk[3] = {kz, ky, kx};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int v;
    if (g.mode == 0) {
      v = c[1 + d] * g.stride[d] - g.pad[d] + kk[d];
    } else {
      const int num = c[1 + d] + g.pad[d] - kk[d];
      if (num < 0 || num % g.stride[d] != 0) return false;
      v = num / g.stride[d];
    }
    if (v < 0 || v >= g.tgt_dims[d]) return false;
    t[d] = v;
  }
  return true;
}

// nbr[k][row] = index of the active voxel at the target of (row, k) in the hashed map, or -1
__global__ __launch_bounds__(256) void spc_neighbors_kernel(const int32_t* __restrict__ coors, long rows,
                                                            const SpGeom g, const long long* __restrict__ tkeys,
                                                            const int32_t* __restrict__ tvals, long cap_mask,
                                                            int32_t* __restrict__ nbr, long ld) {
  const int kvol = g.ksize[0] * g.ksize[1] * g.ksize[2];
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * kvol) return;
  const long row = t % rows;                     // consecutive threads: consecutive rows of one offset
  const int k = (int)(t / rows);
  const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
  const int32_t* c = coors + row * 4;
  int tc[3];
  int idx = -1;
  if (spc_target(g, c, kz, ky, kx, tc)) {
    const long long key = (((long long)c[0] * g.tgt_dims[0] + tc[0]) * g.tgt_dims[1] + tc[1]) * g.tgt_dims[2] + tc[2];
    idx = spc_lookup(tkeys, tvals, cap_mask, key);
  }
  nbr[(long)k * ld + row] = idx;
}

// ---- output sites of a strided convolution, on the device ---------------------------------------------------
// spconv's get_indice_pairs builds the output set with a hash / unique pass and hands its SIZE to the host.  Here the
// set is a bit per output cell (2.7 MB for the 21 x 720 x 720 x 2 map of the first strided layer): inputs MARK the
// cells they reach, a popcount scan ranks the set bits, and the coordinates come out in ascending key order without
// a sort.  Every kernel takes its input count from device memory (n_dev), so a chain of strided layers is built
// back to back and the host reads all their counts in ONE copy (round 3: torch.unique + a boolean-mask index per
// layer, four host reads per pass).
constexpr int kSiteBlk = 256;                             // bitmap words per block of the count / emit kernels

__global__ __launch_bounds__(256) void spc_mark_kernel(const int32_t* __restrict__ coors, const int32_t* __restrict__ n_dev,
                                                       long n_host, const SpGeom g, uint32_t* __restrict__ bitmap) {
  const long n = n_dev != nullptr ? (long)*n_dev : n_host;
  const int kvol = g.ksize[0] * g.ksize[1] * g.ksize[2];
  const long total = n * kvol;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const long row = t % n;                               // consecutive threads: consecutive rows of one offset
    const int k = (int)(t / n);
    const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
    const int32_t* c = coors + row * 4;
    int tc[3];
    if (!spc_target(g, c, kz, ky, kx, tc)) continue;
    const long long key = (((long long)c[0] * g.tgt_dims[0] + tc[0]) * g.tgt_dims[1] + tc[1]) * g.tgt_dims[2] + tc[2];
    const uint32_t bit = 1u << (key & 31);
    uint32_t* w = bitmap + (key >> 5);
    if (!(__builtin_nontemporal_load(w) & bit)) atomicOr(w, bit);      // (a stale read only costs a redundant atomic)
  }
}

__device__ __forceinline__ int spc_block_scan(int v, int* s_w, int& total) {      // exclusive, 256 threads
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < wv; ++i) base += s_w[i];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + inc - v;
}

__global__ __launch_bounds__(256) void spc_site_count_kernel(const uint32_t* __restrict__ bitmap, long words,
                                                             int32_t* __restrict__ sums) {
  __shared__ int s_w[4];
  const long w = (long)blockIdx.x * kSiteBlk + threadIdx.x;
  int total;
  (void)spc_block_scan(w < words ? __popc(bitmap[w]) : 0, s_w, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// sums[0 .. nb) -> exclusive prefix in place, sums[nb] = count_out[0] = the total (one block)
__global__ __launch_bounds__(256) void spc_site_scan_kernel(int32_t* __restrict__ sums, long nb, int32_t* __restrict__ count_out) {
  __shared__ int s_w[4];
  int carry = 0;
  for (long b0 = 0; b0 < nb; b0 += 256) {
    const long i = b0 + threadIdx.x;
    const int v = i < nb ? sums[i] : 0;
    int total;
    const int ex = spc_block_scan(v, s_w, total);
    if (i < nb) sums[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[nb] = carry; count_out[0] = carry; }
}

__global__ __launch_bounds__(256) void spc_site_emit_kernel(const uint32_t* __restrict__ bitmap, long words,
                                                            const int32_t* __restrict__ sums, int D, int H, int W,
                                                            int32_t* __restrict__ coors, long cap) {
  __shared__ int s_w[4];
  const long w = (long)blockIdx.x * kSiteBlk + threadIdx.x;
  uint32_t bits = w < words ? bitmap[w] : 0u;
  int total;
  long rank = (long)sums[blockIdx.x] + spc_block_scan(__popc(bits), s_w, total);
  while (bits) {
    const int b = __builtin_ctz(bits);
    bits &= bits - 1u;
    long long k = (long long)w * 32 + b;
    const int x = (int)(k % W); k /= W;
    const int y = (int)(k % H); k /= H;
    const int z = (int)(k % D); k /= D;
    if (rank < cap) reinterpret_cast<int4*>(coors)[rank] = make_int4((int)k, z, y, x);
    ++rank;
  }
}

// ---- compacted pairs of a neighbour map (spconv's rulebook form, for the weight gradient) -----------------------
// per offset k: the rows with a neighbour, in row order.  Three small kernels: counts per 2048-row chunk, a scan of
// the chunk counts per offset, ranked writes.  (Round 3: a stable torch.argsort over the [kvol, rows] validity map.)
constexpr int kPairChunk = 2048;

__global__ __launch_bounds__(256) void spc_pair_count_kernel(const int32_t* __restrict__ nbr, long ld, long rows,
                                                             int chunks, int32_t* __restrict__ sums) {
  __shared__ int s_w[4];
  const int k = blockIdx.y;
  const long r0 = (long)blockIdx.x * kPairChunk + threadIdx.x * 8;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) c += (r0 + i < rows && nbr[(long)k * ld + r0 + i] >= 0) ? 1 : 0;
  int total;
  (void)spc_block_scan(c, s_w, total);
  if (threadIdx.x == 0) sums[(long)k * chunks + blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void spc_pair_scan_kernel(int32_t* __restrict__ sums, int chunks, int32_t* __restrict__ counts) {
  __shared__ int s_w[4];
  int32_t* mine = sums + (long)blockIdx.x * chunks;
  int carry = 0;
  for (int b0 = 0; b0 < chunks; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int v = i < chunks ? mine[i] : 0;
    int total;
    const int ex = spc_block_scan(v, s_w, total);
    if (i < chunks) mine[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void spc_pair_write_kernel(const int32_t* __restrict__ nbr, long ld, long rows,
                                                             int chunks, const int32_t* __restrict__ sums,
                                                             int32_t* __restrict__ out_rows, int32_t* __restrict__ in_rows) {
  __shared__ int s_w[4];
  const int k = blockIdx.y;
  const long r0 = (long)blockIdx.x * kPairChunk + threadIdx.x * 8;
  int v[8], c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = r0 + i < rows ? nbr[(long)k * ld + r0 + i] : -1;
    c += v[i] >= 0 ? 1 : 0;
  }
  int total;
  long at = (long)k * ld + sums[(long)k * chunks + blockIdx.x] + spc_block_scan(c, s_w, total);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (v[i] >= 0) { out_rows[at] = (int32_t)(r0 + i); in_rows[at] = v[i]; ++at; }
}

// ------------------------------------------------------------------------------------------------
// out[row, :] = sum_k in[nbr[k][row], :] . W_k^T      W given as [kvol][CoutP][Cin] (Cin contiguous)
// T: feature element type (float: split-bf16 products, W as hi + lo bf16 arrays).  NB = CoutP / 32.
template <typename T> struct spc_mma;
template <> struct spc_mma<bf16_t> {
  static __device__ __forceinline__ sf32x16_t run(uint4 a, uint4 b, sf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8_t, a), __builtin_bit_cast(sbf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct spc_mma<f16_t> {
  static __device__ __forceinline__ sf32x16_t run(uint4 a, uint4 b, sf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8_t, a), __builtin_bit_cast(sf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct spc_mma<float> : spc_mma<bf16_t> {};

__device__ __attribute__((aligned(16))) float spc_zero_row[128];     // (the gather target of "no neighbour")

// A block of NW waves owns 64 NW consecutive output rows (a wave: two 32-row MFMA blocks x all output channels, the
// accumulators in registers over all offsets).  The weight block W_k of an offset is staged ONCE PER BLOCK in LDS
// (80 - 272-byte rows: conflict-free 16-byte fragment reads).  History: round 2 fetched the B fragments from L2 in
// every wave (1.8 MB of W per 64 rows at 128 x 128 channels, 578 us per convolution); round 3 staged W_k between two
// barriers per offset (383 us: 340 registers = ONE wave per SIMD at 128 channels, so the barrier, the L2 round trip of
// the copy and every gather latency were exposed 27 times).  Round 4: the channel counts are template parameters
// (KS = Cin / 16), the LDS holds TWO weight buffers and W_{k+1} is copied piecewise in the slots of offset k's
// 16-channel steps (global load in one step, LDS store in the next) — one barrier per offset, nothing waits for
// the copy — and at 128 output channels a block is 8 waves (2 per SIMD, <= 256 registers) so a wave's gather
// latency hides behind its neighbour's MFMAs.
template <typename T, int NB, int KS, int NW>
__global__ __launch_bounds__(64 * NW) void spconv_gather_mma_kernel(const T* __restrict__ feats, const int32_t* __restrict__ nbr,
                                                                    long ld, long rows, const uint16_t* __restrict__ w_hi,
                                                                    const uint16_t* __restrict__ w_lo, T* __restrict__ out,
                                                                    int Cout, int kvol, int nblk) {
  extern __shared__ __attribute__((aligned(16))) uint16_t wlds[];
  constexpr bool SPLIT = sizeof(T) == 4;
  constexpr int RB = 2, THREADS = 64 * NW, Cin = KS * 16;
  constexpr int LD = Cin + 8;                              // halves per staged row
  constexpr int PLANE = NB * 32 * LD, PL = SPLIT ? 2 : 1, BUF = PL * PLANE;
  constexpr int CPR = Cin / 8, PIECES = NB * 32 * CPR;     // 16-byte pieces per row / per plane
  constexpr int IT = (PIECES + THREADS - 1) / THREADS;     // piece rounds per thread and offset
  static_assert(IT <= KS, "one piece round per 16-channel step");
  constexpr int NV = SPLIT ? 2 : 1;                        // 16-byte loads per row and step
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // consecutive row blocks share neighbours: give each XCD (blockIdx mod 8) a contiguous range of them
  const int per = (nblk + 7) >> 3;
  const int bid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const long row0 = ((long)bid * NW + wv) * (32 * RB);
  const bool wave_live = bid < nblk && row0 < rows;        // (idle waves still copy weights and take the barriers)
  const int m = lane & 31, kg = lane >> 5;
  sf32x16_t acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  int idx[RB], idn[RB];
  auto load_idx = [&](int k, int (&d)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const long row = row0 + i * 32 + m;
      d[i] = (k < kvol && wave_live && row < rows) ? nbr[(long)k * ld + row] : -1;
    }
  };
  // A fragments straight from the gathered rows.  Rows without a neighbour read a row of zeros: zeroing the registers
  // behind the load instead made the wave wait for the load it had just issued whenever one of its rows had no
  // neighbour (s_waitcnt vmcnt(0) under the lane mask) — the prefetch was no prefetch.
  auto load_a = [&](const int (&ix)[RB], int c0, uint4 (&dst)[RB][NV]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const T* frow = (ix[i] >= 0 ? feats + (long)ix[i] * Cin : reinterpret_cast<const T*>(spc_zero_row)) + kg * 8 + c0;
#pragma unroll
      for (int v = 0; v < NV; ++v) dst[i][v] = reinterpret_cast<const uint4*>(frow)[v];
    }
  };
  // the weight copy, one round = one 16-byte piece per plane and thread
  struct WPiece { uint4 hi, lo; };
  auto w_fetch = [&](int k, int it) -> WPiece {
    WPiece h;
    h.hi = h.lo = make_uint4(0u, 0u, 0u, 0u);
    const int p = it * THREADS + (int)threadIdx.x;
    if (PIECES % THREADS == 0 || p < PIECES) {
      const int r = p / CPR, c8 = (p - r * CPR) * 8;
      h.hi = *reinterpret_cast<const uint4*>(w_hi + ((long)k * NB * 32 + r) * Cin + c8);
      if constexpr (SPLIT) h.lo = *reinterpret_cast<const uint4*>(w_lo + ((long)k * NB * 32 + r) * Cin + c8);
    }
    return h;
  };
  auto w_store = [&](uint16_t* buf, int it, const WPiece h) {
    const int p = it * THREADS + (int)threadIdx.x;
    if (PIECES % THREADS == 0 || p < PIECES) {
      const int r = p / CPR, c8 = (p - r * CPR) * 8;
      *reinterpret_cast<uint4*>(buf + r * LD + c8) = h.hi;
      if constexpr (SPLIT) *reinterpret_cast<uint4*>(buf + PLANE + r * LD + c8) = h.lo;
    }
  };
  // Offsets NO row of the block has a neighbour at are left out (the rows are in key order at the strided levels:
  // the z = 0 / z = D - 1 planes miss a third of the offsets): their bit mask is gathered first, and the loops
  // below walk the set bits.  (A per-wave test inside the offset loop — round 3 — put a branch around the step loop,
  // and the compiler then carried the accumulators in VGPRs and copied all of them into and out of the accumulation
  // registers for every offset: 340 registers and 2 x 128 moves per offset at 128 channels.)
  __shared__ unsigned s_mask;
  if (threadIdx.x == 0) s_mask = 0u;
  __syncthreads();
  {
    unsigned mine = 0u;
    for (int k = 0; k < kvol; ++k) {
      int d[RB];
      load_idx(k, d);
      bool any = false;
#pragma unroll
      for (int i = 0; i < RB; ++i) any = any || d[i] >= 0;
      if (__ballot(any) != 0ull) mine |= 1u << k;
    }
    if (lane == 0 && mine) atomicOr(&s_mask, mine);
  }
  __syncthreads();
  unsigned mask = s_mask;
  // gathered rows are requested PD 16-channel steps ahead of their MFMAs (a step of one wave is ~0.35 us of matrix
  // work, a gather out of L2 / MALL ~0.8 us: one step ahead left the latency exposed even with two waves per SIMD)
  constexpr int PD = KS % 2 == 0 ? 2 : 1;
  uint4 raw[PD][RB][NV];
  if (mask != 0u) {
    int k = __builtin_ctz(mask);
    mask &= mask - 1u;
    load_idx(k, idx);
    load_a(idx, 0, raw[0]);
    if constexpr (PD == 2) load_a(idx, 16, raw[1]);
#pragma unroll 1
    for (int it = 0; it < IT; ++it) w_store(wlds, it, w_fetch(k, it));
    __syncthreads();
    int par = 0;
    for (;;) {
      const int kn = mask ? __builtin_ctz(mask) : -1;
      mask &= mask - 1u;                                  // (0 stays 0)
      load_idx(kn >= 0 ? kn : kvol, idn);                 // (past the last offset: -1)
      const uint16_t* wk = wlds + par * BUF + m * LD + kg * 8;
      uint16_t* oth = wlds + (par ^ 1) * BUF;
      WPiece hold;
      hold.hi = hold.lo = make_uint4(0u, 0u, 0u, 0u);
      int it_next = kn >= 0 ? 0 : IT;                     // W_kn: the next piece round to request
      bool pending = false;
      auto step = [&](const int s, uint4 (&rw)[RB][NV]) {
        const int c0 = s * 16;
        // last step's piece of W_kn -> LDS, this step's piece -> registers
        if (pending) w_store(oth, it_next - 1, hold);
        pending = it_next < IT;
        if (pending) hold = w_fetch(kn, it_next++);
        uint4 a_hi[RB], a_lo[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          if constexpr (SPLIT) {
            const float4 v0 = __builtin_bit_cast(float4, rw[i][0]);
            const float4 v1 = __builtin_bit_cast(float4, rw[i][1]);
            uint4 h, l;
            h.x = cvt_pk_bf16(v0.x, v0.y); h.y = cvt_pk_bf16(v0.z, v0.w);
            h.z = cvt_pk_bf16(v1.x, v1.y); h.w = cvt_pk_bf16(v1.z, v1.w);
            l.x = cvt_pk_bf16(v0.x - __uint_as_float(h.x << 16), v0.y - __uint_as_float(h.x & 0xffff0000u));
            l.y = cvt_pk_bf16(v0.z - __uint_as_float(h.y << 16), v0.w - __uint_as_float(h.y & 0xffff0000u));
            l.z = cvt_pk_bf16(v1.x - __uint_as_float(h.z << 16), v1.y - __uint_as_float(h.z & 0xffff0000u));
            l.w = cvt_pk_bf16(v1.z - __uint_as_float(h.w << 16), v1.w - __uint_as_float(h.w & 0xffff0000u));
            a_hi[i] = h; a_lo[i] = l;
          } else {
            a_hi[i] = rw[i][0];
            a_lo[i] = make_uint4(0u, 0u, 0u, 0u);
          }
        }
        // step s + PD's rows are requested before this step's MFMAs, into the registers the conversion just freed
        // (PD = 2: the two halves of raw alternate — moving a landing load's registers would wait for it); the last
        // PD steps fetch the next OFFSET's first steps instead
        {
          const int t = s + PD;
          if (t >= KS) load_a(idn, (t - KS) * 16, rw);
          else load_a(idx, t * 16, rw);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint4 b_hi = *reinterpret_cast<const uint4*>(wk + j * 32 * LD + c0);
          uint4 b_lo = make_uint4(0u, 0u, 0u, 0u);
          if constexpr (SPLIT) b_lo = *reinterpret_cast<const uint4*>(wk + PLANE + j * 32 * LD + c0);
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            acc[i][j] = spc_mma<T>::run(a_hi[i], b_hi, acc[i][j]);
            if constexpr (SPLIT) {
              acc[i][j] = spc_mma<T>::run(a_hi[i], b_lo, acc[i][j]);
              acc[i][j] = spc_mma<T>::run(a_lo[i], b_hi, acc[i][j]);
            }
          }
        }
      };
      if constexpr (PD == 2) {
#pragma unroll 1
        for (int s2 = 0; s2 < KS; s2 += 2) {
          step(s2, raw[0]);
          step(s2 + 1, raw[1]);
        }
      } else {
#pragma unroll 1
        for (int s = 0; s < KS; ++s) step(s, raw[0]);
      }
      if (pending) w_store(oth, it_next - 1, hold);
#pragma unroll
      for (int i = 0; i < RB; ++i) idx[i] = idn[i];
      __syncthreads();                                    // W_kn complete, W_k's readers done
      if (kn < 0) break;
      par ^= 1;
    }
  }
  // D[i][n]: this lane holds column n = lane & 31 (output channel within block j), rows (r & 3) + 8 (r >> 2) + 4 kg
  if (!wave_live) return;
  // Round 5: the stores go through a wave-uniform base (this wave's first row) + a 32-bit element offset, and the row
  // bound is tested once per wave — written as `out[orow * Cout + n]` under `if (orow < rows)` the epilogue was 1 570
  // instructions for its 128 stores (a 64-bit multiply, compare and branch per store: 387 quarter-rate multiplies), ~5 %
  // of the kernel's cycles at 128 channels.
  T* __restrict__ ob = out + row0 * Cout;
  const uint32_t lo = (uint32_t)(4 * kg * Cout + m);      // lane's offset inside the wave's 64 rows x Cout block (< 2^13)
  const uint32_t uc = (uint32_t)Cout;
  if (row0 + 32 * RB <= rows) {                           // every row of the wave exists (all waves but the last one)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j * 32 + m >= Cout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ob[lo + (uint32_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * uc + (uint32_t)(j * 32)] = elem<T>::from_float(acc[i][j][r]);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j * 32 + m >= Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
        if (row0 + rr + 4 * kg < rows) ob[lo + (uint32_t)rr * uc + (uint32_t)(j * 32)] = elem<T>::from_float(acc[i][j][r]);
      }
    }
}

template <typename T, int NB, int KS>
static void spconv_launch_one(const void* feats, const int32_t* nbr, long ld, long rows, const void* w_hi, const void* w_lo,
                              void* out, int Cout, int kvol, hipStream_t st) {
  constexpr int NW = NB >= 4 ? 8 : 4;
  constexpr int LD = KS * 16 + 8;
  constexpr size_t lds = (size_t)2 * (sizeof(T) == 4 ? 2 : 1) * NB * 32 * LD * sizeof(uint16_t);
  const int nblk = (int)((rows + 64 * NW - 1) / (64 * NW));
  const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), blk(64 * NW);
  auto fn = spconv_gather_mma_kernel<T, NB, KS, NW>;
  // (more than 64 KB of dynamic LDS at 128 output channels: ask for it explicitly)
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fn, grid, blk, lds, st, (const T*)feats, nbr, ld, rows, (const uint16_t*)w_hi, (const uint16_t*)w_lo,
                     (T*)out, Cout, kvol, nblk);
}

template <typename T>
static int spconv_launch(const void* feats, const int32_t* nbr, long ld, long rows, const void* w_hi, const void* w_lo,
                         void* out, int Cin, int Cout, int kvol, hipStream_t st) {
  const int nb = (Cout + 31) / 32, ks = Cin / 16;
#define UBV_SPC(NBV, KSV)                                                                                       \
  if (nb == NBV && ks == KSV) {                                                                                 \
    spconv_launch_one<T, NBV, KSV>(feats, nbr, ld, rows, w_hi, w_lo, out, Cout, kvol, st);                      \
    return UBV_OK;                                                                                              \
  }
#define UBV_SPC_ROW(NBV) UBV_SPC(NBV, 1) UBV_SPC(NBV, 2) UBV_SPC(NBV, 3) UBV_SPC(NBV, 4) UBV_SPC(NBV, 5) UBV_SPC(NBV, 6) UBV_SPC(NBV, 7) UBV_SPC(NBV, 8)
  UBV_SPC_ROW(1) UBV_SPC_ROW(2) UBV_SPC_ROW(3) UBV_SPC_ROW(4)
#undef UBV_SPC_ROW
#undef UBV_SPC
  return UBV_ERR_UNSUPPORTED;
}

// Weight blocks as stored by spconv, w [kvol][Cin][Cout], -> the product kernel's operand [kvol][rowsP][K]
// (rowsP = rows padded to 32, zero rows past them) in ONE launch:  transpose != 0 (forward): rows = Cout, K = Cin,
// op[k][co][ci] = w[k][ci][co];  transpose == 0 (input gradient): rows = Cin, K = Cout, op[k][ci][co] = w[k'][ci][co]
// with k' = kvol - 1 - k when flip (submanifold layers read the forward map with mirrored offsets).  f32 weights come out
// as bf16 hi + lo halves.  (Round 3: reshape / transpose / flip / pad / contiguous framework copies and a split kernel.)
template <typename T>
__global__ __launch_bounds__(256) void spc_weight_operand_kernel(const T* __restrict__ w, int kvol, int cin, int cout,
                                                                 int rowsP, int transpose, int flip,
                                                                 uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  const int K = transpose ? cin : cout, rows = transpose ? cout : cin;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)kvol * rowsP * K) return;
  const int c = (int)(t % K), r = (int)((t / K) % rowsP), k = (int)(t / ((long)K * rowsP));
  const int ks = flip ? kvol - 1 - k : k;
  if constexpr (sizeof(T) == 4) {
    float v = 0.0f;
    if (r < rows) v = transpose ? w[((long)ks * cin + c) * cout + r] : w[((long)ks * cin + r) * cout + c];
    const uint16_t h = (uint16_t)float_to_bf16_bits(v);
    hi[t] = h;
    lo[t] = (uint16_t)float_to_bf16_bits(v - bf16_bits_to_float(h));
  } else {
    uint16_t v = 0;
    const uint16_t* w16 = reinterpret_cast<const uint16_t*>(w);
    if (r < rows) v = transpose ? w16[((long)ks * cin + c) * cout + r] : w16[((long)ks * cin + r) * cout + c];
    hi[t] = v;
  }
}

static bool sp_geom(SpGeom& g, int B, const int* in_dims, const int* tgt_dims, const int* ksize, const int* stride,
                    const int* pad, int mode) {
  g.B = B; g.mode = mode;
  for (int d = 0; d < 3; ++d) {
    g.in_dims[d] = in_dims[d]; g.tgt_dims[d] = tgt_dims[d]; g.ksize[d] = ksize[d]; g.stride[d] = stride[d];
    g.pad[d] = pad[d];
    if (in_dims[d] <= 0 || tgt_dims[d] <= 0 || ksize[d] <= 0 || stride[d] <= 0 || pad[d] < 0) return false;
  }
  return B > 0;
}

}  // namespace ubv

extern "C" int64_t ubv_spconv_table_slots(int64_t n) {
  int64_t cap = 64;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

extern "C" int ubv_spconv_hash_build(const int32_t* coors, int64_t n, int D, int H, int W, int64_t* table_keys,
                                     int32_t* table_vals, int64_t slots, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(coors && table_keys && table_vals && n >= 0 && D > 0 && H > 0 && W > 0, "spconv_hash_build: bad arguments");
  UBV_CHECK_ARG(slots >= 2 * n && (slots & (slots - 1)) == 0, "spconv_hash_build: slots must be a power of two >= 2 n");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(table_keys, 0xff, (size_t)slots * sizeof(int64_t), st) != hipSuccess) {
    set_error("spconv_hash_build: memset failed");
    return UBV_ERR_LAUNCH;
  }
  if (n > 0)
    hipLaunchKernelGGL(spc_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, coors, (long)n, D, H, W,
                       (long long*)table_keys, table_vals, (long)(slots - 1));
  UBV_CHECK_LAUNCH("spconv_hash_build");
  return UBV_OK;
}

extern "C" int ubv_spconv_neighbors(const int32_t* coors, int64_t rows, int B, const int* row_dims,
                                    const int* target_dims, const int* ksize, const int* stride, const int* pad,
                                    int transposed, const int64_t* table_keys, const int32_t* table_vals,
                                    int64_t slots, int32_t* nbr, int64_t ld, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(coors && table_keys && table_vals && nbr && rows >= 0 && ld >= rows, "spconv_neighbors: bad arguments");
  SpGeom g;
  UBV_CHECK_ARG(sp_geom(g, B, row_dims, target_dims, ksize, stride, pad, transposed ? 1 : 0), "spconv_neighbors: bad geometry");
  if (rows == 0) return UBV_OK;
  const long total = (long)rows * ksize[0] * ksize[1] * ksize[2];
  hipLaunchKernelGGL(spc_neighbors_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), coors,
                     (long)rows, g, (const long long*)table_keys, table_vals, (long)(slots - 1), nbr, (long)ld);
  UBV_CHECK_LAUNCH("spconv_neighbors");
  return UBV_OK;
}

extern "C" int64_t ubv_spconv_sites_words(int B, const int* out_dims) {
  if (!out_dims || B <= 0 || out_dims[0] <= 0 || out_dims[1] <= 0 || out_dims[2] <= 0) return 0;
  const int64_t cells = (int64_t)B * out_dims[0] * out_dims[1] * out_dims[2];
  return (cells + 31) / 32;
}

extern "C" int ubv_spconv_output_sites(const int32_t* coors, const int32_t* n_dev, int64_t n, int B, const int* in_dims,
                                       const int* out_dims, const int* ksize, const int* stride, const int* pad,
                                       int32_t* bitmap, int64_t words, int32_t* sums, int32_t* out_coors, int64_t cap,
                                       int32_t* count_dev, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG((coors || n == 0) && bitmap && sums && out_coors && count_dev && n >= 0 && cap >= 0, "spconv_output_sites: bad arguments");
  SpGeom g;
  UBV_CHECK_ARG(sp_geom(g, B, in_dims, out_dims, ksize, stride, pad, 1), "spconv_output_sites: bad geometry");
  UBV_CHECK_ARG(words == ubv_spconv_sites_words(B, out_dims) && words < (int64_t)1 << 31, "spconv_output_sites: bitmap of %lld words expected", (long long)ubv_spconv_sites_words(B, out_dims));
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(bitmap, 0, (size_t)words * sizeof(int32_t), st) != hipSuccess) {
    set_error("spconv_output_sites: memset failed");
    return UBV_ERR_LAUNCH;
  }
  const long total = (long)n * ksize[0] * ksize[1] * ksize[2];
  if (total > 0) {
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(spc_mark_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, coors, n_dev,
                       (long)n, g, (uint32_t*)bitmap);
  }
  const long nb = (words + kSiteBlk - 1) / kSiteBlk;
  hipLaunchKernelGGL(spc_site_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const uint32_t*)bitmap, (long)words, sums);
  hipLaunchKernelGGL(spc_site_scan_kernel, dim3(1), dim3(256), 0, st, sums, nb, count_dev);
  hipLaunchKernelGGL(spc_site_emit_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const uint32_t*)bitmap, (long)words,
                     (const int32_t*)sums, out_dims[0], out_dims[1], out_dims[2], out_coors, (long)cap);
  UBV_CHECK_LAUNCH("spconv_output_sites");
  return UBV_OK;
}

extern "C" int64_t ubv_spconv_pairs_chunks(int64_t rows) { return rows <= 0 ? 0 : (rows + ubv::kPairChunk - 1) / ubv::kPairChunk; }

extern "C" int ubv_spconv_pairs(const int32_t* nbr, int64_t ld, int64_t rows, int kvol, int32_t* chunk_sums,
                                int32_t* out_rows, int32_t* in_rows, int32_t* counts, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(nbr && chunk_sums && out_rows && in_rows && counts && rows >= 0 && ld >= rows && kvol > 0 && kvol < 65536,
                "spconv_pairs: bad arguments");
  hipStream_t st = as_stream(stream);
  if (rows == 0) {
    if (hipMemsetAsync(counts, 0, (size_t)kvol * sizeof(int32_t), st) != hipSuccess) { set_error("spconv_pairs: memset failed"); return UBV_ERR_LAUNCH; }
    return UBV_OK;
  }
  const int chunks = (int)ubv_spconv_pairs_chunks(rows);
  const dim3 grid((unsigned)chunks, (unsigned)kvol);
  hipLaunchKernelGGL(spc_pair_count_kernel, grid, dim3(256), 0, st, nbr, (long)ld, (long)rows, chunks, chunk_sums);
  hipLaunchKernelGGL(spc_pair_scan_kernel, dim3((unsigned)kvol), dim3(256), 0, st, chunk_sums, chunks, counts);
  hipLaunchKernelGGL(spc_pair_write_kernel, grid, dim3(256), 0, st, nbr, (long)ld, (long)rows, chunks,
                     (const int32_t*)chunk_sums, out_rows, in_rows);
  UBV_CHECK_LAUNCH("spconv_pairs");
  return UBV_OK;
}

extern "C" int ubv_spconv_weight_operand(const void* w, int kvol, int Cin, int Cout, int transpose, int flip, int dtype,
                                         void* w_hi, void* w_lo, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(w && w_hi && kvol > 0 && Cin > 0 && Cout > 0, "spconv_weight_operand: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "spconv_weight_operand: unknown dtype %d", dtype);
  UBV_CHECK_ARG((dtype == UBV_F32) == (w_lo != nullptr), "spconv_weight_operand: f32 weights come out as two halves (hi, lo)");
  const int rows = transpose ? Cout : Cin, K = transpose ? Cin : Cout;
  const int rowsP = (rows + 31) / 32 * 32;
  const long total = (long)kvol * rowsP * K;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  hipStream_t st = as_stream(stream);
  if (dtype == UBV_F32)
    hipLaunchKernelGGL(spc_weight_operand_kernel<float>, grid, blk, 0, st, (const float*)w, kvol, Cin, Cout, rowsP,
                       transpose, flip, (uint16_t*)w_hi, (uint16_t*)w_lo);
  else
    hipLaunchKernelGGL(spc_weight_operand_kernel<uint16_t>, grid, blk, 0, st, (const uint16_t*)w, kvol, Cin, Cout, rowsP,
                       transpose, flip, (uint16_t*)w_hi, (uint16_t*)nullptr);
  UBV_CHECK_LAUNCH("spconv_weight_operand");
  return UBV_OK;
}

extern "C" int ubv_spconv_gather_mma(const void* feats, const int32_t* nbr, int64_t ld, int64_t rows, const void* w_hi,
                                     const void* w_lo, void* out, int Cin, int Cout, int kvol, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(feats && nbr && w_hi && out && rows >= 0 && ld >= rows && kvol > 0, "spconv_gather_mma: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "spconv_gather_mma: unknown dtype %d", dtype);
  UBV_CHECK_ARG((dtype == UBV_F32) == (w_lo != nullptr), "spconv_gather_mma: f32 features take split weights (hi, lo)");
  if (Cin % 16 != 0 || Cin > 128 || Cout <= 0 || Cout > 128 || ((uintptr_t)feats % 16) != 0 || ((uintptr_t)w_hi % 16) != 0) {
    set_error("spconv_gather_mma: Cin=%d must be a multiple of 16 up to 128, Cout=%d at most 128, buffers 16-byte aligned", Cin, Cout);
    return UBV_ERR_UNSUPPORTED;
  }
  if (rows == 0) return UBV_OK;
  hipStream_t st = as_stream(stream);
  int rc;
  if (dtype == UBV_F32) rc = spconv_launch<float>(feats, nbr, ld, rows, w_hi, w_lo, out, Cin, Cout, kvol, st);
  else if (dtype == UBV_F16) rc = spconv_launch<f16_t>(feats, nbr, ld, rows, w_hi, nullptr, out, Cin, Cout, kvol, st);
  else rc = spconv_launch<bf16_t>(feats, nbr, ld, rows, w_hi, nullptr, out, Cin, Cout, kvol, st);
  if (rc != UBV_OK) { set_error("spconv_gather_mma: no kernel for Cin=%d Cout=%d", Cin, Cout); return rc; }
  UBV_CHECK_LAUNCH("spconv_gather_mma");
  return UBV_OK;
}
