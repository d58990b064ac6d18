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

// Candidate output keys of a strided convolution: cand[k][i] = key of the output that input i reaches
// through offset k, or -1.  (The host sorts / uniques them: the output set in ascending key order.)
__global__ __launch_bounds__(256) void spc_candidates_kernel(const int32_t* __restrict__ coors, long n,
                                                             const SpGeom g, long long* __restrict__ cand) {
  const int kvol = g.ksize[0] * g.ksize[1] * g.ksize[2];
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * kvol) return;
  const long row = t % n;
  const int k = (int)(t / n);
  const int kx = k % g.ksize[2], ky = (k / g.ksize[2]) % g.ksize[1], kz = k / (g.ksize[2] * g.ksize[1]);
  const int32_t* c = coors + row * 4;
  int tc[3];
  long long key = -1;
  if (spc_target(g, c, kz, ky, kx, tc))
    key = (((long long)c[0] * g.tgt_dims[0] + tc[0]) * g.tgt_dims[1] + tc[1]) * g.tgt_dims[2] + tc[2];
  cand[t] = key;
}

__global__ __launch_bounds__(256) void spc_keys_to_coors_kernel(const long long* __restrict__ keys, long n, int D,
                                                                int H, int W, int32_t* __restrict__ coors) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long k = keys[i];
  const int x = (int)(k % W); k /= W;
  const int y = (int)(k % H); k /= H;
  const int z = (int)(k % D); k /= D;
  reinterpret_cast<int4*>(coors)[i] = make_int4((int)k, z, y, x);
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

// RB: 32-row blocks per wave.  The weight block W_k of an offset is staged ONCE PER BLOCK in LDS (80 - 272-byte rows:
// conflict-free 16-byte fragment reads) and feeds the 4 waves x RB row blocks of the block.  Before, every wave
// fetched its B fragments from L2 itself: 1.8 MB of W per 64 output rows at 128 x 128 channels — 578 us per
// convolution of the 128-channel stage, bound by those reads (round 2: one row block per wave, 9.6 TB/s of L2 reads).
template <typename T, int NB, int RB>
__global__ __launch_bounds__(256) void spconv_gather_mma_kernel(const T* __restrict__ feats, const int32_t* __restrict__ nbr,
                                                                long ld, long rows, const uint16_t* __restrict__ w_hi,
                                                                const uint16_t* __restrict__ w_lo, T* __restrict__ out,
                                                                int Cin, int Cout, int kvol) {
  extern __shared__ __attribute__((aligned(16))) uint16_t wlds[];
  constexpr bool SPLIT = sizeof(T) == 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * 4 + wv) * (32 * RB);
  const bool wave_live = row0 < rows;                     // (idle waves of the last block still take the barriers)
  const int m = lane & 31, kg = lane >> 5;
  const int LD = Cin + 8;                                 // halves per staged row
  uint16_t* __restrict__ s_hi = wlds;
  uint16_t* __restrict__ s_lo = wlds + NB * 32 * LD;
  const int cpr = Cin / 8, pieces = NB * 32 * cpr;        // 16-byte pieces per row / per plane
  sf32x16_t acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Software pipeline ACROSS the offsets (round 3, session 5): offset k + 1's neighbour indices are fetched before
  // offset k's weights are staged, and its first 16-channel gather is issued in the slot of offset k's last step — a
  // block used to walk index -> barrier -> W copy -> barrier -> first gather as three serial round trips per offset,
  // 27 times (one block per CU-slot: the block's chain IS the kernel's time).
  constexpr int NV = SPLIT ? 2 : 1;                        // 16-byte loads per row and step
  int idx[RB], idn[RB];
  auto load_idx = [&](int k, int (&d)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const long row = row0 + i * 32 + m;
      d[i] = (k < kvol && wave_live && row < rows) ? nbr[(long)k * ld + row] : -1;
    }
  };
  // A fragments straight from the gathered rows (rows without a neighbour: zeros)
  auto load_a = [&](const int (&ix)[RB], int c0, uint4 (&dst)[RB][NV]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const T* frow = feats + (long)(ix[i] >= 0 ? ix[i] : 0) * Cin + kg * 8 + c0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        dst[i][v] = reinterpret_cast<const uint4*>(frow)[v];
        if (ix[i] < 0) dst[i][v] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  uint4 raw[RB][NV], nxt[RB][NV];
  load_idx(0, idx);
  load_a(idx, 0, raw);
  for (int k = 0; k < kvol; ++k) {
    bool any = false;
#pragma unroll
    for (int i = 0; i < RB; ++i) any = any || idx[i] >= 0;
    load_idx(k + 1, idn);                                 // (past the last offset: -1)
    // ---- W_k -> LDS, by the whole block
    __syncthreads();                                      // the previous offset's fragments are consumed
    {
      const uint16_t* gh = w_hi + (long)k * NB * 32 * Cin;
      const uint16_t* gl = SPLIT ? w_lo + (long)k * NB * 32 * Cin : nullptr;
      for (int p = threadIdx.x; p < pieces; p += 256) {
        const int r = p / cpr, c8 = (p - r * cpr) * 8;
        *reinterpret_cast<uint4*>(s_hi + r * LD + c8) = *reinterpret_cast<const uint4*>(gh + (long)r * Cin + c8);
        if constexpr (SPLIT)
          *reinterpret_cast<uint4*>(s_lo + r * LD + c8) = *reinterpret_cast<const uint4*>(gl + (long)r * Cin + c8);
      }
    }
    __syncthreads();
    if (__ballot(any) == 0ull) {                          // no row of this wave has a neighbour at offset k
      load_a(idn, 0, raw);
#pragma unroll
      for (int i = 0; i < RB; ++i) idx[i] = idn[i];
      continue;
    }
    const uint16_t* wk_hi = s_hi + m * LD + kg * 8;
    const uint16_t* wk_lo = SPLIT ? s_lo + m * LD + kg * 8 : nullptr;
    for (int c0 = 0; c0 < Cin; c0 += 16) {
      // the NEXT 16-channel step's loads are issued before this step's MFMAs; the last step fetches the next
      // OFFSET's first step instead
      if (c0 + 16 < Cin) load_a(idx, c0 + 16, nxt);
      else load_a(idn, 0, nxt);
      uint4 a_hi[RB], a_lo[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if constexpr (SPLIT) {
          const float4 v0 = __builtin_bit_cast(float4, raw[i][0]);
          const float4 v1 = __builtin_bit_cast(float4, raw[i][1]);
          uint4 h, l;
          h.x = cvt_pk_bf16(v0.x, v0.y); h.y = cvt_pk_bf16(v0.z, v0.w);
          h.z = cvt_pk_bf16(v1.x, v1.y); h.w = cvt_pk_bf16(v1.z, v1.w);
          l.x = cvt_pk_bf16(v0.x - __uint_as_float(h.x << 16), v0.y - __uint_as_float(h.x & 0xffff0000u));
          l.y = cvt_pk_bf16(v0.z - __uint_as_float(h.y << 16), v0.w - __uint_as_float(h.y & 0xffff0000u));
          l.z = cvt_pk_bf16(v1.x - __uint_as_float(h.z << 16), v1.y - __uint_as_float(h.z & 0xffff0000u));
          l.w = cvt_pk_bf16(v1.z - __uint_as_float(h.w << 16), v1.w - __uint_as_float(h.w & 0xffff0000u));
          a_hi[i] = h; a_lo[i] = l;
        } else {
          a_hi[i] = raw[i][0];
          a_lo[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const uint4 b_hi = *reinterpret_cast<const uint4*>(wk_hi + j * 32 * LD + c0);
        uint4 b_lo = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SPLIT) b_lo = *reinterpret_cast<const uint4*>(wk_lo + j * 32 * LD + c0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          acc[i][j] = spc_mma<T>::run(a_hi[i], b_hi, acc[i][j]);
          if constexpr (SPLIT) {
            acc[i][j] = spc_mma<T>::run(a_hi[i], b_lo, acc[i][j]);
            acc[i][j] = spc_mma<T>::run(a_lo[i], b_hi, acc[i][j]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v) raw[i][v] = nxt[i][v];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) idx[i] = idn[i];
  }
  // D[i][n]: this lane holds column n = lane & 31 (output channel within block j), rows (r & 3) + 8 (r >> 2) + 4 kg
  if (!wave_live) return;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = j * 32 + m;
      if (n >= Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long orow = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (orow < rows) out[orow * Cout + n] = elem<T>::from_float(acc[i][j][r]);
      }
    }
}

template <typename T>
static int spconv_launch(const void* feats, const int32_t* nbr, long ld, long rows, const void* w_hi, const void* w_lo,
                         void* out, int Cin, int Cout, int kvol, hipStream_t st) {
  const int nb = (Cout + 31) / 32;
  constexpr int RB = 2;
  const dim3 grid((unsigned)((rows + 128 * RB - 1) / (128 * RB))), blk(256);
  const size_t lds = (size_t)(sizeof(T) == 4 ? 2 : 1) * nb * 32 * (Cin + 8) * sizeof(uint16_t);
  // (more than 64 KB of dynamic LDS at 128 x 128 f32 channels: ask for it explicitly)
#define UBV_SPC(NBV)                                                                                            \
  case NBV:                                                                                                     \
    if (lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)spconv_gather_mma_kernel<T, NBV, RB>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
    hipLaunchKernelGGL((spconv_gather_mma_kernel<T, NBV, RB>), grid, blk, lds, st, (const T*)feats, nbr, ld, rows, \
                       (const uint16_t*)w_hi, (const uint16_t*)w_lo, (T*)out, Cin, Cout, kvol);                 \
    break;
  switch (nb) {
    UBV_SPC(1) UBV_SPC(2) UBV_SPC(3) UBV_SPC(4)
    default: return UBV_ERR_UNSUPPORTED;
  }
#undef UBV_SPC
  return UBV_OK;
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

extern "C" int ubv_spconv_candidates(const int32_t* coors, int64_t n, int B, const int* in_dims, const int* out_dims,
                                     const int* ksize, const int* stride, const int* pad, int64_t* cand, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(coors && cand && n >= 0, "spconv_candidates: bad arguments");
  SpGeom g;
  UBV_CHECK_ARG(sp_geom(g, B, in_dims, out_dims, ksize, stride, pad, 1), "spconv_candidates: bad geometry");
  if (n == 0) return UBV_OK;
  const long total = (long)n * ksize[0] * ksize[1] * ksize[2];
  hipLaunchKernelGGL(spc_candidates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), coors,
                     (long)n, g, (long long*)cand);
  UBV_CHECK_LAUNCH("spconv_candidates");
  return UBV_OK;
}

extern "C" int ubv_spconv_keys_to_coors(const int64_t* keys, int64_t n, int D, int H, int W, int32_t* coors, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(keys && coors && n >= 0 && D > 0 && H > 0 && W > 0, "spconv_keys_to_coors: bad arguments");
  if (n == 0) return UBV_OK;
  hipLaunchKernelGGL(spc_keys_to_coors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const long long*)keys, (long)n, D, H, W, coors);
  UBV_CHECK_LAUNCH("spconv_keys_to_coors");
  return UBV_OK;
}

extern "C" int ubv_spconv_gather_mma(const void* feats, const int32_t* nbr, int64_t ld, int64_t rows, const void* w_hi,
                                     const void* w_lo, void* out, int Cin, int Cout, int kvol, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(feats && nbr && w_hi && out && rows >= 0 && ld >= rows && kvol > 0, "spconv_gather_mma: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "spconv_gather_mma: unknown dtype %d", dtype);
  UBV_CHECK_ARG((dtype == UBV_F32) == (w_lo != nullptr), "spconv_gather_mma: f32 features take split weights (hi, lo)");
  if (Cin % 16 != 0 || Cout <= 0 || Cout > 128 || ((uintptr_t)feats % 16) != 0 || ((uintptr_t)w_hi % 16) != 0) {
    set_error("spconv_gather_mma: Cin=%d must be a multiple of 16, Cout=%d at most 128, buffers 16-byte aligned", Cin, Cout);
    return UBV_ERR_UNSUPPORTED;
  }
  if (rows == 0) return UBV_OK;
  hipStream_t st = as_stream(stream);
  int rc;
  if (dtype == UBV_F32) rc = spconv_launch<float>(feats, nbr, ld, rows, w_hi, w_lo, out, Cin, Cout, kvol, st);
  else if (dtype == UBV_F16) rc = spconv_launch<f16_t>(feats, nbr, ld, rows, w_hi, nullptr, out, Cin, Cout, kvol, st);
  else rc = spconv_launch<bf16_t>(feats, nbr, ld, rows, w_hi, nullptr, out, Cin, Cout, kvol, st);
  if (rc != UBV_OK) { set_error("spconv_gather_mma: no kernel for Cout=%d", Cout); return rc; }
  UBV_CHECK_LAUNCH("spconv_gather_mma");
  return UBV_OK;
}
