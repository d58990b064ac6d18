// Fused residual + dropout + LayerNorm, forward and backward:  y = LN(identity + dropout(x)).
//
// The post-norm encoder layer of the reference applies this pattern three times per layer
// (BaseTransformerLayer 'self_attn','norm','cross_attn','norm','ffn','norm' with every attention /
// FFN ending in  dropout(out) + identity;  encoder_unibev_detr_img.py:413-481, decoder.py:338,
// spatial_cross_attention_img.py:215, [ext] mmcv FFN).  As separate framework ops it is a dropout
// kernel, an add kernel and a LayerNorm kernel forward, and a masked-scale, three LayerNorm-backward
// kernels and a gradient add backward: ~7 passes over a (bs*40 000) x 256 f32 tensor.  Here it is
// one pass each way: one wave per row, 16-byte vector loads, wave-shuffle mean / variance, the
// dropout mask regenerated from a counter-based hash (nothing stored), gamma/beta gradients reduced
// per block in LDS and flushed with one atomic per column per block.
#include "ubv_common.h"

namespace ubv {

constexpr int kNormChunks = 4;      // C <= 64 lanes * 4 floats * 4 chunks = 1024
constexpr int kNormBwdBlocks = 512; // backward grid: 2048 waves

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) { vec_io<T, 4>::load(p, v); }
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) { vec_io<T, 4>::store(p, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// T: element type of the branch output x (and grad_x); S: element type of the residual stream
// (identity, y and their gradients) — f32, or T itself (the reference's fp16 mode keeps the
// stream in half: mmcv wrap_fp16_model; statistics and the normalisation stay in f32 either way).
template <typename T, typename S>
__global__ __launch_bounds__(256) void add_norm_fwd_kernel(
    const T* __restrict__ x, const S* __restrict__ identity, const float* __restrict__ gamma,
    const float* __restrict__ beta, S* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, long R, int C, float eps, uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // per-step base kept on the device (graph replays)
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wave; r < R; r += nwaves) {
    float s[kNormChunks][4];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kNormChunks; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
        float xv[4], iv[4];
        load4<T>(x + r * C + c, xv);
        load4<S>(identity + r * C + c, iv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float d = xv[i];
          if (thresh != 0u) d = (drop_hash(seed, (uint64_t)(r * C + c + i)) >= thresh) ? d * scale : 0.0f;
          s[k][i] = iv[i] + d;
          sum += s[k][i];
        }
      }
    }
    const float mu = wave_sum(sum) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int k = 0; k < kNormChunks; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = s[k][i] - mu; var = fmaf(d, d, var); }
      }
    }
    const float rs = rsqrtf(wave_sum(var) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < kNormChunks; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
        float g[4], b[4], o[4];
        load4<float>(gamma + c, g);
        load4<float>(beta + c, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (s[k][i] - mu) * rs * g[i] + b[i];
        store4<S>(y + r * C + c, o);
      }
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(256) void add_norm_bwd_kernel(
    const S* __restrict__ gy, const T* __restrict__ x, const S* __restrict__ identity,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    T* __restrict__ gx, S* __restrict__ gid, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dxsum, long R, int C, uint32_t thresh,
    float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // per-step base kept on the device (graph replays)
  __shared__ float red[2][4][kNormChunks * 256];       // [gamma|beta][wave][column]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float ag[kNormChunks][4], ab[kNormChunks][4];
#pragma unroll
  for (int k = 0; k < kNormChunks; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) { ag[k][i] = 0.0f; ab[k][i] = 0.0f; }

  for (long r = wave; r < R; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    float xh[kNormChunks][4], dxh[kNormChunks][4];
    float keep[kNormChunks][4];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < kNormChunks; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
        float xv[4], iv[4], g[4], go[4];
        load4<T>(x + r * C + c, xv);
        load4<S>(identity + r * C + c, iv);
        load4<float>(gamma + c, g);
        load4<S>(gy + r * C + c, go);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          keep[k][i] = 1.0f;
          if (thresh != 0u)
            keep[k][i] = (drop_hash(seed, (uint64_t)(r * C + c + i)) >= thresh) ? scale : 0.0f;
          const float s = iv[i] + xv[i] * keep[k][i];
          xh[k][i] = (s - mu) * rs;
          dxh[k][i] = go[i] * g[i];
          s1 += dxh[k][i];
          s2 = fmaf(dxh[k][i], xh[k][i], s2);
          ag[k][i] = fmaf(go[i], xh[k][i], ag[k][i]);
          ab[k][i] += go[i];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < kNormChunks; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
        float ds[4], dx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ds[i] = rs * (dxh[k][i] - s1 - xh[k][i] * s2);
          dx[i] = ds[i] * keep[k][i];
          if (dxsum != nullptr)        // rare widths: straight atomics
            atomic_add_f32(dxsum + c + i, elem<T>::to_float(elem<T>::from_float(dx[i])));
        }
        store4<S>(gid + r * C + c, ds);
        store4<T>(gx + r * C + c, dx);
      }
    }
  }
  // gamma / beta gradients: waves of the block -> LDS -> one atomic per column per block
#pragma unroll
  for (int k = 0; k < kNormChunks; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[0][wv][(k * 64 + lane) * 4 + i] = ag[k][i];
      red[1][wv][(k * 64 + lane) * 4 + i] = ab[k][i];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float g = 0.0f, b = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { g += red[0][w][c]; b += red[1][w][c]; }
    atomic_add_f32(dgamma + c, g);
    atomic_add_f32(dbeta + c, b);
  }
}

// ---- packed rows: C / VEC lanes per row (VEC = 16 bytes of T), 64 / LPR rows per wave step --------
// For C = 256 in 16-bit data a row is 32 lanes of 16-byte vectors: two rows per wave step, no idle
// lanes, half the dependent iterations of the one-row-per-wave kernels above (which remain for the
// widths this layout does not cover).  Same arithmetic, same hash, same reductions.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = LPR >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <typename T, typename S, int LPR>
__global__ __launch_bounds__(256) void add_norm_fwd_rows_kernel(
    const T* __restrict__ x, const S* __restrict__ identity, const float* __restrict__ gamma,
    const float* __restrict__ beta, S* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, long R, int C, float eps, uint32_t thresh, float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev,
    long period) {
  // period > 0: x and identity hold `period` rows that REPEAT (row r reads row r % period) — the first encoder layer's
  // self-attention, whose queries are one table for every sample of the batch; the dropout mask is per output row
  if (seed_dev != nullptr) seed += *seed_dev;   // per-step base kept on the device (graph replays)
  constexpr int VEC = 16 / elem<T>::kBytes, G = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, c = (lane % LPR) * VEC;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float g[VEC], b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i += 4) {
    float t4[4];
    load4<float>(gamma + c + i, t4);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[i + k] = t4[k];
    load4<float>(beta + c + i, t4);
#pragma unroll
    for (int k = 0; k < 4; ++k) b[i + k] = t4[k];
  }
  for (long r0 = wave * G; r0 < R; r0 += nwaves * G) {
    const long r = r0 + sub;
    const bool ok = r < R;
    const long rr = ok ? r : 0;
    const long rs_ = period > 0 ? (long)((unsigned)rr % (unsigned)period) : rr;      // source row of x / identity
    float xv[VEC], iv[VEC], s[VEC];
    vec_io<T, VEC>::load(x + rs_ * C + c, xv);
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      float t4[4];
      load4<S>(identity + rs_ * C + c + i, t4);
#pragma unroll
      for (int k = 0; k < 4; ++k) iv[i + k] = t4[k];
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float d = xv[i];
      if (thresh != 0u) d = (drop_hash(seed, (uint64_t)(rr * C + c + i)) >= thresh) ? d * scale : 0.0f;
      s[i] = iv[i] + d;
      sum += s[i];
    }
    const float mu = row_sum<LPR>(sum) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { const float d = s[i] - mu; var = fmaf(d, d, var); }
    const float rs = rsqrtf(row_sum<LPR>(var) / (float)C + eps);
    if (ok) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (s[i + k] - mu) * rs * g[i + k] + b[i + k];
        store4<S>(y + r * C + c + i, o);
      }
      if (c == 0) { mean[r] = mu; rstd[r] = rs; }
    }
  }
}

template <typename T, typename S, int LPR>
__global__ __launch_bounds__(256) void add_norm_bwd_rows_kernel(
    const S* __restrict__ gy, const T* __restrict__ x, const S* __restrict__ identity,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    T* __restrict__ gx, S* __restrict__ gid, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dxsum, long R, int C, uint32_t thresh,
    float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev, int* __restrict__ ordered_ws, long period) {
  if (seed_dev != nullptr) seed += *seed_dev;   // per-step base kept on the device (graph replays)
  constexpr int VEC = 16 / elem<T>::kBytes, G = 64 / LPR;
  __shared__ float red[3][4][64][VEC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane / LPR, c = (lane % LPR) * VEC;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float g[VEC], ag[VEC], ab[VEC], ax[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i += 4) {
    float t4[4];
    load4<float>(gamma + c + i, t4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { g[i + k] = t4[k]; ag[i + k] = 0.0f; ab[i + k] = 0.0f; ax[i + k] = 0.0f; }
  }
  // period > 0 (x / identity are `period` rows shared by R / period samples): a lane group walks the SOURCE rows and,
  // per source row, the samples' rows r + b period — x and identity are read once, and grad_x / grad_identity
  // [period, C] come out already summed over the samples (what the shared rows' producers need)
  const long Rs = period > 0 ? period : R;
  const int nrep = period > 0 ? (int)(R / period) : 1;
  for (long r0 = wave * G; r0 < Rs; r0 += nwaves * G) {
    const long r = r0 + sub;
    const bool ok = r < Rs;
    const long rsrc = ok ? r : 0;
    float xv[VEC], iv[VEC], dxs[VEC], dss[VEC];
    vec_io<T, VEC>::load(x + rsrc * C + c, xv);
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      float t4[4];
      load4<S>(identity + rsrc * C + c + i, t4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { iv[i + k] = t4[k]; dxs[i + k] = 0.0f; dss[i + k] = 0.0f; }
    }
#pragma unroll 1
    for (int b = 0; b < nrep; ++b) {
      const long rr = rsrc + (long)b * Rs;
      const float mu = mean[rr], rs = rstd[rr];
      float go[VEC], xh[VEC], dxh[VEC], keep[VEC];
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        float t4[4];
        load4<S>(gy + rr * C + c + i, t4);
#pragma unroll
        for (int k = 0; k < 4; ++k) go[i + k] = ok ? t4[k] : 0.0f;
      }
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        keep[i] = 1.0f;
        if (thresh != 0u) keep[i] = (drop_hash(seed, (uint64_t)(rr * C + c + i)) >= thresh) ? scale : 0.0f;
        const float s = iv[i] + xv[i] * keep[i];
        xh[i] = (s - mu) * rs;
        dxh[i] = go[i] * g[i];
        s1 += dxh[i];
        s2 = fmaf(dxh[i], xh[i], s2);
        ag[i] = fmaf(go[i], xh[i], ag[i]);
        ab[i] += go[i];
      }
      s1 = row_sum<LPR>(s1) / (float)C;
      s2 = row_sum<LPR>(s2) / (float)C;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float ds = rs * (dxh[i] - s1 - xh[i] * s2);
        dss[i] += ds;
        dxs[i] = fmaf(ds, keep[i], dxs[i]);
      }
    }
    if (ok) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        float ds[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ds[k] = dss[i + k];
          // column sums of grad_x AS STORED (rounded to T): the bias gradient of the Linear that
          // produced x, which would otherwise re-read grad_x
          ax[i + k] += elem<T>::to_float(elem<T>::from_float(dxs[i + k]));
        }
        store4<S>(gid + r * C + c + i, ds);
      }
      vec_io<T, VEC>::store(gx + r * C + c, dxs);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    red[0][wv][lane][i] = ag[i]; red[1][wv][lane][i] = ab[i]; red[2][wv][lane][i] = ax[i];
  }
  __syncthreads();
  const int NC = (dxsum != nullptr ? 3 : 2) * C;
  // ORDERED mode (ordered_ws != null): the block's column sums go to its row of a partials table instead of into
  // f32 atomics; add_norm_colsum_final_kernel (next launch) adds the rows in block order: the same bits on every run,
  // whatever order the blocks retire in (f32 atomics gave the 1-D parameter gradients a run-to-run spread).
  // (Summing in this launch — the last block of a group, then the last group, found through tickets — needs an
  //  agent-scope release / acquire per block, i.e. an L2 write-back on every XCD: 15.0 -> 17.7 ms per step measured.)
  float* __restrict__ part = reinterpret_cast<float*>(ordered_ws);
  for (int t = threadIdx.x; t < NC; t += 256) {
    const int which = t / C, col = t - which * C;
    const int cl = col / VEC, e = col - cl * VEC;
    float sum = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int gq = 0; gq < G; ++gq) sum += red[which][w][gq * LPR + cl][e];
    if (part != nullptr) part[(long)blockIdx.x * NC + t] = sum;
    else atomic_add_f32((which == 0 ? dgamma : which == 1 ? dbeta : dxsum) + col, sum);
  }
}

// Ordered mode, second launch: out[col] += sum over the blocks' partial rows IN BLOCK ORDER.  16 columns x 16 row groups
// per block; a thread adds its group's rows one after the other, the 16 groups are then added in group order.
__global__ __launch_bounds__(256) void add_norm_colsum_final_kernel(const float* __restrict__ part, int nrows, int NC, int C,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    float* __restrict__ dxsum) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + cl;
  const int per = (nrows + 15) / 16;
  float sum = 0.0f;
  if (t < NC) {
    // four running sums (rows r, r + 1, r + 2, r + 3 of every quad): one chain of ~80 dependent adds behind ~80 loads was
    // the kernel's whole time (10 us for 1 250 x 512 floats); the order is still fixed
    const int r0 = rg * per, r1 = min(nrows, r0 + per);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      const float a = part[(long)r * NC + t], b = part[(long)(r + 1) * NC + t];
      const float c = part[(long)(r + 2) * NC + t], d = part[(long)(r + 3) * NC + t];
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; r < r1; ++r) s0 += part[(long)r * NC + t];
    sum = (s0 + s1) + (s2 + s3);
  }
  red[rg][cl] = sum;
  __syncthreads();
  if (rg == 0 && t < NC) {
    float tot = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += red[g][cl];
    const int which = t / C, col = t - which * C;
    float* dst = (which == 0 ? dgamma : which == 1 ? dbeta : dxsum) + col;
    *dst += tot;
  }
}

// ---- FFN activation: y = dropout(relu(x)) in one pass --------------------------------------------
// ([ext] mmcv FFN: Sequential(Linear, ReLU, Dropout(ffn_drop)), 80 000 x 512 hidden rows here.)
// Same stateless keep mask as above.  Backward needs only y: y != 0 <=> (x > 0 and kept), so
// grad_x = grad_y * scale where y != 0 — no mask, no pre-activation saved.
template <typename T>
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(const T* __restrict__ x,
                                                               T* __restrict__ y, long n,
                                                               uint32_t thresh, float scale,
                                                               uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev != nullptr) seed += *seed_dev;   // per-step base kept on the device (graph replays)
  constexpr int VEC = 16 / elem<T>::kBytes;
  const long stride = (long)gridDim.x * blockDim.x * VEC;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride) {
    float v[VEC];
    vec_io<T, VEC>::load(x + i, v);
#pragma unroll
    for (int k4 = 0; k4 < VEC; k4 += 4) {
      const uint64_t mix = thresh != 0u ? drop_mix64(seed, (uint64_t)(i + k4) >> 2) : 0ull;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r = fmaxf(v[k4 + e], 0.0f);
        if (thresh != 0u) r = drop_keep16(mix, e, thresh) ? r * scale : 0.0f;
        v[k4 + e] = r;
      }
    }
    vec_io<T, VEC>::store(y + i, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const T* __restrict__ gy,
                                                               const T* __restrict__ y,
                                                               T* __restrict__ gx, long n,
                                                               float scale) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  const long stride = (long)gridDim.x * blockDim.x * VEC;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride) {
    float g[VEC], o[VEC];
    vec_io<T, VEC>::load(gy + i, g);
    vec_io<T, VEC>::load(y + i, o);
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] = (o[k] != 0.0f) ? g[k] * scale : 0.0f;
    vec_io<T, VEC>::store(gx + i, g);
  }
}

// bcast_rows > 0: x / identity are [bcast_rows, C] and repeat over the R rows (rows kernels only: C / (16-byte lanes) in
// {16, 32, 64}; R below 2^31)
static int bcast_check(long R, long bcast_rows, int C, int dtype, const char* who) {
  if (bcast_rows == 0) return UBV_OK;
  const int vec = dtype == UBV_F32 ? 4 : 8;
  const int lpr = (C % vec == 0) ? C / vec : 0;
  if (bcast_rows < 0 || R % bcast_rows != 0 || R >= (1L << 31) || !(lpr == 16 || lpr == 32 || lpr == 64)) {
    set_error("%s: repeated rows need R %% bcast_rows == 0, R < 2^31 and a row of 16, 32 or 64 16-byte lanes (C = %d)", who, C);
    return UBV_ERR_UNSUPPORTED;
  }
  return UBV_OK;
}

static int norm_check(long R, int C, int dtype, int stream_dtype, const char* who) {
  UBV_CHECK_ARG(R >= 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * kNormChunks,
                "%s: C=%d must be a multiple of 4 and <= %d", who, C, 64 * 4 * kNormChunks);
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "%s: unknown dtype %d", who, dtype);
  UBV_CHECK_ARG(stream_dtype == UBV_F32 || stream_dtype == dtype,
                "%s: stream_dtype %d must be f32 or equal dtype %d", who, stream_dtype, dtype);
  return UBV_OK;
}

// Packed-rows kernels where a row is 16, 32 or 64 lanes of 16-byte vectors, else one row per wave.
template <typename T, typename S>
static void norm_fwd_launch(dim3 grid, hipStream_t st, const void* x, const void* identity,
                            const float* gamma, const float* beta, void* y, float* mean,
                            float* rstd, long R, int C, float eps, uint32_t th, float sc,
                            uint64_t seed, const uint64_t* seed_dev, long period) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  const int lpr = (C % VEC == 0) ? C / VEC : 0;
  auto run = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, (const T*)x, (const S*)identity, gamma, beta,
                       (S*)y, mean, rstd, R, C, eps, th, sc, seed, seed_dev, period);
  };
  if (lpr == 64) run(add_norm_fwd_rows_kernel<T, S, 64>);
  else if (lpr == 32) run(add_norm_fwd_rows_kernel<T, S, 32>);
  else if (lpr == 16) run(add_norm_fwd_rows_kernel<T, S, 16>);
  else
    hipLaunchKernelGGL((add_norm_fwd_kernel<T, S>), grid, dim3(256), 0, st, (const T*)x, (const S*)identity, gamma, beta,
                       (S*)y, mean, rstd, R, C, eps, th, sc, seed, seed_dev);
}

template <typename T, typename S>
static void norm_bwd_launch(dim3 grid, hipStream_t st, const void* gy, const void* x,
                            const void* identity, const float* gamma, const float* mean,
                            const float* rstd, void* gx, void* gid, float* dgamma, float* dbeta,
                            float* dxsum, long R, int C, uint32_t th, float sc, uint64_t seed, const uint64_t* seed_dev,
                            int* ordered_ws, long period) {
  constexpr int VEC = 16 / elem<T>::kBytes;
  const int lpr = (C % VEC == 0) ? C / VEC : 0;
  auto run = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, (const S*)gy, (const T*)x, (const S*)identity,
                       gamma, mean, rstd, (T*)gx, (S*)gid, dgamma, dbeta, dxsum, R, C, th, sc, seed, seed_dev, ordered_ws, period);
  };
  auto run_plain = [&](auto kernel) {       // rare widths: one row per wave, atomics
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, (const S*)gy, (const T*)x, (const S*)identity,
                       gamma, mean, rstd, (T*)gx, (S*)gid, dgamma, dbeta, dxsum, R, C, th, sc, seed, seed_dev);
  };
  if (lpr == 64) run(add_norm_bwd_rows_kernel<T, S, 64>);
  else if (lpr == 32) run(add_norm_bwd_rows_kernel<T, S, 32>);
  else if (lpr == 16) run(add_norm_bwd_rows_kernel<T, S, 16>);
  else { run_plain(add_norm_bwd_kernel<T, S>); return; }
  if (ordered_ws != nullptr) {
    const int NC = (dxsum != nullptr ? 3 : 2) * C;
    hipLaunchKernelGGL(add_norm_colsum_final_kernel, dim3((NC + 15) / 16), dim3(256), 0, st, (const float*)ordered_ws,
                       (int)grid.x, NC, C, dgamma, dbeta, dxsum);
  }
}

}  // namespace ubv

extern "C" int ubv_add_dropout_layernorm_forward(const void* x, const void* identity,
                                                 const float* gamma, const float* beta, void* y,
                                                 float* mean, float* rstd, int64_t R, int64_t bcast_rows,
                                                 int C, float eps, float p, uint64_t seed,
                                                 const uint64_t* seed_dev, int dtype,
                                                 int stream_dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && identity && gamma && beta && y && mean && rstd, "add_norm_forward: null pointer");
  int rc = norm_check(R, C, dtype, stream_dtype, "add_norm_forward");
  if (rc) return rc;
  rc = bcast_check(R, bcast_rows, C, dtype, "add_norm_forward");
  if (rc) return rc;
  if (R == 0) return UBV_OK;
  uint32_t th; float sc;
  drop_params(p, th, sc);
  const long waves = R < 8192 ? R : 8192;
  const dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t st = as_stream(stream);
#define UBV_NORM_FWD(T, S) norm_fwd_launch<T, S>(grid, st, x, identity, gamma, beta, y, mean, rstd, (long)R, C, eps, th, sc, seed, seed_dev, (long)bcast_rows)
  const bool lowp = stream_dtype != UBV_F32;
  switch (dtype) {
    case UBV_F32: UBV_NORM_FWD(float, float); break;
    case UBV_F16: if (lowp) UBV_NORM_FWD(f16_t, f16_t); else UBV_NORM_FWD(f16_t, float); break;
    default: if (lowp) UBV_NORM_FWD(bf16_t, bf16_t); else UBV_NORM_FWD(bf16_t, float); break;
  }
#undef UBV_NORM_FWD
  UBV_CHECK_LAUNCH("add_norm_forward");
  return UBV_OK;
}

// ordered-mode workspace: one row of 3 C sums per block
extern "C" int64_t ubv_add_dropout_layernorm_backward_workspace(int C) {
  return (int64_t)ubv::kNormBwdBlocks * 3 * (int64_t)C * 4;
}

extern "C" int ubv_add_dropout_layernorm_backward(const void* grad_y, const void* x,
                                                  const void* identity, const float* gamma,
                                                  const float* mean, const float* rstd, void* grad_x,
                                                  void* grad_identity, float* grad_gamma,
                                                  float* grad_beta, float* grad_x_colsum,
                                                  int64_t R, int64_t bcast_rows, int C, float p,
                                                  uint64_t seed, const uint64_t* seed_dev, int dtype,
                                                  int stream_dtype, void* ordered_workspace, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(grad_y && x && identity && gamma && mean && rstd && grad_x && grad_identity &&
                    grad_gamma && grad_beta, "add_norm_backward: null pointer");
  UBV_CHECK_ARG(((uintptr_t)ordered_workspace % 16) == 0, "add_norm_backward: ordered_workspace must be 16-byte aligned");
  int rc = norm_check(R, C, dtype, stream_dtype, "add_norm_backward");
  if (rc) return rc;
  rc = bcast_check(R, bcast_rows, C, dtype, "add_norm_backward");
  if (rc) return rc;
  if (R == 0) return UBV_OK;
  uint32_t th; float sc;
  drop_params(p, th, sc);
  const long waves = R < 2048 ? R : 2048;
  const dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t st = as_stream(stream);
#define UBV_NORM_BWD(T, S) norm_bwd_launch<T, S>(grid, st, grad_y, x, identity, gamma, mean, rstd, grad_x, grad_identity, grad_gamma, grad_beta, grad_x_colsum, (long)R, C, th, sc, seed, seed_dev, (int*)ordered_workspace, (long)bcast_rows)
  const bool lowp = stream_dtype != UBV_F32;
  switch (dtype) {
    case UBV_F32: UBV_NORM_BWD(float, float); break;
    case UBV_F16: if (lowp) UBV_NORM_BWD(f16_t, f16_t); else UBV_NORM_BWD(f16_t, float); break;
    default: if (lowp) UBV_NORM_BWD(bf16_t, bf16_t); else UBV_NORM_BWD(bf16_t, float); break;
  }
#undef UBV_NORM_BWD
  UBV_CHECK_LAUNCH("add_norm_backward");
  return UBV_OK;
}

extern "C" int ubv_relu_dropout_forward(const void* x, void* y, int64_t n, float p, uint64_t seed,
                                        const uint64_t* seed_dev, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && y && n >= 0, "relu_dropout_forward: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "relu_dropout_forward: unknown dtype %d", dtype);
  const int vec = dtype == UBV_F32 ? 4 : 8;
  UBV_CHECK_ARG(n % vec == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                "relu_dropout_forward: n must be a multiple of %d and the buffers 16-byte aligned", vec);
  if (n == 0) return UBV_OK;
  uint32_t th; float sc;
  drop_params(p, th, sc);
  const long threads = n / vec;
  const dim3 grid((unsigned)min((threads + 255) / 256, 8192L));
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32: hipLaunchKernelGGL(relu_dropout_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, (long)n, th, sc, seed, seed_dev); break;
    case UBV_F16: hipLaunchKernelGGL(relu_dropout_fwd_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (long)n, th, sc, seed, seed_dev); break;
    default: hipLaunchKernelGGL(relu_dropout_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (long)n, th, sc, seed, seed_dev); break;
  }
  UBV_CHECK_LAUNCH("relu_dropout_forward");
  return UBV_OK;
}

extern "C" int ubv_relu_dropout_backward(const void* grad_y, const void* y, void* grad_x, int64_t n,
                                         float p, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(grad_y && y && grad_x && n >= 0, "relu_dropout_backward: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "relu_dropout_backward: unknown dtype %d", dtype);
  const int vec = dtype == UBV_F32 ? 4 : 8;
  UBV_CHECK_ARG(n % vec == 0 && ((uintptr_t)grad_y % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                    ((uintptr_t)grad_x % 16) == 0,
                "relu_dropout_backward: n must be a multiple of %d and the buffers 16-byte aligned", vec);
  if (n == 0) return UBV_OK;
  uint32_t th; float sc;
  drop_params(p, th, sc);
  const long threads = n / vec;
  const dim3 grid((unsigned)min((threads + 255) / 256, 8192L));
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32: hipLaunchKernelGGL(relu_dropout_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)grad_y, (const float*)y, (float*)grad_x, (long)n, sc); break;
    case UBV_F16: hipLaunchKernelGGL(relu_dropout_bwd_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)grad_y, (const f16_t*)y, (f16_t*)grad_x, (long)n, sc); break;
    default: hipLaunchKernelGGL(relu_dropout_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)grad_y, (const bf16_t*)y, (bf16_t*)grad_x, (long)n, sc); break;
  }
  UBV_CHECK_LAUNCH("relu_dropout_backward");
  return UBV_OK;
}
