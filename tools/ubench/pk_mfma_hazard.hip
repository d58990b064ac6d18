// Does a packed f32 VALU instruction go wrong while ANOTHER kernel's MFMA instructions run on the same SIMD?
// (profiles/r04_two_stream_race.txt section 6 claims so from library-level experiments.)
//
// Library-free experiment: kernel A ("aggressor", stream 0) loops one MFMA opcode for a few milliseconds with a small
// register footprint and few waves, so that kernel B ("victim", stream 1) becomes co-resident on the same CUs / SIMDs.
// Kernel B computes a chain of packed f32 instructions (inline asm, so that the opcode is certain) AND the same chain
// with scalar v_fma_f32 / v_add_f32 / v_mul_f32, compares the two bit for bit per lane and counts lanes that differ.
// Rows = aggressor, columns = victim; a cell = launches with >= 1 wrong lane / launches, and wrong lanes in total.
// Aggressors: none, VALU-only, v_mfma_f32_32x32x16_bf16, ..._f16, v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x32_bf16, and
// (if libunibev_hip.so is found beside this repository: argv[1]) this library's own ubv_gemm_nt (f32 split-bf16), the
// kernel the library-level experiments failed beside.
// Victims: pk_fma, pk_add, pk_mul, pk_fma with op_sel (the form the SLP vectoriser emitted for the coordinate
// arithmetic: x and y of a float2 with one operand broadcast), and "coords": the lifting kernels' own coordinate
// arithmetic written in C++ on float2 and left to the compiler (this file is built WITH the SLP vectoriser).
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_mfma_hazard.hip -o tools/ubench/pk_mfma_hazard -ldl
//   tools/ubench/pk_mfma_hazard [path/to/libunibev_hip.so] [launches=1000]
//   GPU_MAX_HW_QUEUES=1 tools/ubench/pk_mfma_hazard ...        (both streams on one hardware queue)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

// ---------------------------------------------------------------------------------------------- aggressors
enum { A_NONE = 0, A_VALU, A_MFMA_32_BF16, A_MFMA_32_F16, A_MFMA_32_F32, A_MFMA_16_BF16, A_MFMA_PLUS_PK, A_MFMA_PLUS_VALU, A_LIB_GEMM, A_COUNT };
static const char* kAggName[A_COUNT] = {"nothing", "valu v_fma_f32 loop", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16",
                                        "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_bf16", "mfma bf16 + v_pk_fma_f32 mixed",
                                        "mfma bf16 + v_fma_f32 mixed", "libunibev ubv_gemm_nt f32"};

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  if constexpr (KIND == A_VALU) {
    float x = lane * 0.001f, y = 1.0f + lane * 1e-6f;
    for (int i = 0; i < iters * 16; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    }
    if (x == 123.456f) sink[0] = x;
  } else if constexpr (KIND == A_MFMA_32_BF16 || KIND == A_MFMA_32_F16) {
    f16v c0 = {}, c1 = {};
    f4 av = {1.0f + lane, 2.0f, 3.0f, 4.0f}, bv = {0.5f, 0.25f, 0.125f, lane * 0.01f};
    for (int i = 0; i < iters; ++i) {
      if constexpr (KIND == A_MFMA_32_BF16) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, bv), __builtin_bit_cast(bf8, av), c1, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, av), __builtin_bit_cast(h8, bv), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, bv), __builtin_bit_cast(h8, av), c1, 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k];
    if (s == 123.456f) sink[0] = s;
  } else if constexpr (KIND == A_MFMA_32_F32) {
    f16v c0 = {}, c1 = {};
    float av = 1.0f + lane * 1e-3f, bv = 0.5f;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, c1, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k];
    if (s == 123.456f) sink[0] = s;
  } else if constexpr (KIND == A_MFMA_PLUS_PK || KIND == A_MFMA_PLUS_VALU) {
    // ONE wave stream that alternates an MFMA with four VALU instructions (packed f32 / scalar f32): what a GEMM's main
    // loop looks like (operand conversion between the matrix instructions) and what the pure loops above do not have
    f16v c0 = {};
    f4 av = {1.0f + lane, 2.0f, 3.0f, 4.0f}, bv = {0.5f, 0.25f, 0.125f, lane * 0.01f};
    f2 x = {1.0f + lane * 0.001f, 2.0f - lane * 0.001f};
    const f2 a = {0.9990234375f, 1.0009765625f}, b = {0.001953125f, -0.0009765625f};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), c0, 0, 0, 0);
      for (int k = 0; k < 4; ++k) {
        if constexpr (KIND == A_MFMA_PLUS_PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(a[1]), "v"(b[1])); }
      }
    }
    float s = x[0] + x[1];
    for (int k = 0; k < 16; ++k) s += c0[k];
    if (s == 123.456f) sink[0] = s;
  } else if constexpr (KIND == A_MFMA_16_BF16) {
    f4 c0 = {}, c1 = {};
    f4 av = {1.0f + lane, 2.0f, 3.0f, 4.0f}, bv = {0.5f, 0.25f, 0.125f, lane * 0.01f};
    for (int i = 0; i < iters * 2; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, bv), __builtin_bit_cast(bf8, av), c1, 0, 0, 0);
    }
    float s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
    if (s == 123.456f) sink[0] = s;
  }
}

// ---------------------------------------------------------------------------------------------- victims
enum { V_PK_FMA = 0, V_PK_ADD, V_PK_MUL, V_PK_FMA_OPSEL, V_COORDS, V_AXPY_LDS, V_COUNT };
static const char* kVicName[V_COUNT] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32 op_sel_hi:[0,1,1]",
                                        "coords (compiler-packed)", "axpy from LDS (compiler-packed)"};

__device__ __forceinline__ float sfma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float sadd(float a, float b) {
  float d;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float smul(float a, float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// victim: `len` dependent packed operations per lane, repeated `reps` times from fresh inputs; wrong[0] += lanes whose
// packed result differs from the scalar one, wrong[1] = one example (lane id | block << 8), wrong[2..3] bits
template <int KIND>
__global__ __launch_bounds__(256) void victim(unsigned* wrong, int len, int reps, float fw, float fh) {
  __shared__ __attribute__((aligned(16))) float lds_pad[1024];                         // (an LDS allocation like the lifting kernels': co-residency)
  lds_pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  unsigned bad = 0;
  for (int r = 0; r < reps; ++r) {
    const float s0 = 1.0f + (float)((t * 7 + r * 13) & 1023) * (1.0f / 1024.0f);
    const float s1 = 2.0f - (float)((t * 5 + r * 11) & 1023) * (1.0f / 2048.0f);
    f2 x = {s0, s1};
    float y0 = s0, y1 = s1;
    const f2 a = {0.9990234375f, 1.0009765625f}, b = {0.001953125f * s1, -0.0009765625f * s0};
    if constexpr (KIND == V_AXPY_LDS) {
      // the lifting kernels' accumulation (bev_lift_tile.hip tile_axpy32): acc[32] += c * (32 floats read from LDS with
      // ds_read_b128), 4 corners x 2 points — the SLP vectoriser turns the FMAs into v_pk_fma_f32 ... op_sel_hi:[0,1,1]
      float acc[32], ref[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) { acc[k] = 0.0f; ref[k] = 0.0f; }
      for (int i = 0; i < len; ++i) {
        const float c = s0 * 0.125f + (float)i * 0.03125f;
        const float* p = lds_pad + ((threadIdx.x * 36 + i * 4) & 991);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f4 v = *reinterpret_cast<const f4*>(p + 4 * q - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3));
          acc[4 * q] = __builtin_fmaf(c, v[0], acc[4 * q]);
          acc[4 * q + 1] = __builtin_fmaf(c, v[1], acc[4 * q + 1]);
          acc[4 * q + 2] = __builtin_fmaf(c, v[2], acc[4 * q + 2]);
          acc[4 * q + 3] = __builtin_fmaf(c, v[3], acc[4 * q + 3]);
          ref[4 * q] = sfma(c, v[0], ref[4 * q]);
          ref[4 * q + 1] = sfma(c, v[1], ref[4 * q + 1]);
          ref[4 * q + 2] = sfma(c, v[2], ref[4 * q + 2]);
          ref[4 * q + 3] = sfma(c, v[3], ref[4 * q + 3]);
        }
      }
      unsigned d = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) d |= __float_as_uint(acc[k]) ^ __float_as_uint(ref[k]);
      x[0] = __uint_as_float(d); y0 = 0.0f; x[1] = 0.0f; y1 = 0.0f;
    } else if constexpr (KIND == V_COORDS) {
      // the lifting kernels' arithmetic (bev_lift_tile.hip tile_points): lx = ref + off / W; rx = lx * W - 0.5
      f2 ref = {s0 * 0.25f, s1 * 0.25f};
      const f2 wh = {fw, fh};
      for (int i = 0; i < len; ++i) {
        f2 off = {x[0] * 3.0f, x[1] * -2.0f};
        f2 l = ref + off / wh;
        x = l * wh - (f2){0.5f, 0.5f};
        x = x * (f2){0.01f, 0.01f} + (f2){1.0f, 1.0f};
        // scalar twin, kept scalar by the asm helpers
        const float o0 = smul(y0, 3.0f), o1 = smul(y1, -2.0f);
        const float l0 = sadd(ref[0], o0 / fw), l1 = sadd(ref[1], o1 / fh);
        y0 = sfma(l0, fw, -0.5f); y1 = sfma(l1, fh, -0.5f);
        y0 = sfma(y0, 0.01f, 1.0f); y1 = sfma(y1, 0.01f, 1.0f);
      }
    } else {
      for (int i = 0; i < len; ++i) {
        if constexpr (KIND == V_PK_FMA) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
          y0 = sfma(y0, a[0], b[0]); y1 = sfma(y1, a[1], b[1]);
        } else if constexpr (KIND == V_PK_ADD) {
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
          y0 = sadd(y0, b[0]); y1 = sadd(y1, b[1]);
        } else if constexpr (KIND == V_PK_MUL) {
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
          y0 = smul(y0, a[0]); y1 = smul(y1, a[1]);
        } else if constexpr (KIND == V_PK_FMA_OPSEL) {
          // hi half of src0 taken from the LOW half of the register pair: d.lo = x.lo a.lo + b.lo, d.hi = x.lo a.hi + b.hi
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]" : "+v"(x) : "v"(a), "v"(b));
          const float n0 = sfma(y0, a[0], b[0]), n1 = sfma(y0, a[1], b[1]);
          y0 = n0; y1 = n1;
        }
      }
    }
    if (__float_as_uint(x[0]) != __float_as_uint(y0) || __float_as_uint(x[1]) != __float_as_uint(y1)) {
      ++bad;
      wrong[1] = (unsigned)(threadIdx.x | (blockIdx.x << 8));
      wrong[2] = __float_as_uint(x[0]) ^ __float_as_uint(y0);
      wrong[3] = __float_as_uint(x[1]) ^ __float_as_uint(y1);
    }
  }
  if (bad) atomicAdd(&wrong[0], bad);
  if (lds_pad[(threadIdx.x * 3) & 1023] < -1.0f) wrong[4] = 1;
}

typedef int (*gemm_nt_fn)(const void*, int64_t, const void*, const void*, int64_t, const float*, const void*, void*, int64_t,
                          int64_t, int, int, int, void*);
typedef int (*split_fn)(const float*, int, int, void*, void*, void*, void*, void*);

template <int K>
static void launch_aggr(hipStream_t st, float* sink, int iters, int blocks) {
  hipLaunchKernelGGL(aggressor<K>, dim3(blocks), dim3(256), 0, st, sink, iters);
}
template <int K>
static void launch_vic(hipStream_t st, unsigned* wrong, int blocks, int len, int reps) {
  hipLaunchKernelGGL(victim<K>, dim3(blocks), dim3(256), 0, st, wrong, len, reps, 180.0f, 180.0f);
}

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : nullptr;
  const int launches = argc > 2 ? atoi(argv[2]) : 1000;
  hipStream_t sa, sv;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  float* sink;
  unsigned* wrong;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&wrong, 64));
  // optional: the library's GEMM as the aggressor
  gemm_nt_fn gemm_nt = nullptr;
  split_fn split = nullptr;
  void *gx = nullptr, *gy = nullptr, *gwh = nullptr, *gwl = nullptr;
  float* gw = nullptr;
  const int64_t M = 80000;
  const int N = 256, K = 256;
  if (libpath && strcmp(libpath, "-") != 0) {
    void* h = dlopen(libpath, RTLD_NOW);
    if (h) {
      gemm_nt = (gemm_nt_fn)dlsym(h, "ubv_gemm_nt");
      split = (split_fn)dlsym(h, "ubv_split_weight");
    }
    if (!gemm_nt || !split) fprintf(stderr, "note: %s not loaded (%s): no library-GEMM row\n", libpath, dlerror());
  }
  if (gemm_nt && split) {
    CK(hipMalloc(&gx, M * K * 4)); CK(hipMalloc(&gy, M * N * 4)); CK(hipMalloc(&gw, N * K * 4));
    CK(hipMalloc(&gwh, N * K * 2)); CK(hipMalloc(&gwl, N * K * 2));
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) & 0xffff) / 65536.0f - 0.5f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) & 0xffff) / 65536.0f - 0.5f;
    CK(hipMemcpy(gx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    if (split(gw, N, K, gwh, gwl, nullptr, nullptr, sa) != 0) { fprintf(stderr, "split_weight failed\n"); gemm_nt = nullptr; }
    CK(hipStreamSynchronize(sa));
  }

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, GPU_MAX_HW_QUEUES=%s, %d launches per cell\n", prop.gcnArchName, cus,
         getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", launches);
  printf("cell = victim launches with >= 1 wrong lane / launches (wrong lanes in total; victim launches that overlapped the aggressor)\n");
  printf("%-28s", "aggressor \\ victim");
  for (int v = 0; v < V_COUNT; ++v) printf(" | %-30s", kVicName[v]);
  printf("\n");

  // calibrate the aggressor length: ~2 ms per launch; victim ~0.3 ms; victims are launched while the aggressor runs
  const int agg_blocks = cus * 3;           // 3 blocks of 4 waves per CU (the GEMM's residency), VGPR-light
  const int vic_blocks = cus * 4;
  int total_bad_cells = 0;
  for (int a = 0; a < A_COUNT; ++a) {
    if (a == A_LIB_GEMM && !gemm_nt) continue;
    printf("%-28s", kAggName[a]);
    for (int v = 0; v < V_COUNT; ++v) {
      int bad_launches = 0, overlapped = 0;
      unsigned long long wrong_lanes = 0;
      unsigned ex[4] = {0, 0, 0, 0};
      hipEvent_t a0, a1, v0, v1;
      CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&v0)); CK(hipEventCreate(&v1));
      const int per_aggr = 4;               // victim launches per aggressor launch
      for (int it = 0; it < launches; it += per_aggr) {
        CK(hipMemsetAsync(wrong, 0, 64, sv));
        CK(hipStreamSynchronize(sv));
        CK(hipEventRecord(a0, sa));
        const int iters = 60000;
        switch (a) {
          case A_VALU: launch_aggr<A_VALU>(sa, sink, iters, agg_blocks); break;
          case A_MFMA_32_BF16: launch_aggr<A_MFMA_32_BF16>(sa, sink, iters, agg_blocks); break;
          case A_MFMA_32_F16: launch_aggr<A_MFMA_32_F16>(sa, sink, iters, agg_blocks); break;
          case A_MFMA_32_F32: launch_aggr<A_MFMA_32_F32>(sa, sink, iters / 4, agg_blocks); break;
          case A_MFMA_16_BF16: launch_aggr<A_MFMA_16_BF16>(sa, sink, iters, agg_blocks); break;
          case A_MFMA_PLUS_PK: launch_aggr<A_MFMA_PLUS_PK>(sa, sink, iters, agg_blocks); break;
          case A_MFMA_PLUS_VALU: launch_aggr<A_MFMA_PLUS_VALU>(sa, sink, iters, agg_blocks); break;
          case A_LIB_GEMM:
            for (int g = 0; g < 24; ++g)
              if (gemm_nt(gx, K, gwh, gwl, K, nullptr, nullptr, gy, N, M, N, K, 0, sa) != 0) { fprintf(stderr, "gemm_nt failed\n"); exit(2); }
            break;
          default: break;
        }
        CK(hipEventRecord(a1, sa));
        for (int k = 0; k < per_aggr; ++k) {
          if (k == 0) CK(hipEventRecord(v0, sv));
          switch (v) {
            case V_PK_FMA: launch_vic<V_PK_FMA>(sv, wrong, vic_blocks, 64, 24); break;
            case V_PK_ADD: launch_vic<V_PK_ADD>(sv, wrong, vic_blocks, 64, 24); break;
            case V_PK_MUL: launch_vic<V_PK_MUL>(sv, wrong, vic_blocks, 64, 24); break;
            case V_PK_FMA_OPSEL: launch_vic<V_PK_FMA_OPSEL>(sv, wrong, vic_blocks, 64, 24); break;
            case V_COORDS: launch_vic<V_COORDS>(sv, wrong, vic_blocks, 16, 24); break;
            case V_AXPY_LDS: launch_vic<V_AXPY_LDS>(sv, wrong, vic_blocks, 16, 12); break;
          }
        }
        CK(hipEventRecord(v1, sv));
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        unsigned hwrong[4];
        CK(hipMemcpy(hwrong, wrong, 16, hipMemcpyDeviceToHost));
        if (hwrong[0]) { bad_launches += per_aggr; wrong_lanes += hwrong[0]; memcpy(ex, hwrong, 16); }
        // overlap: did the victims start before the aggressor ended and end after it started?
        float va = 0.f, av = 0.f;
        if (a != A_NONE && hipEventElapsedTime(&va, v0, a1) == hipSuccess && hipEventElapsedTime(&av, a0, v1) == hipSuccess &&
            va > 0.f && av > 0.f)
          overlapped += per_aggr;
      }
      char cell[96];
      snprintf(cell, sizeof(cell), "%d/%d (%llu; ovl %d)", bad_launches, launches, wrong_lanes, overlapped);
      printf(" | %-30s", cell);
      fflush(stdout);
      if (bad_launches) {
        ++total_bad_cells;
        fprintf(stderr, "  [%s x %s] example: thread %u block %u, xor bits lo %08x hi %08x\n", kAggName[a], kVicName[v], ex[1] & 255,
                ex[1] >> 8, ex[2], ex[3]);
      }
      CK(hipEventDestroy(a0)); CK(hipEventDestroy(a1)); CK(hipEventDestroy(v0)); CK(hipEventDestroy(v1));
    }
    printf("\n");
  }
  printf("%s\n", total_bad_cells ? "RESULT: packed/scalar mismatches observed (see cells)" : "RESULT: no packed/scalar mismatch in any cell");
  return 0;
}
