// Library plumbing: version, arch string, thread-local error text.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ubv_common.h"

namespace ubv {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ubv

// ---- optional per-kernel timing with HIP events on the launch stream ---------------------------------
namespace ubv {
namespace {
struct ProfRec { char name[96]; double bytes; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
bool g_prof_ops_only = false;   // level 2: whole operators only — the scopes of the kernels inside an op put their own
                                // event records between its launches and lengthen what the op-level scope measures
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t st, double bytes) : idx_(-1), st_(st) {
  if (!g_prof_on) return;
  if (g_prof_ops_only && strncmp(name, "bev_lift_fwd<", 13) != 0 && strstr(name, "_op<") == nullptr) return;
  ProfRec r;
  snprintf(r.name, sizeof(r.name), "%s", name);
  r.bytes = bytes;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
  idx_ = (long)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[idx_].e1, st_);
}
}  // namespace ubv

extern "C" int ubv_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(ubv::g_prof_mu);
  for (auto& r : ubv::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  ubv::g_prof.clear();
  ubv::g_prof_on = on != 0;
  ubv::g_prof_ops_only = on == 2;
  return UBV_OK;
}

extern "C" int64_t ubv_profile_read(char* out, int64_t capacity) {
  std::lock_guard<std::mutex> lk(ubv::g_prof_mu);
  struct Agg { long n = 0; double ms = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : ubv::g_prof) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1; a.ms += ms; a.bytes += r.bytes;
  }
  std::string text;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.1f\n", kv.first.c_str(), kv.second.n, kv.second.ms,
             kv.second.bytes / (double)kv.second.n);
    text += line;
  }
  if (out != nullptr && capacity > 0) {
    const size_t n = text.size() < (size_t)capacity - 1 ? text.size() : (size_t)capacity - 1;
    memcpy(out, text.data(), n);
    out[n] = 0;
  }
  return (int64_t)text.size() + 1;
}

// ---- test aid: fill the LDS of every CU with a bit pattern ---------------------------------------------
namespace ubv {
__global__ __launch_bounds__(256) void fill_lds_kernel(unsigned pattern, int words, unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = pattern;
  __syncthreads();
  // (a read the compiler cannot drop keeps the stores)
  if (sink != nullptr && lds_words[(threadIdx.x * 7) % words] != pattern) sink[0] = 1u;
}
}  // namespace ubv

extern "C" int ubv_debug_fill_lds(uint32_t pattern, void* stream) {
  // 2048 blocks x 64 KB: 8 waves of blocks over 256 CUs x 160 KB, every allocation slot is written at least once
  constexpr int kBytes = 64 * 1024;
  hipLaunchKernelGGL(ubv::fill_lds_kernel, dim3(2048), dim3(256), kBytes, (hipStream_t)stream, pattern, kBytes / 4,
                     (unsigned*)nullptr);
  UBV_CHECK_LAUNCH("debug_fill_lds");
  return UBV_OK;
}

// ---- test aid: synthetic "aggressor" kernels for the two-stream hazard study (tools/ab/lift_concurrent.py) ----------
// kind 0: back-to-back v_mfma_f32_32x32x16_bf16, nothing else        3: streaming global loads (a copy's read half)
//      1: gemm-like skeleton: LDS stores + barrier + b128 reads + MFMAs   4: ds_read_b64_tr_b16 loop
//      2: the same LDS traffic and barriers WITHOUT the MFMAs             5: plain VALU loop
//      6: v_cvt_pk_bf16_f32 loop    7: MFMAs on 8 independent accumulators    8: streaming copy (src's first half -> second half)
//      9: v_pk_fma_f32 loop         10: v_pk_fma_f32 between MFMAs   11: scalar v_fma_f32 between MFMAs
namespace ubv {
typedef __attribute__((ext_vector_type(8))) __bf16 dbg_bf8;
typedef __attribute__((ext_vector_type(16))) float dbg_f16v;
typedef __attribute__((ext_vector_type(4))) float dbg_f4;
typedef __attribute__((ext_vector_type(2))) int dbg_i2;
template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(const float* __restrict__ src, float* sink, int iters, long n4) {
  extern __shared__ __attribute__((aligned(16))) unsigned char albs[];
  const int tid = threadIdx.x, lane = tid & 63;
  dbg_f16v c0 = {}, c1 = {};
  dbg_f4 av = {1.0f + lane, 2.0f, 3.0f, 4.0f}, bv = {0.5f, 0.25f, 0.125f, lane * 0.01f};
  float acc = 0.0f;
  if constexpr (KIND == 0) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, av), __builtin_bit_cast(dbg_bf8, bv), c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, bv), __builtin_bit_cast(dbg_bf8, av), c1, 0, 0, 0);
    }
  } else if constexpr (KIND == 1 || KIND == 2) {
    dbg_f4* rows = reinterpret_cast<dbg_f4*>(albs);                 // [256 + 4] x 16 B per "chunk", 80-byte pitch like gemm_nt
    for (int i = 0; i < iters / 8; ++i) {
      __syncthreads();
      *reinterpret_cast<dbg_f4*>(albs + tid * 80) = av;
      *reinterpret_cast<dbg_f4*>(albs + 20480 + tid * 80) = bv;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const dbg_f4 a = *reinterpret_cast<const dbg_f4*>(albs + ((tid + 32 * k) & 255) * 80);
        const dbg_f4 b = *reinterpret_cast<const dbg_f4*>(albs + 20480 + ((tid + 64 * k) & 255) * 80);
        if constexpr (KIND == 1) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, a), __builtin_bit_cast(dbg_bf8, b), c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, b), __builtin_bit_cast(dbg_bf8, a), c1, 0, 0, 0);
        } else {
          acc += a[0] * b[1] + a[2] * b[3];
        }
      }
      av[0] += 1.0f;
    }
    (void)rows;
  } else if constexpr (KIND == 3) {
    const dbg_f4* p = reinterpret_cast<const dbg_f4*>(src);
    for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
      const dbg_f4 v = p[i];
      acc += v[0] + v[1] + v[2] + v[3];
    }
  } else if constexpr (KIND == 4) {
    *reinterpret_cast<dbg_f4*>(albs + tid * 16) = av;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
      dbg_i2 r;
      const unsigned a = (unsigned)(((tid + i) & 255) * 16);
      asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
      acc += (float)(r[0] & 0xff) + (float)(r[1] & 0xff);
    }
  } else if constexpr (KIND == 6) {
    // v_cvt_pk_bf16_f32 loop (the f32 -> bf16 split every GEMM of this library runs on its operands)
    float x = 1.0f + lane * 0.001f, y = 0.5f + lane * 0.002f;
    unsigned r = 0;
    for (int i = 0; i < iters * 8; ++i) {
      unsigned t;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(y));
      r ^= t;
      x += 1.0f;
    }
    acc = (float)(r & 0xff);
  } else if constexpr (KIND == 7) {
    // 8 INDEPENDENT accumulators: no dependency stalls between the MFMAs, the matrix pipe never idles
    dbg_f16v c[8] = {};
    for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, av), __builtin_bit_cast(dbg_bf8, bv), c[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c0 += c[k];
  } else if constexpr (KIND == 8) {
    // streaming copy: 16-byte loads and stores
    const dbg_f4* p = reinterpret_cast<const dbg_f4*>(src);
    dbg_f4* q = reinterpret_cast<dbg_f4*>(const_cast<float*>(src)) + n4;          // second half of the buffer
    for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) q[i] = p[i];
  } else if constexpr (KIND == 9 || KIND == 10 || KIND == 11) {
    // 9: v_pk_fma_f32 loop; 10: the same between MFMAs (a kernel that holds BOTH, like an SLP-built GEMM epilogue)
    typedef float dbg_f2 __attribute__((ext_vector_type(2)));
    dbg_f2 x = {1.0f + lane * 0.001f, 2.0f - lane * 0.001f};
    const dbg_f2 a = {0.9990234375f, 1.0009765625f}, b = {0.001953125f, -0.0009765625f};
    for (int i = 0; i < iters; ++i) {
      if constexpr (KIND >= 10)
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dbg_bf8, av), __builtin_bit_cast(dbg_bf8, bv), c0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (KIND == 11) {                       // the same arithmetic as two scalar FMAs: no packed instruction
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(a[1]), "v"(b[1]));
        } else {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        }
      }
    }
    acc = x[0] + x[1];
  } else {
    float x = lane * 0.001f, y = 1.0f + lane * 1e-6f;
    for (int i = 0; i < iters * 16; ++i) x = __builtin_fmaf(x, y, y);
    acc = x;
  }
  float s = acc;
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k];
  if (s == 123.456f) sink[0] = s;
}
}  // namespace ubv

extern "C" int ubv_debug_aggressor(int kind, int iters, int blocks, int lds_bytes, const float* src, int64_t n_floats,
                                   float* sink, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(kind >= 0 && kind <= 11 && blocks > 0 && lds_bytes >= 40960 && lds_bytes <= 160 * 1024 && sink != nullptr,
                "debug_aggressor: bad arguments");
  UBV_CHECK_ARG((kind != 3 && kind != 8) || (src != nullptr && n_floats >= 4), "debug_aggressor: kind 3 needs a source buffer");
  const dim3 g(blocks), b(256);
  hipStream_t st = (hipStream_t)stream;
  const long n4 = kind == 8 ? n_floats / 8 : n_floats / 4;
  switch (kind) {
    case 0: hipLaunchKernelGGL(aggressor_kernel<0>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 1: hipLaunchKernelGGL(aggressor_kernel<1>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 2: hipLaunchKernelGGL(aggressor_kernel<2>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 3: hipLaunchKernelGGL(aggressor_kernel<3>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 4: hipLaunchKernelGGL(aggressor_kernel<4>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 6: hipLaunchKernelGGL(aggressor_kernel<6>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 7: hipLaunchKernelGGL(aggressor_kernel<7>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 8: hipLaunchKernelGGL(aggressor_kernel<8>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 9: hipLaunchKernelGGL(aggressor_kernel<9>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 10: hipLaunchKernelGGL(aggressor_kernel<10>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    case 11: hipLaunchKernelGGL(aggressor_kernel<11>, g, b, lds_bytes, st, src, sink, iters, n4); break;
    default: hipLaunchKernelGGL(aggressor_kernel<5>, g, b, lds_bytes, st, src, sink, iters, n4); break;
  }
  UBV_CHECK_LAUNCH("debug_aggressor");
  return UBV_OK;
}

extern "C" int ubv_version(void) { return 100; }   // 0.1.0
extern "C" const char* ubv_last_error(void) { return ubv::g_err; }
extern "C" const char* ubv_arch(void) { return "gfx950"; }
