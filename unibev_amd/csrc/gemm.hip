// Library GEMMs of the encoder's Linear layers, called with CACHED plans.
//
//   y[M, N] = x[M, K] . w[N, K]^T (+ bias[N])          row-major, one 16-bit or f32 type, f32 accumulate
//
// The GEMMs themselves are hipBLASLt's (MFMA kernels picked by its heuristic): nothing to gain by
// rewriting them, they run at their HBM bound (M = 80 000, K = N = 256..512).  What this entry point
// removes is HOST time: through the framework every call rebuilds the matmul descriptor and matrix
// layouts and queries the heuristic again — 29 us per F.linear on the host against 27 us of kernel,
// and with 48 Linear layers per pass that made the forward half of the training step host-bound
// (5.3 ms of host work for 3.9 ms of kernels).  Here descriptor, layouts and
// algorithm are built once per (M, N, K, type, bias) and a call is hipblasLtMatmul alone.
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "ubv_common.h"

namespace ubv {

struct GemmPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

static std::mutex g_gemm_mu;
static std::map<int, hipblasLtHandle_t> g_lt;       // one handle per device (created under that device)
static std::map<std::tuple<long, int, int, int, int, int>, GemmPlan> g_plans;
constexpr size_t kGemmWorkspace = 32u << 20;

static hipDataType lt_type(int dtype) {
  return dtype == UBV_F32 ? HIP_R_32F : dtype == UBV_F16 ? HIP_R_16F : HIP_R_16BF;
}

#define UBV_LT(call)                                                        \
  do {                                                                      \
    const hipblasStatus_t s__ = (call);                                     \
    if (s__ != HIPBLAS_STATUS_SUCCESS) {                                    \
      set_error("linear_forward: %s failed with hipblas status %d", #call, (int)s__); \
      return nullptr;                                                       \
    }                                                                       \
  } while (0)

// Column-major view of the row-major product: D^T[N, M] = op_T(W^T[K, N]) . X^T[K, M].
static GemmPlan* gemm_plan(long M, int N, int K, int dtype, int has_bias, int device) {
  const auto key = std::make_tuple(M, N, K, dtype, has_bias, device);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return it->second.ok ? &it->second : nullptr;
  GemmPlan& p = g_plans[key];
  if (g_lt.find(device) == g_lt.end()) {
    hipblasLtHandle_t hnd = nullptr;
    UBV_LT(hipblasLtCreate(&hnd));
    g_lt[device] = hnd;
  }
  const hipDataType t = lt_type(dtype);
  UBV_LT(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  UBV_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  UBV_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  if (has_bias) {
    const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
    UBV_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    const int32_t bt = (int32_t)t;
    UBV_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    const void* dummy = nullptr;       // the heuristic wants to see a bias pointer attribute
    UBV_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy)));
  }
  UBV_LT(hipblasLtMatrixLayoutCreate(&p.la, t, (uint64_t)K, (uint64_t)N, (int64_t)K));
  UBV_LT(hipblasLtMatrixLayoutCreate(&p.lb, t, (uint64_t)K, (uint64_t)M, (int64_t)K));
  UBV_LT(hipblasLtMatrixLayoutCreate(&p.lc, t, (uint64_t)N, (uint64_t)M, (int64_t)N));
  hipblasLtMatmulPreference_t pref = nullptr;
  UBV_LT(hipblasLtMatmulPreferenceCreate(&pref));
  const uint64_t ws = kGemmWorkspace;
  UBV_LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  const hipblasStatus_t hs =
      hipblasLtMatmulAlgoGetHeuristic(g_lt[device], p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (hs != HIPBLAS_STATUS_SUCCESS || found < 1) {
    set_error("linear_forward: no hipBLASLt algorithm for M=%ld N=%d K=%d dtype=%d (status %d)", M, N, K,
              dtype, (int)hs);
    return nullptr;
  }
  p.algo = res[0].algo;
  p.ws = res[0].workspaceSize;
  p.ok = true;
  return &p;
}

}  // namespace ubv

extern "C" int64_t ubv_linear_workspace(void) { return (int64_t)ubv::kGemmWorkspace; }

extern "C" int ubv_linear_forward(const void* x, const void* w, const void* bias, void* y, int64_t M,
                                  int N, int K, int dtype, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(x && w && y && M >= 0 && N > 0 && K > 0, "linear_forward: bad arguments");
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "linear_forward: unknown dtype %d", dtype);
  if (M == 0) return UBV_OK;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lock(g_gemm_mu);
  GemmPlan* p = gemm_plan((long)M, N, K, dtype, bias != nullptr ? 1 : 0, device);
  if (p == nullptr) return UBV_ERR_UNSUPPORTED;
  UBV_CHECK_ARG(p->ws == 0 || (workspace != nullptr && workspace_bytes >= (int64_t)p->ws),
                "linear_forward: workspace of %lld bytes needed, got %lld", (long long)p->ws,
                (long long)workspace_bytes);
  if (bias != nullptr) {
    const hipblasStatus_t s = hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER,
                                                              &bias, sizeof(bias));
    if (s != HIPBLAS_STATUS_SUCCESS) { set_error("linear_forward: bias pointer (status %d)", (int)s); return UBV_ERR_LAUNCH; }
  }
  const float alpha = 1.0f, beta = 0.0f;
  const hipblasStatus_t s = hipblasLtMatmul(g_lt[device], p->desc, &alpha, w, p->la, x, p->lb, &beta, y, p->lc, y,
                                            p->lc, &p->algo, workspace, (size_t)workspace_bytes,
                                            as_stream(stream));
  if (s != HIPBLAS_STATUS_SUCCESS) {
    set_error("linear_forward: hipblasLtMatmul failed with status %d", (int)s);
    return UBV_ERR_LAUNCH;
  }
  return UBV_OK;
}
