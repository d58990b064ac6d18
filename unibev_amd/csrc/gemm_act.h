// The epilogue / operand options shared by the GEMM kernels of gemm_mfma.hip and gemm_ws.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ubv {

// Optional epilogue of the FFN GEMMs (ubv_gemm_nt_act): mode 1 y = dropout(relu(acc + bias)) with the
// keep mask of ubv_relu_dropout_forward (hash of seed and the element's index in the [M, N] output);
// mode 2 y = acc * scale where mask[m][n] != 0, else 0 — the backward of that activation applied to the
// input gradient of the NEXT Linear (mask = the activation's saved output).
struct GemmAct { int mode; const void* mask; uint32_t thresh; float scale; uint64_t seed; const uint64_t* seed_dev;
                 long res_period, res_ld;      // res_period > 0: R is a ROW-PERIODIC term, R[(m % res_period) * res_ld + n]
                 // "dual" form (ubv_gemm_nt_dual; the fused value_proj | offsets | logits GEMM of the BEV self-attention
                 // and its input gradient): X's columns k >= k_split come from a second matrix, Y's columns n >= n_split
                 // go to a second matrix (the row-periodic term then applies to those only); N need not fill the last
                 // column tile
                 const void* x2; long ldx2; int k_split; void* y2; long ldy2; int n_split; };

// Weight-stationary kernel (gemm_ws.hip): true when it took the call.
bool gemm_ws_try(const void* X, long ldx, const void* Wh, const void* Wl, long ldw, const float* bias, const void* R, void* Y,
                 long ldy, long M, int N, int K, const GemmAct& act, hipStream_t st);

}  // namespace ubv
