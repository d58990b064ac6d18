"""Linear + add + dropout + LayerNorm: one kernel (ubv_gemm_nt_addnorm) against ubv_gemm_nt followed by
ubv_add_dropout_layernorm_forward, f32, M = 80 000 rows, operands rotated over 8 buffer sets (HBM-cold, as in the step).
python tools/ab/gemm_addnorm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unibev_amd import functional as UF
from unibev_amd._lib import lib, check
dev = 'cuda'
M = 80000
P = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
for N, K in ((256, 256), (256, 512), (128, 128)):
    NB = 8
    xs = [torch.randn(M, K, device=dev) for _ in range(NB)]
    ids = [torch.randn(M, N, device=dev) for _ in range(NB)]
    w = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    wh, wl, _, _ = UF.split_weight(w, transposed=False)
    lin = [torch.empty(M, N, device=dev) for _ in range(NB)]
    y = [torch.empty(M, N, device=dev) for _ in range(NB)]
    s = [torch.empty(M, N, device=dev) for _ in range(NB)]
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    st = UF._stream()
    def two(i):
        check(lib().ubv_gemm_nt(UF._p(xs[i]), K, UF._p(wh), UF._p(wl), K, UF._p(bias), None, UF._p(lin[i]), N, M, N, K, 0, st), 'g')
        check(lib().ubv_add_dropout_layernorm_forward(UF._p(lin[i]), UF._p(ids[i]), UF._p(gamma), UF._p(beta), UF._p(y[i]), UF._p(mean),
                                                      UF._p(rstd), M, 0, N, 1e-5, P, 7, None, 0, 0, st), 'n')
    def one(i, with_s=True):
        check(lib().ubv_gemm_nt_addnorm(UF._p(xs[i]), K, UF._p(wh), UF._p(wl), K, UF._p(bias), UF._p(ids[i]), UF._p(gamma), UF._p(beta),
                                        UF._p(y[i]), UF._p(s[i]) if with_s else None, UF._p(mean), UF._p(rstd), M, N, K, 1e-5, P, 7, None, st), 'f')
    def gemm_only(i):
        check(lib().ubv_gemm_nt(UF._p(xs[i]), K, UF._p(wh), UF._p(wl), K, UF._p(bias), None, UF._p(lin[i]), N, M, N, K, 0, st), 'g')
    def timeit(fn, n=40):
        for i in range(8): fn(i % NB)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n): fn(i % NB)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1000
    print(f'p={P} N={N} K={K}: gemm_nt {timeit(gemm_only):6.1f} us   gemm_nt + add_norm_fwd {timeit(two):6.1f} us   '
          f'one kernel (y + s) {timeit(one):6.1f} us   one kernel (y only) {timeit(lambda i: one(i, False)):6.1f} us')
