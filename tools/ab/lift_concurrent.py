"""One lifting op (forward) repeated on a side stream while GEMMs run on the current stream: do its outputs stay the same?
   python tools/ab/lift_concurrent.py [pts|self] [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_lift as BL
from unibev_amd import functional as UF
dev = 'cuda'
name = sys.argv[1] if len(sys.argv) > 1 else 'pts'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
value, offlog, ref, vis0, count, gout, geom, is_grid, center = BL.instance(name, 2, torch.float32, dev, True, (256, 704), 32)
B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh = geom
print('ref contiguous:', ref.is_contiguous(), ref.dtype, tuple(ref.shape), tuple(ref.stride()), '| offlog', offlog.is_contiguous(), offlog.dtype, '| value', value.is_contiguous())
if os.environ.get('UBV_REF_CONTIG') == '1':
    ref = ref.float().contiguous()
def lift():
    return UF.bev_lift(value, offlog, ref, Nc, (fh, fw), H, P, vis0=vis0, count=count, query_grid=(qh, qw), ref_is_grid=is_grid)
ref_out = lift().clone()
torch.cuda.synchronize()
x = torch.randn(80000, 256, device=dev); w = torch.randn(256, 256, device=dev)
hi, lo, _, _ = UF.split_weight(w)
prio = int(os.environ.get('UBV_SIDE_PRIORITY', '0'))
main, side = torch.cuda.current_stream(), torch.cuda.Stream(priority=prio)
xb = x.bfloat16(); wb = w.bfloat16().contiguous(); gy = torch.randn(80000, 256, device=dev)
gam = torch.ones(256, device=dev); bet = torch.zeros(256, device=dev)
from unibev_amd._lib import lib, check
# the OTHER build of the library (UBV_OTHER_LIB), called through its C ABI directly: the co-runner's GEMM from a build
# with / without packed f32 instructions beside a victim from the build UBV_LIB_PATH names
import ctypes
other = None
if os.environ.get('UBV_OTHER_LIB'):
    other = ctypes.CDLL(os.environ['UBV_OTHER_LIB'])
    other.ubv_gemm_nt.restype = ctypes.c_int
    other.ubv_gemm_nt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p]
    yo = torch.empty(80000, 256, device=dev)
def other_gemm():
    rc = other.ubv_gemm_nt(x.data_ptr(), 256, hi.data_ptr(), lo.data_ptr(), 256, None, None, yo.data_ptr(), 256, 80000, 256, 256, 0,
                           UF._stream())
    assert rc == 0
sink = torch.zeros(16, device=dev)
xs = torch.randn(2 * 80000 * 256, device=dev)        # (kind 8 copies its first half onto its second)
SYN = {'beside syn mfma loop': 0, 'beside syn lds+barrier+mfma': 1, 'beside syn lds+barrier': 2, 'beside syn global reads': 3,
       'beside syn ds_read_tr': 4, 'beside syn valu loop': 5, 'beside syn cvt_pk_bf16 loop': 6,
       'beside syn mfma 8 accumulators': 7, 'beside syn copy': 8, 'beside syn pk_fma loop': 9, 'beside syn mfma + pk_fma': 10, 'beside syn mfma + scalar fma': 11}
def syn(kind):
    # (blocks x 256 threads, 60 KB of LDS: ubv_gemm_nt's launch geometry, 2 blocks per CU)
    check(lib().ubv_debug_aggressor(kind, 20000 if kind != 4 else 4000, 512, 61440, UF._p(xs), xs.numel(), UF._p(sink), UF._stream()), 'dbg')
modes = ('alone', 'beside gemm_nt', 'beside gemm_nt bf16', 'beside wgrad', 'beside add_norm', 'beside torch.mm') + tuple(SYN)
if os.environ.get('UBV_MODES'):
    modes = tuple(m.strip() for m in os.environ['UBV_MODES'].split(','))
print('library:', os.environ.get('UBV_LIB_PATH', 'in-tree default'))
for mode in modes:
    outs = []
    side.wait_stream(main)
    for i in range(iters):
        with torch.cuda.stream(side):
            outs.append(lift())
        if mode == 'beside gemm_nt':
            for _ in range(3): UF.gemm_nt(x, hi, lo)
        elif mode == 'beside gemm_nt bf16':
            for _ in range(3): UF.gemm_nt(xb, wb, None)
        elif mode == 'beside wgrad':
            for _ in range(2): UF.gemm_wgrad(gy, x)
        elif mode == 'beside add_norm':
            for _ in range(3): UF.add_dropout_layernorm(x, gy, gam, bet, 0.1, True)
        elif mode == 'beside torch.mm':
            for _ in range(3): torch.mm(x, w)
        elif mode == 'beside gemm_nt (other build)':
            for _ in range(3): other_gemm()
        elif mode in SYN:
            syn(SYN[mode])
        elif mode == 'beside add':
            for _ in range(6): x + 1.0
    torch.cuda.synchronize()
    bad = sum(not torch.equal(o, ref_out) for o in outs)
    worst = max(float((o - ref_out).norm() / ref_out.norm()) for o in outs)
    print(f'{name} {mode:16s}: {bad} of {iters} outputs differ, worst {worst:.1e}')
