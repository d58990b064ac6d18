"""Run-to-run spread of the full-size gradient distances (tests/test_modules_gpu.py::_fullsize_gradients): python tools/ab/grad_spread.py [runs]"""
import os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
os.environ['UBV_TEST_SHOW'] = '3'
import test_modules_gpu as T
from unibev_amd.linear import set_f32_gemm
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for gemm in ('mfma', 'library'):
    prev = set_f32_gemm(gemm)
    for i in range(runs):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T._fullsize_gradients('fullsize_smooth', norm_bar=1.0, elem_bar=10.0, fwd_bar=1e-3, allow_1d=1.0, skip_kinks=False)
        print(gemm, i, buf.getvalue().splitlines()[0])
    set_f32_gemm(prev)
