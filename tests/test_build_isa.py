"""The built library must hold NO packed f32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).

The SLP-vectorised build of the lifting kernels (390 packed f32 instructions in one of them) returned wrong results when
it shared a SIMD with another stream's MFMA + VALU kernel — two HIP streams are enough — and the scalar build of the same
source never does (DESIGN section 5 "Two streams", profiles/r04_two_stream_race.txt, profiles/r05_pk_mfma_hazard.txt;
the packed instructions are necessary there, not sufficient in general).  The library is built with -fno-slp-vectorize for
that reason (unibev_amd/csrc/Makefile); this test reads the device code of the built .so back, so that a changed flag, a
vector-typed float expression (an ext_vector add compiles to v_pk_add_f32 without any vectoriser) or a hand-written packed
operation cannot return unnoticed.  CPU-only: it disassembles, nothing runs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_library_has_no_packed_f32_instructions(tmp_path):
    so = os.path.join(ROOT, 'unibev_amd', 'libunibev_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    shutil.copy(so, tmp_path / 'lib.so')                      # (the extraction writes next to its input)
    subprocess.run([OBJDUMP, '--offloading', 'lib.so'], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    objs = [f for f in os.listdir(tmp_path) if f.endswith('gfx950')]
    assert objs, 'no gfx950 code objects in the library'
    packed, kernels = {}, 0
    for f in objs:
        out = subprocess.run([OBJDUMP, '-d', f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        kernels += len(re.findall(r'^[0-9a-f]+ <_Z', out, flags=re.M))
        # (the synthetic co-runners of the hazard study, ubv_debug_aggressor, hold packed instructions on purpose)
        out = re.sub(r'^[0-9a-f]+ <_ZN3ubv16aggressor_kernel[^>]*>:\n.*?s_endpgm', '', out, flags=re.M | re.S)
        for m in re.findall(r'\bv_pk_[a-z0-9_]*f32\b', out):
            packed[m] = packed.get(m, 0) + 1
    assert kernels > 100, f'disassembly looks empty ({kernels} functions)'
    assert not packed, f'packed f32 VALU instructions in the library: {packed} — build with -fno-slp-vectorize'


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_weight_stationary_gemm_kernels_do_not_spill(tmp_path):
    """csrc/gemm_ws.hip counts its vector-memory instructions by hand (its LDS-DMA requests are invisible to hipcc), so a
    scratch access — a spilled register is one — would shift every counted `s_waitcnt vmcnt`.  The kernel descriptors of
    the built library must report no private segment for any gemm_ws_kernel instantiation, and their code no scratch
    instruction."""
    so = os.path.join(ROOT, 'unibev_amd', 'libunibev_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    shutil.copy(so, tmp_path / 'lib.so')
    subprocess.run([OBJDUMP, '--offloading', 'lib.so'], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    found = 0
    for f in [f for f in os.listdir(tmp_path) if f.endswith('gfx950')]:
        out = subprocess.run([OBJDUMP, '-d', f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        # (the product instantiations: timing-study switch ABL = 0, the last template argument)
        for m in re.finditer(r'^[0-9a-f]+ <(_ZN3ubv14gemm_ws_kernelILi\d+ELi\dELi0EEE[^>]*)>:\n(.*?)s_endpgm', out, flags=re.M | re.S):
            found += 1
            body = m.group(2)
            assert 'scratch_' not in body, f'{m.group(1)} spills (scratch instructions in its code)'
            assert 'global_load_lds_dwordx4' in body and 'v_mfma_f32_32x32x16_bf16' in body
    assert found >= 12, f'only {found} gemm_ws_kernel instantiations found in the library'


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_weight_gradient_kernels_keep_their_closures_in_registers(tmp_path):
    """csrc/gemm_mfma.hip's weight-gradient kernels are built from nested lambdas; one arrangement of them (the chunk loop as
    a lambda around the product lambda, two instantiations live) made hipcc keep a closure in scratch memory and re-read it
    every chunk — no spill is reported for that.  No instantiation of gemm_wgrad_kernel / gemm_wgrad_ws_kernel may hold a
    scratch instruction."""
    so = os.path.join(ROOT, 'unibev_amd', 'libunibev_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    shutil.copy(so, tmp_path / 'lib.so')
    subprocess.run([OBJDUMP, '--offloading', 'lib.so'], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    found = 0
    for f in [f for f in os.listdir(tmp_path) if f.endswith('gfx950')]:
        out = subprocess.run([OBJDUMP, '-d', f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r'^[0-9a-f]+ <(_ZN3ubv\d+gemm_wgrad(?:_ws)?_kernelI[^>]*)>:\n(.*?)s_endpgm', out, flags=re.M | re.S):
            found += 1
            assert 'scratch_' not in m.group(2), f'{m.group(1)} uses scratch memory'
    assert found >= 12, f'only {found} weight-gradient kernels found in the library'



@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_weight_stationary_gemm_stage_issues_exactly_the_counted_memory_instructions(tmp_path):
    """csrc/gemm_ws.hip waits for its LDS-DMA tiles with a HAND-COUNTED `s_waitcnt vmcnt(PER)`, PER = E + 4 + PPT: a stage
    of an active wave must issue exactly PPT `global_load_lds_dwordx4` requests, 4 `global_store_dwordx4` and E = 4
    epilogue-operand loads (EXTRA = 1 / 2; else 0), a stage of a wave without columns exactly its PPT requests — a compiler
    that split, merged or hoisted one of them would make the kernel under-wait silently.  Checked on the built
    code: every loop body with ONE barrier (= one stage) of every product instantiation."""
    so = os.path.join(ROOT, 'unibev_amd', 'libunibev_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    shutil.copy(so, tmp_path / 'lib.so')
    subprocess.run([OBJDUMP, '--offloading', 'lib.so'], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    seen = 0
    for f in [f for f in os.listdir(tmp_path) if f.endswith('gfx950')]:
        out = subprocess.run([OBJDUMP, '-d', f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r'^([0-9a-f]+) <(_ZN3ubv14gemm_ws_kernelILi(\d+)ELi(\d)ELi0EEE[^>]*)>:\n(.*?)s_endpgm', out,
                             flags=re.M | re.S):
            base, ks, extra = int(m.group(1), 16), int(m.group(3)), int(m.group(4))
            ppt, e = ks * 16 * 8 // 512, (4 if extra in (1, 2) else 0)
            ins = []
            for ln in m.group(5).splitlines():
                mm = re.match(r'\s*([a-z_0-9]+)\s.*//\s*([0-9A-F]+):', ln)
                if mm:
                    t = re.search(r'\+0x([0-9a-f]+)>', ln)
                    ins.append((int(mm.group(2), 16), mm.group(1), base + int(t.group(1), 16) if t else None))
            kinds = set()
            for a, op, tgt in ins:
                if not (op.startswith('s_cbranch') and tgt is not None and tgt <= a):
                    continue                                    # (backward branches close loops)
                seg = [x[1] for x in ins if tgt <= x[0] <= a]
                if seg.count('s_barrier') != 1:
                    continue                                    # (an outer loop around several stages)
                got = (sum(o.startswith('global_load_lds') for o in seg), sum(o.startswith('global_store') for o in seg),
                       sum(o.startswith('global_load_dword') for o in seg))
                assert got in ((ppt, 4, e), (ppt, 0, 0)), f'{m.group(2)}: a stage issues {got}, counted ({ppt}, 4, {e})'
                assert not any(o.startswith('scratch_') or o.startswith('buffer_') for o in seg)
                kinds.add(got)
            assert (ppt, 4, e) in kinds and (ppt, 0, 0) in kinds, (m.group(2), kinds)
            seen += 1
    assert seen >= 12, f'only {seen} gemm_ws_kernel instantiations found in the library'
