"""Debug mode for the two-stream region of the encoders (modules/transformer.py: the image encoder on the caller's
stream, the point-cloud encoder on a side stream, forked and joined around them — forward, and again when autograd
replays the backward on the forward ops' streams).

Every kernel of this library is a plain-f32 kernel by construction (csrc/Makefile: -fno-slp-vectorize, checked on the
built ISA by tests/test_build_isa.py) because kernels holding packed f32 instructions returned wrong results beside
another stream's MFMA + VALU kernels (profiles/r05_pk_mfma_hazard.txt).  Framework kernels carry no such guarantee, so
none may run while the two streams are both active.  ``ForeignKernelLog`` records every framework (aten) operator that
touches device memory — this library's own kernels are launched through the C ABI and never pass the dispatcher — with
the stream it ran on; ``offenders()`` returns those that ran inside a fork/join window.  tests/test_region_gpu.py fails
on any.
"""
import os
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

# operators that launch nothing: allocation without a fill, metadata, aliases (views are recognised by OpOverload.is_view)
_NO_KERNEL = ('aten.empty', 'aten.empty_like', 'aten.empty_strided', 'aten.new_empty', 'aten.new_empty_strided', 'aten.detach',
              'aten.alias', 'aten.lift_fresh', 'aten._unsafe_view', 'aten.record_stream', 'aten.is_pinned', 'aten.set_',
              'aten.resize_', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.is_same_size', 'aten._reshape_alias',
              'aten.view_as_real', 'aten.result_type', 'aten.is_nonzero', 'aten.unbind', 'aten.split', 'aten.chunk',
              'aten.expand_as', 'aten.reshape', 'aten.flatten', 'aten.contiguous', 'aten.to.', 'aten.type_as',
              'aten._has_compatible_shallow_copy_type', 'aten.view_as', 'aten.squeeze', 'aten.unsqueeze', 'prim.')


def _tensors(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for y in x:
            _tensors(y, out)


class ForeignKernelLog(TorchDispatchMode):
    """``with ForeignKernelLog(side_stream) as log: forward(); log.mark('backward'); backward()`` — then
    ``log.offenders()``.  Also hooks the C ABI (``_lib.set_call_hook``) so that this library's own launches are in the
    log with their stream: a fork/join window is the span from the first to the last side-stream entry of a phase."""

    # entry points that launch nothing (queries of sizes / support, profiling switches)
    _QUERIES = ('_workspace', '_supported', '_splits', '_elems', '_slots', '_words', '_chunks', 'ubv_last_error', 'ubv_version',
                'ubv_arch', 'ubv_profile_', 'ubv_adamw_flat_max_groups', 'ubv_debug_')

    def __init__(self, side_stream):
        super().__init__()
        self.side = side_stream.cuda_stream
        self.device = side_stream.device
        self.phase = 'forward'
        self.log = []                                   # (phase, operator, on side stream, call site, shape)
        self.marks = {}                                 # 'fork' / 'join' / 'bwd_fork' (first) / 'bwd_mark' (last) -> log index

    def __enter__(self):
        from . import _lib
        from .modules import transformer as _tr
        self._prev_hook = _lib.set_call_hook(self._own)
        self._prev_region, _tr._REGION_HOOK[0] = _tr._REGION_HOOK[0], self._region
        return super().__enter__()

    def __exit__(self, *exc):
        from . import _lib
        from .modules import transformer as _tr
        _lib.set_call_hook(self._prev_hook)
        _tr._REGION_HOOK[0] = self._prev_region
        return super().__exit__(*exc)

    def mark(self, phase):
        self.phase = phase

    def _region(self, tag):
        """modules/transformer.py reports the window borders: 'fork' / 'join' around the forward's two streams; in the
        backward 'bwd_fork' when a gradient enters an encoder (the first one opens the window) and 'bwd_mark' when one
        leaves through an encoder input (the last one closes it)."""
        if tag == 'bwd_fork':
            self.marks.setdefault(tag, len(self.log))
        else:
            self.marks[tag] = len(self.log)

    def _on_side(self):
        return torch.cuda.current_stream(self.device).cuda_stream == self.side

    def _own(self, name):
        if any(q in name for q in self._QUERIES):
            return
        self.log.append((self.phase, 'ubv:' + name, self._on_side(), '', ()))

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        ts = []
        _tensors(args, ts)
        _tensors(list((kwargs or {}).values()), ts)
        _tensors(out, ts)
        if not any(t.is_cuda for t in ts):
            return out
        if getattr(func, 'is_view', False) or any(name.startswith(p) for p in _NO_KERNEL):
            return out
        st = traceback.extract_stack()
        site = [f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in st if 'unibev_amd' in f.filename
                and not f.filename.endswith('debug.py')][-2:]
        self.log.append((self.phase, name, self._on_side(), ' <- '.join(reversed(site)) or 'autograd engine',
                         tuple(ts[0].shape) if ts else ()))
        return out

    def windows(self):
        """{phase: (first, last)} log indices of the two-stream windows, from the borders modules/transformer.py reported
        (host order inside a window is one encoder after the other; on the device they overlap)."""
        wins = {}
        if 'fork' in self.marks and 'join' in self.marks:
            wins['forward'] = (self.marks['fork'], self.marks['join'] - 1)
        if 'bwd_fork' in self.marks and 'bwd_mark' in self.marks:
            wins['backward'] = (self.marks['bwd_fork'], self.marks['bwd_mark'] - 1)
        return wins

    def offenders(self):
        """Framework operators launched between a fork and its join, on either stream:
        {(phase, operator, call site, shape): count}."""
        bad = {}
        for phase, (a, b) in self.windows().items():
            for ph, name, _side, site, shape in self.log[a:b + 1]:
                if not name.startswith('ubv:'):
                    bad[(phase, name, site, shape)] = bad.get((phase, name, site, shape), 0) + 1
        return bad

    def summary(self):
        wins = self.windows()
        return {ph: {'entries': b - a + 1, 'own': sum(1 for e in self.log[a:b + 1] if e[1].startswith('ubv:')),
                     'side': sum(1 for e in self.log[a:b + 1] if e[2])} for ph, (a, b) in wins.items()}
