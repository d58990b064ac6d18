"""Autograd operators over the C ABI of ``libunibev_hip.so``.

These are the operator-level drop-ins for the reference's un-vendored extension ops
(SURVEY.md section 8(b)): same names, argument meaning and error behaviour.  Every function
requires CUDA(HIP) tensors; there is no CPU path (``UniBEVHipError`` / ``RuntimeError``).
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, lib

_DT = {torch.float32: _lib.UBV_F32, torch.float16: _lib.UBV_F16, torch.bfloat16: _lib.UBV_BF16}

# optional per-op HIP-event timing (bench.py): name -> list of (start, end) events
_PROFILE = None


def enable_profile(flag=True):
    global _PROFILE
    _PROFILE = {} if flag else None


def profile_results():
    """name -> list of (milliseconds, meta) per call; synchronises."""
    out = {}
    if _PROFILE is None:
        return out
    torch.cuda.synchronize()
    for k, evs in _PROFILE.items():
        out[k] = [(s.elapsed_time(e), m) for s, e, m in evs]
    return out


class _timed:
    """HIP events on the launch stream around one C-ABI call (only while profiling is on)."""

    def __init__(self, name, meta=None):
        self.name = name
        self.meta = meta

    def __enter__(self):
        if _PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if _PROFILE is not None:
            self.e.record()
            _PROFILE.setdefault(self.name, []).append((self.s, self.e, self.meta))


def kernel_profile(enable=None, ops_only=False):
    """Per-kernel HIP-event timing inside the library (``ubv_profile_enable`` /
    ``ubv_profile_read``).  ``kernel_profile(True)`` starts, ``kernel_profile(False)`` stops;
    ``kernel_profile()`` returns {kernel name: dict(launches, total_ms, avg_us, bytes_per_launch)}
    (synchronises on the recorded events)."""
    if enable is not None:
        # ops_only: scopes around whole operators only (the per-kernel scopes record events between an op's
        # launches, which the op-level scope then includes)
        check(lib().ubv_profile_enable((2 if ops_only else 1) if enable else 0), 'profile_enable')
        return None
    n = lib().ubv_profile_read(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    lib().ubv_profile_read(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, nb = line.split('\t')
        cnt, ms, nb = int(cnt), float(ms), float(nb)
        out[name] = dict(launches=cnt, total_ms=ms, avg_us=1e3 * ms / max(cnt, 1),
                         bytes_per_launch=nb)
    return out


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'unsupported dtype {t.dtype}; expected float32, float16 or bfloat16')


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def _need_cuda(*tensors):
    """Validates that every given tensor lives on ONE HIP device and returns a context manager that
    makes it the current device for the duration of the launch (the kernels launch on the current
    device and ``_stream()`` reads the current device's stream, like mmcv's CUDAGuard on
    ``tensor.device()``).  A no-op object when that device already is current."""
    idx = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('unibev_amd ops run on the GPU only (got a CPU tensor); '
                               'there is no CPU fallback')
        if idx is None:
            idx = t.device.index
        elif t.device.index != idx:
            raise RuntimeError(f'unibev_amd op: operands on different devices (cuda:{idx} and '
                               f'cuda:{t.device.index})')
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    # raw handle of torch's current stream on the CURRENT device — every op enters the guard
    # returned by _need_cuda first, so that is the operands' device (no Stream object per launch:
    # the step is host-bound in its forward half, every microsecond per call shows)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


# ----------------------------------------------------------------------------------------------- k1
_K1_PLAN = [os.environ.get('UBV_K1_PLAN', '1') != '0']        # 0: keep the atomic grad_value kernel everywhere


class MultiScaleDeformableAttnFunction(Function):
    """Drop-in for [ext] mmcv ``MultiScaleDeformableAttnFunction`` (call sites
    spatial_cross_attention_img.py:433-435, spatial_cross_attention_pts.py:440-442,
    decoder.py:325-327): ``apply(value, spatial_shapes, level_start_index, sampling_locations,
    attention_weights, im2col_step)`` -> (B, Nq, H*Dh); grads (value, None, None, loc, weight, None).
    """

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step=64):
        with _need_cuda(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                   attention_weights):
            B, S, H, Dh = value.shape
            _, Nq, H2, L, P, two = sampling_locations.shape
            if H2 != H or two != 2 or attention_weights.shape != (B, Nq, H, L, P):
                raise ValueError('ms_deform_attn: inconsistent shapes '
                                 f'{tuple(value.shape)} {tuple(sampling_locations.shape)} '
                                 f'{tuple(attention_weights.shape)}')
            value = value.contiguous()
            ss = value_spatial_shapes.to(torch.int64).contiguous()
            ls = value_level_start_index.to(torch.int64).contiguous()
            loc = sampling_locations.float().contiguous()
            aw = attention_weights.float().contiguous()
            out = torch.empty(B, Nq, H * Dh, dtype=value.dtype, device=value.device)
            # query-grid hint (deform_attn.shapes_tensor(..., query_grid=(qh, qw))): BEV queries in row-major order and a
            # host-known level shape -> the TILE plan (ubv_ms_deform_attn_forward_grid / _backward_grid)
            hw = getattr(value_spatial_shapes, '_ubv_hw', None)
            qg = getattr(value_spatial_shapes, '_ubv_qgrid', None)
            ctx.grid = None
            if qg is not None and hw is not None and len(hw) == 1 and _K1_PLAN[0] and \
                    lib().ubv_ms_deform_attn_grid_supported(H, Dh, L, P, _dt(value), int(hw[0][0]), int(hw[0][1]), Nq,
                                                            int(qg[0]), int(qg[1])) and hw[0][0] * hw[0][1] == S:
                ctx.grid = (int(hw[0][0]), int(hw[0][1]), int(qg[0]), int(qg[1]))
                with _timed('k1_fwd_grid'):
                    check(lib().ubv_ms_deform_attn_forward_grid(_p(value), _p(ss), _p(ls), _p(loc), _p(aw), _p(out), B, S, H,
                                                                Dh, L, Nq, P, _dt(value), *ctx.grid, _stream()),
                          'ms_deform_attn_forward_grid')
            else:
                with _timed('k1_fwd'):
                    check(lib().ubv_ms_deform_attn_forward(_p(value), _p(ss), _p(ls), _p(loc), _p(aw),
                                                           _p(out), B, S, H, Dh, L, Nq, P, _dt(value),
                                                           int(im2col_step), _stream()),
                          'ms_deform_attn_forward')
            ctx.save_for_backward(value, ss, ls, loc, aw)
            # host copy of the shapes when the producer attached one (deform_attn.shapes_tensor): lets the backward
            # plan owner tiles without reading the device tensor back
            ctx.hw = getattr(value_spatial_shapes, '_ubv_hw', None)
            ctx.im2col_step = int(im2col_step)
            ctx.in_dtypes = (sampling_locations.dtype, attention_weights.dtype)
            return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        with _need_cuda(grad_output):
            value, ss, ls, loc, aw = ctx.saved_tensors
            B, S, H, Dh = value.shape
            _, Nq, _, L, P, _ = loc.shape
            go = grad_output.to(value.dtype).contiguous()
            gloc = torch.empty_like(loc)
            gaw = torch.empty_like(aw)
            if ctx.grid is not None:                   # the TILE plan: one block computes d(loc), d(weight) and the records
                fh, fw, qh, qw = ctx.grid
                nws = int(lib().ubv_ms_deform_attn_backward_grid_workspace(B, fh, fw, H, Dh, Nq, P, _dt(value), qh, qw))
                gv = torch.empty(value.shape, dtype=torch.float32, device=value.device)
                ws = _workspace(nws, value.device)
                with _timed('k1_bwd_grid'):
                    rc = lib().ubv_ms_deform_attn_backward_grid(
                        _p(value), _p(loc), _p(aw), _p(go), _p(gv), _p(gloc), _p(gaw), B, S, H, Dh, L, Nq, P, _dt(value),
                        fh, fw, qh, qw, _p(ws), nws, _stream())
                # UBV_ERR_UNSUPPORTED (-3): the plan the forward chose is not available to the backward (include/unibev_hip.h:
                # "the caller falls back to _backward_planned / _backward") — nothing was launched: take the paths below
                if rc != -3:
                    check(rc, 'ms_deform_attn_backward_grid')
                    return (gv.to(value.dtype), None, None, gloc.to(ctx.in_dtypes[0]), gaw.to(ctx.in_dtypes[1]), None)
            # one level with host-known shape: grad_value on the GRID owner-tile plan — sampling points binned by
            # owner tile, every pixel stored once, no f32 atomics (ubv_ms_deform_attn_backward_planned)
            if L == 1 and ctx.hw is not None and len(ctx.hw) == 1 and _K1_PLAN[0]:
                fh, fw = ctx.hw[0]
                nws = int(lib().ubv_ms_deform_attn_backward_workspace(B, fh, fw, H, Dh, Nq, P, _dt(value)))
                if nws > 0 and fh * fw == S:
                    gv = torch.empty(value.shape, dtype=torch.float32, device=value.device)
                    ws = _workspace(nws, value.device)
                    with _timed('k1_bwd_planned'):
                        check(lib().ubv_ms_deform_attn_backward_planned(
                            _p(value), _p(ss), _p(ls), _p(loc), _p(aw), _p(go), _p(gv), _p(gloc), _p(gaw), B, S, H, Dh,
                            L, Nq, P, _dt(value), int(fh), int(fw), _p(ws), nws, _stream()),
                            'ms_deform_attn_backward_planned')
                    return (gv.to(value.dtype), None, None, gloc.to(ctx.in_dtypes[0]),
                            gaw.to(ctx.in_dtypes[1]), None)
            gv = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
            with _timed('k1_bwd'):
                check(lib().ubv_ms_deform_attn_backward(_p(value), _p(ss), _p(ls), _p(loc), _p(aw),
                                                        _p(go), _p(gv), _p(gloc), _p(gaw), B, S, H, Dh,
                                                        L, Nq, P, _dt(value), ctx.im2col_step,
                                                        _stream()),
                      'ms_deform_attn_backward')
            return (gv.to(value.dtype), None, None, gloc.to(ctx.in_dtypes[0]),
                    gaw.to(ctx.in_dtypes[1]), None)


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                   im2col_step=64):
    return MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
                                                  sampling_locations, attention_weights,
                                                  im2col_step)


# ----------------------------------------------------------------------------------------------- lift
def bev_lift_supported(num_heads, head_dim, num_points, dtype):
    return bool(lib().ubv_bev_lift_supported(num_heads, head_dim, num_points, _DT.get(dtype, -1)))


# Overflow-list occupancy of the GRID lifting backward (diagnostics: bench.py's second operating point).  When enabled,
# every grid-plan backward reads its overflow counter back (one blocking 4-byte copy per call) and the largest count per
# instance is kept: {(fh, fw, P): (records that did not fit their owner tile's bucket, sampling points of the call)}.
_OVF_PROBE = None


def lift_overflow_probe(enable=None):
    global _OVF_PROBE
    if enable is None:
        return dict(_OVF_PROBE or {})
    _OVF_PROBE = {} if enable else None


_REF_CACHE = {}


def _ref_contiguous(ref):
    """f32 contiguous copy of a reference-point tensor.  The encoders hand every layer the same cached (permuted) view
    of their reference grid; copying it per call is a framework kernel inside the two-stream window, so the copy of a
    given tensor (storage, layout, version) is made once."""
    if ref.dtype == torch.float32 and ref.is_contiguous():
        return ref
    if ref.requires_grad:
        return ref.float().contiguous()
    key = (ref.data_ptr(), tuple(ref.shape), tuple(ref.stride()), ref.dtype, ref._version, ref.device.index)
    hit = _REF_CACHE.get(key)
    if hit is None:
        if len(_REF_CACHE) >= 32:
            _REF_CACHE.clear()
        hit = _REF_CACHE[key] = (ref, ref.float().contiguous())        # (the source is kept alive: its address is the key)
    return hit[1]


class _BevLift(Function):
    @staticmethod
    def forward(ctx, value, offlog, ref, vis0, count, center, lists, geom):
        B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh, grid = geom
        with _need_cuda(value, offlog, ref, vis0, count, center, lists):
            value = value.contiguous()
            # the kernels read offsets / logits as f32 or in the value's own 16-bit type (autocast)
            lowp = value.dtype != torch.float32 and offlog.dtype == value.dtype
            ol = (offlog if lowp else offlog.float()).contiguous()
            ref = _ref_contiguous(ref)
            row = H * P * 3
            assert ol.shape[-1] == row and ol.numel() == B * Nq * row
            assert value.numel() == B * Nc * fh * fw * H * Dh
            assert ref.numel() == Nc * B * Nq * Z * 2
            out = torch.empty(B, Nq, H * Dh, dtype=value.dtype, device=value.device)
            base = ol.data_ptr()
            nws = lib().ubv_bev_lift_forward_workspace(B, Nc, fh, fw, H, Dh, P, _dt(value))
            ws = _workspace(nws, value.device) if nws > 0 else None
            with _timed('lift_fwd', (geom, value.element_size())):
                check(lib().ubv_bev_lift_forward(
                    _p(value), ctypes.c_void_p(base), row,
                    ctypes.c_void_p(base + H * P * 2 * ol.element_size()), row, _dt(ol),
                    _p(ref), _p(vis0), _p(count), _p(out), B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh,
                    _dt(value), _p(ws), int(nws), _stream()), 'bev_lift_forward')
            if center is not None:
                center = center.detach().float().contiguous()
                assert center.numel() == H * P * 2
            ctx.save_for_backward(value, ol, ref, vis0, count, center, lists)
            ctx.geom = geom
            ctx.ol_dtype = offlog.dtype
            return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        with _need_cuda(grad_output):
            value, ol, ref, vis0, count, center, lists = ctx.saved_tensors
            B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh, grid = ctx.geom
            row = H * P * 3
            go = grad_output.to(value.dtype).contiguous()
            # f32 accumulation map; with 16-bit data the kernels round it once into gv_lp themselves
            gv = torch.empty(value.shape, dtype=torch.float32, device=value.device)
            gv_lp = torch.empty_like(value) if value.dtype != torch.float32 else None
            gol = torch.empty_like(ol)
            base, gbase = ol.data_ptr(), gol.data_ptr()
            off2 = H * P * 2 * ol.element_size()
            nws = lib().ubv_bev_lift_backward_workspace(B, Nc, fh, fw, H, Dh, Nq, P, qw, qh, int(grid))
            ws = _workspace(nws, value.device) if nws > 0 else None
            with _timed('lift_bwd', (ctx.geom, value.element_size())):
                check(lib().ubv_bev_lift_backward(
                    _p(value), ctypes.c_void_p(base), row, ctypes.c_void_p(base + off2), row, _dt(ol),
                    _p(ref), _p(vis0), _p(count), _p(center), _p(go), _p(gv), _p(gv_lp),
                    ctypes.c_void_p(gbase), row, ctypes.c_void_p(gbase + off2), row,
                    B, Nc, fh, fw, H, Dh, Nq, P, Z, qw, qh,
                    int(grid), _dt(value), _p(lists), _p(ws), int(nws), _stream()),
                      'bev_lift_backward')
            if _OVF_PROBE is not None and grid and ws is not None:
                # GRID workspace: [B * H * tiles counters][overflow counter] ... (csrc/bev_lift.hip grid_ws)
                tiles = B * H * ((fw + 7) // 8) * ((fh + 7) // 8)
                n = int(ws[4 * tiles:4 * tiles + 4].view(torch.int32).item())
                key = (fh, fw, P)
                _OVF_PROBE[key] = (max(n, _OVF_PROBE.get(key, (0, 0))[0]), B * Nq * H * P)
            gvalue = gv_lp if gv_lp is not None else gv
            return gvalue, gol.to(ctx.ol_dtype), None, None, None, None, None, None


@torch.no_grad()
def compact_visible(vis0, grid_w=0):
    """Per-camera ordered lists of the visible queries (``ubv_compact_visible_grid``): int32
    [Nc*Nq + Nc], list of camera c at [c*Nq, c*Nq + n_c), the counts n_c at the end.  They depend on
    the visibility alone: one compaction per forward pass serves every layer's backward.
    ``grid_w``: width of the BEV query grid — the lists then come tile by tile (8 x 8 tiles) instead of
    ascending, which keeps a batch of entries on neighbouring pixels of the camera's map."""
    with _need_cuda(vis0):
        Nc, Nq = vis0.shape
        lists = torch.empty(int(lib().ubv_visible_lists_elems(Nc, Nq)), dtype=torch.int32,
                            device=vis0.device)
        check(lib().ubv_compact_visible_grid(_p(vis0.contiguous()), Nc, Nq, int(grid_w), _p(lists), _stream()),
              'compact_visible')
        return lists


def bev_lift(value, offlog, ref, num_cams, feat_hw, num_heads, num_points, vis0=None, count=None,
             query_grid=None, ref_is_grid=False, slot_center=None, visible_lists=None):
    """Fused single-level BEV query lifting (``ubv_bev_lift_forward``).

    value  (B*num_cams, fh*fw, C)  projected features, batch-major / camera-minor
    offlog (B, Nq, H*P*3)          [sampling_offsets | attention logits] raw Linear outputs
    ref    (num_cams, B, Nq, Z, 2) reference points, flat point p uses anchor p % Z
    vis0   (num_cams, Nq) uint8    visibility of batch element 0, or None
    count  (B, Nq) float32         camera count divisor, or None
    query_grid (qh, qw)            BEV grid the Nq queries form (tiling / owner-tile backward)
    ref_is_grid                    ref is exactly that grid's cell centres (and num_cams == 1)
    slot_center (H*P*2,)           accepted for API stability; the bins plan ignores it
    visible_lists                  ``compact_visible(vis0)`` or None (compacted per backward then)
    """
    fh, fw = feat_hw
    BNc, S, C = value.shape[0], value.shape[1], value.shape[-1] if value.dim() == 3 else None
    if value.dim() == 4:
        C = value.shape[2] * value.shape[3]
    B = BNc // num_cams
    Nq = offlog.shape[-2]
    Z = ref.shape[-2]
    Dh = C // num_heads
    qw, qh = (query_grid[1], query_grid[0]) if query_grid is not None else (0, 0)
    grid = bool(ref_is_grid) and num_cams == 1 and qw > 0
    geom = (B, num_cams, fh, fw, num_heads, Dh, Nq, num_points, Z, qw, qh, grid)
    if _STUDY_ROUND[0] is not None and not torch.is_grad_enabled():
        # precision study (tools/ab/value16_study.py, inference only): WHICH rounding of the value-only 16-bit mode costs
        # the accuracy — the stored value map ('value': rounded through the type, kept f32), the sampled output ('out')
        what, dt = _STUDY_ROUND[0]
        if 'value' in what:
            value = value.to(dt).to(value.dtype)
        out = _BevLift.apply(value, offlog, ref, vis0, count, slot_center, visible_lists, geom)
        return out.to(dt).to(out.dtype) if 'out' in what else out
    return _BevLift.apply(value, offlog, ref, vis0, count, slot_center, visible_lists, geom)


_STUDY_ROUND = [None]


def set_study_rounding(what=None, dtype=torch.float16):
    """``what``: None | 'value' | 'out' | 'value+out' (see bev_lift).  Returns the previous setting."""
    prev, _STUDY_ROUND[0] = _STUDY_ROUND[0], (None if what is None else (what, dtype))
    return prev


# ----------------------------------------------------------------------------------------------- geometry
@torch.no_grad()
def point_sampling(lidar2img, xs, ys, zs, pc_range, img_hw):
    """Camera projection + visibility (``ubv_point_sampling``).

    lidar2img (B, Nc, 4, 4) float32 cuda; xs (W,), ys (H,), zs (D,) float32 cuda.
    Returns reference_points_cam (Nc,B,Nq,D,2) f32, bev_mask (Nc,B,Nq,D) bool, vis0 (Nc,Nq) uint8,
    count (B,Nq) f32.
    """
    with _need_cuda(lidar2img, xs, ys, zs):
        B, Nc = lidar2img.shape[:2]
        W, H, D = xs.numel(), ys.numel(), zs.numel()
        Nq = H * W
        dev = lidar2img.device
        l2i = lidar2img.float().contiguous()
        ref_cam = torch.empty(Nc, B, Nq, D, 2, dtype=torch.float32, device=dev)
        mask = torch.empty(Nc, B, Nq, D, dtype=torch.uint8, device=dev)
        vis0 = torch.empty(Nc, Nq, dtype=torch.uint8, device=dev)
        count = torch.empty(B, Nq, dtype=torch.float32, device=dev)
        check(lib().ubv_point_sampling(_p(l2i), _p(xs), _p(ys), _p(zs), _lib.float_array(pc_range),
                                       float(img_hw[0]), float(img_hw[1]), _p(ref_cam), _p(mask),
                                       _p(vis0), _p(count), B, Nc, H, W, D, _stream()),
              'point_sampling')
        return ref_cam, mask.view(torch.bool), vis0, count


# ----------------------------------------------------------------------------------------------- flatten
class _FlattenEmbed(Function):
    @staticmethod
    def forward(ctx, feat, embA, embB, row=None):
        with _need_cuda(feat, embA, embB):
            N, C, HW = feat.shape
            feat = feat.contiguous()
            a = None if embA is None else embA.float().contiguous()
            # ``row``: embB is a (L, C) table and row ``row`` of it is added (the level embedding): the backward then
            # returns a table-shaped gradient written by this library — ``table[row]`` outside would come back through
            # the framework's select_backward (a fill + a copy inside the encoders' two-stream window)
            ctx.row, ctx.tab_rows = row, (None if row is None else embB.shape[0])
            b = None if embB is None else (embB if row is None else embB[row]).float().contiguous()
            out = torch.empty(N, HW, C, dtype=feat.dtype, device=feat.device)
            groups = 1 if a is None else a.shape[0]
            check(lib().ubv_flatten_embed_forward(_p(feat), _p(a), groups, _p(b), _p(out), N, C, HW,
                                                  _dt(feat), _stream()), 'flatten_embed_forward')
            ctx.groups = groups
            ctx.has = (embA is not None, embB is not None)
            ctx.emb_dtypes = (None if embA is None else embA.dtype, None if embB is None else embB.dtype)
            return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        with _need_cuda(grad_out):
            N, HW, C = grad_out.shape
            go = grad_out.contiguous()
            gin = torch.empty(N, C, HW, dtype=go.dtype, device=go.device)
            need_emb = (ctx.has[0] and ctx.needs_input_grad[1]) or (ctx.has[1] and ctx.needs_input_grad[2])
            # (accumulator from the step's zero arena, sums by this library's column-sum kernel: torch.zeros / Tensor.sum
            #  are framework kernels, and this backward runs inside the encoders' two-stream window)
            gemb = zeros_f32(N * C, go.device).view(N, C) if need_emb else None
            check(lib().ubv_flatten_embed_backward(_p(go), _p(gin), _p(gemb), N, C, HW, _dt(go),
                                                   _stream()), 'flatten_embed_backward')
            ga = gb = None
            if ctx.has[0] and ctx.needs_input_grad[1]:
                # sum over the N / groups repeats of each group row: column sums of the [N / groups, groups * C] view
                # (the slice-sum form: N / groups slices of [groups, C])
                ga = linear_grad_reduce(None, gemb.view(N // ctx.groups, ctx.groups, C))[1]
                ga = ga if ga.dtype == ctx.emb_dtypes[0] else ga.to(ctx.emb_dtypes[0])
            if ctx.has[1] and ctx.needs_input_grad[2]:
                if ctx.row is None:
                    gb = linear_grad_reduce(None, gemb.view(N, 1, C))[1].view(C)
                else:
                    gb = zeros_f32(ctx.tab_rows * C, go.device).view(ctx.tab_rows, C)
                    linear_grad_reduce(None, gemb.view(N, 1, C), out=gb[ctx.row])
                gb = gb if gb.dtype == ctx.emb_dtypes[1] else gb.to(ctx.emb_dtypes[1])
            return gin, ga, gb, None


def flatten_embed(feat, embA=None, embB=None, row=None):
    """(N, C, HW) -> (N, HW, C) + embA[n % groups] + embB (``ubv_flatten_embed_forward``).  ``row``: ``embB`` is an
    (L, C) table and its row ``row`` is the term."""
    return _FlattenEmbed.apply(feat, embA, embB, row)


# ----------------------------------------------------------------------------------------------- fusion
class _BevFuse(Function):
    @staticmethod
    def forward(ctx, img, pts, cw_img, cw_pts, sw_img, sw_pts, cat):
        ref = img if img is not None else pts
        with _need_cuda(ref, cw_img, cw_pts):
            B, Nq, C = ref.shape
            img_c = None if img is None else img.contiguous()
            pts_c = None if pts is None else pts.to(ref.dtype).contiguous()
            cwi, cwp = cw_img.float().contiguous(), cw_pts.float().contiguous()
            swi = None if sw_img is None else sw_img.float().contiguous()
            swp = None if sw_pts is None else sw_pts.float().contiguous()
            out = torch.empty(Nq, B, C * (2 if cat else 1), dtype=ref.dtype, device=ref.device)
            check(lib().ubv_bev_fuse_forward(_p(img_c), _p(pts_c), _p(cwi), _p(cwp), _p(swi), _p(swp),
                                             _p(out), B, Nq, C, int(cat), _dt(ref), _stream()),
                  'bev_fuse_forward')
            ctx.save_for_backward(img_c, pts_c, cwi, cwp, swi, swp)
            ctx.cat = int(cat)
            ctx.shape = (B, Nq, C)
            ctx.dt = (cw_img.dtype, cw_pts.dtype, None if sw_img is None else sw_img.dtype,
                      None if sw_pts is None else sw_pts.dtype)
            return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        with _need_cuda(grad_out):
            img, pts, cwi, cwp, swi, swp = ctx.saved_tensors
            B, Nq, C = ctx.shape
            ref = img if img is not None else pts
            go = grad_out.to(ref.dtype).contiguous()
            gimg = torch.empty_like(img) if (img is not None and ctx.needs_input_grad[0]) else None
            gpts = torch.empty_like(pts) if (pts is not None and ctx.needs_input_grad[1]) else None
            need_cw = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
            need_sw = (swi is not None and ctx.needs_input_grad[4]) or \
                      (swp is not None and ctx.needs_input_grad[5])
            gcw = torch.zeros(2, C, dtype=torch.float32, device=go.device) if need_cw else None
            gsw = torch.zeros(2, Nq, dtype=torch.float32, device=go.device) if need_sw else None
            check(lib().ubv_bev_fuse_backward(_p(go), _p(img), _p(pts), _p(cwi), _p(cwp), _p(swi),
                                              _p(swp), _p(gimg), _p(gpts), _p(gcw), _p(gsw), B, Nq, C,
                                              ctx.cat, _dt(ref), _stream()), 'bev_fuse_backward')
            g = [gimg, gpts, None, None, None, None, None]
            if need_cw:
                g[2], g[3] = gcw[0].to(ctx.dt[0]), gcw[1].to(ctx.dt[1])
            if need_sw:
                if swi is not None:
                    g[4] = gsw[0].to(ctx.dt[2])
                if swp is not None:
                    g[5] = gsw[1].to(ctx.dt[3])
            return tuple(g)


def bev_fuse(img, pts, cw_img, cw_pts, sw_img=None, sw_pts=None, cat=False):
    """(B,Nq,C) x2 -> (Nq,B,C*s): per-channel / per-query weighting, add or concat, and the final
    permute (``ubv_bev_fuse_forward``).  A missing modality is ``None`` (zeros)."""
    return _BevFuse.apply(img, pts, cw_img, cw_pts, sw_img, sw_pts, cat)


# ----------------------------------------------------------------------------------------------- add+norm
# Dropout seeds: ONE draw from torch's CPU generator per forward pass (``new_step``) mixed with a
# call counter — reproducible under torch.manual_seed, no device sync, and a few hundred nanoseconds
# per call instead of the two tensor ops of torch.randint(...).item() (24 calls per forward pass on
# a host-bound forward).
_SEED_STATE = [None, 0]
# Optional per-step seed base ON THE DEVICE (an int64 tensor of one element), added to every call's
# seed by the kernels: a captured HIP graph bakes the per-call seeds in, so the replays of a step
# differ only through this value, which the graph itself advances (``graph_step.GraphedStep``).
_SEED_BASE = [None]


def set_seed_base(tensor):
    """Install (or with None remove) the device-side dropout seed base."""
    if tensor is not None:
        assert tensor.is_cuda and tensor.dtype == torch.int64 and tensor.numel() == 1
    _SEED_BASE[0] = tensor


def _next_seed():
    if _SEED_STATE[0] is None:
        _SEED_STATE[0] = int(torch.randint(0, 2 ** 62, (1,)).item())
        _SEED_STATE[1] = 0
    _SEED_STATE[1] += 1
    z = (_SEED_STATE[0] + _SEED_STATE[1] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return z & 0x3FFFFFFFFFFFFFFF


class _AddDropoutNorm(Function):
    @staticmethod
    def forward(ctx, x, identity, gamma, beta, p, eps, batch=0):
        with _need_cuda(x, identity, gamma, beta):
            C = x.shape[-1]
            x2 = x.reshape(-1, C).contiguous()
            # residual stream: f32, or the branch's own 16-bit type when the caller keeps it there
            lowp = x2.dtype != torch.float32 and identity.dtype == x2.dtype
            sdt = x2.dtype if lowp else torch.float32
            id2 = identity.reshape(-1, C).to(sdt).contiguous()
            R = x2.shape[0]
            # batch > 1: x and identity are ONE sample's rows shared by `batch` samples (the first encoder layer's
            # self-attention): the kernel reads them with a row period, y / mean / rstd / the dropout mask are per sample
            period = 0
            if batch and batch > 1:
                assert x.shape[0] == 1 and identity.shape[0] == 1 and id2.shape[0] == R
                period, R = R, R * int(batch)
            g, b = gamma.float().contiguous(), beta.float().contiguous()
            y = torch.empty(R, C, dtype=sdt, device=x.device)
            mean = torch.empty(R, dtype=torch.float32, device=x.device)
            rstd = torch.empty(R, dtype=torch.float32, device=x.device)
            # seed from torch's CPU generator: reproducible under torch.manual_seed, no device sync
            seed = _next_seed() if p > 0 else 0
            check(lib().ubv_add_dropout_layernorm_forward(_p(x2), _p(id2), _p(g), _p(b), _p(y), _p(mean),
                                                          _p(rstd), R, period, C, float(eps), float(p), seed,
                                                          _p(_SEED_BASE[0]), _dt(x2), _DT[sdt],
                                                          _stream()),
                  'add_dropout_layernorm_forward')
            ctx.save_for_backward(x2, id2, g, mean, rstd)
            ctx.p, ctx.seed, ctx.shape = float(p), seed, x.shape
            ctx.seed_base = _SEED_BASE[0]
            ctx.dts = (identity.dtype, gamma.dtype, beta.dtype)
            ctx.period, ctx.rows = period, R
            return y.view(x.shape) if not period else y.view(int(batch), *x.shape[1:])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        with _need_cuda(grad_y):
            x2, id2, g, mean, rstd = ctx.saved_tensors
            R, C = ctx.rows, x2.shape[1]
            gy = grad_y.reshape(R, C).to(id2.dtype).contiguous()
            gx = torch.empty_like(x2)            # (shared rows: [period, C], summed over the samples by the kernel)
            gid = torch.empty_like(id2)
            dg, db = zeros_f32(C, x2.device), zeros_f32(C, x2.device)
            dxs = zeros_f32(C, x2.device)
            check(lib().ubv_add_dropout_layernorm_backward(_p(gy), _p(x2), _p(id2), _p(g), _p(mean),
                                                           _p(rstd), _p(gx), _p(gid), _p(dg), _p(db),
                                                           _p(dxs), R, ctx.period, C, ctx.p, ctx.seed,
                                                           _p(ctx.seed_base), _dt(x2), _dt(id2),
                                                           _p(_ordered_ws(C, x2.device)), _stream()),
                  'add_dropout_layernorm_backward')
            gx = gx.view(ctx.shape)
            # column sums of grad_x ride along: if x came straight out of a Linear, its backward takes
            # them as the bias gradient instead of reducing grad_x again (linear._Linear.backward)
            tag_grad(gx, '_ubv_colsum', dxs)
            gid = gid.view(ctx.shape).to(ctx.dts[0])
            tag_grad(gid, '_ubv_owned', True)   # fresh, single consumer: linear._Linear may accumulate into it
            return (gx, gid, dg.to(ctx.dts[1]), db.to(ctx.dts[2]), None, None, None)


_NORM_ORDERED = os.environ.get('UBV_NORM_ORDERED', '1') != '0'      # 0: f32 atomics instead of ordered sums (A/B runs)


def _ordered_ws(C, device):
    """Scratch of the ordered column sums of ``ubv_add_dropout_layernorm_backward`` (the stream's shared scratch)."""
    if not _NORM_ORDERED:
        return None
    return _workspace(int(lib().ubv_add_dropout_layernorm_backward_workspace(int(C))), device)


def add_dropout_layernorm(x, identity, gamma, beta, p=0.0, training=False, eps=1e-5, batch=0):
    """LayerNorm(identity + dropout(x)) in one pass each way (``ubv_add_dropout_layernorm_*``).  ``batch`` > 1: x and
    identity are (1, ...) — one sample's rows shared by ``batch`` samples — and the result is (batch, ...) with its own
    dropout mask per sample; their gradients are summed over the samples."""
    return _AddDropoutNorm.apply(x, identity, gamma, beta, float(p) if training else 0.0, eps, int(batch or 0))


# ----------------------------------------------------------------------------------------------- zero arena
class _ZeroArena:
    """Small f32 accumulators (bias / gamma / beta gradients) carved out of ONE zero-filled buffer
    per backward pass instead of one ``torch.zeros`` fill kernel each (134 fills, 0.5 ms per step
    measured).  ``reset()`` at the start of every forward drops the buffer; the first ``take`` of
    the following backward allocates a fresh one, so gradients of a finished step are never
    overwritten."""

    def __init__(self):
        self.buf, self.off, self.cap = None, 0, 1 << 16
        self.used = False

    def reset(self):
        if self.buf is not None:
            self.cap = max(self.cap, 2 * self.off)
        self.buf, self.off = None, 0

    def take(self, n, device):
        self.used = True
        n_al = (n + 63) // 64 * 64                     # 256-byte aligned slices
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.cap = max(self.cap, 4 * n_al)
            self.buf = torch.zeros(self.cap, dtype=torch.float32, device=device)
            self.off = 0
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        return out


_ARENAS = {}            # one arena per (device, stream): the fill and its users stay in stream order


def zeros_f32(n, device):
    """Zeroed f32 vector of ``n`` elements from the per-step arena of the current stream."""
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index)) if device.type == 'cuda' else None
    arena = _ARENAS.get(key)
    if arena is None:
        arena = _ARENAS[key] = _ZeroArena()
    return arena.take(int(n), device)


def new_step():
    """Called once per forward pass (``linear.lowp_step_cache``): later backward accumulators come
    from a fresh zero buffer — allocated (and zero-filled: the one framework kernel involved) HERE, ahead of the pass and
    of its two-stream fork, for every arena that was used before, on the caller's stream (the fork orders it before
    anything the side stream does; the buffer lives until the next ``new_step``, after the join)."""
    for key, arena in _ARENAS.items():
        arena.reset()
        if key is not None and arena.used:
            dev = torch.device('cuda', key[0])
            arena.buf = torch.zeros(arena.cap, dtype=torch.float32, device=dev)
            arena.off = 0
            if torch._C._cuda_getCurrentRawStream(key[0]) != key[1] and not torch.cuda.is_current_stream_capturing():
                arena.buf.record_stream(torch.cuda.ExternalStream(key[1], device=dev))    # (used on the arena's stream)
    _SEED_STATE[0] = None             # dropout seeds of this pass: one fresh draw from torch's generator


# ----------------------------------------------------------------------------------------------- relu + dropout
class _ReluDropout(Function):
    @staticmethod
    def forward(ctx, x, p):
        with _need_cuda(x):
            xc = x.contiguous()
            y = torch.empty_like(xc)
            seed = _next_seed() if p > 0 else 0
            check(lib().ubv_relu_dropout_forward(_p(xc), _p(y), xc.numel(), float(p), seed,
                                                 _p(_SEED_BASE[0]), _dt(xc), _stream()),
                  'relu_dropout_forward')
            ctx.save_for_backward(y)
            ctx.p = float(p)
            return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        with _need_cuda(grad_y):
            y, = ctx.saved_tensors
            gy = grad_y.to(y.dtype).contiguous()
            gx = torch.empty_like(y)
            check(lib().ubv_relu_dropout_backward(_p(gy), _p(y), _p(gx), y.numel(), ctx.p, _dt(y),
                                                  _stream()), 'relu_dropout_backward')
            return gx, None


def relu_dropout(x, p=0.0, training=False):
    """``dropout(relu(x), p)`` in one pass each way (``ubv_relu_dropout_*``); the output is the
    only tensor kept for backward."""
    return _ReluDropout.apply(x, float(p) if training else 0.0)


# ----------------------------------------------------------------------------------------------- linear forward
_LINEAR_PLAN_OK = [True]


@torch.no_grad()
def linear_forward(x, weight, bias=None):
    """``F.linear(x, weight, bias)`` through ``ubv_linear_forward`` (hipBLASLt with a cached plan),
    or None when hipBLASLt has no algorithm for the shape (the caller then uses the framework GEMM).
    x (..., K) contiguous, weight (N, K), bias (N,) or None, one dtype."""
    if not _LINEAR_PLAN_OK[0]:
        return None
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    with _need_cuda(x, weight, bias):
        y = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
        nws = int(lib().ubv_linear_workspace())
        ws = _workspace(nws, x.device)
        rc = lib().ubv_linear_forward(_p(x), _p(weight), _p(bias), _p(y), M, N, K, _dt(x), _p(ws),
                                      nws, _stream())
    if rc == -3:                      # UBV_ERR_UNSUPPORTED: hipBLASLt has nothing for this shape
        _LINEAR_PLAN_OK[0] = False
        return None
    check(rc, 'linear_forward')
    return y


# ----------------------------------------------------------------------------------------------- MFMA GEMM
@torch.no_grad()
def split_weight(w, transposed=True):
    """f32 (N, K) -> bf16 halves (w_hi, w_lo, wt_hi, wt_lo) for ``gemm_nt`` (``ubv_split_weight``);
    the transposed pair (K, N) serves the input-gradient GEMM."""
    with _need_cuda(w):
        w = w.detach().float().contiguous()
        N, K = w.shape
        wh = torch.empty(N, K, dtype=torch.bfloat16, device=w.device)
        wl = torch.empty_like(wh)
        wth = torch.empty(K, N, dtype=torch.bfloat16, device=w.device) if transposed else None
        wtl = torch.empty_like(wth) if transposed else None
        check(lib().ubv_split_weight(_p(w), N, K, _p(wh), _p(wl), _p(wth), _p(wtl), _stream()), 'split_weight')
        return wh, wl, wth, wtl


@torch.no_grad()
def split_weights_batched(groups):
    """``split_weight`` for many weights in ONE launch (``ubv_split_weights_batched``).  ``groups``: list of lists of
    f32 CUDA parameters [N_i, K] that are used concatenated along dim 0.  Returns one (w_hi, w_lo, wt_hi, wt_lo)
    per group, carved from a single fresh bf16 buffer."""
    if not groups:
        return []
    dev = groups[0][0].device
    with _need_cuda(*[p for g in groups for p in g]):
        total = sum(4 * sum(p.shape[0] for p in g) * g[0].shape[1] for g in groups)
        buf = torch.empty(total, dtype=torch.bfloat16, device=dev)
        outs, ent, o = [], [], 0
        for g in groups:
            K = g[0].shape[1]
            N = sum(p.shape[0] for p in g)
            wh, wl = buf[o:o + N * K].view(N, K), buf[o + N * K:o + 2 * N * K].view(N, K)
            wth, wtl = buf[o + 2 * N * K:o + 3 * N * K].view(K, N), buf[o + 3 * N * K:o + 4 * N * K].view(K, N)
            o += 4 * N * K
            outs.append((wh, wl, wth, wtl))
            r = 0
            for p in g:
                w = p.detach()
                if w.dtype != torch.float32 or not w.is_contiguous() or w.shape[1] != K:
                    raise ValueError('split_weights_batched: contiguous f32 [N, K] weights of one K per group')
                ent.append((w.data_ptr(), p.shape[0], K, wh[r:].data_ptr(), wl[r:].data_ptr(),
                            wth[:, r:].data_ptr(), wtl[:, r:].data_ptr(), N))
                r += p.shape[0]
        n = len(ent)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        col = lambda i: [e[i] for e in ent]
        check(lib().ubv_split_weights_batched(n, PA(*col(0)), IA(*col(1)), IA(*col(2)), PA(*col(3)), PA(*col(4)),
                                              PA(*col(5)), PA(*col(6)), IA(*col(7)), _stream()), 'split_weights_batched')
        return outs


@torch.no_grad()
def gemm_nt(x, w_hi, w_lo=None, bias=None, residual=None, out=None, row_bias=None):
    """y = x @ w^T (+ bias) (+ residual) on the matrix cores (``ubv_gemm_nt``).  f32 ``x`` takes the
    split weight (w_hi, w_lo); 16-bit ``x`` takes w_hi of its own type.  Returns None when the shape
    is outside the kernel's reach (K % 32, N % 32).  ``row_bias`` [R, N] (a view with a 16-byte aligned row stride is
    fine): + row_bias[m % R] on row m (``ubv_gemm_nt_rowbias``), instead of ``residual``."""
    with _need_cuda(x, w_hi, w_lo, bias, residual, out, row_bias):
        K = x.shape[-1]
        M = x.numel() // K
        N = w_hi.shape[0]
        if K % 32 != 0 or N % 32 != 0 or not x.is_contiguous():
            return None
        y = out if out is not None else torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
        b = None if bias is None else bias.float().contiguous()
        if row_bias is not None:
            assert residual is None and row_bias.dim() == 2 and row_bias.shape[1] == N and row_bias.dtype == x.dtype
            R, ld = row_bias.shape[0], row_bias.stride(0)
            if row_bias.stride(1) != 1 or M % R != 0 or ld % 4 != 0 or row_bias.data_ptr() % 16 != 0:
                return None
            rc = lib().ubv_gemm_nt_rowbias(_p(x), K, _p(w_hi), _p(w_lo), K, _p(b), _p(row_bias), R, ld, _p(y), N, M, N,
                                           K, _dt(x), _stream())
            if rc == -3:
                return None
            check(rc, 'gemm_nt_rowbias')
            return y
        rc = lib().ubv_gemm_nt(_p(x), K, _p(w_hi), _p(w_lo), K, _p(b), _p(residual), _p(y), N, M, N, K,
                               _dt(x), _stream())
        if rc == -3:
            return None
        check(rc, 'gemm_nt')
        return y


def gemm_nt_dual(x, w_hi, w_lo, bias=None, x2=None, residual=None, out=None, y2_cols=0, row_bias=None):
    """``ubv_gemm_nt_dual`` (f32): y = [x | x2] @ w^T (+ bias) (+ residual), or — ``y2_cols`` > 0 — the last ``y2_cols``
    output columns returned as a second tensor with ``row_bias[m % R]`` added to them: (y, y2).  None when the shape is
    outside the kernel's reach."""
    with _need_cuda(x, w_hi, w_lo, bias, x2, residual, out, row_bias):
        K1 = x.shape[-1]
        M = x.numel() // K1
        K = K1 + (x2.shape[-1] if x2 is not None else 0)
        N = w_hi.shape[0]
        ok = x.dtype == torch.float32 and x.is_contiguous() and (x2 is None or (x2.is_contiguous() and x2.dtype == x.dtype)) \
            and K % 32 == 0 and K1 % 32 == 0 and N % 32 == 0 and w_hi.shape[1] == K
        n1 = N - y2_cols
        if not ok or (y2_cols and (n1 <= 0 or n1 % 128 != 0 or y2_cols % 4 != 0)):
            return None
        b = None if bias is None else bias.float().contiguous()
        y = out if out is not None else torch.empty(*x.shape[:-1], n1, dtype=x.dtype, device=x.device)
        y2 = torch.empty(*x.shape[:-1], y2_cols, dtype=x.dtype, device=x.device) if y2_cols else None
        R = ld = 0
        if row_bias is not None:
            assert y2 is not None and row_bias.shape[1] == y2_cols and row_bias.dtype == x.dtype
            R, ld = row_bias.shape[0], row_bias.stride(0)
            if row_bias.stride(1) != 1 or M % R != 0 or ld % 4 != 0 or row_bias.data_ptr() % 16 != 0:
                return None
        rc = lib().ubv_gemm_nt_dual(_p(x), K1, _p(x2), 0 if x2 is None else x2.shape[-1], K1 if x2 is not None else 0,
                                    _p(w_hi), _p(w_lo), K, _p(b), _p(residual), _p(row_bias), R, ld, _p(y), n1,
                                    _p(y2), y2_cols, n1 if y2_cols else 0, M, N, K, _stream())
        if rc == -3:
            return None
        check(rc, 'gemm_nt_dual')
        return (y, y2) if y2_cols else y


@torch.no_grad()
def gemm_wgrad_dual(grad_out, grad_out2, x):
    """``gemm_wgrad`` with grad_out given as two matrices [M, N1] | [M, N2] (N1 a multiple of 128): (grad_weight
    [N1 + N2, K], grad_bias [N1 + N2]) in one pass over ``x`` (``ubv_gemm_wgrad_dual``, f32).  None when out of reach."""
    with _need_cuda(grad_out, grad_out2, x):
        M, N1 = grad_out.shape
        N2 = grad_out2.shape[1]
        K = x.shape[1]
        N = N1 + N2
        if N1 % 128 != 0 or N2 % 4 != 0 or K % 4 != 0 or M == 0 or grad_out2.shape[0] != M or \
                not (grad_out.dtype == grad_out2.dtype == x.dtype == torch.float32) or \
                not (grad_out.is_contiguous() and grad_out2.is_contiguous() and x.is_contiguous()):
            return None
        S = int(lib().ubv_gemm_wgrad_splits(M, N, K))
        part = _workspace(4 * S * (N * K + N), x.device)
        out = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
        rc = lib().ubv_gemm_wgrad_dual(_p(grad_out), _p(grad_out2), N1, _p(x), _p(part), _p(out), M, N, K, S, _stream())
        if rc == -3:
            return None
        check(rc, 'gemm_wgrad_dual')
        return out[:N * K].view(N, K), out[N * K:]


def gemm_nt_act(x, w_hi, w_lo=None, bias=None, act=1, mask=None, p=0.0, seed=0):
    """``ubv_gemm_nt_act``: act 1 -> dropout(relu(x @ w^T + bias), p) with the keep mask of
    ``relu_dropout`` for ``seed``; act 2 -> (x @ w^T) / (1 - p) where ``mask`` != 0, else 0.  None when the
    shape is outside the kernel's reach."""
    with _need_cuda(x, w_hi, w_lo, bias, mask):
        K = x.shape[-1]
        M = x.numel() // K
        N = w_hi.shape[0]
        if K % 32 != 0 or N % 32 != 0 or not x.is_contiguous() or \
                (mask is not None and not (mask.is_contiguous() and mask.dtype == x.dtype and mask.numel() == M * N)):
            return None
        y = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
        b = None if bias is None else bias.float().contiguous()
        rc = lib().ubv_gemm_nt_act(_p(x), K, _p(w_hi), _p(w_lo), K, _p(b), _p(y), N, M, N, K, _dt(x), int(act),
                                   _p(mask), float(p), int(seed), _p(_SEED_BASE[0]), _stream())
        if rc == -3:
            return None
        check(rc, 'gemm_nt_act')
        return y


@torch.no_grad()
def relu_dropout_raw(x, p, seed):
    """dropout(relu(x), p) without an autograd node (``ubv_relu_dropout_forward``)."""
    with _need_cuda(x):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        check(lib().ubv_relu_dropout_forward(_p(xc), _p(y), xc.numel(), float(p), int(seed), _p(_SEED_BASE[0]),
                                             _dt(xc), _stream()), 'relu_dropout_forward')
        return y


@torch.no_grad()
def relu_dropout_grad_raw(grad_y, y, p):
    """grad of ``relu_dropout`` from its output ``y`` (``ubv_relu_dropout_backward``)."""
    with _need_cuda(grad_y, y):
        gy = grad_y.to(y.dtype).contiguous()
        gx = torch.empty_like(y)
        check(lib().ubv_relu_dropout_backward(_p(gy), _p(y), _p(gx), y.numel(), float(p), _dt(y), _stream()),
              'relu_dropout_backward')
        return gx


def next_dropout_seed():
    return _next_seed()


@torch.no_grad()
def gemm_wgrad(grad_out, x):
    """(grad_weight f32 [N, K], grad_bias f32 [N]) of a Linear from grad_out [M, N] and x [M, K] (one
    dtype) on the matrix cores (``ubv_gemm_wgrad``: split-K slabs + their sum).  None when the
    shape is outside the kernel's reach."""
    with _need_cuda(grad_out, x):
        M, N = grad_out.shape
        K = x.shape[1]
        cw = 4 if x.dtype == torch.float32 else 8              # columns per 16-byte load
        if N % cw != 0 or K % cw != 0 or grad_out.dtype != x.dtype or M == 0 or \
                not (grad_out.is_contiguous() and x.is_contiguous()):
            return None
        S = int(lib().ubv_gemm_wgrad_splits(M, N, K))
        part = _workspace(4 * S * (N * K + N), x.device)      # scratch: consumed inside the call
        out = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
        rc = lib().ubv_gemm_wgrad(_p(grad_out), _p(x), _p(part), _p(out), M, N, K, S, _dt(x), _stream())
        if rc == -3:
            return None
        check(rc, 'gemm_wgrad')
        return out[:N * K].view(N, K), out[N * K:]


# ----------------------------------------------------------------------------------------------- linear grads
@torch.no_grad()
def linear_grad_reduce(grad_out=None, partials=None, out=None):
    """(grad_bias f32 [N] or None, grad_weight f32 [N, K] or None) in one launch
    (``ubv_linear_grad_reduce``): column sums of ``grad_out`` [rows, N] and the sum over the
    split-K slices ``partials`` [S, N, K].  Both inputs must share one dtype."""
    gb = gw = None
    go_p = part_p = None
    rows = N = S = NK = 0
    ref = grad_out if grad_out is not None else partials
    with _need_cuda(ref):
        if grad_out is not None:
            grad_out = grad_out.contiguous()
            rows, N = grad_out.shape
            gb = zeros_f32(N, ref.device)
            go_p = _p(grad_out)
        if partials is not None:
            assert grad_out is None or partials.dtype == grad_out.dtype
            partials = partials.contiguous()
            S, NK = partials.shape[0], partials[0].numel()
            # (``out``: where the slice sum goes — e.g. one row of an arena-zeroed table)
            gw = out if out is not None else torch.empty(partials.shape[1:], dtype=torch.float32, device=ref.device)
            assert gw.is_contiguous() and gw.numel() == NK and gw.dtype == torch.float32
            part_p = _p(partials)
        check(lib().ubv_linear_grad_reduce(go_p, rows, N, _p(gb), part_p, S, NK, _p(gw), _dt(ref),
                                           _stream()), 'linear_grad_reduce')
        return gb, gw


@torch.no_grad()
def add2(a, b, out=None):
    """a + b for two f32 CUDA tensors of one shape on this library's kernel (``ubv_add2_f32``); ``out`` may be ``a``."""
    with _need_cuda(a, b, out):
        ac = a if a.is_contiguous() else a.contiguous()
        bc = b if b.is_contiguous() else b.contiguous()
        assert ac.dtype == bc.dtype == torch.float32 and ac.shape == bc.shape
        y = out if out is not None else torch.empty_like(ac)
        check(lib().ubv_add2_f32(_p(ac), _p(bc), _p(y), ac.numel(), _stream()), 'add2')
        return y


@torch.no_grad()
def slice_sum(slices, out):
    """out[r, c] = sum_s slices[s, r, c] (``ubv_slice_sum_f32``): ``slices`` f32 contiguous [S, rows, cols]; ``out`` a
    [rows, cols] view with unit column stride (a column slice of a wider matrix is fine)."""
    with _need_cuda(slices, out):
        S, rows, cols = slices.shape
        assert slices.is_contiguous() and slices.dtype == out.dtype == torch.float32
        assert out.shape == (rows, cols) and out.stride(1) == 1
        check(lib().ubv_slice_sum_f32(_p(slices), S, rows, cols, _p(out), out.stride(0), _stream()), 'slice_sum')
        return out


def _sum2(ga, gb):
    if ga is None or gb is None:
        return ga if gb is None else gb
    if ga.is_cuda and ga.dtype == gb.dtype == torch.float32 and ga.data_ptr() % 16 == 0 and gb.data_ptr() % 16 == 0 \
            and ga.is_contiguous() and gb.is_contiguous():
        return add2(ga, gb)
    return ga + gb


class _FanOut(Function):
    """Two aliases of one tensor whose gradients are summed by ``add2`` instead of by the autograd engine's framework add:
    for a parameter with two consumers inside the encoders' two-stream window.  The aliases carry ``_ubv_master`` (the
    parameter) so that the per-step weight caches of ``unibev_amd.linear`` keep finding their entries."""

    @staticmethod
    def forward(ctx, w):
        ctx.set_materialize_grads(False)
        return w.view_as(w), w.view_as(w)

    @staticmethod
    @once_differentiable
    def backward(ctx, ga, gb):
        return _sum2(ga, gb)


class _FanOutPair(Function):
    """``_FanOut`` for two tensors whose gradients usually arrive as adjacent slices of one buffer (the sampling_offsets |
    attention_weights rows of a fused weight gradient): one ``add2`` over both instead of two."""

    @staticmethod
    def forward(ctx, w1, w2):
        ctx.set_materialize_grads(False)
        return w1.view_as(w1), w1.view_as(w1), w2.view_as(w2), w2.view_as(w2)

    @staticmethod
    @once_differentiable
    def backward(ctx, ga1, gb1, ga2, gb2):
        def adjacent(x, y):
            return x is not None and y is not None and x.is_cuda and x.dtype == y.dtype == torch.float32 and \
                x.is_contiguous() and y.is_contiguous() and x.data_ptr() % 16 == 0 and \
                x.untyped_storage().data_ptr() == y.untyped_storage().data_ptr() and \
                y.storage_offset() == x.storage_offset() + x.numel()        # (slices of ONE buffer, back to back)
        if adjacent(ga1, ga2) and adjacent(gb1, gb2):
            n1, n = ga1.numel(), ga1.numel() + ga2.numel()
            a = torch.as_strided(ga1, (n,), (1,))           # (both slices: one buffer)
            b = torch.as_strided(gb1, (n,), (1,))
            out = add2(a, b)
            return out[:n1].view_as(ga1), out[n1:].view_as(ga2)
        return _sum2(ga1, gb1), _sum2(ga2, gb2)


def _tag_master(w, *aliases):
    master = getattr(w, '_ubv_master', w)
    for t in aliases:
        t._ubv_master = master


def fan_out(w):
    a, b = _FanOut.apply(w)
    _tag_master(w, a, b)
    return a, b


def fan_out_pair(w1, w2):
    """((a1, b1), (a2, b2)): two consumers' aliases of two tensors."""
    a1, b1, a2, b2 = _FanOutPair.apply(w1, w2)
    _tag_master(w1, a1, b1)
    _tag_master(w2, a2, b2)
    return (a1, b1), (a2, b2)


# ----------------------------------------------------------------------------------------------- voxels
_WS = {}


def release_workspaces():
    """Drops the cached scratch buffers (per device and stream; the lifting backward's binning workspace
    is the large one: it is sized for the worst case).  They are re-created on the next call."""
    _WS.clear()


# ---- tags on gradient tensors ------------------------------------------------------------------
# A producer's backward may hand its consumer more than the gradient itself (the column sums it already
# has, the fact that the tensor is fresh and may be accumulated into, ...).  The hint travels as a Python
# attribute on the gradient tensor, together with the tensor's version counter and address at tagging
# time: autograd accumulates a second consumer's gradient IN PLACE into the first one's tensor (and keeps
# its attributes), which bumps the version — the reader then sees a stale tag and ignores it.
def tag_grad(t, name, value):
    setattr(t, name, (value, t._version, t.data_ptr()))
    return t


def grad_tag(t, name):
    """The value tagged on ``t`` under ``name``, or None when absent or stale."""
    rec = getattr(t, name, None)
    if rec is None:
        return None
    value, version, ptr = rec
    return value if (t._version == version and t.data_ptr() == ptr) else None


def grad_tag_stale(t, name):
    rec = getattr(t, name, None)
    return rec is not None and not (t._version == rec[1] and t.data_ptr() == rec[2])


def _workspace(nbytes, device):
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


@torch.no_grad()
def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """Deterministic hard voxelization (``ubv_hard_voxelize``), full-capacity, sync-free form.

    Returns voxels (max_voxels, max_points, F), coors (max_voxels, 3) int32 zyx,
    num_points (max_voxels,) int32 and voxel_num (1,) int32 ON DEVICE; rows >= voxel_num are
    zero / undefined.  ``Voxelization.forward`` slices them to the reference's shapes.
    """
    with _need_cuda(points):
        pts = points.float().contiguous()
        N, F = pts.shape
        dev = pts.device
        voxels = torch.empty(max_voxels, max_points, F, dtype=torch.float32, device=dev)
        coors = torch.empty(max_voxels, 3, dtype=torch.int32, device=dev)       # (rows past the count: zeroed by the op)
        num = torch.empty(max_voxels, dtype=torch.int32, device=dev)
        vnum = torch.empty(1, dtype=torch.int32, device=dev)
        nbytes = lib().ubv_hard_voxelize_workspace(N, max_points, max_voxels)
        ws = _workspace(nbytes, dev)
        check(lib().ubv_hard_voxelize(_p(pts), _p(voxels), _p(coors), _p(num), _p(vnum), _p(ws),
                                      ws.numel(), N, F, _lib.float_array(voxel_size),
                                      _lib.float_array(coors_range), max_points, max_voxels, _stream()),
              'hard_voxelize')
        return voxels, coors, num, vnum


@torch.no_grad()
def hard_voxelize_batch(clouds, voxel_size, coors_range, max_points, max_voxels, with_mean=False):
    """A list of clouds in ONE launch chain (``ubv_hard_voxelize_batch``; up to 16 per call, longer lists in chunks):
    per-sample slabs voxels (B, max_voxels, max_points, F), coors (B, max_voxels, 3) int32 zyx, num_points
    (B, max_voxels) int32, voxel_num (B,) int32 ON DEVICE — sample b bit-identical to ``hard_voxelize(clouds[b])``.
    ``with_mean``: also the HardSimpleVFE means (B, max_voxels, F), written by the same chain
    (``ubv_hard_voxelize_batch_vfe``; equal to ``voxel_mean`` of the slab bit for bit)."""
    with _need_cuda(*clouds):
        pts = [c.float().contiguous() for c in clouds]
        B, F = len(pts), pts[0].shape[1]
        assert B > 0 and all(p.dim() == 2 and p.shape[1] == F for p in pts)
        dev = pts[0].device
        voxels = torch.empty(B, max_voxels, max_points, F, dtype=torch.float32, device=dev)
        coors = torch.empty(B, max_voxels, 3, dtype=torch.int32, device=dev)
        num = torch.empty(B, max_voxels, dtype=torch.int32, device=dev)
        vnum = torch.empty(B, dtype=torch.int32, device=dev)
        mean = torch.empty(B, max_voxels, F, dtype=torch.float32, device=dev) if with_mean else None
        for b0 in range(0, B, 16):
            chunk = pts[b0:b0 + 16]
            nb = len(chunk)
            ns = [int(p.shape[0]) for p in chunk]
            nbytes = lib().ubv_hard_voxelize_batch_workspace(nb, max(ns), max_points, max_voxels)
            ws = _workspace(nbytes, dev)
            ptrs = (ctypes.c_void_p * nb)(*[p.data_ptr() if p.numel() else None for p in chunk])
            counts = (ctypes.c_int * nb)(*ns)
            if with_mean:
                check(lib().ubv_hard_voxelize_batch_vfe(ptrs, counts, nb, _p(voxels[b0:]), _p(coors[b0:]), _p(num[b0:]),
                                                        _p(vnum[b0:]), _p(mean[b0:]), _p(ws), ws.numel(), F,
                                                        _lib.float_array(voxel_size), _lib.float_array(coors_range),
                                                        max_points, max_voxels, _stream()), 'hard_voxelize_batch_vfe')
                continue
            check(lib().ubv_hard_voxelize_batch(ptrs, counts, nb, _p(voxels[b0:]), _p(coors[b0:]), _p(num[b0:]),
                                                _p(vnum[b0:]), _p(ws), ws.numel(), F, _lib.float_array(voxel_size),
                                                _lib.float_array(coors_range), max_points, max_voxels, _stream()),
                  'hard_voxelize_batch')
        return (voxels, coors, num, vnum, mean) if with_mean else (voxels, coors, num, vnum)


@torch.no_grad()
def dynamic_voxelize(points, voxel_size, coors_range):
    """[ext] mmdet3d dynamic_voxelize: (N,3) int32 zyx coords, -1 outside."""
    with _need_cuda(points):
        pts = points.float().contiguous()
        N, F = pts.shape
        coors = torch.empty(N, 3, dtype=torch.int32, device=pts.device)
        check(lib().ubv_dynamic_voxelize(_p(pts), _p(coors), N, F, _lib.float_array(voxel_size),
                                         _lib.float_array(coors_range), _stream()), 'dynamic_voxelize')
        return coors


_REDUCE = {'sum': 0, 'mean': 1, 'max': 2}


@torch.no_grad()
def dynamic_scatter(feats, coors, reduce_type='max'):
    """[ext] mmdet3d ``dynamic_point_to_voxel_forward`` (``ubv_dynamic_point_to_voxel_forward``),
    sync-free full-capacity form.

    feats (N, C) float, coors (N, D) int32 -> voxel_feats (N, C), voxel_coors (N, D) int32,
    point2voxel_map (N,) int32, voxel_points_count (N,) int32, voxel_num (2,) int32 ON DEVICE
    ([0] voxels M, [1] valid points); rows >= M are unspecified.  ``DynamicScatter`` slices them to
    the published op's shapes."""
    if reduce_type not in _REDUCE:
        raise ValueError(f"reduce_type must be 'sum', 'mean' or 'max', got {reduce_type!r}")
    with _need_cuda(feats, coors):
        f = feats.float().contiguous()
        c = coors.to(torch.int32).contiguous()
        N, C = f.shape
        D = c.shape[1]
        dev = f.device
        n = max(N, 1)
        vf = torch.empty(n, C, dtype=torch.float32, device=dev)
        vc = torch.empty(n, D, dtype=torch.int32, device=dev)
        mp = torch.empty(N, dtype=torch.int32, device=dev)
        cnt = torch.empty(n, dtype=torch.int32, device=dev)
        vnum = torch.empty(2, dtype=torch.int32, device=dev)
        nws = int(lib().ubv_dynamic_scatter_workspace(N))
        ws = _workspace(nws, dev) if nws > 0 else None
        check(lib().ubv_dynamic_point_to_voxel_forward(_p(f), _p(c), N, C, D, _REDUCE[reduce_type],
                                                       _p(vf), _p(vc), _p(mp), _p(cnt), _p(vnum),
                                                       _p(ws), nws, _stream()),
              'dynamic_point_to_voxel_forward')
        return vf, vc, mp, cnt, vnum


@torch.no_grad()
def voxel_mean(voxels, num_points, voxel_num=None):
    """[ext] HardSimpleVFE: per-voxel mean of the stored points."""
    with _need_cuda(voxels, num_points):
        M, T, F = voxels.shape
        mean = torch.empty(M, F, dtype=torch.float32, device=voxels.device)     # (rows past voxel_num: zeroed by the op)
        check(lib().ubv_voxel_mean(_p(voxels.contiguous()), _p(num_points.contiguous()), _p(voxel_num),
                                   _p(mean), M, T, F, _stream()), 'voxel_mean')
        return mean


@torch.no_grad()
def sparse_to_dense(feats, coors, batch_size, spatial_shape, m_dev=None):
    """SparseConvTensor.dense(): (M,C) feats at (M,4) int32 (b,z,y,x) -> (B,C,D,H,W)."""
    with _need_cuda(feats, coors):
        M, C = feats.shape
        D, Hs, Ws = spatial_shape
        dense = torch.zeros(batch_size, C, D, Hs, Ws, dtype=torch.float32, device=feats.device)
        check(lib().ubv_sparse_to_dense(_p(feats.float().contiguous()), _p(coors.contiguous()),
                                        _p(m_dev), M, _p(dense), batch_size, C, D, Hs, Ws, _stream()),
              'sparse_to_dense')
        return dense


# ----------------------------------------------------------------------------------------------- sparse 3-D convolution
def _i3(v):
    import ctypes
    return (ctypes.c_int * 3)(*[int(x) for x in v])


def spconv_out_dims(in_dims, ksize, stride, pad):
    """Output (D, H, W) of a strided sparse convolution (dilation 1): floor((d + 2 p - k) / s) + 1."""
    return tuple((int(d) + 2 * int(p) - int(k)) // int(s) + 1 for d, k, s, p in zip(in_dims, ksize, stride, pad))


@torch.no_grad()
def spconv_hash(coors, dims):
    """Hash table of the active voxels ``coors`` [N, 4] int32 (b, z, y, x) of a (D, H, W) map."""
    with _need_cuda(coors):
        N = coors.shape[0]
        slots = int(lib().ubv_spconv_table_slots(N))
        keys = torch.empty(slots, dtype=torch.int64, device=coors.device)
        vals = torch.empty(slots, dtype=torch.int32, device=coors.device)
        check(lib().ubv_spconv_hash_build(_p(coors), N, int(dims[0]), int(dims[1]), int(dims[2]), _p(keys), _p(vals),
                                          slots, _stream()), 'spconv_hash_build')
        return keys, vals


@torch.no_grad()
def spconv_neighbors(coors, batch_size, row_dims, target_dims, ksize, stride, pad, table, transposed=False):
    """Neighbour map [kvol, rows] int32 (``ubv_spconv_neighbors``): the index, in the hashed map, of the voxel
    each (row, kernel offset) reaches, or -1."""
    with _need_cuda(coors, *table):
        rows = coors.shape[0]
        kvol = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
        nbr = torch.empty(kvol, rows, dtype=torch.int32, device=coors.device)
        check(lib().ubv_spconv_neighbors(_p(coors), rows, int(batch_size), _i3(row_dims), _i3(target_dims), _i3(ksize),
                                         _i3(stride), _i3(pad), 1 if transposed else 0, _p(table[0]), _p(table[1]),
                                         table[0].numel(), _p(nbr), rows, _stream()), 'spconv_neighbors')
        return nbr


@torch.no_grad()
def spconv_subm_map(coors, batch_size, dims, ksize):
    """Neighbour map of a submanifold convolution (outputs = the inputs' sites, stride 1, padding k // 2)."""
    coors = coors.contiguous()
    table = spconv_hash(coors, dims)
    pad = [int(k) // 2 for k in ksize]
    return spconv_neighbors(coors, batch_size, dims, dims, ksize, (1, 1, 1), pad, table)


@torch.no_grad()
def spconv_site_chain(coors, batch_size, in_dims, layers):
    """Output sites of a CHAIN of strided sparse convolutions (``layers``: (ksize, stride, pad) each, every layer's
    inputs the previous one's outputs) -> [(out_coors [M, 4] int32 in ascending key order, out_dims)].  Built on the
    device back to back (``ubv_spconv_output_sites``: mark bits, popcount scan, emit); the host reads all the counts
    in ONE copy at the end (spconv's get_indice_pairs reads one per layer)."""
    with _need_cuda(coors):
        cur = coors.contiguous()
        dev = cur.device
        L = len(layers)
        counts = torch.empty(max(L, 1), dtype=torch.int32, device=dev)
        bound, n_dev, dims, bufs = cur.shape[0], None, tuple(int(d) for d in in_dims), []
        for l, (ksize, stride, pad) in enumerate(layers):
            out_dims = spconv_out_dims(dims, ksize, stride, pad)
            words = int(lib().ubv_spconv_sites_words(int(batch_size), _i3(out_dims)))
            if words <= 0:
                raise ValueError(f'sparse convolution: empty output map {out_dims}')
            reach = 1
            for k, s_ in zip(ksize, stride):
                reach *= -(-int(k) // int(s_))                 # offsets per axis that can hit a stride-aligned cell
            cap = max(1, min(reach * bound, int(batch_size) * out_dims[0] * out_dims[1] * out_dims[2]))
            bitmap = torch.empty(words, dtype=torch.int32, device=dev)
            sums = torch.empty(words // 256 + 2, dtype=torch.int32, device=dev)
            buf = torch.empty(cap, 4, dtype=torch.int32, device=dev)
            check(lib().ubv_spconv_output_sites(_p(cur), _p(n_dev), bound, int(batch_size), _i3(dims), _i3(out_dims),
                                                _i3(ksize), _i3(stride), _i3(pad), _p(bitmap), words, _p(sums), _p(buf),
                                                cap, _p(counts[l:l + 1]), _stream()), 'spconv_output_sites')
            bufs.append((buf, out_dims))
            cur, n_dev, bound, dims = buf, counts[l:l + 1], cap, out_dims
        host = counts.cpu().tolist()                           # the one host read
        out = []
        for (buf, od), m in zip(bufs, host):
            out.append((buf[:m].clone() if buf.shape[0] > 2 * m else buf[:m], od))
        return out


@torch.no_grad()
def spconv_strided_maps(coors, batch_size, in_dims, ksize, stride, pad, sites=None):
    """Rulebook of a strided sparse convolution: (out_coors [M, 4] in ascending key order, out_dims,
    nbr_fwd [kvol, M] (inputs of each output), nbr_bwd [kvol, N] (outputs of each input)).  ``sites`` =
    (out_coors, out_dims) when the caller planned the output set already (``spconv_site_chain``); otherwise it is
    built here (one host read of its size, as spconv's get_indice_pairs does)."""
    with _need_cuda(coors):
        coors = coors.contiguous()
        if sites is None:
            sites = spconv_site_chain(coors, batch_size, in_dims, [(ksize, stride, pad)])[0]
        out_coors, out_dims = sites
        t_in = spconv_hash(coors, in_dims)
        t_out = spconv_hash(out_coors, out_dims)
        nbr_fwd = spconv_neighbors(out_coors, batch_size, out_dims, in_dims, ksize, stride, pad, t_in)
        nbr_bwd = spconv_neighbors(coors, batch_size, in_dims, out_dims, ksize, stride, pad, t_out, transposed=True)
        return out_coors, out_dims, nbr_fwd, nbr_bwd


@torch.no_grad()
def spconv_operand(w):
    """Weight blocks [kvol, Cout, Cin] (any float dtype) -> the kernel's operand: rows padded to a multiple of
    32 per block; f32 -> (hi, lo) bf16 halves, 16-bit -> (w, None).  (Generic form; the layers use
    ``spconv_weight_operand`` on the stored layout.)"""
    kvol, cout, cin = w.shape
    coutp = (cout + 31) // 32 * 32
    if coutp != cout:
        w = torch.nn.functional.pad(w, (0, 0, 0, coutp - cout))
    w = w.contiguous()
    if w.dtype == torch.float32:
        hi, lo, _, _ = split_weight(w.view(kvol * coutp, cin), transposed=False)
        return hi, lo
    return w, None


@torch.no_grad()
def spconv_weight_operand(w, transpose, flip=False):
    """spconv's stored weight [kvol, Cin, Cout] -> ``spconv_gather_mma``'s operand in one launch
    (``ubv_spconv_weight_operand``): ``transpose`` = forward ([kvol, CoutP, Cin]), else the input gradient's
    ([kvol, CinP, Cout], offsets mirrored when ``flip``)."""
    with _need_cuda(w):
        w = w.contiguous()
        kvol, cin, cout = w.shape
        rows, K = (cout, cin) if transpose else (cin, cout)
        rowsp = (rows + 31) // 32 * 32
        f32 = w.dtype == torch.float32
        hi = torch.empty(kvol, rowsp, K, dtype=torch.bfloat16 if f32 else w.dtype, device=w.device)
        lo = torch.empty_like(hi) if f32 else None
        check(lib().ubv_spconv_weight_operand(_p(w), kvol, cin, cout, 1 if transpose else 0, 1 if flip else 0, _dt(w),
                                              _p(hi), _p(lo), _stream()), 'spconv_weight_operand')
        return hi, lo


@torch.no_grad()
def spconv_gather_mma(feats, nbr, w_hi, w_lo, cout):
    """out[row] = sum_k feats[nbr[k][row]] . w[k]^T (``ubv_spconv_gather_mma``)."""
    with _need_cuda(feats, nbr, w_hi, w_lo):
        feats = feats.contiguous()
        kvol, rows = nbr.shape
        cin = feats.shape[1]
        out = torch.empty(rows, cout, dtype=feats.dtype, device=feats.device)
        check(lib().ubv_spconv_gather_mma(_p(feats), _p(nbr), rows, rows, _p(w_hi), _p(w_lo), _p(out), cin, int(cout),
                                          kvol, _dt(feats), _stream()), 'spconv_gather_mma')
        return out


_SPWG_MULT = int(os.environ.get('UBV_SPWG_MULT', '4'))
_SPWG_WS_MB = int(os.environ.get('UBV_SPWG_WS_MB', '96'))       # budget of the weight-gradient partial sums


@torch.no_grad()
def spconv_pairs(nbr):
    """Compacted rulebook of a neighbour map [kvol, rows] (``ubv_spconv_pairs``): per offset the rows that HAVE a
    neighbour, in row order -> (out_rows, in_rows, counts) int32; entries past counts[k] are unspecified.
    Deterministic, nothing read back; built once per indice key."""
    with _need_cuda(nbr):
        nbr = nbr.contiguous()
        kvol, rows = nbr.shape
        out_rows, in_rows = torch.empty_like(nbr), torch.empty_like(nbr)
        counts = torch.empty(kvol, dtype=torch.int32, device=nbr.device)
        chunks = int(lib().ubv_spconv_pairs_chunks(rows))
        sums = torch.empty(max(1, kvol * chunks), dtype=torch.int32, device=nbr.device)
        check(lib().ubv_spconv_pairs(_p(nbr), rows, rows, kvol, _p(sums), _p(out_rows), _p(in_rows), _p(counts),
                                     _stream()), 'spconv_pairs')
        return out_rows, in_rows, counts


@torch.no_grad()
def spconv_wgrad(grad_out, feats, nbr, pairs=None):
    """[kvol, Cin, Cout] f32 weight gradient of a sparse convolution (``ubv_spconv_wgrad``; with ``pairs`` =
    ``spconv_pairs(nbr)`` the compacted form ``ubv_spconv_wgrad_pairs``), or None when the channel counts are
    outside the kernel's reach."""
    with _need_cuda(grad_out, feats, nbr):
        kvol, rows = nbr.shape
        cout, cin = grad_out.shape[1], feats.shape[1]
        cw = 4 if feats.dtype == torch.float32 else 8
        if rows == 0 or cout % cw or cin % cw or cout > 128 or cin > 128 or grad_out.dtype != feats.dtype:
            return None
        S = int(lib().ubv_spconv_wgrad_splits(rows, kvol))
        if pairs is not None:
            # the pairs of an offset fill only the first counts[k] of its `rows` slots (30 - 50 % on LiDAR clouds) and the
            # slabs past the count exit at once: finer slabs keep the chip full (measured per encoder backward:
            # 1x 10.9 ms, 2x 7.5 ms, 4x 5.8 ms, 8x 5.4 ms + a 0.4 ms longer slab sum)
            S = max(1, min(_SPWG_MULT * S, (rows + 255) // 256))
        blk = cout * cin + cout
        # the partial sums live in the persistent per-stream scratch: cap them at _SPWG_WS_MB (wide layers fill the chip
        # with their kvol tiles anyway; at 128 x 128 channels, kvol = 27 the uncapped 4 S slabs were ~200 MB)
        S = max(1, min(S, (_SPWG_WS_MB << 20) // (4 * kvol * blk)))
        part = _workspace(4 * S * kvol * blk, feats.device)
        out = torch.empty(kvol, blk, dtype=torch.float32, device=feats.device)
        if pairs is not None:
            out_rows, in_rows, counts = pairs
            rc = lib().ubv_spconv_wgrad_pairs(_p(grad_out.contiguous()), _p(feats.contiguous()), _p(in_rows), _p(out_rows),
                                              _p(counts), rows, rows, _p(part), _p(out), cout, cin, kvol, S, _dt(feats),
                                              _stream())
        else:
            rc = lib().ubv_spconv_wgrad(_p(grad_out.contiguous()), _p(feats.contiguous()), _p(nbr), rows, rows, _p(part),
                                        _p(out), cout, cin, kvol, S, _dt(feats), _stream())
        if rc == -3:
            return None
        check(rc, 'spconv_wgrad')
        return out[:, :cout * cin].view(kvol, cout, cin).transpose(1, 2)


def _f32_scratch(n, device):
    return _workspace(4 * n, device)[:4 * n].view(torch.float32)


class _RowsBatchNorm(Function):
    """BatchNorm1d (+ ReLU) over the rows of a sparse feature matrix (``ubv_rows_bn_forward`` / ``_backward``)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, residual=None):
        with _need_cuda(x, gamma, beta, running_mean, running_var):
            x = x.contiguous()
            N, C = x.shape
            res = None if residual is None else residual.to(x.dtype).contiguous()
            g, b = gamma.float().contiguous(), beta.float().contiguous()
            y = torch.empty_like(x)
            if training:
                mean = torch.empty(C, dtype=torch.float32, device=x.device)
                rstd = torch.empty(C, dtype=torch.float32, device=x.device)
            else:
                mean = running_mean.float().contiguous()
                rstd = torch.rsqrt(running_var.float() + eps).contiguous()
            part = _f32_scratch(int(lib().ubv_rows_bn_partial_elems(C)), x.device) if training else None
            check(lib().ubv_rows_bn_forward(_p(x), _p(res), _p(g), _p(b), _p(running_mean if training else None),
                                            _p(running_var if training else None), _p(mean), _p(rstd), _p(part), _p(y),
                                            N, C, float(eps), float(momentum), int(relu), int(training), _dt(x),
                                            _stream()), 'rows_bn_forward')
            # (with a residual the ReLU mask of the backward is the stored output itself)
            ctx.save_for_backward(x, g, b, mean, rstd, y if res is not None else None)
            ctx.relu, ctx.training, ctx.dts = int(relu), bool(training), (gamma.dtype, beta.dtype)
            ctx.res_dt = None if residual is None else residual.dtype
            return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        with _need_cuda(grad_y):
            x, g, b, mean, rstd, yout = ctx.saved_tensors
            N, C = x.shape
            gy = grad_y.to(x.dtype).contiguous()
            if not ctx.training:                 # running statistics are constants: an affine map (+ ReLU mask)
                xh = (x.float() - mean) * rstd
                pos = (yout > 0) if yout is not None else ((xh * g + b) > 0)
                dyp = gy.float() * pos if ctx.relu else gy.float()
                return ((dyp * (g * rstd)).to(x.dtype), (dyp * xh).sum(0).to(ctx.dts[0]), dyp.sum(0).to(ctx.dts[1]),
                        None, None, None, None, None, None, None if yout is None else dyp.to(ctx.res_dt))
            gx = torch.empty_like(x)
            gres = torch.empty_like(x) if yout is not None else None
            dg = torch.empty(C, dtype=torch.float32, device=x.device)
            db = torch.empty(C, dtype=torch.float32, device=x.device)
            part = _f32_scratch(int(lib().ubv_rows_bn_partial_elems(C)), x.device)
            check(lib().ubv_rows_bn_backward(_p(x), _p(gy), _p(yout), _p(g), _p(b), _p(mean), _p(rstd), _p(part), _p(dg),
                                             _p(db), _p(gx), _p(gres), N, C, ctx.relu, _dt(x), _stream()),
                  'rows_bn_backward')
            return (gx, dg.to(ctx.dts[0]), db.to(ctx.dts[1]), None, None, None, None, None, None,
                    None if gres is None else gres.to(ctx.res_dt))


def rows_batch_norm(x, bn, relu=False, residual=None):
    """``relu?(bn(x) (+ residual))`` for a ``torch.nn.BatchNorm1d`` ``bn`` over the rows of ``x`` [N, C] in one fused pass each way;
    updates ``bn``'s running statistics in training mode exactly as the module would (momentum, unbiased variance,
    ``num_batches_tracked``).  None when the shape is outside the kernels' reach (caller uses the module)."""
    C = x.shape[1]
    if not x.is_cuda or x.dim() != 2 or x.shape[0] == 0 or C % 4 or 256 % (C // 4) or not bn.affine or \
            (bn.training and bn.momentum is None) or x.dtype not in _DT:
        return None
    training = bn.training or bn.running_mean is None
    # the module's own edge cases stay the module's: one row in training mode (torch raises "Expected more than 1 value
    # per channel"), running statistics that are not f32 (the kernels take raw f32 pointers)
    if training and x.shape[0] < 2:
        return None
    if bn.track_running_stats and bn.running_mean is not None and \
            (bn.running_mean.dtype != torch.float32 or bn.running_var.dtype != torch.float32):
        return None
    if residual is not None and (residual.shape != x.shape or not residual.is_cuda):
        return None
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _BN_TICKS is not None:
            _BN_TICKS.append(bn.num_batches_tracked)           # (one fused increment per encoder pass)
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if not training and rm is None:
        return None
    return _RowsBatchNorm.apply(x, bn.weight, bn.bias, rm, rv, training, bn.momentum if bn.momentum is not None else 0.0,
                                bn.eps, relu, residual)


_BN_TICKS = None


class batched_bn_ticks:
    """Inside this context ``rows_batch_norm`` collects the ``num_batches_tracked`` counters it would increment and the
    exit adds 1 to all of them in one multi-tensor launch (21 tiny kernels per middle-encoder pass otherwise)."""

    def __enter__(self):
        global _BN_TICKS
        self._outer, _BN_TICKS = _BN_TICKS, []
        return self

    def __exit__(self, *exc):
        global _BN_TICKS
        ticks, _BN_TICKS = _BN_TICKS, self._outer
        if ticks and exc[0] is None:          # (a pass that raised counted nothing: torch ticks per completed forward)
            if self._outer is not None:
                self._outer.extend(ticks)
            else:
                # a module that ran twice in the pass appears twice: one entry per tensor with its count (a multi-tensor
                # launch updates duplicate entries from unsynchronised blocks and loses increments)
                seen = {}
                for tk in ticks:
                    ent = seen.setdefault(id(tk), [tk, 0])
                    ent[1] += 1
                by_n = {}
                for tk, n in seen.values():
                    by_n.setdefault(n, []).append(tk)
                for n, group in by_n.items():
                    torch._foreach_add_(group, n)
        return False


# ----------------------------------------------------------------------------------------------- GridMask
@torch.no_grad()
def grid_mask(x, d, length, st_h, st_w, use_h=True, use_w=True, mode=1):
    """x (..., h, w) times the GridMask stripe mask of (d, length, st_h, st_w) (``ubv_grid_mask``)."""
    with _need_cuda(x):
        x = x.contiguous()
        h, w = x.shape[-2], x.shape[-1]
        y = torch.empty_like(x)
        check(lib().ubv_grid_mask(_p(x), _p(y), x.numel() // (h * w), h, w, int(d), int(length), int(st_h), int(st_w),
                                  1 if use_h else 0, 1 if use_w else 0, int(mode), _dt(x), _stream()), 'grid_mask')
        return y


# ----------------------------------------------------------------------------------------------- DCNv2
def _dcn_out_size(size, k, s, p, d):
    return (size + 2 * p - (d * (k - 1) + 1)) // s + 1


def _dcn_gemm(a, w):
    """a [M, K] @ w[N, K]^T on the matrix cores where the shape allows it (f32: split weights), else the library."""
    if a.dtype == torch.float32:
        wh, wl, _, _ = split_weight(w, transposed=False)
        y = gemm_nt(a, wh, wl)
    else:
        y = gemm_nt(a, w.to(a.dtype).contiguous())
    return y if y is not None else a @ w.to(a.dtype).t()


class ModulatedDeformConv2dFunction(Function):
    """Drop-in for [ext] mmcv ``ModulatedDeformConv2dFunction`` (same argument order; ``groups`` must be 1, the only
    value the configs use).  x / the result are NCHW tensors as in mmcv; internally channels-last — pass
    ``channels_last`` tensors to avoid the two layout copies."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
        if groups != 1:
            raise NotImplementedError('ModulatedDeformConv2d: groups != 1 is not built')
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(padding), pair(dilation)
        N, C, H, W = x.shape
        Cout, _, kh, kw = weight.shape
        Ho, Wo = _dcn_out_size(H, kh, sh, ph, dh), _dcn_out_size(W, kw, sw, pw, dw)
        with _need_cuda(x, offset, mask, weight, bias):
            xn = x.detach().permute(0, 2, 3, 1).contiguous()
            off = offset.detach().to(x.dtype).contiguous()
            mk = mask.detach().to(x.dtype).contiguous()
            geom = (N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, int(deform_groups))
            col = torch.empty(N * Ho * Wo, kh * kw * C, dtype=x.dtype, device=x.device)
            check(lib().ubv_dcn_im2col(_p(xn), _p(off), _p(mk), _p(col), *geom, _dt(x), _stream()), 'dcn_im2col')
            wp = weight.detach().permute(0, 2, 3, 1).reshape(Cout, kh * kw * C).contiguous()
            y = _dcn_gemm(col, wp)
            if bias is not None:
                y = y + bias.detach().to(y.dtype)
        ctx.save_for_backward(xn, off, mk, col, wp)
        ctx.geom, ctx.has_bias, ctx.wshape = geom, bias is not None, weight.shape
        ctx.dtypes = (offset.dtype, mask.dtype, weight.dtype)
        return y.view(N, Ho, Wo, Cout).permute(0, 3, 1, 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        xn, off, mk, col, wp = ctx.saved_tensors
        geom = ctx.geom
        N, H, W, C, Ho, Wo, kh, kw = geom[:8]
        Cout = ctx.wshape[0]
        with _need_cuda(grad_out, xn):
            go = grad_out.permute(0, 2, 3, 1).contiguous().view(N * Ho * Wo, Cout).to(xn.dtype)
            gcol = _dcn_gemm(go, wp.t().contiguous())                       # [M, K] = dY . Wp
            gx = torch.zeros(N, H, W, C, dtype=torch.float32, device=xn.device)
            goff = torch.empty(off.shape, dtype=torch.float32, device=xn.device)
            gmask = torch.empty(mk.shape, dtype=torch.float32, device=xn.device)
            check(lib().ubv_dcn_col2im(_p(gcol.contiguous()), _p(xn), _p(off), _p(mk), _p(gx), _p(goff), _p(gmask),
                                       *geom, _dt(xn), _stream()), 'dcn_col2im')
            gw = gb = None
            if ctx.needs_input_grad[3] or (ctx.has_bias and ctx.needs_input_grad[4]):
                res = gemm_wgrad(go, col)
                if res is not None:
                    gwf, gb = res
                else:
                    gwf, gb = go.float().t() @ col.float(), go.float().sum(0)
                gw = gwf.view(Cout, kh, kw, C).permute(0, 3, 1, 2).to(ctx.dtypes[2])
                gb = gb.to(ctx.dtypes[2]) if ctx.has_bias else None
        return (gx.permute(0, 3, 1, 2).to(xn.dtype), goff.to(ctx.dtypes[0]), gmask.to(ctx.dtypes[1]), gw, gb,
                None, None, None, None, None)


modulated_deform_conv2d = ModulatedDeformConv2dFunction.apply
