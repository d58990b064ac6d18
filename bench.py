#!/usr/bin/env python3
"""Headline benchmark: samples/s of the UniBEV BEV-encoder hot path, forward + backward, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one training pass of the hot path over one batch of synthetic nuScenes-shaped inputs
already resident in HBM (BASELINE.json north_star: 6 cameras x 256x704 -> 8x22 stride-32 feature
tokens each, a 180x180x256 LiDAR BEV feature map as the voxelize/backbone front end leaves it,
200x200x256 BEV queries, L+C CNW fusion with modality dropout, bs = 2 per GPU = configs[3]):
``UniBEV_Head.forward_bev`` (BEV queries + learned positional encoding -> both 3-layer encoders ->
CNW -> fusion), backward from a fixed random cotangent on ``fused_bev_embed`` to every encoder
parameter and the input features, one RCCL all-reduce of the flat gradient buffer (one process per
GPU), gradient clipping and an AdamW step.  Data parallel only: the per-GPU batch is fixed, so
scaling is weak.  Forward + backward replay as HIP graphs (unibev_amd/graph_step.py).

Precision.  The reference computes in fp32 and BASELINE's bar is 1e-3 on BEV
features.  The HEADLINE (`value`, `dtype`) is the fp32 path (f32 storage, Linear layers as split-bf16
MFMA products with f32 accumulation), the only mode inside the bar on every reference-recorded
fixture.  The 16-bit autocast paths run the same step ~2x faster and are reported as sub-records
under `lowp`.  Every mode's distance to the reference-recorded full-size vectors is MEASURED IN THIS
RUN (`parity`: {fixture: normwise distance}, bar, pass) — the same quantity tests/test_modules_gpu.py
asserts; see DESIGN.md section 4.

Rank 0 prints ONE JSON line on stdout, strict JSON of at most LINE_LIMIT bytes (`compact`): the contract fields,
  roofline      the dominant deformable-sampling OP of the headline run: compulsory bytes per launch
                 / the op's duration, HIP events on the launch stream;
  roofline_ops  every sampling op, forward and backward, as [op, pass, us, frac, PMC traffic / algorithmic bytes];
  phases        graph replay / exposed all-reduce / clip + AdamW milliseconds per step;
  cpu_baseline  the oracle (CPU port of the reference path) timed on this host (N = 1 only);
  and one number per sub-record (spread_value + its ops, ieee_gemm_value, lowp values + parity verdicts, the 256x256
  projection GEMM, the voxel front end, the k1 operator).
The LONG form of every record (per-kernel times, notes, the gemm table, the 16-bit runs' own roofline_ops ...) is written
to --extras-file (bench_extras.json beside this script) and to one `#extras `-prefixed line on stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0      # measured float4-copy peak (same guide); fractions are reported against both
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}   # dense (MI355X_MICROARCH.md)
PARITY_BAR = 1e-3          # BASELINE.json north_star: BEV features within 1e-3 rel of the reference's
# fixtures of reference-recorded vectors per workload (tests/golden/*.npz, recorded from the reference's own modules by
# tests/golden/make_golden.py; inputs are regenerated from seeds and checksummed)
PARITY_FIXTURES = {'LC_cnw': ('fullsize_init', 'fullsize'), 'LC_cat128': ('fullsize_cat128',), 'C': ('C',), 'L': ('L',)}
FIXTURE_NOTE = {'fullsize_init': 'cfg4 shapes, initial sampling parameters, spatially correlated maps (the operating point)',
                'fullsize': 'cfg4 shapes, i.i.d. maps, random offset weights (adversarial)',
                'fullsize_cat128': 'cfg5 shapes (C=128, 6 x 25x45 maps), random parameters',
                'C': 'camera-only small fixture', 'L': 'LiDAR-only small fixture'}


def parity_record(workload, names, device, fp32_stream):
    """Normwise distance of each precision mode's ``fused_bev_embed`` to the REFERENCE-recorded vectors of the workload's
    fixtures, measured now on this device (eval mode, bs = 1 at the fixture's size): {mode: {fixture: distance}} plus a
    verdict against the 1e-3 bar.  Test infrastructure only reads fixtures here; nothing under oracle/ is touched."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _util import encoder_case, t
    from unibev_amd import build_transformer
    out = {n: {} for n in names}
    for fx in PARITY_FIXTURES[workload]:
        cfg, sd, inp, g = encoder_case(fx)
        model = build_transformer(json.loads(json.dumps(cfg))).to(device).eval()
        model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
        img = None if inp['img'] is None else [t(x, device=device) for x in inp['img']]
        pts = None if inp['pts'] is None else [t(x, device=device) for x in inp['pts']]
        for n in names:
            dt = DTYPES[n]
            model.lowp_stream = not fp32_stream
            with torch.no_grad(), value_storage(n), torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32):
                fused = model.encode(img, pts, t(inp['bev_q'], device=device), inp['bev_h'], inp['bev_w'],
                                     bev_pos=t(inp['bev_pos'], device=device), img_metas=inp['metas'])
            f = fused.float().cpu().numpy().reshape(-1)
            ref = g['fused_sub'] if 'fused_sub' in g else g['fused'].reshape(-1)
            f = f[g['fused_idx']] if 'fused_idx' in g else f
            out[n][fx] = float(np.linalg.norm(f - ref) / np.linalg.norm(ref))
        del model
    torch.cuda.empty_cache()
    return {n: {'bar': PARITY_BAR, 'distance': d, 'pass': all(v < PARITY_BAR for v in d.values()),
                'fixtures': {k: FIXTURE_NOTE[k] for k in d}} for n, d in out.items()}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=2, help='samples per GPU (cfg4: 2)')
    ap.add_argument('--dtype', default='all', choices=['all', 'bf16', 'fp16', 'fp32', 'value-fp16'],
                    help="'all': fp32 headline + value-fp16, bf16 and fp16 sub-records.  'value-fp16': the f32 step (f32 residual "
                         "stream, offsets, logits, split-bf16 MFMA Linear layers) with ONLY the projected value maps and the "
                         "sampled outputs stored in fp16 (deform_attn.set_value_storage): the sub-f32 mode that holds the 1e-3 "
                         "bar at the operating point and on cfg5")
    ap.add_argument('--workload', default='LC_cnw', choices=['LC_cnw', 'C', 'L', 'LC_cat128'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP graphs')
    ap.add_argument('--torch-optimizer', dest='flat_optimizer', action='store_false',
                    help='torch clip_grad_norm_ + fused AdamW instead of the flat-buffer kernels')
    ap.add_argument('--single-stream', action='store_true',
                    help='both encoders on one stream (default: image and point-cloud encoders on two)')
    ap.add_argument('--two-streams', action='store_true', help='(the default; kept for command lines of round 4)')
    ap.add_argument('--no-extras', action='store_true', help='skip the gemm / voxel records')
    ap.add_argument('--eval-mode', action='store_true', help='dropout / modality dropout off')
    ap.add_argument('--launcher', default='auto', choices=['auto', 'spawn', 'none'],
                    help="'auto': when --gpus N > 1 and no torchrun environment is present, re-exec under "
                         "torch.distributed.run with N ranks; 'spawn': always (also N = 1: one rank with a process "
                         "group, the multi-GPU code path on a one-GPU box); 'none': never")
    ap.add_argument('--cpu-baseline-plan', action='store_true',
                    help='BASELINE.md section 3 with 1 warm-up + up to 5 timed passes per entry (forward of cfg1-cfg5, forward + '
                         'backward of cfg2 / cfg3) through the oracle on this host, as `cpu_baseline_plan`; a few minutes of CPU '
                         'time.  The DEFAULT run carries the same plan bounded to ~1 minute (1 warm-up + <= 2 timed passes per entry)')
    ap.add_argument('--no-cpu-baseline-plan', action='store_true', help='skip `cpu_baseline_plan` altogether')
    ap.add_argument('--cpu-baseline-child', type=int, default=0, metavar='THREADS',
                    help='(internal) time the CPU baseline pass with THREADS torch threads and print the seconds per pass')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher + process group + gradient-exchange protocol on tiny CPU tensors (gloo when no GPU is '
                         'visible): rank environment, one JSON line from rank 0, `rccl_ranks` == N; no kernels, no timing claim')
    ap.add_argument('--allreduce-algo', default='auto', choices=['auto', 'default', 'ring'],
                    help="RCCL all-reduce algorithm: 'default' = RCCL's own choice (direct on a fully connected xGMI node), "
                         "'ring' = NCCL_ALGO=Ring, 'auto' = ring for --exchange split (its collectives run beside the "
                         "backward's GEMMs: dp.init_distributed), default otherwise.  Recorded in config.collective")
    ap.add_argument('--allow-eager', action='store_true',
                    help='if HIP-graph capture fails, time eager launches instead of exiting non-zero')
    ap.add_argument('--no-parity', action='store_true',
                    help="skip the run-time distance of each precision mode to the reference-recorded full-size vectors")
    ap.add_argument('--params', default='both', choices=['init', 'spread', 'both'],
                    help="sampling parameters of the timed model: 'init' = the reference's init_weights (every query samples "
                         "the same compass-grid offsets: the headline), 'spread' = seeded sampling_offsets weights that "
                         "scatter the offsets by ~3 pixels per query (a trained-looking operating point), 'both' = the "
                         "headline on 'init' and a `spread` sub-record")
    ap.add_argument('--exchange', default='auto', choices=['auto', 'split', 'single'],
                    help="gradient exchange: 'split' = the backward is two HIP graphs cut at the first encoder layers and the "
                         "upper layers' segment of the flat gradient buffer is all-reduced beside the second graph; 'single' = "
                         "one graph, one message after it; 'auto' = single (the overlapped exchange has only ever run with one "
                         "rank: it stays opt-in until a multi-GPU box has executed it — a rank-local failure inside a split-mode "
                         "collective would leave the other ranks waiting).  The overlapped "
                         "segments are SUM all-reduces on RCCL's ring (--allreduce-algo: the ring "
                         "FuncSum kernels hold no packed f32 instruction, profiles/r04_rccl_packed_f32_functions.txt) and are "
                         "divided by the world size when waited for")
    ap.add_argument('--lr', type=float, default=2e-4, help='AdamW learning rate (0: the step runs, the parameters stay put)')
    ap.add_argument('--grad-checksum', action='store_true',
                    help='add order-independent checksums of the flat gradient buffer after the last step (`grad_checksum`) to '
                         'the line: tests compare the exchange modes with them')
    ap.add_argument('--extras-file', default=os.path.join(ROOT, 'bench_extras.json'),
                    help="where the long form of the record goes ('' = nowhere); the final stdout line is the short form")
    ap.add_argument('--no-ieee-gemm', action='store_true', help='skip the `ieee_gemm` sub-record (f32 step on library IEEE GEMMs)')
    ap.add_argument('--fp32-stream', action='store_true',
                    help='keep the encoder residual stream in f32 under autocast (default: the '
                         'autocast dtype, as the reference\'s fp16 mode runs it)')
    return ap.parse_args()


WORKLOADS = {
    # name: (cfg kwargs, image (H, W), lidar feature hw, description)
    'LC_cnw': (dict(embed_dims=256, fusion_method='linear', feature_norm='ChannelNormWeights',
                    drop_modality=0.5), (256, 704), (180, 180),
               'unibev_nus_LC_cnw_256_modality_dropout: L+C CNW, 6x(8x22) img tokens [256x704/32], '
               '180x180 LiDAR BEV feats, 200x200x256 BEV'),
    'C': (dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='C'), (256, 704),
          None, 'unibev_nus_C: camera-only, 6x(8x22) img tokens, 200x200x256 BEV'),
    'L': (dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='L'), (256, 704),
          (180, 180), 'unibev_nus_L: LiDAR-only, 180x180 LiDAR BEV feats, 200x200x256 BEV'),
    'LC_cat128': (dict(embed_dims=128, fusion_method='cat', feature_norm=None, drop_modality=0.5),
                  (800, 1440), (180, 180),
                  'unibev_nus_LC_cat_128_modality_dropout: L+C cat, 6x(25x45) img tokens '
                  '[800x1440/32], 200x200x128 BEV'),
}
DTYPES = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32, 'value-fp16': torch.float32}
VALUE_STORAGE = {'value-fp16': torch.float16}     # modes that store only the projected value maps / sampled outputs in 16 bits


class value_storage:
    """``with value_storage(mode)``: deform_attn.set_value_storage for the modes of VALUE_STORAGE, restored on exit."""

    def __init__(self, name):
        self.st = VALUE_STORAGE.get(name)

    def __enter__(self):
        from unibev_amd.modules.deform_attn import set_value_storage
        self.prev = set_value_storage(self.st)

    def __exit__(self, *exc):
        from unibev_amd.modules.deform_attn import set_value_storage
        set_value_storage(self.prev)


def build_head(workload, device):
    from unibev_amd import configs as cfgs
    from unibev_amd.registry import HEADS
    kw, img_hw, pts_hw, _ = WORKLOADS[workload]
    C = kw['embed_dims']
    tcfg = cfgs.transformer_cfg(decoder=None, **kw)
    tcfg.pop('decoder')
    head = HEADS.build(dict(type='UniBEV_Head', bev_h=200, bev_w=200, num_query=900, num_classes=10,
                            in_channels=C, transformer=tcfg,
                            bbox_coder=dict(pc_range=cfgs.PC_RANGE),
                            positional_encoding=dict(type='LearnedPositionalEncoding',
                                                     num_feats=C // 2, row_num_embed=200,
                                                     col_num_embed=200)))
    head.init_weights()
    # decoder-side parameters are not on the path
    for n, p in head.named_parameters():
        if n.startswith('query_embedding') or n.startswith('transformer.reference_points'):
            p.requires_grad_(False)
    return head.to(device), tcfg


SPREAD_SIGMA_PX = 3.0


def set_sampling_params(head, mode):
    """'init': sampling_offsets weights zero (the reference's init_weights: offsets = the bias' compass grid for every
    query).  'spread': seeded N(0, (sigma / sqrt(C))^2) weights — on LayerNormed queries the offsets of a (head, point)
    then scatter by ~SPREAD_SIGMA_PX pixels around the grid from query to query, as a trained layer's do (what
    tools/bench_lift.py --random-offsets feeds the kernels)."""
    from unibev_amd import linear as UL
    g = torch.Generator(device='cpu').manual_seed(77)
    with torch.no_grad():
        for m in head.modules():
            so = getattr(m, 'sampling_offsets', None)
            if isinstance(so, torch.nn.Linear):
                if mode == 'spread':
                    w = torch.randn(so.weight.shape, generator=g) * (SPREAD_SIGMA_PX / so.in_features ** 0.5)
                    so.weight.copy_(w.to(so.weight.device))
                else:
                    so.weight.zero_()
    UL.mark_weights_changed()


def synth_inputs(workload, bs, dtype, device, rank):
    from unibev_amd import synthetic as syn
    kw, img_hw, pts_hw, _ = WORKLOADS[workload]
    C = kw['embed_dims']
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    img = pts = None
    mods = kw.get('modalities', 'LC')
    if 'C' in mods:
        fh, fw = img_hw[0] // 32, img_hw[1] // 32
        img = [torch.randn(bs, 6, C, fh, fw, generator=g).to(device=device, dtype=dtype)
               .requires_grad_()]
    if 'L' in mods:
        pts = [torch.randn(bs, C, *pts_hw, generator=g).to(device=device, dtype=dtype)
               .requires_grad_()]
    l2i = torch.from_numpy(np.stack([syn.camera_rig(6, img_hw) for _ in range(bs)])).float().to(device)
    metas = [dict(lidar2img=l2i[b], img_shape=[(img_hw[0], img_hw[1], 3)] * 6) for b in range(bs)]
    return img, pts, metas


# ---- compulsory bytes of the sampling ops -------------------------------------
def k1_bytes(B, S, Nq, C, H, P, esize, backward):
    """k1 fwd: B S C e_v + B Nq H L P 3 4 + B Nq C e_o;  bwd: B S C (e_v + e_g) + 2 B Nq H L P 3 4 + B Nq C e_o."""
    loc = B * Nq * H * P * 3 * 4
    if backward:
        return B * S * C * 2 * esize + 2 * loc + B * Nq * C * esize
    return B * S * C * esize + loc + B * Nq * C * esize


def sampling_ops(workload, bs, esize, visible_pairs):
    """op name -> (map tag in the library's kernel names, fwd bytes, bwd bytes) per launch."""
    kw, img_hw, pts_hw, _ = WORKLOADS[workload]
    C, H, Nq = kw['embed_dims'], 8, 200 * 200
    ops = {'self_attn': (f'Nc=1 map=200x200 Nq={Nq} B={bs}', k1_bytes(bs, Nq, Nq, C, H, 4, esize, False),
                         k1_bytes(bs, Nq, Nq, C, H, 4, esize, True))}
    if bs > 1:      # the first layer's self-attention runs once for the batch (modules/encoders.py)
        ops['self_attn (first layer, one sample for the batch)'] = (
            f'Nc=1 map=200x200 Nq={Nq} B=1', k1_bytes(1, Nq, Nq, C, H, 4, esize, False),
            k1_bytes(1, Nq, Nq, C, H, 4, esize, True))
    mods = kw.get('modalities', 'LC')
    if 'L' in mods:
        S = pts_hw[0] * pts_hw[1]
        ops['sca_pts'] = (f'Nc=1 map={pts_hw[0]}x{pts_hw[1]} Nq={Nq} B={bs}', k1_bytes(bs, S, Nq, C, H, 8, esize, False),
                          k1_bytes(bs, S, Nq, C, H, 8, esize, True))
    if 'C' in mods:
        fh, fw = img_hw[0] // 32, img_hw[1] // 32
        # the reference pads every camera to max_len; the contract is the UNPADDED count of visible
        # (camera, query) pairs: value maps of all cameras + one row per pair
        S6 = 6 * fh * fw
        pairs = visible_pairs
        fwd = bs * S6 * C * esize + pairs * H * 8 * 3 * 4 + pairs * C * esize
        bwd = bs * S6 * C * 2 * esize + 2 * pairs * H * 8 * 3 * 4 + pairs * C * esize
        ops['sca_img'] = (f'Nc=6 map={fh}x{fw} Nq={Nq} B={bs}', fwd, bwd)
    return ops


def op_roofline(prof, ops, prof_ops=None):
    """Per sampling op and direction: duration = the library's op-level HIP-event scope, achieved =
    compulsory bytes / duration; plus the kernels the op consists of.  ``prof_ops``: a pass recorded with the
    op-level scopes ALONE (no event records between an op's launches) — the durations come from it, the
    per-kernel times from ``prof``."""
    out = []
    prof_ops = prof_ops or prof
    for op, (tag, fb, bb) in ops.items():
        for direction, scope, nbytes in (('fwd', 'bev_lift_fwd<', fb), ('bwd', 'bev_lift_bwd_op<', bb)):
            # (tag ends with the batch: the first layer's self-attention, computed once for the batch, is its own row)
            hit = [(k, r) for k, r in prof_ops.items() if k.startswith(scope) and k.endswith(tag)]
            if not hit:
                continue
            us = sum(r['avg_us'] for _, r in hit)
            launches = hit[0][1]['launches']
            kern = {k.split('<')[0]: round(r['avg_us'], 1) for k, r in prof.items()
                    if k.endswith(tag) and k.startswith('bev_lift_' + direction) and not k.startswith('bev_lift_bwd_op')}
            gbs = nbytes / (us * 1e-6) / 1e9
            out.append({'op': op, 'pass': direction, 'launches': launches, 'avg_us': us,
                        'compulsory_bytes_per_launch': float(nbytes), 'achieved_GBps': gbs,
                        'frac': gbs / HBM_PEAK_GBS, 'frac_of_copy_peak': gbs / HBM_COPY_GBS, 'kernels_us': kern})
    out.sort(key=lambda d: -d['avg_us'] * d['launches'])
    return out


def traffic_of(op_rec, dtype_name, workload, bs):
    """HBM bytes per launch of an op from the committed PMC passes (profiles/traffic.json, made by
    tools/make_traffic.py with the corrections of MI355X_MICROARCH.md section HBM; taken at bs = 2 on the
    256x704 shapes, bf16 and fp32), or None."""
    tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tfile) or bs != 2 or workload == 'LC_cat128':
        return None
    doc = json.load(open(tfile))
    ops = doc.get('ops', {}).get({'fp16': 'bf16'}.get(dtype_name, dtype_name), {})
    rec = ops.get(f"{op_rec['op']}:{op_rec['pass']}")
    op_rec['traffic_source'] = f"profiles/traffic.json (PMC passes taken at commit {doc.get('commit', 'unrecorded')})"
    return None if rec is None else rec.get('hbm_bytes_per_launch')


def run_mode(args, name, head, world, rank, device, want_ops):
    """Warm up, capture, time K steps of one precision mode.  Returns the mode's record."""
    with value_storage(name):
        return _run_mode(args, name, head, world, rank, device, want_ops)


def _run_mode(args, name, head, world, rank, device, want_ops):
    from unibev_amd import functional as UF
    from unibev_amd.graph_step import GraphedStep
    dtype = DTYPES[name]
    kw = WORKLOADS[args.workload][0]
    mods = kw.get('modalities', 'LC')
    img, pts, metas = synth_inputs(args.workload, args.bs, dtype, device, rank)
    head.transformer.lowp_stream = not args.fp32_stream
    params = [p for p in head.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    opt = None if args.flat_optimizer else torch.optim.AdamW(params, lr=args.lr, weight_decay=0.01, fused=True)
    s = 2 if kw.get('fusion_method') == 'cat' else 1
    C = kw['embed_dims']
    cot = torch.randn(200 * 200, args.bs, C * s, device=device) / 200.0
    split = args.exchange == 'split'
    tr = head.transformer
    encs = [getattr(tr, n) for n in ('img_bev_encoder', 'pts_bev_encoder') if getattr(tr, n, None) is not None]
    cut = encs if split else None
    gs = GraphedStep(head.transformer, lambda: head.forward_bev(img, pts, metas), cot, params,
                     inputs=(img or []) + (pts or []), has_img='C' in mods, has_pts='L' in mods,
                     autocast_dtype=None if dtype == torch.float32 else dtype, split_after=cut)
    params = gs.params                              # (a split backward puts the upper layers' parameters first)

    if args.flat_optimizer:
        # clip (max_norm 35) + AdamW as two streaming passes over the flat parameter / gradient / moment buffers
        from unibev_amd.optim import FlatAdamW
        opt = FlatAdamW(params, gs.grads, lr=args.lr, weight_decay=0.01, max_grad_norm=35.0)

        def finish():
            opt.step()
    else:
        def finish():
            torch.nn.utils.clip_grad_norm_(params, 35.0)
            opt.step()

    graphed = not args.no_graph
    for _ in range(args.warmup):
        gs.eager_step()
        finish()
    if graphed:
        try:
            gs.capture()
        except Exception as e:
            if not args.allow_eager:               # a silently eager number is not the configuration the line names
                raise RuntimeError('HIP graph capture failed; rerun with --allow-eager (or --no-graph) to time eager '
                                   'launches instead') from e
            print(f'[bench] HIP graph capture failed ({type(e).__name__}: {e}); eager launches (--allow-eager)',
                  file=sys.stderr)
            graphed = False
            gs.close()
    step = gs.step if graphed else gs.eager_step

    def barrier():
        from unibev_amd import dp as _dp
        _dp.barrier(device)                        # (async all-reduce: see dp.all_reduce)
        torch.cuda.synchronize()

    step()
    finish()                                       # first replay outside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        finish()
    host_dt = time.perf_counter() - t0             # host-side enqueue time (before the final sync)
    barrier()
    dt = time.perf_counter() - t0
    from unibev_amd import dp
    dt_local = dt
    dt = dp.max_over_ranks(dt, device)
    dt_min = dp.min_over_ranks(dt_local, device)
    # per-phase times of the same step (HIP events on the launch stream; 10 more steps, outside the timed region):
    # graph replay (forward + backward), gradient exchange (one RCCL all-reduce of the flat buffer; 0 for one rank
    # without a process group), clip + AdamW
    phases = None
    if graphed:
        marks = []
        for _ in range(min(args.steps, 10)):
            step(marks)
            finish()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
        torch.cuda.synchronize()
        n4 = len(marks) // 4
        ph = [sum(marks[4 * i + k].elapsed_time(marks[4 * i + k + 1]) for i in range(n4)) / n4 for k in range(3)]
        phases = {'fwd_bwd_graph_ms_per_step': ph[0], 'allreduce_ms_per_step': dp.max_over_ranks(ph[1], device),
                  'clip_adamw_ms_per_step': ph[2], 'gradient_bytes': int(gs.grads.flat.numel() * 4)}
    # pure launch cost: the same step enqueued on an IDLE device (host_dt above includes the time the host spends
    # blocked on the full launch queue once it is steps ahead of the GPU)
    launch_dt = 0.0
    for _ in range(min(args.steps, 10)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        finish()
        launch_dt += time.perf_counter() - t1
    torch.cuda.synchronize()
    launch_dt /= min(args.steps, 10)
    checksum = None
    if args.grad_checksum:                         # (the last step's exchanged gradients; the parameter order is the mode's own)
        f = gs.grads.flat.double()
        checksum = {'l2': float(f.square().sum().sqrt()), 'sum': float(f.sum()), 'abs': float(f.abs().sum()), 'n': int(f.numel())}
    nsteps_run = args.warmup + 1 + args.steps + min(args.steps, 10) * (2 if graphed else 1)       # (profilers divide by this)
    rec = {'dtype': name, 'value': world * args.bs * args.steps / dt, 'ms_per_step': 1e3 * dt / args.steps,
           'ms_per_step_rank_min': 1e3 * dt_min / args.steps, 'ms_per_step_rank_max': 1e3 * dt / args.steps,
           'phases': phases, 'grad_checksum': checksum,
           'host_enqueue_ms_per_step': 1e3 * launch_dt, 'host_loop_ms_per_step': 1e3 * host_dt / args.steps,
           'hip_graphs': graphed, 'steps_run': nsteps_run,
           'gradient_exchange': ('split: 2 HIP graphs, segment 0 (%d of %d bytes) all-reduced beside the second graph'
                                 % (gs.grads.segments[0].numel() * 4, gs.grads.flat.numel() * 4))
           if len(gs.grads.segments) == 2 else ('single message after the graph' + (f' ({gs.split_note})' if getattr(gs, 'split_note', None) else '')),
           'residual_stream': 'f32' if (args.fp32_stream or DTYPES[name] == torch.float32) else name,
           'value_storage': 'fp16' if name in VALUE_STORAGE else ('f32' if name == 'fp32' else name)}
    # ---- per-op roofline: the same step, eager, HIP events on the launch stream around every
    # sampling kernel / op (events cannot be read back from inside a captured graph)
    if want_ops and not args.no_kernel_timing:
        # (one stream: a kernel's duration is its own, not its share of a chip it splits with the other
        #  encoder's kernels — the timed steps above run the two encoders on two streams)
        from unibev_amd.modules import transformer as _tr
        two = _tr._TWO_STREAMS[0]
        _tr.set_two_streams(False)
        UF.set_seed_base(None)
        UF.kernel_profile(True, ops_only=True)          # pass 1: whole operators (their durations)
        for _ in range(min(args.steps, 10)):
            gs.eager_step()
        torch.cuda.synchronize()
        prof_ops = UF.kernel_profile()
        UF.kernel_profile(False)
        # how many sampling-point records did not fit their owner tile's fixed-capacity bucket and went through the
        # (exact, atomic) overflow list of the GRID backward: one more eager step with the counters read back
        UF.lift_overflow_probe(True)
        gs.eager_step()
        torch.cuda.synchronize()
        rec['grid_overflow'] = {f'map {fh}x{fw} P={P}': {'overflow_records': n, 'sampling_points': tot,
                                                        'fraction': n / max(tot, 1)}
                                for (fh, fw, P), (n, tot) in UF.lift_overflow_probe().items()}
        UF.lift_overflow_probe(False)
        UF.kernel_profile(True)                         # pass 2: every kernel inside them
        for _ in range(min(args.steps, 10)):
            gs.eager_step()
        torch.cuda.synchronize()
        prof = UF.kernel_profile()
        UF.kernel_profile(False)
        _tr.set_two_streams(two)
        pairs = args.bs * 40000
        if 'C' in mods:
            from unibev_amd.modules.encoders import pillar_axes, _lidar2img_tensor
            axes = pillar_axes(200, 200, 8, 4, device)
            _, _, vis0, _ = UF.point_sampling(_lidar2img_tensor(metas, device), *axes,
                                              head.transformer.img_bev_encoder.pc_range,
                                              WORKLOADS[args.workload][1])
            pairs = args.bs * int(vis0.sum().item())         # rows of sample 0's visibility (quirk q1) per sample
        ops = op_roofline(prof, sampling_ops(args.workload, args.bs, 4 if name == 'fp32' else 2, pairs), prof_ops)
        # (value-fp16: the op's operands are the fp16 value map, f32 offsets / logits, fp16 rows — priced at 2 bytes like the 16-bit modes)
        for o in ops:
            o['traffic'] = traffic_of(o, name, args.workload, args.bs)
        rec['roofline_ops'] = ops
        if ops:
            dom = ops[0]
            rec['roofline'] = {'bound': 'hbm', 'achieved': dom['achieved_GBps'], 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': dom['frac'], 'traffic': dom['traffic'],
                               'traffic_source': dom.get('traffic_source'),
                               'frac_of_copy_peak': dom['frac_of_copy_peak'], 'copy_peak': HBM_COPY_GBS,
                               'kernel': f"{dom['op']} {dom['pass']}: " + ' + '.join(dom['kernels_us']),
                               'avg_launch_us': dom['avg_us'],
                               'algorithmic_bytes_per_launch': dom['compulsory_bytes_per_launch']}
    gs.close()
    del gs, opt
    for p in params:
        p.grad = None
    torch.cuda.empty_cache()
    return rec


def gemm_record(device, bs):
    """The projection GEMMs of one encoder layer (M = bs x 40 000 rows) through the entry point the layers
    use (unibev_amd.linear.linear: the hand-written MFMA kernels — split-bf16 for f32 data — or hipBLASLt
    where that is faster), forward and weight gradient: time per call, TFLOP/s against the dense bf16 MFMA
    peak (f32 rows: 3 bf16 products per multiply are counted as ONE), operand GB/s."""
    from unibev_amd import functional as UF
    from unibev_amd.linear import linear as ubv_linear
    M, C = bs * 40000, 256
    shapes = [('value_proj / output_proj', C, C), ('offsets+logits (P=8)', 192, C),
              ('offsets+logits (P=4)', 96, C), ('ffn up', 2 * C, C), ('ffn down', C, 2 * C)]

    def clock(fn, n=20):
        """Device time per call: n calls captured in one HIP graph (an eager loop is host-bound below ~25 us
        per call), replayed and timed with events."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        from unibev_amd import dp as _dp
        _dp.drain_watchdog()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode='thread_local'):
            for _ in range(n):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    out = []
    for dname, dt in (('bf16', torch.bfloat16), ('fp32', torch.float32)):
        for label, N, K in shapes:
            x = torch.randn(M, K, device=device, dtype=dt)
            w = torch.nn.Parameter(torch.randn(N, K, device=device) / K ** 0.5)   # f32 master weights, as in the model
            b = torch.nn.Parameter(torch.zeros(N, device=device))
            gy = torch.randn(M, N, device=device, dtype=dt)
            from unibev_amd.linear import lowp_step_cache
            # (inside the step cache: the 16-bit / split copies of the weights are made once per step, not per call)
            with torch.no_grad(), torch.autocast('cuda', dtype=dt, enabled=dt != torch.float32), lowp_step_cache():
                us = clock(lambda: ubv_linear(x, w, b))
            us_w = clock(lambda: UF.gemm_wgrad(gy, x))
            flops = 2.0 * M * N * K
            nb_f, nb_w = (M * K + M * N) * x.element_size(), (M * K + M * N) * x.element_size()
            out.append({'gemm': label, 'dtype': dname, 'M': M, 'N': N, 'K': K, 'us': us,
                        'TFLOPs': flops / us / 1e6, 'mfma_frac': flops / us / 1e6 / MFMA_PEAK_TFLOPS['bf16'],
                        'GBps': nb_f / us / 1e3, 'hbm_frac': nb_f / us / 1e3 / HBM_PEAK_GBS,
                        'wgrad_us': us_w, 'wgrad_GBps': nb_w / us_w / 1e3,
                        'wgrad_hbm_frac': nb_w / us_w / 1e3 / HBM_PEAK_GBS})
    return out


def k1_record(device, bs):
    """The deformable-sampling OPERATOR at the mmcv boundary (``MultiScaleDeformableAttnFunction``: explicit locations and
    weights, what a maintainer binding at INTEGRATION.md level 2 calls) on the self-attention instance (S = Nq = 40 000,
    H = 8, Dh = 32, P = 4, f32): forward, backward on the owner-tile plan, backward with grad_value atomics."""
    from unibev_amd import functional as UF
    from unibev_amd.modules.deform_attn import index_tensor, shapes_tensor
    B, Hq, Wq, H, Dh, P = bs, 200, 200, 8, 32, 4
    S = Nq = Hq * Wq
    g = torch.Generator(device='cpu').manual_seed(3)
    ys, xs = torch.meshgrid(torch.arange(Hq), torch.arange(Wq), indexing='ij')
    ref = torch.stack(((xs + 0.5) / Wq, (ys + 0.5) / Hq), -1).view(1, Nq, 1, 1, 1, 2)
    loc = (ref + 0.015 * torch.randn(B, Nq, H, 1, P, 2, generator=g)).to(device).requires_grad_()
    aw = torch.softmax(torch.randn(B, Nq, H, 1, P, generator=g), -1).to(device).requires_grad_()
    v = torch.randn(B, S, H, Dh, generator=g).to(device).requires_grad_()
    go = torch.randn(B, Nq, H * Dh, generator=g).to(device)
    ss, ls = shapes_tensor([(Hq, Wq)], device), index_tensor([0], device)
    plain = torch.as_tensor([[Hq, Wq]], dtype=torch.long, device=device)      # no host copy attached: atomic kernel

    def timed(fn, n=10):
        """Device time per call: n calls captured in one HIP graph and replayed (the eager python loop of an autograd
        Function is host-bound below ~100 us per call: rounds 3 - 4 reported that as the operator's forward)."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        from unibev_amd import dp as _dp
        _dp.drain_watchdog()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode='thread_local'):
            for _ in range(n):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    def fwd_bwd(shapes):
        v.grad = loc.grad = aw.grad = None
        UF.ms_deform_attn(v, shapes, ls, loc, aw).backward(go)

    sg = shapes_tensor([(Hq, Wq)], device, query_grid=(Hq, Wq))      # + the query-grid hint: the TILE plan
    with torch.no_grad():
        f_us = timed(lambda: UF.ms_deform_attn(v, ss, ls, loc, aw))
        fg_us = timed(lambda: UF.ms_deform_attn(v, sg, ls, loc, aw))
    fb_planned, fb_atomic, fb_grid = timed(lambda: fwd_bwd(ss)), timed(lambda: fwd_bwd(plain)), timed(lambda: fwd_bwd(sg))
    fb, bb = k1_bytes(B, S, Nq, H * Dh, H, P, 4, False), k1_bytes(B, S, Nq, H * Dh, H, P, 4, True)
    return {'shape': f'B={B} S=Nq={S} H={H} Dh={Dh} P={P} f32', 'fwd_us': f_us, 'fwd_frac': fb / f_us / 1e3 / HBM_PEAK_GBS,
            'bwd_planned_us': fb_planned - f_us, 'bwd_planned_frac': bb / (fb_planned - f_us) / 1e3 / HBM_PEAK_GBS,
            'bwd_atomic_us': fb_atomic - f_us, 'bwd_atomic_frac': bb / (fb_atomic - f_us) / 1e3 / HBM_PEAK_GBS,
            'grid_fwd_us': fg_us, 'grid_fwd_frac': fb / fg_us / 1e3 / HBM_PEAK_GBS,
            'grid_bwd_us': fb_grid - fg_us, 'grid_bwd_frac': bb / (fb_grid - fg_us) / 1e3 / HBM_PEAK_GBS,
            'note': 'backward = (forward + backward) - forward, device time of graph replays; bytes: SURVEY.md section 8(d) k1 formulas; '
                    'grid_*: the same operator with the query-grid hint (ubv_ms_deform_attn_forward_grid / _backward_grid)'}


def voxel_record(device):
    """LiDAR front end at cfg3's size: hard voxelization (T = 10, 90 000 voxel budget) of a 30 000
    point cloud + VFE mean + the dense scatter of a SparseEncoder-sized output; nothing read back."""
    from unibev_amd import functional as UF
    from unibev_amd import synthetic as syn
    pts = torch.from_numpy(syn.lidar_points(30000, seed=0)).to(device)
    N, F = pts.shape

    def front():
        voxels, coors, num, vnum = UF.hard_voxelize(pts, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000)
        return UF.voxel_mean(voxels, num, vnum), coors, vnum

    for _ in range(3):
        mean, coors, vnum = front()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mean, coors, vnum = front()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    m = int(vnum.item())
    nbytes = N * F * 4 + N * 3 * 4 + m * (F + 3 + 1) * 4
    # a batch of clouds in ONE launch chain (ubv_hard_voxelize_batch, what UniBEV.voxelize runs): per-cloud time at
    # the per-GPU batch of the reference's configs (1, shipped) and of BASELINE's cfg4 (2), and at 8
    batch = {}
    for nb in (2, 8):
        clouds = [torch.from_numpy(syn.lidar_points(30000, seed=s)).to(device) for s in range(nb)]

        def front_b():             # (the VFE mean rides in the chain's gather launch: what extract_pts_feat runs)
            return UF.hard_voxelize_batch(clouds, syn.VOXEL_SIZE, syn.PC_RANGE, 10, 90000, with_mean=True)[4]
        for _ in range(3):
            front_b()
        e0.record()
        for _ in range(10):
            front_b()
        e1.record()
        torch.cuda.synchronize()
        batch[f'us_per_cloud_batch{nb}'] = 1e3 * e0.elapsed_time(e1) / 10 / nb
    rec = {'points': N, 'voxels': m, 'us_per_cloud': us, 'points_per_s': N / us * 1e6,
           'algorithmic_bytes': nbytes, 'GBps': nbytes / us / 1e3, **batch,
           'note': 'hard voxelize + VFE mean; one cloud per chain (us_per_cloud: ubv_hard_voxelize + ubv_voxel_mean, '
                   '6 launches) is latency-bound; a batch shares one 5-launch chain with the mean written by its gather '
                   '(us_per_cloud_batchN: ubv_hard_voxelize_batch_vfe)'}
    rec['middle_encoder'] = middle_encoder_record(device, mean[:m], coors[:m])
    return rec


def middle_encoder_record(device, feats, coors, bs=2):
    """SparseEncoder of the shipped L / LC configs (41 x 1440 x 1440 grid, basic blocks) on `bs` copies of the
    cloud: forward and forward + backward (training mode), per batch.  ``forward_ms`` / ``forward_backward_ms``
    REBUILD the rulebooks (the strided layers' output sites — one host read of the four counts —, hash tables,
    neighbour maps, compacted pairs) on every pass, as a training step with new clouds does; the ``*_kept_rulebooks`` numbers are the same
    passes with the module's opt-in cache (``keep_rulebooks``: gradient accumulation / checkpointing of one cloud)."""
    from unibev_amd.registry import MIDDLE_ENCODERS, build_from_cfg
    cfg = dict(type='SparseEncoder', in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
               order=('conv', 'norm', 'act'),
               encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
               encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
    torch.manual_seed(0)
    enc = build_from_cfg(cfg, MIDDLE_ENCODERS).to(device).train()
    f = torch.cat([feats] * bs).float().contiguous()
    zyx = coors[:, -3:]                                     # ubv_hard_voxelize: (z, y, x) rows
    c = torch.cat([torch.cat((torch.full_like(zyx[:, :1], b), zyx), 1) for b in range(bs)]).contiguous()

    def fwd():
        with torch.no_grad():
            return enc(f, c, bs)

    def fwd_bwd():
        for p in enc.parameters():
            p.grad = None
        enc(f, c, bs).sum().backward()

    out = {}
    for keep in (False, True):
        enc.keep_rulebooks = keep
        for name, fn in (('forward_ms', fwd), ('forward_backward_ms', fwd_bwd)):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[name + ('_kept_rulebooks' if keep else '')] = e0.elapsed_time(e1) / 5
    out.update(batch=bs, voxels_per_sample=int(feats.shape[0]), out_shape=list(fwd().shape),
               note='SubM / strided sparse convs as gather + MFMA over neighbour maps; weight gradients on the split-K MFMA '
                    'kernel over compacted pairs; forward_ms / forward_backward_ms rebuild the rulebooks every pass on the '
                    'device (ONE host read: the four strided layers\' output counts), *_kept_rulebooks reuse them '
                    '(opt-in cache)')
    return out


def _two_streams():
    from unibev_amd.modules import transformer as _tr
    return bool(_tr._TWO_STREAMS[0])


def self_launch(args):
    """``python bench.py --gpus N`` without a torchrun environment: run the same command line as N ranks of ONE node
    under ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1 at a free port) and pass rank 0's
    JSON line through.  The reference launches its data-parallel job the same way (tools/dist_train.sh ->
    torch.distributed.launch, tools/train_UniBEV.py:242-249)."""
    import socket
    import subprocess
    if not args.dry_run:
        assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
        have = torch.cuda.device_count()
        if have < args.gpus and not share_gpu():
            raise SystemExit(f'--gpus {args.gpus} but only {have} GPU(s) are visible')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               UBV_BENCH_CHILD='1')
    if args.gpus == 1:
        env['UBV_FORCE_DDP'] = '1'                   # one rank WITH a process group: the N > 1 code path
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def share_gpu():
    """UBV_SHARE_GPU=1 (test hook): ranks beyond the visible GPUs wrap around — N processes on one device."""
    return os.environ.get('UBV_SHARE_GPU', '0') == '1'


def allreduce_algo(args):
    """'ring' | 'default' for dp.init_distributed: the overlapped exchange asks for the ring (see --allreduce-algo)."""
    if args.allreduce_algo != 'auto':
        return args.allreduce_algo
    return 'ring' if args.exchange == 'split' else 'default'


def dry_run(args, rank, world, local):
    """The N-rank protocol of this script without its kernels: process group from the launcher's environment (RCCL when
    GPUs are visible, gloo on CPU), identical replicas of a small registry-built module, per-rank synthetic gradients,
    the flat-buffer exchange in the selected mode (`single`: one averaged message; `split`: two segments, the first in
    flight while the second is collected), barrier + max-over-ranks timing, and rank 0's single JSON line.  The averaged
    gradients are checked against their closed form on every rank; a mismatch exits non-zero.  What a first multi-GPU
    run can then still get wrong is RCCL / HIP-graph specific, not the launcher, the rank environment or the exchange
    bookkeeping (tests/test_bench_dryrun.py drives this with world_size 2 on gloo)."""
    from unibev_amd import dp
    from unibev_amd.registry import build_feedforward_network
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device('cuda', local) if use_gpu else torch.device('cpu')
    dp.init_distributed('nccl' if use_gpu else 'gloo', device if use_gpu else None, algo=allreduce_algo(args))
    torch.manual_seed(0)                                          # identical replicas
    net = torch.nn.Sequential(build_feedforward_network(dict(type='FFN', embed_dims=32, feedforward_channels=64, ffn_drop=0.0)),
                              torch.nn.LayerNorm(32)).to(device)
    params = list(net.parameters())
    split = args.exchange == 'split'
    fg = dp.FlatGradients(params, first_segment=2 if split else None)
    fg.attach()

    def step(k):
        grads = [torch.full_like(p, float((rank + 1) * (k + 1))) for p in params]     # what a backward would leave
        if split:
            fg.collect(0, 2, grads[:2])
            fg.start_segment(0)
            fg.collect(2, len(params), grads[2:])
            fg.start_segment(1)
            fg.finish_segments()
        else:
            fg.collect(grads=grads)
            fg.all_reduce_mean()

    for k in range(args.warmup):
        step(k)
    dp.barrier(device if use_gpu else None)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    dp.barrier(device if use_gpu else None)
    dt = dp.max_over_ranks(time.perf_counter() - t0, device)
    want = (args.steps) * (world + 1) / 2.0 if args.steps else 0.0              # mean over ranks of (rank + 1) * steps
    ok = args.steps == 0 or bool(torch.allclose(fg.flat, torch.full_like(fg.flat, want)))
    ok_all = dp.min_over_ranks(1.0 if ok else 0.0, device) == 1.0
    if rank == 0:
        line = {'metric': 'nuScenes samples/sec BEV-encoder fwd+bwd', 'value': None, 'unit': 'samples/s', 'n_gpus': world,
                'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / max(args.steps, 1),
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'fp32', 'data': 'synthetic',
                'dry_run': True, 'exchange_ok': ok_all,
                'config': {'workload': 'dry run: launcher + exchange protocol on a 32-channel FFN + LayerNorm, no kernels',
                           'per_gpu_batch': args.bs, 'global_batch': world * args.bs, 'parallelism': f'dp{world}',
                           'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                           'collective': dp.collective_info(),
                           'gradient_exchange': 'split' if split else 'single',
                           'launcher': 'self (bench.py -> torch.distributed.run)' if os.environ.get('UBV_BENCH_CHILD') == '1'
                                       else ('torchrun' if 'LOCAL_RANK' in os.environ else 'single process')}}
        print(json.dumps(line, allow_nan=False), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if not ok_all:
        raise SystemExit('dry run: the exchanged gradients differ from their closed form')


LINE_LIMIT = 6000          # bytes of the final stdout line (the driver keeps a 9 KB tail of stdout: VERDICT r4 item 1)


def _r(x, nd=5):
    """Round floats (recursively) so that the short line stays short; non-finite values become None (strict JSON)."""
    if isinstance(x, float):
        return float(f'{x:.{nd}g}') if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _ops_short(ops):
    """roofline_ops as rows [op, pass, us, frac, traffic / algorithmic bytes] (None where no PMC pass covers the op)."""
    rows = []
    for o in ops or []:
        tr = o.get('traffic')
        rows.append([o['op'].split(' (')[0] + ('/shared' if '(' in o['op'] else ''), o['pass'], round(o['avg_us'], 1),
                     round(o['frac'], 4), None if not tr else round(tr / o['compulsory_bytes_per_launch'], 2)])
    return rows


def compact(full):
    """The ONE line the driver parses: the contract fields, `roofline` (dominant sampling op), `cpu_baseline`, `phases`,
    every sampling op as a 5-number row, and one number per sub-record.  The long form (`full`) goes to the extras
    file and to a prefixed stderr line."""
    cfg = full['config']
    par = cfg.get('parity') or {}
    short_cfg = {'workload': cfg['workload'].split(':')[0], 'shapes': cfg['workload'].split(': ', 1)[-1],
                 'per_gpu_batch': cfg['per_gpu_batch'], 'global_batch': cfg['global_batch'],
                 'mode': cfg['mode'].split(' (')[0], 'parallelism': cfg['parallelism'], 'rccl_ranks': cfg['rccl_ranks'],
                 'gemm_arithmetic': ('f32 storage, Linear = split-bf16 x3 MFMA, f32 accumulate' if full['dtype'] == 'fp32'
                                     else cfg['gemm_arithmetic']),
                 'collective': cfg.get('collective'),
                 'sampling_params': cfg['sampling_params'].split(':')[0],
                 'streams': 2 if cfg['streams'].startswith('image') else 1,
                 'step': cfg['step'], 'gradient_exchange': cfg['gradient_exchange'].split(':')[0].split(' (')[0],
                 'launcher': cfg['launcher'].split(' (')[0],
                 'parity': {'bar': par.get('bar'), 'pass': par.get('pass'), 'distance': par.get('distance')} if par else None}
    out = {k: full[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                'scaling', 'vs_baseline', 'dtype', 'data')}
    out['config'] = short_cfg
    rf = full.get('roofline')
    if rf:
        out['roofline'] = {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel',
                                                  'avg_launch_us', 'algorithmic_bytes_per_launch')}
    else:
        out['roofline'] = None
    out['roofline_ops_cols'] = ['op', 'pass', 'us', 'frac_of_8TBps', 'pmc_traffic_over_algorithmic']
    out['roofline_ops'] = _ops_short(full.get('roofline_ops'))
    out['phases'] = full.get('phases')
    if full.get('grad_checksum'):
        out['grad_checksum'] = full['grad_checksum']
    out['ms_per_step_rank_min'] = full['ms_per_step_rank_min']
    out['host_enqueue_ms_per_step'] = full['host_enqueue_ms_per_step']
    if 'spread' in full:
        out['spread_value'] = full['spread']['value']
        out['spread_roofline_ops'] = _ops_short(full['spread'].get('roofline_ops'))
    if 'ieee_gemm' in full:
        ip = full['ieee_gemm'].get('parity') or {}
        out['ieee_gemm_value'] = full['ieee_gemm']['value']
        out['ieee_gemm_parity_worst'] = max(ip['distance'].values()) if ip.get('distance') else None
    if 'lowp' in full:
        out['lowp'] = {r['dtype']: {'value': r['value'], 'parity_pass': (r.get('parity') or {}).get('pass'),
                                    'parity_worst': max(r['parity']['distance'].values()) if r.get('parity') else None,
                                    'parity_distance': (r.get('parity') or {}).get('distance')}
                       for r in full['lowp']}
    if 'gemm' in full:
        g = {(x['dtype'], x['N'], x['K']): x for x in full['gemm']}
        x = g.get(('fp32', 256, 256))
        if x:
            out['gemm_256x256_f32'] = {'us': x['us'], 'hbm_frac': x['hbm_frac'], 'mfma_frac': x['mfma_frac'], 'wgrad_us': x['wgrad_us']}
    if 'voxel' in full:
        v = full['voxel']
        out['voxel'] = {'us_per_cloud_batch2': v.get('us_per_cloud_batch2'), 'points_per_s': v['points_per_s'],
                        'middle_encoder_fwd_bwd_ms': v['middle_encoder']['forward_backward_ms'],
                        'middle_encoder_fwd_ms': v['middle_encoder']['forward_ms']}
    if 'k1_operator' in full:
        k = full['k1_operator']
        out['k1_operator'] = {kk: k[kk] for kk in ('fwd_us', 'fwd_frac', 'bwd_planned_us', 'bwd_planned_frac', 'grid_fwd_us',
                                                   'grid_fwd_frac', 'grid_bwd_us', 'grid_bwd_frac') if kk in k}
    if 'cpu_baseline' in full:
        c = full['cpu_baseline']
        out['cpu_baseline'] = {k: c[k] for k in ('value', 'unit', 'cores', 'kind', 'sample')}
        out['cpu_baseline']['host_cores'] = c.get('host_cores')
        # samples/s at other thread counts (all host threads = BASELINE.md section 3's os.cpu_count()); None: timed out
        out['cpu_baseline']['threads_scan'] = {n: v.get('samples_per_s') for n, v in (c.get('threads_scan') or {}).items()}
    if 'cpu_baseline_plan' in full:
        out['cpu_baseline_plan'] = {e['config'].split()[0] + ':' + e['pass']: e['samples_per_s']
                                    for e in full['cpu_baseline_plan']['entries']}
        out['cpu_baseline_plan']['threads'] = full['cpu_baseline_plan']['cores']
    out['extras'] = 'bench_extras.json (long form of every record; also the `#extras ` stderr line)'
    out = _r(out)
    if full.get('grad_checksum'):                 # (compared between runs to 1e-5: not rounded)
        out['grad_checksum'] = full['grad_checksum']
    return out


def emit(full, extras_file):
    """Write the long record to ``extras_file`` and to a prefixed STDERR line, then print the short line, the only line
    on stdout.  The short
    line is strict JSON (no NaN / Infinity) and bounded: if a future field pushes it over LINE_LIMIT the optional
    summaries are dropped one by one instead of letting the contract fields fall out of the driver's tail."""
    long_line = json.dumps(_r(full, 7), allow_nan=False)
    if extras_file:
        try:
            with open(extras_file, 'w') as f:
                f.write(long_line + '\n')
        except OSError as e:
            print(f'[bench] could not write {extras_file}: {e}', file=sys.stderr)
    print('#extras ' + long_line, file=sys.stderr, flush=True)     # (stdout carries exactly one line)
    short = compact(full)
    line = json.dumps(short, allow_nan=False)
    for k in ('cpu_baseline_plan', 'spread_roofline_ops', 'k1_operator', 'voxel', 'gemm_256x256_f32', 'lowp', 'roofline_ops'):
        if len(line) <= LINE_LIMIT:
            break
        short.pop(k, None)
        short['dropped_for_length'] = short.get('dropped_for_length', []) + [k]
        line = json.dumps(short, allow_nan=False)
    assert len(line) <= LINE_LIMIT, len(line)
    json.loads(line)
    # C-level stdout first (RCCL prints its version banner through printf: unflushed, it would land AFTER the line — at
    # process exit — and the last line of stdout would not be the JSON)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(line, flush=True)


def main():
    args = parse()
    if args.cpu_baseline_child > 0:
        return cpu_baseline_child(args)
    torchrun = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if not torchrun and (args.launcher == 'spawn' or (args.launcher == 'auto' and args.gpus > 1)):
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'python -m torch.distributed.run --nproc-per-node {args.gpus} ... or drop the torchrun '
                         f'environment and let bench.py launch its own ranks')
    if args.dry_run:
        return dry_run(args, rank, world, local)
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    if share_gpu():
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # everything runs on one non-default stream: HIP-graph capture needs the gradient accumulation of
    # every parameter pinned to the capturing stream (graph_step.GraphedStep.capture)
    torch.cuda.set_stream(torch.cuda.Stream(device))
    if args.single_stream:
        from unibev_amd.modules import transformer as _tr
        _tr.set_two_streams(False)
    from unibev_amd import dp
    # RCCL over xGMI (no-op for N = 1).  TEST HOOK (tests/test_bench_gpu.py): UBV_DIST_BACKEND=gloo + UBV_SHARE_GPU=1 run the
    # N-rank path of this script — launcher, identical replicas, HIP graphs, the flat-gradient exchange between replays,
    # max-over-ranks timing, rank 0's line — with N processes on ONE GPU (RCCL refuses two ranks per device; gloo moves
    # the gradient buffer through the host).  Its samples/s mean nothing; the line says `collective.backend: gloo`.
    backend = os.environ.get('UBV_DIST_BACKEND', 'nccl')
    dp.init_distributed(backend, device if backend == 'nccl' else None, algo=allreduce_algo(args))

    torch.manual_seed(0)          # identical replicas
    np.random.seed(rank)          # modality dropout is per process, as in the reference
    head, tcfg = build_head(args.workload, device)
    head.train(not args.eval_mode)
    names = ['fp32', 'value-fp16', 'bf16', 'fp16'] if args.dtype == 'all' else [args.dtype]
    set_sampling_params(head, 'spread' if args.params == 'spread' else 'init')
    recs = [run_mode(args, n, head, world, rank, device, want_ops=True) for n in names]
    spread = ieee = None
    if args.params == 'both':
        # second operating point of the headline precision: offsets scattered per query (trained-looking)
        set_sampling_params(head, 'spread')
        spread = run_mode(args, names[0], head, world, rank, device, want_ops=True)
        spread['sampling_params'] = f'spread: seeded sampling_offsets weights, ~{SPREAD_SIGMA_PX:g} px offset scatter per query'
        set_sampling_params(head, 'init')
    if names[0] == 'fp32' and not args.no_ieee_gemm:
        # the same f32 step with every Linear on the library's IEEE f32 GEMM instead of the split-bf16 MFMA kernels
        from unibev_amd.linear import set_f32_gemm
        prev = set_f32_gemm('library')
        try:
            ieee = run_mode(args, 'fp32', head, world, rank, device, want_ops=False)
            if rank == 0 and not args.no_parity:
                ieee['parity'] = parity_record(args.workload, ['fp32'], device, args.fp32_stream)['fp32']
        finally:
            set_f32_gemm(prev)
        ieee['gemm_arithmetic'] = 'IEEE f32 library GEMMs (hipBLASLt)'
    if rank == 0 and not args.no_parity:
        par = parity_record(args.workload, names, device, args.fp32_stream)
        for r in recs:
            r['parity'] = par[r['dtype']]

    if rank == 0:
        main_rec = recs[0]
        full = {
            'metric': 'nuScenes samples/sec BEV-encoder fwd+bwd', 'value': main_rec['value'],
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': main_rec['ms_per_step'],
            'ms_per_step_rank_min': main_rec['ms_per_step_rank_min'], 'ms_per_step_rank_max': main_rec['ms_per_step_rank_max'],
            'phases': main_rec['phases'], 'grad_checksum': main_rec.get('grad_checksum'),
            'host_enqueue_ms_per_step': main_rec['host_enqueue_ms_per_step'],
            'host_loop_ms_per_step': main_rec['host_loop_ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': main_rec['dtype'], 'data': 'synthetic',
            'config': {'workload': WORKLOADS[args.workload][3], 'per_gpu_batch': args.bs,
                       'global_batch': world * args.bs, 'encoder_layers': 3,
                       'mode': 'eval' if args.eval_mode else 'train (dropout 0.1, modality dropout)',
                       'residual_stream': main_rec['residual_stream'],
                       'gemm_arithmetic': ('f32 storage; every Linear is a split-bf16 x3 MFMA product (x_hi w_hi + x_hi w_lo + '
                                           'x_lo w_hi, ~2^-17 per product) with f32 accumulation; sampling kernels plain f32; '
                                           'the IEEE-GEMM run of the same step rides along as `ieee_gemm`')
                       if main_rec['dtype'] == 'fp32' else
                       ('as fp32 (split-bf16 x3 MFMA Linear layers, f32 stream); projected value maps and sampled outputs '
                        'stored in fp16' if main_rec['dtype'] in VALUE_STORAGE else 'autocast ' + main_rec['dtype']),
                       'sampling_params': ('spread' if args.params == 'spread' else
                                           "init: the reference's init_weights (zero offset weights, compass-grid bias); "
                                           "`spread` sub-record: scattered offsets"),
                       'first_layer_self_attention': ('one query table for the batch: computed once per step and shared by '
                                                      'the samples (same values as per-sample; UBV_SHARE_FIRST=0 '
                                                      'computes it per sample)')
                       if os.environ.get('UBV_SHARE_FIRST', '1') != '0' else 'per sample',
                       'step': 'fwd + bwd (HIP graphs) + flat-gradient all-reduce + clip + AdamW'
                               if main_rec['hip_graphs'] else 'fwd + bwd + flat-gradient all-reduce + clip + AdamW',
                       'gradient_exchange': main_rec['gradient_exchange'],
                       'optimizer': 'flat-buffer clip + AdamW kernels' if args.flat_optimizer else 'torch clip_grad_norm_ + fused AdamW',
                       'streams': 'image / point-cloud encoders on 2 HIP streams' if _two_streams() else '1 stream',
                       'launcher': 'self (bench.py -> torch.distributed.run)' if os.environ.get('UBV_BENCH_CHILD') == '1'
                                   else ('torchrun' if 'TORCHELASTIC_RUN_ID' in os.environ or 'LOCAL_RANK' in os.environ
                                         else 'single process'),
                       'parallelism': f'dp{world}', 'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                       'collective': dp.collective_info(),
                       'parity': main_rec.get('parity')},
            'roofline': main_rec.get('roofline'),
            'roofline_ops': main_rec.get('roofline_ops'),
            'grid_overflow': main_rec.get('grid_overflow'),
        }
        if spread is not None:
            full['spread'] = spread
        if ieee is not None:
            full['ieee_gemm'] = ieee
        if len(recs) > 1:
            full['lowp'] = recs[1:]
        if world == 1 and not args.no_extras:
            full['gemm'] = gemm_record(device, args.bs)
            full['voxel'] = voxel_record(device)
            full['k1_operator'] = k1_record(device, args.bs)
        # ---- CPU baseline: the oracle's forward on this host ------------------------------
        if world == 1 and not args.no_cpu_baseline:
            full['cpu_baseline'] = cpu_baseline(args, tcfg, head)
        if world == 1 and not args.no_cpu_baseline and not args.no_cpu_baseline_plan:
            # the default run carries the plan bounded to about a minute; --cpu-baseline-plan: the fuller protocol
            full['cpu_baseline_plan'] = (cpu_baseline_plan() if args.cpu_baseline_plan else
                                         cpu_baseline_plan(max_seconds=4.0, max_passes=2))
        emit(full, args.extras_file)
    if dist.is_initialized():
        dist.destroy_process_group()


def _cpu_pass_fn(args, tcfg, head):
    """The oracle's forward of the bench workload (bs = 1, fp32, eval) as a closure over CPU tensors."""
    from oracle import unibev_ref as R
    sd = {k[len('transformer.'):]: v.detach().float().cpu() for k, v in head.state_dict().items()
          if k.startswith('transformer.')}
    img, pts, metas = synth_inputs(args.workload, 1, torch.float32, 'cpu', 0)
    metas = [dict(lidar2img=[m.numpy() for m in metas[0]['lidar2img']], img_shape=metas[0]['img_shape'])]
    bev_q = head.state_dict()['bev_embedding.weight'].float().cpu()
    pos = R.learned_positional_encoding(head.state_dict()['positional_encoding.row_embed.weight'].float().cpu(),
                                        head.state_dict()['positional_encoding.col_embed.weight'].float().cpu(),
                                        1, 200, 200)
    cfg = json.loads(json.dumps(tcfg))

    def run():
        with torch.no_grad():
            return R.transformer_encode_fuse(sd, cfg, None if img is None else [x.detach() for x in img],
                                             None if pts is None else [x.detach() for x in pts],
                                             bev_q, 200, 200, pos, metas)
    return run


def cpu_baseline_child(args):
    """``--cpu-baseline-child T`` (run by cpu_baseline in a child process it can kill): one warm-up + one timed pass of
    the oracle's forward with T torch threads; prints the seconds of the timed pass."""
    torch.set_num_threads(args.cpu_baseline_child)
    torch.manual_seed(0)
    head, tcfg = build_head(args.workload, 'cpu')
    head.eval()
    run = _cpu_pass_fn(args, tcfg, head)
    run()
    t0 = time.perf_counter()
    run()
    print(f'cpu_pass_seconds {time.perf_counter() - t0:.4f}', flush=True)


def cpu_thread_scan(args, counts, timeout_s):
    """Seconds per pass at other thread counts, each in a child process that is killed after ``timeout_s`` (an
    oversubscribed host does not come back from one pass in bounded time: 78.9 s measured with 256 threads)."""
    import subprocess
    out = {}
    for n in counts:
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', str(n), '--workload', args.workload]
        env = dict(os.environ, OMP_NUM_THREADS=str(n), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
            sec = [float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith('cpu_pass_seconds')]
            out[str(n)] = ({'seconds_per_pass': sec[0], 'samples_per_s': 1.0 / sec[0]} if sec else
                           {'error': (r.stderr or r.stdout).strip().splitlines()[-1][:160] if (r.stderr or r.stdout).strip() else f'rc {r.returncode}'})
        except subprocess.TimeoutExpired:
            out[str(n)] = {'timed_out_after_s': round(time.perf_counter() - t0, 1),
                           'note': 'model build + warm-up + one pass did not finish: slower than the 16-thread figure'}
    return out


def cpu_baseline(args, tcfg, head):
    """The oracle (CPU restatement of the reference path, oracle/unibev_ref.py) timed on the host:
    forward only, fp32, bs = 1, eval mode; 2 warm-up + up to 6 timed passes, median, bounded to ~20 s of timed work
    (fewer passes on a slow host, stated).  The headline figure uses 16 torch threads — all 256 hardware threads of the
    GPU host oversubscribe torch's small CPU kernels — and ``threads_scan`` reports 64 threads and ALL host threads
    (BASELINE.md section 3's ``os.cpu_count()``) beside it, each bounded by a time-out."""
    # a 256-thread host oversubscribes torch's small CPU kernels (78.9 s/pass measured with all
    # threads vs ~3 s with 16): the baseline uses at most 16 threads and says so
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    run = _cpu_pass_fn(args, tcfg, head)
    for _ in range(2):
        run()
    times, spent = [], 0.0
    while len(times) < 6 and spent < 20.0:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
        spent += times[-1]
    med = float(np.median(times))
    import platform
    cpu = platform.processor() or ''
    try:
        cpu = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except Exception:
        pass
    ncpu = os.cpu_count() or 1
    scan = {}
    if ncpu > 16 and os.environ.get('UBV_CPU_SCAN', '1') != '0':
        scan = cpu_thread_scan(args, sorted({n for n in (64, ncpu) if 16 < n <= ncpu}), 30.0)
    best = max([1.0 / med] + [v['samples_per_s'] for v in scan.values() if 'samples_per_s' in v])
    return {'value': 1.0 / med, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu': cpu, 'host_cores': ncpu, 'threads_scan': scan, 'best_samples_per_s': best,
            'sample': f'forward only, fp32, bs=1, eval: 2 warm-up + {len(times)} timed passes of the same '
                      f'workload, median {med:.2f} s, through oracle/unibev_ref.py (torch CPU)'}


def cpu_baseline_plan(threads=16, max_seconds=40.0, max_passes=5):
    """the oracle's forward through ``fused_bev_embed`` for cfg1-cfg5 of BASELINE.json.configs
    (bs = 1, fp32, eval) and forward + backward for cfg2 / cfg3; 1 warm-up + up to 5 timed passes per entry (median),
    each entry bounded to ``max_seconds``.  ``threads`` torch CPU threads: all 256 hardware threads of the GPU host
    oversubscribe torch's small kernels (78.9 s per pass measured against ~3 s with 16) — the count used is reported."""
    import platform
    from oracle import unibev_ref as R
    from unibev_amd import configs as cfgs
    from unibev_amd import synthetic as syn
    torch.set_num_threads(min(os.cpu_count() or 1, threads))
    plan = [('cfg1 unibev_nus_C 2 views (plumbing)', dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='C', num_cams=2), (256, 704), None, 2, False),
            ('cfg2 unibev_nus_C 6x256x704', dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='C'), (256, 704), None, 6, True),
            ('cfg3 unibev_nus_L', dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='L'), (256, 704), (180, 180), 6, True),
            ('cfg4 unibev_nus_LC_cnw_256', dict(embed_dims=256, fusion_method='linear', feature_norm='ChannelNormWeights', drop_modality=0.5), (256, 704), (180, 180), 6, False),
            ('cfg5 unibev_nus_LC_cat_128 6x800x1440', dict(embed_dims=128, fusion_method='cat', feature_norm=None, drop_modality=0.5), (800, 1440), (180, 180), 6, False)]
    out = []
    for name, kw, img_hw, pts_hw, ncam, with_bwd in plan:
        C = kw['embed_dims']
        tcfg = cfgs.transformer_cfg(decoder=None, **kw)
        tcfg.pop('decoder')
        from unibev_amd import build_transformer
        torch.manual_seed(0)
        model = build_transformer(json.loads(json.dumps(tcfg)))
        model.init_weights()
        sd = {k: v.detach().float() for k, v in model.state_dict().items()}
        g = torch.Generator().manual_seed(1000)
        mods = kw.get('modalities', 'LC')
        img = [torch.randn(1, ncam, C, img_hw[0] // 32, img_hw[1] // 32, generator=g)] if 'C' in mods else None
        pts = [torch.randn(1, C, *pts_hw, generator=g)] if 'L' in mods else None
        metas = syn.img_metas(1, ncam, img_hw)
        bev_q = torch.randn(200 * 200, C, generator=g) * 0.1
        pos = torch.randn(1, C, 200, 200, generator=g) * 0.1
        cfg = json.loads(json.dumps(tcfg))

        def run(backward):
            P = {k: (v.clone().requires_grad_() if backward else v) for k, v in sd.items()}
            with torch.set_grad_enabled(backward):
                fused = R.transformer_encode_fuse(P, cfg, img, pts, bev_q, 200, 200, pos, metas)
                if backward:
                    fused.square().mean().backward()
            return fused
        for backward in ((False, True) if with_bwd else (False,)):
            run(backward)
            times, spent = [], 0.0
            while len(times) < max_passes and spent < max_seconds:
                t0 = time.perf_counter()
                run(backward)
                times.append(time.perf_counter() - t0)
                spent += times[-1]
            med = float(np.median(times))
            out.append({'config': name, 'pass': 'fwd+bwd' if backward else 'fwd', 'seconds_per_sample': med,
                        'samples_per_s': 1.0 / med, 'timed_passes': len(times)})
    cpu = platform.processor() or ''
    try:
        cpu = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except Exception:
        pass
    return {'kind': 'port', 'cores': torch.get_num_threads(), 'host_cores': os.cpu_count(), 'cpu': cpu,
            'protocol': f'oracle/unibev_ref.py (torch CPU), fp32, bs = 1, eval, 1 warm-up + <= {max_passes} timed passes '
                        f'(each entry bounded to {max_seconds:g} s of timed work), median', 'entries': out}


if __name__ == '__main__':
    main()
