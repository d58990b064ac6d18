"""One training step of the BEV-encoder hot path as HIP graphs.

The eager step issues ~470 kernel launches from Python; on MI355X the host needs as long to
enqueue them as the GPU needs to run them (15.3 ms against 13.9 ms of kernels, round-1 profile),
so no kernel improvement shows until the launches stop coming from the interpreter.  Here the
forward + backward of ``UniBEV_Head.forward_bev`` is captured once per modality-dropout outcome
and replayed:

* **modality dropout** stays the reference's host-side ``np.random`` draw
  (transformer_fusion.py:463-477): the flags select which captured graph runs — (camera, LiDAR) in
  {(1,1), (1,0), (0,1)}; both encoders execute in every one of them, as in the reference;
* **dropout masks** are a stateless hash of (seed, element): the per-call seeds are baked into a
  graph, a per-step base lives in device memory and is advanced by the graph itself
  (``functional.set_seed_base``), so every replay draws fresh masks and forward / backward of one
  replay agree;
* **gradients** are copied at the end of each graph into ONE flat f32 buffer whose views are the
  parameters' ``.grad``: the data-parallel exchange is a single RCCL all-reduce of that buffer
  (55.6 MB for the encoder-side parameters — one large message per step is what the point-to-point
  xGMI links want), and the optimizer reads the reduced views.  The collective is issued eagerly
  after the replay: nothing about RCCL is captured.

Everything outside — gradient clipping, AdamW — stays eager (about a dozen launches).
"""
import torch

from . import functional as UF
from . import linear as UL
from .dp import FlatGradients


class GraphedStep:
    """fwd + bwd of ``fn(*inputs)`` against a fixed cotangent, captured per modality-flag outcome.

    transformer   the ``UniBEVTransformer`` whose flags are drawn per step
    forward       callable () -> fused BEV features (runs under the caller's autocast settings)
    cotangent     tensor multiplied into the output to form the scalar that is back-propagated
    params        parameters receiving gradients
    """

    def __init__(self, transformer, forward, cotangent, params, inputs=(), has_img=True,
                 has_pts=True, autocast_dtype=None, split_after=None, cut_layer=1):
        """``split_after``: the encoders (``ImgEncoder`` / ``PtsEncoder``) whose backward is cut in two after their first
        ``cut_layer`` layers — see "split backward" below; None = one backward, one gradient message."""
        self.tr = transformer
        self.forward = forward
        self.cot = cotangent
        self.params = list(params)
        self.inputs = [x for x in inputs if x is not None]   # leaf inputs that receive gradients
        self.has = (has_img, has_pts)
        self.adt = autocast_dtype
        self.graphs = {}
        self.missing = {}
        self.pool = None
        self.seed_base = torch.zeros(1, dtype=torch.int64, device=self.params[0].device)
        self.split_after = [m for m in (split_after or []) if m is not None]
        self.cut_layer = int(cut_layer)
        self.split_note = None
        self.n_upper = 0
        if self.split_after:
            self._order_params_for_split()
        self.grads = FlatGradients(self.params, first_segment=self.n_upper)
        self.out = None

    # ---- split backward: the gradient exchange of the upper layers overlaps the backward of the first layer -----
    # The reference's MMDistributedDataParallel overlaps its gradient buckets with the backward (tools/train_UniBEV.py:
    # 242-249).  Here the backward is one captured HIP graph and RCCL stays outside the graphs, so the overlap comes from
    # CUTTING the backward where the first encoder layers end: graph A = forward + backward down to the cut (every
    # parameter used only above the cut — layers 2 and 3 of both encoders, the fusion weights — is then final: the
    # first segment of the flat buffer), graph B = the rest.  step(): replay A, start the all-reduce of segment 0 on
    # RCCL's stream, replay B beside it, all-reduce segment 1, wait.  The encoders sever their autograd graph at the cut
    # themselves (``cut_after``: what the upper layers read from below — the first layers' output, the embedded feature
    # tokens, the positional table — reaches them as fresh leaves, encoders._EncoderBase._sever): part 1 is a backward
    # from the loss to those leaves and the upper parameters, part 2 continues from the severed tensors with the leaves'
    # gradients.
    def _trace(self):
        for e in self.split_after:
            e.cut_after = self.cut_layer
        try:
            with torch.autocast('cuda', dtype=self.adt or torch.bfloat16, enabled=self.adt is not None):
                out = self.forward()
        finally:
            for e in self.split_after:
                e.cut_after = 0
        return out, [c for e in self.split_after for c in getattr(e, '_cuts', [])]

    def _reached(self, roots):
        """Indices of the parameters whose AccumulateGrad nodes are reached from the given graph nodes."""
        pid = {id(p): i for i, p in enumerate(self.params)}
        found, seen, stack = set(), set(), [r for r in roots if r is not None]
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            v = getattr(n, 'variable', None)
            if v is not None and id(v) in pid:
                found.add(pid[id(v)])
            stack.extend(f for f, _ in n.next_functions if f is not None)
        return found

    def _order_params_for_split(self):
        """One traced forward: parameters used only above the cut go first (segment 0 of the flat buffer)."""
        # the partition must hold for EVERY modality combination that will be captured (a parameter above the cut under
        # (1, 1) but below it under (0, 1) would make segment 0 incomplete in that graph): upper = used above the cut and
        # never below it in any combination; every combination must cut somewhere
        up, low, cuts = set(), set(), True
        for flags in self._combos():
            self.tr.forced_flags = flags
            try:
                out, c = self._trace()
                up |= self._reached([out.grad_fn])
                low |= self._reached([o.grad_fn for o, _ in c])
                cuts = cuts and bool(c)
                del out, c
            finally:
                self.tr.forced_flags = None
        mixed = up & low
        upper = [i for i in range(len(self.params)) if i in up]
        rest = [i for i in range(len(self.params)) if i not in up]
        self.split_note = None
        if not cuts or not upper or not rest or mixed:
            self.split_note = ('no split: %d parameter(s) are used above and below the cut' % len(mixed)) if mixed \
                else 'no split: nothing to cut'
            self.split_after = []
            return
        self.params = [self.params[i] for i in upper + rest]
        self.n_upper = len(upper)

    def _part1(self, out, cuts):
        """Backward from the loss down to the cut; segment 0 of the flat buffer is complete afterwards."""
        nu = self.n_upper
        torch.autograd.backward(out, self.cot.to(out.dtype), inputs=self.params[:nu] + [leaf for _, leaf in cuts])
        self.grads.collect(0, nu)
        return cuts

    def _part2(self, cuts):
        """The cut gradients flow on through the first layers; segment 1 is complete afterwards."""
        nu = self.n_upper
        roots = [(o, leaf.grad) for o, leaf in cuts if leaf.grad is not None]
        if roots:
            torch.autograd.backward([o for o, _ in roots], [g for _, g in roots],
                                    inputs=self.params[nu:] + self.inputs)
        self.grads.collect(nu, len(self.params))

    def _combos(self):
        tr = self.tr
        if tr.drop_modality is None or not tr.training or not all(self.has):
            return [(1 if self.has[0] else 0, 1 if self.has[1] else 0)]
        return [(1, 1), (1, 0), (0, 1)]

    def _fwd_bwd(self, part=None, between=None):
        """One forward + backward.  With a split backward: ``part`` 'A' = forward + part 1 (returns the state part 'B'
        continues from), None = both parts back to back (eager), ``between`` called after part 1."""
        if not self.split_after:
            self.seed_base.add_(0x5DEECE66D)             # fresh dropout masks per replay
            with torch.autocast('cuda', dtype=self.adt or torch.bfloat16, enabled=self.adt is not None):
                out = self.forward()
            # d(sum(out * cot)) / d(out) = cot: the cotangent is fed to the backward directly (forming the scalar costs a
            # product, a sum and an expanded product over the 82 MB output)
            torch.autograd.backward(out, self.cot.to(out.dtype))
            self.grads.collect()
            return out
        self.seed_base.add_(0x5DEECE66D)
        out, cuts = self._trace()
        state = self._part1(out, cuts)
        if part == 'A':
            self._state = state
            return out
        if between is not None:
            between()
        self._part2(state)
        return out

    def _clear_grads(self):
        for t in self.params + self.inputs:
            t.grad = None

    def capture(self, warmup=2):
        """Warm up (allocator, plans, lazily built caches), then capture one graph per flag
        combination into a shared memory pool — all on ONE non-default stream, which must also be
        the stream every earlier forward / backward of these parameters ran on: autograd pins each
        parameter's gradient accumulation to the stream it first ran on, and an accumulation that
        hops to another stream (the default one in particular) falls out of the capture.  The
        caller makes a side stream current for the whole run (``torch.cuda.set_stream``, as
        bench.py does); called from the default stream, a fresh side stream is used."""
        UF.set_seed_base(self.seed_base)
        cur = torch.cuda.current_stream()
        self.stream = cur if cur != torch.cuda.default_stream() else torch.cuda.Stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for combo in self._combos():
                self.tr.forced_flags = combo
                for _ in range(warmup):
                    self._clear_grads()
                    self._fwd_bwd()
        torch.cuda.synchronize()
        from . import dp
        dp.drain_watchdog()                              # RCCL's watchdog must hold nothing from this stream (dp.all_reduce)
        for combo in self._combos():
            self.tr.forced_flags = combo
            self._clear_grads()
            UL.mark_weights_changed()                    # the refresh of the 16-bit weight shadows is part of the graph
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: other threads (RCCL's watchdog polling its events) stay free to call into HIP
            with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode='thread_local'):
                self.out = self._fwd_bwd(part='A' if self.split_after else None)
            if self.pool is None:
                self.pool = g.pool()
            if self.split_after:
                g2 = torch.cuda.CUDAGraph()          # the backward below the cut: replayed beside segment 0's all-reduce
                with torch.cuda.graph(g2, pool=self.pool, stream=self.stream, capture_error_mode='thread_local'):
                    self._part2(self._state)
                self._state = None
                g = (g, g2)
            self.graphs[combo] = g
            # which parameters this combination leaves without a gradient (zero-filled in the graph): what the optimizer's
            # "every parameter received a gradient" check must see when THIS graph is replayed
            self.missing[combo] = list(self.grads.missing)
        cur.wait_stream(self.stream)
        self.tr.forced_flags = None
        self.grads.attach()                              # the optimizer reads the flat buffer
        return self

    def eager_step(self):
        """The same step without graphs (fallback, and the pass the per-kernel HIP-event timings
        are taken on): forward + backward, gradients into the flat buffer, rank average."""
        self._clear_grads()
        self.tr.forced_flags = None
        if self.split_after:
            self._fwd_bwd(between=lambda: self.grads.start_segment(0))
            self.grads.attach()
            self.grads.start_segment(1)
            self.grads.finish_segments()
            return
        self._fwd_bwd()
        self.grads.attach()
        self.grads.all_reduce_mean()

    def step(self, marks=None):
        """Replay the graph of this step's modality flags, then average the gradients over the
        ranks.  Returns the flags.  ``marks``: a list that receives three timing events recorded on the current
        stream — before the replay, after it, after the gradient exchange (bench.py's per-phase times)."""
        combo = self.tr.sample_modality_flags(*self.has) if len(self.graphs) > 1 \
            else next(iter(self.graphs))

        def mark():
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        mark()
        g = self.graphs[combo]
        self.grads.missing = self.missing.get(combo, [])
        if isinstance(g, tuple):                 # split backward: segment 0 travels while graph B runs
            g[0].replay()
            self.grads.start_segment(0)
            g[1].replay()
            mark()
            self.grads.start_segment(1)
            self.grads.finish_segments()
        else:
            g.replay()
            mark()
            self.grads.all_reduce_mean()
        mark()
        return combo

    def close(self):
        UF.set_seed_base(None)
        self.tr.forced_flags = None
