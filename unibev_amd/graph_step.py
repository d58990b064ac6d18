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
                 has_pts=True, autocast_dtype=None):
        self.tr = transformer
        self.forward = forward
        self.cot = cotangent
        self.params = list(params)
        self.inputs = [x for x in inputs if x is not None]   # leaf inputs that receive gradients
        self.has = (has_img, has_pts)
        self.adt = autocast_dtype
        self.graphs = {}
        self.pool = None
        self.seed_base = torch.zeros(1, dtype=torch.int64, device=self.params[0].device)
        self.grads = FlatGradients(self.params)
        self.out = None

    def _combos(self):
        tr = self.tr
        if tr.drop_modality is None or not tr.training or not all(self.has):
            return [(1 if self.has[0] else 0, 1 if self.has[1] else 0)]
        return [(1, 1), (1, 0), (0, 1)]

    def _fwd_bwd(self):
        self.seed_base.add_(0x5DEECE66D)                 # fresh dropout masks per replay
        with torch.autocast('cuda', dtype=self.adt or torch.bfloat16, enabled=self.adt is not None):
            out = self.forward()
        (out.float() * self.cot).sum().backward()
        self.grads.collect()
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
                self.out = self._fwd_bwd()
            if self.pool is None:
                self.pool = g.pool()
            self.graphs[combo] = g
        cur.wait_stream(self.stream)
        self.tr.forced_flags = None
        self.grads.attach()                              # the optimizer reads the flat buffer
        return self

    def eager_step(self):
        """The same step without graphs (fallback, and the pass the per-kernel HIP-event timings
        are taken on): forward + backward, gradients into the flat buffer, rank average."""
        self._clear_grads()
        self.tr.forced_flags = None
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
        self.graphs[combo].replay()
        mark()
        self.grads.all_reduce_mean()
        mark()
        return combo

    def close(self):
        UF.set_seed_base(None)
        self.tr.forced_flags = None
