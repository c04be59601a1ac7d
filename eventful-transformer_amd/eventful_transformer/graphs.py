"""HIP-graph replay of the per-frame forward (MI355X addition; the reference has no counterpart -- it runs
eagerly, scripts/time/*.py time `model(frame)` per frame).

One video stream at small r issues ~15 short kernels per block; at B = 1 the host (Python + ctypes) rather than
the GPU sets the frame time.  The gated path is capturable as is: every kernel is enqueued on the current
stream, index lists and the threshold policy's per-clip counts stay on the device, and all temporal state
(gate references, token buffers, q.k^T / A.v states) is updated IN PLACE.  `FrameGraphs` therefore records two
graphs per model -- the dense first frame of a clip and the gated incremental frame -- and replays them:

    frames = FrameGraphs(backbone)
    for clip in clips:
        frames.reset()
        for x in clip:
            y = frames(x)        # y is a static tensor, overwritten by the next call: clone to keep

Replaying the first-frame graph rewrites every state tensor completely, so `reset()` between clips costs
nothing and the model's own `reset()` is only needed before the graphs exist.  Schedule on a fresh object: the
FIRST call runs one first frame and one incremental frame eagerly on the given input (scratch pools of both paths,
split weight planes, window maps), resets the model, captures the first-frame graph, replays it and runs one eager gated
frame (lazily created gated-path state comes into being OUTSIDE a capture), captures the incremental graph and replays the
first-frame graph for the caller's frame; everything after is replay only.  (Until round 5 the incremental graph was warmed and captured lazily on frames 1 and 2: if another
FrameGraphs of the same model, or an eager `model.reset()`, ran in between, that late capture recorded launches against
state tensors that were not this object's.)  The numbers are bit-identical to the eager path (same kernels, same order, same
buffers).

What a capture bakes in, and how it is kept honest: the gate policies (k, threshold, index-list capacity), the weights (the bf16
planes of the split-precision GEMM are built at capture time) and train / eval mode.  `reset()` -- the start of every clip --
compares a signature of those (per gate: policy class + scalar parameters; per parameter and buffer: storage address, version,
dtype, device; plus a flag raised by a `load_state_dict` post-hook, which copies into `.data` without bumping versions) with
the captured one and re-captures when it differs: `set_policies`, `load_state_dict` and `.to()` between clips are safe.

Frame pipelining (`run_pipelined`): block i of frame t+1 only needs block i of frame t (its temporal state) and block
i-1 of frame t+1 (its input), so consecutive frames of ONE stream overlap like a wavefront.  With every launch of a
one-stream frame latency-bound (150-190 dependent launches of 5-80 us that each fill a fraction of the chip), P frames
are captured side by side on P HIP streams ("lanes", each with its own scratch buffers) in one graph, ordered by
per-block events: lane j enters block i once lane j-1 has left block i+1 (one block further than the state dependency:
a chained block hands its MLP token buffer to the next block's first row pass unevaluated, and the next frame must not
scatter into that buffer before it has been read).  Same kernels on the same state in the same per-block order: the
outputs are bit-identical to frame-by-frame replay; the latency of a frame is unchanged, the throughput of the stream
approaches P x.
"""
import weakref

import torch

from eventful_transformer import _native
from eventful_transformer.counting import CountedLinear
from eventful_transformer.utils import PositionEncoding, RelativePositionEmbedding


class FrameGraphs:
    """Graph-replayed `model(x)` for a stream of equally shaped frames (see module docstring)."""

    def __init__(self, model, forward=None):
        """model: a module of this package (reset(), modules()); forward: optional callable run per frame instead of
        `model(x)` when the backbone is wrapped by extra per-frame work (class token, final norm) worth capturing too."""
        _native.require_hip(next(model.parameters()))
        self.model = model
        self._fwd = forward if forward is not None else model
        self._first = None       # (graph, static output) of the first frame of a clip
        self._inc = None         # (graph, static output) of an incremental frame
        self._x = None           # static input
        self._t = 0
        self._pipe = None        # (graph, static inputs, static outputs, side streams) of `run_pipelined`
        self._keep = []          # weight planes / rel-pos tables / sized encodings the captured graphs read
        self._sig = None         # signature of what the captures baked in (policies, weights, mode)
        self._stale = False      # raised by the load_state_dict post-hook
        me = weakref.ref(self)

        def loaded(_module, _incompatible):
            obj = me()
            if obj is not None:
                obj._stale = True
        self._hook = model.register_load_state_dict_post_hook(loaded)

    def __del__(self):
        try:
            self._hook.remove()
        except Exception:
            pass

    def _signature(self):
        sig = [bool(self.model.training)]
        for m in self.model.modules():
            pol = getattr(m, "policy", None)
            if pol is not None:
                # scalars by value; anything else (a tuple, a tensor-valued threshold) by its repr, so that replacing it re-captures
                sig.append((type(pol).__name__,) + tuple(sorted((k, v if isinstance(v, (int, float, bool, str, type(None))) else repr(v))
                                                                for k, v in vars(pol).items())))
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            # (inference tensors have no version counter: reading it raises -- their storage cannot be edited in place either)
            sig.append((t.data_ptr(), -1 if t.is_inference() else t._version, t.dtype, str(t.device)))
        return tuple(sig)

    def _owns_model_state(self):
        """True while the model's Python-side state attributes (gate references, accumulators, first-frame flags) are the ones
        this object's captures created: no other FrameGraphs and no eager reset() has touched the model since."""
        owner = self.model.__dict__.get("_evt_state_owner")
        return owner is not None and owner() is self

    def reset(self):
        """Start a new clip.  Before the first-frame graph exists this resets the model; afterwards the graph
        replay re-initialises all state itself -- unless what the captures baked in has changed (see the module docstring):
        then the graphs are dropped and the next call captures again."""
        self._t = 0
        if self._first is not None and (self._stale or self._signature() != self._sig):
            self.release()
        if self._first is None:
            self.model.reset()

    def release(self):
        """Drop the graphs (and their private memory pools) and reset the model."""
        self._first = self._inc = self._x = self._pipe = None
        self._keep = []
        self._t = 0
        self._sig = None
        self._stale = False
        self.model.reset()

    def _capture(self):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = self._fwd(self._x)
        return graph, y

    def _check_frame(self, x, what):
        """Shared by __call__ and run_pipelined: no MAC counting (it reads counters back to the host, which cannot be captured),
        and the frame must look like the captured one (shape, dtype, device)."""
        if getattr(self.model, "count_mode", False):
            raise RuntimeError("FrameGraphs: operation counting reads counters back to the host; use the eager model")
        if self._x is not None and (x.shape != self._x.shape or x.dtype != self._x.dtype or x.device != self._x.device):
            raise RuntimeError(f"FrameGraphs{what}: frame {tuple(x.shape)} {x.dtype} on {x.device} differs from the captured "
                               f"{tuple(self._x.shape)} {self._x.dtype} on {self._x.device}; call release() to re-capture")

    def _build(self):
        """Everything that is not a replay, in ONE place and in one call: eager warm-up of both frame kinds, then both captures."""
        self.model.reset()
        self._fwd(self._x)               # eager first frame: scratch pool, split weight planes, window maps
        self._fwd(self._x)               # eager incremental frame: the gated path's scratch buffers
        self.model.reset()
        # reset() dropped every cache that is rebuilt lazily: the bf16 weight planes, the rel-pos tables and the
        # sized position encoding.  Rebuild them OUTSIDE the capture -- the bicubic resize creates host tensors
        # and copies them to the device, which must not become graph nodes reading freed host memory.
        # The graphs read these caches at their CAPTURED addresses: keep them alive for as long as this object lives -- a
        # later eager `model.reset()` (or another FrameGraphs capturing the same model at another shape) drops the
        # model's own references and would otherwise leave the replays reading freed memory.
        self._keep = []
        for m in self.model.modules():
            if isinstance(m, CountedLinear):
                self._keep.append(m.split_planes())
            elif isinstance(m, RelativePositionEmbedding):
                self._keep.append(m.tables())
            elif isinstance(m, PositionEncoding):
                self._keep.append(m.sized())
        self._first = self._capture()    # allocates the state tensors (graph-private pool), leaves the model "past its first frame"
        # The gated path creates some of its state lazily on the first gated frame (zero-filled accumulators, transposed gate
        # references): created INSIDE a capture, that zero fill would become a node of the incremental graph and wipe the state on
        # every replay.  So the state is brought to life for real first -- replay the first-frame graph, run ONE eager gated frame --
        # and only then is the gated path captured (it now finds everything in place: no creation inside the capture).  The caller
        # replays the first-frame graph right after this, which re-initialises the clip.
        self._first[0].replay()
        self._fwd(self._x)
        self._inc = self._capture()
        self._sig = self._signature()
        self._stale = False
        self.model.__dict__["_evt_state_owner"] = weakref.ref(self)

    @torch.inference_mode()
    def __call__(self, x):
        self._check_frame(x, "")
        if self._x is None:
            self._x = torch.empty_like(x, memory_format=torch.contiguous_format)
        self._x.copy_(x)
        if self._t == 0:
            if self._first is None:
                self._build()
            graph, y = self._first
        else:
            if self._inc is None:
                raise RuntimeError("FrameGraphs: the first frame of a clip goes through reset() + __call__")
            graph, y = self._inc
        graph.replay()
        self._t += 1
        return y

    # ---------------------------------------------------------------------------------------------
    def _backbones(self):
        from eventful_transformer.backbones import ViTBackbone

        return [m for m in self.model.modules() if isinstance(m, ViTBackbone)]

    def _capture_pipelined(self, xs):
        P = xs.shape[0]
        bbs = self._backbones()
        if len(bbs) != 1:
            raise RuntimeError("FrameGraphs.run_pipelined: the model must contain exactly one ViTBackbone")
        bb = bbs[0]
        side = [torch.cuda.Stream(device=xs.device) for _ in range(P - 1)]
        done = [[torch.cuda.Event() for _ in bb.blocks] for _ in range(P)]
        graph = torch.cuda.CUDAGraph()
        ys = []
        try:
            with torch.cuda.graph(graph):
                main = torch.cuda.current_stream()
                for s in side:                      # fork before any work is issued
                    s.wait_stream(main)
                for j in range(P):
                    st = main if j == 0 else side[j - 1]
                    bb.block_sync = _LaneSync(j, st, done)
                    with torch.cuda.stream(st), _native.lane(j):
                        ys.append(self._fwd(xs[j]))
                for s in side:                      # join
                    main.wait_stream(s)
        finally:
            bb.block_sync = None
        return graph, xs, ys, side

    @torch.inference_mode()
    def run_pipelined(self, xs):
        """xs: (P, ...) = the next P frames of the stream (not the first frame of a clip).  Returns the list of their P
        outputs.  From the second call on these are STATIC tensors that the next call overwrites (clone to keep); the first call
        runs the frames one by one (each lane's scratch buffers come into being outside the capture), returns copies and
        captures the P-lane graph for the later calls.  The clip must have been started through this object (`reset()` then
        `__call__` for its first frame): the temporal state the lanes update is the model's."""
        if self._t == 0 or self._x is None:
            raise RuntimeError("FrameGraphs.run_pipelined: the first frame of a clip goes through __call__")
        if xs.ndim < 2:
            raise RuntimeError("FrameGraphs.run_pipelined: expected a stack (P, ...) of frames")
        self._check_frame(xs[0], ".run_pipelined")
        P = xs.shape[0]
        if self._pipe is not None and (self._pipe[1].shape != xs.shape or self._pipe[1].dtype != xs.dtype or self._pipe[1].device != xs.device):
            raise RuntimeError("FrameGraphs.run_pipelined: frame stack differs from the captured one; call release()")
        if self._pipe is None:
            if not self._owns_model_state():
                # The lanes' eager warm-up and their capture run on the model's Python-side state, which another FrameGraphs (or an
                # eager reset) has taken over since this object captured: replay frame by frame instead (same results).
                return [_clone_out(self(xs[j])) for j in range(P)]
            ys = []
            for j in range(P):           # eager, one by one: each lane's scratch buffers come into being outside the capture
                with _native.lane(j):
                    ys.append(_clone_out(self._fwd(xs[j].contiguous())))
            self._t += P
            self._pipe = self._capture_pipelined(torch.empty_like(xs, memory_format=torch.contiguous_format))   # records, does not run
            return ys
        graph, sx, ys, _ = self._pipe
        sx.copy_(xs)
        graph.replay()
        self._t += P
        return ys


def _clone_out(y):
    """Copy of a forward's output: a tensor, or a dict / list / tuple of outputs (model wrappers return feature dicts)."""
    if isinstance(y, torch.Tensor):
        return y.clone()
    if isinstance(y, dict):
        return {k: _clone_out(v) for k, v in y.items()}
    if isinstance(y, (list, tuple)):
        return type(y)(_clone_out(v) for v in y)
    return y


class _LaneSync:
    """Event ordering of one lane of `run_pipelined` (called by ViTBackbone.forward around every block)."""

    def __init__(self, lane, stream, done):
        self.lane, self.stream, self.done = lane, stream, done

    def before_block(self, i, n):
        if self.lane > 0:
            self.stream.wait_event(self.done[self.lane - 1][min(i + 1, n - 1)])

    def after_block(self, i, n):
        self.done[self.lane][i].record(self.stream)
