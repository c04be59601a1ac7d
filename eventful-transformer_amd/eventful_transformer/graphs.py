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
nothing and the model's own `reset()` is only needed before the graphs exist.  Schedule on a fresh object:
frame 0 eager (fills the scratch pool, splits the weights) then captured and replayed; frame 1 eager; frame 2
captured; everything after is replay only.  The numbers are bit-identical to the eager path (same kernels,
same order, same buffers).  Weights are read at capture time (the bf16 planes of the split-precision GEMM are
built then): call `release()` after loading new weights.
"""
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
        self._inc_warm = False

    def reset(self):
        """Start a new clip.  Before the first-frame graph exists this resets the model; afterwards the graph
        replay re-initialises all state itself."""
        self._t = 0
        if self._first is None:
            self.model.reset()

    def release(self):
        """Drop the graphs (and their private memory pools) and reset the model."""
        self._first = self._inc = self._x = None
        self._t = 0
        self._inc_warm = False
        self.model.reset()

    def _capture(self):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = self._fwd(self._x)
        return graph, y

    @torch.inference_mode()
    def __call__(self, x):
        if getattr(self.model, "count_mode", False):
            raise RuntimeError("FrameGraphs: operation counting reads counters back to the host; use the eager model")
        if self._x is None:
            self._x = torch.empty_like(x, memory_format=torch.contiguous_format)
        elif x.shape != self._x.shape or x.dtype != self._x.dtype or x.device != self._x.device:
            raise RuntimeError(f"FrameGraphs: frame {tuple(x.shape)} {x.dtype} differs from the captured "
                               f"{tuple(self._x.shape)} {self._x.dtype}; call release() to re-capture")
        self._x.copy_(x)
        if self._t == 0:
            if self._first is None:
                self.model.reset()
                self._fwd(self._x)           # eager once: scratch pool, split weight planes, window maps
                self.model.reset()
                # reset() dropped every cache that is rebuilt lazily: the bf16 weight planes, the rel-pos tables and the
                # sized position encoding.  Rebuild them OUTSIDE the capture -- the bicubic resize creates host tensors
                # and copies them to the device, which must not become graph nodes reading freed host memory.
                for m in self.model.modules():
                    if isinstance(m, CountedLinear):
                        m.split_planes()
                    elif isinstance(m, RelativePositionEmbedding):
                        m.tables()
                    elif isinstance(m, PositionEncoding):
                        m.sized()
                self._first = self._capture()
            graph, y = self._first
            graph.replay()
        elif self._inc is None and not self._inc_warm:
            self._inc_warm = True            # eager once: the incremental path's scratch buffers
            y = self._fwd(self._x)
        else:
            if self._inc is None:
                self._inc = self._capture()
            graph, y = self._inc
            graph.replay()
        self._t += 1
        return y
