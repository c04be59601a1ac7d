"""Factorised-encoder ViViT around the gated-token backbone (API of the reference's models/vivit.py).

Constructor kwargs, sub-module names and state_dict keys are the reference's (`embedding.conv.*`,
`spatial_model.{class_token,backbone.*,layer_norm.*}`, `temporal_model.*`, `classifier.*`;
models/vivit.py:24-99,272-291), so its checkpoints and YAML configs load unchanged.  What differs is
how the work reaches the GPU:

  * tubelet embedding (models/vivit.py:153-192): a strided Conv3d with kernel == stride is a GEMM over
    non-overlapping tubelets -- the clip is re-laid as (tubelets, C*t*h*w) rows once and goes through
    the MFMA gated-linear kernel K3 (dense mode) with the conv weight viewed as (dim, C*t*h*w);
  * per-step map of a sub-model (models/vivit.py:293-303): class token + backbone, then the final
    LayerNorm on the class-token ROWS only (LayerNorm is row-wise, so that equals layer_norm(y)[:, 0]);
  * the spatial model steps through time with per-clip state in the gates/buffers of its blocks
    (models/vivit.py:139-150); the temporal model is four dense `Block`s on T+1 tokens (K8 attention);
  * classifier = K3; the view mean / softmax over (clips, classes) is a handful of scalars per clip.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from eventful_transformer import _native
from eventful_transformer.backbones import ViTBackbone
from eventful_transformer.base import ExtendedModule
from eventful_transformer.blocks import LN_EPS
from eventful_transformer.counting import CountedLinear


class ViViTSubModel(ExtendedModule):
    """One factorised sub-model: (B, n, D) tokens -> (B, D) class embedding (models/vivit.py:272-303)."""

    def __init__(self, input_size, backbone_config):
        super().__init__()
        dim = backbone_config["block_config"]["dim"]
        self.class_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.backbone = ViTBackbone(input_size=input_size, has_class_token=True, **backbone_config)
        self.layer_norm = nn.LayerNorm(dim, eps=LN_EPS)

    def forward(self, x):
        _native.require_hip(x)
        B, _, D = x.shape
        tokens = torch.concat([self.class_token.expand(B, 1, D), x], dim=1)
        y = self.backbone(tokens)
        cls_rows = y[:, 0].contiguous()
        out = torch.empty_like(cls_rows)
        _native.row_pass(cls_rows, B, D, ln_w=self.layer_norm.weight, ln_b=self.layer_norm.bias, eps=LN_EPS, c_out=out)
        return out


class TubeletEmbedding(nn.Module):
    """Tubelet (t, h, w) -> token vector.  The parameters live in a Conv3d for state_dict compatibility
    (`conv.weight` (dim, C, t, h, w), `conv.bias`); the arithmetic is one K3 GEMM over tubelet rows."""

    def __init__(self, input_channels, dim, tubelet_shape):
        super().__init__()
        self.tubelet_shape = tuple(tubelet_shape)
        self.conv = nn.Conv3d(in_channels=input_channels, out_channels=dim, kernel_size=self.tubelet_shape,
                              stride=self.tubelet_shape)
        self._split = None

    def _planes(self, w2):
        key = (w2.data_ptr(), -1 if self.conv.weight.is_inference() else self.conv.weight._version, _native.GEMM_MODE)
        if self._split is None or self._split[0] != key:
            self._split = (key, _native.split_weight(w2))
        return self._split[1]

    def forward(self, x):
        """x (B, T, C, H, W) float32 -> (B, T/t, (H/h)*(W/w), dim)."""
        _native.require_hip(x)
        B, T, C, H, W = x.shape
        tt, th, tw = self.tubelet_shape
        nt, nh, nw = T // tt, H // th, W // tw
        # rows ordered (b, time step, patch row, patch col); columns ordered like conv.weight's (C, t, h, w)
        rows = x[:, : nt * tt, :, : nh * th, : nw * tw].reshape(B, nt, tt, C, nh, th, nw, tw)
        rows = rows.permute(0, 1, 4, 6, 3, 2, 5, 7).reshape(B * nt * nh * nw, C * tt * th * tw).contiguous()
        dim, K = self.conv.out_channels, C * tt * th * tw
        w2 = self.conv.weight.detach().view(dim, K)
        out = torch.empty((B, nt, nh * nw, dim), dtype=torch.float32, device=x.device)
        M = rows.shape[0]
        _native.gated_linear(rows, K, None, M, w2, self.conv.bias, out, dim, None, M, None, None, 1, M, K, dim,
                             W_split=self._planes(w2))
        return out


class ViViTPreprocessing(nn.Module):
    """Value normalisation + temporal / spatial view extraction (models/vivit.py:195-269).  Input
    (B, frames, C, H, W) uint8 or float in [0, 1]; returns a list of float32 views (B, t, C, h, w)."""

    def __init__(self, input_shape, normalize_mean, normalize_std, spatial_views, temporal_stride, temporal_views):
        super().__init__()
        self.input_shape = tuple(input_shape)
        self.temporal_stride = temporal_stride
        self.temporal_views = temporal_views
        self.spatial_views = spatial_views
        self.normalize_mean = normalize_mean
        self.normalize_std = normalize_std

    def _normalize(self, x):
        mean = torch.as_tensor(self.normalize_mean, dtype=x.dtype, device=x.device)
        std = torch.as_tensor(self.normalize_std, dtype=x.dtype, device=x.device)
        if mean.ndim:
            mean, std = mean.view(-1, 1, 1), std.view(-1, 1, 1)
        return (x - mean) / std

    @staticmethod
    def _fit(frames, size):
        # utils/image.py:51-58: scale so that the frame covers `size` (bilinear, antialiased); 1.0 -> untouched
        scale = max(size[0] / frames.shape[-2], size[1] / frames.shape[-1])
        if scale == 1.0:
            return frames
        new = [round(scale * frames.shape[-2]), round(scale * frames.shape[-1])]
        flat = frames.reshape((-1,) + frames.shape[-3:])
        return F.interpolate(flat, size=new, mode="bilinear", antialias=True, align_corners=False).reshape(
            frames.shape[:-2] + tuple(new))

    def forward(self, x):
        t, _, h, w = self.input_shape
        span = self.temporal_stride * t
        if x.shape[1] < span:  # repeat the last frame of short videos
            x = torch.concat([x, x[:, -1:].expand(-1, span - x.shape[1], -1, -1, -1)], dim=1)
        if self.temporal_views == 1:
            starts = [(x.shape[1] - span) // 2]
        else:
            gap = (x.shape[1] - span) / (self.temporal_views - 1)
            starts = [int(i * gap) for i in range(self.temporal_views)]
        views = []
        for s in starts:
            v = x[:, s: s + span: self.temporal_stride]
            v = v.float() / 255.0 if v.dtype == torch.uint8 else v
            views.append(self._fit(self._normalize(v), (h, w)))
        H, W = views[0].shape[-2:]
        if self.spatial_views == 1:
            corners = [((H - h) // 2, (W - w) // 2)]
        else:
            gy, gx = (H - h) / (self.spatial_views - 1), (W - w) / (self.spatial_views - 1)
            corners = [(int(i * gy), int(i * gx)) for i in range(self.spatial_views)]
        return [v[..., i: i + h, j: j + w] for i, j in corners for v in views]


class FactorizedViViT(ExtendedModule):
    """Spatio-temporal factorised ViViT classifier (models/vivit.py:18-150): clip -> class probabilities."""

    def __init__(self, classes, input_shape, normalize_mean, normalize_std, spatial_config, spatial_views,
                 temporal_config, temporal_stride, temporal_views, tubelet_shape, batch_views=True, dropout_rate=0.0,
                 spatial_only=False, temporal_only=False):
        super().__init__()
        assert not (spatial_only and temporal_only)
        assert 0.0 <= dropout_rate <= 1.0
        input_t, input_c, input_h, input_w = tuple(input_shape)
        tubelet_shape = tuple(tubelet_shape)
        self.batch_views = batch_views
        self.spatial_only = spatial_only
        self.temporal_only = temporal_only
        self.preprocessing = ViViTPreprocessing(tuple(input_shape), normalize_mean, normalize_std, spatial_views,
                                                temporal_stride, temporal_views)
        dim = spatial_config["block_config"]["dim"]
        self.embedding = TubeletEmbedding(input_c, dim, tubelet_shape)
        self.spatial_model = ViViTSubModel((input_h // tubelet_shape[1], input_w // tubelet_shape[2]), spatial_config)
        self.temporal_model = ViViTSubModel((input_t // tubelet_shape[0],), temporal_config)
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0.0 else nn.Identity()
        self.classifier = CountedLinear(in_features=dim, out_features=classes)
        self._frames, self._frames_in_flight = {}, None   # see use_frame_graphs (None: automatic)

    def forward(self, x):
        clips = x.shape[0]
        if not self.temporal_only:
            x = self._forward_spatial(x)
        if not self.spatial_only:
            x = self._forward_temporal(x, clips)
        return x

    def _forward_spatial(self, x):
        views = self.preprocessing(x)
        if self.batch_views:  # views ride on the batch axis: one pass with (clips * views) streams of per-clip state
            return self._forward_view(torch.stack(views, dim=1).flatten(end_dim=1))
        return torch.stack([self._forward_view(v) for v in views], dim=1).flatten(end_dim=1)

    AUTO_GRAPH_VIEWS = 2        # automatic mode: view batches up to this size replay HIP graphs (eager steps are host-bound there)
    AUTO_FRAMES_IN_FLIGHT = 3
    MAX_FRAME_GRAPHS = 2        # captured shapes kept (each holds its own copy of the per-clip state)

    def use_frame_graphs(self, frames_in_flight=None):
        """MI355X addition (no counterpart in the reference, whose spatial steps run eagerly): replay the spatial sub-model's
        per-step launches as HIP graphs (eventful_transformer.graphs.FrameGraphs), optionally with `frames_in_flight`
        consecutive time steps of a view side by side (`run_pipelined`).  Same kernels on the same state: bit-identical
        probabilities; at batch 1 the eager steps are bound by the host's launch rate.
          None (default): automatic -- graphs with 3 steps in flight when a call carries at most 2 view streams, eager steps for
                          larger batches, while counting MACs, in training mode or when forward hooks are registered;
          0: always eager steps;   n >= 1: always graphs, n steps in flight.
        One FrameGraphs is captured per (views, patches, dim, dtype, device) of the step input -- the stacked-views and per-view
        paths and a smaller final batch each get their own -- and the two most recently used are kept."""
        for frames in self._frames.values():
            frames.release()
        self._frames = {}
        self._frames_in_flight = None if frames_in_flight is None else max(0, int(frames_in_flight))

    def _graph_mode(self, tokens):
        """Steps in flight for this call's graph replay, or 0 for eager steps."""
        if self._frames_in_flight is not None:
            return self._frames_in_flight
        if tokens.shape[0] > self.AUTO_GRAPH_VIEWS or self.training or getattr(self, "count_mode", False) or not tokens.is_cuda:
            return 0
        if any(m._forward_hooks or m._forward_pre_hooks for m in self.spatial_model.modules()):
            return 0      # hooks would only run while capturing
        return self.AUTO_FRAMES_IN_FLIGHT

    def _forward_view(self, x):
        tokens = self.embedding(x.contiguous())  # (views, time, patches, dim)
        P = self._graph_mode(tokens)
        if P >= 1:
            return self._forward_view_graphs(tokens, P)
        self.spatial_model.reset()
        return torch.stack([self.spatial_model(tokens[:, t]) for t in range(tokens.shape[1])], dim=1)

    def _frame_graphs_for(self, step):
        from eventful_transformer.graphs import FrameGraphs

        key = (tuple(step.shape), step.dtype, step.device)
        frames = self._frames.pop(key, None)
        if frames is None:
            while len(self._frames) >= self.MAX_FRAME_GRAPHS:
                self._frames.pop(next(iter(self._frames))).release()
            frames = FrameGraphs(self.spatial_model)
        self._frames[key] = frames      # most recently used last
        return frames

    def _forward_view_graphs(self, tokens, P):
        T = tokens.shape[1]
        steps = tokens.transpose(0, 1).contiguous()          # (time, views, patches, dim)
        frames = self._frame_graphs_for(steps[0])
        frames.reset()
        out = [frames(steps[0]).clone()]
        t = 1
        while t < T:
            if P > 1 and t + P <= T:
                out += [y.clone() for y in frames.run_pipelined(steps[t:t + P])]
                t += P
            else:
                out.append(frames(steps[t]).clone())
                t += 1
        return torch.stack(out, dim=1)

    def _forward_temporal(self, x, clips):
        x = x.reshape((-1,) + tuple(x.shape[-2:])).contiguous()
        x = self.classifier(self.dropout(self.temporal_model(x)))
        return x.view(clips, -1, x.shape[-1]).mean(dim=-2).softmax(dim=-1)
