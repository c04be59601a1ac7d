"""Index helpers, position encodings and drop-path (API of the reference's utils.py).

On the gated path the index broadcasting of `expand_row_index` / `expand_col_index` is index math
inside the kernels (a (B,k) int32 list is all they take); the helpers remain for callers that
gather/scatter with ATen.  `RelativePositionEmbedding` keeps its parameters and its cached
(h, h', dh) tables; inside the blocks the per-frame einsum + two dense N x N adds are fused into
the softmax kernel (K5).
"""
from math import prod

import torch
from torch import nn
from torch.nn import functional as func

from eventful_transformer import _native
from eventful_transformer.base import ExtendedModule
from eventful_transformer.counting import CountedAdd, CountedEinsum


def expand_col_index(index, target_shape):
    """(..., k) -> (..., rows, k) view usable as a dim=-1 gather/scatter index (utils.py:198-203)."""
    lead = index.shape[:-1]
    fill = len(target_shape) - index.ndim
    return index.view(lead + (1,) * fill + index.shape[-1:]).expand(tuple(target_shape[:-1]) + (-1,))


def expand_row_index(index, target_shape):
    """(..., k) -> (..., k, features) view usable as a dim=-2 gather/scatter index (utils.py:206-211)."""
    lead = index.shape[:-1]
    fill = len(target_shape) - index.ndim - 1
    return index.view(lead + (1,) * fill + (index.shape[-1], 1)).expand(
        tuple(target_shape[:-2]) + (-1, target_shape[-1])
    )


class DropPath(ExtendedModule):
    """Stochastic depth; identity in eval mode (utils.py:10-29)."""

    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate

    def forward(self, x):
        if not self.training:
            return x
        keep = torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), device=x.device) > self.drop_rate
        return x.div(1.0 - self.drop_rate) * keep.to(x.dtype)


def _cubic_weights(in_size, out_size, device):
    """(out_size, in_size) matrix of ATen's bicubic taps along one axis (align_corners=False, A = -0.75, border
    indices clamped: UpSampleBicubic2d / upsample_get_cubic_coefficients)."""
    A = -0.75
    # source coordinate in fp32 exactly as ATen forms it (area_pixel_compute_scale / _source_index, cubic=true)
    scale = torch.tensor(in_size, dtype=torch.float32) / torch.tensor(out_size, dtype=torch.float32)
    src = scale * (torch.arange(out_size, dtype=torch.float32) + 0.5) - 0.5
    i0 = torch.floor(src)
    t = (src - i0).double()
    conv1 = lambda x: ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0
    conv2 = lambda x: ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    taps = torch.stack([conv2(t + 1.0), conv1(t), conv1(1.0 - t), conv2(2.0 - t)], dim=1)
    idx = (i0.long().unsqueeze(1) + torch.arange(-1, 3)).clamp_(0, in_size - 1)
    w = torch.zeros(out_size, in_size, dtype=torch.float64)
    w.scatter_add_(1, idx, taps)
    return w.to(torch.float32).to(device)


def bicubic_resize(grid, size):
    """`F.interpolate(grid, size, mode="bicubic", align_corners=False)` for (N, C, H, W) as two small matmuls with the
    same taps (bicubic is separable).  ATen's HIP kernel gives one thread per output pixel a loop over all N*C planes:
    ~1 ms for a 64-channel 64x64 rel-pos table, 8 tables per ViTDet reset; this is two GEMMs of a few MFLOP.
    Same weights, different summation order: agrees to ~1e-6 (tests/test_host_logic.py)."""
    wy = _cubic_weights(grid.shape[-2], size[0], grid.device)
    wx = _cubic_weights(grid.shape[-1], size[1], grid.device)
    return torch.einsum("oh,nchw,pw->ncop", wy, grid, wx)


class PositionEncoding(ExtendedModule):
    """Learned absolute position encoding, bicubically resized to the input grid once and cached in
    eval mode (utils.py:32-105).  `sized()` hands the cached (1,N,D) table to the backbone, which
    folds the add into the first block's row pass."""

    def __init__(self, dim, encoding_size, input_size, has_class_token):
        super().__init__()
        self.encoding_size = tuple(encoding_size)
        self.input_size = tuple(input_size)
        self.has_class_token = has_class_token
        self.encoding = nn.Parameter(torch.zeros(1, prod(self.encoding_size) + int(has_class_token), dim))
        self.add = CountedAdd()
        self.cached_encoding = None

    def sized(self):
        if self.training:
            self.cached_encoding = None
            return self._compute_sized_encoding()
        if self.cached_encoding is None:
            self.cached_encoding = self._compute_sized_encoding().contiguous()
        return self.cached_encoding

    def forward(self, x):
        enc = self.sized()
        if tuple(x.shape[-2:]) != tuple(enc.shape[-2:]):   # (the reference's `x + encoding` fails to broadcast, utils.py:66)
            raise RuntimeError(f"PositionEncoding: input of {tuple(x.shape[-2:])} (tokens, channels) but the encoding is sized for {tuple(enc.shape[-2:])}")
        if x.is_cuda and x.dtype == torch.float32 and enc.dtype == torch.float32 and x.is_contiguous() and enc.is_contiguous() and x.ndim == 3:
            # row pass with the (1,N,D) table broadcast over clips (utils.py:66)
            B, N, D = x.shape
            out = torch.empty_like(x)
            _native.row_pass(x, B * N, D, res=enc, res_rows=N, sum_out=out)
            if self.add.count_mode:
                self.add.counts["add_flops"] += out.numel()
            return out
        _native.require_hip(x)
        return self.add(x, enc)

    def _compute_sized_encoding(self):
        enc = self.encoding
        if self.input_size == self.encoding_size:
            return enc.detach()
        cls = None
        if self.has_class_token:  # the class token comes first (vivit.py:296)
            cls, enc = enc[:, :1], enc[:, 1:]
        grid = enc.transpose(1, 2).reshape(enc.shape[0], enc.shape[2], *self.encoding_size)
        grid = bicubic_resize(grid, self.input_size)   # utils.py:93-97 (F.interpolate bicubic)
        enc = grid.flatten(start_dim=2).transpose(1, 2)
        if cls is not None:
            enc = torch.concat([cls, enc], dim=1)
        return enc.detach()

    def reset_self(self):
        self.cached_encoding = None


class RelativePositionEmbedding(ExtendedModule):
    """Decomposed relative position embedding (utils.py:108-195)."""

    def __init__(self, attention_size, embedding_size, head_dim, pool_size=None):
        super().__init__()
        self.attention_size = tuple(attention_size)
        self.embedding_size = tuple(embedding_size)
        self.pool_size = pool_size
        self.y_embedding = nn.Parameter(torch.zeros(2 * self.embedding_size[0] - 1, head_dim))
        self.x_embedding = nn.Parameter(torch.zeros(2 * self.embedding_size[1] - 1, head_dim))
        self.add = CountedAdd()
        self.einsum = CountedEinsum()
        self.y_relative = None
        self.x_relative = None

    def tables(self):
        """Cached (h, h', dh) and (w, w', dh) tables; rebuilt after reset() (utils.py:151-156)."""
        if self.y_relative is None:
            _native._need_f32("RelativePositionEmbedding", y_embedding=self.y_embedding, x_embedding=self.x_embedding)
            self.y_relative = self._get_relative(self.y_embedding, dim=0).contiguous()
            self.x_relative = self._get_relative(self.x_embedding, dim=1).contiguous()
            if self.y_relative.shape[0] != self.attention_size[0] or self.x_relative.shape[0] != self.attention_size[1]:
                # utils.py:176-183 resizes BOTH decomposed tables to attention_size (h, w): with h != w the row table comes out (w, h',
                # dh) and the reference's einsum fails ("subscript h has size w ..."); same limitation here, said clearly
                shapes = (tuple(self.y_relative.shape), tuple(self.x_relative.shape))
                self.y_relative = self.x_relative = None
                raise RuntimeError(f"RelativePositionEmbedding: resizing a {self.embedding_size} embedding to the non-square attention size "
                                   f"{self.attention_size} yields tables {shapes}; the reference fails on this too -- use a square grid or an "
                                   f"embedding of the attention size")
        return self.y_relative, self.x_relative

    def count_fused(self, batch, heads):
        """Counters of one forward() when K5 does the arithmetic (utils.py:157-168)."""
        if self.count_mode:
            a = self.attention_size
            p = a if self.pool_size is None else (a[0] // self.pool_size[0], a[1] // self.pool_size[1])
            dh = self.y_embedding.shape[-1]
            n = batch * heads * prod(a)
            self.einsum.counts["einsum_flops"] += n * (p[0] + p[1]) * dh
            self.add.counts["add_flops"] += 2 * n * prod(p)

    def forward(self, x, q, inplace=True):
        """ATen form for stand-alone callers: x (B,H,N,N') scores, q (B,H,N,dh) unscaled queries."""
        a = self.attention_size
        p = a if self.pool_size is None else (a[0] // self.pool_size[0], a[1] // self.pool_size[1])
        y_rel, x_rel = self.tables()
        x = x.view(x.shape[:2] + a + p)
        q = q.view(q.shape[:2] + a + q.shape[-1:])
        x = self.add(x, self.einsum("abhwc,hkc->abhwk", q, y_rel).unsqueeze(dim=-1), inplace=inplace)
        x = self.add(x, self.einsum("abhwc,wkc->abhwk", q, x_rel).unsqueeze(dim=-2), inplace=True)
        return x.view(x.shape[:2] + (prod(a), prod(p)))

    def _get_relative(self, embedding, dim):
        size = self.embedding_size[dim]
        steps = torch.arange(size, device=embedding.device)
        rel = embedding.detach()[steps.unsqueeze(1) - steps.unsqueeze(0) + (size - 1)]
        if self.embedding_size != self.attention_size:
            rel = bicubic_resize(rel.transpose(0, 2).unsqueeze(0), self.attention_size)   # utils.py:176-183
            rel = rel.squeeze(0).transpose(0, 2)
        if self.pool_size is not None:
            rel = func.avg_pool1d(rel.transpose(1, 2), self.pool_size[dim]).transpose(1, 2)
        return rel

    def reset_self(self):
        self.y_relative = None
        self.x_relative = None
