"""ViTDet around the gated-token backbone, up to the feature pyramid (API of the reference's models/vitdet.py).

Built here: `LinearEmbedding`, `ViTDetPreprocessing`, `PointwiseLayerNorm2d`, `SimplePyramid` and a `ViTDet` that
runs pre-backbone -> backbone -> pyramid (models/vitdet.py:17-125,211-251).  The region-proposal and ROI heads of the
reference are Detectron2 modules built from a Detectron2 LazyConfig (models/vitdet.py:188-192,204-209); Detectron2 is
an un-vendored dependency and out of scope, so `ViTDet.forward` returns the pyramid features the
heads would consume, under the feature names `p2..p6`.

Sub-module names and state_dict keys are the reference's (`embedding.conv.*`, `backbone.*`,
`pyramid.stages.<i>.<j>.{weight,bias}` with the reference's nn.Sequential positions), so its converted checkpoints
load.  The arithmetic runs TOKEN-MAJOR -- a feature map is (pixels, channels) rows, the layout the backbone already
produces -- on the kernels of the gated path:

  * patch embedding: Conv2d with kernel == stride == non-overlapping patches -> one K3 GEMM over patch rows;
  * ConvTranspose2d(kernel 2, stride 2): each input pixel makes a 2x2 output block -> four K3 GEMMs (one per tap)
    whose scatter epilogue writes the rows of the up-sampled map directly (no (pixels, 4C) intermediate, no permute);
  * 1x1 conv: K3; 3x3 conv (padding 1): im2col rows (zero-padded neighbourhood gather) -> K3;
  * PointwiseLayerNorm2d: the row pass K1a (LayerNorm over channels of each pixel row);
  * max-pools and the exact-erf GELU between the two transposed convs of the 4x stage: elementwise ATen ops on HIP.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from eventful_transformer import _native
from eventful_transformer.backbones import ViTBackbone
from eventful_transformer.base import ExtendedModule, numeric_tuple
from eventful_transformer.blocks import LN_EPS


class _Planes:
    """bf16 hi/lo planes of a derived weight matrix, cached against the parameter's storage + version."""

    def __init__(self):
        self._cache = {}

    def get(self, tag, param, make):
        key = (param.data_ptr(), -1 if param.is_inference() else param._version, _native.GEMM_MODE)
        hit = self._cache.get(tag)
        if hit is None or hit[0] != key:
            w2 = make().contiguous()
            hit = (key, w2, _native.split_weight(w2))
            self._cache[tag] = hit
        return hit[1], hit[2]


def _gemm(rows, w2, planes, bias, out, o_idx=None, o_rows=None):
    """rows (M, K) fp32 @ w2 (Nout, K)^T + bias -> out; with o_idx the M result rows go to rows o_idx of `out`."""
    M, K = rows.shape
    Nout = w2.shape[0]
    if bias is None:
        bias = _native.scratch("zero_bias", (Nout,), torch.float32, rows.device).zero_()
    if o_idx is None:
        _native.gated_linear(rows, K, None, M, w2, bias, out, Nout, None, M, None, None, 1, M, K, Nout, W_split=planes)
    else:
        _native.gated_linear(rows, K, None, M, w2, bias, out, Nout, o_idx, o_rows, None, None, 1, M, K, Nout, W_split=planes)


class LinearEmbedding(nn.Module):
    """Patch -> token vector (models/vitdet.py:17-51).  Parameters live in a Conv2d (`conv.weight` (dim, C, ph, pw))."""

    def __init__(self, input_channels, dim, patch_size):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.conv = nn.Conv2d(in_channels=input_channels, out_channels=dim, kernel_size=self.patch_size, stride=self.patch_size)
        self._planes = _Planes()

    def forward(self, x):
        """x (B, C, H, W) float32 -> (B, patches, dim)."""
        _native.require_hip(x)
        B, C, H, W = x.shape
        ph, pw = self.patch_size
        nh, nw = H // ph, W // pw
        rows = x[:, :, : nh * ph, : nw * pw].reshape(B, C, nh, ph, nw, pw).permute(0, 2, 4, 1, 3, 5)
        rows = rows.reshape(B * nh * nw, C * ph * pw).contiguous()
        dim = self.conv.out_channels
        w2, planes = self._planes.get("w", self.conv.weight, lambda: self.conv.weight.detach().reshape(dim, -1))
        out = torch.empty((B, nh * nw, dim), dtype=torch.float32, device=x.device)
        _gemm(rows, w2, planes, self.conv.bias, out)
        return out


class ViTDetPreprocessing(nn.Module):
    """Value normalisation (on the [0, 255] scale) and bottom-right zero padding to the model's input size
    (models/vitdet.py:223-251).  Expects inputs scaled to [0, 1]."""

    def __init__(self, input_shape, normalize_mean, normalize_std):
        super().__init__()
        self.input_shape = tuple(input_shape)
        self.normalize_mean = normalize_mean
        self.normalize_std = normalize_std
        self._stats = {}   # (dtype, device) -> (mean, std) on the device: built once, outside any HIP-graph capture

    def _mean_std(self, x):
        key = (x.dtype, x.device)
        if key not in self._stats:
            mean = torch.as_tensor(self.normalize_mean, dtype=x.dtype, device=x.device)
            std = torch.as_tensor(self.normalize_std, dtype=x.dtype, device=x.device)
            if mean.ndim:
                mean, std = mean.view(-1, 1, 1), std.view(-1, 1, 1)
            self._stats[key] = (mean, std)
        return self._stats[key]

    def forward(self, x):
        mean, std = self._mean_std(x)
        x = (x * 255.0 - mean) / std
        h, w = self.input_shape[-2:]
        return F.pad(x, (0, w - x.shape[-1], 0, h - x.shape[-2]))


class PointwiseLayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of a (B, C, H, W) map (models/vitdet.py:54-72).  Stand-alone calls take the
    NCHW tensor like the reference; inside `SimplePyramid` the token-major rows go through `rows()`."""

    def rows(self, x):
        """x (pixels, C) fp32 contiguous -> LayerNorm per row (K1a row pass)."""
        out = torch.empty_like(x)
        _native.row_pass(x, x.shape[0], x.shape[1], ln_w=self.weight, ln_b=self.bias, eps=self.eps, c_out=out)
        return out

    def forward(self, x):
        _native.require_hip(x)
        B, C, H, W = x.shape
        y = self.rows(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())
        return y.view(B, H, W, C).permute(0, 3, 1, 2)


class _Map:
    """A token-major feature map: rows (B*h*w, C) + its grid."""

    __slots__ = ("rows", "B", "h", "w")

    def __init__(self, rows, B, h, w):
        self.rows, self.B, self.h, self.w = rows, B, h, w

    def nchw(self):
        return self.rows.view(self.B, self.h, self.w, -1).permute(0, 3, 1, 2)


class SimplePyramid(nn.Module):
    """The ViTDet feature pyramid (models/vitdet.py:75-125): per scale an up/down-sampling stem, then
    1x1 conv -> LN -> 3x3 conv -> LN; plus a stride-2 sub-sampling of the coarsest map."""

    def __init__(self, scale_factors, dim, out_channels):
        super().__init__()
        self.stages = nn.ModuleList(self._build_scale(scale, dim, out_channels) for scale in scale_factors)
        self.max_pool = nn.MaxPool2d(kernel_size=1, stride=2, padding=0)
        self._planes = _Planes()
        self._up_idx = {}

    @staticmethod
    def _build_scale(scale, dim, out_channels):
        assert scale in [4.0, 2.0, 1.0, 0.5]
        if scale == 0.5:
            mid, start = dim, [nn.MaxPool2d(kernel_size=2, stride=2)]
        elif scale == 1.0:
            mid, start = dim, []
        elif scale == 2.0:
            mid, start = dim // 2, [nn.ConvTranspose2d(dim, dim // 2, kernel_size=2, stride=2)]
        else:
            mid = dim // 4
            start = [nn.ConvTranspose2d(dim, dim // 2, kernel_size=2, stride=2), PointwiseLayerNorm2d(dim // 2, eps=LN_EPS),
                     nn.GELU(), nn.ConvTranspose2d(dim // 2, mid, kernel_size=2, stride=2)]
        common = [nn.Conv2d(mid, out_channels, kernel_size=1, bias=False), PointwiseLayerNorm2d(out_channels, eps=LN_EPS),
                  nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1, bias=False),
                  PointwiseLayerNorm2d(out_channels, eps=LN_EPS)]
        return nn.Sequential(*start, *common)

    # -- layer kinds on token-major maps -----------------------------------------------------------------
    def _up_index(self, B, h, w, device):
        """(4, B*h*w) int32: output row (in the (B, 2h, 2w) map) of input pixel p for tap (a, c)."""
        key = (B, h, w, device)
        if key not in self._up_idx:
            b = torch.arange(B, device=device).view(B, 1, 1)
            i = torch.arange(h, device=device).view(1, h, 1)
            j = torch.arange(w, device=device).view(1, 1, w)
            taps = [((b * 2 * h + 2 * i + a) * 2 * w + 2 * j + c).reshape(-1) for a in range(2) for c in range(2)]
            self._up_idx[key] = torch.stack(taps).to(torch.int32).contiguous()
        return self._up_idx[key]

    def _conv_transpose(self, layer, tag, m):
        cout = layer.out_channels
        out = torch.empty((m.B * 4 * m.h * m.w, cout), dtype=torch.float32, device=m.rows.device)
        idx = self._up_index(m.B, m.h, m.w, m.rows.device)
        for t in range(4):
            a, c = divmod(t, 2)
            w2, planes = self._planes.get((tag, t), layer.weight, lambda: layer.weight.detach()[:, :, a, c].t())
            _gemm(m.rows, w2, planes, layer.bias, out, o_idx=idx[t], o_rows=out.shape[0])
        return _Map(out, m.B, 2 * m.h, 2 * m.w)

    def _conv(self, layer, tag, m):
        cout = layer.out_channels
        if layer.kernel_size == (1, 1):
            rows = m.rows
            w2, planes = self._planes.get(tag, layer.weight, lambda: layer.weight.detach().reshape(cout, -1))
        else:  # 3x3, padding 1: rows of the zero-padded 3x3 neighbourhood, columns ordered (tap row, tap col, channel)
            C = m.rows.shape[1]
            g = F.pad(m.rows.view(m.B, m.h, m.w, C), (0, 0, 1, 1, 1, 1))
            rows = g.unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(m.B * m.h * m.w, 9 * C).contiguous()
            w2, planes = self._planes.get(tag, layer.weight, lambda: layer.weight.detach().permute(0, 2, 3, 1).reshape(cout, -1))
        out = torch.empty((rows.shape[0], cout), dtype=torch.float32, device=rows.device)
        _gemm(rows, w2, planes, layer.bias, out)
        return _Map(out, m.B, m.h, m.w)

    @staticmethod
    def _pool(layer, m):
        C = m.rows.shape[1]
        g = m.rows.view(m.B, m.h, m.w, C)
        k, s = numeric_tuple(layer.kernel_size, 2), numeric_tuple(layer.stride, 2)
        if k == (1, 1):
            g = g[:, :: s[0], :: s[1]]
        else:
            assert k == (2, 2) and s == (2, 2)
            g = g[:, : m.h // 2 * 2, : m.w // 2 * 2].reshape(m.B, m.h // 2, 2, m.w // 2, 2, C).amax(dim=(2, 4))
        g = g.contiguous()
        return _Map(g.view(-1, C), m.B, g.shape[1], g.shape[2])

    def _run_stage(self, si, stage, m):
        for li, layer in enumerate(stage):
            tag = (si, li)
            if isinstance(layer, nn.ConvTranspose2d):
                m = self._conv_transpose(layer, tag, m)
            elif isinstance(layer, nn.Conv2d):
                m = self._conv(layer, tag, m)
            elif isinstance(layer, PointwiseLayerNorm2d):
                m = _Map(layer.rows(m.rows), m.B, m.h, m.w)
            elif isinstance(layer, nn.GELU):
                m = _Map(F.gelu(m.rows), m.B, m.h, m.w)
            elif isinstance(layer, nn.MaxPool2d):
                m = self._pool(layer, m)
            else:
                raise RuntimeError(f"SimplePyramid: unexpected layer {type(layer).__name__}")
        return m

    def forward_tokens(self, tokens, grid):
        """tokens (B, h*w, dim) from the backbone -> list of (B, out_channels, H', W') maps, finest first."""
        _native.require_hip(tokens)
        B, _, dim = tokens.shape
        base = _Map(tokens.reshape(-1, dim).contiguous(), B, grid[0], grid[1])
        maps = [self._run_stage(si, stage, base) for si, stage in enumerate(self.stages)]
        maps.append(self._pool(self.max_pool, maps[-1]))
        return [m.nchw() for m in maps]

    def forward(self, x):
        """x (B, dim, h, w), as in the reference."""
        B, dim, h, w = x.shape
        return self.forward_tokens(x.permute(0, 2, 3, 1).reshape(B, h * w, dim), (h, w))


class ViTDet(ExtendedModule):
    """ViTDet up to the pyramid: uint8 / [0,1] frame -> features p2..p6 (models/vitdet.py:128-220 minus the
    Detectron2 heads).  `pre_backbone` / `backbone` / `post_backbone` split the frame time the way
    scripts/time/vitdet_vid.py:33-45 does."""

    FEATURES = ("p2", "p3", "p4", "p5", "p6")

    def __init__(self, backbone_config, input_shape, normalize_mean, normalize_std, output_channels, patch_size,
                 scale_factors, classes=None, detectron2_config=None):
        super().__init__()
        input_c, input_h, input_w = input_shape
        patch_size = numeric_tuple(patch_size, length=2)
        self.backbone_input_size = (input_h // patch_size[0], input_w // patch_size[1])
        self.preprocessing = ViTDetPreprocessing(input_shape, normalize_mean, normalize_std)
        dim = backbone_config["block_config"]["dim"]
        self.embedding = LinearEmbedding(input_c, dim, patch_size)
        self.backbone = ViTBackbone(input_size=self.backbone_input_size, **backbone_config)
        self.pyramid = SimplePyramid(scale_factors, dim, output_channels)

    def pre_backbone(self, x):
        x = x.float() / 255.0 if x.dtype == torch.uint8 else x
        images = self.preprocessing(x)
        return images, self.embedding(images.contiguous())

    def post_backbone(self, images, x):
        return dict(zip(self.FEATURES, self.pyramid.forward_tokens(x, self.backbone_input_size)))

    def forward(self, x):
        images, x = self.pre_backbone(x)
        x = self.backbone(x)
        return self.post_backbone(images, x)
