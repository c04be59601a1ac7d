"""Op wrappers that account MACs into `self.counts` (API of the reference's counting.py).

`CountedLinear` is on the gated-token path: its forward runs the MFMA gated-linear kernel (K3) in
dense mode.  Inside the blocks the fused path calls the same kernel with the gate's index list and
books the same counters, so `total_counts()` is identical whichever way a linear is reached.
`CountedMatmul` / `CountedAdd` / `CountedEinsum` exist for API
parity (sub-modules of `MatmulBuffer`, `RelativePositionEmbedding`, ...); inside the fused blocks
their arithmetic is part of kernels K4-K6 and only their counters are touched.  (The reference's `CountedBias` /
`CountedConv` are dead code there -- nothing instantiates them -- and are not mirrored.)
"""
import torch
import torch.nn as nn

from eventful_transformer import _native
from eventful_transformer.base import ExtendedModule


class CountedAdd(ExtendedModule):
    """a + b (optionally in place), counted as one op per output element (counting.py:9-22)."""

    def forward(self, a, b, inplace=False):
        if inplace:
            a += b
            out = a
        else:
            out = a + b
        if self.count_mode:
            self.counts["add_flops"] += out.numel()
        return out


class CountedEinsum(ExtendedModule):
    """torch.einsum with a MAC count obtained by contracting all-ones operands (counting.py:113-124)."""

    def forward(self, equation, *operands):
        if self.count_mode:
            ones = [torch.ones_like(x) for x in operands]
            self.counts["einsum_flops"] += int(torch.einsum(equation, *ones).sum())
        return torch.einsum(equation, *operands)


class CountedLinear(ExtendedModule):
    """y = x W^T + b with weights (out_features, in_features), zero-initialised like the reference
    (counting.py:127-162).  Forward = kernel K3 (fp32 MFMA) over all rows of x."""

    def __init__(self, in_features, out_features, device=None, dtype=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.zeros((out_features, in_features), device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(out_features, device=device, dtype=dtype))
        self._split = None
        # load_state_dict copies into `.data` without bumping the tensor version the cache below is keyed on
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.reset_self())

    def split_planes(self):
        """bf16 hi/lo planes of `weight` for the split-precision MFMA path (None when it does not apply).
        Cached against the weight's storage address + version and dropped on reset(), the same policy the
        reference applies to its cached rel-pos tables / position encodings ("just in case new weights get
        loaded", utils.py:102-105,191-195) and on load_state_dict().  Other in-place writes through `weight.data` keep
        pointer and version: call reset() after them."""
        w = self.weight
        # (inference tensors -- a model built or loaded under torch.inference_mode() -- track no version: the address alone keys them)
        key = (w.data_ptr(), -1 if w.is_inference() else w._version, _native.GEMM_MODE)
        if self._split is None or self._split[0] != key:
            self._split = (key, _native.split_weight(w))
        return self._split[1]

    def reset_self(self):
        self._split = None

    def count_rows(self, rows):
        """Book the MACs of `rows` token rows going through this layer (used by the fused blocks)."""
        if self.count_mode:
            self.counts["bias_flops"] += rows * self.out_features
            self.counts["linear_flops"] += rows * self.in_features * self.out_features

    def forward_bias(self, x):
        out = x + self.bias
        if self.count_mode:
            self.counts["bias_flops"] += out.numel()
        return out

    def forward_linear(self, x):
        if self.count_mode:
            self.counts["linear_flops"] += x.numel() * self.out_features
        zero = _native.scratch("zero_bias", (self.out_features,), torch.float32, x.device).zero_()
        return self._run(x, zero)

    def forward(self, x):
        self.count_rows(x.numel() // self.in_features)
        return self._run(x, self.bias)

    def _run(self, x, bias):
        _native.require_hip(x, self.weight)
        if x.dtype != torch.float32:
            raise RuntimeError(f"CountedLinear: float32 activations expected, got {x.dtype}")
        x2 = x.reshape(-1, self.in_features)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        out = torch.empty(x.shape[:-1] + (self.out_features,), dtype=torch.float32, device=x.device)
        _native.gated_linear(x2, self.in_features, None, rows, self.weight, bias, out, self.out_features, None, rows,
                             None, None, 1, rows, self.in_features, self.out_features, W_split=self.split_planes())
        return out


class CountedMatmul(ExtendedModule):
    """Batched a @ b with MAC counting (counting.py:165-175)."""

    def count_product(self, out_numel, inner):
        if self.count_mode:
            self.counts["matmul_flops"] += out_numel * inner

    def forward(self, a, b):
        out = a @ b
        self.count_product(out.numel(), a.shape[-1])
        return out
