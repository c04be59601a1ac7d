"""Token-selection policies (API of the reference's policies.py).

A policy maps a tensor of per-token deltas to the indices of the tokens to recompute.  Here the
L2 norm is one row pass (K1a) and the selection is the LDS radix-select / ordered chunk-scan
compaction kernel (K1).  Indices come back int64 and ASCENDING -- a documented tightening of the
reference, whose `topk(sorted=False)` order is implementation-defined (policies.py:63); ties at
the k-th norm go to the lowest token index.

Inside the blocks the gates call `select_into()` with the norms their fused row pass already
produced, so the delta tensor is never materialised and a threshold policy's variable count never
leaves the device (the reference's `nonzero()` is a host sync, policies.py:27-28).
"""
import torch

from eventful_transformer import _native
from eventful_transformer.base import ExtendedModule


def _token_norms(x, dim, order=2):
    """Norm of order 1 / 2 / inf over `dim` (-1: tokens along -2; -2: tokens along -1) -> (B', N) float32 on device."""
    _native.require_hip(x)
    if dim not in (-1, -2, x.ndim - 1, x.ndim - 2):
        raise RuntimeError(f"policy: the norm must reduce one of the last two dims, got dim={dim}")
    if dim in (-2, x.ndim - 2):
        x = x.transpose(-1, -2)
    x = x.float().contiguous()
    N, D = x.shape[-2], x.shape[-1]
    lead = x.shape[:-2]
    rows = x.numel() // D
    pad = (-D) % 4
    if pad:  # kernels read rows as float4
        x = torch.nn.functional.pad(x, (0, pad))
        D += pad
    norms = torch.empty(rows, dtype=torch.float32, device=x.device)
    _native.row_pass(x, rows, D, norms=norms, order=order)
    return norms, lead, N


class _NormPolicy(ExtendedModule):
    order = 2

    def _check_order(self):
        _native.norm_order(self.order)   # 1, 2, inf (policies.py:11,44,76); anything else raises NotImplementedError

    # -- used by the fused gates ---------------------------------------------------------------
    def capacity(self, n_tokens):
        """Upper bound on the number of selected tokens per clip (row stride of the index list)."""
        raise NotImplementedError

    def fixed_count(self, n_tokens):
        """Exact per-clip count if it is data-independent, else None."""
        raise NotImplementedError

    def select_into(self, norms, B, N, idx, count, rest=None, parts=0):
        """norms (B,N) f32 -> idx (B,capacity) int32 ascending; `count` (B,) int32 if data-dependent;
        `rest` (B,N) int32, optional: the complement list (unselected tokens, ascending).
        parts > 0: `norms` holds (B,N,parts) partial sums of squares instead (fused attention epilogue)."""
        raise NotImplementedError


class TokenNormThreshold(_NormPolicy):
    """Selects tokens whose delta norm exceeds a threshold (policies.py:6-32)."""

    def __init__(self, threshold=0.0, order=2):
        super().__init__()
        self.threshold = threshold
        self.order = order

    def capacity(self, n_tokens):
        return n_tokens

    def fixed_count(self, n_tokens):
        return None

    def select_into(self, norms, B, N, idx, count, rest=None, parts=0):
        self._check_order()
        assert not parts or self.order == 2, "per-head partial squares only make an L2 norm"
        _native.select_threshold(norms, B, N, self.threshold, idx.shape[-1], idx, count, rest, parts=parts)

    def forward(self, x, dim=-1):
        # The reference asserts batch 1 because nonzero() flattens the batch (policies.py:25).
        assert all(size == 1 for size in x.shape[:-2])
        norms, lead, N = _token_norms(x, dim, self.order)
        idx = torch.empty((1, N), dtype=torch.int32, device=x.device)
        count = torch.empty(1, dtype=torch.int32, device=x.device)
        self.select_into(norms, 1, N, idx, count)
        r = int(count.item())  # stand-alone API returns a sized tensor: one readback
        return idx[0, :r].long().view((1,) * (x.ndim - 2) + (-1,))


class TokenNormTopK(_NormPolicy):
    """Selects the k tokens with the largest delta norm (policies.py:39-68)."""

    def __init__(self, k, order=2, save_status=False):
        super().__init__()
        self.k = k
        self.order = order
        self.save_status = save_status
        self.last_input = None
        self.last_output = None

    def _k(self, n_tokens):
        return self.k

    # k = 0 (`topk(0)`: an empty index, policies.py:63 -- e.g. TokenNormTopFraction with fraction * N < 1): every gate forwards nothing.
    # The fused gates take it as a device-side count of 0 over an index list of capacity 1, the route of a threshold policy that
    # selects nothing (the kernels address index lists of at least one slot).
    def capacity(self, n_tokens):
        return max(1, self._k(n_tokens))

    def fixed_count(self, n_tokens):
        return self._k(n_tokens) or None

    def select_into(self, norms, B, N, idx, count, rest=None, parts=0):
        self._check_order()
        assert not parts or self.order == 2, "per-head partial squares only make an L2 norm"
        if self._k(N) == 0:
            count.zero_()
            if rest is not None:
                rest.copy_(torch.arange(N, dtype=torch.int32, device=rest.device).expand(B, N))
            return
        _native.select_topk(norms, B, N, self._k(N), idx, rest, parts=parts)

    def forward(self, x, dim=-1):
        if self._k(x.shape[-2 if dim in (-1, x.ndim - 1) else -1]) == 0:
            _native.require_hip(x)
            out = torch.empty(tuple(x.shape[:-2]) + (0,), dtype=torch.int64, device=x.device)
            if getattr(self, "save_status", False):
                self.last_input, self.last_output = x.clone(), out.clone()
            return out
        norms, lead, N = _token_norms(x, dim, self.order)
        B = norms.numel() // N
        k = self._k(N)
        idx = torch.empty((B, k), dtype=torch.int32, device=x.device)
        self.select_into(norms, B, N, idx, None)
        out = idx.long().view(tuple(lead) + (k,))
        if getattr(self, "save_status", False):
            self.last_input = x.clone()
            self.last_output = out.clone()
        return out


class TokenNormTopFraction(TokenNormTopK):
    """Selects int(fraction * N) tokens with the largest delta norm (policies.py:71-95)."""

    def __init__(self, fraction, order=2):
        assert not (fraction < 0.0 or fraction > 1.0)
        super().__init__(k=None, order=order)
        self.fraction = fraction

    def _k(self, n_tokens):
        return int(self.fraction * n_tokens)
