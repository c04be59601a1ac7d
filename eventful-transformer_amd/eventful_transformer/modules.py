"""Gates, buffers and accumulators (API of the reference's modules.py), on HIP kernels.

Each class keeps the reference's constructor, call signature, return values and state attribute
(`p`, `b`, `product`, `first`, `policy`), so they stay discoverable by `set_policies`
(utils/misc.py:140-143) and usable stand-alone.  The per-frame work is done by libevt_hip:

  TokenGate / TokenDeltaGate (rows) : K1a delta-norm -> K1 select -> K2 gather + reference update
  TokenBuffer (rows)                : row scatter
  MatmulBuffer                      : K4 (rows + columns of the q.k^T state in one launch)
  MatmulDeltaAccumulator            : K6 (two MFMA products accumulated into the state)

Inside `blocks.py` the fused path drives the same kernels directly on the packed token buffer and
uses these modules only as the owners of the per-clip state.  Column-structured gates/buffers
called stand-alone (the reference only ever uses them through EventfulBlock, where the fused
attention kernels do the work) and non-fp32 / non-contiguous stand-alone inputs run on the generic
kernels evt_gate_cols / evt_scatter_cols / evt_gate_rows_any / evt_move_rows_any.

Index tensors returned to Python are int64 and ascending.
"""
import torch

from eventful_transformer import _native
from eventful_transformer.base import ExtendedModule
from eventful_transformer.counting import CountedMatmul
from eventful_transformer.policies import _NormPolicy


def _rows_fast(*tensors):
    """Kernel path precondition for row-structured token tensors."""
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] % 4 == 0
               for t in tensors)


def _require_shape(who, what, got, want):
    """The state tensors of a clip have the first frame's sizes and the kernels address them with those: the reference fails in its
    gather / scatter with a shape error when a later frame differs (modules.py:90-96, 154-164); here that is an error up front instead
    of an out-of-bounds access."""
    if tuple(got) != tuple(want):
        raise RuntimeError(f"{who}: {what} of shape {tuple(got)} does not fit this clip's state {tuple(want)}; call reset() between clips")


def _index_i32(index, lead, device):
    """(…,k) int index broadcast over the leading dims `lead` -> contiguous (prod(lead), k) int32."""
    k = index.shape[-1]
    shape = index.shape[:-1] + (1,) * (len(lead) - (index.ndim - 1))
    return index.view(shape + (k,)).expand(tuple(lead) + (k,)).reshape(-1, k).to(device=device, dtype=torch.int32).contiguous()


class _GateBase(ExtendedModule):
    _tiles = None   # (EventfulBlock.matmul_gate on evt_attention_gated) the reference in the kernel's tiled layout
    _p = None

    def __init__(self, structure="row"):
        super().__init__()
        assert structure in ["row", "col"]
        self.structure = structure
        self.first = True
        self.policy = None
        self.p = None
        self._state_t = None   # EventfulBlock's large-N path keeps the "col" gate reference transposed; `p` is a view of it

    # `p` is the reference's state attribute (modules.py:122).  EventfulBlock's <= 256-token attention kernel keeps the "col" gate's
    # reference in a tiled layout (`use_tiles`): reading `p` then returns the logical (B,H,N,N) tensor as a COPY, assigning a tensor
    # of that shape writes it into the tiles (a documented deviation: in-place edits of the returned tensor do not reach the state).
    @property
    def p(self):
        if self._tiles is not None:
            return _native.tiles_to_logical(self._tiles, self._tiles_n)
        return self._p

    @p.setter
    def p(self, value):
        if self._tiles is not None and torch.is_tensor(value) and value.ndim == 4 and value.shape[-1] == value.shape[-2] == self._tiles_n \
                and tuple(value.shape[:2]) == tuple(self._tiles.shape[:2]):
            _native.logical_to_tiles(value, self._tiles)
            return
        self._tiles = None
        self._p = value

    def _p_shape(self):
        """Shape of the reference without materialising a tiled one."""
        return tuple(self._tiles.shape[:2]) + (self._tiles_n, self._tiles_n) if self._tiles is not None else tuple(self._p.shape)

    def use_tiles(self, tiles, n):
        """The reference lives in `tiles` (evt_attention_gated layout) from now on; `p` reads / writes go through it."""
        self._p = None
        self._tiles, self._tiles_n = tiles, n

    def reset_self(self):
        self.first = True
        self.p = None
        self._tiles = None
        self._state_t = None

    # -- selection --------------------------------------------------------------------------------
    def _select_rows(self, c, forced_index):
        """Returns (idx32 (B',cap), count or None, index64 shaped like the reference's)."""
        lead, N, D = c.shape[:-2], c.shape[-2], c.shape[-1]
        rows = c.numel() // D
        Bp = rows // N
        if forced_index is not None:
            return _index_i32(forced_index, lead, c.device), None, forced_index
        if isinstance(self.policy, _NormPolicy) and getattr(self.policy, "_k", lambda n: 1)(N) == 0:   # top-k with k = 0: nothing forwarded
            idx = torch.empty((Bp, 0), dtype=torch.int32, device=c.device)
            return idx, None, idx.long().view(tuple(lead) + (0,))
        if isinstance(self.policy, _NormPolicy):
            norms = _native.scratch("gate_norms", (rows,), torch.float32, c.device)
            _native.row_pass(c, rows, D, p=self.p, norms=norms, order=getattr(self.policy, "order", 2))
            cap = self.policy.capacity(N)
            idx = torch.empty((Bp, cap), dtype=torch.int32, device=c.device)
            fixed = self.policy.fixed_count(N)
            count = None if fixed is not None else torch.empty(Bp, dtype=torch.int32, device=c.device)
            self.policy.select_into(norms, Bp, N, idx, count)
            if count is not None:
                # Stand-alone API hands back a sized tensor: one readback (policies.py:25 => batch 1).
                assert Bp == 1, "threshold-type policies are batch-1 (policies.py:25)"
                r = int(count.item())
                idx = idx[:, :r].contiguous()
                count = None
            return idx, count, idx.long().view(tuple(lead) + (idx.shape[-1],))
        # Arbitrary user policy: give it the delta tensor, as the reference does (modules.py:149).
        index = self.policy(c - self.p, dim=-1)
        return _index_i32(index, lead, c.device), None, index


class TokenGate(_GateBase):
    """Token gate: forwards the tokens that changed most since they were last forwarded
    (modules.py:104-168).  `policy` picks them; `p` is the last-forwarded reference."""

    def forward(self, c, forced_index=None):
        _native.require_hip(c)
        if self.first:
            return self.forward_first(c)
        return self.forward_incremental(c, forced_index=forced_index)

    def forward_first(self, c):
        # Like the reference, `p` keeps a REFERENCE to the first input (modules.py:140).
        self.first = False
        self.p = c
        return c, None

    def _count_gate(self):
        if self.count_mode:
            self.counts["gate_flops"] += self.p.numel()

    def _incremental(self, c, forced_index, want_delta):
        _require_shape(type(self).__name__, "input", c.shape, self._p_shape())
        self._count_gate()
        if self.structure == "row" and _rows_fast(c, self.p):
            idx, count, index = self._select_rows(c, forced_index)
            lead, N, D = c.shape[:-2], c.shape[-2], c.shape[-1]
            Bp, cap = idx.shape
            c_t = torch.empty(tuple(lead) + (cap, D), dtype=torch.float32, device=c.device)
            e_t = torch.empty_like(c_t) if want_delta else None
            if cap:
                _native.gate_gather_update(c, self.p, idx, count, Bp, N, D, cap, c_tilde=c_t, e_tilde=e_t, update_p=True)
            return c_t, e_t, index
        return self._incremental_any(c, forced_index, want_delta, update_p=True)

    def _incremental_any(self, c, forced_index, want_delta, update_p):
        """Any other structure / element type / layout: the policy sees the delta tensor as in the reference (modules.py:149), the
        gather, the delta and the reference update run on evt_gate_rows_any / evt_gate_cols (fp32 / bf16 / fp16)."""
        row = self.structure == "row"
        if c.dtype not in _native._STORE_OF:
            raise NotImplementedError(f"MI355X build: gates take float32 / bfloat16 / float16 tensors, not {c.dtype}")
        index = forced_index if forced_index is not None else self.policy(c - self.p, dim=(-1 if row else -2))
        cc = c if c.is_contiguous() else c.contiguous()
        if not self.p.is_contiguous() or self.p.dtype != c.dtype:   # (the first frame kept a reference to its input, whatever its layout)
            self.p = self.p.to(c.dtype).contiguous()
        lead, R, L = tuple(c.shape[:-2]), c.shape[-2], c.shape[-1]
        idx = _index_i32(index, lead, c.device)
        Bp, k = idx.shape
        shape = lead + ((k, L) if row else (R, k))
        c_t = torch.empty(shape, dtype=c.dtype, device=c.device)
        e_t = torch.empty(shape, dtype=c.dtype, device=c.device) if want_delta else None
        if row:
            _native.gate_rows_any(cc, self.p, idx, None, Bp, R, L, k, c_tilde=c_t, e_tilde=e_t, update_p=update_p)
        else:
            _native.gate_cols(cc, self.p, idx, None, Bp, R, L, k, c_tilde=c_t, e_tilde=e_t, update_p=update_p)
        return c_t, e_t, index

    def forward_incremental(self, c, forced_index=None):
        c_t, _, index = self._incremental(c, forced_index, want_delta=False)
        return c_t, index


class TokenDeltaGate(TokenGate):
    """Token gate that also returns the delta of the forwarded tokens (modules.py:171-201)."""

    def forward_first(self, c):
        c = super().forward_first(c)[0]
        return c, None, None

    def forward_incremental(self, c, forced_index=None):
        return self._incremental(c, forced_index, want_delta=True)


class SimpleSTGTGate(_GateBase):
    """Baseline gate of "Spatio-Temporal Gated Transformers": the reference is the PREVIOUS INPUT,
    replaced wholesale every frame (modules.py:6-49)."""

    def __init__(self, structure="row"):
        assert structure == "row"
        super().__init__(structure=structure)

    def forward(self, c):
        _native.require_hip(c)
        if self.first:
            return self.forward_first(c)
        return self.forward_incremental(c)

    def forward_first(self, c):
        self.first = False
        self.p = c
        return c, None

    def forward_incremental(self, c):
        _require_shape(type(self).__name__, "input", c.shape, self._p_shape())
        if self.count_mode:
            self.counts["gate_flops"] += c.numel()
        if _rows_fast(c, self.p):
            idx, count, index = self._select_rows(c, None)
            lead, N, D = c.shape[:-2], c.shape[-2], c.shape[-1]
            Bp, cap = idx.shape
            c_t = torch.empty(tuple(lead) + (cap, D), dtype=torch.float32, device=c.device)
            if cap:
                _native.gate_gather_update(c, None, idx, count, Bp, N, D, cap, c_tilde=c_t, update_p=False)
        else:
            c_t, _, index = TokenGate._incremental_any(self, c, None, False, update_p=False)
        self.p = c
        return c_t, index


class TokenBuffer(ExtendedModule):
    """Token buffer: holds the latest output for every token; gated tokens overwrite their rows
    (modules.py:52-101).  forward() returns a reference to the state `b`."""

    def __init__(self, structure="row"):
        super().__init__()
        assert structure in ["row", "col"]
        self.structure = structure
        self.first = True
        self.b = None

    def forward(self, x, index):
        _native.require_hip(x)
        if self.first:
            return self.forward_first(x)
        return self.forward_incremental(x, index)

    def forward_first(self, x):
        self.first = False
        self.b = x.clone()
        return self.b

    def forward_incremental(self, x, index):
        k = index.shape[-1]
        want = tuple(self.b.shape[:-2]) + ((k, self.b.shape[-1]) if self.structure == "row" else (self.b.shape[-2], k))
        _require_shape(type(self).__name__, f"gated tokens ({k} indices)", x.shape, want)
        if self.structure == "row" and _rows_fast(x, self.b):
            lead, N, F = self.b.shape[:-2], self.b.shape[-2], self.b.shape[-1]
            idx = _index_i32(index, lead, x.device)
            _native.scatter_rows(x, self.b, idx, None, idx.shape[0], N, F, idx.shape[1])
        else:   # any other structure / element type / layout: evt_move_rows_any / evt_scatter_cols
            if self.b.dtype not in _native._STORE_OF:
                raise NotImplementedError(f"MI355X build: buffers hold float32 / bfloat16 / float16 tensors, not {self.b.dtype}")
            if not self.b.is_contiguous():
                self.b = self.b.contiguous()
            lead, R, L = tuple(self.b.shape[:-2]), self.b.shape[-2], self.b.shape[-1]
            idx = _index_i32(index, lead, x.device)
            xc = x.to(self.b.dtype)
            xc = xc if xc.is_contiguous() else xc.contiguous()
            if self.structure == "row":
                _native.move_rows_any(xc, idx, idx.shape[0], R, L, idx.shape[1], self.b, scatter=True)
            else:
                _native.scatter_cols(xc, self.b, idx, None, idx.shape[0], R, L, idx.shape[1])
        return self.b

    def reset_self(self):
        self.first = True
        self.b = None


class MatmulBuffer(ExtendedModule):
    """Query-key product state (modules.py:204-252): rows `index_q` and columns `index_k` of
    `product` are recomputed from the (already updated) q and k; the result is exact (I1)."""

    def __init__(self):
        super().__init__()
        self.first = True
        self._product = None
        self._refresh = None
        self.matmul = CountedMatmul()

    # `product` is the reference's state attribute.  On gated frames EventfulBlock may compute the scores inside the
    # fused attention kernel without touching this tensor; it then registers a refresh (one K4 launch over the
    # current token buffer) that runs when -- and only if -- somebody reads the state.
    @property
    def product(self):
        if self._refresh is not None:
            fn, self._refresh = self._refresh, None
            fn()
        return self._product

    @product.setter
    def product(self, value):
        self._product = value
        self._refresh = None

    def defer(self, refresh):
        """Mark the state stale; `refresh()` recomputes it in place on the next read."""
        self._refresh = refresh

    def forward(self, q, k, index_q, index_k):
        """q: (B,H,Nq,dh) (already divided by the scale); k: (B,H,dh,Nk) as in the reference."""
        _native.require_hip(q, k)
        if self.first:
            return self.forward_first(q, k)
        return self.forward_incremental(q, k, index_q, index_k)

    @staticmethod
    def _operands(q, k):
        q = q.float().contiguous()
        k_rows = k.float().transpose(-2, -1).contiguous()
        return q, k_rows

    def forward_first(self, q, k):
        self.first = False
        qc, kr = self._operands(q, k)
        B, H, Nq, dh = qc.shape
        Nk = kr.shape[2]
        self.matmul.count_product(B * H * Nq * Nk, dh)
        self.product = torch.empty((B, H, Nq, Nk), dtype=torch.float32, device=q.device)
        _native.qk_strided(qc, kr, self.product, 1.0)
        return self.product

    def forward_incremental(self, q, k, index_q, index_k):
        qc, kr = self._operands(q, k)
        B, H, Nq, dh = qc.shape
        Nk = kr.shape[2]
        _require_shape(type(self).__name__, "q . k^T", (B, H, Nq, Nk), self.product.shape)
        iq = _index_i32(index_q, (B,), q.device)
        ik = _index_i32(index_k, (B,), q.device)
        self.matmul.count_product(B * H * iq.shape[1] * Nk, dh)
        self.matmul.count_product(B * H * Nq * ik.shape[1], dh)
        _native.qk_strided(qc, kr, self.product, 1.0, idx_q=iq, kcap_q=iq.shape[1], idx_k=ik, kcap_k=ik.shape[1])
        return self.product

    def reset_self(self):
        self.first = True
        self.product = None


class MatmulDeltaAccumulator(ExtendedModule):
    """Attention-value product state, updated from the gated columns of A and rows of v
    (modules.py:255-299).  An approximation by design -- reproduced, not fixed."""

    def __init__(self):
        super().__init__()
        self.first = True
        self.product = None
        self.matmul = CountedMatmul()

    def forward(self, a_n_tilde, v_n_tilde, a_delta_tilde, v_delta_tilde):
        _native.require_hip(a_n_tilde, v_n_tilde)
        if self.first:
            return self.forward_first(a_n_tilde, v_n_tilde)
        return self.forward_incremental(a_n_tilde, v_n_tilde, a_delta_tilde, v_delta_tilde)

    @staticmethod
    def _merged(v):
        # (B,H,K,dh) -> (B,K,H*dh): the head-merged layout K6 consumes
        B, H, K, dh = v.shape
        return v.permute(0, 2, 1, 3).reshape(B, K, H * dh).contiguous()

    def _view(self, merged, B, H, N, dh):
        return merged.view(B, N, H, dh).permute(0, 2, 1, 3)

    def forward_first(self, a, v):
        self.first = False
        B, H, N, K = a.shape
        dh = v.shape[-1]
        store = _native.store_code(a.dtype)
        self.matmul.count_product(B * H * N * dh, K)
        self._state = torch.empty((B, N, H * dh), dtype=a.dtype, device=a.device)
        _native.av(a.contiguous(), self._merged(v.to(a.dtype)), K, B, H, N, K, H * dh, store, pv=self._state)
        self.product = self._view(self._state, B, H, N, dh)
        return self.product

    def forward_incremental(self, a_n_tilde, v_n_tilde, a_delta_tilde, v_delta_tilde):
        B, H, N, K = a_n_tilde.shape
        dh = v_n_tilde.shape[-1]
        _require_shape(type(self).__name__, "A . v", (B, H, N, dh), self.product.shape)
        _require_shape(type(self).__name__, "gated value rows", v_n_tilde.shape, (B, H, K, dh))
        if self.count_mode:
            self.counts["accumulator_flops"] += v_n_tilde.numel() + 2 * self.product.numel()
        self.matmul.count_product(B * H * N * dh, K)
        self.matmul.count_product(B * H * N * dh, K)
        dt = self._state.dtype
        store = _native.store_code(dt)
        v_old = v_n_tilde - v_delta_tilde
        _native.av(a_n_tilde.to(dt).contiguous(), self._merged(v_delta_tilde.to(dt)), K, B, H, N, K, H * dh, store,
                   pv=self._state, a2=a_delta_tilde.to(dt).contiguous(), v2=self._merged(v_old.to(dt)), gated=True)
        return self.product

    def reset_self(self):
        self.first = True
        self.product = None
        self._state = None
