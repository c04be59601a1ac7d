"""Transformer blocks (API of the reference's blocks.py) driving the gfx950 kernels.

Class names, constructor kwargs, sub-module attribute names and state_dict keys are the
reference's (blocks.py:35-47,404), because `ViTBackbone` instantiates blocks by
class NAME from a config string (backbones.py:46-59) and checkpoints are loaded by key.

Each forward is a fixed sequence of C-ABI calls on the current HIP stream -- no host sync, no
shape-dependent allocation after the first frame of a clip except the returned tensor:

  frame t >= 1, EventfulBlock                                        kernel
  ---------------------------------------------------------------   -------------------------------
  [+pos-enc] LN1 + ||c - p||                                          K1a  evt_row_pass
  top-k / threshold                                                   K1   evt_select_*
  QKV on gathered rows -> qkv buffer rows, refresh p                  K3   evt_gated_linear
  q.k^T state: rows + cols idx                                        K4   evt_qk
  [rel-pos] softmax, gather cols idx, A delta gate                    K5   evt_softmax_gate
  v delta gate                                                        K6a  evt_v_gate
  A.v state += a~.dv~ + da~.v_old ; merge heads                       K6   evt_av
  ||attn - p||, select, projection -> buffer rows                     K1a, K1, K3
  +skip, LN2 + ||c - p||, select                                      K1a, K1
  MLP on gathered rows -> buffer rows                                 K7   evt_gated_mlp
  +skip                                                               K1a

Per-clip state lives on the gate / buffer sub-modules under the reference's attribute names
(`qkv_gate.p`, `qkv_accumulator.b`, `matmul_accumulator_1.product`, ...), so `reset()` and state
inspection work as in the reference.  `pool_size` (K/V token pooling, blocks.py:303-326,525-540) is
supported on global blocks: keys/values are pooled by `evt_pool_kv`, the gate's index list is mapped to
pooled cells by `evt_pool_index`, and K4/K5/K6 run with Nq != Nk.  `ats_fraction` (adaptive token sampling,
blocks.py:150-181) runs off the fast path (`_ats_attention`) and, like the reference, only when batch == heads.
Not implemented: pooling inside windowed blocks and ATS combined with pooling / windows (raise).
"""
from math import prod, sqrt

import torch
import torch.nn as nn

from eventful_transformer import _native
from eventful_transformer.base import ExtendedModule, numeric_tuple
from eventful_transformer.counting import CountedAdd, CountedLinear, CountedMatmul
from eventful_transformer.modules import (
    MatmulBuffer,
    MatmulDeltaAccumulator,
    SimpleSTGTGate,
    TokenBuffer,
    TokenDeltaGate,
    TokenGate,
)
from eventful_transformer.policies import _NormPolicy
from eventful_transformer.utils import DropPath, RelativePositionEmbedding

LN_EPS = 1e-6


# q.k^T state update: above this fraction of changed state entries the whole product is recomputed instead of the
# row + column panels (measured break-even ~0.68 of the entries, i.e. k/N ~ 0.43)
QK_FULL_RATIO = 0.7
# The fused attention kernels emit the projection gate's delta norm per head from their epilogue, and with a bf16 cast the
# projection reads the A.v state (no fp32 attention output).  Windowed / dense blocks (resident K8) emit the same norm from
# FUSE_DENSE_NORM_ROWS token rows per launch on (one stream: 1.84 vs 1.81 ms per 672^2 frame -- the epilogue's reference reads and
# the 12-partial selection cost what the 6 us row pass did; eight streams 6.37 vs 6.41 ms).
FUSE_DENSE_NORM_ROWS = 8192
PROJ_FROM_STATE = True   # (module constant: tests/big_tile_check.py turns it off to compare with the fp32-output path, bit for bit)
# Module constant, not an environment switch: the tests set it to False to run pooled blocks with > 256 tokens on the K4 + K5+K6
# chain (the path of the grids whose tile does not fit evt_attention_stream) and compare.
STREAM_POOLED = True
# Diagnostic tap: a callable (block, gate tag, idx (B,cap) int32, count or None) invoked after every fused gate selection
# with the DEVICE index list (scratch memory: clone to keep).  bench.py's self-check records the timed run's index sets
# through it -- unlike forward hooks it leaves the launch sequence (block chaining included) untouched.
INDEX_TAP = None


def _norm_order(gate):
    """The `order` of the gate's norm policy (policies.py:11,44,76); 2 for anything that is not a norm policy."""
    return getattr(gate.policy, "order", 2)


def _PREFETCH_MAP(blk, nxt):
    """gate tag -> the linears whose weight planes its selection launch prefetches: three to five launches ahead.  Measured worse:
    one launch ahead (the riders of two linears outlast the selection); a second range with the gate reference the next row pass
    compares against (5.4 MB, written a frame ago): 672^2 1.65 vs 1.62 ms, 1024^2 2.55 vs 2.51."""
    return {"qkv": (blk.mlp_1,), "projection": (blk.mlp_2,), "mlp": (None if nxt is None else nxt.qkv,)}


class PendingSum:
    """Block output handed to the next block as an UNEVALUATED residual sum `src + res` (backbone-internal).

    The last step of a block is `x_out = mlp_buffer + x_mid` and the first step of the next block reads x_out once to
    normalise it; `ViTBackbone` lets consecutive eventful blocks pass (mlp_buffer, x_mid) instead, and the next block's
    first row pass does the add, the LayerNorm and the delta norm in one sweep (writing x_out on the way, as its skip
    connection): one HBM read of x_out and one kernel launch per block boundary less.  Never visible to callers of
    the backbone; blocks called directly always take and return tensors."""

    __slots__ = ("src", "res")

    def __init__(self, src, res):
        self.src, self.res = src, res

    @property
    def shape(self):
        return self.src.shape

    def materialize(self):
        out = torch.empty_like(self.src)
        B, N, D = self.src.shape
        _native.row_pass(self.src, B * N, D, res=self.res, sum_out=out)
        return out


def _window_map(input_size, window_size, device):
    """(windows, window_len) int32: clip-row index of each window token, -1 for padding.

    Window order and in-window order follow Block._partition_windows of the reference
    (blocks.py:257-301): windows row-major over the padded grid, tokens row-major inside a window."""
    h, w = input_size
    d0, d1 = window_size
    th, tw = h + (-h % d0), w + (-w % d1)
    ys = torch.arange(th).view(th // d0, 1, d0, 1)
    xs = torch.arange(tw).view(1, tw // d1, 1, d1)
    rows = torch.where((ys < h) & (xs < w), ys * w + xs, torch.full((1,), -1, dtype=torch.long))
    return rows.reshape(-1, d0 * d1).to(device=device, dtype=torch.int32).contiguous()


class Block(ExtendedModule):
    """Dense pre-LN Transformer block (blocks.py:26-137): LN -> QKV -> MHSA (optionally windowed
    and/or with decomposed relative position) -> projection -> +skip -> LN -> MLP(GELU) -> +skip."""

    def __init__(self, dim, heads, input_size, mlp_ratio, ats_fraction=None, drop_path_rate=0.0,
                 relative_embedding_size=None, matmul_2_cast=None, pool_size=None, window_size=None):
        super().__init__()
        if ats_fraction is not None:
            assert not (ats_fraction < 0.0 or ats_fraction > 1.0)
            # the reference asserts the same (blocks.py:71-73)
            assert pool_size is None and window_size is None, "ats_fraction excludes pool_size and window_size"
        assert not (drop_path_rate < 0.0 or drop_path_rate > 1.0)
        assert matmul_2_cast in [None, "float16", "bfloat16"]
        self.dim = dim
        self.heads = heads
        self.input_size = tuple(input_size)
        self.ats_fraction = ats_fraction
        self.last_ats_indices = None
        self.matmul_2_cast = matmul_2_cast
        self.pool_size = None if pool_size is None else numeric_tuple(pool_size, length=2)
        if window_size is None:
            self.window_size = None
            attention_size = self.input_size
        else:
            self.window_size = numeric_tuple(window_size, length=2)
            attention_size = self.window_size
            if relative_embedding_size is not None:
                relative_embedding_size = self.window_size  # blocks.py:90-91
        self.scale = sqrt(dim // heads)

        self.input_layer_norm = nn.LayerNorm(dim, eps=LN_EPS)
        self.qkv = CountedLinear(in_features=dim, out_features=dim * 3)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        if relative_embedding_size is not None:
            self.relative_position = RelativePositionEmbedding(attention_size, relative_embedding_size, dim // heads,
                                                               pool_size=self.pool_size)
        else:
            self.relative_position = None
        self.matmul = CountedMatmul()
        self.projection = CountedLinear(in_features=dim, out_features=dim)
        self.add = CountedAdd()
        self.mlp_layer_norm = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp_1 = CountedLinear(in_features=dim, out_features=dim * mlp_ratio)
        self.gelu = nn.GELU()
        self.mlp_2 = CountedLinear(in_features=dim * mlp_ratio, out_features=dim)
        self._wmap = None
        self._norm_fusable = {}

    # ---------------------------------------------------------------------------------------------
    # helpers shared by all block classes
    # ---------------------------------------------------------------------------------------------
    def reset_self(self):
        self.last_ats_indices = None
        self._last_ats_i32 = None

    def _store_dtype(self):
        return torch.float32 if self.matmul_2_cast is None else getattr(torch, self.matmul_2_cast)

    def _check_input(self, x):
        _native.require_hip(x)
        if self.training and isinstance(self.drop_path, DropPath):
            raise RuntimeError("MI355X path is inference-only: call .eval() (DropPath is train-time only)")
        if x.ndim != 3 or x.shape[-1] != self.dim:
            raise RuntimeError(f"block input must be (batch, tokens, {self.dim}), got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise RuntimeError(f"block input must be float32, got {x.dtype}")
        if (self.window_size is not None or self.relative_position is not None or self.pool_size is not None) \
                and x.shape[1] != self.input_size[0] * self.input_size[1]:
            # windows, rel-pos terms and pooled cells are laid out on the input grid (blocks.py:257-326 reshape to it and fail)
            raise RuntimeError(f"block input has {x.shape[1]} tokens but input_size = {tuple(self.input_size)}")
        return x if x.is_contiguous() else x.contiguous()

    def _compute_window_padding(self):
        return (-self.input_size[0] % self.window_size[0], -self.input_size[1] % self.window_size[1])

    def _windows(self, device):
        if self._wmap is None or self._wmap.device != device:
            self._wmap = _window_map(self.input_size, self.window_size, device)
        return self._wmap

    def _rel_tables(self):
        """(rel_y, rel_x, key-grid height, key-grid width, query-grid width) for K5; tables are (q, k, dh)."""
        if self.relative_position is None:
            return None, None, 0, 0, 0
        ry, rx = self.relative_position.tables()
        return ry, rx, ry.shape[1], rx.shape[1], rx.shape[0]

    def _pool_kv(self, qkv, B, N):
        """K/V token pooling (blocks.py:303-326).  Returns (kv (B,Nk,2D) or None, Nk)."""
        if self.pool_size is None:
            return None, N
        qh, qw = self.input_size
        p0, p1 = self.pool_size
        assert qh * qw == N, "token pooling needs tokens == prod(input_size) (no class token), as in the reference"
        Nk = (qh // p0) * (qw // p1)
        kv = self._ws("pooled_kv", (B, Nk, 2 * self.dim), torch.float32, qkv)
        _native.pool_kv(qkv, B, qh, qw, self.dim, p0, p1, kv)
        return kv, Nk

    def _v_full(self, qkv, kv, G, n, Nk, v_s, store, **win):
        """K6a FULL: value state from the packed buffer, or from the pooled buffer's value half."""
        D = self.dim
        if kv is None:
            _native.v_gate(qkv, None, None, G, n, D, 0, v_s, None, None, store, False, **win)
        else:
            _native.v_gate(kv, None, None, G, Nk, D, 0, v_s, None, None, store, False, v_offset=D, v_rs=2 * D)

    def _ws(self, name, shape, dtype, x):
        return _native.scratch(name, shape, dtype, x.device)

    def _count_add(self, n):
        if self.add.count_mode:
            self.add.counts["add_flops"] += n

    def _ln(self, which):
        ln = self.input_layer_norm if which == 1 else self.mlp_layer_norm
        return ln.weight, ln.bias

    # -- adaptive token sampling (blocks.py:150-181, 378-391, 196-203) --------------------------------
    # Off the fast path: the q.k^T / softmax / A.v contractions run on K4 / K5 / K6, the token scoring on evt_ats_scores, the
    # selection on evt_select_topk, the stabilisation on evt_ats_stabilize, the row gathers on evt_gather_rows_map /
    # evt_move_rows_any -- and the selection stays on the device (the reference moves it to the CPU for
    # `_stabilize_ats_indices`).  The reference sums its scores over the BATCH axis (`scores.sum(dim=-3)` on a (B, H, N)
    # tensor, blocks.py:163), so -- like the reference -- this only runs when batch == heads: clip b then keeps the tokens ranked by
    # head b's scores summed over all clips.
    def _ats_select(self, a, v):
        """a (B,H,N,N) probabilities, v (B,H,N,dh) (both already in the dtype the reference scores in) -> (H, n) int64."""
        B, H, N = a.shape[0], a.shape[1], a.shape[2]
        if B != H:
            raise RuntimeError(f"ats_fraction: the reference's adaptive token sampling sums its scores over the batch axis "
                               f"(blocks.py:163) and only runs when batch == heads; got batch {B}, heads {H}")
        scores = self._ws("ats_scores", (H, N), torch.float32, a)
        _native.ats_scores(a if a.is_contiguous() else a.contiguous(), v, B, H, N, v.shape[-1], scores)
        n_select = int(self.ats_fraction * (N - 1)) + 1
        now = self._ws("ats_now", (H, n_select), torch.int32, a)
        _native.select_topk(scores, H, N, n_select, now)          # ascending lists (blocks.py:380 sorts them)
        index32 = torch.empty((H, n_select), dtype=torch.int32, device=a.device)
        last = self.__dict__.get("_last_ats_i32")
        if last is not None and self.last_ats_indices is not None:   # keep every surviving token at last frame's position (blocks.py:378-391)
            _native.ats_stabilize(last, now, H, n_select, N, index32)
        else:
            index32.copy_(now)
        self._last_ats_i32 = index32
        self.last_ats_indices = index32.long()
        return self.last_ats_indices

    def _ats_rows(self, x, index):
        """x (B, N, F) -> rows index[b] of clip b: (B, n, F) (`_gather_ats_skip`, blocks.py:196-203)."""
        B, N, F = x.shape
        n = index.shape[-1]
        out = torch.empty((B, n, F), dtype=x.dtype, device=x.device)
        _native.gather_rows_map(x if x.is_contiguous() else x.contiguous(), self._last_ats_i32, B, N, F, n, out, map_per_batch=True)
        return out

    def _heads_v(self, qkv, B, N):
        """Value heads (B,H,N,dh) as a view of the packed (B,N,3D) buffer (blocks.py:248-255)."""
        D, H = self.dim, self.heads
        return qkv.view(B, N, 3, H, D // H)[:, :, 2].permute(0, 2, 1, 3)

    def _ats_attention(self, product, qkv, B, N, eventful):
        """Attention tail with ATS from the (B,H,N,N) score state `product`: softmax (K5) -> scores -> selection ->
        A.v.  Returns (attention output (B, n, D) fp32, indices (H, n)).
        eventful=False (Block / EventfulTokenwiseBlock / EventfulMatmul1Block): ATS scores in fp32, before the
        matmul_2 cast; the selected rows of a.v are rows of the full product (K6 over all rows, then a gather).
        eventful=True (EventfulBlock): cast first, then ATS, then the v / A delta gates and the delta accumulator on
        the gathered rows, with the stand-alone gate / accumulator modules (blocks.py:558-575)."""
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        ry, rx, gh, gw, qw = self._rel_tables()
        a32 = self._ws("ats_probs", (B, H, N, N), torch.float32, qkv)
        _native.softmax_gate(product, a32, B, H, N, N, D, _native.EVT_F32, qkv=qkv, rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw)
        v = self._heads_v(qkv, B, N)
        if not eventful:
            index = self._ats_select(a32, v)
            a = a32 if sdt == torch.float32 else a32.to(sdt)
            v_s = self._ws("attn_values", (B, N, D), sdt, qkv)
            self._v_full(qkv, None, B, N, N, v_s, _native.store_code(sdt))
            full = self._ws("attn_out", (B, N, D), torch.float32, qkv)
            _native.av(a, v_s, N, B, H, N, N, D, _native.store_code(sdt), out_f32=full)
            self.matmul.count_product(B * H * index.shape[-1] * dh, N)
            return self._ats_rows(full, index), index
        a = a32 if sdt == torch.float32 else a32.to(sdt)
        v = v.to(sdt) if sdt != torch.float32 else v.clone()
        index = self._ats_select(a, v)
        n_sel = index.shape[-1]
        a_rows = torch.empty((B, H, n_sel, N), dtype=a.dtype, device=a.device)
        _native.move_rows_any(a if a.is_contiguous() else a.contiguous(), self._last_ats_i32, B * H, N, N, n_sel, a_rows, rep=H)   # rows index[b] of every head of clip b
        a = a_rows
        idx_v = None if self._ats_idx_k is None else self._ats_idx_k
        v_n, v_d, index_v = self.v_gate(v, forced_index=idx_v)
        a_n, a_d, _ = self.matmul_gate(a, forced_index=index_v)
        x = self.matmul_accumulator_2(a_n, v_n, a_d, v_d)
        x = x.permute(0, 2, 1, 3).reshape(B, index.shape[-1], D)
        return x.float().contiguous(), index

    # -- dense attention (also the windowed attention of ViTDet's EventfulTokenwiseBlocks) -----------
    def _attention_dense(self, qkv, B, N, out, norm=None):
        """qkv (B,N,3D) -> out (B,N,D) fp32.  Block._forward_attention (blocks.py:205-240).
        norm = (reference (B,N,D), partials (B,N,H)): the launch also emits, per token and head, || out - reference ||^2 (the
        caller has checked `_dense_norm_fusable`: the resident K8 kernel runs)."""
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        ry, rx, gh, gw, qw = self._rel_tables()
        if self.window_size is None:
            G, n, tok_map, gpc, pad = B, N, None, 1, None
        else:
            tok_map = self._windows(qkv.device)
            gpc, n = tok_map.shape
            G, pad = B * gpc, self.qkv.bias
            assert prod(self.input_size) == N, "windowed attention needs tokens == prod(input_size)"
        if self.ats_fraction is not None:
            prod_s = self._ws("attn_scores", (B, H, N, N), torch.float32, qkv)
            _native.qk_packed(qkv, B, N, D, H, self.scale, prod_s)
            self.matmul.count_product(B * H * N * N, dh)
            if self.relative_position is not None:
                self.relative_position.count_fused(B, H)
            return self._ats_attention(prod_s, qkv, B, N, eventful=False)
        if self.pool_size is not None and self.window_size is not None:
            return self._attention_window_pooled(qkv, B, N, out, tok_map)
        if self.pool_size is None and _native.attention_dense_fits(n, D, H) and _native.DENSE_FUSED:
            # K8: the whole group in one launch, no score / probability tensors in HBM
            _native.attention_dense(qkv, G, H, n, D, self.scale, store, out_f32=out, rel_y=ry, rel_x=rx, gh=gh, gw=gw,
                                    qw=qw, tok_map=tok_map, groups_per_clip=gpc, clip_rows=N, pad_row=pad,
                                    norm_ref=None if norm is None else norm[0], norm_parts=None if norm is None else norm[1])
            self.matmul.count_product(G * H * n * n, dh)
            if self.relative_position is not None:
                self.relative_position.count_fused(G, H)
            self.matmul.count_product(G * H * n * dh, n)
            return
        kv, nk = self._pool_kv(qkv, B, N) if self.window_size is None else (None, n)
        win = dict(tok_map=tok_map, groups_per_clip=gpc, clip_rows=N, pad_row=pad)
        prod_s = self._ws("attn_scores", (G, H, n, nk), torch.float32, qkv)
        a_s = self._ws("attn_probs", (G, H, n, nk), sdt, qkv)
        v_s = self._ws("attn_values", (G, nk, D), sdt, qkv)
        _native.qk_packed(qkv, G, n, D, H, self.scale, prod_s, kv=kv, Nk=nk, **win)
        self.matmul.count_product(G * H * n * nk, dh)
        if self.relative_position is not None:
            self.relative_position.count_fused(G, H)
        _native.softmax_gate(prod_s, a_s, G, H, n, nk, D, store, qkv=qkv, rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw, **win)
        self._v_full(qkv, kv, G, n, nk, v_s, store, **win)
        _native.av(a_s, v_s, nk, G, H, n, nk, D, store, out_f32=out, out_map=tok_map, groups_per_clip=gpc, clip_rows=N)
        self.matmul.count_product(G * H * n * dh, nk)

    def _attention_window_pooled(self, qkv, B, N, out, tok_map):
        """window_size together with pool_size (Block._forward_attention, blocks.py:205-240: `_partition_windows` in the qkv domain,
        then `_pool_tokens` on the WINDOW grid, blocks.py:308).  No reference config uses the combination, so it is kept simple:
        the windows are gathered into contiguous groups (padding tokens = the qkv bias row, blocks.py:270-283;
        evt_gather_rows_map / evt_scatter_rows_map), every window is pooled like a small clip (evt_pool_kv) and runs through the q.k^T / softmax / A.v launches with
        pooled keys; the rows are copied back to their tokens."""
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        ry, rx, gh, gw, qw = self._rel_tables()
        gpc, n = tok_map.shape
        G = B * gpc
        d0, d1 = self.window_size
        p0, p1 = self.pool_size
        assert d0 % p0 == 0 and d1 % p1 == 0, "token pooling needs a window size divisible by the pool size, as in the reference"
        nk = (d0 // p0) * (d1 // p1)
        # windows as contiguous groups: rows through the window map, padding tokens = the qkv bias row (blocks.py:270-283)
        wq = self._ws("window_qkv", (B, gpc * n, 3 * D), torch.float32, qkv)
        _native.gather_rows_map(qkv, tok_map, B, N, 3 * D, gpc * n, wq, pad_row=self.qkv.bias)
        wq = wq.view(G, n, 3 * D)
        kv = self._ws("pooled_kv", (G, nk, 2 * D), torch.float32, qkv)
        _native.pool_kv(wq, G, d0, d1, D, p0, p1, kv)
        prod_s = self._ws("attn_scores", (G, H, n, nk), torch.float32, qkv)
        a_s = self._ws("attn_probs", (G, H, n, nk), sdt, qkv)
        v_s = self._ws("attn_values", (G, nk, D), sdt, qkv)
        out_w = self._ws("window_out", (B, gpc * n, D), torch.float32, qkv)
        _native.qk_packed(wq, G, n, D, H, self.scale, prod_s, kv=kv, Nk=nk)
        self.matmul.count_product(G * H * n * nk, dh)
        if self.relative_position is not None:
            self.relative_position.count_fused(G, H)
        _native.softmax_gate(prod_s, a_s, G, H, n, nk, D, store, qkv=wq, rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw)
        self._v_full(wq, kv, G, n, nk, v_s, store)
        _native.av(a_s, v_s, nk, G, H, n, nk, D, store, out_f32=out_w.view(G, n, D))
        self.matmul.count_product(G * H * n * dh, nk)
        _native.scatter_rows_map(out_w, tok_map, B, gpc * n, N, D, out)   # un-window: padding rows dropped (blocks.py:346-376)

    def _dense_norm_fusable(self, N):
        """True when `_attention_dense` runs the resident K8 kernel for this block (windowed or one group of <= 256 tokens, head dim
        64, no pooling / ATS, planes within a CU's LDS): the kernel that can emit the next gate's per-head delta norms."""
        if self.ats_fraction is not None or self.pool_size is not None or not _native.DENSE_FUSED:
            return False
        key = (N, _native.QK_SPLIT)
        hit = self._norm_fusable.get(key)
        if hit is None:
            n = N if self.window_size is None else prod(self.window_size)
            _, _, gh, gw, _ = self._rel_tables()
            hit = _native.attention_dense_fits(n, self.dim, self.heads) and \
                _native.attention_dense_resident(n, self.dim, self.heads, _native.store_code(self._store_dtype()), gh, gw)
            self._norm_fusable[key] = hit
        return hit

    def _dense_tail(self, attn, skip, B, N):
        """projection + skip + LN2 + MLP + skip over all tokens (Block.forward, blocks.py:127-137)."""
        D, rows = self.dim, B * N
        proj = self._ws("dense_proj", (B, N, D), torch.float32, attn)
        _native.gated_linear(attn, D, None, rows, self.projection.weight, self.projection.bias, proj, D, None, rows,
                             None, None, 1, rows, D, D, W_split=self.projection.split_planes())
        self.projection.count_rows(rows)
        x2 = self._ws("x_mid", (B, N, D), torch.float32, attn)
        c = self._ws("gate_in", (B, N, D), torch.float32, attn)
        w, b = self._ln(2)
        _native.row_pass(proj, rows, D, res=skip, sum_out=x2, ln_w=w, ln_b=b, eps=LN_EPS, c_out=c)
        self._count_add(rows * D)
        Dh = self.mlp_1.out_features
        hidden = self._ws("mlp_hidden", (rows, Dh), torch.float32, attn)
        mlp = self._ws("dense_mlp", (B, N, D), torch.float32, attn)
        _native.gated_mlp(c, D, None, rows, self.mlp_1.weight, self.mlp_1.bias, self.mlp_2.weight, self.mlp_2.bias,
                          hidden, mlp, D, None, None, 1, rows, D, Dh, W1_split=self.mlp_1.split_planes(),
                          W2_split=self.mlp_2.split_planes())
        self.mlp_1.count_rows(rows)
        self.mlp_2.count_rows(rows)
        out = torch.empty((B, N, D), dtype=torch.float32, device=attn.device)
        _native.row_pass(mlp, rows, D, res=x2, sum_out=out)
        self._count_add(rows * D)
        return out

    def forward(self, x):
        x = self._check_input(x)
        B, N, D = x.shape
        rows = B * N
        c = self._ws("gate_in", (B, N, D), torch.float32, x)
        w, b = self._ln(1)
        _native.row_pass(x, rows, D, ln_w=w, ln_b=b, eps=LN_EPS, c_out=c)
        qkv = self._ws("dense_qkv", (B, N, 3 * D), torch.float32, x)
        _native.gated_linear(c, D, None, rows, self.qkv.weight, self.qkv.bias, qkv, 3 * D, None, rows, None, None, 1,
                             rows, D, 3 * D, W_split=self.qkv.split_planes())
        self.qkv.count_rows(rows)
        attn = self._ws("attn_out", (B, N, D), torch.float32, x)
        ats = self._attention_dense(qkv, B, N, attn)
        if ats is not None:   # adaptive token sampling: the block continues on the selected tokens only
            attn, index = ats
            return self._dense_tail(attn, self._ats_rows(x, index), B, attn.shape[1])
        return self._dense_tail(attn, x, B, N)


class EventfulTokenwiseBlock(Block):
    """Block with gated token-wise operations: three gate -> op -> buffer groups around QKV,
    projection and MLP (blocks.py:399-463).  Attention itself stays dense (and may be windowed)."""

    def __init__(self, gate_before_ln=False, stgt=False, **super_kwargs):
        super().__init__(**super_kwargs)
        self.gate_before_ln = gate_before_ln
        token_gate_class = SimpleSTGTGate if stgt else TokenGate
        self.qkv_gate = token_gate_class()
        self.qkv_accumulator = TokenBuffer()
        self.projection_gate = token_gate_class()
        self.projection_accumulator = TokenBuffer()
        self.mlp_gate = token_gate_class()
        self.mlp_accumulator = TokenBuffer()
        self._wants_rest = False   # set by subclasses that keep a q.k^T product state
        self._rest = None
        self._clip_shape = None    # (batch, tokens) of the clip's first frame: what the state tensors were created for

    def _check_clip_shape(self, bn):
        """The per-clip state (gate references, token buffers, attention states) has the first frame's (batch, tokens): the kernels
        address it with those sizes.  The reference fails in its scatter / gather with a shape error when they change without
        reset() (modules.py:90-96, 154-164); here it would be an out-of-bounds access, so it is an error up front."""
        if self.qkv_gate.first or self._clip_shape is None:   # (the gates' `first` flags are what reset() clears)
            self._clip_shape = bn
        elif self._clip_shape != bn:
            raise RuntimeError(f"{type(self).__name__}: frame of (batch, tokens) = {bn} but this clip's state was created for "
                               f"{self._clip_shape}; call reset() between clips")

    # ---------------------------------------------------------------------------------------------
    # one gate -> linear(s) -> buffer group
    # ---------------------------------------------------------------------------------------------
    def _select(self, gate, c, norms, B, N, tag, parts=0):
        """Runs the gate's policy on the norms already produced by the row pass (parts > 0: on the per-head partial
        sums of squares produced by the fused attention epilogue).  Returns (idx, count, cap)."""
        policy = gate.policy
        if isinstance(policy, _NormPolicy):
            cap = policy.capacity(N)
            idx = self._ws("idx_" + tag, (B, cap), torch.int32, c)
            count = None if policy.fixed_count(N) is not None else self._ws("cnt_" + tag, (B,), torch.int32, c)
            # the qkv gate of blocks with a q.k^T state also wants the complement list (K4 skips re-written rows)
            rest = self._ws("idx_rest", (B, N), torch.int32, c) if (tag == "qkv" and self._wants_rest) else None
            self._rest = rest if tag == "qkv" else self._rest
            if B * N <= _native.PREFETCH_MAX_ROWS:
                # one stream: the single-workgroup selection launch also pulls the weight planes of the gated linears one or two
                # launches ahead into the memory-side cache (MLP-1 behind the projection gate; MLP-2 and the NEXT block's QKV
                # behind the MLP gate -- close to their use: the attention state of a 1024^2 frame would evict them) -- a
                # frame walks 340 MB of planes, colder than any cache by the time it comes round again
                nxt = self.__dict__.get("_next_block")
                for lin in _PREFETCH_MAP(self, nxt).get(tag, ()):
                    if lin is not None:
                        _native.select_prefetch_next(lin.split_planes())
            policy.select_into(norms, B, N, idx, count, rest, parts=parts)
            if INDEX_TAP is not None:
                INDEX_TAP(self, tag, idx, count)
            return idx, count, cap
        # Any other callable gets the delta tensor like in the reference (modules.py:149).
        index = policy(c - gate.p, dim=-1)
        idx = index.reshape(B, -1).to(torch.int32).contiguous()
        if tag == "qkv":
            self._rest = None
        if idx.shape[1] == 0:
            # an empty selection (e.g. a threshold nothing exceeds): the kernels address index lists of at least one slot, so it
            # travels as a device-side count of 0 over a one-slot list -- the route of the built-in policies (policies.py)
            return torch.zeros((B, 1), dtype=torch.int32, device=c.device), torch.zeros((B,), dtype=torch.int32, device=c.device), 1
        return idx, None, idx.shape[1]

    def _count_gate(self, gate, n):
        if gate.count_mode:
            gate.counts["gate_flops"] += n

    def _n_rows(self, B, cap, count):
        """Gated row total for MAC accounting (one readback, counting mode only)."""
        return B * cap if count is None else int(count.sum().item())

    def _group(self, gate, buffer, src, res, ln, linear_fn, out_features, tag, sum_out=None, count_res=True,
               norm_parts=None, state_src=None):
        """Generic gate group.  src (+res) -> [LN] -> gate -> linear_fn on gated rows -> buffer rows.

        norm_parts / state_src (projection group only): what the fused attention kernel of THIS forward call handed over
        -- ((B,N,parts) partial squares of ||src - p||, parts) and the bf16 A.v state standing in for an unwritten `src`.
        Returns (buffer state, idx, count, cap); idx is None on the first frame of a clip."""
        B, N, D = src.shape
        rows = B * N
        stgt = isinstance(gate, SimpleSTGTGate)
        ln_w, ln_b = (None, None) if ln is None else self._ln(ln)
        if res is not None and count_res:
            self._count_add(rows * D)
        if gate.first:
            gate.first = False
            buffer.first = False
            gate.p = torch.empty((B, N, D), dtype=torch.float32, device=src.device)
            buffer.b = torch.empty((B, N, out_features), dtype=torch.float32, device=src.device)
            if self.gate_before_ln and ln is not None:
                # reference: p = x (pre-LN), the linear sees LN(x)
                c = self._ws("gate_in", (B, N, D), torch.float32, src)
                _native.row_pass(src, rows, D, res=res, sum_out=gate.p)
                if sum_out is not None:
                    sum_out.copy_(gate.p)
                _native.row_pass(gate.p, rows, D, ln_w=ln_w, ln_b=ln_b, eps=LN_EPS, c_out=c)
                a = c
            else:
                _native.row_pass(src, rows, D, res=res, sum_out=sum_out, ln_w=ln_w, ln_b=ln_b, eps=LN_EPS, c_out=gate.p)
                a = gate.p
            linear_fn(a, None, None, buffer.b, None, B, N)
            return buffer.b, None, None, N

        c = self._ws("gate_in", (B, N, D), torch.float32, src)
        norms = self._ws("gate_norms", (B, N), torch.float32, src)
        self._count_gate(gate, rows * D)
        if self.gate_before_ln and ln is not None:
            raw = self._ws("gate_raw", (B, N, D), torch.float32, src)
            _native.row_pass(src, rows, D, res=res, sum_out=sum_out, c_out=raw, p=gate.p, norms=norms, order=_norm_order(gate))
            idx, count, cap = self._select(gate, raw, norms, B, N, tag)
            _native.row_pass(raw, rows, D, ln_w=ln_w, ln_b=ln_b, eps=LN_EPS, c_out=c)
            if stgt:
                _native.row_pass(raw, rows, D, c_out=gate.p)
            else:
                _native.gate_gather_update(raw, gate.p, idx, count, B, N, D, cap, update_p=True)
            linear_fn(c, idx, count, buffer.b, None, B, cap)
        else:
            parts = 0
            src16 = None
            if ln is None and res is None:
                c = src  # the gate input already exists in HBM: only the norms are new
                src16 = state_src
                if norm_parts is not None:
                    norms, parts = norm_parts   # ||src - p||^2 per head came out of the fused attention epilogue
                else:
                    _native.row_pass(src, rows, D, p=gate.p, norms=norms, order=_norm_order(gate))
            else:
                _native.row_pass(src, rows, D, res=res, sum_out=sum_out, ln_w=ln_w, ln_b=ln_b, eps=LN_EPS, c_out=c,
                                 p=gate.p, norms=norms, order=_norm_order(gate))
            idx, count, cap = self._select(gate, c, norms, B, N, tag, parts=parts)
            if src16 is not None:   # the fp32 attention output was not written: the projection reads the bf16 A.v state
                assert parts and count is None and not stgt
                linear_fn(src16, idx, count, buffer.b, gate.p, B, cap, a_bf16=True)
            else:
                linear_fn(c, idx, count, buffer.b, None if stgt else gate.p, B, cap)
            if stgt:
                _native.row_pass(c, rows, D, c_out=gate.p)
        return buffer.b, idx, count, cap

    def _linear_fn(self, layer):
        def run(a, idx, count, out, p_upd, B, cap, a_bf16=False):
            N = out.shape[1]
            _native.gated_linear(a, layer.in_features, idx, N if idx is not None else cap, layer.weight, layer.bias,
                                 out, layer.out_features, idx, N if idx is not None else cap, count, p_upd, B, cap,
                                 layer.in_features, layer.out_features, W_split=layer.split_planes(), a_bf16=a_bf16)
            if layer.count_mode:
                layer.count_rows(self._n_rows(B, cap, count))
        return run

    def _mlp_fn(self, a, idx, count, out, p_upd, B, cap):
        N = out.shape[1]
        D, Dh = self.dim, self.mlp_1.out_features
        hidden = self._ws("mlp_hidden", (B * cap, Dh), torch.float32, a)
        _native.gated_mlp(a, D, idx, N if idx is not None else cap, self.mlp_1.weight, self.mlp_1.bias,
                          self.mlp_2.weight, self.mlp_2.bias, hidden, out, D, count, p_upd, B, cap, D, Dh,
                          W1_split=self.mlp_1.split_planes(), W2_split=self.mlp_2.split_planes())
        if self.mlp_1.count_mode:
            n = self._n_rows(B, cap, count)
            self.mlp_1.count_rows(n)
            self.mlp_2.count_rows(n)

    # ---------------------------------------------------------------------------------------------
    def _forward_pre_attention(self, x, res=None, sum_out=None):
        """LN1 -> qkv gate -> QKV -> qkv buffer (blocks.py:452-463 + :491).  With `res` the block input is the pending
        sum x + res of the previous block, written to `sum_out` by the same row pass (the add was counted there)."""
        return self._group(self.qkv_gate, self.qkv_accumulator, x, res, 1, self._linear_fn(self.qkv), 3 * self.dim,
                           "qkv", sum_out=sum_out, count_res=False)

    def _forward_post_attention(self, attn, skip, defer=False, fused=None):
        """projection group, +skip, LN2, mlp group, +skip (blocks.py:430-450).  defer: leave the last add to the
        next block's first row pass (returns a PendingSum).  fused: hand-over of the fused attention kernel
        (`norm_parts`, `state_src`; see _group), valid for this call only."""
        B, N, D = attn.shape
        fused = fused or {}
        proj, _, _, _ = self._group(self.projection_gate, self.projection_accumulator, attn, None, None,
                                    self._linear_fn(self.projection), D, "projection",
                                    norm_parts=fused.get("norm_parts"), state_src=fused.get("state_src"))
        x2 = self._ws("x_mid", (B, N, D), torch.float32, attn)
        mlp, _, _, _ = self._group(self.mlp_gate, self.mlp_accumulator, proj, skip, 2, self._mlp_fn, D, "mlp",
                                   sum_out=x2)
        self._count_add(B * N * D)
        if defer:
            return PendingSum(mlp, x2)
        out = torch.empty((B, N, D), dtype=torch.float32, device=attn.device)
        _native.row_pass(mlp, B * N, D, res=x2, sum_out=out)
        return out

    def _forward_attention(self, qkv, idx, count, cap, B, N):
        """-> (attention output, ATS indices or None, fused hand-over dict or None)"""
        attn = self._ws("attn_out", (B, N, self.dim), torch.float32, qkv)
        # The projection gate's delta norm || attn - p ||^2 comes out of the attention epilogue, per head (the select kernel adds the H
        # partials): no separate pass over the attention output -- as in the global blocks' fused attention kernels.
        pg = self.projection_gate
        norm = None
        if B * N >= FUSE_DENSE_NORM_ROWS and not pg.first and pg.p is not None and isinstance(pg.policy, _NormPolicy) and pg.policy.order == 2 \
                and self._dense_norm_fusable(N):
            norm = (pg.p, self._ws("norm_parts", (B, N, self.heads), torch.float32, qkv))
        ats = self._attention_dense(qkv, B, N, attn, norm=norm)
        if ats is not None:
            return ats[0], ats[1], None
        return attn, None, (None if norm is None else dict(norm_parts=(norm[1], self.heads), state_src=None))

    def forward(self, x, _defer_output=False):
        """_defer_output (ViTBackbone only, never with hooks registered): return the block output as a PendingSum."""
        defer = _defer_output
        self._check_clip_shape(tuple(x.shape[:2]))
        if isinstance(x, PendingSum):
            B, N, D = x.shape
            xin = self._ws("x_in", (B, N, D), torch.float32, x.src)
            qkv, idx, count, cap = self._forward_pre_attention(x.src, res=x.res, sum_out=xin)
            x = xin
        else:
            x = self._check_input(x)
            B, N, _ = x.shape
            qkv, idx, count, cap = self._forward_pre_attention(x)
        attn, ats_index, fused = self._forward_attention(qkv, idx, count, cap, B, N)
        skip = x if ats_index is None else self._ats_rows(x, ats_index)   # blocks.py:426,493
        return self._forward_post_attention(attn, skip, defer=defer and ats_index is None, fused=fused)


class EventfulMatmul1Block(EventfulTokenwiseBlock):
    """EventfulTokenwiseBlock + gated query-key product (blocks.py:466-540)."""

    def __init__(self, **super_kwargs):
        super().__init__(**super_kwargs)
        assert self.window_size is None  # blocks.py:485
        self.matmul_accumulator_1 = MatmulBuffer()
        self._wants_rest = True

    def _scores(self, qkv, idx, count, cap, B, N):
        """q.k^T state update (K4), with pooled keys when `pool_size` is set.

        Returns (product (B,H,N,Nk) fp32 state, kv or None, Nk, idx_k, count_k, cap_k): the key-side index list
        is the gate's own list, or its pooled / de-duplicated image (`_pool_index`, blocks.py:525-540)."""
        D, H = self.dim, self.heads
        acc = self.matmul_accumulator_1
        kv, Nk = self._pool_kv(qkv, B, N)
        idx_k, count_k, cap_k = idx, count, cap
        if kv is not None and idx is not None:
            cap_k = min(cap, Nk)
            # Pooled image of the index list, de-duplicated PER CLIP (blocks.py:525-540).  Batch 1 -- every reference config that pools --
            # is the reference's list exactly.  For a batch the reference's `index.unique(dim=-1)` de-duplicates COLUMNS of the (B, k)
            # tensor (whole batch vectors), which leaves duplicate cells inside a clip's row; its accumulator then adds a duplicated
            # key's delta twice (modules.py:285-295) and a clip's output depends on its neighbours (0.4-0.6 off its own batch-1 result on
            # the small golden shapes).  Here a clip in a batch gets its batch-1 result (tests/test_gpu_properties.py).
            idx_k = self._ws("idx_k", (B, cap_k), torch.int32, qkv)
            count_k = self._ws("cnt_k", (B,), torch.int32, qkv)
            p0, p1 = self.pool_size
            _native.pool_index(idx, count, B, cap, self.input_size[1], p0, p1, self.input_size[1] // p1, Nk, cap_k,
                               idx_k, count_k)
        if acc.first:
            acc.first = False
            acc.product = torch.empty((B, H, N, Nk), dtype=torch.float32, device=qkv.device)
            _native.qk_packed(qkv, B, N, D, H, self.scale, acc.product, kv=kv, Nk=Nk)
            acc.matmul.count_product(B * H * N * Nk, D // H)
        elif count is None and count_k is None and (cap * Nk + N * cap_k - cap * cap_k) > QK_FULL_RATIO * N * Nk:
            # Most of the state changes (k/N = 0.65 at the ViViT operating point: the row + column panels are 88 % of
            # the N x N state): recomputing ALL of q.k^T from the updated buffer gives the same state (I1) and is
            # cheaper than the two panels with their scattered column writes (288 vs 374 us at B = 256).
            _native.qk_packed(qkv, B, N, D, H, self.scale, acc.product, kv=kv, Nk=Nk)
            if acc.matmul.count_mode:   # the reference's count is that of the delta update (modules.py:232-247)
                acc.matmul.count_product(H * Nk * B * cap, D // H)
                acc.matmul.count_product(H * N * B * cap_k, D // H)
        else:
            _native.qk_packed(qkv, B, N, D, H, self.scale, acc.product, idx=idx, count=count, kcap=cap, kv=kv, Nk=Nk,
                              idx_k=idx_k, count_k=count_k, kcap_k=cap_k, idx_rest=self._rest)
            if acc.matmul.count_mode:
                acc.matmul.count_product(H * Nk * self._n_rows(B, cap, count), D // H)
                acc.matmul.count_product(H * N * self._n_rows(B, cap_k, count_k), D // H)
        if self.relative_position is not None:
            self.relative_position.count_fused(B, H)
        return acc.product, kv, Nk, idx_k, count_k, cap_k

    def _defer_scores(self, B, N):
        """Marks `matmul_accumulator_1.product` stale: it is recomputed (one K4 launch over the current token buffer,
        allocating the tensor if need be) when -- and only if -- somebody reads the attribute."""
        acc, owner, D, H, scale = self.matmul_accumulator_1, self.qkv_accumulator, self.dim, self.heads, self.scale

        def refresh():
            buf = owner.b
            kv, Nk = self._pool_kv(buf, B, N)   # (None, N) without pool_size
            if acc._product is None or acc._product.shape != (B, H, N, Nk):
                acc._product = torch.empty((B, H, N, Nk), dtype=torch.float32, device=buf.device)
            _native.qk_packed(buf, B, N, D, H, scale, acc._product, kv=kv, Nk=Nk)
        acc.defer(refresh)

    def _first_frame_fused(self, qkv, B, N, attn, a_state=None, pv=None):
        """First frame of a clip through K8: q.k^T state, probabilities, A.v state and the block's attention output
        from ONE launch (MatmulBuffer.forward_first, modules.py:224-230, + blocks.py:518-522 + modules.py:277-283).
        Returns False when K8 does not cover the shape (pooled keys, > 256 tokens, head dim != 64)."""
        D, H = self.dim, self.heads
        acc = self.matmul_accumulator_1
        if not (acc.first and self.pool_size is None and _native.DENSE_FUSED and _native.attention_dense_fits(N, D, H)):
            return False
        acc.first = False
        if a_state is not None and _native.FUSED_QK:
            # EventfulBlock whose gated frames compute the scores inside the fused kernel: nobody reads the q.k^T state,
            # so it is neither written here (477 MB per block at B = 256) nor even allocated until somebody asks for it
            product = None
            self._defer_scores(B, N)
        else:
            product = acc.product = torch.empty((B, H, N, N), dtype=torch.float32, device=qkv.device)
        ry, rx, gh, gw, qw = self._rel_tables()
        _native.attention_dense(qkv, B, H, N, D, self.scale, _native.store_code(self._store_dtype()), out_f32=attn,
                                rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw, product=product, a_state=a_state, pv=pv)
        acc.matmul.count_product(B * H * N * N, D // H)
        if self.relative_position is not None:
            self.relative_position.count_fused(B, H)
        return True

    def _forward_attention(self, qkv, idx, count, cap, B, N):
        D, H = self.dim, self.heads
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        if self.ats_fraction is not None:
            product = self._scores(qkv, idx, count, cap, B, N)[0]
            return self._ats_attention(product, qkv, B, N, eventful=False) + (None,)
        if idx is None:
            attn = self._ws("attn_out", (B, N, D), torch.float32, qkv)
            if self._first_frame_fused(qkv, B, N, attn):
                self.matmul.count_product(B * H * N * (D // H), N)
                return attn, None, None
        product, kv, Nk, _, _, _ = self._scores(qkv, idx, count, cap, B, N)
        ry, rx, gh, gw, qw = self._rel_tables()
        a_s = self._ws("attn_probs", (B, H, N, Nk), sdt, qkv)
        v_s = self._ws("attn_values", (B, Nk, D), sdt, qkv)
        _native.softmax_gate(product, a_s, B, H, N, Nk, D, store, qkv=qkv, rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw)
        self._v_full(qkv, kv, B, N, Nk, v_s, store)
        attn = self._ws("attn_out", (B, N, D), torch.float32, qkv)
        _native.av(a_s, v_s, Nk, B, H, N, Nk, D, store, out_f32=attn)
        self.matmul.count_product(B * H * N * (D // H), Nk)
        return attn, None, None


class EventfulBlock(EventfulMatmul1Block):
    """EventfulMatmul1Block + gated attention-value product (blocks.py:543-575)."""

    def __init__(self, **super_kwargs):
        super().__init__(**super_kwargs)
        self.v_gate = TokenDeltaGate()
        self.matmul_gate = TokenDeltaGate(structure="col")
        self.matmul_accumulator_2 = MatmulDeltaAccumulator()

    def _forward_attention(self, qkv, idx, count, cap, B, N):
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        attn = self._ws("attn_out", (B, N, D), torch.float32, qkv)
        vg, ag, acc = self.v_gate, self.matmul_gate, self.matmul_accumulator_2
        if self.ats_fraction is not None:
            if count is not None:
                raise NotImplementedError("ats_fraction with a variable-count policy: the gated A.v path needs one index "
                                          "count for the whole batch (and ATS needs batch == heads > 1)")
            product = self._scores(qkv, idx, count, cap, B, N)[0]
            self._ats_idx_k = None if idx is None else idx.long()
            return self._ats_attention(product, qkv, B, N, eventful=True) + (None,)
        if self.pool_size is None or STREAM_POOLED:
            _, _, gh_, gw_, _ = self._rel_tables()
            if _native.attention_stream_fits(N, D, H, store, gh_, gw_):
                return self._attention_stream(qkv, idx, count, cap, B, N)
        if self.pool_size is None and self.relative_position is None and _native.attention_gated_fits(N, D, H, store) and \
                (ag._tiles is not None or (acc.first and idx is None and self.matmul_accumulator_1.first)):
            return self._attention_resident(qkv, idx, count, cap, B, N)
        if acc.first and idx is None and self.matmul_accumulator_1.first:
            a_state = torch.empty((B, H, N, N), dtype=sdt, device=qkv.device)
            pv = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
            if self._first_frame_fused(qkv, B, N, attn, a_state=a_state, pv=pv):
                vg.first = ag.first = acc.first = False
                ag.p, acc._state = a_state, pv
                vg._state = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
                vg.p = vg._state.view(B, N, H, dh).permute(0, 2, 1, 3)
                acc.product = acc._state.view(B, N, H, dh).permute(0, 2, 1, 3)
                self._v_full(qkv, None, B, N, N, vg._state, store)
                acc.matmul.count_product(B * H * N * dh, N)
                return attn, None, None
        acc1 = self.matmul_accumulator_1
        in_kernel_qk = (not acc.first and not acc1.first and self.pool_size is None and
                        _native.fused_qk_fits(N, N, D, H, cap))
        if in_kernel_qk:
            # Gated frame, <= 256 tokens, head dim 64: the fused kernel computes (q / scale) k^T itself from the updated
            # token buffer; K4 and the write + read of the (B,H,N,N) score state are skipped.  The state attribute is
            # refreshed lazily if anybody reads it (MatmulBuffer.defer).
            product, kv, Nk, idx_k, count_k, cap_k = None, None, N, idx, count, cap
            self._defer_scores(B, N)
            if acc1.matmul.count_mode:
                n_sel = self._n_rows(B, cap, count)
                acc1.matmul.count_product(2 * H * N * n_sel, dh)
            if self.relative_position is not None:
                self.relative_position.count_fused(B, H)
        else:
            product, kv, Nk, idx_k, count_k, cap_k = self._scores(qkv, idx, count, cap, B, N)
        ry, rx, gh, gw, qw = self._rel_tables()
        rel = dict(qkv=qkv, rel_y=ry, rel_x=rx, gh=gh, gw=gw, qw=qw)
        # value source for K6a: packed buffer, or the value half of the pooled buffer
        vsrc, vkw = (qkv, {}) if kv is None else (kv, dict(v_offset=D, v_rs=2 * D))
        if acc.first:
            vg.first = ag.first = acc.first = False
            ag.p = torch.empty((B, H, N, Nk), dtype=sdt, device=qkv.device)
            vg._state = torch.empty((B, Nk, D), dtype=sdt, device=qkv.device)
            acc._state = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
            # head-split views with the reference's logical shapes (B,H,tokens,dh)
            vg.p = vg._state.view(B, Nk, H, dh).permute(0, 2, 1, 3)
            acc.product = acc._state.view(B, N, H, dh).permute(0, 2, 1, 3)
            _native.softmax_gate(product, ag.p, B, H, N, Nk, D, store, **rel)
            self._v_full(qkv, kv, B, N, Nk, vg._state, store)
            _native.av(ag.p, vg._state, Nk, B, H, N, Nk, D, store, pv=acc._state, out_f32=attn)
            acc.matmul.count_product(B * H * N * dh, Nk)
            return attn, None, None
        fused = None
        if dh in (64, 128):
            # K6a with k-contiguous outputs + fused K5/K6: a~ / da~ stay in LDS
            v_delta = self._ws("v_delta_t", (B, D, cap_k), sdt, qkv)
            v_old = self._ws("v_old_t", (B, D, cap_k), sdt, qkv)
            _native.v_gate(vsrc, idx_k, count_k, B, Nk, D, cap_k, vg._state, v_delta, v_old, store, True,
                           transposed=True, **vkw)
            # The projection gate's delta norm ||attn - p||^2 comes out of the same epilogue, per head (the select kernel
            # adds the H partials): no separate pass over the attention output.
            pg = self.projection_gate
            fuse_norm = not pg.first and isinstance(pg.policy, _NormPolicy) and pg.policy.order == 2 and pg.p is not None
            nparts = self._ws("norm_parts", (B, N, H), torch.float32, qkv) if fuse_norm else None
            # bf16 `matmul_2_cast`: the attention output IS the A.v state (out == pv.float()).  When the projection then runs
            # on the persistent GEMM (which takes bf16 activations: half the bytes, no split, two MFMAs of three), it reads the
            # state directly and this launch does not write the fp32 copy at all.
            state_src = False
            if fuse_norm and PROJ_FROM_STATE and sdt == torch.bfloat16 and not isinstance(pg, SimpleSTGTGate) \
                    and self.projection.split_planes() is not None and pg.policy.fixed_count(N) is not None:
                cap_p = pg.policy.capacity(N)
                state_src = _native.gated_linear_big_tile(D, True, N, D, True, N, False, B, cap_p, D, D) != 0
            # rel-pos terms of all query tokens once per frame (evt_rel_terms): each 32-row workgroup of the fused kernel
            # would otherwise re-read 32 x (gh + gw) table rows (head dim 64, un-pooled key grid only)
            terms = None
            if ry is not None and dh == 64:
                terms = self._ws("rel_terms", (B, H, N, gh + gw), torch.float32, qkv)
                _native.rel_terms(qkv, ry, rx, B, H, N, D, gh, gw, qw, terms)
            _native.softmax_av_gated(product, ag.p, idx_k, count_k, cap_k, v_delta, v_old, acc._state,
                                     None if state_src else attn, B, H, N, D,
                                     store, Nk=Nk, scale=self.scale, norm_ref=pg.p if fuse_norm else None,
                                     norm_parts=nparts, rel_terms=terms, **rel)
            # handed to the projection group of this call (with `state_src` the fp32 `attn` scratch is NOT written)
            fused = dict(norm_parts=(nparts, H) if fuse_norm else None, state_src=acc._state if state_src else None)
        else:
            a_new = self._ws("a_new", (B, H, N, cap_k), sdt, qkv)
            a_delta = self._ws("a_delta", (B, H, N, cap_k), sdt, qkv)
            v_delta = self._ws("v_delta", (B, cap_k, D), sdt, qkv)
            v_old = self._ws("v_old", (B, cap_k, D), sdt, qkv)
            _native.softmax_gate(product, ag.p, B, H, N, Nk, D, store, a_new=a_new, a_delta=a_delta, idx=idx_k,
                                 count=count_k, kcap=cap_k, gated=True, **rel)
            _native.v_gate(vsrc, idx_k, count_k, B, Nk, D, cap_k, vg._state, v_delta, v_old, store, True, **vkw)
            _native.av(a_new, v_delta, cap_k, B, H, N, cap_k, D, store, pv=acc._state, out_f32=attn, a2=a_delta,
                       v2=v_old, count=count_k, gated=True)
        if self.count_mode or acc.count_mode or vg.count_mode:
            n = self._n_rows(B, cap_k, count_k)
            self._count_gate(vg, B * Nk * D)
            self._count_gate(ag, B * H * N * Nk)
            if acc.count_mode:
                acc.counts["accumulator_flops"] += n * D + 2 * B * N * D
            acc.matmul.count_product(2 * B * N * D, n // B if B else 0)
        return attn, None, fused

    def _attention_resident(self, qkv, idx, count, cap, B, N):
        """At most 256 tokens, head dim 64, a 16-bit `matmul_2_cast`, no relative position (every ViViT frame): ONE
        evt_attention_gated launch per frame -- value delta gate, scores from the token buffer, softmax, A delta gate and both
        accumulator products (blocks.py:558-575), one workgroup per (clip, head).  `matmul_gate.p` lives in the kernel's tiled layout
        (`_GateBase.use_tiles`: reads return the logical tensor), `matmul_accumulator_1.product` is refreshed lazily if read."""
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        vg, ag, acc, acc1 = self.v_gate, self.matmul_gate, self.matmul_accumulator_2, self.matmul_accumulator_1
        attn = self._ws("attn_out", (B, N, D), torch.float32, qkv)
        if acc.first:
            vg.first = ag.first = acc.first = acc1.first = False
            ag.use_tiles(_native.gated_tiles_empty(B, H, N, sdt, qkv.device), N)
            vg._state = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
            acc._state = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
            vg.p = vg._state.view(B, N, H, dh).permute(0, 2, 1, 3)
            acc.product = acc._state.view(B, N, H, dh).permute(0, 2, 1, 3)
            _native.attention_gated(qkv, ag._tiles, vg._state, acc._state, B, H, N, D, self.scale, store, True, out_f32=attn)
            self._defer_scores(B, N)
            acc1.matmul.count_product(B * H * N * N, dh)
            acc.matmul.count_product(B * H * N * dh, N)
            return attn, None, None
        pg = self.projection_gate
        fuse_norm = not pg.first and isinstance(pg.policy, _NormPolicy) and pg.policy.order == 2 and pg.p is not None
        nparts = self._ws("norm_parts", (B, N, H), torch.float32, qkv) if fuse_norm else None
        state_src = False   # bf16 cast: the projection reads the A.v state (see _forward_attention)
        if fuse_norm and PROJ_FROM_STATE and sdt == torch.bfloat16 and not isinstance(pg, SimpleSTGTGate) \
                and self.projection.split_planes() is not None and pg.policy.fixed_count(N) is not None:
            cap_p = pg.policy.capacity(N)
            state_src = _native.gated_linear_big_tile(D, True, N, D, True, N, False, B, cap_p, D, D) != 0
        _native.attention_gated(qkv, ag._tiles, vg._state, acc._state, B, H, N, D, self.scale, store, False, idx=idx, count=count,
                                kcap=cap, out_f32=None if state_src else attn, norm_ref=pg.p if fuse_norm else None, norm_parts=nparts)
        self._defer_scores(B, N)
        if self.count_mode or acc.count_mode or vg.count_mode or acc1.matmul.count_mode:
            n = self._n_rows(B, cap, count)
            if acc1.matmul.count_mode:   # the reference's delta update of the q.k^T state (modules.py:232-247)
                acc1.matmul.count_product(2 * H * N * n, dh)
            self._count_gate(vg, B * N * D)
            self._count_gate(ag, B * H * N * N)
            if acc.count_mode:
                acc.counts["accumulator_flops"] += n * D + 2 * B * N * D
            acc.matmul.count_product(2 * B * N * D, n // B if B else 0)
        fused = dict(norm_parts=(nparts, H) if fuse_norm else None, state_src=acc._state if state_src else None)
        return attn, None, fused

    def _attention_stream(self, qkv, idx, count, cap, B, N):
        """More than 256 tokens at head dim 64 (ViTDet global blocks): ONE evt_attention_stream launch per frame computes
        the scores in the kernel from the token buffer -- no q.k^T state, no K4 -- with the matmul_gate reference stored
        transposed (`matmul_gate.p` is a transposed view of it, same logical (B,H,N,Nk) tensor as in the reference).
        `matmul_accumulator_1.product` is refreshed lazily if anybody reads it (MatmulBuffer.defer).
        With `pool_size` (blocks.py:303-326, 509-511, 525-540) the keys and values are the pooled cells: Nk = N / (p0 p1) rows of
        the evt_pool_kv buffer, the key-side index list is the gate's list mapped to cells and de-duplicated (evt_pool_index)."""
        D, H = self.dim, self.heads
        dh = D // H
        sdt = self._store_dtype()
        store = _native.store_code(sdt)
        vg, ag, acc, acc1 = self.v_gate, self.matmul_gate, self.matmul_accumulator_2, self.matmul_accumulator_1
        attn = self._ws("attn_out", (B, N, D), torch.float32, qkv)
        ry, rx, gh, gw, qw = self._rel_tables()
        kv, Nk = self._pool_kv(qkv, B, N)
        pooled = dict(kv=kv, Nk=Nk) if kv is not None else {}
        idx_k, count_k, cap_k = idx, count, cap
        if kv is not None and idx is not None:
            cap_k = min(cap, Nk)
            idx_k = self._ws("idx_k", (B, cap_k), torch.int32, qkv)
            count_k = self._ws("cnt_k", (B,), torch.int32, qkv)
            p0, p1 = self.pool_size
            _native.pool_index(idx, count, B, cap, self.input_size[1], p0, p1, self.input_size[1] // p1, Nk, cap_k, idx_k, count_k)
        terms = None
        prep = ry is not None and _native.stream_prep_fits(D, H, cap_k, True)
        if ry is not None:   # decomposed rel-pos terms of every query token, once per frame (utils.py:159-168)
            terms = self._ws("rel_terms", (B, H, N, gh + gw), torch.float32, qkv)
            # (a gated frame computes them together with the key plane and the value gate: evt_stream_prep, below)
            if acc.first or not prep:
                _native.rel_terms(qkv, ry, rx, B, H, N, D, gh, gw, qw, terms)
            self.relative_position.count_fused(B, H)
        if acc.first:   # first frame of the clip: gate reference, value state, A.v state and the output in one launch
            vg.first = ag.first = acc.first = acc1.first = False
            ag._state_t = torch.empty((B, H, Nk, N), dtype=sdt, device=qkv.device)   # [b][h][key][row]
            ag.p = ag._state_t.transpose(-1, -2)
            vg._state = torch.empty((B, Nk, D), dtype=sdt, device=qkv.device)
            acc._state = torch.empty((B, N, D), dtype=sdt, device=qkv.device)
            vg.p = vg._state.view(B, Nk, H, dh).permute(0, 2, 1, 3)
            acc.product = acc._state.view(B, N, H, dh).permute(0, 2, 1, 3)
            self._v_full(qkv, kv, B, N, Nk, vg._state, store)
            _native.attention_stream(qkv, ag._state_t, acc._state, B, H, N, D, self.scale, store, True, rel_terms=terms,
                                     gh=gh, gw=gw, v_state=vg._state, out_f32=attn, **pooled)
            self._defer_scores(B, N)
            acc1.matmul.count_product(B * H * N * Nk, dh)
            acc.matmul.count_product(B * H * N * dh, Nk)
            return attn, None, None
        v_delta = self._ws("v_delta_t", (B, D, cap_k), sdt, qkv)
        v_old = self._ws("v_old_t", (B, D, cap_k), sdt, qkv)
        if prep:
            # rel-pos terms, key plane and value gate depend only on the updated token buffer (and the index list): one launch
            _native.stream_prep(qkv, ry, rx, terms, idx_k, count_k, cap_k, vg._state, v_delta, v_old, B, H, N, D, gh, gw, qw, store, **pooled)
        elif kv is None:
            _native.v_gate(qkv, idx, count, B, N, D, cap, vg._state, v_delta, v_old, store, True, transposed=True)
        else:
            _native.v_gate(kv, idx_k, count_k, B, Nk, D, cap_k, vg._state, v_delta, v_old, store, True, transposed=True,
                           v_offset=D, v_rs=2 * D)
        pg = self.projection_gate
        fuse_norm = not pg.first and isinstance(pg.policy, _NormPolicy) and pg.policy.order == 2 and pg.p is not None
        nparts = self._ws("norm_parts", (B, N, H), torch.float32, qkv) if fuse_norm else None
        state_src = False   # bf16 cast: the projection reads the A.v state (see _forward_attention)
        if fuse_norm and PROJ_FROM_STATE and sdt == torch.bfloat16 and not isinstance(pg, SimpleSTGTGate) \
                and self.projection.split_planes() is not None and pg.policy.fixed_count(N) is not None:
            cap_p = pg.policy.capacity(N)
            state_src = _native.gated_linear_big_tile(D, True, N, D, True, N, False, B, cap_p, D, D) != 0
        _native.attention_stream(qkv, ag._state_t, acc._state, B, H, N, D, self.scale, store, False, rel_terms=terms,
                                 gh=gh, gw=gw, idx=idx_k, count=count_k, kcap=cap_k, v_delta_t=v_delta, v_old_t=v_old,
                                 out_f32=None if state_src else attn, norm_ref=pg.p if fuse_norm else None,
                                 norm_parts=nparts, k_split_ready=prep, **pooled)
        self._defer_scores(B, N)
        if self.count_mode or acc.count_mode or vg.count_mode or acc1.matmul.count_mode:
            n = self._n_rows(B, cap, count)
            n_k = self._n_rows(B, cap_k, count_k)
            if acc1.matmul.count_mode:   # the reference's delta update of the q.k^T state (modules.py:232-247)
                acc1.matmul.count_product(H * Nk * n, dh)
                acc1.matmul.count_product(H * N * n_k, dh)
            self._count_gate(vg, B * Nk * D)
            self._count_gate(ag, B * H * N * Nk)
            if acc.count_mode:
                acc.counts["accumulator_flops"] += n_k * D + 2 * B * N * D
            acc.matmul.count_product(2 * B * N * D, n_k // B if B else 0)
        fused = dict(norm_parts=(nparts, H) if fuse_norm else None, state_src=acc._state if state_src else None)
        return attn, None, fused

    def reset_self(self):
        super().reset_self()
