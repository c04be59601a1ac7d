"""CPU oracle for the Eventful Transformer gated-token inference path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.  The product
package (`eventful-transformer_amd/eventful_transformer`) never imports anything from `oracle/`
and raises if the HIP library is missing.

What it is: a functional restatement, on torch-CPU fp32 ATen ops, of the algorithm in the
reference's `eventful_transformer/{modules,policies,blocks,utils,backbones}.py` and of the
per-step map of `models/vivit.py::ViViTSubModel`.  State lives in small explicit objects instead
of nn.Module attributes, blocks are plain functions over a parameter dict, and every function
cites the reference lines it restates.  The ATen op ORDER is kept identical to the reference so
that on one machine the oracle is bit-identical to the reference, free-running (same index sets).

Parity pinning: `oracle/gen_golden.py` imports the real reference (read-only mount, this container
only) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against
those vectors.  The reference itself has no tests or golden vectors.

Index convention: the reference's `topk(sorted=False)` returns an implementation-defined ORDER
(policies.py:63).  The oracle returns whatever ATen returns; comparisons are made on
ascending-sorted index sets (SURVEY.md §7 hard part 1).
"""
from math import prod, sqrt

import torch
import torch.nn.functional as F
from torch.linalg import vector_norm

LN_EPS = 1e-6  # blocks.py:23


# --------------------------------------------------------------------------------------------
# index broadcasting helpers (utils.py:198-211)
# --------------------------------------------------------------------------------------------
def rows_index(index, shape):
    """(…,k) -> gather/scatter index along dim -2 of a tensor of `shape` (utils.py:206-211)."""
    extra = len(shape) - index.ndim
    view = index.shape[:-1] + (1,) * (extra - 1) + (index.shape[-1], 1)
    return index.view(view).expand(tuple(shape[:-2]) + (-1, shape[-1]))


def cols_index(index, shape):
    """(…,k) -> gather/scatter index along dim -1 of a tensor of `shape` (utils.py:198-203)."""
    extra = len(shape) - index.ndim
    view = index.shape[:-1] + (1,) * extra + index.shape[-1:]
    return index.view(view).expand(tuple(shape[:-1]) + (-1,))


# --------------------------------------------------------------------------------------------
# policies (policies.py)
# --------------------------------------------------------------------------------------------
class TopK:
    """policies.py:39-68."""

    def __init__(self, k, order=2):
        self.k = k
        self.order = order
        self.last_input = None
        self.last_output = None

    def __call__(self, e, dim=-1):
        out = vector_norm(e, ord=self.order, dim=dim).topk(self.k, sorted=False)[1]
        self.last_input, self.last_output = e, out
        return out


class TopFraction:
    """policies.py:71-95."""

    def __init__(self, fraction, order=2):
        self.fraction = fraction
        self.order = order

    def __call__(self, e, dim=-1):
        n = vector_norm(e, ord=self.order, dim=dim)
        return n.topk(int(self.fraction * n.shape[-1]), sorted=False)[1]


class Threshold:
    """policies.py:6-32 (batch 1 only, ascending indices, data-dependent count)."""

    def __init__(self, threshold, save_status=False, order=2):
        self.threshold = threshold
        self.order = order
        self.save_status = save_status
        # save_status: how close the call was -- min over tokens of | ||e|| - threshold | / threshold, and the tokens within
        # NEAR of the threshold (indices, relative distances): the only ones another summation order may decide differently
        self.last_margin = None
        self.last_near = None

    NEAR = 1e-3

    def __call__(self, e, dim=-1):
        assert all(s == 1 for s in e.shape[:-2])
        norms = vector_norm(e, ord=self.order, dim=dim)
        if self.save_status:
            rel = ((norms.double() - self.threshold).abs() / self.threshold).reshape(-1)
            self.last_margin = float(rel.min())
            near = (rel < self.NEAR).nonzero().reshape(-1)
            self.last_near = (near, rel[near])
        hit = norms.gt(self.threshold).nonzero()
        return hit[..., -1].view((1,) * (e.ndim - 2) + (-1,))


# --------------------------------------------------------------------------------------------
# gates / buffers (modules.py)
# --------------------------------------------------------------------------------------------
class Slot:
    """One piece of per-clip state (`p`, `b` or `product` in the reference); None == first frame."""

    __slots__ = ("t",)

    def __init__(self):
        self.t = None


def _gate_index(e, policy, forced, structure):
    # modules.py:154-164
    if forced is None:
        index = policy(e, dim=(-1 if structure == "row" else -2))
    else:
        index = forced
    if structure == "row":
        return -2, rows_index(index, e.shape), index
    return -1, cols_index(index, e.shape), index


def token_gate(slot, c, policy, forced=None, structure="row"):
    """TokenGate.forward (modules.py:123-152).  First frame keeps a REFERENCE to c (modules.py:140)."""
    if slot.t is None:
        slot.t = c
        return c, None
    dim, wide, index = _gate_index(c - slot.t, policy, forced, structure)
    picked = c.gather(dim=dim, index=wide)
    slot.t.scatter_(dim=dim, index=wide, src=picked)
    return picked, index


def token_delta_gate(slot, c, policy, forced=None, structure="row"):
    """TokenDeltaGate.forward (modules.py:183-201): also returns the gathered delta."""
    if slot.t is None:
        slot.t = c
        return c, None, None
    e = c - slot.t
    dim, wide, index = _gate_index(e, policy, forced, structure)
    picked = c.gather(dim=dim, index=wide)
    picked_e = e.gather(dim=dim, index=wide)
    slot.t.scatter_(dim=dim, index=wide, src=picked)
    return picked, picked_e, index


def stgt_gate(slot, c, policy):
    """SimpleSTGTGate (modules.py:6-49): reference = previous INPUT, replaced every frame."""
    if slot.t is None:
        slot.t = c
        return c, None
    index = policy(c - slot.t, dim=-1)
    picked = c.gather(dim=-2, index=rows_index(index, c.shape))
    slot.t = c
    return picked, index


def token_buffer(slot, x, index, structure="row"):
    """TokenBuffer.forward (modules.py:68-97).  Returns a reference to the state."""
    if slot.t is None:
        slot.t = x.clone()
        return slot.t
    if structure == "row":
        slot.t.scatter_(dim=-2, index=rows_index(index, slot.t.shape), src=x)
    else:
        slot.t.scatter_(dim=-1, index=cols_index(index, slot.t.shape), src=x)
    return slot.t


def qk_buffer(slot, q, kt, index_q, index_k):
    """MatmulBuffer.forward (modules.py:214-248): rows index_q then columns index_k, both exact."""
    if slot.t is None:
        slot.t = q @ kt
        return slot.t
    q_rows = q.gather(dim=-2, index=rows_index(index_q, q.shape))
    k_cols = kt.gather(dim=-1, index=cols_index(index_k, kt.shape))
    slot.t.scatter_(dim=-2, index=rows_index(index_q, slot.t.shape), src=q_rows @ kt)
    slot.t.scatter_(dim=-1, index=cols_index(index_k, slot.t.shape), src=q @ k_cols)
    return slot.t


def av_accumulator(slot, a_new, v_new, a_delta, v_delta):
    """MatmulDeltaAccumulator.forward (modules.py:265-295) -- the approximate A.v update (I4)."""
    if slot.t is None:
        slot.t = a_new @ v_new
        return slot.t
    slot.t += a_new @ v_delta
    slot.t += a_delta @ (v_new - v_delta)
    return slot.t


# --------------------------------------------------------------------------------------------
# relative position embedding (utils.py:108-195)
# --------------------------------------------------------------------------------------------
def relative_table(embedding, emb_size, attn_size, axis, pool=None):
    """RelativePositionEmbedding._get_relative (utils.py:175-189); `pool` = pool_size or None."""
    n = emb_size[axis]
    offs = torch.arange(n).unsqueeze(1) - torch.arange(n).unsqueeze(0) + n - 1
    rel = embedding[offs]
    if tuple(emb_size) != tuple(attn_size):
        rel = rel.transpose(0, 2).unsqueeze(0)
        rel = F.interpolate(rel, tuple(attn_size), mode="bicubic", align_corners=False)
        rel = rel.squeeze(0).transpose(0, 2)
    if pool is not None:  # utils.py:185-188: average the KEY axis
        rel = F.avg_pool1d(rel.transpose(1, 2), pool[axis]).transpose(1, 2)
    return rel


def add_relative(x, q, rel_y, rel_x, attn_size, inplace, pool=None):
    """RelativePositionEmbedding.forward (utils.py:139-173).  q is UNSCALED; keys live on the pooled grid."""
    a = tuple(attn_size)
    kgrid = a if pool is None else (a[0] // pool[0], a[1] // pool[1])
    x = x.view(x.shape[:2] + a + kgrid)
    q = q.view(q.shape[:2] + a + q.shape[-1:])
    ty = torch.einsum("abhwc,hkc->abhwk", q, rel_y).unsqueeze(-1)
    tx = torch.einsum("abhwc,wkc->abhwk", q, rel_x).unsqueeze(-2)
    if inplace:
        x += ty
    else:
        x = x + ty
    x += tx
    return x.view(x.shape[:2] + (prod(a), prod(kgrid)))


# --------------------------------------------------------------------------------------------
# blocks (blocks.py)
# --------------------------------------------------------------------------------------------
PARAM_KEYS = (
    "input_layer_norm.weight", "input_layer_norm.bias", "qkv.weight", "qkv.bias",
    "projection.weight", "projection.bias", "mlp_layer_norm.weight", "mlp_layer_norm.bias",
    "mlp_1.weight", "mlp_1.bias", "mlp_2.weight", "mlp_2.bias",
)


class BlockOracle:
    """One transformer block of any of the four reference classes, as state + pure functions.

    kind: "Block" | "EventfulTokenwiseBlock" | "EventfulMatmul1Block" | "EventfulBlock"
    params: dict with the reference's state_dict key names (blocks.py:94-116).
    `ats_fraction` restates the reference's adaptive token sampling as written (blocks.py:150-181,378-391): its
    scores are summed over the BATCH axis (blocks.py:163), so it only runs when batch == heads.
    """

    def __init__(self, kind, params, dim, heads, input_size, window_size=None,
                 relative_embedding_size=None, matmul_2_cast=None, gate_before_ln=False, stgt=False, pool_size=None,
                 ats_fraction=None):
        self.kind = kind
        self.p = params
        self.dim, self.heads = dim, heads
        self.input_size = tuple(input_size)
        self.window_size = None if window_size is None else tuple(window_size)
        self.cast = None if matmul_2_cast is None else getattr(torch, matmul_2_cast)
        self.gate_before_ln = gate_before_ln
        self.stgt = stgt
        self.pool = None if pool_size is None else ((pool_size,) * 2 if isinstance(pool_size, int) else tuple(pool_size))
        self.scale = sqrt(dim // heads)  # blocks.py:92
        if self.window_size is not None:
            attn = self.window_size
            if relative_embedding_size is not None:
                relative_embedding_size = self.window_size  # blocks.py:90-91
        else:
            attn = self.input_size
        self.attn_size = attn
        self.rel_size = None if relative_embedding_size is None else tuple(relative_embedding_size)
        self.rel_y = self.rel_x = None
        if kind in ("EventfulMatmul1Block", "EventfulBlock"):
            assert self.window_size is None  # blocks.py:485
        self.ats_fraction = ats_fraction
        self.last_ats = None
        self.policy = {}  # gate name -> policy callable (utils/misc.py:140-143: one per gate)
        self.reset()

    # -- harness -----------------------------------------------------------------------------
    GATES = ("qkv_gate", "projection_gate", "mlp_gate", "v_gate", "matmul_gate")

    def set_policy(self, factory):
        self.policy = {g: factory() for g in self.GATES}

    def reset(self):
        names = ("qkv_gate", "qkv_accumulator", "projection_gate", "projection_accumulator",
                 "mlp_gate", "mlp_accumulator", "matmul_accumulator_1", "v_gate", "matmul_gate",
                 "matmul_accumulator_2")
        self.s = {n: Slot() for n in names}
        self.rel_y = self.rel_x = None  # utils.py:191-195
        self.last_ats = None            # blocks.py:139-140
        self.trace = {}

    # -- pieces ------------------------------------------------------------------------------
    def _ln(self, x, which):
        return F.layer_norm(x, (self.dim,), self.p[which + ".weight"], self.p[which + ".bias"], LN_EPS)

    def _lin(self, x, which):
        return F.linear(x, self.p[which + ".weight"], self.p[which + ".bias"])

    def _mlp(self, x):
        return self._lin(F.gelu(self._lin(x, "mlp_1")), "mlp_2")  # blocks.py:242-246, exact erf

    def _gate(self, name, c):
        if self.stgt:
            return stgt_gate(self.s[name], c, self.policy.get(name))
        return token_gate(self.s[name], c, self.policy.get(name))

    def _heads(self, x):
        # blocks.py:248-255: (B,N,3D) -> three non-contiguous (B,H,N,dh) views
        x = x.view(x.shape[:-1] + (3, self.heads, x.shape[-1] // (3 * self.heads)))
        return x.permute(2, 0, 3, 1, 4)

    @staticmethod
    def _merge(x):
        # blocks.py:328-344 -- incl. its assertion that the reshape COPIED (with one head, or one token, it is a view: a gate would be handed a raw
        # reference to an accumulator state and see a delta of zero for ever; the reference refuses, blocks.py:341)
        x = x.permute(0, 2, 1, 3)
        y = x.reshape(x.shape[:-2] + (-1,))
        assert x.data_ptr() != y.data_ptr(), "blocks.py:341: recombining the heads must copy (one head, or one token, is not supported by the reference)"
        return y

    def _window_pad(self):
        return (-self.input_size[0] % self.window_size[0], -self.input_size[1] % self.window_size[1])

    def _to_windows(self, x):
        # blocks.py:257-301 (input already in the QKV domain: pad value = qkv bias, :280-281)
        if self.window_size is None:
            return x
        ph, pw = self._window_pad()
        d = self.window_size
        x = x.view(x.shape[:1] + self.input_size + x.shape[2:])
        if ph or pw:
            fill = (torch.zeros((1,) * (x.ndim - 1) + x.shape[-1:], dtype=x.dtype) + self.p["qkv.bias"])
            # utils/image.py:31-49: pad last dim first (no-op), then width, then height
            if pw:
                shape = list(x.shape)
                shape[-2] = pw
                x = torch.concat([x, fill.expand(shape)], -2)
            if ph:
                shape = list(x.shape)
                shape[-3] = ph
                x = torch.concat([x, fill.expand(shape)], -3)
        s = x.shape
        x = x.view(-1, s[-3] // d[0], d[0], s[-2] // d[1], d[1], s[-1]).transpose(-3, -4)
        return x.reshape(-1, prod(d), s[-1])

    def _from_windows(self, x):
        # blocks.py:346-376
        if self.window_size is None:
            return x
        ph, pw = self._window_pad()
        d, s = self.window_size, self.input_size
        th, tw = s[0] + ph, s[1] + pw
        x = x.view(-1, th // d[0], tw // d[1], d[0], d[1], x.shape[-1]).transpose(-3, -4)
        x = x.reshape(-1, th, tw, x.shape[-1])
        if ph or pw:
            x = x[:, : s[0], : s[1]]
        return x.flatten(start_dim=1, end_dim=2)

    def _rel(self, x, q, inplace):
        if self.rel_size is None:
            return x
        if self.rel_y is None:
            self.rel_y = relative_table(self.p["relative_position.y_embedding"], self.rel_size, self.attn_size, 0, self.pool)
            self.rel_x = relative_table(self.p["relative_position.x_embedding"], self.rel_size, self.attn_size, 1, self.pool)
        return add_relative(x, q, self.rel_y, self.rel_x, self.attn_size, inplace, self.pool)

    def _pool_tokens(self, x):
        # blocks.py:303-326: average pooling of keys / values over the token grid
        if self.pool is None:
            return x
        w = self.input_size if self.window_size is None else self.window_size
        s = x.shape
        x = x.reshape((-1,) + w + x.shape[-1:]).permute(0, 3, 1, 2)
        x = F.avg_pool2d(x, self.pool).permute(0, 2, 3, 1)
        return x.view(s[:-2] + (-1,) + s[-1:])

    def _pool_index(self, index):
        # blocks.py:525-540
        if self.pool is None or index is None:
            return index
        width = self.input_size[1]
        iy = index.div(width, rounding_mode="floor").div(self.pool[0], rounding_mode="floor")
        ix = index.remainder(width).div(self.pool[1], rounding_mode="floor")
        return (iy * (width // self.pool[1]) + ix).unique(dim=-1)

    def _cast2(self, a, v):
        # blocks.py:183-189
        if self.cast is None:
            return a, v, a.dtype
        return a.to(self.cast), v.to(self.cast), a.dtype

    # -- adaptive token sampling (blocks.py:150-181, 378-391, 196-203) ---------------------------
    def _ats(self, a, v):
        """a (B,H,N,N) probabilities, v (B,H,N,dh) -> (a rows gathered (B,H,n,N), indices (H,n)) or (a, None)."""
        if self.ats_fraction is None:
            return a, None
        class_scores = a[..., 0]
        raw = class_scores * vector_norm(v[...], dim=-1)
        scores = raw / raw[..., 1:].sum(dim=-1, keepdim=True)
        scores[..., 0] = float("inf")           # the class token always stays
        scores = scores.sum(dim=-3)             # blocks.py:163 ("sum scores over heads": reduces the BATCH axis)
        n_select = int(self.ats_fraction * (scores.shape[-1] - 1)) + 1
        self.trace["ats_scores"] = scores
        index = scores.topk(n_select, sorted=False)[1]
        index = self._stabilize_ats(index)
        self.last_ats = index
        self.trace["ats_index"] = index
        return a.gather(dim=-2, index=rows_index(index, a.shape)), index

    def _stabilize_ats(self, index):
        # blocks.py:378-391: ascending, then keep every token that survives at the position it had last frame
        index = index.sort(dim=-1)[0]
        if self.last_ats is None:
            return index
        new, old = index.flatten(end_dim=-2), self.last_ats.flatten(end_dim=-2)
        out = old.clone()
        for i in range(new.shape[0]):
            gone = torch.isin(old[i], new[i], invert=True)
            fresh = torch.isin(new[i], old[i], invert=True)
            out[i, gone] = new[i, fresh]
        return out.view(index.shape)

    @staticmethod
    def _ats_skip(skip, index):
        # blocks.py:196-203
        return skip if index is None else skip.gather(dim=-2, index=rows_index(index, skip.shape))

    # -- attention variants --------------------------------------------------------------------
    def _attention_dense(self, qkv):
        # Block._forward_attention (blocks.py:205-240)
        q, k, v = self._heads(self._to_windows(qkv))
        k, v = self._pool_tokens(k), self._pool_tokens(v)
        x = (q / self.scale) @ k.transpose(-2, -1)
        x = self._rel(x, q, inplace=True)
        x = x.softmax(dim=-1)
        x, self._ats_index = self._ats(x, v)
        x, v, old = self._cast2(x, v)
        x = self._from_windows(self._merge(x @ v))
        return x.to(old) if self.cast is not None else x

    def _scores_gated(self, qkv, index):
        # EventfulMatmul1Block._forward_matmul_1 (blocks.py:506-523)
        q, k, v = self._heads(qkv)
        k, v = self._pool_tokens(k), self._pool_tokens(v)
        index_k = self._pool_index(index)
        self.trace["index_k"] = index_k
        x = qk_buffer(self.s["matmul_accumulator_1"], q / self.scale, k.transpose(-2, -1), index, index_k)
        x = self._rel(x, q, inplace=False)
        return x.softmax(dim=-1), v, index_k

    def _attention_matmul1(self, qkv, index):
        # EventfulMatmul1Block._forward_attention (blocks.py:497-504)
        a, v, _ = self._scores_gated(qkv, index)
        a, self._ats_index = self._ats(a, v)
        a, v, old = self._cast2(a, v)
        x = self._merge(a @ v)
        return x.to(old) if self.cast is not None else x

    def _attention_eventful(self, qkv, index):
        # EventfulBlock._forward_attention (blocks.py:558-575)
        a, v, index_k = self._scores_gated(qkv, index)
        a, v, old = self._cast2(a, v)
        a, self._ats_index = self._ats(a, v)      # after the cast here (blocks.py:561-562)
        if self.cast is None:
            v = v.clone()
        v_new, v_delta, index_v = token_delta_gate(self.s["v_gate"], v, None, forced=index_k)
        a_new, a_delta, _ = token_delta_gate(self.s["matmul_gate"], a, None, forced=index_v, structure="col")
        x = av_accumulator(self.s["matmul_accumulator_2"], a_new, v_new, a_delta, v_delta)
        self.trace["attn_state"] = x
        x = self._merge(x)
        return x.to(old) if self.cast is not None else x

    # -- forward -------------------------------------------------------------------------------
    def forward(self, x):
        tr = self.trace = {}
        if self.kind == "Block":
            # Block.forward (blocks.py:117-137)
            skip = x
            x = self._lin(self._ln(x, "input_layer_norm"), "qkv")
            x = self._attention_dense(x)
            x = self._lin(x, "projection") + self._ats_skip(skip, self._ats_index)
            return self._mlp(self._ln(x, "mlp_layer_norm")) + x

        # EventfulTokenwiseBlock._forward_pre_attention (blocks.py:452-463)
        skip = x
        if self.gate_before_ln:
            x, index = self._gate("qkv_gate", x)
            x = self._ln(x, "input_layer_norm")
        else:
            x = self._ln(x, "input_layer_norm")
            tr["qkv_gate_in"] = x
            x, index = self._gate("qkv_gate", x)
        tr["qkv_index"] = index
        x = self._lin(x, "qkv")
        x = token_buffer(self.s["qkv_accumulator"], x, index)
        tr["qkv_buffer"] = x
        if self.kind == "EventfulTokenwiseBlock":
            x = self._attention_dense(x)
        elif self.kind == "EventfulMatmul1Block":
            x = self._attention_matmul1(x, index)
        else:
            x = self._attention_eventful(x, index)
        tr["attn_out"] = x
        skip = self._ats_skip(skip, self._ats_index)   # blocks.py:426,493

        # EventfulTokenwiseBlock._forward_post_attention (blocks.py:430-450)
        x, index = self._gate("projection_gate", x)
        tr["projection_index"] = index
        x = self._lin(x, "projection")
        x = token_buffer(self.s["projection_accumulator"], x, index)
        x = x + skip
        skip = x
        tr["mid"] = x
        if self.gate_before_ln:
            x, index = self._gate("mlp_gate", x)
            x = self._ln(x, "mlp_layer_norm")
        else:
            x = self._ln(x, "mlp_layer_norm")
            x, index = self._gate("mlp_gate", x)
        tr["mlp_index"] = index
        x = self._mlp(x)
        x = token_buffer(self.s["mlp_accumulator"], x, index)
        return x + skip


# --------------------------------------------------------------------------------------------
# backbone + ViViT spatial sub-model step (backbones.py:61-64, vivit.py:293-303)
# --------------------------------------------------------------------------------------------
def sized_position_encoding(encoding, enc_size, input_size, has_class_token):
    """PositionEncoding._compute_sized_encoding (utils.py:69-100)."""
    enc_size, input_size = tuple(enc_size), tuple(input_size)
    if enc_size == input_size:
        return encoding
    cls = None
    if has_class_token:
        cls, encoding = encoding[:, :1], encoding[:, 1:]
    e = encoding.transpose(1, 2)
    e = e.view(e.shape[:-1] + enc_size)
    e = F.interpolate(e, input_size, mode="bicubic", align_corners=False)
    e = e.flatten(start_dim=2).transpose(1, 2)
    if has_class_token:
        e = torch.concat([cls, e], dim=1)
    return e


class BackboneOracle:
    """ViTBackbone (backbones.py:8-64): position encoding add, then the blocks in order."""

    def __init__(self, blocks, position_encoding, enc_size, input_size, has_class_token):
        self.blocks = blocks
        self.encoding = sized_position_encoding(position_encoding, enc_size, input_size, has_class_token)

    def set_policy(self, factory):
        for b in self.blocks:
            b.set_policy(factory)

    def reset(self):
        for b in self.blocks:
            b.reset()

    def forward(self, x):
        x = x + self.encoding
        for b in self.blocks:
            x = b.forward(x)
        return x


class ViViTSpatialOracle:
    """ViViTSubModel.forward (vivit.py:293-303): class token, backbone, final LN, token 0."""

    def __init__(self, backbone, class_token, ln_weight, ln_bias):
        self.backbone, self.class_token = backbone, class_token
        self.ln_weight, self.ln_bias = ln_weight, ln_bias

    def reset(self):
        self.backbone.reset()

    def forward(self, x):
        cls = self.class_token.expand((x.shape[0],) + self.class_token.shape[1:])
        x = self.backbone.forward(torch.concat([cls, x], dim=1))
        x = F.layer_norm(x, x.shape[-1:], self.ln_weight, self.ln_bias, LN_EPS)
        return x[:, 0]


# --------------------------------------------------------------------------------------------
# deterministic synthetic parameters / inputs shared by the oracle, the tests and bench.py
# --------------------------------------------------------------------------------------------
def make_block_params(dim, mlp_ratio, seed, std=0.02, rel_sizes=None, head_dim=None):
    """Seeded block parameters under the reference's state_dict names.

    The reference initialises every CountedLinear / embedding to zeros (counting.py:143-144,
    utils.py:125-130), so "random init" is defined here: numpy RandomState (version-stable),
    normal(0, std), LN weight 1 + small noise / bias small noise so LN affine is exercised.
    """
    import numpy as np

    rs = np.random.RandomState(seed)

    def n(*shape, s=std):
        return torch.from_numpy((rs.standard_normal(shape) * s).astype(np.float32))

    p = {
        "input_layer_norm.weight": 1.0 + n(dim, s=0.05), "input_layer_norm.bias": n(dim, s=0.05),
        "qkv.weight": n(3 * dim, dim), "qkv.bias": n(3 * dim),
        "projection.weight": n(dim, dim), "projection.bias": n(dim),
        "mlp_layer_norm.weight": 1.0 + n(dim, s=0.05), "mlp_layer_norm.bias": n(dim, s=0.05),
        "mlp_1.weight": n(dim * mlp_ratio, dim), "mlp_1.bias": n(dim * mlp_ratio),
        "mlp_2.weight": n(dim, dim * mlp_ratio), "mlp_2.bias": n(dim),
    }
    if rel_sizes is not None:
        p["relative_position.y_embedding"] = n(2 * rel_sizes[0] - 1, head_dim)
        p["relative_position.x_embedding"] = n(2 * rel_sizes[1] - 1, head_dim)
    return p


def sharpen_qk(params, dim, qk_std, std=0.02):
    """Rescales the query and key thirds of a block's `qkv.weight` / `qkv.bias` (rows [0, 2*dim)) from std to qk_std, in place.

    With std-0.02 random weights the attention logits have a standard deviation of ~0.3: near-uniform attention, whose output
    barely depends on which tokens moved -- the projection gate's delta norms are then near-tied (median top-k margin 1.7e-4,
    SURVEY.md section 7-1).  qk_std = 0.06 gives logits of std ~2.8 and projection-gate margins with a median >= 1e-3."""
    for key in ("qkv.weight", "qkv.bias"):
        w = params[key].clone()
        w[: 2 * dim] *= qk_std / std
        params[key] = w
    return params


def make_token_stream(batch, tokens, dim, steps, n_changed, seed, scale=1.0, small=0.0):
    """Synthetic (steps, batch, tokens, dim) token stream (SURVEY.md §8d "Synthetic inputs").

    Step 0 is N(0, scale^2); every later step re-randomises exactly `n_changed` tokens per clip
    (so top-k with k == n_changed has wide margins at the qkv gate) and adds N(0, small^2) noise
    to all the others (small > 0 keeps norms distinct, no zero-norm ties).
    """
    import numpy as np

    rs = np.random.RandomState(seed)
    out = np.empty((steps, batch, tokens, dim), dtype=np.float32)
    cur = (rs.standard_normal((batch, tokens, dim)) * scale).astype(np.float32)
    out[0] = cur
    for t in range(1, steps):
        cur = cur.copy()
        if small > 0:
            cur += (rs.standard_normal(cur.shape) * small).astype(np.float32)
        for b in range(batch):
            pick = rs.permutation(tokens)[:n_changed]
            cur[b, pick] = (rs.standard_normal((n_changed, dim)) * scale).astype(np.float32)
        out[t] = cur
    return torch.from_numpy(out)


def make_gate_case(seed, batch, tokens, dim):
    """Seeded (c, p) pair for gate-level known-answer tests: p ~ N(0,1), c = p + e with per-token
    log-normal delta magnitudes so the delta norms are spread out (wide top-k margins)."""
    import numpy as np

    rs = np.random.RandomState(seed)
    p = rs.standard_normal((batch, tokens, dim)).astype(np.float32)
    mag = (0.3 * np.exp(rs.standard_normal((batch, tokens, 1)))).astype(np.float32)
    e = rs.standard_normal((batch, tokens, dim)).astype(np.float32) * mag
    return torch.from_numpy(p + e), torch.from_numpy(p)


def make_threshold_case(seed, tokens, dim, count, threshold=1.0):
    """Seeded batch-1 (c, p, threshold) where exactly `count` tokens have ||c-p|| > threshold
    (norms in [1.5,3] x threshold for the selected tokens, [0.1,0.6] x threshold for the others)."""
    import numpy as np

    rs = np.random.RandomState(seed)
    p = rs.standard_normal((1, tokens, dim)).astype(np.float32)
    u = rs.standard_normal((1, tokens, dim))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    target = rs.uniform(0.1, 0.6, size=(1, tokens, 1)) * threshold
    pick = rs.permutation(tokens)[:count]
    target[0, pick, 0] = rs.uniform(1.5, 3.0, size=count) * threshold
    e = (u * target).astype(np.float32)
    return torch.from_numpy(p + e), torch.from_numpy(p), float(threshold)


def make_threshold_stream(tokens, dim, steps, seed, frac=0.1, big=0.5, small=1e-3):
    """Batch-1 token stream for threshold-policy runs: each step ~`frac` of the
    tokens move by N(0, big^2) and the rest by N(0, small^2)."""
    import numpy as np

    rs = np.random.RandomState(seed)
    out = np.empty((steps, 1, tokens, dim), dtype=np.float32)
    cur = rs.standard_normal((1, tokens, dim)).astype(np.float32)
    out[0] = cur
    n_big = int(frac * tokens)
    for t in range(1, steps):
        cur = cur + (rs.standard_normal(cur.shape) * small).astype(np.float32)
        pick = rs.permutation(tokens)[:n_big]
        cur[0, pick] += (rs.standard_normal((n_big, dim)) * big).astype(np.float32)
        out[t] = cur
    return torch.from_numpy(out)


def make_varied_threshold_stream(tokens, dim, steps, seed, lo=1e-4, hi=1.0, frac=(0.02, 0.15)):
    """Batch-1 token stream with CONTINUOUS perturbation magnitudes for threshold-policy runs (BASELINE config 5, "variable r
    per frame"): each step a random fraction f ~ U(frac) of the tokens moves by N(0, s^2) with a per-token log-uniform
    s in [lo, hi]; the others jitter by N(0, lo^2).  The selected-token count of every gate then depends on the data, on the
    frame and on the threshold (make_threshold_stream is bimodal: the same ~10 % of tokens at every threshold)."""
    import numpy as np

    rs = np.random.RandomState(seed)
    out = np.empty((steps, 1, tokens, dim), dtype=np.float32)
    cur = rs.standard_normal((1, tokens, dim)).astype(np.float32)
    out[0] = cur
    for t in range(1, steps):
        f = rs.uniform(*frac)
        moving = rs.uniform(size=tokens) < f
        scale = np.exp(rs.uniform(np.log(lo), np.log(hi), size=tokens))
        scale[~moving] = lo
        cur = cur + (rs.standard_normal(cur.shape) * scale[None, :, None]).astype(np.float32)
        out[t] = cur
    return torch.from_numpy(out)
