"""ctypes binding of libevt_hip.so (C ABI in include/evt_abi.h) + the scratch-tensor pool.

PyTorch-ROCm is plumbing here: it owns device memory and the HIP stream; every kernel on the
gated-token path is a hand-written gfx950 kernel reached through this module.  There is no CPU
or ATen fallback for that path: if the library cannot be loaded, or a tensor is not on a HIP
device, the call raises.
"""
import ctypes
import os
import threading
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVT_LIB", os.path.join(_HERE, "libevt_hip.so"))   # EVT_LIB: A/B another build of the same ABI

EVT_F32, EVT_BF16, EVT_F16 = 0, 1, 2
ACT_NONE, ACT_GELU = 0, 1

_STORE_OF = {torch.float32: EVT_F32, torch.bfloat16: EVT_BF16, torch.float16: EVT_F16}

# Every symbol include/evt_abi.h declares (tests check that the .so exports all of them).
ABI_VERSION = 9   # include/evt_abi.h EVT_ABI_VERSION
ABI_SYMBOLS = (
    "evt_version", "evt_last_error_string", "evt_target_arch", "evt_row_pass", "evt_row_pass_ord", "evt_select_topk",
    "evt_select_threshold", "evt_select_topk_sq", "evt_select_threshold_sq", "evt_gate_gather_update", "evt_scatter_rows", "evt_gated_linear",
    "evt_gated_linear_workspace_bytes", "evt_gated_linear_big_tile", "evt_gated_mlp", "evt_split_weights", "evt_split_weights_bytes", "evt_qk", "evt_softmax_gate", "evt_v_gate", "evt_av", "evt_softmax_av_gated", "evt_rel_terms", "evt_pool_kv", "evt_pool_index",
    "evt_attention_dense", "evt_attention_stream", "evt_attention_stream_lds_bytes", "evt_attention_stream_key_blocks", "evt_stream_prep", "evt_prefetch", "evt_select_prefetch_next", "evt_attention_dense_resident",
    "evt_attention_gated", "evt_attention_gated_fits", "evt_attention_gated_tile_bytes",
    "evt_gate_cols", "evt_scatter_cols", "evt_gate_rows_any", "evt_move_rows_any", "evt_gather_rows_map", "evt_scatter_rows_map",
    "evt_ats_scores", "evt_ats_stabilize", "evt_set_cu_budget",
)


class LinearDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("a_idx", c_void_p), ("a_rows", c_int32),
        ("W", c_void_p), ("bias", c_void_p), ("out", c_void_p), ("ldo", c_int64),
        ("o_idx", c_void_p), ("o_rows", c_int32), ("count", c_void_p), ("p_upd", c_void_p),
        ("B", c_int32), ("kcap", c_int32), ("K", c_int32), ("Nout", c_int32), ("act", c_int32),
        ("W_split", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64), ("a_bf16", c_int32),
    ]


class MlpDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("idx", c_void_p), ("rows", c_int32),
        ("W1", c_void_p), ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p),
        ("hidden", c_void_p), ("out", c_void_p), ("ldo", c_int64), ("count", c_void_p),
        ("p_upd", c_void_p), ("B", c_int32), ("kcap", c_int32), ("D", c_int32), ("Dh", c_int32),
        ("W1_split", c_void_p), ("W2_split", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class QkDesc(Structure):
    _fields_ = [
        ("q", c_void_p), ("q_bs", c_int64), ("q_hs", c_int64), ("q_rs", c_int64),
        ("k", c_void_p), ("k_bs", c_int64), ("k_hs", c_int64), ("k_rs", c_int64),
        ("product", c_void_p),
        ("idx_q", c_void_p), ("count_q", c_void_p), ("kcap_q", c_int32), ("idx_q_rest", c_void_p),
        ("idx_k", c_void_p), ("count_k", c_void_p), ("kcap_k", c_int32),
        ("tok_map", c_void_p), ("groups_per_clip", c_int32), ("pad_q", c_void_p), ("pad_k", c_void_p),
        ("G", c_int32), ("H", c_int32), ("Nq", c_int32), ("Nk", c_int32), ("dh", c_int32),
        ("scale", c_float), ("delta", c_int32), ("split", c_int32),
    ]


class SoftmaxDesc(Structure):
    _fields_ = [
        ("product", c_void_p), ("qkv", c_void_p), ("rel_y", c_void_p), ("rel_x", c_void_p),
        ("gh", c_int32), ("gw", c_int32), ("tok_map", c_void_p), ("groups_per_clip", c_int32),
        ("clip_rows", c_int32), ("pad_row", c_void_p), ("a_state", c_void_p), ("a_new", c_void_p),
        ("a_delta", c_void_p), ("idx", c_void_p), ("count", c_void_p),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("Nk", c_int32), ("D", c_int32),
        ("kcap", c_int32), ("store", c_int32), ("gated", c_int32), ("qw", c_int32),
    ]


class AvDesc(Structure):
    _fields_ = [
        ("a1", c_void_p), ("v1", c_void_p), ("a2", c_void_p), ("v2", c_void_p), ("lda", c_int64),
        ("count", c_void_p), ("pv", c_void_p), ("out_f32", c_void_p), ("out_map", c_void_p),
        ("groups_per_clip", c_int32), ("clip_rows", c_int32),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("K", c_int32), ("D", c_int32),
        ("store", c_int32), ("gated", c_int32),
    ]


class SoftmaxAvDesc(Structure):
    _fields_ = [
        ("product", c_void_p), ("qkv", c_void_p), ("rel_y", c_void_p), ("rel_x", c_void_p),
        ("gh", c_int32), ("gw", c_int32), ("a_state", c_void_p), ("idx", c_void_p), ("count", c_void_p),
        ("kcap", c_int32), ("v_delta_t", c_void_p), ("v_old_t", c_void_p), ("pv", c_void_p),
        ("out_f32", c_void_p), ("B", c_int32), ("H", c_int32), ("N", c_int32), ("D", c_int32), ("dh", c_int32),
        ("store", c_int32), ("Nk", c_int32), ("qw", c_int32), ("scale", c_float), ("qk_split", c_int32),
        ("norm_ref", c_void_p), ("norm_parts", c_void_p), ("rel_terms", c_void_p),
    ]


class AttnDenseDesc(Structure):
    _fields_ = [
        ("qkv", c_void_p), ("rel_y", c_void_p), ("rel_x", c_void_p), ("gh", c_int32), ("gw", c_int32), ("qw", c_int32),
        ("tok_map", c_void_p), ("groups_per_clip", c_int32), ("clip_rows", c_int32), ("pad_row", c_void_p),
        ("out_f32", c_void_p), ("product", c_void_p), ("a_state", c_void_p), ("pv", c_void_p),
        ("G", c_int32), ("H", c_int32), ("N", c_int32), ("D", c_int32), ("scale", c_float), ("store", c_int32),
        ("qk_split", c_int32), ("norm_ref", c_void_p), ("norm_parts", c_void_p),
    ]


class AttnStreamDesc(Structure):
    _fields_ = [
        ("qkv", c_void_p), ("rel_terms", c_void_p), ("gh", c_int32), ("gw", c_int32), ("a_state_t", c_void_p),
        ("idx", c_void_p), ("count", c_void_p), ("kcap", c_int32), ("v_delta_t", c_void_p), ("v_old_t", c_void_p),
        ("v_state", c_void_p), ("pv", c_void_p), ("out_f32", c_void_p), ("norm_ref", c_void_p), ("norm_parts", c_void_p),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("D", c_int32), ("store", c_int32), ("scale", c_float),
        ("qk_split", c_int32), ("first", c_int32), ("k_split", c_void_p), ("k_split_ready", c_int32),
        ("kv", c_void_p), ("Nk", c_int32),
    ]


class StreamPrepDesc(Structure):
    _fields_ = [
        ("qkv", c_void_p), ("rel_y", c_void_p), ("rel_x", c_void_p), ("terms", c_void_p), ("k_split", c_void_p),
        ("idx", c_void_p), ("count", c_void_p), ("kcap", c_int32), ("v_state", c_void_p), ("v_delta_t", c_void_p), ("v_old_t", c_void_p),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("D", c_int32), ("gh", c_int32), ("gw", c_int32), ("qw", c_int32), ("store", c_int32),
        ("kv", c_void_p), ("Nk", c_int32),
    ]


class AttnGatedDesc(Structure):
    _fields_ = [
        ("qkv", c_void_p), ("a_tiles", c_void_p), ("idx", c_void_p), ("count", c_void_p), ("kcap", c_int32),
        ("v_state", c_void_p), ("pv", c_void_p), ("out_f32", c_void_p), ("norm_ref", c_void_p), ("norm_parts", c_void_p),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("D", c_int32), ("store", c_int32), ("scale", c_float),
        ("qk_split", c_int32), ("first", c_int32),
    ]


_lib = None


def _bind(lib):
    P, I, F = c_void_p, c_int, c_float
    lib.evt_version.restype = c_int
    lib.evt_last_error_string.restype = c_char_p
    lib.evt_target_arch.restype = c_char_p
    lib.evt_gated_linear_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32]
    lib.evt_gated_linear_workspace_bytes.restype = c_int64
    lib.evt_split_weights_bytes.argtypes = [c_int64, c_int64]
    lib.evt_split_weights_bytes.restype = c_int64
    lib.evt_attention_stream_lds_bytes.argtypes = [c_int32, c_int32, c_int32]
    lib.evt_attention_stream_lds_bytes.restype = c_int64
    lib.evt_prefetch.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    lib.evt_select_prefetch_next.argtypes = [c_void_p, c_int64, c_void_p]
    lib.evt_attention_stream_key_blocks.argtypes = [c_int32, c_int32, c_int32]
    lib.evt_attention_stream_key_blocks.restype = c_int64
    lib.evt_attention_dense_resident.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32]
    lib.evt_attention_dense_resident.restype = c_int
    lib.evt_attention_gated_fits.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32]
    lib.evt_attention_gated_fits.restype = c_int
    lib.evt_attention_gated_tile_bytes.argtypes = [c_int32, c_int32, c_int32]
    lib.evt_attention_gated_tile_bytes.restype = c_int64
    sigs = {
        "evt_row_pass": [P, P, I, P, P, P, F, P, P, P, I, I, P],
        "evt_row_pass_ord": [P, P, I, P, P, P, F, P, P, P, I, I, I, P],
        "evt_select_topk": [P, I, I, I, P, P, P],
        "evt_select_threshold": [P, I, I, F, I, P, P, P, P],
        "evt_select_topk_sq": [P, I, I, I, I, P, P, P],
        "evt_select_threshold_sq": [P, I, I, I, F, I, P, P, P, P],
        "evt_gate_gather_update": [P, P, P, P, I, I, I, I, P, P, I, P],
        "evt_scatter_rows": [P, P, P, P, I, I, I, I, P],
        "evt_gated_linear": [POINTER(LinearDesc), P],
        "evt_gated_linear_big_tile": [POINTER(LinearDesc)],
        "evt_gated_mlp": [POINTER(MlpDesc), P],
        "evt_split_weights": [P, P, c_int64, c_int64, P],
        "evt_qk": [POINTER(QkDesc), P],
        "evt_softmax_gate": [POINTER(SoftmaxDesc), P],
        "evt_v_gate": [P, c_int64, P, P, I, I, I, I, P, P, P, I, I, I, P, I, I, P, P],
        "evt_pool_kv": [P, I, I, I, I, I, I, P, P],
        "evt_pool_index": [P, P, I, I, I, I, I, I, I, I, P, P, P],
        "evt_softmax_av_gated": [POINTER(SoftmaxAvDesc), P],
        "evt_rel_terms": [P, P, P, I, I, I, I, I, I, I, I, P, P],
        "evt_attention_dense": [POINTER(AttnDenseDesc), P],
        "evt_attention_stream": [POINTER(AttnStreamDesc), P],
        "evt_stream_prep": [POINTER(StreamPrepDesc), P],
        "evt_attention_gated": [POINTER(AttnGatedDesc), P],
        "evt_gate_cols": [P, P, P, P, I, I, I, I, I, P, P, I, P],
        "evt_scatter_cols": [P, P, P, P, I, I, I, I, I, P],
        "evt_gate_rows_any": [P, P, P, P, I, I, I, I, I, P, P, I, P],
        "evt_move_rows_any": [P, P, I, I, I, I, I, I, I, P, P],
        "evt_gather_rows_map": [P, P, P, I, I, I, I, I, P, P],
        "evt_scatter_rows_map": [P, P, I, I, I, I, P, P],
        "evt_ats_scores": [P, P, c_int64, c_int64, c_int64, I, I, I, I, I, P, P],
        "evt_ats_stabilize": [P, P, I, I, I, P, P],
        "evt_av": [POINTER(AvDesc), P],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int


def load():
    """Loads libevt_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python eventful-transformer_amd/build.py` "
                "(hipcc --offload-arch=gfx950).  The gated-token path has no non-HIP fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        _bind(lib)
        if lib.evt_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {lib.evt_version()}, this package binds version {ABI_VERSION}: "
                               "rebuild it with `python eventful-transformer_amd/build.py --force`")
        _lib = lib
    return _lib


def is_loaded():
    return _lib is not None


def _check(rc):
    if rc != 0:
        msg = load().evt_last_error_string().decode()
        raise RuntimeError(f"libevt_hip: {msg} (status {rc})")


_lane_stream = {}   # (thread, device, work lane) -> (stream handle, stream) that last launched there


def _stream():
    """The current HIP stream's handle.  Work buffers (`scratch`) are shared by everything one host thread runs in one work
    lane of a device, which is only sound while that work is ordered: when the launching stream CHANGES within a lane, the new
    stream first waits for everything the previous one has been given (one event, only at the switch), so two streams driven
    from one thread can never use a lane's index lists or hidden rows at the same time -- they serialise at the switch instead.
    Run them concurrently with `with _native.lane(i):` around each one's calls.  Inside a graph capture the
    switch is only recorded: torch synchronises the device before a capture begins, and a captured stream runs nothing."""
    st = torch.cuda.current_stream()
    h = st.cuda_stream
    key = (threading.get_ident(), st.device_index, _current_lane())
    last = _lane_stream.get(key)
    if last is None or last[0] != h:
        if last is not None and not torch.cuda.is_current_stream_capturing() and not last[2]:
            st.wait_stream(last[1])
        _lane_stream[key] = (h, st, torch.cuda.is_current_stream_capturing())
    return c_void_p(h)


def require_hip(*tensors):
    """The product path only runs on a HIP device; fail loudly otherwise."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "eventful_transformer (MI355X build): tensors must live on a HIP device; there is no "
                "CPU path in this package (the CPU oracle under oracle/ is test infrastructure only)."
            )


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


def _need_f32(who, **tensors):
    """The C ABI takes these operands as float32 (include/evt_abi.h): weights, biases, LayerNorm / rel-pos parameters, fp32 activations.
    A model converted with .half() / .bfloat16() / .double() would hand over pointers to other element sizes, a model left on the CPU host
    pointers -- an out-of-bounds access on the device (found as GPU memory faults by scripts/probes/weight_dtype_probe.py), so both are
    errors here.  (The per-launch wrappers call `_f32dev` on the operands that can come from a user's module; this form names several.)"""
    for name, t in tensors.items():
        _f32dev(t, who, name)


def _f32dev(t, who, name):
    if t is not None and (t.dtype is not torch.float32 or not t.is_cuda):
        if not t.is_cuda:   # e.g. a model left on the CPU fed device tensors: a host pointer in a device kernel
            raise RuntimeError(f"{who}: `{name}` lives on {t.device}, not on a HIP device: move the model with .to('cuda') (there is no CPU path)")
        raise RuntimeError(f"{who}: `{name}` must be float32, got {t.dtype} (the MI355X path keeps parameters and the residual stream in "
                           f"fp32, like the reference's default; `matmul_2_cast` selects the 16-bit stage)")


def store_code(dtype):
    return _STORE_OF[dtype]


# --------------------------------------------------------------------------------------------------
# scratch pool: fixed-address work buffers shared by all blocks on a device (blocks run one after
# another on one stream, so the hidden/ã/Δã/... scratch never needs to exist per block).  The pool is keyed by
# (name, shape, dtype, device, work lane, host thread) -- NOT by stream or model: two same-shaped models driven by one
# thread share buffers, which is correct because their launches are ordered (`_stream` orders a stream switch within a lane)
# and is what `lane` is for when they should overlap; every host thread has its own buffers.
# --------------------------------------------------------------------------------------------------
_pool = {}
# Frames of one stream captured side by side on several HIP streams (graphs.FrameGraphs.run_pipelined), or independent batches
# overlapped on two streams (bench.py --overlap), each get their own set of work buffers: the lane is part of the pool key.  The
# current lane is per HOST THREAD (two threads inside `with lane(i)` must not overwrite each other's lane mid-call).
_lane_local = threading.local()


def _current_lane():
    return getattr(_lane_local, "index", 0)


class lane:
    """Context manager: scratch buffers requested inside belong to work lane `index` (of the calling host thread)."""

    def __init__(self, index):
        self.index = int(index)

    def __enter__(self):
        self._saved = _current_lane()
        _lane_local.index = self.index
        return self

    def __exit__(self, *exc):
        _lane_local.index = self._saved


def scratch(name, shape, dtype, device):
    key = (name, tuple(shape), dtype, device.index if device.index is not None else torch.cuda.current_device(), _current_lane(),
           threading.get_ident())
    t = _pool.get(key)
    if t is None:
        t = torch.empty(tuple(shape), dtype=dtype, device=device)
        _pool[key] = t
    return t


def clear_scratch():
    _pool.clear()


# --------------------------------------------------------------------------------------------------
# thin wrappers (argument checking beyond dtype/contiguity lives in the C library)
# --------------------------------------------------------------------------------------------------
def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} strides {t.stride()}")
    return t


NORM_ORDERS = {2: 2, 2.0: 2, 1: 1, 1.0: 1, float("inf"): 0}   # policy `order` -> evt_norm_order


def norm_order(order):
    """The reference hands `order` to torch.linalg.vector_norm (policies.py:28,63); the row pass implements 1, 2 and inf."""
    try:
        return NORM_ORDERS[order]
    except (KeyError, TypeError):
        raise NotImplementedError(f"MI355X build: delta norms of order 1, 2 and inf are implemented in the gate kernels, not {order!r}") from None


def row_pass(x, rows, D, res=None, res_rows=0, sum_out=None, ln_w=None, ln_b=None, eps=1e-6, c_out=None, p=None,
             norms=None, order=2):
    # "rows" family (bench.py): algorithmic bytes = every (rows, D) fp32 tensor the pass reads or writes, once (+ the norms)
    _f32dev(ln_w, "evt_row_pass", "ln_w")   # (parameters of the caller's module; the other operands are this package's own buffers or
    _f32dev(ln_b, "evt_row_pass", "ln_b")   #  inputs the block / PositionEncoding entry points have checked)
    _f32dev(res, "evt_row_pass", "res")
    tensors = 1 + (res is not None) + (sum_out is not None) + (c_out is not None) + (p is not None)
    _timed("rows", 4.0 * rows * D * tensors + (4.0 * rows if norms is not None else 0.0),
           lambda: _check(load().evt_row_pass_ord(_p(x), _p(res), res_rows, _p(sum_out), _p(ln_w), _p(ln_b), eps, _p(c_out), _p(p),
                                                  _p(norms), rows, D, norm_order(order), _stream())))


def select_topk(norms, B, N, k, idx, rest=None, parts=0):
    """parts > 0: `norms` is (B, N, parts) partial sums of squares (evt_softmax_av_gated norm_parts)."""
    if parts:
        _check(load().evt_select_topk_sq(_p(norms), parts, B, N, k, _p(idx), _p(rest), _stream()))
    else:
        _check(load().evt_select_topk(_p(norms), B, N, k, _p(idx), _p(rest), _stream()))


def select_threshold(norms, B, N, threshold, kcap, idx, count, rest=None, parts=0):
    if parts:
        _check(load().evt_select_threshold_sq(_p(norms), parts, B, N, float(threshold), kcap, _p(idx), _p(count), _p(rest), _stream()))
    else:
        _check(load().evt_select_threshold(_p(norms), B, N, float(threshold), kcap, _p(idx), _p(count), _p(rest), _stream()))


def gate_gather_update(c, p, idx, count, B, N, D, kcap, c_tilde=None, e_tilde=None, update_p=True):
    _check(load().evt_gate_gather_update(_p(c), _p(p), _p(idx), _p(count), B, N, D, kcap, _p(c_tilde), _p(e_tilde),
                                         int(update_p), _stream()))


def scatter_rows(x, buf, idx, count, B, N, F, kcap):
    _check(load().evt_scatter_rows(_p(x), _p(buf), _p(idx), _p(count), B, N, F, kcap, _stream()))


# bench.py brackets every launch of a workload's dominant kernel family with HIP events on the launch stream:
# set_kernel_events("gemm" | "attn" | "rows", list) -> entries (start_event, end_event, algorithmic work, launches), where
# work is FLOP for "gemm" (K3/K7), bytes for "attn" (the EventfulBlock attention kernels K5+K6 / K8 / K9 / K10) and bytes for "rows"
# (the HBM-bound row kernels: row passes -- residual add, LayerNorm, delta norm, gather-free -- and the stand-alone value gate).
_EVENTS = {}


def set_kernel_events(family, sink):
    if sink is None:
        _EVENTS.pop(family, None)
    else:
        _EVENTS[family] = sink


def _timed(family, work, fn, launches=1, count=None, per_row=0.0, issue=1.0):
    """work: algorithmic FLOP / bytes of the launch; issue: matrix-core products issued per algorithmic product (gemm family).  count (device int32 (B,), threshold policy): the launch touches
    `per_row` more work for every LIVE selected row -- the capacity of the index list says nothing about it -- so the entry's
    work becomes a callable that reads a copy of the counts (taken before the start event) once the timed region is over."""
    sink = _EVENTS.get(family)
    if sink is None:
        return fn()
    if count is not None:
        live = count.clone()
        base = work
        work = lambda: base + per_row * float(live.sum().item())   # noqa: E731
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    sink.append((s, e, work, launches, issue))


def event_work(entry):
    """Algorithmic work of one set_kernel_events entry (evaluates the live-count form; call after the timed region)."""
    return entry[2]() if callable(entry[2]) else entry[2]


def gemm_kernel_name():
    return "gated_linear_pipe_kernel" if GEMM_MODE == "split" else "gated_linear_kernel"


# GEMM arithmetic of K3/K7: "split" = bf16 hi/lo planes, 3 bf16 MFMAs per fp32 product, fp32 accumulate
# (~1e-5 relative to fp32, ~5x the fp32-MFMA rate); "f32" = exact fp32-input MFMA.  EVT_GEMM overrides.
GEMM_MODE = os.environ.get("EVT_GEMM", "split")
# Score arithmetic of the attention kernels follows the GEMM mode: bf16 hi + lo split products, or exact fp32-input MFMA.
QK_SPLIT = GEMM_MODE == "split"
# Module constants, not environment switches: the tests set FUSED_QK = False to run the K4 + stored-state chain -- the path of
# every shape the in-kernel-score kernels do not cover -- on a shape they do cover, and compare.
FUSED_QK = True          # gated frames: q.k^T inside K5+K6 / K9 / K10 (no K4, no score-state traffic)
DENSE_FUSED = True       # dense / windowed attention of <= 256-token groups in one launch (K8) instead of the K4 + K5 + K6 chain


def split_weight(W):
    """Fresh bf16 hi/lo planes of a (out, in) weight matrix via evt_split_weights, or None when the split path does
    not apply.  Layout "hl32" (include/evt_abi.h): (out, ceil(in / 32), 2, 32) -- per row and group of 32 input
    features, 32 hi values then 32 lo values (one 128-byte line).  Callers cache the result (CountedLinear.split_planes)."""
    _need_f32("evt_split_weights", weight=W)
    if GEMM_MODE != "split" or W.ndim != 2 or (W.shape[-1] % 8) != 0:
        return None
    rows, cols = W.shape
    Wc = W.detach()
    if not Wc.is_contiguous():
        Wc = Wc.contiguous()
    planes = torch.empty((rows, (cols + 31) // 32, 2, 32), dtype=torch.bfloat16, device=W.device)
    assert planes.numel() * 2 == load().evt_split_weights_bytes(rows, cols)
    _check(load().evt_split_weights(_p(Wc), _p(planes), rows, cols, _stream()))
    return planes


SPLITK_WS_LIMIT = 1 << 30


def _splitk_workspace(device, has_count, *shapes):
    """Split-K partial-sum workspace for small launches (evt_gated_linear_workspace_bytes); (None, 0) if unused."""
    need = max(load().evt_gated_linear_workspace_bytes(*s, int(has_count)) for s in shapes)
    if need == 0 or need > SPLITK_WS_LIMIT:
        return None, 0
    return scratch("splitk_ws", (need // 4,), torch.float32, device), need


def gated_linear_big_tile(lda, gathered, a_rows, ldo, scattered, o_rows, has_count, B, kcap, K, Nout, has_split=True):
    """Tile configuration of the persistent 256-row kernel a launch of this shape runs on (0: the 128x128 kernel): the
    launches that accept bf16 activations (`a_bf16`).  Shape-only -- the descriptor's pointers are only compared with NULL."""
    one = ctypes.c_void_p(1)
    d = LinearDesc(one, lda, one if gathered else None, a_rows, one, one, one, ldo, one if scattered else None, o_rows,
                   one if has_count else None, None, B, kcap, K, Nout, ACT_NONE, one if has_split else None, None, 0, 0)
    return int(load().evt_gated_linear_big_tile(ctypes.byref(d)))


def gated_linear(A, lda, a_idx, a_rows, W, bias, out, ldo, o_idx, o_rows, count, p_upd, B, kcap, K, Nout, act=ACT_NONE,
                 W_split=None, a_bf16=False):
    """a_bf16: A is a bfloat16 tensor of exactly representable activations (the A.v state; see evt_abi.h)."""
    _f32dev(bias, "evt_gated_linear", "bias")
    if W_split is None:   # (the split planes were built from a weight that evt_split_weights' wrapper checked)
        _f32dev(W, "evt_gated_linear", "weight")
    if not a_bf16:
        _f32dev(A, "evt_gated_linear", "A")
    ws, ws_bytes = _splitk_workspace(out.device, count is not None, (B, kcap, K, Nout)) if W_split is not None else (None, 0)
    d = LinearDesc(_p(A), lda, _p(a_idx), a_rows, _p(W), _p(bias), _p(out), ldo, _p(o_idx), o_rows, _p(count),
                   _p(p_upd), B, kcap, K, Nout, act, _p(W_split), _p(ws), ws_bytes, int(a_bf16))
    # split arithmetic: 3 bf16 MFMA products per fp32 product; 2 when the activations are exactly bf16 (no lo plane)
    issue = (2.0 if a_bf16 else 3.0) if (GEMM_MODE == "split" and W_split is not None) else 1.0
    _timed("gemm", 2.0 * B * kcap * K * Nout, lambda: _check(load().evt_gated_linear(ctypes.byref(d), _stream())), issue=issue)


def gated_mlp(A, lda, idx, rows, W1, b1, W2, b2, hidden, out, ldo, count, p_upd, B, kcap, D, Dh, W1_split=None,
              W2_split=None):
    _f32dev(b1, "evt_gated_mlp", "bias_1")
    _f32dev(b2, "evt_gated_mlp", "bias_2")
    _f32dev(A, "evt_gated_mlp", "A")
    if W1_split is None or W2_split is None:
        _f32dev(W1, "evt_gated_mlp", "weight_1")
        _f32dev(W2, "evt_gated_mlp", "weight_2")
    s1, s2 = W1_split, W2_split
    if s1 is None or s2 is None:
        s1 = s2 = None
    ws, ws_bytes = _splitk_workspace(out.device, count is not None, (B, kcap, D, Dh), (B, kcap, Dh, D)) if s1 is not None else (None, 0)
    d = MlpDesc(_p(A), lda, _p(idx), rows, _p(W1), _p(b1), _p(W2), _p(b2), _p(hidden), _p(out), ldo, _p(count),
                _p(p_upd), B, kcap, D, Dh, _p(s1), _p(s2), _p(ws), ws_bytes)
    _timed("gemm", 4.0 * B * kcap * D * Dh, lambda: _check(load().evt_gated_mlp(ctypes.byref(d), _stream())), launches=2,
           issue=3.0 if (GEMM_MODE == "split" and s1 is not None) else 1.0)


def _ptr_off(t, elems):
    return c_void_p(t.data_ptr() + 4 * elems)


def qk_packed(qkv, B, N, D, H, scale, product, idx=None, count=None, kcap=0, tok_map=None, groups_per_clip=1,
              clip_rows=0, pad_row=None, kv=None, Nk=None, idx_k=None, count_k=None, kcap_k=0, idx_rest=None):
    """K4 on the packed (B, rows, 3D) token buffer; idx given -> delta update of rows idx / columns idx_k.
    kv: pooled (B, Nk, 2D) key/value buffer (evt_pool_kv) -- keys then come from it, with their own index list."""
    dh = D // H
    rows = clip_rows if tok_map is not None else N
    if kv is None:
        kptr, k_bs, k_rs, Nk_ = _ptr_off(qkv, D), rows * 3 * D, 3 * D, N
        idx_k, count_k, kcap_k = idx, count, kcap
    else:
        kptr, k_bs, k_rs, Nk_ = _p(kv), Nk * 2 * D, 2 * D, Nk
    d = QkDesc(_p(qkv), rows * 3 * D, dh, 3 * D, kptr, k_bs, dh, k_rs, _p(product),
               _p(idx), _p(count), kcap, _p(idx_rest), _p(idx_k), _p(count_k), kcap_k, _p(tok_map), groups_per_clip,
               _p(pad_row), None if pad_row is None else _ptr_off(pad_row, D), B, H, N, Nk_, dh, float(scale),
               int(idx is not None), int(QK_SPLIT))
    _check(load().evt_qk(ctypes.byref(d), _stream()))


def qk_strided(q, k, product, scale, idx_q=None, count_q=None, kcap_q=0, idx_k=None, count_k=None, kcap_k=0):
    """K4 on free-standing contiguous q (B,H,Nq,dh) and k (B,H,Nk,dh) tensors."""
    B, H, Nq, dh = q.shape
    Nk = k.shape[2]
    d = QkDesc(_p(q), H * Nq * dh, Nq * dh, dh, _p(k), H * Nk * dh, Nk * dh, dh, _p(product),
               _p(idx_q), _p(count_q), kcap_q, None, _p(idx_k), _p(count_k), kcap_k, None, 1, None, None,
               B, H, Nq, Nk, dh, float(scale), int(idx_q is not None), 0)   # stand-alone MatmulBuffer API: exact fp32 products
    _check(load().evt_qk(ctypes.byref(d), _stream()))


def softmax_gate(product, a_state, B, H, N, Nk, D, store, qkv=None, rel_y=None, rel_x=None, gh=0, gw=0, tok_map=None,
                 groups_per_clip=1, clip_rows=0, pad_row=None, a_new=None, a_delta=None, idx=None, count=None, kcap=0,
                 gated=False, qw=None):
    d = SoftmaxDesc(_p(product), _p(qkv), _p(rel_y), _p(rel_x), gh, gw, _p(tok_map), groups_per_clip, clip_rows,
                    _p(pad_row), _p(a_state), _p(a_new), _p(a_delta), _p(idx), _p(count), B, H, N, Nk, D, kcap, store,
                    int(gated), gw if qw is None else qw)
    _check(load().evt_softmax_gate(ctypes.byref(d), _stream()))


def v_gate(qkv, idx, count, B, N, D, kcap, v_state, v_delta, v_old, store, gated, tok_map=None, groups_per_clip=1,
           clip_rows=0, pad_row=None, transposed=False, v_offset=None, v_rs=None):
    """K6a.  Default source: the value slice of the packed (B,N,3D) buffer `qkv`; with v_offset / v_rs the value
    slice of any row-major buffer (e.g. the pooled (B,Nk,2D) buffer: v_offset=D, v_rs=2D)."""
    off = 2 * D if v_offset is None else v_offset
    rs = 3 * D if v_rs is None else v_rs
    pad = None if pad_row is None else _ptr_off(pad_row, off)
    es = 4 if store == EVT_F32 else 2
    n_rows = B * (kcap if gated else N)   # (capacity for a device-side count: the launch is priced by its list, the kernel skips dead rows)
    _timed("rows", n_rows * D * (4.0 + (4.0 * es if gated else 1.0 * es)),
           lambda: _check(load().evt_v_gate(_ptr_off(qkv, off), rs, _p(idx), _p(count), B, N, D, kcap, _p(v_state), _p(v_delta),
                                            _p(v_old), store, int(gated), int(transposed), _p(tok_map), groups_per_clip, clip_rows,
                                            pad, _stream())))


def pool_kv(qkv, B, qh, qw, D, p0, p1, kv):
    _check(load().evt_pool_kv(_p(qkv), B, qh, qw, D, p0, p1, _p(kv), _stream()))


def pool_index(idx, count, B, kcap, qw, p0, p1, kw, Nk, kcap_k, idx_k, count_k):
    _check(load().evt_pool_index(_p(idx), _p(count), B, kcap, qw, p0, p1, kw, Nk, kcap_k, _p(idx_k), _p(count_k),
                                 _stream()))


def rel_terms(qkv, rel_y, rel_x, B, H, N, D, gh, gw, qw, out, split=None):
    """Decomposed rel-pos terms of every query token (utils.py:159-168): out (B,H,N,gh+gw), read by the fused attention
    kernels instead of recomputing them per 32-row workgroup.  split (default: the scores' arithmetic, EVT_QK_SPLIT): bf16
    hi/lo MFMA products vs fp32 FMA chains."""
    _check(load().evt_rel_terms(_p(qkv), _p(rel_y), _p(rel_x), B, H, N, D, gh, gw, qw,
                                int(QK_SPLIT if split is None else split), _p(out), _stream()))


def softmax_av_gated(product, a_state, idx, count, kcap, v_delta_t, v_old_t, pv, out_f32, B, H, N, D, store,
                     qkv=None, rel_y=None, rel_x=None, gh=0, gw=0, Nk=None, qw=None, scale=0.0, qk_split=None,
                     norm_ref=None, norm_parts=None, rel_terms=None):
    """K5+K6.  product None: the score rows are computed in the kernel from `qkv` ((q / scale) k^T; head dim 64,
    N == Nk <= 256) instead of being read from the q.k^T state."""
    d = SoftmaxAvDesc(_p(product), _p(qkv), _p(rel_y), _p(rel_x), gh, gw, _p(a_state), _p(idx), _p(count), kcap,
                      _p(v_delta_t), _p(v_old_t), _p(pv), _p(out_f32), B, H, N, D, D // H, store,
                      N if Nk is None else Nk, gw if qw is None else qw, float(scale),
                      int(QK_SPLIT if qk_split is None else qk_split), _p(norm_ref), _p(norm_parts), _p(rel_terms))
    # algorithmic bytes: q.k^T state read once, gate-reference columns read + rewritten, v delta / old reads,
    # A.v state read-modify-write, fp32 output
    es, nk = (4 if store == EVT_F32 else 2), (N if Nk is None else Nk)
    state_read = 4.0 * H * N * nk if product is not None else 8.0 * N * D   # QK mode reads q and k instead of the state
    per_row = 2.0 * es * H * N + 2.0 * es * D          # per selected key: its gate-reference column read + rewritten, v delta / old
    fixed = B * (state_read + 2.0 * es * N * D + (4.0 * N * D if out_f32 is not None else 0.0))
    _timed("attn", fixed + (B * kcap * per_row if count is None else 0.0),
           lambda: _check(load().evt_softmax_av_gated(ctypes.byref(d), _stream())), count=count, per_row=per_row)


def fused_qk_fits(N, Nk, D, H, kcap):
    """K5+K6 can compute the score rows itself (product=None): head dim 64, un-pooled, at most 256 tokens."""
    return FUSED_QK and D == 64 * H and N == Nk and 0 < N <= 256 and kcap > 0


STREAM_MIN_N = 257   # evt_attention_stream takes more than 256 tokens (K8 / K10 / the in-LDS QK mode cover the rest)


LDS_PER_CU = 160 * 1024


def attention_stream_fits(N, D, H, store=EVT_F32, gh=0, gw=0):
    """evt_attention_stream: head dim 64, more than 256 tokens (K8 / the in-LDS QK mode cover the rest), and a 32-row tile with
    the rel-pos terms of a gh x gw key grid inside a CU's LDS (evt_attention_stream_lds_bytes; larger grids take the
    evt_qk + evt_softmax_av_gated path)."""
    if not (FUSED_QK and D == 64 * H and STREAM_MIN_N <= N <= 32767):   # (32-bit offsets into a head's N x N reference)
        return False
    return 0 < load().evt_attention_stream_lds_bytes(store, gh, gw) <= LDS_PER_CU




def prefetch(t, sink):
    """Read a read-only tensor (weight planes) into the memory-side cache on the CURRENT stream (evt_prefetch); graphs.py issues it
    on a side stream one block ahead.  sink: an int32 tensor of >= 1 element on the same device."""
    _check(load().evt_prefetch(_p(t), t.numel() * t.element_size(), _p(sink), _stream()))


# One stream: the selection launches carry prefetch riders for the weight planes of a gated linear a few launches ahead
# (evt_select_prefetch_next), below PREFETCH_MAX_ROWS token rows per launch.
PREFETCH_MAX_ROWS = 8192


def select_prefetch_next(t):
    """Arm the next select launch of this thread with a prefetch of tensor t (weight planes)."""
    if t is None:
        return
    sink = scratch("prefetch_sink", (4,), torch.int32, t.device)
    _check(load().evt_select_prefetch_next(_p(t), t.numel() * t.element_size(), _p(sink)))


def k_split_plane(qkv, B, H, N, gh=0, gw=0):
    """The key-plane workspace of evt_attention_stream / evt_stream_prep (4 KB per 16-key block and head; with a rel-pos key grid
    every grid row starts a new block: evt_attention_stream_key_blocks)."""
    blocks = int(load().evt_attention_stream_key_blocks(N, gh, gw))
    return scratch("k_split", (B, H, blocks, 2048), torch.bfloat16, qkv.device)


def stream_prep_fits(D, H, kcap, has_rel):
    return QK_SPLIT and has_rel and D == 64 * H and kcap > 0 and kcap % 8 == 0


def stream_prep(qkv, rel_y, rel_x, terms, idx, count, kcap, v_state, v_delta_t, v_old_t, B, H, N, D, gh, gw, qw, store, kv=None, Nk=None):
    """rel-pos terms + key plane + transposed value gate of a gated frame in one launch; attention_stream(..., k_split_ready=True)
    then skips its key-plane pre-kernel.  kv, Nk: pooled keys / values (idx / count / kcap / v_state are then the pooled ones)."""
    nk = N if kv is None else int(Nk)
    d = StreamPrepDesc(_p(qkv), _p(rel_y), _p(rel_x), _p(terms), _p(k_split_plane(qkv, B, H, nk, gh, gw)), _p(idx), _p(count), kcap, _p(v_state),
                       _p(v_delta_t), _p(v_old_t), B, H, N, D, gh, gw, qw, store, _p(kv), 0 if kv is None else nk)
    _check(load().evt_stream_prep(ctypes.byref(d), _stream()))


def attention_stream(qkv, a_state_t, pv, B, H, N, D, scale, store, first, rel_terms=None, gh=0, gw=0, idx=None, count=None,
                     kcap=0, v_delta_t=None, v_old_t=None, v_state=None, out_f32=None, norm_ref=None, norm_parts=None,
                     qk_split=None, k_split_ready=False, kv=None, Nk=None):
    """K5+K6 / first frame for N > 256 with in-kernel scores; a_state_t is the TRANSPOSED gate reference (B,H,Nk,N).
    kv, Nk: pooled keys / values (the (B,Nk,2D) buffer of pool_kv); idx / count / kcap, v_state, v_delta_t / v_old_t are then the
    pooled ones and gh x gw == Nk."""
    split = int(QK_SPLIT if qk_split is None else qk_split)
    nk = N if kv is None else int(Nk)
    # split arithmetic: the frame's key rows as bf16 hi / lo fragments, written by the call's pre-kernel (4 KB per 16 keys and head)
    ksp = k_split_plane(qkv, B, H, nk, *((gh, gw) if rel_terms is not None else (0, 0))) if split else None
    d = AttnStreamDesc(_p(qkv), _p(rel_terms), gh, gw, _p(a_state_t), _p(idx), _p(count), kcap, _p(v_delta_t), _p(v_old_t),
                       _p(v_state), _p(pv), _p(out_f32), _p(norm_ref), _p(norm_parts), B, H, N, D, store, float(scale),
                       split, int(first), _p(ksp), int(bool(k_split_ready) and bool(split)), _p(kv), 0 if kv is None else nk)
    # algorithmic bytes: q, k read once per clip (8ND), rel terms, gate-reference columns read + rewritten (first frame:
    # written whole), v pieces, A.v state read-modify-write (first frame: v state read, state written), fp32 output
    # The gated form is priced by the LIVE selected-key count (threshold policy: `count` on the device, capacity N), never by kcap.
    es = 4 if store == EVT_F32 else 2
    rel_b = 4.0 * H * N * (gh + gw) if rel_terms is not None else 0.0
    call = lambda: _check(load().evt_attention_stream(ctypes.byref(d), _stream()))   # noqa: E731
    if first:
        _timed("attn", B * (4.0 * (N + nk) * D + rel_b + 1.0 * es * H * N * nk + es * (N + nk) * D + 4.0 * N * D), call)
    else:
        per_row = 2.0 * es * H * N + 2.0 * es * D
        fixed = B * (4.0 * (N + nk) * D + rel_b + 2.0 * es * N * D + (4.0 * N * D if out_f32 is not None else 0.0))
        _timed("attn", fixed + (B * kcap * per_row if count is None else 0.0), call, count=count, per_row=per_row)


# ------------------------------------------------------------------------------------------------
# K10 (evt_attention_gated): EventfulBlock's attention for <= 256 tokens, one workgroup per (clip, head), value gate included.
# The gate reference lives in a TILED layout (include/evt_abi.h): (B, H, NT, NT, 2, 2, 32, 2, 4) = per (row tile, key block) a
# 2 KB tile [t = c / 16][half = c / 4 % 2][row][g2 = c / 8 % 2][e = c % 4] for key-in-block c.
# ------------------------------------------------------------------------------------------------
def attention_gated_fits(N, D, H, store):
    return bool(load().evt_attention_gated_fits(N, D, H, store, int(QK_SPLIT))) and FUSED_QK


def gated_tiles_empty(B, H, N, dtype, device):
    nt = (N + 31) // 32
    t = torch.empty((B, H, nt, nt, 2, 2, 32, 2, 4), dtype=dtype, device=device)
    assert t.numel() * t.element_size() == load().evt_attention_gated_tile_bytes(B, H, N)
    return t


def tiles_to_logical(tiles, N):
    """Tiled gate reference -> the logical (B, H, N, N) tensor (a copy)."""
    B, H, nt = tiles.shape[:3]
    return tiles.permute(0, 1, 2, 6, 3, 4, 7, 5, 8).reshape(B, H, nt * 32, nt * 32)[:, :, :N, :N].contiguous()


def logical_to_tiles(p, tiles):
    """Writes a logical (B, H, N, N) tensor into the tiled gate reference (padding rows / keys become zero)."""
    B, H, nt = tiles.shape[:3]
    N = p.shape[-1]
    padded = torch.zeros((B, H, nt * 32, nt * 32), dtype=tiles.dtype, device=tiles.device)
    padded[:, :, :N, :N] = p.to(device=tiles.device, dtype=tiles.dtype)
    tiles.permute(0, 1, 2, 6, 3, 4, 7, 5, 8).copy_(padded.view(B, H, nt, 32, nt, 2, 2, 2, 4))


def attention_gated(qkv, a_tiles, v_state, pv, B, H, N, D, scale, store, first, idx=None, count=None, kcap=0, out_f32=None,
                    norm_ref=None, norm_parts=None):
    d = AttnGatedDesc(_p(qkv), _p(a_tiles), _p(idx), _p(count), kcap, _p(v_state), _p(pv), _p(out_f32), _p(norm_ref), _p(norm_parts),
                      B, H, N, D, store, float(scale), 1, int(first))
    # algorithmic bytes: q, k read once (8 N D); per selected key its reference column read + rewritten, its value row read and
    # its value-reference row read-modify-written; A.v state read-modify-write (first frame: written); fp32 output
    es = 2
    call = lambda: _check(load().evt_attention_gated(ctypes.byref(d), _stream()))   # noqa: E731
    out_b = 4.0 * N * D if out_f32 is not None else 0.0
    if first:
        _timed("attn", B * (12.0 * N * D + 1.0 * es * H * N * N + 2.0 * es * N * D + out_b), call)
    else:
        per_row = 2.0 * es * H * N + 4.0 * D + 2.0 * es * D
        fixed = B * (8.0 * N * D + 2.0 * es * N * D + out_b)
        _timed("attn", fixed + (B * kcap * per_row if count is None else 0.0), call, count=count, per_row=per_row)


# ------------------------------------------------------------------------------------------------
# Index-structured operations off the fused path (evt_gather.hip, evt_ats.hip): stand-alone gates / buffers, window partition, ATS
# ------------------------------------------------------------------------------------------------
def gate_cols(c, p, idx, count, Bp, R, N, kcap, c_tilde=None, e_tilde=None, update_p=True):
    _check(load().evt_gate_cols(_p(c), _p(p), _p(idx), _p(count), Bp, R, N, kcap, store_code(c.dtype), _p(c_tilde), _p(e_tilde), int(update_p), _stream()))


def scatter_cols(x, buf, idx, count, Bp, R, N, kcap):
    _check(load().evt_scatter_cols(_p(x), _p(buf), _p(idx), _p(count), Bp, R, N, kcap, store_code(buf.dtype), _stream()))


def gate_rows_any(c, p, idx, count, Bp, N, F, kcap, c_tilde=None, e_tilde=None, update_p=True):
    _check(load().evt_gate_rows_any(_p(c), _p(p), _p(idx), _p(count), Bp, N, F, kcap, store_code(c.dtype), _p(c_tilde), _p(e_tilde), int(update_p), _stream()))


def move_rows_any(x, index_map, B, N, F, n, out, rep=1, scatter=False):
    _check(load().evt_move_rows_any(_p(x), _p(index_map), B, N, F, n, rep, int(scatter), store_code(x.dtype), _p(out), _stream()))


def gather_rows_map(x, index_map, B, N, F, n_out, out, pad_row=None, map_per_batch=False):
    _check(load().evt_gather_rows_map(_p(x), _p(index_map), _p(pad_row), B, N, F, n_out, int(map_per_batch), _p(out), _stream()))


def scatter_rows_map(x, index_map, B, n_in, N, F, out):
    _check(load().evt_scatter_rows_map(_p(x), _p(index_map), B, n_in, N, F, _p(out), _stream()))


def ats_scores(a, v, B, H, N, dh, scores):
    """a (B,H,N,N) contiguous, v (B,H,N,dh) any strides with contiguous channels, same dtype -> scores (H,N) fp32."""
    assert a.dtype == v.dtype and a.is_contiguous() and v.stride(-1) == 1
    _check(load().evt_ats_scores(_p(a), _p(v), v.stride(0), v.stride(1), v.stride(2), B, H, N, dh, store_code(a.dtype), _p(scores), _stream()))


def ats_stabilize(last, now, rows, n, N, out):
    _check(load().evt_ats_stabilize(_p(last), _p(now), rows, n, N, _p(out), _stream()))


def set_cu_budget(cus):
    """Persistent launches of this host thread size their grids for `cus` compute units (a stream with a CU mask); 0: the whole device."""
    lib = load()
    lib.evt_set_cu_budget.argtypes = [c_int32]
    _check(lib.evt_set_cu_budget(int(cus)))


def attention_dense_fits(N, D, H):
    """K8 handles groups of <= 256 tokens at head dim 64."""
    return D == 64 * H and 0 < N <= 256


def attention_dense_resident(N, D, H, store, gh=0, gw=0, qk_split=None):
    """True when an evt_attention_dense launch of this shape without state outputs runs the resident kernel (the one that can
    also emit the next gate's per-head delta norms, `norm_ref` / `norm_parts`)."""
    if D != 64 * H:
        return False
    return bool(load().evt_attention_dense_resident(N, gh, gw, store, int(QK_SPLIT if qk_split is None else qk_split)))


def attention_dense(qkv, G, H, N, D, scale, store, out_f32=None, rel_y=None, rel_x=None, gh=0, gw=0, qw=0, tok_map=None,
                    groups_per_clip=1, clip_rows=0, pad_row=None, product=None, a_state=None, pv=None, qk_split=None,
                    norm_ref=None, norm_parts=None):
    """K8: q.k^T + rel-pos + softmax + A.v of whole groups in one launch (scores never reach HBM).
    norm_ref / norm_parts (resident kernel only): per (token, head) || out - ref ||^2 for the gate that consumes `out`."""
    d = AttnDenseDesc(_p(qkv), _p(rel_y), _p(rel_x), gh, gw, qw, _p(tok_map), groups_per_clip, clip_rows, _p(pad_row),
                      _p(out_f32), _p(product), _p(a_state), _p(pv), G, H, N, D, float(scale), store,
                      int(QK_SPLIT if qk_split is None else qk_split), _p(norm_ref), _p(norm_parts))
    es = 4 if store == EVT_F32 else 2
    work = G * (12.0 * N * D + 4.0 * N * D) + (G * H * N * N * (4.0 + es) + G * N * D * es if product is not None else 0.0)
    if product is not None:   # only the state-producing (global-block) form is part of the "attn" family
        _timed("attn", work, lambda: _check(load().evt_attention_dense(ctypes.byref(d), _stream())))
    else:
        _check(load().evt_attention_dense(ctypes.byref(d), _stream()))


def av(a1, v1, lda, B, H, N, K, D, store, pv=None, out_f32=None, a2=None, v2=None, count=None, gated=False,
       out_map=None, groups_per_clip=1, clip_rows=0):
    d = AvDesc(_p(a1), _p(v1), _p(a2), _p(v2), lda, _p(count), _p(pv), _p(out_f32), _p(out_map), groups_per_clip,
               clip_rows, B, H, N, K, D, store, int(gated))
    _check(load().evt_av(ctypes.byref(d), _stream()))
