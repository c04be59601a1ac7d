"""CPU: C-ABI surface, host-side API parity with the reference's package, and the fail-loudly rule."""
import ctypes
import inspect
import os
import re

import pytest
import torch

import eventful_oracle as O
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_header_symbols_are_exported():
    from eventful_transformer import _native
    header = open(os.path.join(ROOT, "include", "evt_abi.h")).read()
    declared = re.findall(r"^EVT_API\s+[\w\s\*]+?\b(evt_\w+)\s*\(", header, flags=re.M)
    assert len(declared) >= 14 and len(set(declared)) == len(declared)
    assert set(declared) == set(_native.ABI_SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/evt_abi.h but not exported"
    lib.evt_version.restype = ctypes.c_int
    lib.evt_target_arch.restype = ctypes.c_char_p
    assert lib.evt_version() == _native.ABI_VERSION == 9 and lib.evt_target_arch() == b"gfx950"
    # the boundary's documentation names every entry point: the binding guide, and a header comment citing the reference lines it replaces
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert [n for n in declared if n not in guide] == []
    assert header.count("modules.py:") + header.count("blocks.py:") + header.count("policies.py:") + header.count("utils.py:") >= 40


def test_abi_argument_errors_without_gpu():
    """Argument validation runs before any HIP call, so it is checkable on a GPU-less box."""
    from eventful_transformer import _native
    lib = _native.load()
    rc = lib.evt_select_topk(None, 1, 8, 2, None, None, None)
    assert rc == -1 and b"null pointer" in lib.evt_last_error_string()
    rc = lib.evt_row_pass(ctypes.c_void_p(16), None, 0, None, None, None, 1e-6, None, None, None, 4, 6, None)
    assert rc == -2 and b"multiple of 4" in lib.evt_last_error_string()
    rc = lib.evt_gated_linear(None, None)
    assert rc == -1
    # shape-only query (ABI 6): ViTDet's 64 x 64 key grid fits a CU's LDS with every store type, an absurd grid does not
    for store in (_native.EVT_F32, _native.EVT_BF16, _native.EVT_F16):
        assert 0 < lib.evt_attention_stream_lds_bytes(store, 64, 64) <= _native.LDS_PER_CU
    assert lib.evt_attention_stream_lds_bytes(_native.EVT_F32, 1000, 1000) > _native.LDS_PER_CU
    assert lib.evt_attention_stream_lds_bytes(7, 8, 8) < 0
    # key plane: 16-key blocks; with a rel-pos key grid every grid row starts a new block
    assert lib.evt_attention_stream_key_blocks(197, 0, 0) == 13 and lib.evt_attention_stream_key_blocks(1764, 42, 42) == 42 * 3
    assert lib.evt_attention_stream_key_blocks(4096, 64, 64) == 256 and lib.evt_attention_stream_key_blocks(0, 0, 0) < 0
    assert not _native.attention_stream_fits(1000000, 768, 12, _native.EVT_F32, 1000, 1000)
    assert _native.attention_stream_fits(4096, 768, 12, _native.EVT_BF16, 64, 64)


def test_every_descriptor_entry_point_rejects_null_and_zeroed_descriptors_without_a_gpu():
    """The C ABI is the drop-in boundary (include/evt_abi.h): a binding written in another language will get arguments wrong.  Every
    entry point that takes a descriptor must answer a NULL descriptor and an all-zero descriptor (null pointers, zero sizes) with an
    error code and a message -- before any HIP call, so this runs on a GPU-less box -- never with a crash."""
    import ctypes
    from eventful_transformer import _native as n
    lib = n.load()
    pairs = [("evt_gated_linear", n.LinearDesc), ("evt_gated_mlp", n.MlpDesc), ("evt_qk", n.QkDesc), ("evt_softmax_gate", n.SoftmaxDesc),
             ("evt_av", n.AvDesc), ("evt_softmax_av_gated", n.SoftmaxAvDesc), ("evt_attention_dense", n.AttnDenseDesc),
             ("evt_attention_stream", n.AttnStreamDesc), ("evt_stream_prep", n.StreamPrepDesc), ("evt_attention_gated", n.AttnGatedDesc)]
    for name, cls in pairs:
        fn = getattr(lib, name)
        assert fn(None, None) < 0, name
        assert lib.evt_last_error_string(), name
        d = cls()
        ctypes.memset(ctypes.byref(d), 0, ctypes.sizeof(d))
        assert fn(ctypes.byref(d), None) < 0, f"{name} accepted an all-zero descriptor"
        assert lib.evt_last_error_string(), name
    # positional entry points: null pointers / non-positive sizes
    assert lib.evt_gate_cols(None, None, None, None, 1, 1, 1, 1, n.EVT_F32, None, None, 0, None) < 0
    assert lib.evt_scatter_cols(None, None, None, None, 1, 1, 1, 1, n.EVT_F32, None) < 0
    assert lib.evt_gate_rows_any(None, None, None, None, 1, 1, 1, 1, n.EVT_F32, None, None, 0, None) < 0
    assert lib.evt_ats_scores(None, None, 0, 0, 0, 1, 1, 2, 1, n.EVT_F32, None, None) < 0
    assert lib.evt_ats_stabilize(None, None, 1, 1, 1, None, None) < 0
    assert lib.evt_attention_gated_tile_bytes(-1, 1, 1) < 0 and lib.evt_attention_gated_tile_bytes(1, 12, 197) == 12 * 49 * 2048
    assert lib.evt_attention_gated_fits(197, 768, 12, n.EVT_BF16, 1) == 1 and lib.evt_attention_gated_fits(257, 768, 12, n.EVT_BF16, 1) == 0
    assert lib.evt_attention_gated_fits(197, 768, 12, n.EVT_F32, 1) == 0 and lib.evt_attention_gated_fits(197, 768, 8, n.EVT_BF16, 1) == 0


def test_splitk_workspace_query_is_shape_only():
    """Split-K is chosen from (rows, K, Nout) alone: few-tile launches get S planes, chip-filling ones none."""
    from eventful_transformer import _native
    q = _native.load().evt_gated_linear_workspace_bytes
    assert q(1, 256, 768, 2304, 0) == 6 * 256 * 2304 * 4      # 36 tiles, 24 k-tiles -> 6 splits of 4
    assert q(1, 256, 3072, 768, 0) == 16 * 256 * 768 * 4      # 12 tiles, 96 k-tiles -> 16 splits of 6
    assert q(256, 128, 768, 2304, 0) == 0                     # 4608 tiles: single pass
    assert q(1, 12, 64, 192, 0) == 0                          # K too short to split
    assert q(0, 128, 768, 768, 0) == 0 and q(1, 128, 0, 768, 1) == 0
    # per-clip counts: the device picks S, the workspace covers the largest one (a single live row tile)
    assert q(1, 4096, 768, 2304, 1) == 6 * 4096 * 2304 * 4 and q(1, 4096, 768, 2304, 0) == 0
    assert q(64, 4096, 768, 2304, 1) == 0                  # too many clips for the per-workgroup count scan


def test_product_path_fails_loudly_on_cpu_tensors():
    from eventful_transformer import blocks, modules, policies
    blk = blocks.EventfulBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4).eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        blk(torch.zeros(1, 37, 64))
    with pytest.raises(RuntimeError, match="HIP device"):
        modules.TokenGate()(torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match="HIP device"):
        policies.TokenNormTopK(k=2)(torch.zeros(1, 4, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "eventful-transformer_amd", "eventful_transformer")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("CPU oracle under oracle/ is test infrastructure only", "") or \
                not re.search(r"^\s*(import|from)\s+\S*oracle", src, flags=re.M), fn
            assert not re.search(r"^\s*(import|from)\s+(eventful_oracle|_refimport)", src, flags=re.M), fn


def test_state_dict_keys_and_constructor_contract():
    from eventful_transformer import blocks
    from eventful_transformer.backbones import ViTBackbone
    blk = blocks.EventfulBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, relative_embedding_size=(6, 6),
                               matmul_2_cast="bfloat16", gate_before_ln=False, stgt=False)
    assert set(blk.state_dict().keys()) == set(O.PARAM_KEYS) | {"relative_position.y_embedding",
                                                                 "relative_position.x_embedding"}
    assert blk.qkv.weight.shape == (192, 64) and blk.mlp_1.weight.shape == (256, 64)
    assert blk.relative_position.y_embedding.shape == (11, 16)
    for name in ("qkv_gate", "qkv_accumulator", "projection_gate", "projection_accumulator", "mlp_gate",
                 "mlp_accumulator", "matmul_accumulator_1", "v_gate", "matmul_gate", "matmul_accumulator_2"):
        assert hasattr(blk, name), name
    sig = inspect.signature(blocks.Block.__init__)
    assert list(sig.parameters)[1:] == ["dim", "heads", "input_size", "mlp_ratio", "ats_fraction", "drop_path_rate",
                                        "relative_embedding_size", "matmul_2_cast", "pool_size", "window_size"]
    with pytest.raises(AssertionError):
        blocks.EventfulBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, window_size=(3, 3))  # blocks.py:485
    winpool = blocks.Block(dim=64, heads=4, input_size=(8, 8), mlp_ratio=4, pool_size=2, window_size=(4, 4))   # blocks.py:308
    assert winpool.pool_size == (2, 2) and winpool.window_size == (4, 4)
    ats = blocks.Block(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, ats_fraction=0.5)
    assert ats.ats_fraction == 0.5 and ats.last_ats_indices is None
    with pytest.raises(AssertionError):   # ATS excludes pooling and windows, as in the reference (blocks.py:71-73)
        blocks.Block(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, ats_fraction=0.5, pool_size=2)
    pooled = blocks.EventfulBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, pool_size=2, relative_embedding_size=(6, 6))
    assert pooled.pool_size == (2, 2) and pooled.relative_position.pool_size == (2, 2)
    bb = ViTBackbone(block_config=dict(dim=64, heads=4, mlp_ratio=4, relative_embedding_size=(8, 8), window_size=(3, 3)),
                     depth=3, position_encoding_size=(3, 3), input_size=(6, 6), block_class="EventfulBlock",
                     windowed_class="EventfulTokenwiseBlock", window_indices=(0, 2),
                     windowed_overrides=dict(matmul_2_cast=None))
    kinds = [type(b).__name__ for b in bb.blocks]
    assert kinds == ["EventfulTokenwiseBlock", "EventfulBlock", "EventfulTokenwiseBlock"]
    assert bb.blocks[0].window_size == (3, 3) and bb.blocks[1].window_size is None
    assert bb.blocks[0].relative_position.y_embedding.shape == (5, 16)   # sized by the window (blocks.py:90-91)
    assert bb.blocks[1].relative_position.y_embedding.shape == (15, 16)
    assert bb.position_encoding.encoding.shape == (1, 9, 64)


def test_set_policies_reset_and_counting_hooks():
    from eventful_transformer import blocks, modules, policies
    blk = blocks.EventfulBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4)
    H.set_policies(blk, policies.TokenNormTopK, k=5)
    gates = [m for m in blk.modules() if isinstance(m, (modules.TokenGate, modules.SimpleSTGTGate))]
    assert len(gates) == 5 and len({id(g.policy) for g in gates}) == 5  # one fresh policy per gate
    assert all(g.policy.k == 5 for g in gates)
    assert len(list(blk.extended_modules())) == 19 + 5  # 19 in the reference + 5 policies just attached
    blk.counting()
    assert all(m.count_mode for m in blk.extended_modules())
    blk.qkv.count_rows(10)
    assert blk.total_counts()["linear_flops"] == 10 * 64 * 192 and blk.total_counts()["bias_flops"] == 10 * 192
    blk.clear_counts()
    assert len(blk.total_counts()) == 0
    blk.no_counting()
    blk.qkv_gate.first, blk.qkv_gate.p = False, torch.zeros(1)
    blk.qkv_accumulator.first, blk.qkv_accumulator.b = False, torch.zeros(1)
    blk.reset()
    assert blk.qkv_gate.first and blk.qkv_gate.p is None and blk.qkv_accumulator.b is None
    stg = blocks.EventfulTokenwiseBlock(dim=64, heads=4, input_size=(6, 6), mlp_ratio=4, stgt=True)
    assert isinstance(stg.qkv_gate, modules.SimpleSTGTGate)
    tf = policies.TokenNormTopFraction(0.5)
    assert tf.capacity(37) == 18 and tf.fixed_count(37) == 18
    thr = policies.TokenNormThreshold(threshold=0.3)
    assert thr.capacity(37) == 37 and thr.fixed_count(37) is None


def test_index_helpers_match_reference_semantics():
    from eventful_transformer.utils import expand_col_index, expand_row_index
    idx = torch.tensor([[0, 2], [1, 3]])
    for shape in [(2, 5, 7), (2, 3, 5, 7)]:
        assert torch.equal(expand_row_index(idx, shape), O.rows_index(idx, shape))
        assert torch.equal(expand_col_index(idx, shape), O.cols_index(idx, shape))
        x = torch.arange(float(torch.tensor(shape).prod())).view(shape)
        assert torch.equal(x.gather(-2, expand_row_index(idx, shape)), x.gather(-2, O.rows_index(idx, shape)))


@pytest.mark.parametrize("isz,win", [((6, 6), (3, 3)), ((7, 5), (3, 3)), ((5, 9), (4, 2))])
def test_window_map_matches_oracle_partition(isz, win):
    """The int32 window map used by K4/K5/K6 == Block._partition_windows of the reference (via the oracle)."""
    from eventful_transformer.blocks import _window_map
    n = isz[0] * isz[1]
    params = O.make_block_params(8, 1, seed=0)
    params["qkv.bias"] = torch.full((24,), -7.0)
    blk = O.BlockOracle("EventfulTokenwiseBlock", params, 8, 1, isz, window_size=win)
    tokens = torch.arange(float(n)).view(1, n, 1).expand(1, n, 24).contiguous()
    parts = blk._to_windows(tokens)[..., 0]            # (windows, window_len), padding == -7
    wm = _window_map(isz, win, torch.device("cpu"))
    assert wm.shape == parts.shape
    assert torch.equal(torch.where(wm < 0, torch.tensor(-7.0), wm.float()), parts)
    back = blk._from_windows(parts.unsqueeze(-1))[0, :, 0]
    assert torch.equal(back, torch.arange(float(n)))


def test_counts_arithmetic():
    from eventful_transformer.base import Counts, dict_csv_header, dict_csv_line, numeric_tuple
    a, b = Counts(), Counts()
    a["x"] += 3
    b["x"] += 1
    b["y"] += 2
    assert dict(a + b) == {"x": 4, "y": 2} and dict(a - b) == {"x": 2, "y": -2}
    assert dict(2 * a) == {"x": 6} and dict(a / 2) == {"x": 1.5} and dict(5 - a) == {"x": 2}
    assert dict(sum([a, b])) == {"x": 4, "y": 2}
    assert dict_csv_header(a + b) == "x,y" and dict_csv_line(a + b) == "4,2"
    assert numeric_tuple(3, 2) == (3, 3) and numeric_tuple([1, 2], 2) == (1, 2)


def test_bicubic_resize_matches_aten():
    """utils.bicubic_resize (two matmuls with ATen's cubic taps) against F.interpolate(mode="bicubic"), the call the
    reference makes for rel-pos tables and position encodings (utils.py:93-97,176-183)."""
    import torch.nn.functional as F
    from eventful_transformer.utils import bicubic_resize
    g = torch.Generator().manual_seed(0)
    for shape, size in [((1, 64, 64, 64), (42, 42)), ((1, 768, 14, 14), (42, 42)), ((1, 16, 8, 8), (6, 6)),
                        ((1, 32, 14, 14), (64, 64)), ((2, 8, 7, 5), (9, 11)), ((1, 4, 4, 4), (1, 3)), ((1, 8, 9, 9), (9, 9))]:
        x = torch.randn(*shape, generator=g)
        want = F.interpolate(x, size, mode="bicubic", align_corners=False)
        got = bicubic_resize(x, size)
        assert got.shape == want.shape
        assert float((got - want).abs().max()) < 1e-5, (shape, size)


def test_graft_entry_build_hook():
    """The driver's build hook: compiles every HIP source for gfx950 and checks the ABI of the freshly built library."""
    import __graft_entry__ as entry

    entry.build()
    from eventful_transformer import _native

    assert _native.load().evt_version() == _native.ABI_VERSION


def test_big_tile_choice_respects_the_4gb_weight_plane_bound():
    """The persistent 256-row GEMM addresses the hl32 weight planes with 32-bit byte offsets: a weight matrix whose planes reach
    4 GB (Nout * pitch(K) * 2 >= 2^32) must fall back to the 128x128 kernel (evt_gated_linear_big_tile == 0), while the same
    launch at ViT-B's size picks a 256-row tile.  Shape-only query: no GPU needed."""
    from eventful_transformer import _native
    B, kcap, N = 256, 128, 197
    assert _native.gated_linear_big_tile(768, True, N, 2304, True, N, False, B, kcap, 768, 2304) != 0
    big_k, big_n = 65536, 32768          # 32768 rows x 65536 x 4 bytes of hl32 lines = 8 GB
    assert _native.gated_linear_big_tile(big_k, True, N, big_n, True, N, False, B, kcap, big_k, big_n) == 0


def test_gated_tile_layout_matches_the_abi_formula():
    """The tiled gate reference of evt_attention_gated: the host's permute-based conversion places element (row i, key j) where
    include/evt_abi.h says -- tile (i / 32, j / 32), offset 512 (c / 16) + 8 (32 (c / 4 % 2) + r) + 4 (c / 8 % 2) + c % 4 -- and the
    setter / getter of `matmul_gate.p` round-trip through it."""
    import torch

    from eventful_transformer import _native
    from eventful_transformer.modules import TokenDeltaGate

    B, H, N = 2, 3, 70
    nt = (N + 31) // 32
    p = torch.arange(B * H * N * N, dtype=torch.float32).view(B, H, N, N)
    tiles = torch.full((B, H, nt, nt, 2, 2, 32, 2, 4), -1.0)
    _native.logical_to_tiles(p, tiles)
    flat = tiles.view(B, H, nt * nt, 1024)
    for (b, h, i, j) in [(0, 0, 0, 0), (1, 2, 69, 69), (0, 1, 33, 5), (1, 0, 7, 63), (0, 2, 40, 47), (1, 1, 31, 32), (0, 0, 64, 12)]:
        r, c = i % 32, j % 32
        off = 512 * (c // 16) + 8 * (32 * ((c // 4) % 2) + r) + 4 * ((c // 8) % 2) + c % 4
        assert flat[b, h, (i // 32) * nt + j // 32, off] == p[b, h, i, j]
    assert torch.equal(_native.tiles_to_logical(tiles, N), p)
    gate = TokenDeltaGate(structure="col")
    gate.use_tiles(torch.zeros_like(tiles), N)
    gate.p = p
    assert torch.equal(gate.p, p) and gate._tiles is not None
    gate.reset()
    assert gate.p is None and gate._tiles is None
