"""GPU: block / backbone level parity of the HIP path (through the reference's Python API) against
(a) the committed golden vectors produced by the real reference and (b) the CPU oracle run here.

Tolerances (fp32 residual stream, north_star: 1e-3):
  * fp32 mode                : 2e-4 absolute on block outputs (|y| ~ 4)
  * bf16 / fp16 A.v cast mode: 1e-3 absolute on block outputs.  Inside the cast stage a different
    fp32 accumulation order can flip one bf16 rounding (1 ulp = 2^-8 relative) of an A.v element;
    the projection (weights ~ 0.08) shrinks that below 1e-3 on the residual stream.
Gate indices: compared as ascending sets; must be identical whenever the reference's recorded
margin between the k-th and (k+1)-th norm is >= 1e-3 (all small fixtures are generated that way).
"""
import os

import numpy as np
import pytest
import torch

import eventful_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tol(kw):
    return 2e-4 if kw.get("matmul_2_cast") is None else 1e-3


@pytest.mark.parametrize("name", list(H.small_cases().keys()))
def test_small_blocks_vs_golden(golden_dir, name):
    from eventful_transformer import _native
    g = H.load_npz(os.path.join(golden_dir, "blocks_small.npz"))
    case = H.small_cases()[name]
    kind, isz, has_cls, kw, pol = case
    params = H.small_case_params(name, case, g[f"{name}__param_seed"])
    blk = H.product_block(kind, params, H.SMALL["dim"], H.SMALL["heads"], isz, **kw)
    H.product_policy(blk, pol)
    xs = torch.from_numpy(g[f"{name}__x"])
    ys = torch.from_numpy(g[f"{name}__y"])
    with torch.inference_mode():
        for t in range(xs.shape[0]):
            y = blk(xs[t].to(DEV)).cpu()
            err = float((y - ys[t]).abs().max())
            assert err <= _tol(kw), (name, t, err)
            if kind != "Block" and t > 0:
                B = xs.shape[1]
                for gname in ("qkv", "projection", "mlp"):
                    want = torch.from_numpy(g[f"{name}__idx_{gname}_{t}"]).long()
                    pol_obj = getattr(blk, f"{gname}_gate").policy
                    cap = pol_obj.capacity(xs.shape[2])
                    idx = _native.scratch(f"idx_{gname}", (B, cap), torch.int32, torch.device(DEV, 0)).cpu().long()
                    if pol[0] == "thr":
                        cnt = int(_native.scratch(f"cnt_{gname}", (B,), torch.int32, torch.device(DEV, 0)).cpu()[0])
                        assert cnt == want.shape[-1], (name, t, gname, cnt, want.shape)
                        idx = idx[:, :cnt]
                    assert torch.equal(idx, want), (name, t, gname)
    assert _native.is_loaded()


def test_reset_and_state_attributes():
    """reset() re-arms the first-frame branch; state is reachable under the reference's names."""
    from eventful_transformer import policies
    params = O.make_block_params(64, 4, seed=3, std=0.08)
    blk = H.product_block("EventfulBlock", params, 64, 4, (6, 6), matmul_2_cast="bfloat16")
    H.set_policies(blk, policies.TokenNormTopK, k=12)
    xs = O.make_token_stream(2, 37, 64, 3, 12, seed=4, small=0.02)
    with torch.inference_mode():
        first = [blk(xs[t].to(DEV)).cpu() for t in range(3)]
        assert blk.qkv_gate.p.shape == (2, 37, 64) and blk.qkv_accumulator.b.shape == (2, 37, 192)
        assert blk.matmul_accumulator_1.product.shape == (2, 4, 37, 37)
        assert blk.v_gate.p.shape == (2, 4, 37, 16) and blk.v_gate.p.dtype == torch.bfloat16
        assert blk.matmul_gate.p.shape == (2, 4, 37, 37) and blk.matmul_gate.p.dtype == torch.bfloat16
        assert blk.matmul_accumulator_2.product.shape == (2, 4, 37, 16)
        # I1 on the HIP state
        q, k, _ = blk.qkv_accumulator.b.cpu().view(2, 37, 3, 4, 16).permute(2, 0, 3, 1, 4)
        assert torch.allclose(blk.matmul_accumulator_1.product.cpu(), (q / 4.0) @ k.transpose(-2, -1), atol=1e-5)
        blk.reset()
        assert blk.qkv_gate.first and blk.qkv_gate.p is None and blk.matmul_accumulator_2.product is None
        again = [blk(xs[t].to(DEV)).cpu() for t in range(3)]
    for a, b in zip(first, again):
        assert torch.equal(a, b)  # deterministic (I7)


@pytest.mark.parametrize("cast,grid,has_cls,policy", [("bfloat16", (6, 6), True, ("thr", 1.0)), ("float16", (9, 7), False, ("topk", 20)),
                                                      ("bfloat16", (16, 16), False, ("topk", 100)), ("float16", (5, 5), True, ("thr", 1e9))])
def test_eventful_block_on_the_resident_attention_kernel(cast, grid, has_cls, policy):
    """EventfulBlock at head dim 64 with a 16-bit cast and no relative position runs its attention on evt_attention_gated (K10: value
    gate + scores + softmax + A gate + both accumulator products in one launch, `matmul_gate.p` tiled): block outputs against the oracle
    over 5 frames -- the threshold policy (variable, device-side count; batch 1 as in the reference, also a threshold nothing passes),
    top-k with and without a class token, N = 256 -- and the per-clip state through the reference's attribute names: `matmul_gate.p`
    (read through the tiled layout), `v_gate.p`, `matmul_accumulator_2.product`, the lazily refreshed `matmul_accumulator_1.product`."""
    dim, heads = 128, 2
    N = grid[0] * grid[1] + int(has_cls)
    B = 1 if policy[0] == "thr" else 3
    params = O.make_block_params(dim, 4, seed=51, std=0.06, head_dim=64)
    blk = H.product_block("EventfulBlock", params, dim, heads, grid, matmul_2_cast=cast)
    ob = O.BlockOracle("EventfulBlock", params, dim, heads, grid, matmul_2_cast=cast)
    ob.set_policy(H.oracle_policy(policy))
    forced = policy[0] == "topk"   # top-k: decisions teacher-forced gate by gate (a near-tie at a downstream gate would fork the two runs);
    if forced:                     # threshold: free-running -- the variable, device-side count is the point (margins are wide on this stream)
        for gn in ("qkv_gate", "projection_gate", "mlp_gate", "v_gate", "matmul_gate"):
            getattr(blk, gn).policy = _ForcedPolicy(policy[1])
    else:
        H.product_policy(blk, policy)
    xs = O.make_token_stream(B, N, dim, 5, policy[1] if policy[0] == "topk" else 12, seed=52, small=0.01)   # (wide margins at the qkv gate)
    sdt = getattr(torch, cast)
    with torch.inference_mode():
        for t in range(5):
            y_ref = ob.forward(xs[t].clone())
            if forced and t > 0:
                for gn, tk in (("qkv_gate", "qkv_index"), ("projection_gate", "projection_index"), ("mlp_gate", "mlp_index")):
                    getattr(blk, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].to(DEV)
            y = blk(xs[t].to(DEV)).cpu()
            err = float((y - y_ref).abs().max())
            assert err <= 2e-3, (cast, t, err)
            assert blk.matmul_gate._tiles is not None   # the resident kernel took the block
        p = blk.matmul_gate.p
        assert p.shape == (B, heads, N, N) and p.dtype == sdt
        atol_p = 4e-3 if cast == "bfloat16" else 5e-4
        assert torch.allclose(p.float().cpu(), ob.s["matmul_gate"].t.float(), atol=atol_p)
        assert blk.v_gate.p.shape == (B, heads, N, 64)
        assert torch.allclose(blk.v_gate.p.float().cpu(), ob.s["v_gate"].t.float(), atol=2 * atol_p * float(ob.s["v_gate"].t.float().abs().max()))
        assert blk.matmul_accumulator_2.product.shape == (B, heads, N, 64)
        q, k, _ = blk.qkv_accumulator.b.cpu().view(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
        assert torch.allclose(blk.matmul_accumulator_1.product.cpu(), (q / 8.0) @ k.transpose(-2, -1), atol=2e-4)
        # assigning the logical tensor writes the tiles; the next frame then runs from that state
        blk.matmul_gate.p = ob.s["matmul_gate"].t.to(DEV)
        assert torch.equal(blk.matmul_gate.p.cpu(), ob.s["matmul_gate"].t)
    blk.reset()
    assert blk.matmul_gate.p is None and blk.matmul_gate._tiles is None


@pytest.mark.parametrize("order", [1, float("inf")])
@pytest.mark.parametrize("kind,cast", [("EventfulBlock", "bfloat16"), ("EventfulTokenwiseBlock", None)])
def test_blocks_with_l1_and_linf_norm_policies(kind, cast, order):
    """`TokenNormTopK(order=...)` (policies.py:44,63): the gates select on the L1 / L-infinity norm of the delta.  The fused projection-gate
    norm (per-head partial squares out of the attention epilogue) only makes an L2 norm, so these orders take the row pass; outputs and
    index sets against the CPU oracle with the same `order`."""
    from eventful_transformer import policies
    params = O.make_block_params(64, 4, seed=13, std=0.08)
    kw = dict(matmul_2_cast=cast) if cast else {}
    blk = H.product_block(kind, params, 64, 4, (6, 6), **kw)
    H.set_policies(blk, policies.TokenNormTopK, k=12, order=order)
    ob = O.BlockOracle(kind, params, 64, 4, (6, 6), **kw)
    ob.set_policy(lambda: O.TopK(12, order=order))
    xs = O.make_token_stream(2, 36, 64, 4, 12, seed=14, small=0.02)
    got = []
    hooks = _grab_index_sets(type("BB", (), {"blocks": [blk]})(), 12, got)
    with torch.inference_mode():
        for t in range(4):
            y = blk(xs[t].to(DEV)).cpu()
            y_ref = ob.forward(xs[t].clone())
            assert float((y - y_ref).abs().max()) <= (1e-3 if cast else 2e-4), (t, float((y - y_ref).abs().max()))
            if t > 0:
                for gi, key in enumerate(("qkv_index", "projection_index", "mlp_index")):
                    want = ob.trace[key].sort(dim=-1)[0][0].numpy().astype(np.int64)
                    assert np.array_equal(got[t][gi], want), (t, key)
    for h in hooks:
        h.remove()


def test_threshold_zero_selected_freezes_buffers():
    """I6: r = 0 everywhere -> all buffers frozen, output moves only through the residual."""
    from eventful_transformer import policies
    params = O.make_block_params(64, 4, seed=6, std=0.08)
    blk = H.product_block("EventfulBlock", params, 64, 4, (6, 6))
    H.set_policies(blk, policies.TokenNormThreshold, threshold=1e9)
    xs = O.make_token_stream(1, 36, 64, 3, 9, seed=2, small=0.02)
    with torch.inference_mode():
        y0 = blk(xs[0].to(DEV)).cpu()
        state = {n: getattr(blk, n).b.clone() for n in ("qkv_accumulator", "projection_accumulator", "mlp_accumulator")}
        y1 = blk(xs[1].to(DEV)).cpu()
        for n, s in state.items():
            assert torch.equal(getattr(blk, n).b, s)
    assert torch.allclose(y1 - y0, xs[1] - xs[0], atol=1e-5)


def test_counts_match_closed_form():
    """I5-style accounting: MAC counters of the fused path equal the reference's formulas."""
    from eventful_transformer import policies
    B, N, D, H_, k = 2, 37, 64, 4, 12
    params = O.make_block_params(D, 4, seed=3, std=0.08)
    blk = H.product_block("EventfulBlock", params, D, H_, (6, 6))
    H.set_policies(blk, policies.TokenNormTopK, k=k)
    xs = O.make_token_stream(B, N, D, 2, k, seed=4, small=0.02)
    blk.counting()
    with torch.inference_mode():
        blk(xs[0].to(DEV))
        c0 = blk.total_counts()
        blk.clear_counts()
        blk(xs[1].to(DEV))
        c1 = blk.total_counts()
    assert c0["linear_flops"] == B * N * 12 * D * D and c0["matmul_flops"] == 2 * B * N * N * D
    assert c1["linear_flops"] == B * k * 12 * D * D           # 12 k D^2 per block per frame
    assert c1["matmul_flops"] == 4 * B * k * N * D            # 4 k N D
    assert c1["gate_flops"] == 3 * B * N * D + B * N * D + B * H_ * N * N
    assert c1["accumulator_flops"] == B * k * D + 2 * B * N * D
    assert c1["add_flops"] == 2 * B * N * D
    assert c1["bias_flops"] == B * k * (3 * D + D + 4 * D + D)


def _grab_index_sets(bb, k, sink):
    """Forward hooks that copy clip 0's three gate index lists out of the shared scratch after every block."""
    from eventful_transformer import _native
    dev0 = torch.device(DEV, 0)

    def grab(_m, inp, _o):
        B = inp[0].shape[0]
        sink.append([_native.scratch(f"idx_{g}", (B, k), torch.int32, dev0)[0].cpu().numpy().astype(np.int64)
                     for g in ("qkv", "projection", "mlp")])
    return [blk.register_forward_hook(grab) for blk in bb.blocks]


@pytest.mark.parametrize("fixture,k,mode,cast,tol", [("vivit_b.npz", 128, "fp32", None, 1e-3), ("vivit_b.npz", 128, "bf16", "bfloat16", None),
                                                     ("vivit_b_k64.npz", 64, "fp32", None, 1e-3), ("vivit_b_k64.npz", 64, "bf16", "bfloat16", None),
                                                     ("vivit_b_sharp.npz", 128, "fp32", None, 1e-3)])
def test_vivit_b_full_size(golden_dir, fixture, k, mode, cast, tol):
    """ViViT-B spatial model, 197 tokens, 12 EventfulBlocks, B=1, FREE-RUNNING against the reference's golden
    class-token features and gate index sets: BASELINE config 2 (k=128, 6 frames) and config 4's shape (k=64,
    T=32 frames).  Reports the index-set agreement rate over all 36 gates per frame and requires identical sets
    wherever the reference's margin between the k-th and (k+1)-th norm is >= 1e-3."""
    from eventful_transformer import policies
    g = H.load_npz(os.path.join(golden_dir, fixture))
    seed = int(g[f"{mode}__seed"])
    qk_std = float(g["qk_std"]) if "qk_std" in g.files else None
    _, sd, cls, ln_w, ln_b = H.vivit_oracle(cast, seed=seed, k=k, qk_std=qk_std)
    bb = H.product_vivit(sd, cast)
    H.set_policies(bb, policies.TokenNormTopK, k=k)
    feats = torch.from_numpy(g[f"{mode}__features"])
    idx_gold = g[f"{mode}__idx"]
    margins = g[f"{mode}__margins"]
    xs = O.make_token_stream(1, 196, 768, feats.shape[0], k, seed=seed + 2, small=0.01)
    cls_d, w_d, b_d = cls.to(DEV), ln_w.to(DEV), ln_b.to(DEV)
    got = []
    hooks = _grab_index_sets(bb, k, got)
    agree = total = strict = strict_ok = 0
    proj = proj_ok = 0            # projection gates at margin >= 1e-3 (the sharp fixture has them)
    smallest_equal = 1.0          # the smallest reference margin among the differing sets (1.0: none differ)
    worst = 0.0
    with torch.inference_mode():
        for t in range(feats.shape[0]):
            x = torch.concat([cls_d.expand(1, 1, 768), xs[t].to(DEV)], dim=1)
            y = bb(x)
            f = torch.nn.functional.layer_norm(y, (768,), w_d, b_d, 1e-6)[:, 0].cpu()
            worst = max(worst, float((f - feats[t]).abs().max()))
            if t == 0:
                continue
            for bi in range(12):
                for gi in range(3):
                    same = np.array_equal(got[t * 12 + bi][gi], idx_gold[t - 1, bi, gi, 0].astype(np.int64))
                    total += 1
                    agree += same
                    if not same:
                        smallest_equal = min(smallest_equal, float(margins[t - 1, bi, gi]))
                    if margins[t - 1, bi, gi] >= 1e-3:
                        strict += 1
                        strict_ok += same
                        if gi == 1:
                            proj += 1
                            proj_ok += same
    for h in hooks:
        h.remove()
    H.report(f"\n[free-running {fixture} {mode}] index-set agreement {agree}/{total} = {agree / total:.4f}; "
             f"margin>=1e-3: {strict_ok}/{strict} (projection gates among them: {proj_ok}/{proj}); max feature error {worst:.3e}; "
             f"reference margin of the first differing set: {smallest_equal:.2e}")
    # The bar is the reference's OWN free-running divergence when only the fp32 summation order changes (1, 2, 4 vs 8 ATen
    # threads; oracle/gen_golden.py::gen_envelope -> tests/golden/envelope.npz) -- a different summation order is exactly what
    # a GPU kernel is.  fp32: the reference agrees with itself on 100 % of the index sets to 3e-6; so must the HIP path (to
    # 1e-3 on features, north_star).  bf16 A.v cast: every flipped bf16 rounding of an A.v state element persists and
    # gates with margins down to 1e-9 fork; the reference then agrees with itself on 74-82 % of all sets (k = 128: 100 %
    # of the sets with margin >= 1e-3; k = 64, 31 gated frames: 86-87 %) with feature gaps of 6.7e-2 / 8.2e-2.  The HIP
    # path must stay within a measured, arithmetic-mode-dependent distance of the reference's lowest self-agreement (below)
    # and within 1.5 x its largest feature gap; the strict bf16 check is the teacher-forced test below.
    env = H.load_npz(os.path.join(golden_dir, "envelope.npz"))
    tag = "k128" if k == 128 else "k64"
    ref_all = float(env[f"{tag}__{mode}__agreement_all"].min())
    ref_strict = float(env[f"{tag}__{mode}__agreement_margin_1e-3"].min())
    ref_gap = float(env[f"{tag}__{mode}__feature_gap"].max())
    H.report(f"    reference vs itself (1/2/4 vs 8 threads): agreement >= {ref_all:.4f}, margin>=1e-3 >= {ref_strict:.4f}, "
             f"feature gap <= {ref_gap:.3e}")
    if cast is None:
        # fp32 mode: EVERY index set must equal the reference's (as the reference's own re-runs do), whatever its margin
        assert ref_all == 1.0 and strict_ok == strict and strict >= total // 2, (strict_ok, strict, total)
        assert agree == total, (agree, total, smallest_equal)
        assert worst <= tol, (mode, worst)
        if qk_std is not None:
            assert proj >= 60, proj
    else:
        # How far below the reference's own self-agreement?  Measured (profiles/r05/envelope_split_vs_f32.txt): with EXACT-fp32 GEMMs
        # (EVT_GEMM=f32: the fp32-input MFMA, bitwise an fmaf chain) the HIP path lands inside / at the edge of the envelope
        # (k = 128: 0.8167 vs >= 0.7833; k = 64: 0.7294 vs >= 0.7446, 0.8526 vs >= 0.8581 on the margin subset); with the
        # split-precision GEMMs (1e-5 instead of 1e-6 relative per product) near-tied gates fork a few frames earlier: 0.7167,
        # 0.7007 / 0.8426.  The slack is therefore tied to the arithmetic mode: 0.03 / 0.02 for exact fp32 (test_envelope_exact_fp32_gemm
        # runs this test in that mode), 0.08 / 0.03 for the split mode.
        from eventful_transformer import _native
        slack_all, slack_strict = (0.03, 0.02) if _native.GEMM_MODE == "f32" else (0.08, 0.03)
        assert agree / total >= ref_all - slack_all, (agree, total, ref_all)
        assert strict_ok / strict >= ref_strict - slack_strict, (strict_ok, strict, ref_strict)
        assert worst <= 1.5 * ref_gap, (mode, worst, ref_gap)


def test_envelope_exact_fp32_gemm():
    """The bf16-mode free-running tests once more with exact-fp32 GEMMs (the library reads EVT_GEMM once per process): the HIP path
    must then sit within 0.03 / 0.02 of the reference's own self-agreement -- the evidence that the rest of the distance seen in the
    default mode is the split arithmetic and not a defect of the gated path."""
    import subprocess
    import sys
    env = dict(os.environ, EVT_GEMM="f32")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "vivit_b_full_size and bf16",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_vivit_b_dense_config1():
    """BASELINE config 1 on the GPU: ViViT-B spatial model with 12 dense `Block`s (gating off), N = 197, two clips x
    two frames, class-token features against the CPU oracle (1e-3)."""
    seed = 77
    sd = H.backbone_params(12, 768, 4, seed, 197)
    rs = np.random.RandomState(seed + 1)
    cls = torch.from_numpy((rs.standard_normal((1, 1, 768)) * 0.02).astype(np.float32))
    ln_w = torch.from_numpy((1 + rs.standard_normal(768) * 0.05).astype(np.float32))
    ln_b = torch.from_numpy((rs.standard_normal(768) * 0.05).astype(np.float32))
    blocks = [O.BlockOracle("Block", H.block_params_of(sd, i), 768, 12, (14, 14)) for i in range(12)]
    ora = O.ViViTSpatialOracle(O.BackboneOracle(blocks, sd["position_encoding.encoding"], (14, 14), (14, 14), True), cls, ln_w, ln_b)
    from eventful_transformer.backbones import ViTBackbone
    bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4), depth=12, position_encoding_size=(14, 14),
                     input_size=(14, 14), block_class="Block", has_class_token=True)
    bb.load_state_dict(sd, strict=True)
    bb = bb.eval().to(DEV)
    xs = O.make_token_stream(2, 196, 768, 2, 128, seed=seed + 2, small=0.01)
    with torch.inference_mode():
        for t in range(2):
            x = torch.concat([cls.to(DEV).expand(2, 1, 768), xs[t].to(DEV)], dim=1)
            f = torch.nn.functional.layer_norm(bb(x), (768,), ln_w.to(DEV), ln_b.to(DEV), 1e-6)[:, 0].cpu()
            ref = ora.forward(xs[t])
            assert float((f - ref).abs().max()) <= 1e-3, (t, float((f - ref).abs().max()))


class _ForcedPolicy:
    """Test double for teacher-forcing DECISIONS: records what the real HIP top-k policy selects on the
    gate's delta, but hands the block the oracle's index set, so both sides refresh the same tokens."""

    def __init__(self, k):
        from eventful_transformer import policies
        self.real = policies.TokenNormTopK(k)
        self.force = None
        self.mine = None

    def __call__(self, e, dim=-1):
        self.mine = self.real(e, dim=dim)
        return self.force


@pytest.mark.parametrize("fixture,k,mode,cast,out_tol,min_margin", [("vivit_b.npz", 128, "fp32", None, 5e-4, 1e-4),
                                                                    ("vivit_b.npz", 128, "bf16", "bfloat16", 1e-3, 1e-3),
                                                                    ("vivit_b_k64.npz", 64, "bf16", "bfloat16", 1e-3, 2e-3),
                                                                    ("vivit_b_sharp.npz", 128, "fp32", None, 5e-4, 1e-3)])
def test_vivit_b_teacher_forced(golden_dir, fixture, k, mode, cast, out_tol, min_margin):
    """Full-size ViViT-B, teacher-forced block by block AND gate by gate:
      * every block is fed the ORACLE's input for that block;
      * every gate is handed the oracle's index set (so near-tie decisions cannot fork the states), while
        the set the HIP policy would have chosen is recorded;
      * each block's fp32 output must match the oracle's (5e-4 fp32 mode, 1e-3 with the bf16 A.v cast);
      * the recorded HIP selections must equal the reference's golden sets wherever the reference's margin
        between the k-th and (k+1)-th norm is >= min_margin (1e-4 in fp32 mode; with the bf16 A.v cast the
        projection gate's input carries the bf16 rounding of the A.v accumulators, ~2^-9 per element, and the
        one set that differed in these runs had a reference margin of 1.06e-3: bar 1e-3 at k = 128, 2e-3 at k = 64).
    `vivit_b_sharp.npz` (sharp attention, O.sharpen_qk; all 12 frames; fp32 mode, where north_star's bit-exact bar holds): the
    projection gate's norms are spread out there, so its sets are compared too -- at least 60 projection-gate sets with a
    reference margin >= 1e-3 must be bit-equal.  (The bf16 A.v cast with sharp attention: test_vivit_b_sharp_bf16_projection_gates.)"""
    g = H.load_npz(os.path.join(golden_dir, fixture))
    seed = int(g[f"{mode}__seed"])
    sharp = "qk_std" in g.files
    model, sd, cls, ln_w, ln_b = H.vivit_oracle(cast, seed=seed, k=k, qk_std=float(g["qk_std"]) if sharp else None)
    bb = H.product_vivit(sd, cast)
    steps = g[f"{mode}__features"].shape[0] if sharp else 4
    checked, mismatched, per_gate, worst = _teacher_forced_clip(model, bb, cls, g, mode, k, steps, seed + 2, out_tol, min_margin, fixture)
    H.report(f"\n[teacher-forced {fixture} {mode}] {steps} frames: worst block-output error {worst:.3e} (bar {out_tol:.0e}); index sets at "
             f"reference margin >= {min_margin:.0e}: {checked - mismatched}/{checked} equal; per gate (checked, equal): {per_gate}")
    assert checked >= 60 and mismatched == 0, (checked, mismatched)
    if sharp:
        assert per_gate["projection_gate"][0] >= 60, per_gate


def test_vivit_b_sharp_bf16_projection_gates(golden_dir):
    """The headline's own arithmetic mode (bf16 A.v cast) where it is most fragile: PROJECTION-gate index sets under sharp attention,
    index-only (no output bar: a sharp attention output of magnitude ~1 carries a 2^-9 rounding step of its own).
    `vivit_b_sharp_clips.npz` = the REAL reference on 192 short clips (3 frames: 2 gated frames each; short because with the cast the
    bf16 state of any two implementations also drifts apart frame by frame), all gates' sets and margins, teacher-forced here block by
    block and gate by gate; the clips that hold a large-margin projection gate are replayed.
    What can be asked for -- measured, not assumed (profiles/r05/bf16_sharp_projection_gates_ab.txt): the projection gate's input is
    the bf16 A.v state, so a token's frame-to-frame delta is a handful of bf16 steps in a handful of elements; ONE probability a~
    rounded the other way (any 1e-7-level difference in exp / the softmax sum does that to a few of the 465k a~ per block) moves up to
    a~ x 64 elements of one token's output by a step and with it that token's delta norm by ~1 %.  The HIP sets equal the reference's
    on 79 % of the projection gates with a reference margin >= 1e-3, and the rate does NOT rise with the margin (83 % at >= 3e-3,
    86 % at >= 7e-3, 10 of 13 at >= 1e-2) nor with exact-fp32 q.k^T or exact-fp32 GEMMs (635 vs 634 of 807): it is the quantised
    state, not the split arithmetic.  The reference arithmetic itself, re-run at another ATen thread count, does not re-order these
    sums (`selfdiff` all false) and offers no noise floor of its own.  Required: EVERY qkv / mlp gate at margin >= 1e-3 bit-equal
    (2592 of them), projection gates: at least 70 % of those at margin >= 1e-3 (>= 60 checked), the table by margin bar reported.
    The strict projection-gate claim is the fp32-mode test (test_vivit_b_teacher_forced[sharp]: 68 / 68).
    ROUND 6: what the 79 % is made of was then measured (test_vivit_b_sharp_bf16_projection_gates_state_forced, below): with every
    block-frame started from the oracle's state the HIP projection gates equal the oracle's on 554 / 554 sets at margin >= 1e-4 -- and the
    oracle run on the GPU box's CPU reproduces these GOLDEN sets (the same arithmetic, run on the build machine, free-running) on only
    381 / 548.  The rate below is the drift of the bf16 states between two machines' fp32 summation orders, not the kernels' arithmetic."""
    g = H.load_npz(os.path.join(golden_dir, "vivit_b_sharp_clips.npz"))
    k, cast, steps = int(g["k"]), "bfloat16", int(g["steps"])
    margins_all, idx_all = g["margins"], g["idx"]
    model, sd, cls, ln_w, ln_b = H.vivit_oracle(cast, seed=int(g["seed"]), k=k, qk_std=float(g["qk_std"]))
    bb = H.product_vivit(sd, cast)
    bars = (1e-3, 3e-3, 5e-3, 7e-3, 1e-2)
    proj = {b_: [0, 0] for b_ in bars}         # bar -> [checked, equal]
    other = [0, 0]
    worst_all, worst_differing = 0.0, 0.0
    # the projection gates with a large margin are rare (~3 %): replay the clips that hold one, until 70 of them have been seen
    want = [c for c in range(int(g["clips"])) if (margins_all[c, :, :, 1] >= 7e-3).any()][:32]   # (all 70: the table in DESIGN.md section 3; 32: 474 checked / 372 equal = 78 %; 16 clips: 237 / 172 = 73 %, too close to the bar)
    seen = 0
    for c in want:
        model.backbone.reset()
        bb.reset()
        view = {"x__margins": margins_all[c], "x__idx": idx_all[c][:, :, :, None, :]}
        same = _teacher_forced_clip(model, bb, cls, view, "x", k, steps, int(g["stream_seeds"][c]), None, 1e-3, f"vivit_b_sharp_clips.npz clip {c}",
                                    per_set=True)
        worst_all = max(worst_all, same["worst"])
        for (t, bi, gi), eq in same["sets"].items():
            m = float(margins_all[c, t - 1, bi, gi])
            if gi == 1:
                for b_ in bars:
                    if m >= b_:
                        proj[b_][0] += 1
                        proj[b_][1] += eq
                if not eq:
                    worst_differing = max(worst_differing, m)
            elif m >= 1e-3:
                other[0] += 1
                other[1] += eq
        seen = proj[7e-3][0]
        if seen >= 70:
            break
    H.report(f"\n[teacher-forced vivit_b_sharp_clips.npz, bf16 cast + sharp attention, {len(want)} clips x 2 gated frames] projection-gate sets equal to "
             f"the reference's by margin bar (checked, equal): {proj}; largest reference margin of a differing projection set {worst_differing:.2e}; "
             f"qkv + mlp gates at margin >= 1e-3: {other[1]}/{other[0]}; worst block-output error {worst_all:.3e} (not bounded)")
    assert other[0] >= 500 and other[1] == other[0], other
    assert proj[1e-3][0] >= 60 and proj[1e-3][1] >= 0.70 * proj[1e-3][0], proj


def test_vivit_b_sharp_bf16_projection_gates_state_forced(golden_dir):
    """Separates 'the states drifted' from 'the kernel rounds differently' for the bf16-cast projection gate (round-5 review item 2a).
    Same golden as test_vivit_b_sharp_bf16_projection_gates (the REAL reference, bf16 cast, sharp attention, short clips), same
    block-by-block / gate-by-gate protocol -- but before EVERY gated block-frame the product block's whole per-clip state is overwritten
    with the reference's: the fp32 gate references and token buffers AND the bf16 states `matmul_gate.p`, `v_gate.p`,
    `matmul_accumulator_2.product`.  The block then runs ONE frame from the reference's state on the reference's input with the
    reference's decisions, and only the set its projection gate WOULD select is compared.  What is left can only come from the
    arithmetic of that one block-frame.  Beside it, under the identical protocol, an ORACLE TWIN with the reference's rounding points
    and another summation order (_ExactSumTwin: fp64 accumulation, one rounding per op result): the rate at which THAT disagrees with the
    reference is what the order of fp32 additions alone does to these near-tied, bf16-quantised deltas.  Required: qkv / mlp gates at
    margin >= 1e-3 all equal; EVERY projection-gate set at an in-situ margin >= 1e-4 equal to the oracle's (the north_star's "bit-exact
    gate indices" in the headline's own arithmetic mode); at >= 1e-5 no more than two sets behind the twin.  Measured beside it: how often
    the oracle run in situ (this machine's CPU) reproduces the GOLDEN sets of the same arithmetic run on the build machine."""
    g = H.load_npz(os.path.join(golden_dir, "vivit_b_sharp_clips.npz"))
    k, cast, steps = int(g["k"]), "bfloat16", int(g["steps"])
    margins_all, idx_all = g["margins"], g["idx"]
    model, sd, cls, ln_w, ln_b = H.vivit_oracle(cast, seed=int(g["seed"]), k=k, qk_std=float(g["qk_std"]))
    bb = H.product_vivit(sd, cast)
    twin = [_ExactSumTwin("EventfulBlock", H.block_params_of(sd, i), 768, 12, (14, 14), matmul_2_cast=cast) for i in range(12)]
    bars = (1e-5, 1e-4, 1e-3, 3e-3, 7e-3)
    hip = {b_: [0, 0] for b_ in bars}      # vs the oracle run in situ, by ITS margin
    tw = {b_: [0, 0] for b_ in bars}
    gold = {b_: [0, 0] for b_ in bars}     # the in-situ oracle vs the golden reference run (another machine), by the golden margin
    hip_gold = {b_: [0, 0] for b_ in bars}
    other = [0, 0]
    want = [c for c in range(int(g["clips"])) if (margins_all[c, :, :, 1] >= 7e-3).any()][:10]   # (24 clips: 554 / 554; 14: 320 / 320; 10 keep the test near a minute)
    for c in want:
        model.backbone.reset()
        bb.reset()
        view = {"x__margins": margins_all[c], "x__idx": idx_all[c][:, :, :, None, :]}
        same = _teacher_forced_clip(model, bb, cls, view, "x", k, steps, int(g["stream_seeds"][c]), None, 1e-3, f"vivit_b_sharp_clips.npz clip {c}",
                                    per_set=True, force_state=True, twin=twin)
        for (t, bi, gi), (eq, teq, m_here, gold_eq) in same["local"].items():
            m_gold = float(margins_all[c, t - 1, bi, gi])
            if gi == 1:
                for b_ in bars:
                    if m_here >= b_:
                        hip[b_][0] += 1
                        hip[b_][1] += eq
                        tw[b_][0] += 1
                        tw[b_][1] += teq
                    if m_gold >= b_:
                        gold[b_][0] += 1
                        gold[b_][1] += gold_eq
                        hip_gold[b_][0] += 1
                        hip_gold[b_][1] += same["sets"][(t, bi, gi)]
            elif m_here >= 1e-3:
                other[0] += 1
                other[1] += eq
    H.report(f"\n[STATE-FORCED vivit_b_sharp_clips.npz, bf16 cast + sharp attention, {len(want)} clips x 2 gated frames: every block-frame starts from the "
             f"oracle's state] PROJECTION-gate sets by margin bar (checked, equal) -- HIP vs the oracle run in situ: {hip}; oracle twin with fp64 "
             f"accumulation (same rounding points, other summation order) vs the oracle in situ: {tw}; the oracle in situ vs the golden reference run "
             f"(same arithmetic on two machines, free-running): {gold}; HIP vs the golden run: {hip_gold}; qkv + mlp gates at margin >= 1e-3 (HIP vs "
             f"in situ): {other[1]}/{other[0]}")
    assert other[0] >= 200 and other[1] == other[0], other
    # observed: 554 / 554 at margin >= 1e-4 (573 / 574 at >= 1e-5) for the HIP path AND for the twin; the in-situ oracle agrees with the
    # golden run (another machine's BLAS summation order, free-running) on 381 / 548 -- the "79 %" of the test above is that drift
    assert hip[1e-4][0] >= 200 and hip[1e-4][1] == hip[1e-4][0], hip
    assert hip[1e-5][1] >= tw[1e-5][1] - 2, (hip, tw)


_STATE_SLOTS = ("qkv_gate", "qkv_accumulator", "projection_gate", "projection_accumulator", "mlp_gate", "mlp_accumulator", "v_gate",
                "matmul_gate", "matmul_accumulator_2")


def _upload_state(pb, snap):
    """Overwrites EVERY piece of per-clip state of a product EventfulBlock with the oracle's (`snap`: slot name -> tensor taken
    before the oracle ran the frame): gate references, token buffers, and the three store-type attention states -- `v_gate.p`,
    `matmul_gate.p` (through its setter: the tiled layout of evt_attention_gated, or the plain tensor) and
    `matmul_accumulator_2.product` (head-merged (B,N,D) storage behind the (B,H,N,dh) view)."""
    for name in ("qkv_gate", "projection_gate", "mlp_gate"):
        getattr(pb, name).p.copy_(snap[name].to(DEV))
    for name in ("qkv_accumulator", "projection_accumulator", "mlp_accumulator"):
        getattr(pb, name).b.copy_(snap[name].to(DEV))
    B, Hh, N, dh = snap["v_gate"].shape
    pb.v_gate._state.copy_(snap["v_gate"].permute(0, 2, 1, 3).reshape(B, N, Hh * dh).to(DEV))
    pb.matmul_accumulator_2._state.copy_(snap["matmul_accumulator_2"].permute(0, 2, 1, 3).reshape(B, N, Hh * dh).to(DEV))
    if pb.matmul_gate._tiles is not None:
        pb.matmul_gate.p = snap["matmul_gate"].to(DEV)
    else:
        pb.matmul_gate.p.copy_(snap["matmul_gate"].to(DEV))


class _TwinPolicy:
    """Oracle-side counterpart of _ForcedPolicy: records the set TopK would select on the delta, returns the forced one."""

    def __init__(self, k):
        self.real = O.TopK(k)
        self.force = None
        self.mine = None

    def __call__(self, e, dim=-1):
        self.mine = self.real(e, dim=dim).sort(dim=-1)[0]
        return self.force


class _ExactSumTwin(O.BlockOracle):
    """The oracle with the SAME rounding points but another summation order: every linear layer and both attention-value products
    accumulate in fp64 and round once to the dtype the reference's op returns (fp32 / the `matmul_2_cast` type).  Whatever index
    sets this twin selects differently from the reference -- with all state and all decisions forced to the reference's -- differ
    because of the order of fp32 additions alone."""

    def _lin(self, x, which):
        return torch.nn.functional.linear(x.double(), self.p[which + ".weight"].double(), self.p[which + ".bias"].double()).float()

    def forward(self, x):
        saved = O.av_accumulator

        def exact(slot, a_new, v_new, a_delta, v_delta):
            dt = a_new.dtype
            if slot.t is None:
                slot.t = (a_new.double() @ v_new.double()).to(dt)
                return slot.t
            slot.t += (a_new.double() @ v_delta.double()).to(dt)
            slot.t += (a_delta.double() @ (v_new - v_delta).double()).to(dt)
            return slot.t
        O.av_accumulator = exact
        try:
            return super().forward(x)
        finally:
            O.av_accumulator = saved


def _teacher_forced_clip(model, bb, cls, g, mode, k, steps, stream_seed, out_tol, min_margin, fixture, per_set=False, force_state=False,
                         twin=None):
    """One clip, teacher-forced block by block and gate by gate (see test_vivit_b_teacher_forced) -> (sets checked, sets that differ,
    per gate [checked, equal], worst block-output error); per_set: {"sets": {(frame, block, gate): equal}, "worst": ...} instead, silently.
    force_state: before every gated block-frame the product block's WHOLE per-clip state is overwritten with the oracle's (_upload_state).
    twin: a list of _ExactSumTwin blocks run under the same protocol (state and decisions forced); "twin_sets" in the result."""
    sets = {}
    twin_sets = {}
    local = {}   # (frame, block, gate) -> (HIP set == the IN-SITU oracle's own selection, twin set == it, its margin, oracle's selection == golden)
    gate_names = ("qkv_gate", "projection_gate", "mlp_gate")
    trace_keys = ("qkv_index", "projection_index", "mlp_index")
    for blk in bb.blocks:
        for gn in gate_names + ("v_gate", "matmul_gate"):
            getattr(blk, gn).policy = _ForcedPolicy(k)
    xs = O.make_token_stream(1, model.backbone.encoding.shape[1] - 1, 768, steps, k, seed=stream_seed, small=0.01)
    margins = g[f"{mode}__margins"]
    idx_gold = g[f"{mode}__idx"]
    checked = mismatched = 0
    per_gate = {gn: [0, 0] for gn in gate_names}   # [checked, equal]
    worst = 0.0
    with torch.inference_mode():
        for t in range(steps):
            x = torch.concat([cls.expand(1, 1, 768), xs[t]], dim=1) + model.backbone.encoding
            for bi, (ob, pb) in enumerate(zip(model.backbone.blocks, bb.blocks)):
                snap = None
                if (force_state or twin is not None) and t > 0:
                    snap = {n_: ob.s[n_].t.clone() for n_ in _STATE_SLOTS}
                y_ref = ob.forward(x)
                if t > 0:
                    for gn, tk in zip(gate_names, trace_keys):
                        getattr(pb, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].to(DEV)
                    if force_state:
                        _upload_state(pb, snap)
                if twin is not None:
                    tw = twin[bi]
                    if t == 0:
                        tw.reset()
                        tw.policy = {gn: _TwinPolicy(k) for gn in tw.GATES}
                    else:
                        for n_ in _STATE_SLOTS:
                            tw.s[n_].t = snap[n_].clone()
                        tw.s["matmul_accumulator_1"].t = ob.s["matmul_accumulator_1"].t.clone()   # (exact either way: recomputed rows / columns)
                        for gn, tk in zip(gate_names, trace_keys):
                            tw.policy[gn].force = ob.trace[tk]
                    tw.forward(x)
                    if t > 0:
                        for gi, gn in enumerate(gate_names):
                            twin_sets[(t, bi, gi)] = bool(np.array_equal(tw.policy[gn].mine.reshape(-1).numpy(), idx_gold[t - 1, bi, gi].astype(np.int64).reshape(-1)))
                y_dev = pb(x.to(DEV)).cpu()
                err = float((y_dev - y_ref).abs().max())
                worst = max(worst, err)
                assert out_tol is None or err <= out_tol, (mode, t, bi, err)
                if t > 0:
                    for gi, gn in enumerate(gate_names):
                        mine = getattr(pb, gn).policy.mine.cpu().numpy()
                        same = np.array_equal(mine, idx_gold[t - 1, bi, gi].astype(np.int64))
                        sets[(t, bi, gi)] = bool(same)
                        if force_state:
                            here = ob.trace[trace_keys[gi]].sort(dim=-1)[0].numpy()          # what the oracle selected on THIS machine
                            e = ob.policy[gn].last_input
                            nrm = torch.linalg.vector_norm(e.double(), dim=-1).sort(dim=-1, descending=True)[0]
                            m_here = float(((nrm[..., k - 1] - nrm[..., k]) / nrm[..., k - 1]).min())
                            tw_eq = None if twin is None else bool(np.array_equal(twin[bi].policy[gn].mine.reshape(-1).numpy(), here.reshape(-1)))
                            local[(t, bi, gi)] = (bool(np.array_equal(mine.reshape(-1), here.reshape(-1))), tw_eq, m_here,
                                                  bool(np.array_equal(here.reshape(-1), idx_gold[t - 1, bi, gi].astype(np.int64).reshape(-1))))
                        if per_set:
                            continue
                        if margins[t - 1, bi, gi] >= min_margin:
                            checked += 1
                            mismatched += (not same)
                            per_gate[gn][0] += 1
                            per_gate[gn][1] += same
                            if not same:
                                H.report(f"\n[teacher-forced {fixture} {mode}] frame {t} block {bi} {gn}: HIP set differs, "
                                         f"reference margin {margins[t - 1, bi, gi]:.3e}")
                x = y_ref
    if per_set:
        return {"sets": sets, "worst": worst, "twin_sets": twin_sets, "local": local}
    return checked, mismatched, per_gate, worst


@pytest.mark.parametrize("case,k,grid,seed", [("vivit_fp16", 128, 14, 77), ("vivit401_fp16", 50, 20, 79)])
def test_timing_configs_vivit(golden_dir, case, k, grid, seed):
    """The reference's GPU timing setting `matmul_2_cast: "float16"` (configs/time/vivit_epic_kitchens/_cuda.yml:5) at FULL size: config 2's
    model (197 tokens, k = 128) and the EPIC-Kitchens model the reference times (20 x 20 + class token = 401 tokens -- the fused
    attention path for more than 256 tokens, evt_attention_stream, without a key grid -- k = 50,
    configs/models/vivit_b_epic_kitchens.yml:5-8, configs/time/vivit_epic_kitchens/temporal_cuda.yml:5).  Teacher-forced block by block
    against the REAL reference's golden index sets (`timing_configs.npz`): block outputs within 1e-3, every set at margin >= 1e-3 equal."""
    g = H.load_npz(os.path.join(golden_dir, "timing_configs.npz"))
    assert int(g[f"{case}__seed"]) == seed and int(g[f"{case}__k"]) == k
    model, sd, cls, ln_w, ln_b = H.vivit_oracle("float16", seed=seed, k=k, grid=grid)
    bb = H.product_vivit(sd, "float16", grid=grid)
    checked, mismatched, per_gate, worst = _teacher_forced_clip(model, bb, cls, g, case, k, 4, seed + 2, 1e-3, 1e-3, "timing_configs.npz")
    H.report(f"\n[teacher-forced timing_configs.npz {case}: {grid * grid + 1} tokens, k = {k}, float16 cast] worst block-output error {worst:.3e} "
             f"(bar 1e-3); index sets at reference margin >= 1e-3: {checked - mismatched}/{checked} equal; per gate (checked, equal): {per_gate}")
    assert checked >= 60 and mismatched == 0, (checked, mismatched)


@pytest.mark.parametrize("case,grid,k,pool,stride", [("vitdet672_fp16", 42, 256, None, 16), ("vitdet672_pool2", 42, 256, 2, 16),
                                                     ("vitdet1024_k512", 64, 512, None, 64)])
def test_timing_configs_vitdet(golden_dir, case, grid, k, pool, stride):
    """ViTDet-B in the reference's GPU timing / evaluation settings at FULL size against the REAL reference (`timing_configs.npz`),
    decisions teacher-forced ON THE DEVICE (after every selection the tap compares the HIP list with the reference's and overwrites the
    device-side list with it: with thousands of tokens some gate always has a margin of ~1e-6, and ONE token refreshed a frame earlier or
    later shows up as a 1e-2 difference in that row): float16 A.v cast in the global blocks (configs/time/vitdet_vid/_cuda.yml:5-7) at 672^2 top-k 256; the
    paper's 'spatiotemporal' variant -- K / V pooled 2 x 2 in the global blocks (configs/evaluate/vitdet_vid/_spatial.yml:4-6,
    blocks.py:303-326,525-540), N = 1764 queries against 441 pooled keys; and 1024^2 with top-k 512
    (configs/time/vitdet_vid/temporal_1024_cuda.yml:5).  Outputs (sparse slice and the full refreshed rows) within 1e-3; every gate's
    index set with a reference margin >= 1e-3 bit-equal (the others are reported)."""
    from eventful_transformer import blocks as evt_blocks
    g = H.load_npz(os.path.join(golden_dir, "timing_configs.npz"))
    view = {key[len(case) + 2:]: g[key] for key in g.files if key.startswith(case + "__")}
    idx_gold, margins = view["idx"], view["margins"]
    gi_of = {"qkv": 0, "projection": 1, "mlp": 2}
    st = {"n": 0, "equal": 0, "checked": 0, "checked_equal": 0, "differ": []}
    buckets = {lo: [0, 0] for lo in (0.0, 1e-6, 1e-5, 1e-4, 1e-3)}   # reference margin in [lo, next lo): [sets, equal]

    def tap(_blk, tag, idx, count):
        n = st["n"]
        st["n"] += 1
        t, bi, gi = n // 36, (n // 3) % 12, gi_of[tag]
        want = torch.from_numpy(idx_gold[t, bi, gi, 0].astype(np.int64))
        same = torch.equal(idx[0].cpu().long(), want)
        m = float(margins[t, bi, gi])
        st["equal"] += same
        if m >= 1e-3:
            st["checked"] += 1
            st["checked_equal"] += same
        if not same:
            st["differ"].append((t + 1, bi, tag, f"{m:.1e}"))
            idx[0] = want.to(idx.device, torch.int32)

    evt_blocks.INDEX_TAP = tap
    try:
        _vitdet_run(golden_dir, view, grid, "TokenNormTopK", dict(k=k), "float16",
                    lambda steps, g_: O.make_token_stream(1, grid * grid, 768, steps, k, seed=int(g_["seed"]) + 2, small=0.01),
                    stride, 1e-3, pool_size=pool, name=f"timing_configs.npz {case}")
    finally:
        evt_blocks.INDEX_TAP = None
    H.report(f"    [{case}] gate index sets equal to the reference's: {st['equal']}/{st['n']}; at reference margin >= 1e-3: "
             f"{st['checked_equal']}/{st['checked']}; differing sets (frame, block, gate, margin): {st['differ']}")
    assert st["n"] == 72 and st["checked"] >= 40 and st["checked_equal"] == st["checked"], st


def _vitdet_run(golden_dir, fixture, grid, policy_cls, policy_kw, cast, stream_fn, stride, tol, pool_size=None, name=None):
    """Runs the product ViTDet backbone over the fixture's stream and compares every frame with the REAL reference's outputs:
    the sparse slice (every `stride`-th token: mostly tokens no gate touched, whose error is the dense first frame's) and --
    separately -- the FULL rows of the tokens the last block's MLP gate refreshed in that frame (`yrow_<t>`), i.e. the
    gated update itself."""
    from eventful_transformer import policies
    g = fixture if isinstance(fixture, dict) else H.load_npz(os.path.join(golden_dir, fixture))
    fixture = name or fixture
    seed = int(g["seed"])

    def rel_for(i):
        return (14, 14) if i in H.VITDET_WINDOWED else (64, 64)

    sd = H.backbone_params(12, 768, 4, seed, 14 * 14, rel_for=rel_for, qk_std=float(g["qk_std"]) if "qk_std" in g else None)
    bb = H.product_vitdet(grid, sd, cast, pool_size=pool_size)
    H.set_policies(bb, getattr(policies, policy_cls), **policy_kw)
    want = torch.from_numpy(g["y_slice"])
    xs = stream_fn(want.shape[0], g)
    worst, worst_rows, n_rows = [], [], []
    with torch.inference_mode():
        for t in range(want.shape[0]):
            y_full = bb(xs[t].to(DEV)).cpu()
            worst.append(float((y_full[:, ::stride] - want[t]).abs().max()))
            if t > 0:
                rows = torch.from_numpy(g[f"yrowidx_{t}"].astype(np.int64))
                n_rows.append(int(rows.numel()))
                worst_rows.append(float((y_full[0, rows] - torch.from_numpy(g[f"yrow_{t}"])).abs().max()) if rows.numel() else 0.0)
    H.report(f"\n[{fixture}] max |out - reference| per frame, every {stride}-th token: {[f'{w:.2e}' for w in worst]}; FULL rows of the "
             f"tokens the last block's MLP gate refreshed ({n_rows} rows): {[f'{w:.2e}' for w in worst_rows]} (tolerance {tol:.0e})")
    assert max(worst) <= tol, worst
    assert len(worst_rows) == want.shape[0] - 1 and max(worst_rows) <= tol, worst_rows
    return g, bb


@pytest.mark.parametrize("forced", [False, True])
@pytest.mark.parametrize("dense_norm_rows", [None, 0])
def test_vitdet_672_topk(golden_dir, monkeypatch, dense_norm_rows, forced):
    """BASELINE config 3: ViTDet-B backbone 672^2 (N=1764; 8 windowed EventfulTokenwiseBlocks with 14x14
    windows + rel-pos, 4 global EventfulBlocks with rel-pos resized 64->42), top-k 256, fp32, against the REAL reference's
    golden outputs AND its 72 gate index sets (2 gated frames x 12 blocks x 3 gates; metric: "gate-index bit-exact").
      * free-running (forced = False): every set whose reference margin between the k-th and (k+1)-th norm is >= 1e-3 must be
        bit-equal (the sets below that margin are reported);
      * teacher-forced on the device (forced = True): after every selection the tap compares the HIP list with the reference's
        and then overwrites the device-side list with the reference's, so a near-tie cannot fork the states -- every set with
        a reference margin >= 1e-4 must be bit-equal, whatever came before it (1e-4: the gate input carries the 1e-5 relative
        error of the split-precision products, on a delta norm that is itself ~1e-1 of the token norm).
    dense_norm_rows = 0: the windowed blocks' projection gates select on the per-head norms from the resident K8 epilogue (the
    batched-streams path) instead of a row pass."""
    from eventful_transformer import blocks as evt_blocks
    if dense_norm_rows is not None:
        monkeypatch.setattr(evt_blocks, "FUSE_DENSE_NORM_ROWS", dense_norm_rows)
    gold = H.load_npz(os.path.join(golden_dir, "vitdet_672.npz"))
    idx_gold, margins = gold["idx"], gold["margins"]
    gi_of = {"qkv": 0, "projection": 1, "mlp": 2}
    bar = 1e-4 if forced else 1e-3
    st = {"n": 0, "equal": 0, "checked": 0, "checked_equal": 0, "differ": []}
    buckets = {lo: [0, 0] for lo in (0.0, 1e-6, 1e-5, 1e-4, 1e-3)}   # reference margin in [lo, next lo): [sets, equal]

    def tap(_blk, tag, idx, count):
        n = st["n"]
        st["n"] += 1
        t, bi, gi = n // 36, (n // 3) % 12, gi_of[tag]
        assert n % 3 == gi and count is None
        want = torch.from_numpy(idx_gold[t, bi, gi, 0].astype(np.int64))
        same = torch.equal(idx[0].cpu().long(), want)
        m = float(margins[t, bi, gi])
        st["equal"] += same
        lo = max(b_ for b_ in buckets if m >= b_)
        buckets[lo][0] += 1
        buckets[lo][1] += same
        if m >= bar:
            st["checked"] += 1
            st["checked_equal"] += same
        if not same:
            st["differ"].append((t + 1, bi, tag, m))
            if forced:
                idx[0] = want.to(idx.device, torch.int32)

    evt_blocks.INDEX_TAP = tap
    try:
        _vitdet_run(golden_dir, "vitdet_672.npz", 42, "TokenNormTopK", dict(k=256), None,
                    lambda steps, g: O.make_token_stream(1, 42 * 42, 768, steps, 256, seed=int(g["seed"]) + 2, small=0.01),
                    16, 1e-3)
    finally:
        evt_blocks.INDEX_TAP = None
    H.report(f"    [{'teacher-forced' if forced else 'free-running'}, dense_norm_rows={dense_norm_rows}] gate index sets equal to the "
             f"reference's: {st['equal']}/{st['n']}; at reference margin >= {bar:.0e}: {st['checked_equal']}/{st['checked']}; by reference margin "
             f"bucket [lo, next) (sets, equal): { {f'{lo:.0e}': v for lo, v in buckets.items()} }; differing "
             f"sets (frame, block, gate, reference margin): {[(t, b, g_, f'{m:.1e}') for t, b, g_, m in st['differ']]}")
    assert st["n"] == idx_gold.shape[0] * 36 == 72
    assert st["checked"] >= (60 if forced else 50) and st["checked_equal"] == st["checked"], st


@pytest.mark.parametrize("thr,fixture", [(1.0, "vitdet_1024.npz"), (0.2, "vitdet_1024_thr0.2.npz"), (5.0, "vitdet_1024_thr5.npz")])
def test_vitdet_1024_threshold(golden_dir, thr, fixture):
    """BASELINE config 5: ViTDet-B backbone 1024^2 (N=4096, windows padded 64->70), threshold policy with data-dependent r
    per gate AND per frame kept on the device, global blocks bf16 A.v cast -- at all three thresholds of
    configs/evaluate/vitdet_vid/threshold_1024.yml:5, over 4 gated frames of a stream with continuous perturbation magnitudes
    (O.make_varied_threshold_stream) against the REAL reference's golden outputs, counts and index lists.

    With thousands of tokens and continuous norms some token always sits within ~1e-6 of the threshold (fixture `margins`), so
    the decisions are teacher-forced ON THE DEVICE: after every selection the diagnostic tap compares the HIP kernel's own list
    and count with the reference's and then overwrites the device-side list and count with the reference's -- the launches
    that consume them are the product's own (device-side count, capacity N).  Required: every HIP list equals the reference's
    except for tokens the fixture lists as NEAR the threshold (within 1e-3 relative); the counts agree accordingly; the
    backbone output of every frame is within 1e-3 of the reference's."""
    from eventful_transformer import blocks as evt_blocks
    g = H.load_npz(os.path.join(golden_dir, fixture))
    keys = {"qkv": "qkv_index", "projection": "projection_index", "mlp": "mlp_index"}
    state = {"n": 0, "equal": 0, "near_only": 0, "worst_rel": 0.0, "bad": [], "strict": 0, "strict_bad": []}
    gate_col = {"qkv": 0, "projection": 1, "mlp": 2}
    counts = []

    def tap(_blk, tag, idx, count):
        n = state["n"]
        state["n"] += 1
        t, bi = 1 + n // 36, (n // 3) % 12
        want = torch.from_numpy(g[f"idx_{t}_{bi}_{keys[tag]}"].reshape(-1).astype(np.int64))
        mine = idx[0, : int(count[0])].cpu().long()
        counts.append(int(mine.numel()))
        # a gate none of whose tokens sits within 1e-3 (relative) of the threshold leaves no room for another summation order:
        # its list (and so its count) must be EXACTLY the reference's
        if float(g["margins"][t - 1, bi, gate_col[tag]]) >= 1e-3:
            state["strict"] += 1
            if not torch.equal(mine, want):
                state["strict_bad"].append((t, bi, tag))
        if torch.equal(mine, want):
            state["equal"] += 1
        else:
            near = g[f"near_{t}_{bi}_{keys[tag]}"].astype(np.int64)
            rel = g[f"nearrel_{t}_{bi}_{keys[tag]}"]
            diff = np.setxor1d(mine.numpy(), want.numpy())
            if np.isin(diff, near).all():
                state["near_only"] += 1
                state["worst_rel"] = max(state["worst_rel"], float(max(rel[near == d][0] for d in diff)))
            else:
                state["bad"].append((t, bi, tag, int(mine.numel()), int(want.numel())))
            # force the reference's decision into the device-side list and count the consuming launches read
            idx[0, : want.numel()] = want.to(idx.device, torch.int32)
            count[0] = want.numel()

    evt_blocks.INDEX_TAP = tap
    try:
        _vitdet_run(golden_dir, fixture, 64, "TokenNormThreshold", dict(threshold=thr), "bfloat16",
                    lambda steps, g_: O.make_varied_threshold_stream(64 * 64, 768, steps, int(g_["stream_seed"])), 64, 1e-3)
    finally:
        evt_blocks.INDEX_TAP = None
    assert float(g["threshold"]) == thr
    frames = g["counts"].shape[0]
    assert frames >= 4 and state["n"] == frames * 36
    ref_counts = g["counts"]
    H.report(f"    thr {thr}: reference gate counts range {ref_counts.min()}..{ref_counts.max()}, {len(set(ref_counts.reshape(-1).tolist()))} "
             f"distinct values over {ref_counts.size} gates; per frame (qkv of block 0 / projection of block 2 / mlp of block 11): "
             f"{[(int(a[0, 0]), int(a[2, 1]), int(a[11, 2])) for a in ref_counts]}; closest token to the threshold {float(g['margins'].min()):.1e}; "
             f"HIP lists equal to the reference's: {state['equal']}/{state['n']}, differing in NEAR tokens only: {state['near_only']} "
             f"(largest relative distance of a flipped token {state['worst_rel']:.1e}), otherwise: {len(state['bad'])}")
    H.report(f"    gates with every token at least 1e-3 (relative) away from the threshold: {state['strict']}, all of them with exactly the reference's list: "
             f"{not state['strict_bad']}")
    assert not state["bad"], state["bad"]
    assert not state["strict_bad"] and state["strict"] >= 20, (state["strict"], state["strict_bad"])
    assert state["equal"] >= 0.9 * state["n"], state
    assert len(set(ref_counts.reshape(-1).tolist())) >= 6          # r really varies with the frame (and, mildly, with the gate)
    got = np.asarray(counts).reshape(frames, 12, 3)
    assert np.abs(got - ref_counts).max() <= 3, np.abs(got - ref_counts).max()


def test_forward_hooks_see_tensors(golden_dir):
    """Block chaining (blocks.PendingSum) is backbone-internal: a forward hook on a block -- the reference's users inspect
    block outputs that way -- receives a plain tensor, and hooking does not change the backbone's output."""
    from eventful_transformer import policies
    sd = H.backbone_params(3, 64, 4, 5, 37)
    from eventful_transformer.backbones import ViTBackbone
    cfg = dict(dim=64, heads=4, mlp_ratio=4)
    outs = []
    for hooked in (False, True):
        bb = ViTBackbone(block_config=cfg, depth=3, position_encoding_size=(6, 6), input_size=(6, 6), block_class="EventfulBlock",
                         has_class_token=True)
        bb.load_state_dict(sd, strict=True)
        bb = bb.eval().to(DEV)
        H.set_policies(bb, policies.TokenNormTopK, k=12)
        seen = []
        if hooked:
            bb.blocks[1].register_forward_hook(lambda m, i, o: seen.append((type(i[0]), type(o))))
            bb.blocks.register_forward_hook(lambda m, i, o: seen.append(("container", type(o))))
        xs = O.make_token_stream(2, 37, 64, 3, 12, seed=6, small=0.02)
        with torch.inference_mode():
            ys = [bb(xs[t].to(DEV)).cpu() for t in range(3)]
        outs.append(ys)
        if hooked:
            assert len(seen) == 6 and all(t is torch.Tensor for pair in seen for t in pair if t != "container"), seen
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_pooled_block_batch_is_per_clip():
    """K/V pooling with batch > 1: the HIP path de-duplicates pooled indices PER CLIP, so a batch equals the
    clips run one by one (the reference's `.unique(dim=-1)` couples clips for batch > 1 -- blocks.py:535-539 --
    and is only exact at batch 1, the regime its pooled configs use).  Checked against batch-1 oracle runs."""
    from eventful_transformer import policies
    params = O.make_block_params(64, 4, seed=21, std=0.08, rel_sizes=(6, 6), head_dim=16)
    kw = dict(pool_size=2, relative_embedding_size=(6, 6))
    blk = H.product_block("EventfulBlock", params, 64, 4, (6, 6), **kw)
    H.set_policies(blk, policies.TokenNormTopK, k=10)
    xs = O.make_token_stream(3, 36, 64, 4, 10, seed=22, small=0.02)
    oracles = []
    for b in range(3):
        o = O.BlockOracle("EventfulBlock", params, 64, 4, (6, 6), **kw)
        o.set_policy(lambda: O.TopK(10))
        oracles.append(o)
    with torch.inference_mode():
        for t in range(4):
            y = blk(xs[t].to(DEV)).cpu()
            ref = torch.cat([oracles[b].forward(xs[t, b:b + 1].clone()) for b in range(3)])
            assert float((y - ref).abs().max()) <= 2e-4, t


@pytest.mark.parametrize("cast,policy", [(None, ("topk", 20)), ("bfloat16", ("topk", 20)), (None, ("thr", 0.8))])
def test_pooled_eventful_block_head_dim_64(cast, policy):
    """Pooled K/V through the FUSED K5+K6 kernel (head dim 64, N = 64 queries x Nk = 16 pooled keys, rel-pos
    tables pooled along the key axis), batch 1, 4 frames, against the oracle."""
    from eventful_transformer import policies
    dim, heads, grid = 256, 4, (8, 8)
    kw = dict(pool_size=2, relative_embedding_size=(8, 8))
    if cast:
        kw["matmul_2_cast"] = cast
    params = O.make_block_params(dim, 4, seed=31, std=0.05, rel_sizes=(8, 8), head_dim=64)
    blk = H.product_block("EventfulBlock", params, dim, heads, grid, **kw)
    ora = O.BlockOracle("EventfulBlock", params, dim, heads, grid, **kw)
    H.product_policy(blk, policy)
    ora.set_policy(H.oracle_policy(policy))
    xs = O.make_token_stream(1, 64, dim, 4, 20, seed=32, small=0.02)
    tol = 3e-4 if cast is None else 1e-3
    with torch.inference_mode():
        for t in range(4):
            y = blk(xs[t].to(DEV)).cpu()
            ref = ora.forward(xs[t].clone())
            assert float((y - ref).abs().max()) <= tol, (cast, policy, t, float((y - ref).abs().max()))
            if t:
                assert blk.matmul_gate.p.shape == (1, 4, 64, 16) and blk.v_gate.p.shape == (1, 4, 16, 64)


@pytest.mark.parametrize("cast,policy,rel", [(None, ("topk", 90), True), ("bfloat16", ("topk", 90), True), (None, ("thr", 0.8), True),
                                             ("float16", ("topk", 50), False)])
def test_pooled_eventful_block_on_the_stream_kernel(cast, policy, rel):
    """Pooled K/V with more than 256 tokens (the 'spatiotemporal' ViTDet variant's global blocks, blocks.py:303-326, 509-511,
    525-540): ONE evt_attention_stream launch per frame with Nk = N / 4 pooled cells (ABI 8) instead of K4 + K5+K6 with an fp32
    (B,H,N,Nk) score state.  N = 324 queries x 81 cells, batch 1, 4 frames against the oracle; the other path (EVT_STREAM_POOLED=0)
    gives the same outputs to rounding."""
    from eventful_transformer import blocks as blocks_mod
    dim, heads, grid = 128, 2, (18, 18)
    kw = dict(pool_size=2)
    if rel:
        kw["relative_embedding_size"] = (18, 18)
    if cast:
        kw["matmul_2_cast"] = cast
    params = O.make_block_params(dim, 4, seed=41, std=0.05, rel_sizes=(18, 18) if rel else None, head_dim=64)
    xs = O.make_token_stream(1, 324, dim, 4, 90, seed=42, small=0.02)
    ora = O.BlockOracle("EventfulBlock", params, dim, heads, grid, **kw)
    ora.set_policy(H.oracle_policy(policy))
    tol = 3e-4 if cast is None else 1e-3
    outs = {}
    for pooled_stream in (True, False):
        old = blocks_mod.STREAM_POOLED
        blocks_mod.STREAM_POOLED = pooled_stream
        try:
            blk = H.product_block("EventfulBlock", params, dim, heads, grid, **kw)
            H.product_policy(blk, policy)
            ys = []
            with torch.inference_mode():
                for t in range(4):
                    ys.append(blk(xs[t].to(DEV)).cpu())
            from eventful_transformer import _native
            on_stream = pooled_stream and _native.FUSED_QK
            assert (getattr(blk.matmul_gate, "_state_t", None) is not None) == on_stream
            if on_stream:
                assert blk.matmul_gate._state_t.shape == (1, 2, 81, 324) and blk.matmul_gate.p.shape == (1, 2, 324, 81)
                assert blk.v_gate.p.shape == (1, 2, 81, 64)
                assert blk.matmul_accumulator_1.product.shape == (1, 2, 324, 81)   # refreshed lazily from the token buffer
            outs[pooled_stream] = ys
        finally:
            blocks_mod.STREAM_POOLED = old
    for t in range(4):
        ref = ora.forward(xs[t].clone())
        err = float((outs[True][t] - ref).abs().max())
        assert err <= tol, (cast, policy, t, err)
        assert float((outs[True][t] - outs[False][t]).abs().max()) <= tol, (cast, policy, t)


@pytest.mark.parametrize("policy,kw,windowed,resized", [("TokenNormTopK", dict(k=20), False, False), ("TokenNormThreshold", dict(threshold=0.8), False, False),
                                                        ("TokenNormTopK", dict(k=30), True, False), ("TokenNormTopK", dict(k=20), False, True)])
def test_frame_graphs_replay_is_bit_identical(policy, kw, windowed, resized):
    """HIP-graph replay of the first / incremental frame (graphs.py) against the eager path: same kernels on the
    same buffers, so outputs must be bit-identical over several clips, including the clip boundary handled by
    replaying the first-frame graph instead of reset()."""
    from eventful_transformer import policies
    from eventful_transformer.backbones import ViTBackbone
    from eventful_transformer.graphs import FrameGraphs

    def make():
        torch.manual_seed(3)
        cfg = dict(dim=128, heads=2, mlp_ratio=2, matmul_2_cast="bfloat16")
        extra = {}
        if windowed:
            cfg.update(window_size=(4, 4), relative_embedding_size=(8, 8))
            extra = dict(window_indices=(0, 2), windowed_class="EventfulTokenwiseBlock", windowed_overrides=dict(matmul_2_cast=None))
        if resized:   # rel-pos tables and the position encoding are bicubically resized: caches rebuilt after reset(),
            cfg.update(relative_embedding_size=(6, 6))   # which must happen OUTSIDE the graph capture (graphs.py)
        bb = ViTBackbone(block_config=cfg, depth=3, position_encoding_size=(4, 4) if resized else (8, 8), input_size=(8, 8),
                         block_class="EventfulBlock", **extra)
        for p_ in bb.parameters():
            torch.nn.init.normal_(p_, std=0.05)
        bb = bb.eval().to(DEV)
        H.set_policies(bb, getattr(policies, policy), **kw)
        return bb

    clips = [O.make_token_stream(2, 64, 128, 6, 20, seed=40 + c, small=0.01).to(DEV) for c in range(3)]
    eager, graphed = make(), FrameGraphs(make())
    with torch.inference_mode():
        for clip in clips:
            eager.reset()
            graphed.reset()
            for t in range(clip.shape[0]):
                want = eager(clip[t])
                got = graphed(clip[t])
                assert torch.equal(got, want), (t, float((got - want).abs().max()))
    assert graphed._first is not None and graphed._inc is not None
    with pytest.raises(RuntimeError, match="differs from the captured"):
        graphed(clips[0][0][:1])


@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("policy,kw", [("TokenNormTopK", dict(k=30)), ("TokenNormThreshold", dict(threshold=0.8))])
def test_pipelined_frames_are_bit_identical(policy, kw, lanes):
    """graphs.FrameGraphs.run_pipelined: `lanes` consecutive frames of one stream captured side by side on `lanes` HIP streams
    (block i of frame t+1 behind block i+1 of frame t) must give bit for bit what frame-by-frame execution gives, on a
    backbone with windowed and global blocks, rel-pos terms and chained (PendingSum) block boundaries."""
    from eventful_transformer import policies
    from eventful_transformer.backbones import ViTBackbone
    from eventful_transformer.graphs import FrameGraphs

    def make():
        torch.manual_seed(3)
        cfg = dict(dim=128, heads=2, mlp_ratio=2, matmul_2_cast="bfloat16", window_size=(4, 4), relative_embedding_size=(8, 8))
        bb = ViTBackbone(block_config=cfg, depth=4, position_encoding_size=(8, 8), input_size=(8, 8), block_class="EventfulBlock",
                         window_indices=(0, 2), windowed_class="EventfulTokenwiseBlock", windowed_overrides=dict(matmul_2_cast=None))
        for p_ in bb.parameters():
            torch.nn.init.normal_(p_, std=0.05)
        bb = bb.eval().to(DEV)
        H.set_policies(bb, getattr(policies, policy), **kw)
        return bb

    T = 1 + 4 * lanes
    clips = [O.make_token_stream(1, 64, 128, T, 20, seed=60 + c, small=0.01).to(DEV) for c in range(2)]
    eager, piped = make(), FrameGraphs(make())
    with torch.inference_mode():
        for clip in clips:
            eager.reset()
            piped.reset()
            want = [eager(clip[t]).clone() for t in range(T)]
            got = [piped(clip[0]).clone()]
            for t in range(1, T, lanes):
                got += [y.clone() for y in piped.run_pipelined(clip[t:t + lanes])]
            for t in range(T):
                assert torch.equal(got[t], want[t]), (t, float((got[t] - want[t]).abs().max()))
    assert piped._pipe is not None
    with pytest.raises(RuntimeError, match="first frame"):
        piped.reset()
        piped.run_pipelined(clips[0][1:1 + lanes])


def test_vivit_sized_backbone_reruns_are_bit_identical():
    """I7 at the benchmark's shapes (N = 197, D = 768, 12 heads, bf16 A.v cast, k = 128; 2 blocks, 3 clips x 4 frames):
    every kernel on the path -- K8 with state outputs on the first frame, split-precision K3/K4, the fused K5+K6,
    split-K for this small batch -- is deterministic, so two runs from reset() agree bit for bit."""
    from eventful_transformer import policies
    from eventful_transformer.backbones import ViTBackbone

    torch.manual_seed(11)
    bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4, matmul_2_cast="bfloat16"), depth=2,
                     position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
    for p_ in bb.parameters():
        torch.nn.init.normal_(p_, std=0.02)
    bb = bb.eval().to(DEV)
    H.set_policies(bb, policies.TokenNormTopK, k=128)
    xs = O.make_token_stream(3, 197, 768, 4, 128, seed=77, small=0.01).to(DEV)
    runs = []
    with torch.inference_mode():
        for _ in range(2):
            bb.reset()
            runs.append(torch.stack([bb(xs[t]).clone() for t in range(xs.shape[0])]))
    assert torch.isfinite(runs[0]).all()
    assert torch.equal(runs[0], runs[1])


def _per_clip_margin(e, k):
    n = torch.linalg.vector_norm(e.double(), dim=-1)
    s = n.sort(dim=-1, descending=True)[0]
    return ((s[:, k - 1] - s[:, k]) / s[:, k - 1]).numpy()


@pytest.mark.parametrize("cast,out_tol", [(None, 5e-4), ("bfloat16", 1e-3)])
def test_two_blocks_batch64_operating_point(cast, out_tol):
    """The launch configuration the headline number is measured at, checked against the CPU oracle: two ViViT-B
    `EventfulBlock`s (N = 197, D = 768, k = 128), B = 64 clips => M = 8192 gated rows: >= 128 output tiles, so the
    gated linears run WITHOUT split-K (bias / GELU / scatter in the GEMM epilogue, XCD tile map, gathered A rows with
    the fused refresh of the gate reference) and K4 / K5+K6 run at gridDim.y = B*H = 768.  Three frames,
    teacher-forced gate by gate (both sides refresh the same tokens, the HIP selection is recorded):
      * EVERY clip's block output within tolerance of the oracle's;
      * the HIP policy's own index sets identical to the oracle's for every (clip, gate) whose margin is >= 1e-3."""
    B, k, steps, depth = 64, 128, 3, 2
    kw = dict(matmul_2_cast=cast) if cast else {}
    sd = H.backbone_params(depth, 768, 4, 123, 197)
    oras, blks = [], []
    for i in range(depth):
        o = O.BlockOracle("EventfulBlock", H.block_params_of(sd, i), 768, 12, (14, 14), **kw)
        o.set_policy(lambda: O.TopK(k))
        oras.append(o)
        b = H.product_block("EventfulBlock", H.block_params_of(sd, i), 768, 12, (14, 14), **kw)
        for gn in ("qkv_gate", "projection_gate", "mlp_gate", "v_gate", "matmul_gate"):
            getattr(b, gn).policy = _ForcedPolicy(k)
        blks.append(b)
    xs = O.make_token_stream(B, 197, 768, steps, k, seed=321, small=0.01)
    gates = (("qkv_gate", "qkv_index"), ("projection_gate", "projection_index"), ("mlp_gate", "mlp_index"))
    checked = mismatched = 0
    worst = 0.0
    with torch.inference_mode():
        for t in range(steps):
            x = xs[t]
            for ob, pb in zip(oras, blks):
                y_ref = ob.forward(x.clone())
                if t > 0:
                    for gn, tk in gates:
                        getattr(pb, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].to(DEV)
                y_dev = pb(x.to(DEV)).cpu()
                err = (y_dev - y_ref).abs().amax(dim=(1, 2))      # per clip
                worst = max(worst, float(err.max()))
                assert float(err.max()) <= out_tol, (cast, t, int(err.argmax()), float(err.max()))
                if t > 0:
                    for gn, tk in gates:
                        mine = getattr(pb, gn).policy.mine.cpu()
                        want = ob.trace[tk].sort(dim=-1)[0]
                        margin = _per_clip_margin(ob.policy[gn].last_input, k)
                        for b_ in range(B):
                            if margin[b_] >= 1e-3:
                                checked += 1
                                mismatched += not torch.equal(mine[b_], want[b_])
                x = y_ref
    H.report(f"\n[B=64 operating point cast={cast}] worst block-output error {worst:.3e}; index sets {checked - mismatched}/{checked}")
    assert checked >= 2 * B and mismatched == 0, (checked, mismatched)


def _ats_cases():
    cs = {}
    for kind in ("Block", "EventfulTokenwiseBlock", "EventfulMatmul1Block", "EventfulBlock"):
        cs[f"{kind}_ats"] = (kind, {})
    cs["EventfulBlock_ats_bf16"] = ("EventfulBlock", dict(matmul_2_cast="bfloat16"))
    cs["EventfulMatmul1Block_ats_bf16"] = ("EventfulMatmul1Block", dict(matmul_2_cast="bfloat16"))
    return cs


@pytest.mark.parametrize("name", list(_ats_cases().keys()))
def test_adaptive_token_sampling_vs_golden(golden_dir, name):
    """`ats_fraction` (blocks.py:150-181,378-391,196-203) for all four block classes against outputs of the REAL
    reference, at the only shape the reference's ATS executes (batch == heads: it sums the scores over the batch
    axis): block outputs on the selected tokens and the stabilised per-clip index lists, 4 frames."""
    from eventful_transformer import policies
    g = H.load_npz(os.path.join(golden_dir, "ats.npz"))
    kind, kw = _ats_cases()[name]
    params = O.make_block_params(64, 4, seed=int(g[f"{name}__param_seed"]), std=0.08)
    blk = H.product_block(kind, params, 64, 4, (6, 6), ats_fraction=float(g["fraction"]), **kw)
    if kind != "Block":
        H.set_policies(blk, policies.TokenNormTopK, k=int(g["k"]))
    xs, ys = torch.from_numpy(g[f"{name}__x"]), torch.from_numpy(g[f"{name}__y"])
    tol = 2e-4 if not kw else 2e-3
    with torch.inference_mode():
        for t in range(xs.shape[0]):
            y = blk(xs[t].to(DEV)).cpu()
            assert y.shape == ys[t].shape, (name, t, y.shape)
            assert np.array_equal(blk.last_ats_indices.cpu().numpy(), g[f"{name}__ats_index"][t]), (name, t)
            err = float((y - ys[t]).abs().max())
            assert err <= tol, (name, t, err)
    blk.reset()
    assert blk.last_ats_indices is None


def test_adaptive_token_sampling_needs_batch_equal_heads():
    params = O.make_block_params(64, 4, seed=1, std=0.08)
    blk = H.product_block("Block", params, 64, 4, (6, 6), ats_fraction=0.5)
    with pytest.raises(RuntimeError, match="batch == heads"):
        blk(torch.randn(2, 37, 64, device=DEV))


def test_lazy_qk_state_is_exact_when_read():
    """ViViT-shaped EventfulBlock (head dim 64, <= 256 tokens): on gated frames the scores are computed inside the fused
    attention kernel and `matmul_accumulator_1.product` is only refreshed when read -- it must then be the exact q.k^T of the
    CURRENT token buffer (I1), and EVT_FUSED_QK=0 (K4 + stored state) must give the same block outputs to rounding."""
    from eventful_transformer import _native, policies
    sd = H.backbone_params(1, 768, 4, 5, 197)
    xs = O.make_token_stream(2, 197, 768, 3, 128, seed=9, small=0.01)
    outs = {}
    for fused in (True, False):
        old, _native.FUSED_QK = _native.FUSED_QK, fused
        try:
            blk = H.product_block("EventfulBlock", H.block_params_of(sd, 0), 768, 12, (14, 14), matmul_2_cast="bfloat16")
            H.set_policies(blk, policies.TokenNormTopK, k=128)
            with torch.inference_mode():
                outs[fused] = [blk(xs[t].to(DEV)).cpu() for t in range(3)]
                state = blk.matmul_accumulator_1.product            # read: triggers the refresh when stale
                q, k, _ = blk.qkv_accumulator.b.cpu().view(2, 197, 3, 12, 64).permute(2, 0, 3, 1, 4)
                want = (q / 8.0) @ k.transpose(-2, -1)
                assert state.shape == (2, 12, 197, 197)
                assert torch.allclose(state.cpu(), want, atol=2e-4), float((state.cpu() - want).abs().max())
        finally:
            _native.FUSED_QK = old
    for a, b_ in zip(outs[True], outs[False]):
        assert float((a - b_).abs().max()) <= 1e-3


@pytest.mark.parametrize("dim,heads,cast", [(384, 8, None), (1280, 16, None), (768, 8, None), (896, 8, None), (1280, 16, "bfloat16"), (896, 8, "bfloat16"),
                                            (384, 6, None), (1024, 16, "bfloat16")])
def test_other_widths_and_head_dims_match_the_oracle(dim, heads, cast):
    """The reference takes any dim / heads (blocks.py:102-116).  Widths other than ViT-B's: head dim 64 at 384 / 1024 (ViT-S / ViT-L:
    the fast paths with other head counts and ragged GEMM column tiles) and head dims 48 / 80 (ViT-H: 1280 / 16) / 96 / 112, which
    take the generic attention kernels (evt_qk, evt_softmax_gate, evt_av: any multiple of 16 up to 128).  One EventfulBlock, N = 197,
    top-k 128, first frame + 2 gated frames on a designed-margin stream, 2 clips against the CPU oracle (decisions teacher-forced)."""
    from eventful_transformer import policies
    n, k = 197, 128
    params = O.make_block_params(dim, 4, seed=dim, std=0.02, head_dim=dim // heads)
    kw = dict(matmul_2_cast=cast) if cast else {}
    ob = O.BlockOracle("EventfulBlock", params, dim, heads, (1, n), **kw)
    ob.set_policy(lambda: O.TopK(k))
    blk = H.product_block("EventfulBlock", params, dim, heads, (1, n), **kw)
    gates = ("qkv_gate", "projection_gate", "mlp_gate")
    for gn in gates:   # decisions teacher-forced: a near-tie at the projection gate (16-bit casts) must not fork the comparison
        getattr(blk, gn).policy = _ForcedPolicy(k)
    xs = O.make_token_stream(2, n, dim, 3, k, seed=dim + 1, small=0.01)
    tol = 2e-4 if cast is None else 2e-3
    with torch.inference_mode():
        for t in range(3):
            y_ref = ob.forward(xs[t])
            if t:
                for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                    getattr(blk, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].to(DEV)
            y = blk(xs[t].to(DEV)).cpu()
            err = float((y - y_ref).abs().max())
            assert err <= tol, (dim, heads, cast, t, err)
            if t and cast is None:   # fp32: the product's own selections are the oracle's
                for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                    assert torch.equal(getattr(blk, gn).policy.mine.sort(dim=-1)[0].cpu(), ob.trace[tk].sort(dim=-1)[0]), (dim, heads, t, gn)
