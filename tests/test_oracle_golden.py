"""CPU: pin oracle/eventful_oracle.py to the golden vectors produced by the REAL reference
(oracle/gen_golden.py).  Same torch build => expected to agree to the last bit on this machine;
the tolerance only absorbs a different host ISA (AVX2 vs AVX-512 ATen kernels) on another box."""
import os

import numpy as np
import pytest
import torch

import eventful_oracle as O
import helpers as H

ATOL = 2e-5


def test_gate_cases(golden_dir):
    g = H.load_npz(os.path.join(golden_dir, "gates.npz"))
    n = int(g["n_cases"])
    assert n >= 15
    for i in range(n):
        kind = bytes(g[f"c{i}_kind"]).decode()
        seed, B, N, D, k = (int(g[f"c{i}_{f}"]) for f in ("seed", "B", "N", "D", "k"))
        want = torch.from_numpy(g[f"c{i}_idx"]).long()
        slot = O.Slot()
        if kind == "topk":
            c, p = O.make_gate_case(seed, B, N, D)
            pol = O.TopK(k)
        else:
            c, p, thr = O.make_threshold_case(seed, N, D, k)
            assert thr == float(g[f"c{i}_thr"])
            pol = O.Threshold(thr)
        O.token_gate(slot, p.clone(), pol)
        c_t, idx = O.token_gate(slot, c.clone(), pol)
        got = idx.sort(dim=-1)[0]
        assert torch.equal(got, want), (i, kind)
        # I3: p[idx] == c[idx] bitwise afterwards; idx unique per row
        assert torch.equal(slot.t.gather(-2, O.rows_index(idx, c.shape)), c_t)
        assert all(len(set(r.tolist())) == r.numel() for r in idx)


@pytest.mark.parametrize("name", list(H.small_cases().keys()))
def test_small_blocks(golden_dir, name):
    g = H.load_npz(os.path.join(golden_dir, "blocks_small.npz"))
    kind, isz, has_cls, kw, pol = H.small_cases()[name]
    params = H.small_case_params(name, H.small_cases()[name], g[f"{name}__param_seed"])
    blk = O.BlockOracle(kind, params, H.SMALL["dim"], H.SMALL["heads"], isz, **kw)
    blk.set_policy(H.oracle_policy(pol))
    xs = torch.from_numpy(g[f"{name}__x"])
    ys = torch.from_numpy(g[f"{name}__y"])
    with torch.inference_mode():
        for t in range(xs.shape[0]):
            y = blk.forward(xs[t].clone())
            assert torch.allclose(y, ys[t], atol=ATOL, rtol=0), (name, t, float((y - ys[t]).abs().max()))
            if kind != "Block" and t > 0:
                for gname, key in (("qkv", "qkv_index"), ("projection", "projection_index"), ("mlp", "mlp_index")):
                    want = torch.from_numpy(g[f"{name}__idx_{gname}_{t}"]).long()
                    assert torch.equal(blk.trace[key].sort(dim=-1)[0], want), (name, t, gname)


def test_invariants_small():
    """I0 (k == N equals dense), I1 (q.k^T state exact), I2/I3 (buffers / gate state)."""
    params = O.make_block_params(64, 4, seed=5, std=0.08)
    xs = O.make_token_stream(2, 37, 64, 4, 12, seed=9, small=0.02)
    dense = O.BlockOracle("Block", params, 64, 4, (6, 6))
    ev = O.BlockOracle("EventfulBlock", params, 64, 4, (6, 6))
    ev.set_policy(lambda: O.TopK(37))
    part = O.BlockOracle("EventfulBlock", params, 64, 4, (6, 6))
    part.set_policy(lambda: O.TopK(12))
    for t in range(4):
        yd = dense.forward(xs[t].clone())
        ye = ev.forward(xs[t].clone())
        assert torch.allclose(yd, ye, atol=1e-4), float((yd - ye).abs().max())  # I0
        before = None if t == 0 else part.s["qkv_accumulator"].t.clone()
        part.forward(xs[t].clone())
        buf = part.s["qkv_accumulator"].t
        q, k, _ = part._heads(buf)
        want = (q / part.scale) @ k.transpose(-2, -1)
        assert torch.allclose(part.s["matmul_accumulator_1"].t, want, atol=1e-5)  # I1
        if t > 0:
            idx = part.trace["qkv_index"]
            mask = torch.ones(2, 37, dtype=torch.bool)
            mask.scatter_(1, idx, False)
            assert torch.equal(buf[mask], before[mask])  # I2: untouched rows bit-unchanged


@pytest.mark.parametrize("fixture,k,mode,cast", [("vivit_b.npz", 128, "fp32", None), ("vivit_b.npz", 128, "bf16", "bfloat16"),
                                                 ("vivit_b_k64.npz", 64, "bf16", "bfloat16"),
                                                 ("vivit_b_sharp.npz", 128, "fp32", None), ("vivit_b_sharp.npz", 128, "bf16", "bfloat16")])
def test_vivit_b_features(golden_dir, fixture, k, mode, cast):
    """ViViT-B spatial sub-model, free-running oracle vs the reference's golden features / index sets: BASELINE
    config 2 (k = 128, 6 frames), config 4's shape (k = 64, T = 32 frames) and config 2 with sharp attention (12 frames)."""
    g = H.load_npz(os.path.join(golden_dir, fixture))
    qk_std = float(g["qk_std"]) if "qk_std" in g.files else None
    model, sd, *_ = H.vivit_oracle(cast, seed=int(g[f"{mode}__seed"]), k=k, qk_std=qk_std)
    feats = torch.from_numpy(g[f"{mode}__features"])
    idx = g[f"{mode}__idx"]
    xs = O.make_token_stream(1, 196, 768, feats.shape[0], k, seed=int(g[f"{mode}__seed"]) + 2, small=0.01)
    with torch.inference_mode():
        for t in range(feats.shape[0]):
            y = model.forward(xs[t])
            assert torch.allclose(y, feats[t], atol=ATOL, rtol=0), (mode, t, float((y - feats[t]).abs().max()))
            if t > 0:
                for bi, blk in enumerate(model.backbone.blocks):
                    for gi, key in enumerate(("qkv_index", "projection_index", "mlp_index")):
                        got = blk.trace[key].sort(dim=-1)[0].numpy()
                        if float(g[f"{mode}__margins"][t - 1, bi, gi]) > 1e-5:
                            assert np.array_equal(got, idx[t - 1, bi, gi].astype(np.int64)), (mode, t, bi, key)


def test_vitdet_672(golden_dir):
    g = H.load_npz(os.path.join(golden_dir, "vitdet_672.npz"))
    seed = int(g["seed"])
    ob, sd = H.vitdet_oracle(42, lambda: O.TopK(256), None, seed)
    want = torch.from_numpy(g["y_slice"])
    xs = O.make_token_stream(1, 42 * 42, 768, want.shape[0], 256, seed=seed + 2, small=0.01)
    with torch.inference_mode():
        for t in range(want.shape[0]):
            y = ob.forward(xs[t].clone())[:, ::16]
            assert torch.allclose(y, want[t], atol=ATOL, rtol=0), (t, float((y - want[t]).abs().max()))


def test_vitdet_1024_threshold(golden_dir):
    """Config 5 fixture (variable r: continuous-magnitude stream, sharp attention): the oracle against the reference's outputs,
    counts and index lists over the first two gated frames (the GPU test walks all of them)."""
    g = H.load_npz(os.path.join(golden_dir, "vitdet_1024.npz"))
    seed, thr = int(g["seed"]), float(g["threshold"])
    ob, sd = H.vitdet_oracle(64, lambda: O.Threshold(thr), "bfloat16", seed, qk_std=float(g["qk_std"]) if "qk_std" in g.files else None)
    want = torch.from_numpy(g["y_slice"])
    frames = min(3, want.shape[0])
    xs = O.make_varied_threshold_stream(64 * 64, 768, want.shape[0], int(g["stream_seed"]))
    with torch.inference_mode():
        for t in range(frames):
            y = ob.forward(xs[t].clone())[:, ::64]
            assert torch.allclose(y, want[t], atol=ATOL, rtol=0), (t, float((y - want[t]).abs().max()))
            if t > 0:
                counts = [blk.trace[k].shape[-1] for blk in ob.blocks for k in ("qkv_index", "projection_index", "mlp_index")]
                assert counts == g["counts"][t - 1].reshape(-1).tolist()
                for bi, blk in enumerate(ob.blocks):
                    for k in ("qkv_index", "projection_index", "mlp_index"):
                        assert np.array_equal(blk.trace[k].reshape(-1).numpy(), g[f"idx_{t}_{bi}_{k}"].reshape(-1).astype(np.int64))
    assert len(set(g["counts"].reshape(-1).tolist())) >= 6, "the fixture counts must vary"


def _ats_cases():
    cs = {}
    for kind in ("Block", "EventfulTokenwiseBlock", "EventfulMatmul1Block", "EventfulBlock"):
        cs[f"{kind}_ats"] = (kind, {})
    cs["EventfulBlock_ats_bf16"] = ("EventfulBlock", dict(matmul_2_cast="bfloat16"))
    cs["EventfulMatmul1Block_ats_bf16"] = ("EventfulMatmul1Block", dict(matmul_2_cast="bfloat16"))
    return cs


@pytest.mark.parametrize("name", list(_ats_cases().keys()))
def test_adaptive_token_sampling_vs_golden(golden_dir, name):
    """ATS (blocks.py:150-181,378-391) as the reference runs it -- batch == heads, scores summed over the batch axis --
    for all four block classes: outputs and the stabilised index sets of the real reference."""
    g = H.load_npz(os.path.join(golden_dir, "ats.npz"))
    kind, kw = _ats_cases()[name]
    params = O.make_block_params(64, 4, seed=int(g[f"{name}__param_seed"]), std=0.08)
    ora = O.BlockOracle(kind, params, 64, 4, (6, 6), ats_fraction=float(g["fraction"]), **kw)
    if kind != "Block":
        ora.set_policy(lambda: O.TopK(int(g["k"])))
    xs, ys = torch.from_numpy(g[f"{name}__x"]), torch.from_numpy(g[f"{name}__y"])
    for t in range(xs.shape[0]):
        y = ora.forward(xs[t].clone())
        assert torch.allclose(y, ys[t], atol=ATOL, rtol=0), (name, t, float((y - ys[t]).abs().max()))
        assert np.array_equal(ora.trace["ats_index"].numpy(), g[f"{name}__ats_index"][t])
