"""Run in a subprocess by test_gpu_big_tiles.py with EVT_GEMM_BIG set (the library reads it once per process):
gated linear launches that the forced 256-row tile covers with edge rows, edge columns, several tiles per persistent
workgroup, gather / scatter, fused gate-reference refresh, GELU and dense mode, checked against fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eventful-transformer_amd"))
from eventful_transformer import _native as n  # noqa: E402

DEV = torch.device("cuda", 0)
CASES = [  # B, N, K, Nout, k, act
    (3, 300, 96, 520, 140, 1),      # edge rows (M = 420) and edge columns, three k-tiles
    (2, 70, 64, 192, 64, 0),        # two k-tiles (the minimum), one row tile
    (64, 197, 768, 2304, 128, 0),   # M = 8192: 288 ... 576 tiles, several per persistent workgroup
    (40, 197, 256, 768, 131, 1),    # M = 5240: ragged last row tile, every workgroup crosses tile boundaries
    (1, 50, 64, 64, 20, 0),         # one ragged tile, fewer columns than the narrowest tile, two k-tiles
    (5, 197, 3072, 768, 128, 0),    # the MLP-2 shape at a small batch: 96 k-tiles per tile
    (130, 197, 96, 1536, 128, 1),   # M = 16640: 520 .. 780 tiles -> two or three tiles per persistent workgroup, three k-tiles each,
                                    # ragged last row tile: the tile descriptions prepared two tiles ahead are all in use
]


def main():
    for B, N, K, Nout, k, act in CASES:
        g = torch.Generator().manual_seed(B * 1000 + N + K + Nout)
        A = torch.randn(B, N, K, generator=g)
        W = torch.randn(Nout, K, generator=g) * 0.05
        bias = torch.randn(Nout, generator=g)
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)]).int()
        buf0 = torch.randn(B, N, Nout, generator=g)
        p0 = torch.randn(B, N, K, generator=g)
        rows = A.gather(1, idx.long().unsqueeze(-1).expand(-1, -1, K))
        y = torch.nn.functional.linear(rows.double(), W.double(), bias.double())
        if act:
            y = torch.nn.functional.gelu(y)
        ref = buf0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, Nout), y.float())
        p_ref = p0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, K), rows)
        Ad, Wd, bd, idxd = (t.to(DEV) for t in (A, W, bias, idx))
        Ws = n.split_weight(Wd)
        assert Ws is not None
        outs = []
        for _ in range(2):
            buf, pd = buf0.to(DEV), p0.to(DEV)
            n.gated_linear(Ad, K, idxd, N, Wd, bd, buf, Nout, idxd, N, None, pd, B, k, K, Nout, act, W_split=Ws)
            outs.append(buf.cpu())
            assert torch.equal(pd.cpu(), p_ref), "gate reference refresh"
        assert torch.equal(outs[0], outs[1]), "reruns bit-identical"
        assert torch.allclose(outs[0], ref, atol=2e-4, rtol=1e-4), float((outs[0] - ref).abs().max())
        mask = torch.ones(B, N, dtype=torch.bool)
        mask.scatter_(1, idx.long(), False)
        assert torch.equal(outs[0][mask], buf0[mask]), "rows outside idx bit-unchanged"
        # the same launch through the 128x128 kernel (EVT_GEMM_BIG only steers the automatic choice; a threshold-policy
        # count list keeps a launch on the 128x128 kernel): the two kernels agree bit for bit
        buf2 = buf0.to(DEV)
        count = torch.full((B,), k, dtype=torch.int32, device=DEV)
        n.gated_linear(Ad, K, idxd, N, Wd, bd, buf2, Nout, idxd, N, count, None, B, k, K, Nout, act, W_split=Ws)
        if n.load().evt_gated_linear_workspace_bytes(B, k, K, Nout, 1) == 0:
            assert torch.equal(buf2.cpu(), outs[0]), "256-row tiles and 128x128 tiles differ"
        else:   # few tiles: the 128x128 kernel runs split-K (partial sums over K slices, another summation order)
            assert torch.allclose(buf2.cpu(), outs[0], atol=2e-5, rtol=2e-5), float((buf2.cpu() - outs[0]).abs().max())
        # dense mode (first frame of a clip)
        out = torch.empty(B * N, Nout, device=DEV)
        n.gated_linear(Ad, K, None, B * N, Wd, bd, out, Nout, None, B * N, None, None, 1, B * N, K, Nout, act, W_split=Ws)
        yd = torch.nn.functional.linear(A.double().reshape(-1, K), W.double(), bias.double())
        if act:
            yd = torch.nn.functional.gelu(yd)
        assert torch.allclose(out.cpu(), yd.float(), atol=2e-4, rtol=1e-4)
    # bf16 activations (the A.v state of a bf16 matmul_2_cast feeding the projection): bit-identical to the same values as fp32
    for B, N, K, Nout, k in [(3, 300, 96, 520, 140), (64, 197, 768, 768, 128)]:
        g = torch.Generator().manual_seed(B + N + K + Nout + 7)
        Ab = (torch.randn(B, N, K, generator=g) * 2).to(torch.bfloat16)
        W = torch.randn(Nout, K, generator=g) * 0.05
        bias = torch.randn(Nout, generator=g)
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)]).int()
        buf0 = torch.randn(B, N, Nout, generator=g)
        p0 = torch.randn(B, N, K, generator=g)
        Wd, bd, idxd = W.to(DEV), bias.to(DEV), idx.to(DEV)
        Ws = n.split_weight(Wd)
        assert n.gated_linear_big_tile(K, True, N, Nout, True, N, False, B, k, K, Nout) != 0
        res = []
        for a_bf16 in (False, True):
            A = Ab.to(DEV) if a_bf16 else Ab.float().to(DEV)
            buf, pd = buf0.to(DEV), p0.to(DEV)
            n.gated_linear(A, K, idxd, N, Wd, bd, buf, Nout, idxd, N, None, pd, B, k, K, Nout, 0, W_split=Ws, a_bf16=a_bf16)
            res.append((buf.cpu(), pd.cpu()))
        assert torch.equal(res[0][0], res[1][0]), "bf16 activations: output differs from the fp32 launch"
        assert torch.equal(res[0][1], res[1][1]), "bf16 activations: gate reference refresh differs"
        rows = Ab.float().gather(1, idx.long().unsqueeze(-1).expand(-1, -1, K))
        y = torch.nn.functional.linear(rows.double(), W.double(), bias.double())
        ref = buf0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, Nout), y.float())
        assert torch.allclose(res[1][0], ref, atol=3e-4, rtol=1e-4), float((res[1][0] - ref).abs().max())
    # gated MLP: under a forced tile both launches run on the 256-row kernel, so the hidden scratch holds pre-split hl32
    # lines (written by the first launch's epilogue, staged without conversion by the second)
    for B, N, D, Dh, k in [(3, 300, 96, 160, 140), (64, 197, 768, 3072, 128)]:
        g = torch.Generator().manual_seed(B + N + D + Dh)
        A = torch.randn(B, N, D, generator=g)
        W1 = torch.randn(Dh, D, generator=g) * 0.05
        b1 = torch.randn(Dh, generator=g) * 0.1
        W2 = torch.randn(D, Dh, generator=g) * 0.05
        b2 = torch.randn(D, generator=g) * 0.1
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)]).int()
        buf0 = torch.randn(B, N, D, generator=g)
        p0 = torch.randn(B, N, D, generator=g)
        rows = A.gather(1, idx.long().unsqueeze(-1).expand(-1, -1, D))
        h = torch.nn.functional.gelu(torch.nn.functional.linear(rows.double(), W1.double(), b1.double()))
        y = torch.nn.functional.linear(h, W2.double(), b2.double())
        ref = buf0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, D), y.float())
        p_ref = p0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, D), rows)
        Ad, W1d, b1d, W2d, b2d, idxd = (t.to(DEV) for t in (A, W1, b1, W2, b2, idx))
        S1, S2 = n.split_weight(W1d), n.split_weight(W2d)
        hidden = torch.empty(B * k, Dh, device=DEV)
        buf, pd = buf0.to(DEV), p0.to(DEV)
        n.gated_mlp(Ad, D, idxd, N, W1d, b1d, W2d, b2d, hidden, buf, D, None, pd, B, k, D, Dh, W1_split=S1, W2_split=S2)
        assert torch.equal(pd.cpu(), p_ref), "MLP gate reference refresh"
        scale = float(y.abs().max())
        err = float((buf.cpu().double() - ref.double()).abs().max()) / scale
        assert err < 3e-5, err
        # the same MLP as two gated-linear launches with an fp32 hidden tensor: bit-identical
        hid2 = torch.empty(B * k, Dh, device=DEV)
        buf2 = buf0.to(DEV)
        n.gated_linear(Ad, D, idxd, N, W1d, b1d, hid2, Dh, None, k, None, None, B, k, D, Dh, n.ACT_GELU, W_split=S1)
        n.gated_linear(hid2, Dh, None, k, W2d, b2d, buf2, D, idxd, N, None, None, B, k, Dh, D, 0, W_split=S2)
        assert torch.equal(buf2.cpu(), buf.cpu()), "pre-split hidden differs from fp32 hidden"
    # block level: with a bf16 A.v cast the projection reads the A.v state (no fp32 attention output is written) whenever
    # its launch runs on the 256-row kernel -- forced here at a small batch.  Two ViViT-sized EventfulBlocks, 3 clips x 4
    # frames: bit-identical to the fp32-output path, and the state path is really taken.
    from eventful_transformer import blocks as EB, policies
    from eventful_transformer.backbones import ViTBackbone
    torch.manual_seed(11)
    bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4, matmul_2_cast="bfloat16"), depth=2,
                     position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
    for p_ in bb.parameters():
        torch.nn.init.normal_(p_, std=0.02)
    bb = bb.eval().to(DEV)
    for m in bb.modules():
        if hasattr(m, "policy"):
            m.policy = policies.TokenNormTopK(k=128)
    g = torch.Generator().manual_seed(5)
    xs = (torch.randn(4, 3, 197, 768, generator=g) * 0.5).to(DEV)
    xs[1:] = xs[0] + 0.02 * torch.randn(3, 3, 197, 768, generator=g).to(DEV)
    taken = []
    orig = n.gated_linear

    def spy(*args, **kw):
        taken.append(bool(kw.get("a_bf16", False)))
        return orig(*args, **kw)

    n.gated_linear = spy
    runs = []
    with torch.inference_mode():
        for on in (True, False):
            EB.PROJ_FROM_STATE = on
            taken.clear()
            bb.reset()
            runs.append(torch.stack([bb(xs[t]).clone() for t in range(xs.shape[0])]))
            assert any(taken) == on, (on, taken)
    n.gated_linear = orig
    assert torch.isfinite(runs[0]).all() and torch.equal(runs[0], runs[1]), "projection from the A.v state differs"
    print("BIG_TILES_OK")


if __name__ == "__main__":
    main()
