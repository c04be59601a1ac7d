"""GPU: each C-ABI entry point of libevt_hip.so against the oracle / plain torch fp32 on CPU."""
import os

import numpy as np
import pytest
import torch

import eventful_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def native():
    from eventful_transformer import _native
    return _native


def test_library_is_loaded_and_targets_gfx950():
    n = native()
    lib = n.load()
    assert lib.evt_version() == n.ABI_VERSION == 9
    assert lib.evt_target_arch() == b"gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


@pytest.mark.parametrize("rows,D", [(197, 768), (5, 64), (33, 1024), (7, 2048), (3, 4096), (1000, 768)])
def test_row_pass_layernorm_residual_norm(rows, D):
    n = native()
    g = torch.Generator().manual_seed(rows * 7 + D)
    x, res, p = (torch.randn(rows, D, generator=g) for _ in range(3))
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    s_ref = x + res
    c_ref = torch.nn.functional.layer_norm(s_ref, (D,), w, b, 1e-6)
    n_ref = torch.linalg.vector_norm(c_ref - p, dim=-1)
    xd, rd, pd, wd, bd = (t.to(DEV) for t in (x, res, p, w, b))
    s_out, c_out = torch.empty_like(xd), torch.empty_like(xd)
    norms = torch.empty(rows, device=DEV)
    n.row_pass(xd, rows, D, res=rd, sum_out=s_out, ln_w=wd, ln_b=bd, eps=1e-6, c_out=c_out, p=pd, norms=norms)
    assert torch.equal(s_out.cpu(), s_ref)  # a single fp32 add: bit-exact
    assert torch.allclose(c_out.cpu(), c_ref, atol=2e-5, rtol=1e-5)
    assert torch.allclose(norms.cpu(), n_ref, rtol=1e-5)
    # broadcast residual (position encoding) and norm-without-reference
    enc = torch.randn(rows // 2 if rows % 2 == 0 else rows, D, generator=g)
    if rows % enc.shape[0] == 0:
        out = torch.empty_like(xd)
        n.row_pass(xd, rows, D, res=enc.to(DEV), res_rows=enc.shape[0], sum_out=out)
        assert torch.equal(out.cpu(), x + enc.repeat(rows // enc.shape[0], 1))
    n.row_pass(xd, rows, D, norms=norms)
    assert torch.allclose(norms.cpu(), torch.linalg.vector_norm(x, dim=-1), rtol=1e-5)


def test_gate_select_golden(golden_dir):
    """Teacher-forced gate parity: the reference's (c, p) -> identical ascending index set."""
    n = native()
    g = H.load_npz(os.path.join(golden_dir, "gates.npz"))
    for i in range(int(g["n_cases"])):
        kind = bytes(g[f"c{i}_kind"]).decode()
        seed, B, N, D, k = (int(g[f"c{i}_{f}"]) for f in ("seed", "B", "N", "D", "k"))
        want = torch.from_numpy(g[f"c{i}_idx"]).int()
        if kind == "topk":
            c, p = O.make_gate_case(seed, B, N, D)
        else:
            c, p, thr = O.make_threshold_case(seed, N, D, k)
        cd, pd = c.to(DEV), p.to(DEV)
        norms = torch.empty(B * N, device=DEV)
        n.row_pass(cd, B * N, D, p=pd, norms=norms)
        if kind == "topk":
            idx = torch.full((B, k), -1, dtype=torch.int32, device=DEV)
            n.select_topk(norms, B, N, k, idx)
            assert torch.equal(idx.cpu(), want), (i, "topk")
            cap, count = k, None
        else:
            idx = torch.full((1, N), -1, dtype=torch.int32, device=DEV)
            count = torch.zeros(1, dtype=torch.int32, device=DEV)
            n.select_threshold(norms, 1, N, thr, N, idx, count)
            assert int(count.item()) == k
            assert torch.equal(idx.cpu()[:, :k], want), (i, "threshold")
            cap = N
        # K2: gather + reference update (I3: p[idx] == c[idx] bitwise, other rows untouched)
        c_t = torch.zeros(B, cap, D, device=DEV)
        e_t = torch.zeros(B, cap, D, device=DEV)
        p_before = pd.clone()
        n.gate_gather_update(cd, pd, idx, count, B, N, D, cap, c_tilde=c_t, e_tilde=e_t, update_p=True)
        wl = want.long()
        ref_ct = c.gather(1, wl.unsqueeze(-1).expand(-1, -1, D))
        assert torch.equal(c_t.cpu()[:, :k], ref_ct)
        assert torch.equal(e_t.cpu()[:, :k], ref_ct - p.gather(1, wl.unsqueeze(-1).expand(-1, -1, D)))
        p_ref = p.clone().scatter_(1, wl.unsqueeze(-1).expand(-1, -1, D), ref_ct)
        assert torch.equal(pd.cpu(), p_ref)
        del p_before


@pytest.mark.parametrize("N,k", [(197, 128), (1764, 256), (4096, 400)])
def test_select_prefetch_rider_changes_nothing(N, k):
    """evt_select_prefetch_next: the next selection launch also reads the armed range(s) with extra workgroups of its grid --
    same lists with one, two or no riders, the rider is consumed by that launch, and a selection that launches nothing drops it."""
    n = native()
    g = torch.Generator(device=DEV).manual_seed(N)
    norms = torch.rand(2, N, device=DEV, generator=g)
    planes = [torch.randn(3 << 20, device=DEV, generator=g), torch.randn((1 << 20) + 12, device=DEV, generator=g)]
    want, rest_w = torch.empty(2, k, dtype=torch.int32, device=DEV), torch.empty(2, N, dtype=torch.int32, device=DEV)
    n.select_topk(norms, 2, N, k, want, rest_w)
    for riders in (planes[:1], planes, []):
        got, rest = torch.full((2, k), -1, dtype=torch.int32, device=DEV), torch.full((2, N), -1, dtype=torch.int32, device=DEV)
        for t in riders:
            n.select_prefetch_next(t)
        n.select_topk(norms, 2, N, k, got, rest)
        assert torch.equal(got, want) and torch.equal(rest[:, :N - k], rest_w[:, :N - k])
    cnt_w, cnt = torch.empty(2, dtype=torch.int32, device=DEV), torch.empty(2, dtype=torch.int32, device=DEV)
    thr_w, thr_g = torch.empty(2, N, dtype=torch.int32, device=DEV), torch.empty(2, N, dtype=torch.int32, device=DEV)
    n.select_threshold(norms, 2, N, 0.7, N, thr_w, cnt_w, None)
    n.select_prefetch_next(planes[0])
    n.select_threshold(norms, 2, N, 0.7, N, thr_g, cnt, None)
    assert torch.equal(cnt, cnt_w) and all(torch.equal(thr_g[b, :int(cnt[b])], thr_w[b, :int(cnt[b])]) for b in range(2))
    n.select_prefetch_next(planes[1])
    n.select_topk(norms, 2, N, 0, got, None)          # k == 0 launches nothing: the rider must not wait for a later launch
    del planes
    n.select_topk(norms, 2, N, k, got, None)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_select_tie_policy_and_edges():
    """Ties are tie-policy-defined (lowest index first), not reference-pinned."""
    n = native()
    norms = torch.tensor([[1, 3, 3, 3, 0, 3, 2, 3], [0, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.float32, device=DEV)
    idx = torch.empty((2, 2), dtype=torch.int32, device=DEV)
    n.select_topk(norms, 2, 8, 2, idx)
    assert idx.cpu().tolist() == [[1, 2], [0, 1]]
    idx = torch.empty((2, 8), dtype=torch.int32, device=DEV)
    n.select_topk(norms, 2, 8, 8, idx)
    assert idx.cpu().tolist() == [list(range(8))] * 2
    # big N, many rounds of ballot compaction; compare with a stable CPU selection
    g = torch.Generator().manual_seed(3)
    big = torch.rand(3, 5000, generator=g)
    big[:, 100:200] = big[:, 300:400]  # force exact duplicates
    for k in (1, 63, 64, 65, 2500, 4999, 5000):
        idx = torch.empty((3, k), dtype=torch.int32, device=DEV)
        rest = torch.full((3, 5000), -1, dtype=torch.int32, device=DEV)
        n.select_topk(big.to(DEV), 3, 5000, k, idx, rest)
        order = np.lexsort((np.arange(5000)[None].repeat(3, 0), -big.numpy()), axis=-1)
        want = np.sort(order[:, :k], axis=-1)
        assert np.array_equal(idx.cpu().numpy(), want), k
        want_rest = np.sort(order[:, k:], axis=-1)   # complement list: unselected tokens, ascending
        assert np.array_equal(rest.cpu().numpy()[:, : 5000 - k], want_rest), k
    cnt = torch.empty(3, dtype=torch.int32, device=DEV)
    idx = torch.empty((3, 5000), dtype=torch.int32, device=DEV)
    n.select_threshold(big.to(DEV), 3, 5000, 0.5, 5000, idx, cnt)
    for b in range(3):
        want = torch.nonzero(big[b] > 0.5).flatten()
        assert int(cnt[b]) == want.numel()
        assert torch.equal(idx[b, : want.numel()].cpu().long(), want)
    with pytest.raises(RuntimeError):
        n.select_topk(norms, 2, 8, 9, idx)  # k > N raises like torch.topk


@pytest.fixture(params=["split", "f32"])
def gemm_mode(request):
    """K3/K7 arithmetic: 'split' (bf16 hi/lo planes, 3 MFMAs per product) and 'f32' (exact fp32 MFMA)."""
    n = native()
    old = n.GEMM_MODE
    n.GEMM_MODE = request.param
    yield request.param
    n.GEMM_MODE = old


@pytest.mark.parametrize("rows,cols", [(96, 64), (50, 72), (7, 200), (33, 768)])
def test_split_weights_planes(rows, cols):
    """evt_split_weights: hi = rne_bf16(w), lo = rne_bf16(w - hi) in the hl32 layout (rows, ceil(cols/32), 2, 32),
    zero-filled past `cols`."""
    n = native()
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(rows, cols, generator=g) * torch.logspace(-6, 3, cols)).to(DEV)
    old, n.GEMM_MODE = n.GEMM_MODE, "split"
    try:
        planes = n.split_weight(W)
    finally:
        n.GEMM_MODE = old
    groups = (cols + 31) // 32
    assert planes.shape == (rows, groups, 2, 32)
    Wp = torch.nn.functional.pad(W, (0, groups * 32 - cols)).view(rows, groups, 32)
    hi = Wp.to(torch.bfloat16)
    lo = (Wp - hi.float()).to(torch.bfloat16)
    assert torch.equal(planes[:, :, 0], hi) and torch.equal(planes[:, :, 1], lo)
    rel = ((planes[:, :, 0].float() + planes[:, :, 1].float() - Wp).abs() / Wp.abs().clamp_min(1e-30)).max()
    assert float(rel) < 2.0 ** -15


def test_split_gemm_error_vs_fp64():
    """The split-precision path is an fp32-accurate matmul: error vs fp64 ~1e-5 of the output scale, same
    order as the fp32 MFMA kernel's own rounding noise, far inside the 1e-3 activation tolerance."""
    n = native()
    g = torch.Generator().manual_seed(7)
    M, K, Nout = 512, 3072, 768
    A = torch.randn(M, K, generator=g)
    W = torch.randn(Nout, K, generator=g) * 0.02
    bias = torch.zeros(Nout)
    ref = A.double() @ W.double().T
    errs = {}
    for mode in ("f32", "split"):
        old, n.GEMM_MODE = n.GEMM_MODE, mode
        try:
            out = torch.empty(M, Nout, device=DEV)
            Wd = W.to(DEV)
            n.gated_linear(A.to(DEV), K, None, M, Wd, bias.to(DEV), out, Nout, None, M, None, None, 1, M, K, Nout,
                           W_split=n.split_weight(Wd))
        finally:
            n.GEMM_MODE = old
        errs[mode] = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert errs["f32"] < 2e-6, errs
    assert errs["split"] < 3e-5, errs


@pytest.mark.parametrize("B,N,K,Nout,k,act", [(2, 197, 768, 2304, 128, 0), (1, 37, 64, 192, 12, 0),
                                              (3, 50, 256, 64, 50, 1), (1, 300, 768, 3072, 131, 1),
                                              (2, 64, 3072, 768, 64, 0), (1, 50, 72, 136, 20, 0), (5, 300, 200, 520, 140, 1),
                                              # the benchmark's operating point: M = B*k = 16384 rows, >= 128 output tiles =>
                                              # no split-K, bias / GELU / scatter in the GEMM epilogue, XCD tile map, gathered
                                              # A rows + fused p refresh
                                              (128, 197, 768, 2304, 128, 0), (128, 197, 768, 3072, 128, 1),
                                              (128, 197, 3072, 768, 128, 0), (128, 197, 768, 768, 128, 0)])
def test_gated_linear(B, N, K, Nout, k, act, gemm_mode):
    n = native()
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
    Ad, Wd, bd, idxd, buf, pd = (t.to(DEV) for t in (A, W, bias, idx, buf0, p0))
    Ws = n.split_weight(Wd)
    assert (Ws is not None) == (gemm_mode == "split")
    n.gated_linear(Ad, K, idxd, N, Wd, bd, buf, Nout, idxd, N, None, pd, B, k, K, Nout, act, W_split=Ws)
    assert torch.allclose(buf.cpu(), ref, atol=2e-4, rtol=1e-4), float((buf.cpu() - ref).abs().max())
    mask = torch.ones(B, N, dtype=torch.bool)
    mask.scatter_(1, idx.long(), False)
    assert torch.equal(buf.cpu()[mask], buf0[mask])  # I2: rows outside idx bit-unchanged
    assert torch.equal(pd.cpu(), p_ref)              # fused K2
    # variable count per clip (threshold policy): rows beyond count untouched
    count = torch.tensor([k // 2] + [k] * (B - 1), dtype=torch.int32)
    buf2 = buf0.to(DEV)
    n.gated_linear(Ad, K, idxd, N, Wd, bd, buf2, Nout, idxd, N, count.to(DEV), None, B, k, K, Nout, act, W_split=Ws)
    ref2 = buf0.clone()
    for b in range(B):
        sel = idx[b, : int(count[b])].long()
        ref2[b, sel] = ref[b, sel]
    assert torch.allclose(buf2.cpu(), ref2, atol=2e-4, rtol=1e-4)
    # dense mode (no index lists), as used on the first frame of a clip
    out = torch.empty(B * N, Nout, device=DEV)
    n.gated_linear(Ad, K, None, B * N, Wd, bd, out, Nout, None, B * N, None, None, 1, B * N, K, Nout, act, W_split=Ws)
    yd = torch.nn.functional.linear(A.double().reshape(-1, K), W.double(), bias.double())
    if act:
        yd = torch.nn.functional.gelu(yd)
    assert torch.allclose(out.cpu(), yd.float(), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("M,K,Nout,act", [(256, 3072, 768, 0), (256, 768, 3072, 1), (196, 768, 2304, 0), (50, 1024, 1024, 0),
                                          (409, 768, 2304, 0), (409, 768, 3072, 1), (1000, 3072, 768, 0)])
def test_splitk_small_launch(M, K, Nout, act):
    """Few-tile launches (one video stream: the small-row-count kernel with its in-CU K split, or split-K over workgroups with
    fixed-order partial sums): same accuracy, bit-reproducible, scatter / count / fused p refresh unchanged.  M = 409 and
    1000: (column, row) tile counts that do not divide by the 8 XCD runs of the tile order."""
    n = native()
    if n.GEMM_MODE != "split":
        pytest.skip("split-K belongs to the split-precision kernel")
    assert n.load().evt_gated_linear_workspace_bytes(1, M, K, Nout, 0) > 0
    g = torch.Generator().manual_seed(M + K + Nout)
    N = M + 60
    A = torch.randn(1, N, K, generator=g)
    W = torch.randn(Nout, K, generator=g) * 0.03
    bias = torch.randn(Nout, generator=g)
    idx = torch.randperm(N, generator=g)[:M].sort()[0].int().unsqueeze(0)
    count = torch.tensor([M - 7], dtype=torch.int32)
    buf0 = torch.randn(1, N, Nout, generator=g)
    p0 = torch.randn(1, N, K, generator=g)
    sel = idx[0, : M - 7].long()
    y = torch.nn.functional.linear(A[0, sel].double(), W.double(), bias.double())
    if act:
        y = torch.nn.functional.gelu(y)
    ref = buf0.clone()
    ref[0, sel] = y.float()
    p_ref = p0.clone()
    p_ref[0, sel] = A[0, sel]
    Ad, Wd, bd, idxd, cd = (t.to(DEV) for t in (A, W, bias, idx, count))
    Ws = n.split_weight(Wd)
    outs = []
    for _ in range(2):
        buf, pd = buf0.to(DEV), p0.to(DEV)
        n.gated_linear(Ad, K, idxd, N, Wd, bd, buf, Nout, idxd, N, cd, pd, 1, M, K, Nout, act, W_split=Ws)
        outs.append(buf.cpu())
        assert torch.equal(pd.cpu(), p_ref)
    assert torch.equal(outs[0], outs[1])  # I7: reruns bit-identical
    assert torch.allclose(outs[0], ref, atol=2e-4, rtol=1e-4), float((outs[0] - ref).abs().max())
    err = (outs[0][0, sel].double() - y).abs().max() / y.abs().max()
    assert err < 2e-5, float(err)


@pytest.mark.parametrize("counts", [[100], [1000], [0], [130, 5, 1024], [409], [1]])
def test_splitk_dynamic_counts(counts):
    """Threshold-policy launches (kcap = N, per-clip counts on the device): the split-K factor is picked by the
    workgroups from the counts -- few live tiles -> split + finish kernel, many -> single pass under the same
    launch, none -> nothing written."""
    n = native()
    if n.GEMM_MODE != "split":
        pytest.skip("split-K belongs to the split-precision kernel")
    B, N, K, Nout = len(counts), 1024, 768, 2304
    assert n.load().evt_gated_linear_workspace_bytes(B, N, K, Nout, 1) > 0
    g = torch.Generator().manual_seed(sum(counts) + B)
    A = torch.randn(B, N, K, generator=g)
    W = torch.randn(Nout, K, generator=g) * 0.03
    bias = torch.randn(Nout, generator=g)
    idx = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).int()
    buf0 = torch.randn(B, N, Nout, generator=g)
    ref = buf0.clone()
    for b, c in enumerate(counts):
        sel = idx[b, :c].long()
        ref[b, sel] = torch.nn.functional.linear(A[b, sel].double(), W.double(), bias.double()).float()
    Ad, Wd, bd, idxd = (t.to(DEV) for t in (A, W, bias, idx))
    cd = torch.tensor(counts, dtype=torch.int32, device=DEV)
    Ws = n.split_weight(Wd)
    outs = []
    for _ in range(2):
        buf = buf0.to(DEV)
        n.gated_linear(Ad, K, idxd, N, Wd, bd, buf, Nout, idxd, N, cd, None, B, N, K, Nout, 0, W_split=Ws)
        outs.append(buf.cpu())
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(outs[0], ref, atol=2e-4, rtol=1e-4), float((outs[0] - ref).abs().max())
    for b, c in enumerate(counts):   # rows past count[b] bit-unchanged
        dead = idx[b, c:].long()
        assert torch.equal(outs[0][b, dead], buf0[b, dead])


@pytest.mark.parametrize("B,N,D,Dh,k", [(2, 40, 64, 256, 9), (128, 197, 768, 3072, 128)])
def test_gated_mlp_matches_two_linears(B, N, D, Dh, k, gemm_mode):
    """K7 (linear -> exact-erf GELU -> linear on the gated rows, scatter into the token buffer, fused refresh of the
    gate reference) against fp64; the second case is the benchmark's operating point (M = 16384 rows)."""
    n = native()
    g = torch.Generator().manual_seed(11)
    s1, s2 = (0.1, 0.1) if D == 64 else (0.03, 0.02)
    A, W1, b1 = torch.randn(B, N, D, generator=g), torch.randn(Dh, D, generator=g) * s1, torch.randn(Dh, generator=g)
    W2, b2 = torch.randn(D, Dh, generator=g) * s2, torch.randn(D, generator=g)
    idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)]).int()
    buf0 = torch.randn(B, N, D, generator=g)
    p0 = torch.randn(B, N, D, generator=g)
    rows32 = A.gather(1, idx.long().unsqueeze(-1).expand(-1, -1, D))
    rows = rows32.double()
    y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(rows, W1.double(), b1.double())),
                                   W2.double(), b2.double()).float()
    ref = buf0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, D), y)
    p_ref = p0.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, D), rows32)
    buf, pd = buf0.to(DEV), p0.to(DEV)
    hidden = torch.empty(B * k, Dh, device=DEV)
    W1d, W2d = W1.to(DEV), W2.to(DEV)
    n.gated_mlp(A.to(DEV), D, idx.to(DEV), N, W1d, b1.to(DEV), W2d, b2.to(DEV), hidden, buf, D, None,
                pd, B, k, D, Dh, W1_split=n.split_weight(W1d), W2_split=n.split_weight(W2d))
    assert torch.allclose(buf.cpu(), ref, atol=3e-4, rtol=1e-4), float((buf.cpu() - ref).abs().max())
    assert torch.equal(pd.cpu(), p_ref)
    mask = torch.ones(B, N, dtype=torch.bool)
    mask.scatter_(1, idx.long(), False)
    assert torch.equal(buf.cpu()[mask], buf0[mask])


@pytest.mark.parametrize("B,H,N,dh,k", [(2, 12, 197, 64, 128), (1, 4, 37, 16, 12), (1, 2, 130, 32, 1)])
def test_qk_state_full_and_delta(B, H, N, dh, k):
    """I1: after every update the product state equals (q/scale) k^T of the CURRENT buffer."""
    n = native()
    D = H * dh
    g = torch.Generator().manual_seed(N + k)
    buf = torch.randn(B, N, 3 * D, generator=g)
    scale = float(np.sqrt(dh))

    def ref(bf):
        q, kk, _ = bf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        return (q / scale) @ kk.transpose(-2, -1)

    bd = buf.to(DEV)
    prod_d = torch.empty(B, H, N, N, device=DEV)
    n.qk_packed(bd, B, N, D, H, scale, prod_d)
    assert torch.allclose(prod_d.cpu(), ref(buf), atol=1e-4, rtol=1e-4)
    idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)]).int()
    new_rows = torch.randn(B, k, 3 * D, generator=g)
    buf2 = buf.clone().scatter_(1, idx.long().unsqueeze(-1).expand(-1, -1, 3 * D), new_rows)
    before = prod_d.clone()
    n.qk_packed(buf2.to(DEV), B, N, D, H, scale, prod_d, idx=idx.to(DEV), kcap=k)
    assert torch.allclose(prod_d.cpu(), ref(buf2), atol=1e-4, rtol=1e-4)
    keep = torch.ones(B, N, dtype=torch.bool)
    keep.scatter_(1, idx.long(), False)
    for b in range(B):
        sub_new = prod_d[b][:, keep[b]][:, :, keep[b]].cpu()
        sub_old = before[b][:, keep[b]][:, :, keep[b]].cpu()
        assert torch.equal(sub_new, sub_old)  # entries outside rows/cols idx are bit-unchanged


@pytest.mark.parametrize("cast", [None, "bfloat16", "float16"])
def test_attention_value_path_matches_oracle(cast):
    """K5 + K6a + K6 against the oracle's delta gates / accumulator on identical inputs, 3 frames."""
    n = native()
    B, H, N, dh, k = 2, 4, 37, 16, 12
    D = H * dh
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(5)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    ap = torch.empty(B, H, N, N, dtype=sdt, device=DEV)
    vp = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    tol = 1e-5 if cast is None else (2e-2 if cast == "bfloat16" else 3e-3)
    for t in range(3):
        scores = torch.randn(B, H, N, N, generator=g) * 2
        buf = torch.randn(B, N, 3 * D, generator=g)
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        # oracle (blocks.py:558-575)
        a = scores.softmax(dim=-1)
        v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)[2]
        if cast is not None:
            a, v = a.to(sdt), v.to(sdt)
        else:
            v = v.clone()
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        # HIP
        sd, bd, idxd = scores.to(DEV), buf.to(DEV), idx.int().to(DEV)
        if t == 0:
            n.softmax_gate(sd, ap, B, H, N, N, D, store)
            n.v_gate(bd, None, None, B, N, D, 0, vp, None, None, store, False)
            n.av(ap, vp, N, B, H, N, N, D, store, pv=pv, out_f32=out)
        else:
            a_new = torch.empty(B, H, N, k, dtype=sdt, device=DEV)
            a_del = torch.empty_like(a_new)
            v_del = torch.empty(B, k, D, dtype=sdt, device=DEV)
            v_old = torch.empty_like(v_del)
            n.softmax_gate(sd, ap, B, H, N, N, D, store, a_new=a_new, a_delta=a_del, idx=idxd, kcap=k, gated=True)
            n.v_gate(bd, idxd, None, B, N, D, k, vp, v_del, v_old, store, True)
            n.av(a_new, v_del, k, B, H, N, k, D, store, pv=pv, out_f32=out, a2=a_del, v2=v_old, gated=True)
            assert torch.allclose(a_new.float().cpu(), a_n.float(), atol=tol * 0.1 + 1e-6)
            assert torch.allclose(v_del.float().cpu().view(B, k, H, dh).permute(0, 2, 1, 3), v_d.float(), atol=1e-6)
        assert torch.allclose(ap.float().cpu(), ag.t.float(), atol=tol * 0.1 + 1e-6)
        assert torch.allclose(out.cpu(), ref, atol=tol, rtol=0), (cast, t, float((out.cpu() - ref).abs().max()))
        assert torch.equal(out.cpu(), pv.float().cpu())


def test_rel_pos_softmax_matches_oracle():
    n = native()
    B, H, gh, gw, dh = 2, 4, 6, 5, 16
    N, D = gh * gw, H * dh
    g = torch.Generator().manual_seed(8)
    scores = torch.randn(B, H, N, N, generator=g)
    buf = torch.randn(B, N, 3 * D, generator=g)
    ry, rx = torch.randn(gh, gh, dh, generator=g), torch.randn(gw, gw, dh, generator=g)
    q = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)[0]
    ref = O.add_relative(scores.clone(), q, ry, rx, (gh, gw), inplace=False).softmax(dim=-1)
    ap = torch.empty(B, H, N, N, device=DEV)
    n.softmax_gate(scores.to(DEV), ap, B, H, N, N, D, 0, qkv=buf.to(DEV), rel_y=ry.to(DEV), rel_x=rx.to(DEV), gh=gh,
                   gw=gw)
    assert torch.allclose(ap.cpu(), ref, atol=2e-6, rtol=1e-4), float((ap.cpu() - ref).abs().max())


@pytest.mark.parametrize("cast,N,k,rel", [(None, 70, 12, False), ("bfloat16", 70, 40, False), ("float16", 36, 20, True),
                                          ("bfloat16", 300, 200, False), (None, 130, 130, False),
                                          ("bfloat16", 42, 17, True)])
def test_fused_softmax_av_gated_matches_oracle(cast, N, k, rel):
    """K6a(transposed) + fused K5/K6 (evt_softmax_av_gated) vs the oracle's gates/accumulator, dh = 64,
    incl. multi-chunk k > 128, partial row blocks, rel-pos terms and a device-side count < kcap."""
    n = native()
    B, H, dh = 2, 2, 64
    D = H * dh
    gh, gw = (6, N // 6) if rel else (0, 0)
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(N * 31 + k)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    ap = torch.empty(B, H, N, N, dtype=sdt, device=DEV)
    vp = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    ry = torch.randn(gh, gh, dh, generator=g) * 0.2 if rel else None
    rx = torch.randn(gw, gw, dh, generator=g) * 0.2 if rel else None
    tol = 2e-5 if cast is None else (2e-2 if cast == "bfloat16" else 3e-3)
    for t in range(3):
        scores = torch.randn(B, H, N, N, generator=g) * 2
        buf = torch.randn(B, N, 3 * D, generator=g)
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        q, _, v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        logits = O.add_relative(scores.clone(), q, ry, rx, (gh, gw), inplace=False) if rel else scores
        a = logits.softmax(dim=-1)
        if cast is not None:
            a, v = a.to(sdt), v.to(sdt)
        else:
            v = v.clone()
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        sd, bd, idxd = scores.to(DEV), buf.to(DEV), idx.int().to(DEV)
        relkw = dict(qkv=bd, rel_y=ry.to(DEV), rel_x=rx.to(DEV), gh=gh, gw=gw) if rel else {}
        if t == 0:
            n.softmax_gate(sd, ap, B, H, N, N, D, store, **relkw)
            n.v_gate(bd, None, None, B, N, D, 0, vp, None, None, store, False)
            n.av(ap, vp, N, B, H, N, N, D, store, pv=pv, out_f32=out)
        else:
            # frame 2 exercises the device-side count: capacity k + 5, valid k
            cap = k if t == 1 else k + 5
            idx_cap = torch.full((B, cap), 0, dtype=torch.int32, device=DEV)
            idx_cap[:, :k] = idxd
            count = None if t == 1 else torch.full((B,), k, dtype=torch.int32, device=DEV)
            v_del = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            v_old = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            n.v_gate(bd, idx_cap, count, B, N, D, cap, vp, v_del, v_old, store, True, transposed=True)
            n.softmax_av_gated(sd, ap, idx_cap, count, cap, v_del, v_old, pv, out, B, H, N, D, store, **relkw)
        assert torch.allclose(ap.float().cpu(), ag.t.float(), atol=tol * 0.1 + 2e-6), (cast, t)
        err = float((out.cpu() - ref).abs().max())
        assert err <= tol, (cast, N, k, t, err)
        assert torch.equal(out.cpu(), pv.float().cpu())


@pytest.mark.parametrize("cast,N,k,rel", [(None, 70, 12, False), ("bfloat16", 197, 128, False), ("float16", 36, 20, True),
                                          (None, 256, 100, False), ("bfloat16", 42, 17, True), (None, 197, 197, False)])
@pytest.mark.parametrize("qk_split", [0, 1])
def test_fused_attention_in_kernel_qk(cast, N, k, rel, qk_split):
    """evt_softmax_av_gated with product == NULL: the score rows are (q / scale) k^T computed INSIDE the kernel from the
    token buffer (no q.k^T state, no K4) -- against the oracle's softmax / delta gates / accumulator on the same
    buffer, 3 gated frames incl. a device-side count < kcap, rel-pos, partial row blocks and N = 256."""
    n = native()
    B, H, dh, scale = 2, 2, 64, 8.0
    D = H * dh
    gh, gw = (6, N // 6) if rel else (0, 0)
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(N * 17 + k)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    ap = torch.empty(B, H, N, N, dtype=sdt, device=DEV)
    vp = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    ry = torch.randn(gh, gh, dh, generator=g) * 0.2 if rel else None
    rx = torch.randn(gw, gw, dh, generator=g) * 0.2 if rel else None
    # qk_split = 1: q, k as bf16 hi + lo (~1e-5 relative on the logits, the arithmetic of evt_qk's split mode)
    tol = (3e-5 if not qk_split else 3e-4) if cast is None else (2e-2 if cast == "bfloat16" else 3e-3)
    for t in range(4):
        buf = torch.randn(B, N, 3 * D, generator=g) * 1.5
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        q, kk, v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        scores = (q / scale) @ kk.transpose(-2, -1)
        logits = O.add_relative(scores.clone(), q, ry, rx, (gh, gw), inplace=False) if rel else scores
        a = logits.softmax(dim=-1)
        if cast is not None:
            a, v = a.to(sdt), v.to(sdt)
        else:
            v = v.clone()
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        bd, idxd = buf.to(DEV), idx.int().to(DEV)
        relkw = dict(rel_y=ry.to(DEV), rel_x=rx.to(DEV), gh=gh, gw=gw) if rel else {}
        if t == 0:
            n.attention_dense(bd, B, H, N, D, scale, store, out_f32=out, a_state=ap, pv=pv, qw=gw, qk_split=qk_split, **relkw)
            n.v_gate(bd, None, None, B, N, D, 0, vp, None, None, store, False)
        else:
            cap = k if (t != 2 or k == N) else k + 5
            idx_cap = torch.full((B, cap), 0, dtype=torch.int32, device=DEV)
            idx_cap[:, :k] = idxd
            count = None if cap == k else torch.full((B,), k, dtype=torch.int32, device=DEV)
            v_del = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            v_old = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            n.v_gate(bd, idx_cap, count, B, N, D, cap, vp, v_del, v_old, store, True, transposed=True)
            ref_next = torch.randn(B, N, D, generator=g).to(DEV)      # stands for the projection gate's reference
            parts = torch.full((B, N, H), float("nan"), device=DEV)
            n.softmax_av_gated(None, ap, idx_cap, count, cap, v_del, v_old, pv, out, B, H, N, D, store, qkv=bd, scale=scale,
                               qk_split=qk_split, norm_ref=ref_next, norm_parts=parts, **relkw)
            # fused delta norm: per-head ||out - ref||^2, and the selection from those partials == selection from the norms
            want_parts = (out - ref_next).view(B, N, H, dh).pow(2).sum(-1)
            assert torch.allclose(parts, want_parts, rtol=1e-5, atol=1e-6), float((parts - want_parts).abs().max())
            kk_sel = max(1, N // 3)
            i1 = torch.empty(B, kk_sel, dtype=torch.int32, device=DEV)
            i2 = torch.empty(B, kk_sel, dtype=torch.int32, device=DEV)
            n.select_topk(parts.sum(-1).sqrt().contiguous(), B, N, kk_sel, i1)
            n.select_topk(parts, B, N, kk_sel, i2, parts=H)
            assert torch.equal(i1, i2)
        # probabilities are rounded to the store type from scores computed in a different fp32 summation order than the
        # CPU's: allow one ulp of the store type at p <= 1 (bf16 2^-8, fp16 2^-11)
        atol_p = {None: tol * 0.1 + 3e-6, "bfloat16": 4e-3, "float16": 5e-4}[cast]
        assert torch.allclose(ap.float().cpu(), ag.t.float(), atol=atol_p), (cast, t, float((ap.float().cpu() - ag.t.float()).abs().max()))
        err = float((out.cpu() - ref).abs().max())
        # cast modes: the output IS the store-type state, so a flipped rounding shows as one ulp of the store type at
        # the output's magnitude (bf16 2^-8, fp16 2^-11 relative)
        # (the store-type state PERSISTS: a rounding flipped in an earlier frame stays, so from the first gated frame on an element may be
        # off by two ulps -- scripts/probes/random_attention_probe.py met 1.5 ulps of the top binade at t = 2 on random shapes)
        bar = tol if cast is None else max(tol, (1 if t == 0 else 2) * float(ref.abs().max()) * (2.0 ** -7 if cast == "bfloat16" else 2.0 ** -10))
        assert err <= bar, (cast, N, k, t, err, bar)
        assert torch.equal(out.cpu(), pv.float().cpu())
    with pytest.raises(RuntimeError, match="in-kernel"):
        n.softmax_av_gated(None, torch.empty(1, 2, 300, 300, device=DEV), idxd[:1], None, k, v_del, v_old,
                           torch.empty(1, 300, D, device=DEV), torch.empty(1, 300, D, device=DEV), 1, H, 300, D, 0, qkv=bd, scale=scale)


@pytest.mark.parametrize("cast,N,rel", [(None, 197, False), ("bfloat16", 197, False), (None, 196, True), ("float16", 100, True),
                                        (None, 256, True), (None, 33, False)])
@pytest.mark.parametrize("qk_split", [0, 1])
def test_attention_dense_fused_vs_fp64(cast, N, rel, qk_split):
    """K8 (evt_attention_dense): one launch for q.k^T + rel-pos + softmax + A.v against an fp64 restatement of
    blocks.py:205-240 with the reference's rounding points, plus its state outputs (first frame of a clip)."""
    n = native()
    B, H, dh = 2, 3, 64
    D = H * dh
    g = torch.Generator().manual_seed(N + (7 if rel else 0))
    qkv = torch.randn(B, N, 3 * D, generator=g)
    scale = float(np.sqrt(dh))
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    side = int(round(N ** 0.5))
    ry = rx = None
    if rel:
        assert side * side == N
        ry = torch.randn(side, side, dh, generator=g) * 0.2
        rx = torch.randn(side, side, dh, generator=g) * 0.2
    q, k, v = qkv.double().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    s = ((q.float() / scale).double()) @ k.transpose(-2, -1)
    s32 = s.clone()
    if rel:
        qg = q.reshape(B, H, side, side, dh)
        ty = torch.einsum("bhyxd,ykd->bhyxk", qg, ry.double())
        tx = torch.einsum("bhyxd,xkd->bhyxk", qg, rx.double())
        s = (s.view(B, H, side, side, side, side) + ty[..., :, None] + tx[..., None, :]).reshape(B, H, N, N)
    p = torch.softmax(s, dim=-1).to(sdt)
    vv = v.float().to(sdt)
    want = (p.double() @ vv.double()).to(sdt).float().permute(0, 2, 1, 3).reshape(B, N, D)
    out = torch.empty(B, N, D, device=DEV)
    product = torch.empty(B, H, N, N, device=DEV)
    a_state = torch.empty(B, H, N, N, dtype=sdt, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    kw = dict(rel_y=ry.to(DEV), rel_x=rx.to(DEV), gh=side, gw=side, qw=side) if rel else {}
    n.attention_dense(qkv.to(DEV), B, H, N, D, scale, store, out_f32=out, product=product, a_state=a_state, pv=pv, qk_split=qk_split, **kw)
    # qk_split = 1: q, k and the rel-pos tables as bf16 hi + lo (~1e-5 relative on the logits, the arithmetic of evt_qk's split mode)
    tol = {None: 2e-5 if not qk_split else 2e-4, "bfloat16": 1.6e-2, "float16": 2e-3}[cast]   # one rounding step of the store type on O(1) values
    assert torch.allclose(product.cpu().double(), s32, atol=1e-4 if not qk_split else 5e-4, rtol=1e-5)
    assert float((a_state.cpu().double() - p.double()).abs().max()) <= tol / 4
    assert float((out.cpu() - want).abs().max()) <= tol, float((out.cpu() - want).abs().max())
    assert torch.equal(pv.cpu().float(), out.cpu())
    # the unfused chain (K4 + K5 + K6a + K6) on the same buffer: same rounding points
    prod2 = torch.empty(B, H, N, N, device=DEV)
    a2 = torch.empty(B, H, N, N, dtype=sdt, device=DEV)
    v2 = torch.empty(B, N, D, dtype=sdt, device=DEV)
    out2 = torch.empty(B, N, D, device=DEV)
    qd = qkv.to(DEV)
    n.qk_packed(qd, B, N, D, H, scale, prod2)
    n.softmax_gate(prod2, a2, B, H, N, N, D, store, qkv=qd, **kw)
    n.v_gate(qd, None, None, B, N, D, 0, v2, None, None, store, False)
    n.av(a2, v2, N, B, H, N, N, D, store, out_f32=out2)
    assert float((out.cpu() - out2.cpu()).abs().max()) <= tol


@pytest.mark.parametrize("cast,N,rel", [(None, 197, False), ("bfloat16", 197, False), (None, 196, True), ("float16", 100, True),
                                        (None, 256, False), (None, 33, False), ("bfloat16", 196, True), (None, 144, True)])
@pytest.mark.parametrize("qk_split", [0, 1])
@pytest.mark.parametrize("B,H", [(2, 3), (25, 6)])
def test_attention_dense_resident_vs_fp64(cast, N, rel, qk_split, B, H):
    """K8, resident form (evt_attn_window.hip: the launches of evt_attention_dense WITHOUT state outputs -- ViTDet's windowed
    blocks on every frame, dense `Block`s): one workgroup per (group, head) with K / V staged once, scores and probabilities
    in registers.  Against an fp64 restatement of blocks.py:205-240 with the reference's rounding points and against the tiled
    kernel (same call with a state output), for both workgroup shapes: 6 (group, head) pairs -> two 4-wave workgroups per
    pair; 150 pairs -> one 8-wave workgroup."""
    n = native()
    dh = 64
    D = H * dh
    g = torch.Generator().manual_seed(N + (7 if rel else 0) + B)
    qkv = torch.randn(B, N, 3 * D, generator=g)
    scale = float(np.sqrt(dh))
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    side = int(round(N ** 0.5))
    ry = rx = None
    if rel:
        assert side * side == N
        ry = torch.randn(side, side, dh, generator=g) * 0.2
        rx = torch.randn(side, side, dh, generator=g) * 0.2
    q, k, v = qkv.double().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    s = ((q.float() / scale).double()) @ k.transpose(-2, -1)
    if rel:
        qg = q.reshape(B, H, side, side, dh)
        ty = torch.einsum("bhyxd,ykd->bhyxk", qg, ry.double())
        tx = torch.einsum("bhyxd,xkd->bhyxk", qg, rx.double())
        s = (s.view(B, H, side, side, side, side) + ty[..., :, None] + tx[..., None, :]).reshape(B, H, N, N)
    p = torch.softmax(s, dim=-1).to(sdt)
    vv = v.float().to(sdt)
    want = (p.double() @ vv.double()).to(sdt).float().permute(0, 2, 1, 3).reshape(B, N, D)
    kw = dict(rel_y=ry.to(DEV), rel_x=rx.to(DEV), gh=side, gw=side, qw=side) if rel else {}
    qd = qkv.to(DEV)
    out = torch.full((B, N, D), float("nan"), device=DEV)
    assert n.attention_dense_resident(N, D, H, store, side if rel else 0, side if rel else 0, qk_split=qk_split)
    ref_next = torch.randn(B, N, D, generator=g).to(DEV)      # stands for the projection gate's reference
    parts = torch.full((B, N, H), float("nan"), device=DEV)
    n.attention_dense(qd, B, H, N, D, scale, store, out_f32=out, qk_split=qk_split, norm_ref=ref_next, norm_parts=parts, **kw)
    # fused delta norm: per-head ||out - ref||^2, and the selection from those partials == the selection from the norms
    want_parts = (out - ref_next).view(B, N, H, dh).pow(2).sum(-1)
    assert torch.allclose(parts, want_parts, rtol=1e-5, atol=1e-6), float((parts - want_parts).abs().max())
    kk_sel = max(1, N // 3)
    i1 = torch.empty(B, kk_sel, dtype=torch.int32, device=DEV)
    i2 = torch.empty(B, kk_sel, dtype=torch.int32, device=DEV)
    n.select_topk(want_parts.sum(-1).sqrt().contiguous(), B, N, kk_sel, i1)
    n.select_topk(parts, B, N, kk_sel, i2, parts=H)
    assert torch.equal(i1, i2) or float((parts - want_parts).abs().max()) > 0
    # split arithmetic with an fp32 store type also runs P.V as bf16 hi / lo products (~1e-5 relative)
    tol = {None: 2e-5 if not qk_split else 2e-4, "bfloat16": 1.6e-2, "float16": 2e-3}[cast]
    err = float((out.cpu() - want).abs().max())
    assert err <= tol, (cast, N, rel, qk_split, B, err)
    out_t = torch.empty(B, N, D, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    n.attention_dense(qd, B, H, N, D, scale, store, out_f32=out_t, pv=pv, qk_split=qk_split, **kw)   # a state output: tiled kernel
    assert float((out - out_t).abs().max()) <= tol
    with pytest.raises(RuntimeError, match="resident kernel only"):
        n.attention_dense(qd, B, H, N, D, scale, store, out_f32=out_t, pv=pv, qk_split=qk_split, norm_ref=ref_next, norm_parts=parts, **kw)


def test_attention_dense_windowed_block_matches_chain():
    """Windowed dense attention (14x14 windows on a padded 20x20 grid, rel-pos): the K8 launch against the
    K4+K5+K6 chain through the same Block, padding rows dropped on un-windowing."""
    from eventful_transformer import _native, blocks
    torch.manual_seed(5)
    blk = blocks.Block(dim=128, heads=2, input_size=(20, 20), mlp_ratio=2, window_size=(14, 14), relative_embedding_size=(14, 14))
    for p_ in blk.parameters():
        torch.nn.init.normal_(p_, std=0.08)
    blk = blk.eval().to(DEV)
    x = torch.randn(2, 400, 128, device=DEV)
    outs = []
    try:
        for fused in (True, False):
            _native.DENSE_FUSED = fused
            blk.reset()
            with torch.inference_mode():
                outs.append(blk(x).cpu())
    finally:
        _native.DENSE_FUSED = True
    assert torch.isfinite(outs[0]).all()
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5 * float(outs[1].abs().max() + 1)


def test_attention_dense_argument_errors():
    n = native()
    qkv = torch.zeros(1, 300, 3 * 64, device=DEV)
    out = torch.zeros(1, 300, 64, device=DEV)
    with pytest.raises(RuntimeError, match="at most 256"):
        n.attention_dense(qkv, 1, 1, 300, 64, 8.0, n.store_code(torch.float32), out_f32=out)
    with pytest.raises(RuntimeError, match="head dim must be 64"):
        n.attention_dense(qkv, 1, 2, 100, 64, 8.0, n.store_code(torch.float32), out_f32=out)


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("B,H,qh,qw,gh,gw", [(1, 12, 42, 42, 42, 42), (2, 2, 6, 7, 6, 7), (1, 3, 8, 6, 4, 3), (1, 2, 64, 64, 64, 64),
                                             (1, 2, 70, 3, 70, 3)])
def test_rel_terms(B, H, qh, qw, gh, gw, split):
    """evt_rel_terms against the reference's two einsums (utils.py:159-168), incl. a pooled key grid (gh x gw != qh x qw) and
    more than 64 rows per grid column; fp32 FMA chains (split = 0) and bf16 hi/lo MFMA products (split = 1, ~1e-5 relative)."""
    n = native()
    dh, N = 64, qh * qw
    D = H * dh
    g = torch.Generator().manual_seed(qh * 100 + qw)
    qkv = torch.randn(B, N, 3 * D, generator=g)
    ry = torch.randn(qh, gh, dh, generator=g) * 0.3
    rx = torch.randn(qw, gw, dh, generator=g) * 0.3
    q = qkv[..., :D].view(B, qh, qw, H, dh).double()
    want_y = torch.einsum("byxhc,ykc->bhyxk", q, ry.double())
    want_x = torch.einsum("byxhc,xkc->bhyxk", q, rx.double())
    want = torch.cat([want_y, want_x], dim=-1).reshape(B, H, N, gh + gw)
    out = torch.full((B, H, N, gh + gw), float("nan"), device=DEV)
    n.rel_terms(qkv.to(DEV), ry.to(DEV), rx.to(DEV), B, H, N, D, gh, gw, qw, out, split=split)
    atol = 1e-4 if split else 2e-5
    assert torch.allclose(out.cpu().double(), want, atol=atol, rtol=1e-5), float((out.cpu().double() - want).abs().max())


@pytest.mark.parametrize("cast,N,k", [(None, 70, 12), ("bfloat16", 42, 17), (None, 1764, 256)])
def test_fused_attention_with_precomputed_rel_terms(cast, N, k):
    """evt_softmax_av_gated reading the rel-pos terms from evt_rel_terms == the same launch computing them itself
    (to fp32 rounding of the 64-term dots; one ulp of the store type where a rounding flips), state mode, streamed
    (N = 1764) and register-resident rows."""
    n = native()
    B, H, dh = 1, 2, 64
    D = H * dh
    gw = {70: 10, 42: 7, 1764: 42}[N]
    gh = N // gw
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator(device=DEV).manual_seed(N + k)
    qkv = torch.randn(B, N, 3 * D, device=DEV, generator=g)
    ry = torch.randn(gh, gh, dh, device=DEV, generator=g) * 0.2
    rx = torch.randn(gw, gw, dh, device=DEV, generator=g) * 0.2
    product = torch.randn(B, H, N, N, device=DEV, generator=g)
    idx = torch.stack([torch.randperm(N, device=DEV, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
    a0 = torch.rand(B, H, N, N, device=DEV, generator=g).to(sdt)
    vp = torch.randn(B, N, D, device=DEV, generator=g).to(sdt)
    pv0 = torch.randn(B, N, D, device=DEV, generator=g).to(sdt)
    vd = torch.empty(B, D, k, device=DEV, dtype=sdt)
    vo = torch.empty(B, D, k, device=DEV, dtype=sdt)
    n.v_gate(qkv, idx, None, B, N, D, k, vp, vd, vo, store, True, transposed=True)
    terms = torch.empty(B, H, N, gh + gw, device=DEV)
    n.rel_terms(qkv, ry, rx, B, H, N, D, gh, gw, gw, terms, split=0)   # fp32 chains: the in-kernel computation's arithmetic
    res = []
    for t in (None, terms):
        a_s, pv, out = a0.clone(), pv0.clone(), torch.empty(B, N, D, device=DEV)
        n.softmax_av_gated(product, a_s, idx, None, k, vd, vo, pv, out, B, H, N, D, store, qkv=qkv, rel_y=ry, rel_x=rx,
                           gh=gh, gw=gw, rel_terms=t)
        res.append((a_s.float().cpu(), pv.float().cpu(), out.cpu()))
    tol = 2e-5 if cast is None else 8e-3   # bf16: one ulp at |x| <= 1 (probabilities) .. 2 (accumulated output)
    for x, y in zip(*res):
        assert torch.allclose(x, y, atol=tol * max(1.0, float(y.abs().max())), rtol=0), float((x - y).abs().max())



@pytest.mark.parametrize("cast,N,gw,k,counted", [(None, 1764, 42, 256, False), ("bfloat16", 4096, 64, 400, True), ("float16", 324, 18, 104, False),
                                                 (None, 272, 17, 40, True)])
def test_stream_prep_matches_the_three_launches(cast, N, gw, k, counted):
    """evt_stream_prep (rel-pos terms + key plane + transposed value gate as roles of ONE launch) against evt_rel_terms +
    evt_v_gate + evt_attention_stream's own key-plane pre-kernel: the terms, the value-gate outputs and state, and everything
    the following evt_attention_stream launch produces (with k_split_ready) must be bit-identical -- top-k and a device-side
    count (capacity N), fp32 / bf16 / fp16 store, ViTDet's 42 x 42 and 64 x 64 grids."""
    n = native()
    B, H, dh, scale = 1, 12, 64, 8.0
    D = H * dh
    gh = N // gw
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator(device=DEV).manual_seed(N + k)
    qkv = torch.randn(B, N, 3 * D, device=DEV, generator=g)
    ry = torch.randn(gh, gh, dh, device=DEV, generator=g) * 0.2
    rx = torch.randn(gw, gw, dh, device=DEV, generator=g) * 0.2
    cap = N if counted else k
    idx = torch.zeros(B, cap, dtype=torch.int32, device=DEV)
    idx[0, :k] = torch.randperm(N, device=DEV, generator=g)[:k].sort()[0].int()
    count = torch.full((B,), k, dtype=torch.int32, device=DEV) if counted else None
    apT0 = torch.rand(B, H, N, N, device=DEV, generator=g).to(sdt)
    vp0 = torch.randn(B, N, D, device=DEV, generator=g).to(sdt)
    pv0 = torch.randn(B, N, D, device=DEV, generator=g).to(sdt)
    pref = torch.randn(B, N, D, device=DEV, generator=g)
    res = []
    for prep in (False, True):
        n.clear_scratch()
        apT, vp, pv = apT0.clone(), vp0.clone(), pv0.clone()
        terms = torch.full((B, H, N, gh + gw), float("nan"), device=DEV)
        vd = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
        vo = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
        out = torch.empty(B, N, D, device=DEV)
        parts = torch.empty(B, N, H, device=DEV)
        if prep:
            assert n.stream_prep_fits(D, H, cap, True)
            n.stream_prep(qkv, ry, rx, terms, idx, count, cap, vp, vd, vo, B, H, N, D, gh, gw, gw, store)
        else:
            n.rel_terms(qkv, ry, rx, B, H, N, D, gh, gw, gw, terms)
            n.v_gate(qkv, idx, count, B, N, D, cap, vp, vd, vo, store, True, transposed=True)
        n.attention_stream(qkv, apT, pv, B, H, N, D, scale, store, False, rel_terms=terms, gh=gh, gw=gw, idx=idx, count=count, kcap=cap,
                           v_delta_t=vd, v_old_t=vo, out_f32=out, norm_ref=pref, norm_parts=parts, k_split_ready=prep)
        kplane = n.k_split_plane(qkv, B, H, N, gh, gw).clone()
        res.append([t.cpu() for t in (terms, vd[..., :k], vo[..., :k], vp, kplane.view(torch.int16), apT, pv, out, parts)])
    for name, a_, b_ in zip(("terms", "v_delta", "v_old", "v_state", "key plane", "a_state_t", "pv", "out", "norm_parts"), res[0], res[1]):
        assert torch.equal(a_.view(torch.uint8) if a_.dtype == torch.bfloat16 else a_, b_.view(torch.uint8) if b_.dtype == torch.bfloat16 else b_), name
    assert torch.isfinite(res[1][7]).all()


@pytest.mark.parametrize("cast,qh,qw,pool,k,counted", [(None, 42, 42, (2, 2), 256, False), ("bfloat16", 64, 64, (2, 2), 400, True),
                                                       ("float16", 18, 24, (3, 2), 40, False)])
def test_stream_prep_pooled_matches_the_three_launches(cast, qh, qw, pool, k, counted):
    """evt_stream_prep with pooled keys / values (ABI 8): the key plane and the value gate read the (B,Nk,2D) buffer of evt_pool_kv,
    the rel-pos terms the N query tokens -- bit-identical to evt_rel_terms + evt_v_gate + evt_attention_stream's own key-plane
    pre-kernel, and so is everything the following evt_attention_stream launch produces."""
    n = native()
    B, H, dh, scale = 1, 12, 64, 8.0
    D = H * dh
    N, gh, gw = qh * qw, qh // pool[0], qw // pool[1]
    Nk = gh * gw
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator(device=DEV).manual_seed(N + k)
    qkv = torch.randn(B, N, 3 * D, device=DEV, generator=g)
    kv = torch.empty(B, Nk, 2 * D, device=DEV)
    n.pool_kv(qkv, B, qh, qw, D, pool[0], pool[1], kv)
    ry = torch.randn(qh, gh, dh, device=DEV, generator=g) * 0.2
    rx = torch.randn(qw, gw, dh, device=DEV, generator=g) * 0.2
    cap = Nk if counted else k
    cap += (-cap) % 8
    idx = torch.zeros(B, cap, dtype=torch.int32, device=DEV)
    idx[0, :k] = torch.randperm(Nk, device=DEV, generator=g)[:k].sort()[0].int()
    count = torch.full((B,), k, dtype=torch.int32, device=DEV) if (counted or cap != k) else None
    apT0 = torch.rand(B, H, Nk, N, device=DEV, generator=g).to(sdt)
    vp0 = torch.randn(B, Nk, D, device=DEV, generator=g).to(sdt)
    pv0 = torch.randn(B, N, D, device=DEV, generator=g).to(sdt)
    pref = torch.randn(B, N, D, device=DEV, generator=g)
    res = []
    for prep in (False, True):
        n.clear_scratch()
        apT, vp, pv = apT0.clone(), vp0.clone(), pv0.clone()
        terms = torch.full((B, H, N, gh + gw), float("nan"), device=DEV)
        vd = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
        vo = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
        out = torch.empty(B, N, D, device=DEV)
        parts = torch.empty(B, N, H, device=DEV)
        if prep:
            assert n.stream_prep_fits(D, H, cap, True)
            n.stream_prep(qkv, ry, rx, terms, idx, count, cap, vp, vd, vo, B, H, N, D, gh, gw, qw, store, kv=kv, Nk=Nk)
        else:
            n.rel_terms(qkv, ry, rx, B, H, N, D, gh, gw, qw, terms)
            n.v_gate(kv, idx, count, B, Nk, D, cap, vp, vd, vo, store, True, transposed=True, v_offset=D, v_rs=2 * D)
        n.attention_stream(qkv, apT, pv, B, H, N, D, scale, store, False, rel_terms=terms, gh=gh, gw=gw, idx=idx, count=count, kcap=cap,
                           v_delta_t=vd, v_old_t=vo, out_f32=out, norm_ref=pref, norm_parts=parts, k_split_ready=prep, kv=kv, Nk=Nk)
        kplane = n.k_split_plane(qkv, B, H, Nk, gh, gw).clone()
        res.append([t.cpu() for t in (terms, vd[..., :k], vo[..., :k], vp, kplane.view(torch.int16), apT, pv, out, parts)])
    for name, a_, b_ in zip(("terms", "v_delta", "v_old", "v_state", "key plane", "a_state_t", "pv", "out", "norm_parts"), res[0], res[1]):
        assert torch.equal(a_.view(torch.uint8) if a_.dtype == torch.bfloat16 else a_, b_.view(torch.uint8) if b_.dtype == torch.bfloat16 else b_), name
    assert torch.isfinite(res[1][7]).all()


@pytest.mark.parametrize("cast,qh,qw,pool,k,rel", [(None, 18, 18, (2, 2), 30, True), ("bfloat16", 42, 42, (2, 2), 200, True),
                                                   ("float16", 20, 16, (2, 4), 25, False), (None, 24, 36, (3, 2), 60, True),
                                                   ("bfloat16", 64, 40, (2, 2), 333, True)])
@pytest.mark.parametrize("qk_split", [0, 1])
def test_attention_stream_pooled_keys_matches_oracle(cast, qh, qw, pool, k, rel, qk_split):
    """evt_attention_stream with POOLED keys / values (ABI 8: `kv`, `Nk`; blocks.py:303-326, 509-511): N = qh x qw query tokens
    against Nk = N / (p0 p1) pooled cells -- the (B,Nk,2D) buffer of evt_pool_kv, the (B,H,Nk,N) transposed gate reference, the
    pooled value state, index lists over the cells (incl. a device-side count < kcap), rel-pos terms against the pooled key grid
    (utils.py:139-173 with the tables pooled along the key axis).  First frame + 3 gated frames against the oracle's softmax /
    delta gates / accumulator on the same buffers, and the fused per-head ||out - ref||^2 partials."""
    n = native()
    import torch.nn.functional as F
    B, H, dh, scale = 2, 2, 64, 8.0
    D = H * dh
    N, gh, gw = qh * qw, qh // pool[0], qw // pool[1]
    Nk = gh * gw
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(N * 7 + k)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    apT = torch.full((B, H, Nk, N), float("nan"), dtype=sdt, device=DEV)   # [b][h][cell][row]
    vp = torch.empty(B, Nk, D, dtype=sdt, device=DEV)
    pv = torch.full((B, N, D), float("nan"), dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    ry = torch.randn(qh, gh, dh, generator=g) * 0.2 if rel else None
    rx = torch.randn(qw, gw, dh, generator=g) * 0.2 if rel else None
    tol = (3e-5 if not qk_split else 3e-4) if cast is None else (2e-2 if cast == "bfloat16" else 3e-3)

    def pooled(x):   # (B,H,N,dh) -> (B,H,Nk,dh): Block._pool_tokens (blocks.py:303-326)
        y = x.reshape(B * H, qh, qw, dh).permute(0, 3, 1, 2)
        return F.avg_pool2d(y, pool).permute(0, 2, 3, 1).reshape(B, H, Nk, dh)

    for t in range(4):
        buf = torch.randn(B, N, 3 * D, generator=g) * 1.5
        idx = torch.stack([torch.randperm(Nk, generator=g)[:k].sort()[0] for _ in range(B)])
        q, kk, v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        kk, v = pooled(kk), pooled(v)
        scores = (q / scale) @ kk.transpose(-2, -1)
        if rel:
            qg = q.reshape(B, H, qh, qw, dh)
            ty = torch.einsum("abhwc,hkc->abhwk", qg, ry).unsqueeze(-1)
            tx = torch.einsum("abhwc,wkc->abhwk", qg, rx).unsqueeze(-2)
            logits = ((scores.view(B, H, qh, qw, gh, gw) + ty) + tx).view(B, H, N, Nk)
        else:
            logits = scores
        a = logits.softmax(dim=-1)
        if cast is not None:
            a, v = a.to(sdt), v.to(sdt)
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        bd, idxd = buf.to(DEV), idx.int().to(DEV)
        kv = torch.full((B, Nk, 2 * D), float("nan"), device=DEV)
        n.pool_kv(bd, B, qh, qw, D, pool[0], pool[1], kv)
        terms = None
        if rel:
            terms = torch.empty(B, H, N, gh + gw, device=DEV)
            n.rel_terms(bd, ry.to(DEV), rx.to(DEV), B, H, N, D, gh, gw, qw, terms, split=qk_split)
        relkw = dict(rel_terms=terms, gh=gh, gw=gw) if rel else {}
        if t == 0:
            n.v_gate(kv, None, None, B, Nk, D, 0, vp, None, None, store, False, v_offset=D, v_rs=2 * D)
            n.attention_stream(bd, apT, pv, B, H, N, D, scale, store, True, v_state=vp, out_f32=out, qk_split=qk_split, kv=kv, Nk=Nk, **relkw)
        else:
            cap = k if t != 2 else k + 8
            idx_cap = torch.full((B, cap), 0, dtype=torch.int32, device=DEV)
            idx_cap[:, :k] = idxd
            count = None if cap == k else torch.full((B,), k, dtype=torch.int32, device=DEV)
            v_del = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            v_old = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            n.v_gate(kv, idx_cap, count, B, Nk, D, cap, vp, v_del, v_old, store, True, transposed=True, v_offset=D, v_rs=2 * D)
            ref_next = torch.randn(B, N, D, generator=g).to(DEV)
            parts = torch.full((B, N, H), float("nan"), device=DEV)
            n.attention_stream(bd, apT, pv, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap,
                               v_delta_t=v_del, v_old_t=v_old, out_f32=out, norm_ref=ref_next, norm_parts=parts,
                               qk_split=qk_split, kv=kv, Nk=Nk, **relkw)
            want_parts = (out - ref_next).view(B, N, H, dh).pow(2).sum(-1)
            assert torch.allclose(parts, want_parts, rtol=1e-5, atol=1e-6), float((parts - want_parts).abs().max())
        atol_p = {None: tol * 0.1 + 3e-6, "bfloat16": 4e-3, "float16": 5e-4}[cast]
        got_p = apT.float().cpu().transpose(-1, -2)
        assert torch.allclose(got_p, ag.t.float(), atol=atol_p), (cast, t, float((got_p - ag.t.float()).abs().max()))
        err = float((out.cpu() - ref).abs().max())
        bar = tol if cast is None else max(tol, (1 if t == 0 else 2) * float(ref.abs().max()) * (2.0 ** -7 if cast == "bfloat16" else 2.0 ** -10))   # (persisting roundings: see above)
        assert err <= bar, (cast, N, Nk, k, t, err, bar)
        assert torch.equal(out.cpu(), pv.float().cpu())


@pytest.mark.parametrize("cast,N,k", [("bfloat16", 197, 128), ("float16", 197, 128), ("bfloat16", 37, 12), ("float16", 70, 40),
                                      ("bfloat16", 256, 100), ("bfloat16", 130, 130), ("float16", 225, 1), ("bfloat16", 64, 33)])
def test_attention_gated_resident_matches_oracle(cast, N, k):
    """evt_attention_gated (K10: one workgroup per (clip, head); value gate, scores, softmax, A gate and both accumulator products in
    one launch; TILED gate reference): first frame + 3 gated frames against the oracle's value gate / softmax / delta gates /
    accumulator on the same token buffers -- incl. a device-side count < kcap, partial last tiles (N % 32 != 0), N = 256, every key
    selected, a single key selected, and the fused per-head ||out - ref||^2 partials."""
    n = native()
    B, H, dh, scale = 2, 3, 64, 8.0
    D = H * dh
    sdt = getattr(torch, cast)
    store = n.store_code(sdt)
    assert n.attention_gated_fits(N, D, H, store)
    g = torch.Generator().manual_seed(N * 19 + k)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    tiles = n.gated_tiles_empty(B, H, N, sdt, DEV)
    tiles.view(torch.int16).fill_(0x7fc0 if cast == "bfloat16" else 0x7e00)   # NaN everywhere: the first frame must write every tile
    vp = torch.full((B, N, D), float("nan"), dtype=sdt, device=DEV)
    pv = torch.full((B, N, D), float("nan"), dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    tol = 2e-2 if cast == "bfloat16" else 3e-3
    for t in range(4):
        buf = torch.randn(B, N, 3 * D, generator=g) * 1.5
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        q, kk, v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        a = ((q / scale) @ kk.transpose(-2, -1)).softmax(dim=-1).to(sdt)
        v = v.to(sdt)
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        bd, idxd = buf.to(DEV), idx.int().to(DEV)
        if t == 0:
            n.attention_gated(bd, tiles, vp, pv, B, H, N, D, scale, store, True, out_f32=out)
        else:
            cap = k if (t != 2 or k == N) else min(N, k + 5)
            idx_cap = torch.full((B, cap), 0, dtype=torch.int32, device=DEV)
            idx_cap[:, :k] = idxd
            count = None if cap == k else torch.full((B,), k, dtype=torch.int32, device=DEV)
            ref_next = torch.randn(B, N, D, generator=g).to(DEV)      # stands for the projection gate's reference
            parts = torch.full((B, N, H), float("nan"), device=DEV)
            n.attention_gated(bd, tiles, vp, pv, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap, out_f32=out,
                              norm_ref=ref_next, norm_parts=parts)
            want_parts = (out - ref_next).view(B, N, H, dh).pow(2).sum(-1)
            assert torch.allclose(parts, want_parts, rtol=1e-5, atol=1e-6), float((parts - want_parts).abs().max())
        # the value reference is elementwise: exact
        got_v = vp.view(B, N, H, dh).permute(0, 2, 1, 3).float().cpu()
        assert torch.equal(got_v, vs.t.float()), (cast, t, float((got_v - vs.t.float()).abs().max()))
        # probabilities are rounded to the store type from scores computed in a different fp32 summation order than the CPU's:
        # one ulp of the store type at p <= 1 (bf16 2^-8, fp16 2^-11)
        atol_p = {"bfloat16": 4e-3, "float16": 5e-4}[cast]
        got_p = n.tiles_to_logical(tiles, N).float().cpu()
        assert torch.isfinite(got_p).all()
        assert torch.allclose(got_p, ag.t.float(), atol=atol_p), (cast, t, float((got_p - ag.t.float()).abs().max()))
        err = float((out.cpu() - ref).abs().max())
        # the output IS the store-type A.v state, and the state persists: a rounding flipped in one frame (the fp32 sums are added in
        # another order than the CPU's) stays, so an element may be off by one ulp of the store type per rounding that flipped so
        # far -- two are seen (fp16, ~0.2 % of the roundings flip), never more on these streams
        ulps = 1 if t == 0 else 2
        bar = max(tol, ulps * float(ref.abs().max()) * (2.0 ** -7 if cast == "bfloat16" else 2.0 ** -10))
        assert err <= bar, (cast, N, k, t, err, bar)
        assert torch.equal(out.cpu(), pv.float().cpu())
    # the fp32 output may be omitted (the caller reads the A.v state): same state
    pv2, tiles2, vp2 = pv.clone(), tiles.clone(), vp.clone()
    n.attention_gated(bd, tiles2, vp2, pv2, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap)
    pv3, tiles3, vp3 = pv.clone(), tiles.clone(), vp.clone()
    n.attention_gated(bd, tiles3, vp3, pv3, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap, out_f32=out)
    assert torch.equal(pv2.view(torch.int16), pv3.view(torch.int16)) and torch.equal(out, pv3.float())
    # layout round trip through the host helpers (what matmul_gate.p's setter / getter do)
    tiles4 = torch.zeros_like(tiles)
    n.logical_to_tiles(n.tiles_to_logical(tiles, N), tiles4)
    assert torch.equal(n.tiles_to_logical(tiles4, N).view(torch.int16), n.tiles_to_logical(tiles, N).view(torch.int16))
    with pytest.raises(RuntimeError, match="evt_attention_gated"):
        n.attention_gated(torch.empty(1, 300, 3 * D, device=DEV), tiles, vp, pv, 1, H, 300, D, scale, store, True, out_f32=out)


@pytest.mark.parametrize("cast,N,counts", [("bfloat16", 100, (0, 7, 100)), ("float16", 197, (197, 0, 1)), ("bfloat16", 33, (33, 32, 0))])
def test_attention_gated_resident_ragged_device_counts(cast, N, counts):
    """evt_attention_gated with a DIFFERENT device-side count per clip (the threshold policy's case, policies.py:58-68): a clip whose
    gate selects nothing (its state must not move: modules.py:187-201 with an empty index), one that selects every key, and one in
    between, in the same launch; the index slots behind a clip's count hold garbage tokens that must not be read as selected."""
    n = native()
    B, H, dh, scale = len(counts), 2, 64, 8.0
    D = H * dh
    sdt = getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(N * 7 + sum(counts))
    slots = [(O.Slot(), O.Slot(), O.Slot()) for _ in range(B)]
    tiles = n.gated_tiles_empty(B, H, N, sdt, DEV)
    vp = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv = torch.empty(B, N, D, dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    tol = 2e-2 if cast == "bfloat16" else 3e-3
    for t in range(3):
        buf = torch.randn(B, N, 3 * D, generator=g) * 1.5
        idx_cap = torch.randint(0, N, (B, N), generator=g).int()      # garbage behind the counts
        refs, lists = [], []
        for b in range(B):
            c = counts[(b + t) % B] if t else N                           # the counts rotate over the clips from frame to frame
            idx = torch.randperm(N, generator=g)[:c].sort()[0]
            lists.append(idx)
            idx_cap[b, :c] = idx.int()
            q, kk, v = buf[b:b + 1].view(1, N, 3, H, dh).permute(2, 0, 3, 1, 4)
            a = ((q / scale) @ kk.transpose(-2, -1)).softmax(dim=-1).to(sdt)
            v = v.to(sdt)
            vs, ag, acc = slots[b]
            forced = idx.unsqueeze(0) if t else None
            v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=forced)
            a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=forced, structure="col")
            refs.append(O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float())
        ref = torch.cat(refs)
        before = (pv.clone(), tiles.clone(), vp.clone())
        if t == 0:
            n.attention_gated(buf.to(DEV), tiles, vp, pv, B, H, N, D, scale, store, True, out_f32=out)
        else:
            count = torch.tensor([len(l) for l in lists], dtype=torch.int32, device=DEV)
            n.attention_gated(buf.to(DEV), tiles, vp, pv, B, H, N, D, scale, store, False, idx=idx_cap.to(DEV), count=count, kcap=N, out_f32=out)
            for b in range(B):
                if len(lists[b]) == 0:   # nothing selected: the clip's three states are untouched, bit for bit
                    for was, now in zip(before, (pv, tiles, vp)):
                        assert torch.equal(was[b].view(torch.int16), now[b].view(torch.int16)), (cast, t, b)
        got_v = torch.stack([vp[b].view(N, H, dh).permute(1, 0, 2).float().cpu() for b in range(B)])
        want_v = torch.cat([slots[b][0].t.float() for b in range(B)])
        assert torch.equal(got_v, want_v), (cast, t)
        want_p = torch.cat([slots[b][1].t.float() for b in range(B)])
        got_p = n.tiles_to_logical(tiles, N).float().cpu()
        assert torch.allclose(got_p, want_p, atol={"bfloat16": 4e-3, "float16": 5e-4}[cast]), (cast, t)
        err = float((out.cpu() - ref).abs().max())
        bar = max(tol, (1 if t == 0 else 2) * float(ref.abs().max()) * (2.0 ** -7 if cast == "bfloat16" else 2.0 ** -10))
        assert err <= bar, (cast, N, t, err, bar)
        assert torch.equal(out.cpu(), pv.float().cpu())


def test_resident_attention_kernels_are_repeatable_under_load():
    """Race check for the two resident attention kernels, whose waves reuse LDS regions across barriers (the q / epilogue blocks
    inside the K planes): 25 runs of the same launch at the headline's occupancy (3 workgroups per CU queued) must agree bit for bit
    -- evt_attention_gated (gated frame, N = 197, k = 128, bf16) on cloned states, evt_attention_dense's resident form (196-token
    windows with rel-pos, fp32)."""
    n = native()
    H, dh, N, k, B = 12, 64, 197, 128, 64
    D = H * dh
    g = torch.Generator(device=DEV).manual_seed(5)
    sdt = torch.bfloat16
    store = n.store_code(sdt)
    buf = torch.randn(B, N, 3 * D, device=DEV, generator=g)
    tiles0 = n.gated_tiles_empty(B, H, N, sdt, DEV)
    vp0 = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv0 = torch.empty(B, N, D, dtype=sdt, device=DEV)
    n.attention_gated(buf, tiles0, vp0, pv0, B, H, N, D, 8.0, store, True)
    buf2 = buf + 0.3 * torch.randn(B, N, 3 * D, device=DEV, generator=g)
    idx = torch.stack([torch.randperm(N, device=DEV, generator=g)[:k].sort()[0] for _ in range(B)]).int()
    ref_next = torch.randn(B, N, D, device=DEV, generator=g)
    first = None
    for _ in range(25):
        tiles, vp, pv = tiles0.clone(), vp0.clone(), pv0.clone()
        out = torch.empty(B, N, D, device=DEV)
        parts = torch.empty(B, N, H, device=DEV)
        n.attention_gated(buf2, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, out_f32=out, norm_ref=ref_next, norm_parts=parts)
        got = (tiles.view(torch.int16), vp.view(torch.int16), pv.view(torch.int16), out, parts)
        if first is None:
            first = got
            assert torch.isfinite(out).all() and torch.isfinite(parts).all()
        else:
            for a_, b_ in zip(first, got):
                assert torch.equal(a_, b_)
    G = 72
    wqkv = torch.randn(G, 196, 3 * D, device=DEV, generator=g)
    ry = torch.randn(14, 14, dh, device=DEV, generator=g) * 0.2
    rx = torch.randn(14, 14, dh, device=DEV, generator=g) * 0.2
    first = None
    for _ in range(25):
        wout = torch.empty(G, 196, D, device=DEV)
        n.attention_dense(wqkv, G, H, 196, D, 8.0, n.EVT_F32, out_f32=wout, rel_y=ry, rel_x=rx, gh=14, gw=14, qw=14)
        if first is None:
            first = wout
            assert torch.isfinite(wout).all()
        else:
            assert torch.equal(first, wout)


@pytest.mark.parametrize("cast,N,gw,k,rel", [(None, 260, 13, 40, True), ("bfloat16", 324, 18, 100, True), ("float16", 288, 16, 64, False),
                                             (None, 1764, 42, 256, True), ("bfloat16", 1024, 32, 333, True),
                                             ("bfloat16", 512, 64, 100, True), (None, 350, 70, 60, True), (None, 280, 20, 50, True),
                                             ("bfloat16", 197, 197, 128, False), (None, 262, 131, 77, False)])
@pytest.mark.parametrize("qk_split", [0, 1])
def test_attention_stream_matches_oracle(cast, N, gw, k, rel, qk_split):
    """evt_attention_stream (N > 256, scores computed in the kernel, TRANSPOSED gate reference): first frame + 3 gated
    frames against the oracle's softmax / delta gates / accumulator on the same token buffers -- incl. a device-side
    count < kcap, rel-pos terms from evt_rel_terms, a partial last row tile (N % 32 != 0), odd token counts (ViViT's 197), grids whose
    rows end early in a 16-key block (gw = 18, 20: lanes of the statistics pass that never see a valid key) and the fused per-head
    ||out - ref||^2 partials."""
    n = native()
    B, H, dh, scale = 2, 2, 64, 8.0
    D = H * dh
    gh = N // gw
    sdt = torch.float32 if cast is None else getattr(torch, cast)
    store = n.store_code(sdt)
    g = torch.Generator().manual_seed(N * 13 + k)
    vs, ag, acc = O.Slot(), O.Slot(), O.Slot()
    apT = torch.full((B, H, N, N), float("nan"), dtype=sdt, device=DEV)   # [b][h][key][row]
    vp = torch.empty(B, N, D, dtype=sdt, device=DEV)
    pv = torch.full((B, N, D), float("nan"), dtype=sdt, device=DEV)
    out = torch.empty(B, N, D, device=DEV)
    ry = torch.randn(gh, gh, dh, generator=g) * 0.2 if rel else None
    rx = torch.randn(gw, gw, dh, generator=g) * 0.2 if rel else None
    tol = (3e-5 if not qk_split else 3e-4) if cast is None else (2e-2 if cast == "bfloat16" else 3e-3)
    for t in range(4):
        buf = torch.randn(B, N, 3 * D, generator=g) * 1.5
        idx = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        q, kk, v = buf.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        scores = (q / scale) @ kk.transpose(-2, -1)
        logits = O.add_relative(scores.clone(), q, ry, rx, (gh, gw), inplace=False) if rel else scores
        a = logits.softmax(dim=-1)
        if cast is not None:
            a, v = a.to(sdt), v.to(sdt)
        else:
            v = v.clone()
        v_n, v_d, _ = O.token_delta_gate(vs, v, None, forced=idx if t else None)
        a_n, a_d, _ = O.token_delta_gate(ag, a, None, forced=idx if t else None, structure="col")
        ref = O.BlockOracle._merge(O.av_accumulator(acc, a_n, v_n, a_d, v_d)).float()
        bd, idxd = buf.to(DEV), idx.int().to(DEV)
        terms = None
        if rel:
            terms = torch.empty(B, H, N, gh + gw, device=DEV)
            n.rel_terms(bd, ry.to(DEV), rx.to(DEV), B, H, N, D, gh, gw, gw, terms, split=qk_split)
        relkw = dict(rel_terms=terms, gh=gh, gw=gw) if rel else {}
        if t == 0:
            n.v_gate(bd, None, None, B, N, D, 0, vp, None, None, store, False)
            n.attention_stream(bd, apT, pv, B, H, N, D, scale, store, True, v_state=vp, out_f32=out, qk_split=qk_split, **relkw)
        else:
            cap = k if t != 2 else k + 8
            idx_cap = torch.full((B, cap), 0, dtype=torch.int32, device=DEV)
            idx_cap[:, :k] = idxd
            count = None if cap == k else torch.full((B,), k, dtype=torch.int32, device=DEV)
            v_del = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            v_old = torch.full((B, D, cap), float("nan"), dtype=sdt, device=DEV)
            n.v_gate(bd, idx_cap, count, B, N, D, cap, vp, v_del, v_old, store, True, transposed=True)
            ref_next = torch.randn(B, N, D, generator=g).to(DEV)
            parts = torch.full((B, N, H), float("nan"), device=DEV)
            n.attention_stream(bd, apT, pv, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap,
                               v_delta_t=v_del, v_old_t=v_old, out_f32=out, norm_ref=ref_next, norm_parts=parts,
                               qk_split=qk_split, **relkw)
            want_parts = (out - ref_next).view(B, N, H, dh).pow(2).sum(-1)
            assert torch.allclose(parts, want_parts, rtol=1e-5, atol=1e-6), float((parts - want_parts).abs().max())
        atol_p = {None: tol * 0.1 + 3e-6, "bfloat16": 4e-3, "float16": 5e-4}[cast]
        got_p = apT.float().cpu().transpose(-1, -2)
        assert torch.allclose(got_p, ag.t.float(), atol=atol_p), (cast, t, float((got_p - ag.t.float()).abs().max()))
        err = float((out.cpu() - ref).abs().max())
        # (the store-type state PERSISTS: a rounding flipped in an earlier frame stays, so from the first gated frame on an element may be
        # off by two ulps -- scripts/probes/random_attention_probe.py met 1.5 ulps of the top binade at t = 2 on random shapes)
        bar = tol if cast is None else max(tol, (1 if t == 0 else 2) * float(ref.abs().max()) * (2.0 ** -7 if cast == "bfloat16" else 2.0 ** -10))
        assert err <= bar, (cast, N, k, t, err, bar)
        assert torch.equal(out.cpu(), pv.float().cpu())
    if cast is not None:   # 16-bit store: the fp32 output may be omitted (the caller reads the A.v state)
        pv2 = pv.clone()
        n.attention_stream(bd, apT.clone(), pv2, B, H, N, D, scale, store, False, idx=idx_cap, count=count, kcap=cap,
                           v_delta_t=v_del, v_old_t=v_old, out_f32=None, qk_split=qk_split, **relkw)
