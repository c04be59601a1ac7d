"""GPU: the reference's stand-alone module / policy API on HIP tensors
against the oracle, plus reference-pinned MAC counters (I5)."""
import os

import numpy as np
import pytest
import torch

import eventful_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _asc(index):
    return index.sort(dim=-1)[0]


def test_policies_standalone():
    from eventful_transformer import policies
    c, p = O.make_gate_case(3, 2, 197, 768)
    e = c - p
    top = policies.TokenNormTopK(k=64, save_status=True)
    got = top(e.to(DEV))
    assert got.dtype == torch.int64 and got.shape == (2, 64) and got.device.type == "cuda"
    assert torch.equal(got.cpu(), _asc(O.TopK(64)(e)))
    assert torch.equal(top.last_output, got) and torch.equal(top.last_input.cpu(), e)
    frac = policies.TokenNormTopFraction(0.25)
    assert torch.equal(frac(e.to(DEV)).cpu(), _asc(O.TopFraction(0.25)(e)))
    # norm over dim=-2 (column structure): tokens along the last dim
    assert torch.equal(top(e.transpose(1, 2).contiguous().to(DEV), dim=-2).cpu(), _asc(O.TopK(64)(e)))
    c1, p1, thr = O.make_threshold_case(5, 300, 128, 41)
    thr_pol = policies.TokenNormThreshold(threshold=thr)
    got = thr_pol((c1 - p1).to(DEV))
    assert got.shape == (1, 41) and torch.equal(got.cpu(), O.Threshold(thr)(c1 - p1))
    with pytest.raises(AssertionError):
        thr_pol(e.to(DEV))  # batch > 1, policies.py:25
    # the policies' `order` argument (policies.py:11,44,76 -> vector_norm(ord=order)): L1 and L-infinity delta norms
    for order in (1, float("inf")):
        assert torch.equal(policies.TokenNormTopK(k=64, order=order)(e.to(DEV)).cpu(), _asc(O.TopK(64, order=order)(e)))
        assert torch.equal(policies.TokenNormTopFraction(0.25, order=order)(e.to(DEV)).cpu(), _asc(O.TopFraction(0.25, order=order)(e)))
        d1 = c1 - p1
        n1 = torch.linalg.vector_norm(d1, ord=order, dim=-1)[0].sort()[0]
        t1 = float((n1[-30] + n1[-31]) / 2)      # a threshold between two norms: 30 tokens above it
        got = policies.TokenNormThreshold(threshold=t1, order=order)(d1.to(DEV))
        assert got.shape == (1, 30) and torch.equal(got.cpu(), O.Threshold(t1, order=order)(d1))
    with pytest.raises(NotImplementedError):
        policies.TokenNormTopK(k=4, order=3)(e.to(DEV))


@pytest.mark.parametrize("delta", [False, True])
def test_token_gates_standalone(delta):
    """TokenGate / TokenDeltaGate .forward(c[, forced_index]) with the reference's returns and state."""
    from eventful_transformer import modules, policies
    cls = modules.TokenDeltaGate if delta else modules.TokenGate
    ogate = O.token_delta_gate if delta else O.token_gate
    gate, slot = cls(), O.Slot()
    gate.policy = policies.TokenNormTopK(k=9)
    xs = O.make_token_stream(2, 40, 64, 4, 9, seed=1, small=0.05)
    for t in range(3):
        out = gate(xs[t].clone().to(DEV))
        ref = ogate(slot, xs[t].clone(), O.TopK(9))
        if t == 0:
            assert out[-1] is None and torch.equal(out[0].cpu(), ref[0])
            continue
        assert torch.equal(out[-1].cpu(), _asc(ref[-1]))
        order = ref[-1].sort(dim=-1)[1]
        for got, want in zip(out[:-1], ref[:-1]):
            want = want.gather(1, order.unsqueeze(-1).expand(-1, -1, 64))
            assert torch.equal(got.cpu(), want)
        assert torch.equal(gate.p.cpu(), slot.t)  # I3
    # forced index on a head-split tensor (B, H, N, dh): the (B, k) index broadcasts over heads
    g2, s2 = cls(), O.Slot()
    v0, v1 = torch.randn(2, 3, 20, 16), torch.randn(2, 3, 20, 16)
    forced = torch.stack([torch.randperm(20)[:5].sort()[0] for _ in range(2)])
    g2(v0.clone().to(DEV)); ogate(s2, v0.clone(), None)
    out = g2(v1.clone().to(DEV), forced_index=forced.to(DEV))
    ref = ogate(s2, v1.clone(), None, forced=forced)
    for got, want in zip(out[:-1], ref[:-1]):
        assert torch.equal(got.cpu(), want)
    assert torch.equal(g2.p.cpu(), s2.t)
    gate.reset()
    assert gate.first and gate.p is None


def test_col_gate_and_buffers_standalone():
    from eventful_transformer import modules
    a0, a1 = torch.rand(2, 3, 10, 10), torch.rand(2, 3, 10, 10)
    forced = torch.stack([torch.randperm(10)[:4].sort()[0] for _ in range(2)])
    g, s = modules.TokenDeltaGate(structure="col"), O.Slot()
    g(a0.clone().to(DEV)); O.token_delta_gate(s, a0.clone(), None, structure="col")
    out = g(a1.clone().to(DEV), forced_index=forced.to(DEV))
    ref = O.token_delta_gate(s, a1.clone(), None, forced=forced, structure="col")
    assert torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref[1]) and torch.equal(g.p.cpu(), s.t)
    # TokenBuffer rows / cols
    for structure, shape_x in (("row", (2, 4, 32)), ("col", (2, 3, 10, 4))):
        buf, sb = modules.TokenBuffer(structure=structure), O.Slot()
        first = torch.randn(2, 10, 32) if structure == "row" else torch.randn(2, 3, 10, 10)
        x = torch.randn(*shape_x)
        b0 = buf(first.clone().to(DEV), None)
        O.token_buffer(sb, first.clone(), None, structure=structure)
        assert b0 is buf.b
        out = buf(x.to(DEV), forced.to(DEV))
        ref = O.token_buffer(sb, x, forced, structure=structure)
        assert out is buf.b and torch.equal(out.cpu(), ref)
    # SimpleSTGTGate: reference replaced wholesale each frame
    from eventful_transformer import policies
    st, ss = modules.SimpleSTGTGate(), O.Slot()
    st.policy = policies.TokenNormTopK(k=3)
    xs = O.make_token_stream(1, 12, 32, 3, 3, seed=2, small=0.05)
    for t in range(3):
        out = st(xs[t].clone().to(DEV))
        ref = O.stgt_gate(ss, xs[t].clone(), O.TopK(3))
        if t:
            assert torch.equal(out[1].cpu(), _asc(ref[1]))
            assert torch.equal(st.p.cpu(), xs[t])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gates_and_buffers_of_any_structure_and_dtype(dtype):
    """Stand-alone gates / buffers off the fp32 row fast path -- column structure, 16-bit element types, odd row lengths, a
    non-contiguous first input (the reference keeps a REFERENCE to it, modules.py:140) -- on evt_gate_cols / evt_scatter_cols /
    evt_gate_rows_any / evt_move_rows_any, against the oracle's gather / delta / scatter in the same dtype (bit-equal: the delta is one
    subtraction rounded to the dtype)."""
    from eventful_transformer import modules
    g_ = torch.Generator().manual_seed(11)
    for structure, shape in (("col", (2, 3, 9, 13)), ("row", (2, 3, 13, 7)), ("row", (3, 11, 6))):
        k = 4
        tokens = shape[-1] if structure == "col" else shape[-2]
        lead = shape[:1]
        c0 = torch.randn(*shape, generator=g_).to(dtype)
        c1 = torch.randn(*shape, generator=g_).to(dtype)
        c2 = torch.randn(*shape, generator=g_).to(dtype)
        forced = [torch.stack([torch.randperm(tokens, generator=g_)[:k].sort()[0] for _ in range(lead[0])]) for _ in range(2)]
        gate, slot = modules.TokenDeltaGate(structure=structure), O.Slot()
        # first input handed over as a permuted (non-contiguous) view of a transposed tensor
        first_dev = c0.transpose(-1, -2).contiguous().to(DEV).transpose(-1, -2)
        assert not first_dev.is_contiguous()
        gate(first_dev)
        O.token_delta_gate(slot, c0.clone(), None, structure=structure)
        for c, f in ((c1, forced[0]), (c2, forced[1])):
            out = gate(c.clone().to(DEV), forced_index=f.to(DEV))
            ref = O.token_delta_gate(slot, c.clone(), None, forced=f, structure=structure)
            assert out[0].dtype == dtype and torch.equal(out[0].cpu(), ref[0]) and torch.equal(out[1].cpu(), ref[1])
            assert torch.equal(gate.p.cpu(), slot.t)
        buf, sb = modules.TokenBuffer(structure=structure), O.Slot()
        buf(c0.clone().to(DEV), None)
        O.token_buffer(sb, c0.clone(), None, structure=structure)
        xs = (torch.randn(*(shape[:-1] + (k,)), generator=g_) if structure == "col" else torch.randn(*(shape[:-2] + (k, shape[-1])), generator=g_)).to(dtype)
        out = buf(xs.to(DEV), forced[0].to(DEV))
        ref = O.token_buffer(sb, xs, forced[0], structure=structure)
        assert out is buf.b and torch.equal(out.cpu(), ref)


def test_row_map_kernels_window_partition_and_back():
    """evt_gather_rows_map / evt_scatter_rows_map: the window partition of a token buffer with padding tokens (map < 0 -> the pad row)
    and its inverse (padding dropped), and a per-batch map (the ATS row gather); evt_move_rows_any with heads sharing a clip's map."""
    from eventful_transformer import _native as n, blocks
    B, F = 2, 24
    tok_map = blocks._window_map((7, 5), (3, 3), torch.device(DEV))          # (windows, 9), -1 = padding
    x = torch.randn(B, 35, F, device=DEV)
    pad = torch.randn(F, device=DEV)
    out = torch.full((B, tok_map.numel(), F), float("nan"), device=DEV)
    n.gather_rows_map(x, tok_map, B, 35, F, tok_map.numel(), out, pad_row=pad)
    flat = tok_map.reshape(-1).long()
    want = torch.where((flat >= 0)[None, :, None], x[:, flat.clamp(min=0)], pad[None, None, :].expand(B, flat.numel(), F))
    assert torch.equal(out, want)
    back = torch.zeros(B, 35, F, device=DEV)
    n.scatter_rows_map(out, tok_map, B, tok_map.numel(), 35, F, back)
    assert torch.equal(back, x)
    per = torch.stack([torch.randperm(35)[:6].sort()[0] for _ in range(B)]).int().to(DEV)
    got = torch.empty(B, 6, F, device=DEV)
    n.gather_rows_map(x, per, B, 35, F, 6, got, map_per_batch=True)
    assert torch.equal(got, torch.stack([x[b, per[b].long()] for b in range(B)]))
    a = torch.randn(B, 3, 35, 35, device=DEV).to(torch.bfloat16)     # heads of a clip share its map
    rows = torch.empty(B, 3, 6, 35, device=DEV, dtype=torch.bfloat16)
    n.move_rows_any(a, per, B * 3, 35, 35, 6, rows, rep=3)
    assert torch.equal(rows, torch.stack([a[b][:, per[b].long()] for b in range(B)]))


def test_matmul_buffer_and_accumulator_standalone():
    from eventful_transformer import modules
    B, Hh, N, dh, k = 2, 3, 21, 16, 6
    mb, sm = modules.MatmulBuffer(), O.Slot()
    acc, sa = modules.MatmulDeltaAccumulator(), O.Slot()
    g = torch.Generator().manual_seed(0)
    for t in range(3):
        q, kT = torch.randn(B, Hh, N, dh, generator=g), torch.randn(B, Hh, dh, N, generator=g)
        iq = torch.stack([torch.randperm(N, generator=g)[:k].sort()[0] for _ in range(B)])
        ik = torch.stack([torch.randperm(N, generator=g)[:k + 1].sort()[0] for _ in range(B)])
        out = mb(q.to(DEV), kT.to(DEV), iq.to(DEV) if t else None, ik.to(DEV) if t else None)
        ref = O.qk_buffer(sm, q, kT, iq, ik)
        assert out is mb.product and torch.allclose(out.cpu(), ref, atol=1e-5)
        a_n, a_d = torch.rand(B, Hh, N, k if t else N, generator=g), torch.randn(B, Hh, N, k if t else N, generator=g) * 0.1
        v_n, v_d = torch.randn(B, Hh, k if t else N, dh, generator=g), torch.randn(B, Hh, k if t else N, dh, generator=g) * 0.1
        out = acc(a_n.to(DEV), v_n.to(DEV), a_d.to(DEV) if t else None, v_d.to(DEV) if t else None)
        ref = O.av_accumulator(sa, a_n, v_n, a_d, v_d)
        assert out.shape == ref.shape and torch.allclose(out.cpu(), ref, atol=2e-5)
    mb.reset(); acc.reset()
    assert mb.first and mb.product is None and acc.product is None


def test_counted_linear_and_position_encoding():
    from eventful_transformer.counting import CountedLinear
    from eventful_transformer.utils import PositionEncoding
    lin = CountedLinear(64, 96)
    w, b = torch.randn(96, 64) * 0.1, torch.randn(96)
    lin.load_state_dict({"weight": w, "bias": b})
    lin = lin.to(DEV)
    x = torch.randn(3, 5, 64)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double()).float()
    assert torch.allclose(lin(x.to(DEV)).cpu(), ref, atol=1e-4)
    assert torch.allclose(lin.forward_linear(x.to(DEV)).cpu(), ref - b, atol=1e-4)
    assert torch.equal(lin.forward_bias(x.new_zeros(1, 96).to(DEV)).cpu(), b.unsqueeze(0))
    pe = PositionEncoding(64, (3, 3), (6, 6), True).eval()
    enc = torch.randn(1, 10, 64) * 0.1
    pe.load_state_dict({"encoding": enc})
    pe = pe.to(DEV)
    xx = torch.randn(2, 37, 64)
    want = xx + O.sized_position_encoding(enc, (3, 3), (6, 6), True)
    assert torch.allclose(pe(xx.to(DEV)).cpu(), want, atol=2e-6)


def test_mac_counters_match_reference(golden_dir):
    """I5: MAC counters of the fused HIP path == the REFERENCE's counters (golden) on a 3-block backbone
    with a windowed rel-pos block and two global EventfulBlocks, first frame and two gated frames."""
    from eventful_transformer import policies
    from eventful_transformer.backbones import ViTBackbone
    g = H.load_npz(os.path.join(golden_dir, "counts.npz"))
    cfg = dict(dim=64, heads=4, mlp_ratio=4, relative_embedding_size=(8, 8), window_size=(3, 3))
    bb = ViTBackbone(block_config=cfg, depth=3, position_encoding_size=(3, 3), input_size=(6, 6),
                     block_class="EventfulBlock", windowed_class="EventfulTokenwiseBlock", window_indices=(0,)).eval().to(DEV)
    H.set_policies(bb, policies.TokenNormTopK, k=12)
    xs = O.make_token_stream(2, 36, 64, 3, 12, seed=11, small=0.02)
    bb.counting()
    keys = ("add_flops", "bias_flops", "linear_flops", "einsum_flops", "matmul_flops", "gate_flops", "accumulator_flops")
    with torch.inference_mode():
        for t in range(3):
            bb.clear_counts()
            bb(xs[t].to(DEV))
            c = bb.total_counts()
            for key in keys:
                want = int(g[f"t{t}__{key}"]) if f"t{t}__{key}" in g.files else 0
                assert int(c[key]) == want, (t, key, int(c[key]), want)


def test_two_streams_share_a_work_lane_safely_and_overlap_with_lanes():
    """The scratch pool is keyed by (name, shape, dtype, device, work lane, host thread), not by stream: two same-shaped models
    driven on two HIP streams must still give what they give one after the other -- with their own lanes (they may overlap) and
    WITHOUT lanes (a stream switch inside a lane makes the new stream wait for the previous one's work)."""
    from eventful_transformer import _native, policies
    params = O.make_block_params(64, 4, seed=3, std=0.08)
    xs = O.make_token_stream(2, 37, 64, 3, 12, seed=4, small=0.02).to(DEV)

    def make():
        blk = H.product_block("EventfulBlock", params, 64, 4, (6, 6))
        H.set_policies(blk, policies.TokenNormTopK, k=12)
        return blk

    ref = make()
    with torch.inference_mode():
        want = [ref(xs[t]).clone() for t in range(3)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=DEV)
    for lanes in (True, False):
        a, b = make(), make()
        outs = {0: [], 1: []}
        with torch.inference_mode():
            for t in range(3):
                for i, (m, st) in enumerate(((a, sa), (b, sb))):
                    with torch.cuda.stream(st), _native.lane(10 + i if lanes else 0):
                        if not lanes:
                            big2 = big @ big          # keeps this stream busy while the other one enters the shared lane
                        outs[i].append(m(xs[t]).clone())
        torch.cuda.synchronize()
        for i in (0, 1):
            for t in range(3):
                assert torch.equal(outs[i][t], want[t]), (lanes, i, t)


def test_frames_that_do_not_fit_the_clip_state_are_errors():
    """A later frame with another batch size or token count than the clip's first frame, without reset(): the reference fails in its
    gather / scatter with a shape error (modules.py:90-96, 154-164; utils.py:66; blocks.py:257-326) -- the kernels would read and write
    out of bounds (found as a GPU memory fault by scripts/probes/api_robustness_probe.py), so every entry point checks first.  After
    reset() the new shape is fine."""
    from eventful_transformer import modules as M, policies
    from eventful_transformer.backbones import ViTBackbone
    bb = ViTBackbone(block_config=dict(dim=64, heads=4, mlp_ratio=4), depth=2, position_encoding_size=(6, 6), input_size=(6, 6),
                     block_class="EventfulBlock", has_class_token=True).eval().to(DEV)
    H.set_policies(bb, policies.TokenNormTopK, k=12)
    x4, x2 = torch.randn(4, 37, 64, device=DEV), torch.randn(2, 37, 64, device=DEV)
    with torch.inference_mode():
        bb(x4)
        with pytest.raises(RuntimeError, match="reset"):
            bb(x2)
        bb.reset()
        y = bb(x2)
        assert torch.isfinite(bb(x2 + 0.1)).all() and y.shape == (2, 37, 64)
        with pytest.raises(RuntimeError, match="reset"):
            bb(x4)
        bb.reset()
        for n_bad in (30, 40):   # the position encoding is sized for 37 tokens
            with pytest.raises(RuntimeError, match="PositionEncoding"):
                bb(torch.randn(2, n_bad, 64, device=DEV))
        # a windowed / rel-pos block lays its tokens out on input_size
        params = O.make_block_params(64, 4, seed=3, std=0.08, rel_sizes=(3, 3), head_dim=16)
        blk = H.product_block("EventfulTokenwiseBlock", params, 64, 4, (6, 6), window_size=(3, 3), relative_embedding_size=(8, 8))
        H.set_policies(blk, policies.TokenNormTopK, k=12)
        with pytest.raises(RuntimeError, match="input_size"):
            blk(torch.randn(2, 35, 64, device=DEV))
        blk(torch.randn(2, 36, 64, device=DEV))
        # stand-alone modules
        gate = M.TokenGate()
        gate.policy = policies.TokenNormTopK(8)
        gate(torch.randn(4, 50, 64, device=DEV))
        with pytest.raises(RuntimeError, match="reset"):
            gate(torch.randn(2, 50, 64, device=DEV))
        with pytest.raises(RuntimeError, match="reset"):
            gate(torch.randn(4, 51, 64, device=DEV))
        dgate = M.TokenDeltaGate(structure="col")
        dgate.policy = policies.TokenNormTopK(8)
        dgate(torch.randn(2, 3, 20, 20, device=DEV))
        with pytest.raises(RuntimeError, match="reset"):
            dgate(torch.randn(2, 3, 20, 21, device=DEV), forced_index=torch.arange(8, device=DEV).expand(2, 8))
        buf = M.TokenBuffer()
        buf(torch.randn(4, 50, 64, device=DEV), None)
        with pytest.raises(RuntimeError, match="reset"):
            buf(torch.randn(2, 8, 64, device=DEV), torch.zeros(2, 8, dtype=torch.long, device=DEV))
        with pytest.raises(RuntimeError, match="reset"):
            buf(torch.randn(4, 9, 64, device=DEV), torch.zeros(4, 8, dtype=torch.long, device=DEV))
        mb = M.MatmulBuffer()
        mb(torch.randn(2, 3, 20, 16, device=DEV), torch.randn(2, 3, 16, 20, device=DEV), None, None)
        ix = torch.arange(5, device=DEV).expand(2, 5)
        with pytest.raises(RuntimeError, match="reset"):
            mb(torch.randn(2, 3, 21, 16, device=DEV), torch.randn(2, 3, 16, 20, device=DEV), ix, ix)
        acc = M.MatmulDeltaAccumulator()
        acc(torch.rand(2, 3, 20, 20, device=DEV), torch.randn(2, 3, 20, 16, device=DEV), None, None)
        with pytest.raises(RuntimeError, match="reset"):
            acc(torch.rand(2, 3, 21, 5, device=DEV), torch.randn(2, 3, 5, 16, device=DEV), torch.rand(2, 3, 21, 5, device=DEV), torch.randn(2, 3, 5, 16, device=DEV))


def test_a_model_built_under_inference_mode_runs():
    """Weights that are inference tensors (the model constructed or loaded inside torch.inference_mode()) track no version counter;
    the weight-plane caches key them by address alone instead of raising."""
    from eventful_transformer import policies
    params = O.make_block_params(64, 4, seed=3, std=0.08)
    xs = O.make_token_stream(2, 37, 64, 3, 12, seed=4, small=0.02).to(DEV)
    outside = H.product_block("EventfulBlock", params, 64, 4, (6, 6))
    H.set_policies(outside, policies.TokenNormTopK, k=12)
    with torch.inference_mode():
        inside = H.product_block("EventfulBlock", params, 64, 4, (6, 6))
        H.set_policies(inside, policies.TokenNormTopK, k=12)
        assert next(inside.parameters()).is_inference()
        for t in range(3):
            assert torch.equal(inside(xs[t]), outside(xs[t]))


def test_host_threads_driving_models_on_their_own_streams():
    """Four host threads, each running its own ViViT-B backbone (bf16 / fp32 / fp16 casts, different batch sizes) on its own HIP stream at
    the same time: scratch buffers are pooled per host thread and every launch goes to the thread's current stream, so the results are
    bit for bit those of the same models run one after the other (three repetitions)."""
    import threading
    from eventful_transformer import policies
    sd = H.backbone_params(12, 768, 4, 41, 197)
    jobs = [("bfloat16", 32, 1), (None, 8, 2), ("float16", 16, 3), ("bfloat16", 32, 4)]
    models, data = [], []
    for cast, b, seed in jobs:
        bb = H.product_vivit(sd, cast)
        H.set_policies(bb, policies.TokenNormTopK, k=128)
        models.append(bb)
        g = torch.Generator(device=DEV).manual_seed(seed)
        xs = [torch.randn(b, 197, 768, device=DEV, generator=g)]
        for t in range(4):
            xs.append(xs[-1] + 0.25 * torch.randn(b, 197, 768, device=DEV, generator=g))
        data.append(xs)
    torch.cuda.synchronize()
    errors = []

    def run(i, out, stream):
        try:
            with torch.inference_mode():
                if stream is None:
                    models[i].reset()
                    out[i] = [models[i](x).clone() for x in data[i]]
                else:
                    with torch.cuda.stream(stream):
                        models[i].reset()
                        out[i] = [models[i](x).clone() for x in data[i]]
                    stream.synchronize()
        except Exception as e:   # surfaces in the main thread below
            errors.append((i, repr(e)))

    seq = {}
    for i in range(len(jobs)):
        run(i, seq, None)
    torch.cuda.synchronize()
    assert not errors, errors
    for rep in range(3):
        par = {}
        threads = [threading.Thread(target=run, args=(i, par, torch.cuda.Stream())) for i in range(len(jobs))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for i in range(len(jobs)):
            assert all(torch.equal(a, b) for a, b in zip(seq[i], par[i])), (rep, i)


def test_converted_or_misplaced_models_are_errors_not_memory_faults():
    """A model converted with .half() / .bfloat16() / .double(), or left on the CPU while the frames are on the device, would hand the
    kernels pointers to other element sizes or to host memory (a GPU memory fault before round 6's guards): a RuntimeError that says what
    to do instead.  train() mode without DropPath is fine (inference kernels, no dropout)."""
    from eventful_transformer import policies
    from eventful_transformer.backbones import ViTBackbone

    def model(**kw):
        bb = ViTBackbone(block_config=dict(dim=64, heads=4, mlp_ratio=4, **kw), depth=2, position_encoding_size=(6, 6), input_size=(6, 6),
                         block_class="EventfulBlock", has_class_token=False).eval().to(DEV)
        H.set_policies(bb, policies.TokenNormTopK, k=12)
        return bb

    x = torch.randn(2, 36, 64, device=DEV)
    with torch.inference_mode():
        want = model()(x)
        for conv in (lambda m: m.half(), lambda m: m.bfloat16(), lambda m: m.double()):
            with pytest.raises(RuntimeError, match="float32"):
                conv(model())(x)
        with pytest.raises(RuntimeError, match="float32"):
            conv(model(relative_embedding_size=(6, 6))).blocks[0].relative_position.tables()
        with pytest.raises(RuntimeError, match="HIP device"):
            model().cpu()(x)
        with pytest.raises(RuntimeError, match="HIP device"):
            model()(x.cpu())
        assert torch.isfinite(model().train()(x)).all() and want.shape == (2, 36, 64)
