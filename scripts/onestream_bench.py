#!/usr/bin/env python3
"""Micro-benchmark of the one-stream (ViTDet) kernels at their real shapes: B = 1, D = 768, H = 12.
  python scripts/onestream_bench.py [--only stream,select,...]      (EVT_LIB=<other build> to A/B)
Each kernel is timed in a chain of `iters` back-to-back launches after a 150 ms spin (warm clock)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("EVT_PKG_ROOT") or os.path.join(ROOT, "eventful-transformer_amd"))   # EVT_PKG_ROOT: another tree's host package (A/B across an ABI change)
import torch
from eventful_transformer import _native as n


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=1, help="streams per launch (the attention kernels only: --only stream,stream_first)")
    a = ap.parse_args()
    Bn = a.batch
    only = set(a.only.split(",")) if a.only else None
    dev = torch.device("cuda", 0)
    D, H, dh = 768, 12, 64
    g = torch.Generator(device=dev).manual_seed(0)
    out = {}
    for name, N, gw, k, cast, thr in (("672", 1764, 42, 256, None, False), ("1024", 4096, 64, 400, "bfloat16", True)):
        sdt = torch.float32 if cast is None else getattr(torch, cast)
        store = n.store_code(sdt)
        gh = N // gw
        qkv = torch.randn(Bn, N, 3 * D, device=dev, generator=g)
        ry = torch.randn(gh, gh, dh, device=dev, generator=g) * 0.2
        rx = torch.randn(gw, gw, dh, device=dev, generator=g) * 0.2
        terms = torch.empty(Bn, H, N, gh + gw, device=dev)
        n.rel_terms(qkv, ry, rx, Bn, H, N, D, gh, gw, gw, terms)
        cap = N if thr else k
        idx = torch.zeros(Bn, cap, dtype=torch.int32, device=dev)
        for bi in range(Bn):
            idx[bi, :k] = torch.randperm(N, device=dev, generator=g)[:k].sort()[0].int()
        count = torch.full((Bn,), k, dtype=torch.int32, device=dev) if thr else None
        apT = torch.rand(Bn, H, N, N, device=dev, generator=g).to(sdt)
        vp = torch.randn(Bn, N, D, device=dev, generator=g).to(sdt)
        pv = torch.randn(Bn, N, D, device=dev, generator=g).to(sdt)
        o32 = torch.empty(Bn, N, D, device=dev)
        vd = torch.zeros(Bn, D, cap, device=dev, dtype=sdt)
        vo = torch.zeros(Bn, D, cap, device=dev, dtype=sdt)
        pref = torch.randn(Bn, N, D, device=dev, generator=g)
        parts = torch.empty(Bn, N, H, device=dev)
        norms = torch.rand(1, N, device=dev, generator=g)
        sel = torch.empty(1, cap, dtype=torch.int32, device=dev)
        rest = torch.empty(1, N, dtype=torch.int32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        x = torch.randn(1, N, D, device=dev, generator=g)
        p = torch.randn(1, N, D, device=dev, generator=g)
        c = torch.empty_like(x)
        lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        nrm = torch.empty(N, device=dev)
        Wq = torch.randn(3 * D, D, device=dev, generator=g) * 0.02
        W1 = torch.randn(4 * D, D, device=dev, generator=g) * 0.02
        W2 = torch.randn(D, 4 * D, device=dev, generator=g) * 0.02
        Wp = torch.randn(D, D, device=dev, generator=g) * 0.02
        sq, s1, s2, sp = (n.split_weight(w) for w in (Wq, W1, W2, Wp))
        b3, b4, b1 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)
        qkv_out = torch.empty(1, N, 3 * D, device=dev)
        o1 = torch.empty(1, N, D, device=dev)
        hidden = torch.empty(cap, 4 * D, device=dev)
        wmap_blocks = None
        # the windowed blocks' attention (K8 resident): the frame's 14 x 14 windows (672^2: 3 x 3; 1024^2: 5 x 5 on the padded 70 x 70 grid), rel-pos, fp32
        wG = 9 if name == "672" else 25
        wqkv = torch.randn(wG, 196, 3 * D, device=dev, generator=g)
        wout = torch.empty(wG, 196, D, device=dev)
        wry = torch.randn(14, 14, dh, device=dev, generator=g) * 0.2
        wrx = torch.randn(14, 14, dh, device=dev, generator=g) * 0.2
        kernels = {
            "stream": lambda: n.attention_stream(qkv, apT, pv, Bn, H, N, D, 8.0, store, False, rel_terms=terms, gh=gh, gw=gw, idx=idx,
                                                 count=count, kcap=cap, v_delta_t=vd, v_old_t=vo, out_f32=o32, norm_ref=pref, norm_parts=parts),
            "stream_first": lambda: n.attention_stream(qkv, apT, pv, Bn, H, N, D, 8.0, store, True, rel_terms=terms, gh=gh, gw=gw,
                                                       v_state=vp, out_f32=o32),
            "window": lambda: n.attention_dense(wqkv, wG, H, 196, D, 8.0, n.EVT_F32, out_f32=wout, rel_y=wry, rel_x=wrx, gh=14, gw=14, qw=14),
            "rel_terms": lambda: n.rel_terms(qkv, ry, rx, 1, H, N, D, gh, gw, gw, terms),
            "v_gate_t": lambda: n.v_gate(qkv, idx, count, 1, N, D, cap, vp, vd, vo, store, True, transposed=True),
            "select": (lambda: n.select_threshold(norms, 1, N, 0.9, cap, sel, cnt, rest)) if thr else (lambda: n.select_topk(norms, 1, N, k, sel, rest)),
            "row_pass": lambda: n.row_pass(x, N, D, res=p, sum_out=o1, ln_w=lw, ln_b=lb, c_out=c, p=p, norms=nrm),
            "qkv": lambda: n.gated_linear(c, D, idx, N, Wq, b3, qkv_out, 3 * D, idx, N, count, p, 1, cap, D, 3 * D, W_split=sq),
            "proj": lambda: n.gated_linear(c, D, idx, N, Wp, b1, o1, D, idx, N, count, p, 1, cap, D, D, W_split=sp),
            "mlp": lambda: n.gated_mlp(c, D, idx, N, W1, b4, W2, b1, hidden, o1, D, count, p, 1, cap, D, 4 * D, W1_split=s1, W2_split=s2),
        }
        one = torch.zeros(64, device=dev)
        kernels["null"] = lambda: one.add_(1.0)   # launch floor of a back-to-back chain (one 64-element ATen kernel)
        if Bn == 1:   # the same QKV / MLP launches cycling through 40 different weight sets (~1.1 GB: colder than any cache)
            cold = [(n.split_weight(torch.randn(3 * D, D, device=dev, generator=g) * 0.02), n.split_weight(torch.randn(4 * D, D, device=dev, generator=g) * 0.02),
                     n.split_weight(torch.randn(D, 4 * D, device=dev, generator=g) * 0.02)) for _ in range(40)]
            turn = [0]
            def qkv_cold():
                turn[0] = (turn[0] + 1) % 40
                n.gated_linear(c, D, idx, N, Wq, b3, qkv_out, 3 * D, idx, N, count, p, 1, cap, D, 3 * D, W_split=cold[turn[0]][0])
            def mlp_cold():
                turn[0] = (turn[0] + 1) % 40
                n.gated_mlp(c, D, idx, N, W1, b4, W2, b1, hidden, o1, D, count, p, 1, cap, D, 4 * D, W1_split=cold[turn[0]][1], W2_split=cold[turn[0]][2])
            kernels["qkv_cold_weights"] = qkv_cold
            kernels["mlp_cold_weights"] = mlp_cold
        for kn, fn in kernels.items():
            if (only and kn not in only) or (Bn > 1 and not kn.startswith("stream")):
                continue
            us = timeit(fn)
            out[f"{name}.{kn}"] = round(us, 1)
            print(f"{name:5s} {kn:14s} {us:8.1f} us", flush=True)
            if kn.startswith("stream") and hasattr(n.load(), "evt_debug_prof_stream"):
                import ctypes
                buf = (ctypes.c_ulonglong * 16)()
                torch.cuda.synchronize()
                n.load().evt_debug_prof_stream(buf)
                names = ["prologue", "passA", "combine", "passB-rest", "epilogue", "B:scores+gate", "B:Vstage", "B:bar1", "B:sweep", "B:bar2"]
                tot = sum(buf[:10]) or 1
                print("      phase ticks (wave 0 of workgroup 100, 100 MHz s_memtime... units as read): " +
                      ", ".join(f"{nm} {buf[i]} ({100 * buf[i] // tot}%)" for i, nm in enumerate(names)), flush=True)
    return out


if __name__ == "__main__":
    main()
