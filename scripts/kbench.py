#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the headline shapes (ViViT-B, B clips, k=128, bf16 A.v cast).
Usage: python scripts/kbench.py [--clips 64] [--only name,name] [--iters 20]
Prints one line per kernel: avg us, algorithmic GB/s or TFLOP/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch  # noqa: E402

from eventful_transformer import _native as n  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:   # a cold process runs its first ~10 ms of kernels at a lower clock
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--cast", default="bfloat16")
    a = ap.parse_args()
    B, N, D, H, k = a.clips, a.tokens, 768, 12, a.k
    dh = D // H
    dev = torch.device("cuda", 0)
    sdt = torch.float32 if a.cast in ("none", "fp32") else getattr(torch, a.cast)
    store = n.store_code(sdt)
    es = torch.empty(0, dtype=sdt).element_size()
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, N, D, device=dev, generator=g)
    p = torch.randn(B, N, D, device=dev, generator=g)
    w = torch.randn(D, device=dev, generator=g)
    c = torch.empty_like(x)
    norms = torch.empty(B * N, device=dev)
    idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
    Wqkv = torch.randn(3 * D, D, device=dev, generator=g) * 0.02
    W1 = torch.randn(4 * D, D, device=dev, generator=g) * 0.02
    W2 = torch.randn(D, 4 * D, device=dev, generator=g) * 0.02
    Wp = torch.randn(D, D, device=dev, generator=g) * 0.02
    b3, b4, b1 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)
    qkv = torch.randn(B, N, 3 * D, device=dev, generator=g)
    buf = torch.empty(B, N, D, device=dev)
    hidden = torch.randn(B * k, 4 * D, device=dev, generator=g)
    product = torch.randn(B, H, N, N, device=dev, generator=g)
    ap_ = torch.rand(B, H, N, N, device=dev, generator=g).to(sdt)
    vp = torch.randn(B, N, D, device=dev, generator=g).to(sdt)
    pv = torch.zeros(B, N, D, device=dev, dtype=sdt)
    out = torch.empty(B, N, D, device=dev)
    vd_t = torch.empty(B, D, k, device=dev, dtype=sdt)
    vo_t = torch.empty(B, D, k, device=dev, dtype=sdt)
    vd = torch.empty(B, k, D, device=dev, dtype=sdt)
    vo = torch.empty(B, k, D, device=dev, dtype=sdt)
    nparts = torch.empty(B, N, H, device=dev)
    a_new = torch.empty(B, H, N, k, device=dev, dtype=sdt)
    a_del = torch.empty(B, H, N, k, device=dev, dtype=sdt)
    n.v_gate(qkv, idx, None, B, N, D, k, vp, vd_t, vo_t, store, True, transposed=True)
    n.v_gate(qkv, idx, None, B, N, D, k, vp, vd, vo, store, True)

    tiles = None
    if sdt != torch.float32 and n.attention_gated_fits(N, D, H, store):
        tiles = n.gated_tiles_empty(B, H, N, sdt, dev)
        n.logical_to_tiles(ap_, tiles)
    fused_bytes = B * H * N * (2 * es * k) + B * N * D * (8 + 2 * es) + B * k * D * (4 + 2 * es)   # reference columns, q / k, A.v state, value gate

    M = B * k
    Sq, Sp, S1, S2 = (n.split_weight(t) for t in (Wqkv, Wp, W1, W2))
    cases = {
        "row_pass_ln_norm": (lambda: n.row_pass(x, B * N, D, ln_w=w, ln_b=w, c_out=c, p=p, norms=norms),
                             ("GB/s", 4 * B * N * D * 3)),
        "row_pass_norm_only": (lambda: n.row_pass(x, B * N, D, p=p, norms=norms), ("GB/s", 4 * B * N * D * 2)),
        "row_pass_add": (lambda: n.row_pass(x, B * N, D, res=p, sum_out=c), ("GB/s", 4 * B * N * D * 3)),
        "select_topk": (lambda: n.select_topk(norms, B, N, k, idx), ("GB/s", 4 * B * N)),
        "linear_qkv": (lambda: n.gated_linear(x, D, idx, N, Wqkv, b3, qkv, 3 * D, idx, N, None, p, B, k, D, 3 * D, W_split=Sq),
                       ("TF", 2.0 * M * D * 3 * D)),
        "linear_qkv_nop": (lambda: n.gated_linear(x, D, idx, N, Wqkv, b3, qkv, 3 * D, idx, N, None, None, B, k, D, 3 * D, W_split=Sq),
                           ("TF", 2.0 * M * D * 3 * D)),
        "linear_qkv_gather_only": (lambda: n.gated_linear(x, D, idx, N, Wqkv, b3, qkv, 3 * D, None, k, None, None, B, k, D, 3 * D, W_split=Sq),
                                   ("TF", 2.0 * M * D * 3 * D)),
        "linear_qkv_scatter_only": (lambda: n.gated_linear(x, D, None, k, Wqkv, b3, qkv, 3 * D, idx, N, None, None, B, k, D, 3 * D, W_split=Sq),
                                    ("TF", 2.0 * M * D * 3 * D)),
        "linear_qkv_compact": (lambda: n.gated_linear(x, D, None, k, Wqkv, b3, qkv, 3 * D, None, k, None, None, B, k, D, 3 * D, W_split=Sq),
                               ("TF", 2.0 * M * D * 3 * D)),
        "linear_proj": (lambda: n.gated_linear(x, D, idx, N, Wp, b1, buf, D, idx, N, None, p, B, k, D, D, W_split=Sp),
                        ("TF", 2.0 * M * D * D)),
        "linear_proj_bf16": (lambda: n.gated_linear(vp, D, idx, N, Wp, b1, buf, D, idx, N, None, p, B, k, D, D, W_split=Sp, a_bf16=True),
                             ("TF", 2.0 * M * D * D)),
        "linear_mlp1_gelu": (lambda: n.gated_linear(x, D, idx, N, W1, b4, hidden, 4 * D, None, k, None, p, B, k, D, 4 * D, act=n.ACT_GELU,
                                                    W_split=S1), ("TF", 2.0 * M * D * 4 * D)),
        "linear_mlp2": (lambda: n.gated_linear(hidden, 4 * D, None, k, W2, b1, buf, D, idx, N, None, None, B, k, 4 * D, D, W_split=S2),
                        ("TF", 2.0 * M * D * 4 * D)),
        "mlp": (lambda: n.gated_mlp(x, D, idx, N, W1, b4, W2, b1, hidden, buf, D, None, p, B, k, D, 4 * D, W1_split=S1, W2_split=S2),
                ("TF", 4.0 * M * D * 4 * D)),
        "linear_dense_qkv": (lambda: n.gated_linear(x, D, None, B * N, Wqkv, b3, qkv, 3 * D, None, B * N, None, None,
                                                    1, B * N, D, 3 * D, W_split=Sq), ("TF", 2.0 * B * N * D * 3 * D)),
        "qk_delta": (lambda: n.qk_packed(qkv, B, N, D, H, 8.0, product, idx=idx, kcap=k),
                     ("TF", 2.0 * 2 * B * k * N * D)),
        "qk_full": (lambda: n.qk_packed(qkv, B, N, D, H, 8.0, product), ("TF", 2.0 * B * N * N * D)),
        "v_gate_t": (lambda: n.v_gate(qkv, idx, None, B, N, D, k, vp, vd_t, vo_t, store, True, transposed=True),
                     ("GB/s", B * k * D * (4 + 4 * es))),
        "softmax_av_fused": (lambda: n.softmax_av_gated(product, ap_, idx, None, k, vd_t, vo_t, pv, out, B, H, N, D, store),
                             ("GB/s", B * H * N * (4 * N + 2 * es * k) + B * N * D * (4 + 2 * es))),
        "softmax_av_fused_qk": (lambda: n.softmax_av_gated(None, ap_, idx, None, k, vd_t, vo_t, pv, out, B, H, N, D, store, qkv=qkv, scale=8.0),
                                ("GB/s", B * H * N * (2 * es * k) + B * N * D * (8 + 4 + 2 * es))),
        "softmax_av_fused_qk_norm_noout": (lambda: n.softmax_av_gated(None, ap_, idx, None, k, vd_t, vo_t, pv, None, B, H, N, D, store, qkv=qkv, scale=8.0,
                                                                      norm_ref=p, norm_parts=nparts),
                                           ("GB/s", B * H * N * (2 * es * k) + B * N * D * (8 + 4 + 2 * es))),
        "softmax_av_fused_qk_norm": (lambda: n.softmax_av_gated(None, ap_, idx, None, k, vd_t, vo_t, pv, out, B, H, N, D, store, qkv=qkv, scale=8.0,
                                                                norm_ref=p, norm_parts=nparts),
                                     ("GB/s", B * H * N * (2 * es * k) + B * N * D * (8 + 4 + 4 + 2 * es))),
        "gated_resident_norm_noout": (lambda: n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, norm_ref=p, norm_parts=nparts),
                                      ("GB/s", fused_bytes + 4 * B * N * D)),
        "gated_resident_norm": (lambda: n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, out_f32=out, norm_ref=p, norm_parts=nparts),
                                ("GB/s", fused_bytes + 8 * B * N * D)),
        "gated_resident_plain": (lambda: n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, out_f32=out),
                                 ("GB/s", fused_bytes + 4 * B * N * D)),
        "gated_resident_first": (lambda: n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, True, out_f32=out),
                                 ("GB/s", B * H * N * N * es + B * N * D * (12 + 4 + 2 * es))),
        "softmax_gated": (lambda: n.softmax_gate(product, ap_, B, H, N, N, D, store, a_new=a_new, a_delta=a_del,
                                                 idx=idx, kcap=k, gated=True),
                          ("GB/s", B * H * N * (4 * N + 4 * es * k))),
        "av_gated": (lambda: n.av(a_new, vd, k, B, H, N, k, D, store, pv=pv, out_f32=out, a2=a_del, v2=vo, gated=True),
                     ("TF", 2.0 * 2 * B * N * k * D)),
        "softmax_full": (lambda: n.softmax_gate(product, ap_, B, H, N, N, D, store), ("GB/s", B * H * N * N * (4 + es))),
        "av_full": (lambda: n.av(ap_, vp, N, B, H, N, N, D, store, pv=pv, out_f32=out), ("TF", 2.0 * B * N * N * D)),
        # K8: first frame of a clip in one launch (q.k^T state + probabilities + A.v state + output)
        "attention_dense_states": (lambda: n.attention_dense(qkv, B, H, N, D, 8.0, store, out_f32=out, product=product,
                                                             a_state=ap_, pv=pv),
                                   ("GB/s", B * H * N * N * (4 + es) + B * N * D * (12 + 4 + es))),
        "attention_dense": (lambda: n.attention_dense(qkv, B, H, N, D, 8.0, store, out_f32=out),
                            ("TF", 4.0 * B * N * N * D)),
    }
    only = [s for s in a.only.split(",") if s]
    print(f"# B={B} N={N} k={k} D={D} cast={a.cast}")
    for name, (fn, (unit, work)) in cases.items():
        if (only and name not in only) or (name.startswith("gated_resident") and tiles is None):
            continue
        us = timeit(fn, a.iters)
        rate = work / us * 1e-3 if unit == "GB/s" else work / us * 1e-6
        print(f"{name:22s} {us:9.1f} us   {rate:9.1f} {unit}")


if __name__ == "__main__":
    main()
