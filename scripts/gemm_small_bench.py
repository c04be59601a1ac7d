#!/usr/bin/env python3
"""Graph-replayed timing of the small-M gated linears (no host overhead): hot weights (one layer repeated) vs cold weights
(12 layers in rotation, as in a backbone).  python scripts/gemm_small_bench.py [--m 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=256)
ap.add_argument("--tokens", type=int, default=1764)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--count", action="store_true", help="threshold-policy form: capacity = all tokens, the row count on the device")
a = ap.parse_args()
dev = torch.device("cuda", 0)
D, N, k, L = 768, a.tokens, a.m, a.layers
g = torch.Generator(device=dev).manual_seed(0)
c = torch.randn(1, N, D, device=dev, generator=g)
p = torch.randn(1, N, D, device=dev, generator=g)
idx = torch.randperm(N, device=dev, generator=g)[:k].sort()[0].int().view(1, k).contiguous()
count, cap = None, k
if a.count:
    cap = N
    idx = torch.cat([idx, torch.zeros(1, N - k, dtype=torch.int32, device=dev)], dim=1).contiguous()
    count = torch.full((1,), k, dtype=torch.int32, device=dev)
shapes = {"qkv": (D, 3 * D), "proj": (D, D), "mlp1": (D, 4 * D), "mlp2": (4 * D, D)}
hidden = torch.randn(cap, 4 * D, device=dev, generator=g)
for name, (K, Nout) in shapes.items():
    Ws = [torch.randn(Nout, K, device=dev, generator=g) * 0.02 for _ in range(L)]
    Ss = [n.split_weight(w) for w in Ws]
    bias = torch.zeros(Nout, device=dev)
    out = torch.empty(1, N, Nout, device=dev)
    A, lda, a_rows, aidx = (c, D, N, idx) if K == D else (hidden, K, cap, None)
    act = n.ACT_GELU if name == "mlp1" else n.ACT_NONE

    def launch(l):
        n.gated_linear(A, lda, aidx, a_rows, Ws[l], bias, out, Nout, idx, N, count, p if aidx is not None else None, 1, cap, K, Nout,
                       act=act, W_split=Ss[l])
    res = {}
    for mode in ("hot", "cold"):
        launch(0)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(2 * L):
                launch(0 if mode == "hot" else i % L)
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            gr.replay()
        e.record()
        torch.cuda.synchronize()
        res[mode] = s.elapsed_time(e) * 1e3 / (10 * 2 * L)
    print(f"{name:5s} M={k}{' (device count, capacity ' + str(cap) + ')' if a.count else ''} K={K} Nout={Nout}: hot {res['hot']:6.1f} us  cold {res['cold']:6.1f} us per launch (incl. ~1.5 us boundary)", flush=True)
