#!/usr/bin/env python3
"""One-stream gated linears with HOT vs COLD weights, graph-replayed (no host launch cost): 40 launches per graph, either all
with the same split-weight planes or each with its own (40 x 7-19 MB: colder than L2 / MALL between replays)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n

dev = torch.device("cuda", 0)
D, H = 768, 12
g = torch.Generator(device=dev).manual_seed(0)
for name, N, k, thr in (("672", 1764, 256, False), ("1024", 4096, 400, True)):
    cap = N if thr else k
    idx = torch.zeros(1, cap, dtype=torch.int32, device=dev)
    idx[0, :k] = torch.randperm(N, device=dev, generator=g)[:k].sort()[0].int()
    count = torch.full((1,), k, dtype=torch.int32, device=dev) if thr else None
    c = torch.randn(1, N, D, device=dev, generator=g)
    p = torch.randn(1, N, D, device=dev, generator=g)
    shapes = {"qkv": (3 * D, D), "proj": (D, D), "mlp1": (4 * D, D), "mlp2": (D, 4 * D)}
    for kn, (no, ki) in shapes.items():
        W = torch.randn(no, ki, device=dev, generator=g) * 0.02
        bias = torch.zeros(no, device=dev)
        sets = [n.split_weight(torch.randn(no, ki, device=dev, generator=g) * 0.02) for _ in range(40)]
        a_in = c if ki == D else torch.randn(1, N, ki, device=dev, generator=g)
        pp = p if ki == D else torch.randn(1, N, ki, device=dev, generator=g)
        out = torch.empty(1, N, no, device=dev)
        def launch(ws):
            n.gated_linear(a_in, ki, idx, N, W, bias, out, no, idx, N, count, pp, 1, cap, ki, no, W_split=ws)
        res = {}
        for mode in ("hot", "cold"):
            for ws in sets[:2]:
                launch(ws)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(40):
                    launch(sets[0] if mode == "hot" else sets[i])
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                gr.replay()
            e.record()
            torch.cuda.synchronize()
            res[mode] = s.elapsed_time(e) * 1e3 / (20 * 40)
        # does a prefetch through arbitrary XCDs (memory-side cache only) help?  pairs [evt_prefetch(X); gemm(W_i)] with X = the NEXT
        # launch's planes (they arrive one pair early) vs X = an unrelated buffer of the same size
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        others = [torch.empty_like(sets[0]) for _ in range(40)]
        for mode in ("prefetched", "unrelated"):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(40):
                    n.prefetch(sets[(i + 1) % 40] if mode == "prefetched" else others[i], sink)
                    launch(sets[i])
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                gr.replay()
            e.record()
            torch.cuda.synchronize()
            res[mode] = s.elapsed_time(e) * 1e3 / (20 * 40)
        print(f"{name:5s} {kn:5s} M={k:4d}  hot {res['hot']:6.2f} us   cold {res['cold']:6.2f} us per launch;  prefetch + launch: next planes {res['prefetched']:6.2f}  unrelated buffer {res['unrelated']:6.2f}", flush=True)
