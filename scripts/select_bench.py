#!/usr/bin/env python3
"""Graph-replayed timing of the stand-alone selection kernel (one workgroup per clip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for N, k, thr, parts in ((1764, 256, None, 0), (1764, 256, None, 12), (4096, 0, 0.9, 0), (4096, 0, 0.9, 12), (197, 128, None, 0), (2048, 512, None, 0),
                         (4096, 400, None, 0)):
    norms = torch.rand(1, N, parts, device=dev, generator=g) if parts else torch.rand(1, N, device=dev, generator=g)
    idx = torch.empty(1, N if thr is not None else k, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    rest = torch.empty(1, N, dtype=torch.int32, device=dev)
    fn = (lambda: n.select_threshold(norms, 1, N, thr, N, idx, cnt, rest, parts=parts)) if thr is not None else (lambda: n.select_topk(norms, 1, N, k, idx, rest, parts=parts))
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(40):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f"select N={N} k={k} thr={thr} parts={parts}: {s.elapsed_time(e) * 1e3 / 400:6.2f} us per launch (incl. ~1.5 us boundary)", flush=True)
