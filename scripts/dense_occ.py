#!/usr/bin/env python3
"""K8 (evt_attention_dense) launch time vs the number of windows: finds how many workgroups run at once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
D, H, N = 768, 12, 196
ry = torch.randn(14, 14, 64, device=dev, generator=g) * 0.2
rx = torch.randn(14, 14, 64, device=dev, generator=g) * 0.2
for G in (3, 6, 9, 12, 18, 25, 27):
    qkv = torch.randn(G, N, 3 * D, device=dev, generator=g)
    out = torch.empty(G, N, D, device=dev)
    fn = lambda: n.attention_dense(qkv, G, H, N, D, 8.0, 0, out_f32=out, rel_y=ry, rel_x=rx, gh=14, gw=14, qw=14)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f"G={G:3d} workgroups={7 * G * H:5d}  {s.elapsed_time(e) * 1e3 / 200:7.1f} us per launch (incl. boundary)", flush=True)
