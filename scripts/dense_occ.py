#!/usr/bin/env python3
"""K8 (evt_attention_dense) launch time vs the number of windows, at ViTDet's window shape (196 tokens, 12 heads, rel-pos):
9 windows = 672^2, 25 = 1024^2.  EVT_DENSE_TILED=1 times the tiled kernel (evt_attn_dense.hip) instead of the resident one
(evt_attn_window.hip); EVT_WINDOW_NW=4|8 forces the resident kernel's workgroup shape."""
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
print(f"EVT_DENSE_TILED={os.environ.get('EVT_DENSE_TILED', '0')} EVT_WINDOW_NW={os.environ.get('EVT_WINDOW_NW', 'auto')}")
for G, split, store in [(g_, s_, 0) for s_ in (1, 0) for g_ in (3, 9, 12, 18, 25, 27, 64, 256)] + [(256, 1, 1)]:
    qkv = torch.randn(G, N, 3 * D, device=dev, generator=g)
    out = torch.empty(G, N, D, device=dev)
    fn = lambda: n.attention_dense(qkv, G, H, N, D, 8.0, store, out_f32=out, rel_y=ry, rel_x=rx, gh=14, gw=14, qw=14, qk_split=split)
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
    us = s.elapsed_time(e) * 1e3 / 200
    flop = 4.0 * G * H * N * N * 64
    print(f"G={G:3d} split={split} store={store} (group, head) pairs={G * H:5d}  {us:7.1f} us per launch (incl. boundary)  {flop / us * 1e-6:7.1f} TFLOP/s", flush=True)
