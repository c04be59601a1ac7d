#!/usr/bin/env python3
"""Does the prefetch rider of the select launch run, and what does it cost / give?  Graph-replayed chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, k, D = 1764, 256, 768
norms = torch.rand(1, N, device=dev, generator=g)
idx = torch.empty(1, k, dtype=torch.int32, device=dev)
c = torch.randn(1, N, D, device=dev, generator=g)
p = torch.randn(1, N, D, device=dev, generator=g)
W = torch.randn(3 * D, D, device=dev, generator=g) * 0.02
bias = torch.zeros(3 * D, device=dev)
out = torch.empty(1, N, 3 * D, device=dev)
sets = [n.split_weight(torch.randn(3 * D, D, device=dev, generator=g) * 0.02) for _ in range(40)]
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)


def run(body, reps=40):
    body(0); body(1)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(reps):
            body(i)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        gr.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (20 * reps)


def sel(i, rider=None):
    if rider is not None:
        n.select_prefetch_next(rider)
    n.select_topk(norms, 1, N, k, idx, None)


def gemm(i):
    n.gated_linear(c, D, idx, N, W, bias, out, 3 * D, idx, N, None, p, 1, k, D, 3 * D, W_split=sets[i % 40])


print("select alone                        %.2f us" % run(lambda i: sel(i)))
print("select + rider (7 MB planes)        %.2f us" % run(lambda i: sel(i, sets[i % 40])))
print("select + rider (64 MB buffer)       %.2f us" % run(lambda i: sel(i, big)))
print("select ; gemm(cold planes)          %.2f us per pair" % run(lambda i: (sel(i), gemm(i))))
print("select+rider(planes i) ; gemm(i)    %.2f us per pair" % run(lambda i: (sel(i, sets[i % 40]), gemm(i))))
print("select+rider(planes i+1) ; gemm(i)  %.2f us per pair" % run(lambda i: (sel(i, sets[(i + 1) % 40]), gemm(i))))
