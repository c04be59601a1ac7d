#!/usr/bin/env python3
"""t(K) = overhead + k-tiles x slope for the gated-linear kernel: dense M x K @ (N x K)^T at M = 32768.
   EVT_GEMM_PP=0|2 python scripts/gemm_k_sweep.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n
dev = torch.device("cuda", 0)
M = 32768
for N in [int(x) for x in sys.argv[1:]] or [2304, 768]:
    res = []
    for K in (768, 1536, 3072):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev); S = n.split_weight(W)
        f = lambda: n.gated_linear(A, K, None, M, W, b, out, N, None, M, None, None, 1, M, K, N, W_split=S)
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        res.append((K, us))
        print(f"N={N} K={K}: {us:8.1f} us  {2.0 * M * K * N / us * 1e-6:7.1f} TF")
    slope = (res[2][1] - res[0][1]) / (96 - 24)
    print(f"  -> per 32-wide k-tile {slope:.3f} us per launch; overhead (K -> 0) {res[0][1] - 24 * slope:.1f} us = {100 * (res[0][1] - 24 * slope) / res[0][1]:.0f} % of the K=768 launch")
