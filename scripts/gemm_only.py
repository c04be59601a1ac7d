#!/usr/bin/env python3
"""Runs ONE gated-linear shape repeatedly (for rocprofv3 --pmc passes on the GEMM alone).
   python scripts/gemm_only.py [M=32768] [K=768] [N=2304] [iters=30] [act=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n
M, K, N, iters, act = [int(x) for x in (sys.argv[1:] + ["32768", "768", "2304", "30", "0"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev); S = n.split_weight(W)
for _ in range(iters):
    n.gated_linear(A, K, None, M, W, b, out, N, None, M, None, None, 1, M, K, N, act=act, W_split=S)
torch.cuda.synchronize()
