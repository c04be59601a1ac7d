#!/usr/bin/env python3
"""Phase timing inside the persistent gated-linear kernel (library built with -DEVT_PROF, see csrc/evt_linear_big.hip):
cycles per k-tile iteration that wave 0 (stage-first group) and wave 4 (multiply-first group) of one workgroup spend in
each phase.  Usage: EVT_LIB=<prof build> python scripts/gemm_prof.py [--clips 256] [--shape qkv|proj|mlp1|mlp2]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch  # noqa: E402

from eventful_transformer import _native as n  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--shape", default="qkv")
a = ap.parse_args()
B, N, D, k = a.clips, 197, 768, 128
K, Nout, act = {"qkv": (D, 3 * D, 0), "proj": (D, D, 0), "mlp1": (D, 4 * D, 1), "mlp2": (4 * D, D, 0), "mlp": (D, D, 0)}[a.shape]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, N, K, device=dev, generator=g)
W = torch.randn(Nout, K, device=dev, generator=g) * 0.02
b = torch.zeros(Nout, device=dev)
out = torch.empty(B, N, Nout, device=dev)
idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
S = n.split_weight(W)
if a.shape == "mlp":   # the whole gated MLP: the profile that remains is the second launch's (pre-split hidden)
    W1 = torch.randn(4 * D, D, device=dev, generator=g) * 0.02
    W2 = torch.randn(D, 4 * D, device=dev, generator=g) * 0.02
    S1, S2 = n.split_weight(W1), n.split_weight(W2)
    b4 = torch.zeros(4 * D, device=dev)
    hidden = torch.empty(B * k, 4 * D, device=dev)
    for _ in range(100):
        n.gated_mlp(x, D, idx, N, W1, b4, W2, b, hidden, out, D, None, None, B, k, D, 4 * D, W1_split=S1, W2_split=S2)
else:
  for _ in range(200):
    n.gated_linear(x, K, idx, N, W, b, out, Nout, idx, N, None, None, B, k, K, Nout, act, W_split=S)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib = n.load()
# the persistent 256-row kernel (evt_linear_pipe.hip, 8 waves); needs a -DEVT_PROF build (EVT_LIB)
lib.evt_debug_prof_pipe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.evt_debug_prof_pipe(buf) == 0
v = [buf[q] for q in range(8)]
tot = sum(v)
print(f"{a.shape}: wave 0 of workgroup 8, total {tot} ticks")
for q, nm in enumerate(["refresh + fetch-side bookkeeping", "wait: loads of the next k-tile, fragment reads", "first k-half (MFMA + staging)",
                        "barrier", "second k-half (MFMA + reads)", "epilogue"]):
    print(f"   {nm:48s} {v[q]:12d}  {100.0 * v[q] / max(tot, 1):5.1f} %")
