#!/usr/bin/env python3
"""Debug: constant input, fp32 mode, full size -- which gate keeps selecting non-trivial sets, and how large is the frame-to-frame motion?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies, blocks as EB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cast = None if len(sys.argv) < 3 or sys.argv[2] == "none" else sys.argv[2]
N, D, K = 197, 768, 128
sd = H.backbone_params(12, D, 4, 41, N)
bb = H.product_vivit(sd, cast)
H.set_policies(bb, policies.TokenNormTopK, k=K)
g = torch.Generator(device="cuda").manual_seed(3)
tok = lambda: torch.randn(B, N, D, device="cuda", generator=g)
frames = [tok(), tok(), tok()]
const = frames[-1]
seen = []
EB.INDEX_TAP = lambda blk, tag, idx, count: seen.append((blk, tag, idx.clone()))
ar = torch.arange(K, device="cuda", dtype=torch.int32)
with torch.inference_mode():
    bb.reset()
    for x in frames:
        y = bb(x).clone()
    for t in range(60):
        seen.clear()
        y2 = bb(const).clone()
        d = float((y2 - y).abs().max())
        nontriv = [(list(bb.blocks).index(blk), tag, int((idx != ar).any(dim=1).sum())) for blk, tag, idx in seen if bool((idx != ar).any())]
        print(t, f"max|dy| {d:.3e}", "non-trivial lists:", nontriv[:6], "..." if len(nontriv) > 6 else "", flush=True)
        y = y2
