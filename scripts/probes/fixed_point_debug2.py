#!/usr/bin/env python3
"""Debug: constant input, fp32 mode -- which state of block 0 still moves from frame to frame?"""
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
b0 = bb.blocks[0]
def snap():
    return dict(qkv_buf=b0.qkv_accumulator.b.clone(), qkv_p=b0.qkv_gate.p.clone(), v_p=b0.v_gate._state.clone(), a_p=b0.matmul_gate.p.clone(),
                av=b0.matmul_accumulator_2._state.clone(), proj_p=b0.projection_gate.p.clone(), proj_buf=b0.projection_accumulator.b.clone(),
                mlp_p=b0.mlp_gate.p.clone(), mlp_buf=b0.mlp_accumulator.b.clone())
with torch.inference_mode():
    bb.reset()
    for x in frames:
        bb(x)
    prev = None
    for t in range(12):
        bb(const)
        cur = snap()
        if prev is not None:
            print(t, {k: f"{float((cur[k].float() - prev[k].float()).abs().max()):.2e}" for k in cur}, flush=True)
        prev = cur
