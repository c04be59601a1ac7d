#!/usr/bin/env python3
"""Probe: a long stream (300 gated frames, no reset) of one ViT-B EventfulBlock against the oracle, fp32 -- does the difference grow?  And mixed
policies per gate (top-k / threshold / top-fraction)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
torch.set_num_threads(8)
N, dim, heads, k = 197, 768, 12, 128
params = O.make_block_params(dim, 4, seed=9, std=0.02, head_dim=64)
for cast in (None,):
    kw = dict(matmul_2_cast=cast) if cast else {}
    ob = O.BlockOracle("EventfulBlock", params, dim, heads, (1, N), **kw); ob.set_policy(lambda: O.TopK(k))
    blk = H.product_block("EventfulBlock", params, dim, heads, (1, N), **kw); H.set_policies(blk, policies.TokenNormTopK, k=k)
    xs = O.make_token_stream(1, N, dim, 301, k, seed=10, small=0.01)
    errs = []
    with torch.inference_mode():
        for t in range(301):
            errs.append(float((blk(xs[t].cuda()).cpu() - ob.forward(xs[t])).abs().max()))
    print(f"long stream cast {cast}: err at frames 0/10/50/100/200/300: {[('%.1e' % errs[i]) for i in (0, 10, 50, 100, 200, 300)]}, max {max(errs):.1e}", flush=True)
# mixed policies per gate
ob = O.BlockOracle("EventfulBlock", params, dim, heads, (1, N)); blk = H.product_block("EventfulBlock", params, dim, heads, (1, N))
ob.policy = {"qkv_gate": O.TopK(100), "projection_gate": O.Threshold(0.05), "mlp_gate": O.TopFraction(0.3), "v_gate": O.TopK(100), "matmul_gate": O.TopK(100)}
blk.qkv_gate.policy = policies.TokenNormTopK(100); blk.projection_gate.policy = policies.TokenNormThreshold(0.05); blk.mlp_gate.policy = policies.TokenNormTopFraction(0.3)
xs = O.make_token_stream(1, N, dim, 6, 100, seed=11, small=0.01)
with torch.inference_mode():
    errs = [float((blk(xs[t].cuda()).cpu() - ob.forward(xs[t])).abs().max()) for t in range(6)]
print("mixed policies per gate (top-k 100 / threshold 0.05 / fraction 0.3), fp32:", ['%.1e' % e for e in errs], flush=True)
