#!/usr/bin/env python3
"""Probe: pooled EventfulBlock with B clips per launch vs the oracle run per clip (B = 1 each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies, blocks as EB
torch.set_num_threads(8)
D = 768
for kind, isz, pool, cast, B in (("EventfulBlock", (18, 18), 3, None, 3), ("EventfulBlock", (18, 18), 3, None, 1), ("EventfulBlock", (12, 12), 2, None, 3), ("EventfulBlock", (12, 12), 2, None, 2)):
    n = isz[0] * isz[1]; k = n // 3
    kw = dict(pool_size=pool)
    params = O.make_block_params(D, 4, seed=n, std=0.02, head_dim=64)
    blk = H.product_block(kind, params, D, 12, isz, **kw)
    H.set_policies(blk, policies.TokenNormTopK, k=k)
    xs = O.make_token_stream(3, n, D, 3, k, seed=n + 1, small=0.01)[:, :B]
    oracles = []
    for b in range(B):
        ob = O.BlockOracle(kind, params, D, 12, isz, **kw); ob.set_policy(lambda: O.TopK(k)); oracles.append(ob)
    with torch.inference_mode():
        for t in range(3):
            seen = {}
            EB.INDEX_TAP = lambda b_, tag, idx, count: seen.__setitem__(tag, idx.clone())
            y = blk(xs[t].cuda()).cpu()
            EB.INDEX_TAP = None
            row = []
            for b, ob in enumerate(oracles):
                yr = ob.forward(xs[t][b:b + 1])
                same = "" if t == 0 else "".join("=" if torch.equal(ob.trace[tag + "_index"].reshape(-1).sort()[0], seen[tag][b].long().cpu()) else "X" for tag in ("qkv", "projection", "mlp"))
                row.append(f"{float((y[b:b + 1] - yr).abs().max()):.1e}{same}")
            print(f"{kind} {isz} pool {pool} B={B} frame {t}: per clip err + gate sets(qkv,proj,mlp) {row}", flush=True)
