#!/usr/bin/env python3
"""Probe: pooled keys (pool_size) at several grids / pool sizes / store types against the oracle, with index-set comparison (fork or bug?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies, blocks as EB
torch.set_num_threads(8)
dim, heads = 768, 12
cases = [((18, 18), 3, None), ((18, 18), 2, None), ((12, 12), 3, None), ((12, 12), 2, None), ((18, 18), (3, 2), None), ((20, 20), 2, None), ((20, 20), 4, None), ((18, 18), 3, "float16"),
         ((14, 14), 2, "bfloat16"), ((14, 14), 2, None), ((42, 42), 2, None), ((42, 42), 3, None)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for isz, pool, cast in cases:
    n = isz[0] * isz[1]
    k = max(1, n // 3)
    kw = dict(pool_size=pool)
    if cast: kw["matmul_2_cast"] = cast
    try:
        params = O.make_block_params(dim, 4, seed=n, std=0.02, head_dim=64)
        ob = O.BlockOracle("EventfulBlock", params, dim, heads, isz, **kw)
        ob.set_policy(lambda: O.TopK(k))
        blk = H.product_block("EventfulBlock", params, dim, heads, isz, **kw)
        H.set_policies(blk, policies.TokenNormTopK, k=k)
        xs = O.make_token_stream(1, n, dim, 3, k, seed=n + 1, small=0.01)
        errs, notes = [], []
        with torch.inference_mode():
            for t in range(3):
                seen = {}
                EB.INDEX_TAP = lambda b_, tag, idx, count: seen.__setitem__(tag, idx.clone())
                y = blk(xs[t].cuda()).cpu()
                EB.INDEX_TAP = None
                errs.append(float((y - ob.forward(xs[t])).abs().max()))
                if t:
                    for tag in ("qkv", "projection", "mlp"):
                        if not torch.equal(ob.trace[tag + "_index"].reshape(1, -1).sort(dim=-1)[0], seen[tag].long().cpu()):
                            notes.append(f"f{t} {tag} differ")
        print(f"grid {isz} pool {pool} {cast}: err {['%.0e' % e for e in errs]} {notes}", flush=True)
    except Exception as e:
        print(f"grid {isz} pool {pool} {cast}: RAISED {type(e).__name__}: {str(e)[:150]}", flush=True)
