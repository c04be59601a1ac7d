#!/usr/bin/env python3
"""Probe: policy edge cases through a block -- top-k with k = 0, k = N, k > N; fraction 0 / 1; threshold 0 / inf; order 1 / inf."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
for dim, heads, N, cast in ((64, 4, 37, None), (768, 12, 197, "bfloat16"), (768, 12, 300, None)):
    params = O.make_block_params(dim, 4, seed=N, std=0.05, head_dim=dim // heads)
    kw = dict(matmul_2_cast=cast) if cast else {}
    xs = O.make_token_stream(1, N, dim, 3, max(1, N // 3), seed=N + 1, small=0.01)
    cases = [("TopK k=0", policies.TokenNormTopK, dict(k=0), lambda: O.TopK(0)),
             ("TopK k=N", policies.TokenNormTopK, dict(k=N), lambda: O.TopK(N)),
             ("TopK k=N+5", policies.TokenNormTopK, dict(k=N + 5), None),
             ("TopFraction 0.0", policies.TokenNormTopFraction, dict(fraction=0.0), lambda: O.TopFraction(0.0)),
             ("TopFraction 1.0", policies.TokenNormTopFraction, dict(fraction=1.0), lambda: O.TopFraction(1.0)),
             ("TopFraction 0.37", policies.TokenNormTopFraction, dict(fraction=0.37), lambda: O.TopFraction(0.37)),
             ("Threshold 0", policies.TokenNormThreshold, dict(threshold=0.0), lambda: O.Threshold(0.0)),
             ("Threshold inf", policies.TokenNormThreshold, dict(threshold=float("inf")), lambda: O.Threshold(float("inf"))),
             ("TopK order=1", policies.TokenNormTopK, dict(k=N // 3, order=1), lambda: O.TopK(N // 3, order=1)),
             ("TopK order=inf", policies.TokenNormTopK, dict(k=N // 3, order=float("inf")), lambda: O.TopK(N // 3, order=float("inf")))]
    for name, pc, pkw, ofac in cases:
        try:
            blk = H.product_block("EventfulBlock", params, dim, heads, (1, N), **kw)
            H.set_policies(blk, pc, **pkw)
            ob = None
            if ofac is not None:
                ob = O.BlockOracle("EventfulBlock", params, dim, heads, (1, N), **kw)
                ob.set_policy(ofac)
            errs = []
            with torch.inference_mode():
                for t in range(3):
                    y = blk(xs[t].cuda()).cpu()
                    if ob is not None:
                        errs.append(float((y - ob.forward(xs[t])).abs().max()))
            print(f"dim {dim} N {N} {str(cast):9s} {name:16s} ok, finite {bool(torch.isfinite(y).all())}, err vs oracle {['%.1e' % e for e in errs]}", flush=True)
        except Exception as e:
            print(f"dim {dim} N {N} {str(cast):9s} {name:16s} RAISED {type(e).__name__}: {str(e)[:150]}", flush=True)
