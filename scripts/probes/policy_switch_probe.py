#!/usr/bin/env python3
"""Probe: the policy changed in the middle of a clip (no reset) -- top-k 128 -> top-k 40 -> threshold -> top-k 197 -> top-k 128 -- against the oracle doing the same."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies, blocks as EB
torch.set_num_threads(8)
N, dim, heads = 197, 768, 12
for cast in (None, "bfloat16"):
    params = O.make_block_params(dim, 4, seed=9, std=0.02, head_dim=64)
    kw = dict(matmul_2_cast=cast) if cast else {}
    ob = O.BlockOracle("EventfulBlock", params, dim, heads, (1, N), **kw)
    blk = H.product_block("EventfulBlock", params, dim, heads, (1, N), **kw)
    xs = O.make_token_stream(1, N, dim, 11, 60, seed=10, small=0.01)
    plan = [("topk", 128)] * 3 + [("topk", 40)] * 2 + [("thr", 0.5)] * 2 + [("topk", 197)] * 2 + [("topk", 128)] * 2
    errs = []
    with torch.inference_mode():
        for t, (kind, val) in enumerate(plan):
            if kind == "topk":
                ob.set_policy(lambda: O.TopK(val)); H.set_policies(blk, policies.TokenNormTopK, k=val)
            else:
                ob.set_policy(lambda: O.Threshold(val)); H.set_policies(blk, policies.TokenNormThreshold, threshold=val)
            seen = {}
            EB.INDEX_TAP = lambda b_, tag, idx, count: seen.__setitem__(tag, (idx.clone(), None if count is None else count.clone()))
            y_ref = ob.forward(xs[t])
            y = blk(xs[t].cuda()).cpu()
            EB.INDEX_TAP = None
            errs.append(float((y - y_ref).abs().max()))
            if t:
                for tag in ("qkv", "projection", "mlp"):
                    want = ob.trace[tag + "_index"].reshape(-1).sort()[0]
                    idx, cnt = seen[tag]
                    got = idx[0, :int(cnt[0])].long().cpu() if cnt is not None else idx[0].long().cpu()
                    if not torch.equal(want, got):
                        print(f"   frame {t} {tag}: sets differ ({len(want)} vs {len(got)} tokens, {len(set(want.tolist()) ^ set(got.tolist()))} in the symmetric difference)")
    print(f"cast {cast}: err per frame {['%.1e' % e for e in errs]}", flush=True)
