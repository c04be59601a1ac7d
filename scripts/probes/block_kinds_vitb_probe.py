#!/usr/bin/env python3
"""Probe: the three eventful block kinds and their options at ViT-B width (head dim 64: the fast kernels) against the CPU oracle + fixed point + r = N."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
torch.set_num_threads(8)
dim, heads = 768, 12
cases = [("EventfulTokenwiseBlock", (14, 14), True, {}), ("EventfulMatmul1Block", (14, 14), True, {}), ("EventfulBlock", (14, 14), True, {}),
         ("EventfulTokenwiseBlock", (14, 14), True, dict(stgt=True)), ("EventfulBlock", (14, 14), True, dict(gate_before_ln=True)),
         ("EventfulMatmul1Block", (14, 14), False, dict(pool_size=2)), ("EventfulBlock", (14, 14), False, dict(pool_size=2, matmul_2_cast="bfloat16")),
         ("EventfulBlock", (14, 14), False, dict(relative_embedding_size=(14, 14))), ("EventfulBlock", (16, 18), False, dict(relative_embedding_size=(8, 8), matmul_2_cast="float16")),
         ("EventfulTokenwiseBlock", (28, 28), False, dict(window_size=(14, 14), relative_embedding_size=(14, 14))),
         ("EventfulTokenwiseBlock", (20, 23), False, dict(window_size=(7, 7), relative_embedding_size=(16, 16))),
         ("EventfulMatmul1Block", (18, 18), False, {}), ("EventfulBlock", (18, 18), False, dict(pool_size=3))]
for kind, isz, cls, kw in cases:
    n = isz[0] * isz[1] + int(cls)
    k = max(1, n // 3)
    rel = kw.get("relative_embedding_size")
    if rel is not None and kw.get("window_size"):
        rel = kw["window_size"]
    try:
        params = O.make_block_params(dim, 4, seed=n, std=0.02, rel_sizes=rel, head_dim=64)
        ob = O.BlockOracle(kind, params, dim, heads, isz, **kw)
        ob.set_policy(lambda: O.TopK(k))
        blk = H.product_block(kind, params, dim, heads, isz, **kw)
        H.set_policies(blk, policies.TokenNormTopK, k=k)
        xs = O.make_token_stream(2, n, dim, 4, k, seed=n + 1, small=0.01)
        errs = []
        with torch.inference_mode():
            for t in range(4):
                y = blk(xs[t].cuda()).cpu()
                errs.append(float((y - ob.forward(xs[t])).abs().max()))
            # fixed point
            y = blk(xs[-1].cuda()).clone(); stable = -1
            for t in range(3 * (-(-n // k)) + 4):
                y2 = blk(xs[-1].cuda()).clone()
                if torch.equal(y, y2):
                    stable = t; break
                y = y2
            # r = N
            H.set_policies(blk, policies.TokenNormTopK, k=n)
            blk.reset(); ys = [blk(x.cuda()).clone() for x in xs]
            dn = []
            for t, x in enumerate(xs):
                blk.reset(); dn.append(float((ys[t] - blk(x.cuda())).abs().max()))
        print(f"{kind:24s} {str(isz):9s} cls={int(cls)} {str(kw)[:70]:70s}: err {['%.0e' % e for e in errs]} fixed point after {stable} r=N {max(dn):.1e}", flush=True)
    except Exception as e:
        print(f"{kind:24s} {str(isz):9s} {str(kw)[:70]:70s}: RAISED {type(e).__name__}: {str(e)[:150]}", flush=True)
