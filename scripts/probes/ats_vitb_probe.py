#!/usr/bin/env python3
"""Probe: ats_fraction (blocks.py:150-181, 378-391) at ViT-B width (12 heads => batch 12, the reference's batch == heads quirk) vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
torch.set_num_threads(8)
dim, heads, isz, frac, k = 768, 12, (14, 14), 0.7, 64
n = 197
for kind, kw in (("Block", {}), ("EventfulTokenwiseBlock", {}), ("EventfulMatmul1Block", {}), ("EventfulBlock", {}), ("EventfulBlock", dict(matmul_2_cast="bfloat16")),
                 ("Block", dict(matmul_2_cast="float16"))):
    try:
        params = O.make_block_params(dim, 4, seed=77, std=0.02, head_dim=64)
        ob = O.BlockOracle(kind, params, dim, heads, isz, ats_fraction=frac, **kw)
        blk = H.product_block(kind, params, dim, heads, isz, ats_fraction=frac, **kw)
        if kind != "Block":
            ob.set_policy(lambda: O.TopK(k)); H.set_policies(blk, policies.TokenNormTopK, k=k)
        xs = O.make_token_stream(heads, n, dim, 3, k, seed=78, small=0.02)
        out = []
        with torch.inference_mode():
            for t in range(3):
                y_ref = ob.forward(xs[t].clone())
                y = blk(xs[t].cuda()).cpu()
                same_idx = torch.equal(blk.last_ats_indices.cpu(), ob.trace["ats_index"])
                out.append(f"{float((y - y_ref).abs().max()):.1e}{'=' if same_idx else 'X'} shape {tuple(y.shape)}")
        print(f"{kind:24s} {str(kw):36s}: {out}", flush=True)
    except Exception as e:
        print(f"{kind:24s} {str(kw):36s}: RAISED {type(e).__name__}: {str(e)[:160]}", flush=True)
