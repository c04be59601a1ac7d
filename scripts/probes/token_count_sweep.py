#!/usr/bin/env python3
"""Probe: one EventfulBlock of ViT-B width (768, 12 heads) at awkward token counts against the CPU oracle, free-running on streams
with a designed top-k margin (O.make_token_stream): first frame + 3 gated frames, fp32 and bf16 cast."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies
torch.set_num_threads(8)
bad = 0
for N in [int(a) for a in sys.argv[1:]] or (1, 2, 3, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 258, 288, 300):
    for cast in (None, "bfloat16"):
        k = max(1, N // 2)
        params = O.make_block_params(768, 4, seed=N, std=0.02, head_dim=64)
        kw = dict(matmul_2_cast=cast) if cast else {}
        ob = O.BlockOracle("EventfulBlock", params, 768, 12, (1, N), **kw)
        ob.set_policy(lambda: O.TopK(k))
        try:
            blk = H.product_block("EventfulBlock", params, 768, 12, (1, N), **kw)
            H.set_policies(blk, policies.TokenNormTopK, k=k)
            xs = O.make_token_stream(2, N, 768, 4, k, seed=N + 1, small=0.01)
            errs, notes = [], []
            from eventful_transformer import blocks as EB
            seen = {}
            EB.INDEX_TAP = lambda b_, tag, idx, count: seen.__setitem__(tag, idx.clone())
            with torch.inference_mode():
                for t in range(4):
                    seen.clear()
                    y_ref = ob.forward(xs[t])
                    y = blk(xs[t].cuda()).cpu()
                    errs.append(float((y - y_ref).abs().max()))
                    if t:
                        for tag, gn in (("qkv", "qkv_gate"), ("projection", "projection_gate"), ("mlp", "mlp_gate")):
                            want = ob.trace[tag + "_index"].sort(dim=-1)[0]
                            got = seen[tag].long().cpu()
                            if not torch.equal(want, got):
                                e = ob.policy[gn].last_input
                                nrm = torch.linalg.vector_norm(e.double(), dim=-1).sort(dim=-1, descending=True)[0]
                                margin = ((nrm[:, k - 1] - nrm[:, k]) / nrm[:, k - 1]).min() if k < N else float("nan")
                                notes.append(f"frame {t} {tag}: sets differ, oracle margin {float(margin):.1e}")
            EB.INDEX_TAP = None
            tol = 2e-4 if cast is None else 2e-3
            flag = "" if max(errs) <= tol else "   <-- ABOVE TOLERANCE"
            bad += bool(flag)
            print(f"N={N:4d} k={k:4d} {str(cast):9s} max err per frame {['%.1e' % e for e in errs]}{flag} {notes[:3]}", flush=True)
        except Exception as e:
            bad += 1
            print(f"N={N:4d} k={k:4d} {str(cast):9s} RAISED {type(e).__name__}: {str(e)[:200]}", flush=True)
print("cases with problems:", bad)
