#!/usr/bin/env python3
"""Probe: the ViTDet path (K8 windows, K9 streamed attention with its transposed (B,H,N,N) gate reference, small / split GEMMs) at 32 and 48
streams per launch (gate reference 4.8 / 7.2 GB per global block): permuting the streams must permute the outputs, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies
rel_for = lambda i: (14, 14) if i in H.VITDET_WINDOWED else (64, 64)
sd = H.backbone_params(12, 768, 4, 91, 14 * 14, rel_for=rel_for)
for cast, S in ((None, 32), ("float16", 48)):
    bb = H.product_vitdet(42, sd, cast)
    H.set_policies(bb, policies.TokenNormTopK, k=256)
    n = 42 * 42
    g = torch.Generator(device="cuda").manual_seed(S)
    xs = [torch.randn(S, n, 768, device="cuda", generator=g)]
    for t in range(2):
        xs.append(xs[-1] + 0.25 * torch.randn(S, n, 768, device="cuda", generator=g))
    perm = torch.randperm(S, device="cuda", generator=g)
    with torch.inference_mode():
        bb.reset(); ys = [bb(x).clone() for x in xs]
        bb.reset(); yp = [bb(x[perm]).clone() for x in xs]
    print(f"cast {cast} streams {S}: finite {all(bool(torch.isfinite(y).all()) for y in ys)}; permuted equal per frame {[bool(torch.equal(yp[t], ys[t][perm])) for t in range(3)]}; "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del bb, ys, yp, xs
    torch.cuda.empty_cache()
