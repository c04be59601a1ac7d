#!/usr/bin/env python3
"""Probe: tensors beyond 2 GB / 4 GB in one launch (32-bit offset overflow?) -- 2 ViViT-B EventfulBlocks at B = 1024 / 2048 clips: the hidden
MLP tensor of the first frame is B x 197 x 3072 x 4 bytes = 2.5 / 5 GB, the packed token buffer 1.9 / 3.7 GB.  A clip's outputs must not
depend on the batch it runs in: the first and the last 256 clips against runs of just those clips, bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies
from eventful_transformer.backbones import ViTBackbone
N, D, K = 197, 768, 128
sd = H.backbone_params(2, D, 4, 41, N)
def model(cast):
    cfg = dict(dim=768, heads=12, mlp_ratio=4)
    if cast: cfg["matmul_2_cast"] = cast
    bb = ViTBackbone(block_config=cfg, depth=2, position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
    bb.load_state_dict(sd); bb = bb.eval().cuda()
    H.set_policies(bb, policies.TokenNormTopK, k=K)
    return bb
for cast in ("bfloat16", None):
    for B in (1024, 2048):
        g = torch.Generator(device="cuda").manual_seed(B)
        xs = [torch.randn(B, N, D, device="cuda", generator=g)]
        for t in range(2):
            xs.append(xs[-1] + 0.25 * torch.randn(B, N, D, device="cuda", generator=g))
        bb = model(cast)
        with torch.inference_mode():
            bb.reset(); big = [bb(x).clone() for x in xs]
            res = []
            for lo in (0, B - 256):
                bb.reset()
                small = [bb(x[lo:lo + 256]).clone() for x in xs]
                res.append([bool(torch.equal(small[t], big[t][lo:lo + 256])) for t in range(3)])
                if not all(res[-1]):
                    print("   max diff", [float((small[t] - big[t][lo:lo + 256]).abs().max()) for t in range(3)])
        print(f"cast {cast} B {B}: finite {all(bool(torch.isfinite(b).all()) for b in big)}; first 256 clips equal per frame {res[0]}, last 256 {res[1]}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
        del bb, big, xs
        torch.cuda.empty_cache()
