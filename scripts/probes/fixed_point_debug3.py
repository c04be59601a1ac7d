#!/usr/bin/env python3
"""Debug: constant input, fp32 mode -- where does block 0's A.v state move?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies, blocks as EB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, D, K = 197, 768, 128
sd = H.backbone_params(1, D, 4, 41, N)
from eventful_transformer.backbones import ViTBackbone
bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4), depth=1, position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
bb.load_state_dict(sd); bb = bb.eval().cuda()
H.set_policies(bb, policies.TokenNormTopK, k=K)
g = torch.Generator(device="cuda").manual_seed(3)
tok = lambda: torch.randn(B, N, D, device="cuda", generator=g)
frames = [tok(), tok(), tok()]
const = frames[-1]
b0 = bb.blocks[0]
seen = []
EB.INDEX_TAP = lambda blk, tag, idx, count: seen.append((tag, idx.clone()))
with torch.inference_mode():
    bb.reset()
    for x in frames:
        bb(x)
    prev = None
    for t in range(6):
        seen.clear()
        bb(const)
        av = b0.matmul_accumulator_2._state.clone()
        if prev is not None:
            d = (av - prev)
            nz = d.nonzero()
            print(t, "moved elements", nz.shape[0], "of", av.numel(), "max", float(d.abs().max()))
            if nz.shape[0]:
                rows = nz[:, 1].unique()
                qidx = [i for tg, i in seen if tg == "qkv"][0]
                print("   rows moved:", rows[:20].tolist(), " qkv idx[0][:8]:", qidx[0][:8].tolist(), " heads:", (nz[:, 2] // 64).unique().tolist()[:12])
                print("   sample:", [(int(a), int(b), int(c), float(prev[a, b, c]), float(av[a, b, c])) for a, b, c in nz[:4].tolist()])
        prev = av
