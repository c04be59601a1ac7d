#!/usr/bin/env python3
"""Debug: is evt_softmax_av_gated (fp32 store, in-kernel scores) a fixed point when called again with identical inputs?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies, blocks as EB, _native as n
B = 4
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
orig = n.softmax_av_gated
calls = []
def wrapped(product, a_state, idx, count, kcap, v_delta_t, v_old_t, pv, out_f32, *args, **kw):
    orig(product, a_state, idx, count, kcap, v_delta_t, v_old_t, pv, out_f32, *args, **kw)
    if calls and calls[0]:
        torch.cuda.synchronize()
        print("  v_delta max", float(v_delta_t.float().abs().max()), " kcap", kcap, "product", product is not None)
        for rep in range(3):
            pv0, a0 = pv.clone(), a_state.clone()
            orig(product, a_state, idx, count, kcap, v_delta_t, v_old_t, pv, out_f32, *args, **kw)
            torch.cuda.synchronize()
            d = (pv - pv0)
            print(f"  re-call {rep}: pv moved {int((d != 0).sum())} elements, max {float(d.abs().max()):.3e}; a_state moved {int((a_state != a0).sum())}")
n.softmax_av_gated = wrapped
EB._native.softmax_av_gated = wrapped
with torch.inference_mode():
    bb.reset()
    for x in frames:
        bb(x)
    for t in range(4):
        calls[:] = [t == 3]
        bb(const)
