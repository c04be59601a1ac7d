#!/usr/bin/env python3
"""Probe: inputs a user could plausibly pass -- non-contiguous frames, a changed batch size / token count without reset(), autograd mode."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies
N, D, K = 197, 768, 128
sd = H.backbone_params(2, D, 4, 41, N)
from eventful_transformer.backbones import ViTBackbone
def model(cast="bfloat16"):
    bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4, matmul_2_cast=cast), depth=2, position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
    bb.load_state_dict(sd); bb = bb.eval().cuda()
    H.set_policies(bb, policies.TokenNormTopK, k=K)
    return bb
g = torch.Generator(device="cuda").manual_seed(3)
B = 4
xs = [torch.randn(B, N, D, device="cuda", generator=g) for _ in range(3)]
bb = model()
with torch.inference_mode():
    bb.reset(); ref = [bb(x).clone() for x in xs]
    # 1. non-contiguous frames (a transposed view of (N, B, D) storage)
    bb.reset()
    outs = [bb(x.transpose(0, 1).contiguous().transpose(0, 1)).clone() for x in xs]
    print("non-contiguous input equal:", all(torch.equal(a, b) for a, b in zip(ref, outs)), [float((a - b).abs().max()) for a, b in zip(ref, outs)])
    # 1b. a strided slice along D (every other channel of a 2D-wide tensor)
    bb.reset()
    wide = [torch.stack([x, x * 0 + 7.0], dim=-1).reshape(B, N, 2 * D) for x in xs]
    outs = [bb(w[..., ::2]).clone() for w in wide]
    print("strided-channel input equal:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
# 2. without inference_mode (autograd on, weights require grad)
bb.reset()
try:
    outs = [bb(x).detach().clone() for x in xs]
    print("autograd-mode run equal:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
except Exception as e:
    print("autograd-mode run raised:", type(e).__name__, str(e)[:200])
# 3. batch size changes without reset
with torch.inference_mode():
    bb.reset(); bb(xs[0])
    try:
        y = bb(xs[1][:2])
        print("batch 4 -> 2 without reset: NO error, output finite:", bool(torch.isfinite(y).all()), tuple(y.shape))
    except Exception as e:
        print("batch 4 -> 2 without reset raised:", type(e).__name__, str(e)[:200])
    bb.reset(); bb(xs[0][:2])
    try:
        y = bb(xs[1])
        print("batch 2 -> 4 without reset: NO error", tuple(y.shape))
    except Exception as e:
        print("batch 2 -> 4 without reset raised:", type(e).__name__, str(e)[:200])
    # 4. double / half input
    bb.reset()
    for dt in (torch.float64, torch.float16):
        try:
            y = bb(xs[0].to(dt))
            print(f"{dt} input: NO error, out dtype {y.dtype}, close to fp32 run: {float((y.float() - ref[0]).abs().max()):.3e}")
        except Exception as e:
            print(f"{dt} input raised:", type(e).__name__, str(e)[:160])
        bb.reset()
    # 5. token count that does not match the model (first frame)
    for n_bad in (150, 200):
        bb.reset()
        try:
            y = bb(torch.randn(2, n_bad, D, device="cuda"))
            print(f"{n_bad} tokens into a 197-token model: NO error", tuple(y.shape))
        except Exception as e:
            print(f"{n_bad} tokens raised:", type(e).__name__, str(e)[:160])
    # 6. stand-alone modules with a changed batch
    from eventful_transformer import modules as M
    gate = M.TokenGate(); gate.policy = policies.TokenNormTopK(8)
    gate(torch.randn(4, 50, 64, device="cuda"))
    try:
        gate(torch.randn(2, 50, 64, device="cuda")); print("TokenGate batch 4 -> 2: NO error")
    except Exception as e:
        print("TokenGate batch 4 -> 2 raised:", type(e).__name__, str(e)[:120])
    buf = M.TokenBuffer(); buf(torch.randn(4, 50, 64, device="cuda"), None)
    try:
        buf(torch.randn(2, 8, 64, device="cuda"), torch.zeros(2, 8, dtype=torch.long, device="cuda")); print("TokenBuffer batch 4 -> 2: NO error")
    except Exception as e:
        print("TokenBuffer batch 4 -> 2 raised:", type(e).__name__, str(e)[:120])
