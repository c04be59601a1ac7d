import os, sys
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(os.getcwd(), p))
import torch
import helpers as H
from eventful_transformer import policies
sd = H.backbone_params(2, 768, 4, 41, 197)
from eventful_transformer.backbones import ViTBackbone
def model():
    bb = ViTBackbone(block_config=dict(dim=768, heads=12, mlp_ratio=4, matmul_2_cast="bfloat16"), depth=2, position_encoding_size=(14, 14), input_size=(14, 14), block_class="EventfulBlock", has_class_token=True)
    bb.load_state_dict(sd); bb = bb.eval().cuda(); H.set_policies(bb, policies.TokenNormTopK, k=128); return bb
x = torch.randn(2, 197, 768, device="cuda")
for name, conv in (("half()", lambda m: m.half()), ("bfloat16()", lambda m: m.bfloat16()), ("double()", lambda m: m.double()), ("train()", lambda m: m.train()), ("cpu()", lambda m: m.cpu())):
    bb = conv(model())
    for xin, xn in ((x, "fp32 x"), (x.to(next(bb.parameters()).dtype).to(next(bb.parameters()).device), "matching x")):
        try:
            with torch.inference_mode():
                y = bb(xin); y2 = bb(xin + 0.1)
            print(f"{name:12s} {xn:10s}: NO error, out {y.dtype} finite {bool(torch.isfinite(y2).all())}", flush=True)
        except Exception as e:
            print(f"{name:12s} {xn:10s}: RAISED {type(e).__name__}: {str(e)[:140]}", flush=True)
        bb.reset()
