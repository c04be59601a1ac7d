#!/usr/bin/env python3
"""Probe: the model wrappers converted with .half() / left on the CPU; stand-alone modules with index tensors on the CPU / of int32 / out of range."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies, modules as M
from models.vivit import FactorizedViViT
from models.vitdet import ViTDet
DEV = "cuda"
def attempt(name, fn):
    try:
        out = fn(); torch.cuda.synchronize()
        print(f"{name}: NO error", flush=True)
    except Exception as e:
        print(f"{name}: RAISED {type(e).__name__}: {str(e)[:150]}", flush=True)
clip = H.synthetic_video(6)
def vivit():
    m = FactorizedViViT(**H.VIVIT_B_CONFIG); m.load_state_dict(H.seeded_module_params(m, 5), strict=True); m = m.eval().to(DEV)
    H.set_policies(m, policies.TokenNormTopK, k=128); return m
with torch.inference_mode():
    attempt("vivit .half()", lambda: vivit().half()(clip.to(DEV)))
    attempt("vivit on cpu, clip on device", lambda: vivit().cpu()(clip.to(DEV)))
    attempt("vivit on device, clip on cpu", lambda: vivit()(clip))
bcfg = dict(block_config=dict(dim=768, heads=12, mlp_ratio=4, relative_embedding_size=(64, 64), window_size=(14, 14)),
            depth=12, position_encoding_size=(14, 14), block_class="EventfulBlock", windowed_class="EventfulTokenwiseBlock", window_indices=H.VITDET_WINDOWED)
def det():
    d = ViTDet(bcfg, (3, 448, 448), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], 256, (16, 16), [4.0, 2.0, 1.0, 0.5])
    d.load_state_dict(H.seeded_module_params(d, 6), strict=True); d = d.eval().to(DEV); H.set_policies(d, policies.TokenNormTopK, k=128); return d
img = torch.randint(0, 256, (1, 3, 448, 448), dtype=torch.uint8)
with torch.inference_mode():
    attempt("vitdet .half()", lambda: det().half()(img.to(DEV)))
    attempt("vitdet on cpu, frame on device", lambda: det().cpu()(img.to(DEV)))
    attempt("vitdet on device, frame on cpu", lambda: det()(img))
    # stand-alone modules
    buf = M.TokenBuffer(); buf(torch.randn(2, 50, 64, device=DEV), None)
    attempt("TokenBuffer index on cpu", lambda: buf(torch.randn(2, 8, 64, device=DEV), torch.arange(8).expand(2, 8)))
    attempt("TokenBuffer int32 index", lambda: buf(torch.randn(2, 8, 64, device=DEV), torch.arange(8, dtype=torch.int32, device=DEV).expand(2, 8)))
    g = M.TokenGate(); g.policy = policies.TokenNormTopK(8); g(torch.randn(2, 50, 64, device=DEV))
    attempt("TokenGate forced_index on cpu", lambda: g(torch.randn(2, 50, 64, device=DEV), forced_index=torch.arange(8).expand(2, 8)))
