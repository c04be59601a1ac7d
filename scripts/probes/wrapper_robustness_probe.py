#!/usr/bin/env python3
"""Probe: the model wrappers with inputs a user could plausibly pass -- other resolutions / frame counts / dtypes / batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies
from models.vivit import FactorizedViViT
from models.vitdet import ViTDet
DEV = "cuda"
def attempt(name, fn):
    try:
        out = fn()
        torch.cuda.synchronize()
        desc = {k: tuple(v.shape) for k, v in out.items()} if isinstance(out, dict) else tuple(out.shape)
        fin = all(bool(torch.isfinite(v).all()) for v in (out.values() if isinstance(out, dict) else [out]))
        print(f"{name}: ok {desc} finite={fin}", flush=True)
    except Exception as e:
        print(f"{name}: RAISED {type(e).__name__}: {str(e)[:170]}", flush=True)
model = FactorizedViViT(**H.VIVIT_B_CONFIG)
model.load_state_dict(H.seeded_module_params(model, 5), strict=True)
model = model.eval().to(DEV)
H.set_policies(model, policies.TokenNormTopK, k=128)
with torch.inference_mode():
    clip = H.synthetic_video(6).to(DEV)
    print("reference clip", tuple(clip.shape), clip.dtype)
    attempt("vivit uint8 (1,80,3,224,224)", lambda: model(clip))
    attempt("vivit float32 clip", lambda: model(clip.float()))
    attempt("vivit 2 clips", lambda: model(torch.cat([clip, clip.flip(1)])))
    attempt("vivit 64 frames", lambda: model(clip[:, :64]))
    attempt("vivit 100 frames", lambda: model(torch.cat([clip, clip[:, :20]], dim=1)))
    attempt("vivit 256x256", lambda: model(torch.nn.functional.interpolate(clip[0].float(), size=(256, 256)).to(torch.uint8)[None]))
    attempt("vivit 192x224", lambda: model(clip[..., :192, :]))
    attempt("vivit again after errors", lambda: model(clip))
bcfg = dict(block_config=dict(dim=768, heads=12, mlp_ratio=4, relative_embedding_size=(64, 64), window_size=(14, 14)),
            depth=12, position_encoding_size=(14, 14), block_class="EventfulBlock", windowed_class="EventfulTokenwiseBlock", window_indices=H.VITDET_WINDOWED)
det = ViTDet(bcfg, (3, 448, 448), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], 256, (16, 16), [4.0, 2.0, 1.0, 0.5])
det.load_state_dict(H.seeded_module_params(det, 6), strict=True)
det = det.eval().to(DEV)
H.set_policies(det, policies.TokenNormTopK, k=128)
g = torch.Generator(device=DEV).manual_seed(3)
img = lambda *s: torch.randint(0, 256, s, dtype=torch.uint8, device=DEV, generator=g)
with torch.inference_mode():
    attempt("vitdet 448x448 frame 0", lambda: det(img(1, 3, 448, 448)))
    attempt("vitdet 448x448 frame 1", lambda: det(img(1, 3, 448, 448)))
    attempt("vitdet 400x300 without reset", lambda: det(img(1, 3, 400, 300)))
    det.reset()
    attempt("vitdet 400x300 after reset", lambda: det(img(1, 3, 400, 300)))
    attempt("vitdet 400x300 frame 1", lambda: det(img(1, 3, 400, 300)))
    det.reset()
    attempt("vitdet 512x512 (larger than input_shape)", lambda: det(img(1, 3, 512, 512)))
    det.reset()
    attempt("vitdet batch 2", lambda: det(img(2, 3, 448, 448)))
    attempt("vitdet batch 2 frame 1", lambda: det(img(2, 3, 448, 448)))
    det.reset()
    attempt("vitdet float input 0..255", lambda: det(img(1, 3, 448, 448).float()))
    det.reset()
    attempt("vitdet 448x448 again", lambda: det(img(1, 3, 448, 448)))
