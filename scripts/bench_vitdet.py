#!/usr/bin/env python3
"""Secondary benchmark: ViTDet-B backbone frames/s (BASELINE configs 3 and 5), B = 1 video stream.
  python scripts/bench_vitdet.py --grid 42 --policy topk --k 256            (config 3, fp32)
  python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16   (config 5)
Timing protocol of scripts/time/vitdet_vid.py:28-55: device-synchronised wall clock around the backbone
per frame; mean over all frames and over non-first frames."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=42)
ap.add_argument("--policy", default="topk")
ap.add_argument("--k", type=int, default=256)
ap.add_argument("--thr", type=float, default=1.0)
ap.add_argument("--cast", default="none")
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--graphs", action="store_true", help="replay HIP graphs of the first / incremental frame (graphs.py)")
ap.add_argument("--fuse-dense-norm-rows", type=int, default=None, help="A/B: blocks.FUSE_DENSE_NORM_ROWS (0 = the windowed blocks' attention always emits the projection gate's norm)")
a = ap.parse_args()
if a.fuse_dense_norm_rows is not None:
    from eventful_transformer import blocks as _EB
    _EB.FUSE_DENSE_NORM_ROWS = a.fuse_dense_norm_rows
cast = None if a.cast == "none" else a.cast
from eventful_transformer import policies
rel_for = lambda i: (14, 14) if i in H.VITDET_WINDOWED else (64, 64)
sd = H.backbone_params(12, 768, 4, 91, 14 * 14, rel_for=rel_for)
bb = H.product_vitdet(a.grid, sd, cast)
if a.policy == "topk":
    H.set_policies(bb, policies.TokenNormTopK, k=a.k)
    xs = torch.cat([O.make_token_stream(1, a.grid ** 2, 768, a.frames, a.k, seed=5 + b, small=0.01) for b in range(a.batch)], dim=1)
else:
    H.set_policies(bb, policies.TokenNormThreshold, threshold=a.thr)
    xs = O.make_threshold_stream(a.grid ** 2, 768, a.frames, 7)
xs = xs.cuda()
times = []
run = bb
if a.graphs:
    from eventful_transformer.graphs import FrameGraphs
    run = FrameGraphs(bb)
with torch.inference_mode():
    for rep in range(3 if a.graphs else 2):
        run.reset()
        times = []
        for t in range(a.frames):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(xs[t])
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
nf = times[1:]
print(json.dumps({"config": f"ViTDet-B backbone {a.grid*16}^2 N={a.grid**2} policy={a.policy} k={a.k} thr={a.thr} cast={cast} B={a.batch} graphs={a.graphs}",
                  "first_frame_ms": round(times[0] * 1e3, 2), "non_first_ms": round(sum(nf) / len(nf) * 1e3, 2),
                  "frames_per_s_non_first": round(a.batch * len(nf) / sum(nf), 1),
                  "frames_per_s_all": round(a.batch * len(times) / sum(times), 1)}))
