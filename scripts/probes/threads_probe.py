#!/usr/bin/env python3
"""Probe: two host threads, each driving its own model on its own HIP stream at the same time (scratch pools are per thread) -- results must be
those of the same models run one after the other."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import helpers as H
from eventful_transformer import policies
sd = H.backbone_params(12, 768, 4, 41, 197)
def make(cast):
    bb = H.product_vivit(sd, cast); H.set_policies(bb, policies.TokenNormTopK, k=128); return bb
def frames(seed, B):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xs = [torch.randn(B, 197, 768, device="cuda", generator=g)]
    for t in range(5): xs.append(xs[-1] + 0.25 * torch.randn(B, 197, 768, device="cuda", generator=g))
    return xs
jobs = [("bfloat16", 32, 1), (None, 8, 2), ("float16", 16, 3), ("bfloat16", 32, 4)]
models = [make(c) for c, _, _ in jobs]
data = [frames(s, b) for _, b, s in jobs]
torch.cuda.synchronize()
def run(i, out, stream=None):
    with torch.inference_mode():
        if stream is None:
            models[i].reset(); out[i] = [models[i](x).clone() for x in data[i]]
        else:
            with torch.cuda.stream(stream):
                models[i].reset(); out[i] = [models[i](x).clone() for x in data[i]]
            stream.synchronize()
seq = {}
for i in range(len(jobs)): run(i, seq)
torch.cuda.synchronize()
for rep in range(3):
    par = {}
    ths = [threading.Thread(target=run, args=(i, par, torch.cuda.Stream())) for i in range(len(jobs))]
    for th in ths: th.start()
    for th in ths: th.join()
    torch.cuda.synchronize()
    print("rep", rep, "threads == sequential:", [all(torch.equal(a, b) for a, b in zip(seq[i], par[i])) for i in range(len(jobs))], flush=True)
