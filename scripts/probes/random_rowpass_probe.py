#!/usr/bin/env python3
"""Probe: tests/test_gpu_kernels.py::test_row_pass_layernorm_residual_norm driven with random (rows, D) incl. D = 4 .. 8192 and one-row launches,
plus the norm orders 1 / inf."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import test_gpu_kernels as T
from eventful_transformer import _native as n
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
cases = 150
for case in range(cases):
    rows = rng.choice([1, 2, 3, 63, 64, 65, 197, 1000]) if rng.random() < 0.4 else rng.randint(1, 3000)
    D = 4 * rng.randint(1, 1024) if rng.random() < 0.6 else rng.choice([4, 8, 64, 768, 1024, 1280, 3072, 4096])   # (evt_row_pass: D <= 4096, a row in registers)
    try:
        T.test_row_pass_layernorm_residual_norm(rows, D)
        x = torch.randn(rows, D, device="cuda"); p = torch.randn(rows, D, device="cuda")
        for order, ref in ((1, (x - p).abs().sum(-1)), (float("inf"), (x - p).abs().amax(-1))):
            norms = torch.empty(rows, device="cuda")
            n.row_pass(x, rows, D, p=p, norms=norms, order=order)
            assert torch.allclose(norms, ref, rtol=2e-5), (order, float((norms - ref).abs().max()))
    except AssertionError as e:
        bad += 1; print(f"MISS rows {rows} D {D}: {str(e)[:160]}", flush=True)
    except Exception as e:
        bad += 1; print(f"RAISED rows {rows} D {D}: {type(e).__name__} {str(e)[:160]}", flush=True)
print(f"{cases} random row passes, {bad} to look at", flush=True)
