#!/usr/bin/env python3
"""Probe: the kernel-level attention tests of tests/test_gpu_kernels.py driven with RANDOM shapes -- evt_attention_stream (N = gh x gw in (256, 2100], any k,
store type, rel-pos, exact / split scores) and evt_attention_gated (N <= 256, any k, bf16 / fp16) against the oracle's gates / softmax / accumulator."""
import os, sys, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import test_gpu_kernels as T
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(cases):
    if rng.random() < 0.6:
        while True:
            gh, gw = rng.randint(1, 70), rng.randint(1, 70)
            N = gh * gw
            if 256 < N <= 2100:
                break
        k = rng.randint(1, N)
        cast = rng.choice([None, "bfloat16", "float16"])
        rel = rng.random() < 0.6 and gh > 1 and gw > 1
        qk = rng.choice([0, 1])
        name, fn, args = "stream", T.test_attention_stream_matches_oracle, (cast, N, gw, k, rel, qk)
    else:
        N = rng.randint(2, 256)   # (one token: the reference asserts when recombining the heads, blocks.py:341)
        k = rng.randint(1, N)
        cast = rng.choice(["bfloat16", "float16"])
        name, fn, args = "gated", T.test_attention_gated_resident_matches_oracle, (cast, N, k)
    try:
        fn(*args)
    except AssertionError as e:
        bad += 1
        print(f"MISS #{case} {name} {args}: {str(e)[:200]}", flush=True)
    except Exception as e:
        bad += 1
        print(f"RAISED #{case} {name} {args}: {type(e).__name__} {str(e)[:200]}", flush=True)
print(f"{cases} random attention kernel cases, {bad} to look at", flush=True)
