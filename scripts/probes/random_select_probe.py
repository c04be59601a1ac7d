#!/usr/bin/env python3
"""Probe: evt_select_topk / evt_select_threshold (and the _sq forms) on random sizes and value patterns -- ties, zeros, inf, tiny / huge magnitudes --
against a stable descending sort (ties to the lowest index), incl. the complement list."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from eventful_transformer import _native as n
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for case in range(cases):
    N = rng.choice([1, 2, 3, 63, 64, 65, 197, 255, 256, 257, 1764, 2048, 2049, 4096, 8191, 16384]) if rng.random() < 0.5 else rng.randint(1, 16384)
    B = rng.choice([1, 1, 2, 5, 33])
    k = rng.randint(1, N)
    g = torch.Generator().manual_seed(case)
    pat = rng.choice(["normal", "ties", "zeros", "mixed", "inf", "tiny"])
    x = torch.rand(B, N, generator=g)
    if pat == "ties":
        x = (x * rng.choice([2, 5, 50])).floor()
    elif pat == "zeros":
        x = x * (torch.rand(B, N, generator=g) < 0.3)
    elif pat == "mixed":
        x = x * torch.pow(10.0, torch.randint(-30, 30, (B, N), generator=g).float())
    elif pat == "inf":
        x[torch.rand(B, N, generator=g) < 0.05] = float("inf")
    elif pat == "tiny":
        x = x * 1e-40
    parts = rng.choice([0, 0, 12, 3])
    xd = x.cuda()
    if parts:   # partial sums of squares whose sqrt-of-sum is the norm: split x^2 over `parts` addends (exactly representable split: first part carries all)
        sq = torch.zeros(B, N, parts)
        sq[..., 0] = x * x
        norms_in, ref_vals = sq.cuda(), torch.sqrt((x * x))
    else:
        norms_in, ref_vals = xd, x
    idx = torch.full((B, k), -7, dtype=torch.int32, device="cuda")
    rest = torch.full((B, N), -7, dtype=torch.int32, device="cuda")
    try:
        n.select_topk(norms_in, B, N, k, idx, rest, parts=parts)
        order = torch.sort(ref_vals, dim=-1, descending=True, stable=True)[1]
        want = order[:, :k].sort(dim=-1)[0]
        got = idx.long().cpu()
        ok = torch.equal(got, want)
        if N > k:
            want_rest = order[:, k:].sort(dim=-1)[0]
            ok = ok and torch.equal(rest[:, :N - k].long().cpu(), want_rest)
        # threshold at a random value
        thr = float(ref_vals.flatten()[rng.randrange(B * N)]) if pat != "inf" else 0.5
        idx2 = torch.full((B, N), -7, dtype=torch.int32, device="cuda")
        cnt = torch.full((B,), -7, dtype=torch.int32, device="cuda")
        n.select_threshold(norms_in, B, N, thr, N, idx2, cnt, None, parts=parts)
        for b in range(B):
            w = (ref_vals[b] > thr).nonzero().flatten()
            c = int(cnt[b])
            ok = ok and c == len(w) and torch.equal(idx2[b, :c].long().cpu(), w)
        if not ok:
            bad += 1
            print(f"MISS #{case} N {N} B {B} k {k} pattern {pat} parts {parts}", flush=True)
    except Exception as e:
        bad += 1
        print(f"RAISED #{case} N {N} B {B} k {k} pattern {pat} parts {parts}: {type(e).__name__} {str(e)[:150]}", flush=True)
print(f"{cases} random selections, {bad} to look at", flush=True)
