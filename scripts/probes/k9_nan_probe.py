#!/usr/bin/env python3
"""Debug probe: evt_attention_stream first frame on a rel-pos grid, where do non-finite reference values appear?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n

dev = torch.device("cuda", 0)
import os
CASES = [(324, 18, torch.bfloat16), (272, 17, torch.float32), (260, 13, torch.float32), (1764, 42, torch.float32), (1024, 32, torch.float32), (31 * 9, 31, torch.float32), (29 * 10, 29, torch.float32), (20 * 14, 20, torch.float32), (24 * 12, 24, torch.float32), (36 * 9, 36, torch.float32), (33 * 9, 33, torch.float32)]
for (N, gw, sdt) in CASES:
    B, H, dh = 2, 2, 64
    D = H * dh
    gh = N // gw
    g = torch.Generator().manual_seed(N)
    buf = (torch.randn(B, N, 3 * D, generator=g) * 1.5).to(dev)
    ry = (torch.randn(gh, gh, dh, generator=g) * 0.2).to(dev)
    rx = (torch.randn(gw, gw, dh, generator=g) * 0.2).to(dev)
    store = n.store_code(sdt)
    terms = torch.empty(B, H, N, gh + gw, device=dev)
    n.rel_terms(buf, ry, rx, B, H, N, D, gh, gw, gw, terms, split=1)
    apT = torch.full((B, H, N, N), 7.0, dtype=sdt, device=dev)
    vp = torch.empty(B, N, D, dtype=sdt, device=dev)
    pv = torch.empty(B, N, D, dtype=sdt, device=dev)
    out = torch.empty(B, N, D, device=dev)
    n.v_gate(buf, None, None, B, N, D, 0, vp, None, None, store, False)
    n.attention_stream(buf, apT, pv, B, H, N, D, 8.0, store, True, v_state=vp, out_f32=out, qk_split=1, rel_terms=terms, gh=gh, gw=gw)
    torch.cuda.synchronize()
    a = apT.float().transpose(-1, -2)   # [b][h][row][key]
    bad = ~torch.isfinite(a)
    rows = bad.any(-1)                  # (B,H,N)
    print(f"N={N} gw={gw} {sdt}: non-finite entries {int(bad.sum())}, rows with any {int(rows.sum())} of {rows.numel()}; terms finite {bool(torch.isfinite(terms).all())}")
    if bad.any():
        r = rows.nonzero()
        print("   first bad rows (b,h,row):", r[:12].tolist())
        print("   bad rows mod 32 histogram:", torch.bincount(r[:, 2] % 32, minlength=32).tolist())
        b0, h0, r0 = r[0].tolist()
        print("   row", r0, "bad keys:", bad[b0, h0, r0].nonzero().flatten()[:40].tolist(), "count", int(bad[b0, h0, r0].sum()))
        print("   row sums of finite rows (should be 1):", a[~rows][:4].sum(-1).tolist() if (~rows).any() else None)
