#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n
dev = torch.device("cuda", 0)
N, gw, sdt = 280, 20, torch.float32
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
out = torch.zeros(B, N, D, device=dev)
n.v_gate(buf, None, None, B, N, D, 0, vp, None, None, store, False)
n.attention_stream(buf, apT, pv, B, H, N, D, 8.0, store, True, v_state=vp, out_f32=out, qk_split=1, rel_terms=terms, gh=gh, gw=gw)
torch.cuda.synchronize()
d = out.flatten()[:256 * 2 * 2].view(256, 2, 2).cpu()   # [tid][hr][(m, s)]
for wave in range(4):
    for kg in range(4):
        t = wave * 64 + kg * 16
        print(f"wave {wave} kg {kg} lane l15=0: hr0 (m, s) = ({d[t,0,0].item():.6g}, {d[t,0,1].item():.6g})   hr1 = ({d[t,1,0].item():.6g}, {d[t,1,1].item():.6g})")
print("any nan:", bool(torch.isnan(d).any()), " any inf:", bool(torch.isinf(d).any()))
