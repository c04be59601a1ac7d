#!/usr/bin/env python3
"""Row pass + gated linear at the headline shape (B = 256 clips, N = 197, k = 128, D = 768): the gate input as fp32 against the
gate input as three bf16 planes (evt_row_pass_split + evt_linear_desc.a_lo2), per launch and as the pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch
from eventful_transformer import _native as n

dev = torch.device("cuda", 0)
B, N, D, k = 256, 197, 768, 128
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, N, D, device=dev, generator=g)
res = torch.randn(B, N, D, device=dev, generator=g)
p = torch.randn(B, N, D, device=dev, generator=g)
lw, lb = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
s_out, c32, norms = torch.empty_like(x), torch.empty_like(x), torch.empty(B * N, device=dev)
planes = torch.empty(B, N, 2 * D, dtype=torch.bfloat16, device=dev)
lo2 = torch.empty(B, N, D, dtype=torch.bfloat16, device=dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


rp32 = lambda: n.row_pass(x, B * N, D, res=res, sum_out=s_out, ln_w=lw, ln_b=lb, c_out=c32, p=p, norms=norms)
rp3 = lambda: n.row_pass_split(x, B * N, D, planes, lo2, res=res, sum_out=s_out, ln_w=lw, ln_b=lb, p=p, norms=norms)
print(f"row pass fp32 gate input {timeit(rp32):7.1f} us   three planes {timeit(rp3):7.1f} us")
for name, Nout in (("QKV", 3 * D), ("projection-like (768)", D)):
    W = torch.randn(Nout, D, device=dev, generator=g) * 0.02
    bias = torch.zeros(Nout, device=dev)
    Ws = n.split_weight(W)
    out = torch.zeros(B, N, Nout, device=dev)
    f32 = lambda: n.gated_linear(c32, D, idx, N, W, bias, out, Nout, idx, N, None, p, B, k, D, Nout, 0, W_split=Ws)
    pl = lambda: n.gated_linear(planes, D, idx, N, W, bias, out, Nout, idx, N, None, p, B, k, D, Nout, 0, W_split=Ws, a_lo2=lo2)
    flop = 2.0 * B * k * D * Nout
    a, b = timeit(f32), timeit(pl)
    print(f"{name:24s} fp32 A {a:7.1f} us ({flop / a * 1e-6:6.1f} TF)   planes {b:7.1f} us ({flop / b * 1e-6:6.1f} TF)   tile {n.gated_linear_big_tile(D, True, N, Nout, True, N, False, B, k, D, Nout, planes=True)}")
Dh = 4 * D
W1 = torch.randn(Dh, D, device=dev, generator=g) * 0.02
W2 = torch.randn(D, Dh, device=dev, generator=g) * 0.02
b1, b2 = torch.zeros(Dh, device=dev), torch.zeros(D, device=dev)
S1, S2 = n.split_weight(W1), n.split_weight(W2)
hidden = torch.empty(B * k, Dh, device=dev)
out = torch.zeros(B, N, D, device=dev)
m32 = lambda: n.gated_mlp(c32, D, idx, N, W1, b1, W2, b2, hidden, out, D, None, p, B, k, D, Dh, W1_split=S1, W2_split=S2)
mpl = lambda: n.gated_mlp(planes, D, idx, N, W1, b1, W2, b2, hidden, out, D, None, p, B, k, D, Dh, W1_split=S1, W2_split=S2, a_lo2=lo2)
flop = 4.0 * B * k * D * Dh
a, b = timeit(m32), timeit(mpl)
print(f"{'MLP (both launches)':24s} fp32 A {a:7.1f} us ({flop / a * 1e-6:6.1f} TF)   planes {b:7.1f} us ({flop / b * 1e-6:6.1f} TF)")
