#!/usr/bin/env python3
"""Timing of evt_attention_dense on a ViTDet window partition (grid 42 or 64, 14x14 windows, rel-pos; `norel` drops the
tables).  The phase numbers in DESIGN.md came from temporary early-return hooks in the kernel, not kept in the product."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
from eventful_transformer import _native as n
from eventful_transformer.blocks import _window_map
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 42
D, H = 768, 12
N = grid * grid
dev = torch.device("cuda")
qkv = torch.randn(1, N, 3 * D, device=dev)
tm = _window_map((grid, grid), (14, 14), dev)
gpc, nw = tm.shape
pad = torch.randn(3 * D, device=dev)
ry = torch.randn(14, 14, 64, device=dev); rx = torch.randn(14, 14, 64, device=dev)
out = torch.zeros(1, N, D, device=dev)
norel = len(sys.argv) > 2 and sys.argv[2] == "norel"
kw = {} if norel else dict(rel_y=ry, rel_x=rx, gh=14, gw=14, qw=14)
def run():
    n.attention_dense(qkv, gpc, H, nw, D, 8.0, n.store_code(torch.float32), out_f32=out,
                      tok_map=tm, groups_per_clip=gpc, clip_rows=N, pad_row=pad, **kw)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"grid {grid} groups {gpc} norel={norel}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
