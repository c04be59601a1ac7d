#!/usr/bin/env python3
"""Phase timing inside the fused gated attention kernel (library built with -DEVT_PROF): s_memtime ticks wave 0 of one
workgroup spends per phase of the ViViT-B launch (B clips, k = 128, bf16 store, in-kernel q.k^T, fused projection norm).
Usage: EVT_LIB=<prof build> python scripts/attn_prof.py [--clips 256]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch  # noqa: E402

from eventful_transformer import _native as n  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--dense", choices=["vivit", "window"], default=None,
                help="K8 (evt_attention_dense) instead: vivit = first frame of a clip (B clips, N = 197, bf16, states written); "
                     "window = ViTDet 14 x 14 windows of a 42 x 42 frame (fp32, rel-pos)")
ap.add_argument("--gated", action="store_true", help="K10 (evt_attention_gated): ViViT-B gated frame, one workgroup per (clip, head)")
ap.add_argument("--vitdet", action="store_true", help="one stream, N = 1764 (42 x 42 grid, rel-pos), k = 256, fp32 store, score state read from HBM")
a = ap.parse_args()
if a.dense is not None:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    D, H = 768, 12
    if a.dense == "vivit":
        B, N = a.clips, 197
        sdt = torch.bfloat16
        qkv = torch.randn(B, N, 3 * D, device=dev, generator=g)
        out = torch.empty(B, N, D, device=dev)
        ap_ = torch.empty(B, H, N, N, device=dev, dtype=sdt)
        pv = torch.empty(B, N, D, device=dev, dtype=sdt)
        for _ in range(30):
            n.attention_dense(qkv, B, H, N, D, 8.0, n.store_code(sdt), out_f32=out, a_state=ap_, pv=pv)
    else:
        G, N = 9, 196   # nine windows of one 42 x 42 frame, identity window map
        qkv = torch.randn(G, N, 3 * D, device=dev, generator=g)
        out = torch.empty(G, N, D, device=dev)
        ry = torch.randn(14, 14, 64, device=dev, generator=g) * 0.2
        rx = torch.randn(14, 14, 64, device=dev, generator=g) * 0.2
        for _ in range(100):
            n.attention_dense(qkv, G, H, N, D, 8.0, n.store_code(torch.float32), out_f32=out, rel_y=ry, rel_x=rx, gh=14, gw=14, qw=14)
    torch.cuda.synchronize()
    lib = n.load()
    if a.dense == "window" and not os.environ.get("EVT_DENSE_TILED"):   # the resident kernel (evt_attn_window.hip)
        buf = (ctypes.c_ulonglong * 12)()
        lib.evt_debug_prof_window.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
        assert lib.evt_debug_prof_window(buf) == 0
        names = ["window map + q requests", "K / V / table requests issued", "K / V planes written", "q blocks + barrier + rel-pos items", "barrier", "S^T products",
                 "rel-pos adds + masks + row max", "exps + row sum", "P conversion + P.V", "epilogue"]
        tot = sum(buf[q] for q in range(10))
        print(f"K8 resident, window launch: wave 0 of one workgroup: {tot} ticks (100 MHz: {tot / 100:.1f} us)")
        for q, nm in enumerate(names):
            print(f"   {nm:32s} {buf[q]:8d}  {100.0 * buf[q] / max(1, tot):5.1f} %")
        sys.exit(0)
    buf = (ctypes.c_ulonglong * 8)()
    lib.evt_debug_prof_dense.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.evt_debug_prof_dense(buf) == 0
    names = ["q rows + K chunk 0 staged", "rel-pos dots + q fragments", "q.k^T chunks", "softmax + state write", "P.V chunks", "epilogue"]
    tot = sum(buf[q] for q in range(6))
    print(f"K8 {a.dense}: wave 0 of one workgroup: {tot} ticks")
    for q, nm in enumerate(names):
        print(f"   {nm:30s} {buf[q]:8d}  {100.0 * buf[q] / max(1, tot):5.1f} %")
    sys.exit(0)
if a.gated:
    B, N, D, H, k = a.clips, 197, 768, 12, 128
    dev = torch.device("cuda", 0)
    sdt = torch.bfloat16
    store = n.store_code(sdt)
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(B, N, 3 * D, device=dev, generator=g)
    p = torch.randn(B, N, D, device=dev, generator=g)
    idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
    tiles = n.gated_tiles_empty(B, H, N, sdt, dev)
    tiles.zero_()
    vp = torch.randn(B, N, D, device=dev, generator=g).to(sdt)
    pv = torch.zeros(B, N, D, device=dev, dtype=sdt)
    nparts = torch.empty(B, N, H, device=dev)
    for _ in range(100):
        n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, norm_ref=p, norm_parts=nparts)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    lib = n.load()
    lib.evt_debug_prof_gated.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.evt_debug_prof_gated(buf) == 0
    names = ["zero fill + requests issued", "q fragments (through LDS) + barrier", "planes written (K split, value gate) + barrier", "S^T products",
             "softmax", "pack + pass 1 (a~ . dv~)", "pass 2 (A gate, da~ . v_old, tile stores)", "epilogue"]
    tot = sum(buf[q] for q in range(8))
    print(f"K10 gated frame: wave 0 of one workgroup: {tot} ticks")
    for q, nm in enumerate(names):
        print(f"   {nm:48s} {buf[q]:8d}  {100.0 * buf[q] / max(1, tot):5.1f} %")
    sys.exit(0)
B, N, D, H, k = (1, 1764, 768, 12, 256) if a.vitdet else (a.clips, 197, 768, 12, 128)
dev = torch.device("cuda", 0)
sdt = torch.float32 if a.vitdet else torch.bfloat16
store = n.store_code(sdt)
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device=dev, generator=g)
p = torch.randn(B, N, D, device=dev, generator=g)
idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
ap_ = torch.rand(B, H, N, N, device=dev, generator=g).to(sdt)
vp = torch.randn(B, N, D, device=dev, generator=g).to(sdt)
pv = torch.zeros(B, N, D, device=dev, dtype=sdt)
out = torch.empty(B, N, D, device=dev)
vd_t = torch.empty(B, D, k, device=dev, dtype=sdt)
vo_t = torch.empty(B, D, k, device=dev, dtype=sdt)
nparts = torch.empty(B, N, H, device=dev)
n.v_gate(qkv, idx, None, B, N, D, k, vp, vd_t, vo_t, store, True, transposed=True)
if a.vitdet:
    product = torch.randn(B, H, N, N, device=dev, generator=g)
    ry = torch.randn(42, 42, D // H, device=dev, generator=g) * 0.2
    rx = torch.randn(42, 42, D // H, device=dev, generator=g) * 0.2
    for _ in range(50):
        n.softmax_av_gated(product, ap_, idx, None, k, vd_t, vo_t, pv, out, B, H, N, D, store, qkv=qkv, rel_y=ry, rel_x=rx, gh=42, gw=42)
else:
  for _ in range(100):
    n.softmax_av_gated(None, ap_, idx, None, k, vd_t, vo_t, pv, out, B, H, N, D, store, qkv=qkv, scale=8.0, norm_ref=p, norm_parts=nparts)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib = n.load()
lib.evt_debug_prof_attn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.evt_debug_prof_attn(buf) == 0
names = ["prefetch issue + q.k^T", "softmax statistics", "barrier after staging", "MFMA sweeps (+ barrier)", "epilogue",
         "2a: A gate + state scatter", "2b: V staging + next V request"]
tot = sum(buf[q] for q in range(7)) + buf[9]
if a.vitdet:
    print(f"   (streamed path) before the row pass: index loads, rel-pos terms {buf[9]}")
print(f"wave 0 of one workgroup: {tot} ticks")
for q, nm in enumerate(names):
    print(f"   {nm:30s} {buf[q]:8d}  {100.0 * buf[q] / max(1, tot):5.1f} %")
