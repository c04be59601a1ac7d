#!/usr/bin/env python3
"""Can the power-limited, MFMA-bound gated GEMMs and the memory-bound kernels (K10 attention, row passes) of two resident batches share
the chip SPATIALLY instead of taking turns?  Two HIP streams with complementary CU masks (hipExtStreamCreateWithCUMask): the GEMM stream
gets `--gemm-cus` compute units (its persistent launches sized for them: evt_set_cu_budget), the other stream the rest.
Reports each stream's time per iteration alone on the whole chip, alone on its partition, and with both running; "combined" =
t_alone_full / t_concurrent summed over the two streams (1.0 = what time-slicing the whole chip gives).
Usage: python scripts/probes/cu_partition_probe.py [--clips 256] [--gemm-cus 176,192,208]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))
import torch  # noqa: E402

from eventful_transformer import _native as n  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--gemm-cus", default="160,176,192,208,224")
ap.add_argument("--iters", type=int, default=12)
a = ap.parse_args()
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi):
    """A torch stream whose kernels run on CUs [lo, hi) of the mask's bit order (interleaved over the XCDs by the driver)."""
    words = (CUS + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


B, N, D, H, k = a.clips, 197, 768, 12, 128
g = torch.Generator(device=dev).manual_seed(0)
sdt = torch.bfloat16
store = n.store_code(sdt)


def make_set():
    x = torch.randn(B, N, D, device=dev, generator=g)
    p = torch.randn(B, N, D, device=dev, generator=g)
    c = torch.empty_like(x)
    w = torch.randn(D, device=dev, generator=g)
    norms = torch.empty(B * N, device=dev)
    idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:k].sort()[0] for _ in range(B)]).int().contiguous()
    Wq = torch.randn(3 * D, D, device=dev, generator=g) * 0.02
    W1 = torch.randn(4 * D, D, device=dev, generator=g) * 0.02
    W2 = torch.randn(D, 4 * D, device=dev, generator=g) * 0.02
    b3, b4, b1 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev)
    qkv = torch.randn(B, N, 3 * D, device=dev, generator=g)
    hidden = torch.empty(B * k, 4 * D, device=dev)
    buf = torch.empty(B, N, D, device=dev)
    tiles = n.gated_tiles_empty(B, H, N, sdt, dev)
    tiles.zero_()
    vp = torch.randn(B, N, D, device=dev, generator=g).to(sdt)
    pv = torch.zeros(B, N, D, device=dev, dtype=sdt)
    parts = torch.empty(B, N, H, device=dev)
    Sq, S1, S2 = (n.split_weight(t) for t in (Wq, W1, W2))

    def gemms():   # the GEMM work of one block and frame: QKV + MLP pair (the projection is 7 % of it)
        n.gated_linear(x, D, idx, N, Wq, b3, qkv, 3 * D, idx, N, None, p, B, k, D, 3 * D, W_split=Sq)
        n.gated_mlp(x, D, idx, N, W1, b4, W2, b1, hidden, buf, D, None, p, B, k, D, 4 * D, W1_split=S1, W2_split=S2)

    def mem():     # the memory-bound work of one block and frame: K10 + two row passes
        n.attention_gated(qkv, tiles, vp, pv, B, H, N, D, 8.0, store, False, idx=idx, kcap=k, norm_ref=p, norm_parts=parts)
        n.row_pass(x, B * N, D, res=p, sum_out=buf, ln_w=w, ln_b=w, c_out=c, p=p, norms=norms)
        n.row_pass(x, B * N, D, res=p, sum_out=buf, ln_w=w, ln_b=w, c_out=c, p=p, norms=norms)
    return gemms, mem


gemms_a, mem_a = make_set()
gemms_b, mem_b = make_set()


def run(stream, fn, iters, budget):
    with torch.cuda.stream(stream), n.lane(id(stream) % 97):
        n.set_cu_budget(budget)
        for _ in range(iters):
            fn()
        n.set_cu_budget(0)


def timed(pairs, iters):
    """pairs: [(stream, fn, budget)] launched interleaved from this thread; -> seconds per iteration of each, measured by events."""
    for s, fn, bud in pairs:
        run(s, fn, 2, bud)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in pairs]
    for (s, _, _), (e0, _) in zip(pairs, ev):
        e0.record(s)
    for it in range(iters):
        for s, fn, bud in pairs:
            run(s, fn, 1, bud)
    for (s, _, _), (_, e1) in zip(pairs, ev):
        e1.record(s)
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) * 1e3 / iters for e0, e1 in ev]


full = torch.cuda.Stream(device=dev)
t_g = timed([(full, gemms_a, 0)], a.iters)[0]
t_m = timed([(full, mem_a, 0)], a.iters)[0]
# what the bench does today: two batches on two unmasked streams, each running gemms then mem
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
both = lambda ga, ma: (lambda: (ga(), ma()))   # noqa: E731
t_ts = timed([(s1, both(gemms_a, mem_a), 0), (s2, both(gemms_b, mem_b), 0)], a.iters)
t_op = timed([(s1, both(gemms_a, mem_a), 0), (s2, (lambda: (mem_b(), gemms_b())), 0)], a.iters)   # the second stream starts in the other phase
print(f"# two unmasked streams, the second one starting with its memory-bound kernels (opposite phase): {max(t_op):.0f} us per pair = {max(t_op) / 2:.0f} us each")
print(f"# B={B}: GEMMs of a block-frame alone on the chip {t_g:.0f} us, memory-bound kernels alone {t_m:.0f} us, serial sum {t_g + t_m:.0f} us; "
      f"two batches on two unmasked streams: {max(t_ts):.0f} us per pair of block-frames = {max(t_ts) / 2:.0f} us each")
for gc in [int(v) for v in a.gemm_cus.split(",")]:
    sg, sm = masked_stream(0, gc), masked_stream(gc, CUS)
    tg_part = timed([(sg, gemms_a, gc)], a.iters)[0]
    tm_part = timed([(sm, mem_a, 0)], a.iters)[0]
    tg_c, tm_c = timed([(sg, gemms_a, gc), (sm, mem_b, 0)], a.iters)
    print(f"gemm CUs {gc:3d} | mem CUs {CUS - gc:3d}: alone on its partition: gemm {tg_part:.0f} us ({t_g / tg_part:.2f} of full-chip rate), mem {tm_part:.0f} us ({t_m / tm_part:.2f}); "
          f"concurrent: gemm {tg_c:.0f} us ({t_g / tg_c:.2f}), mem {tm_c:.0f} us ({t_m / tm_c:.2f}); combined {t_g / tg_c + t_m / tm_c:.2f}; "
          f"a pair of block-frames would take max(2 gemm, 2 mem) = {2 * max(tg_c, tm_c):.0f} us = {max(tg_c, tm_c):.0f} us each")
