#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the ViViT-B spatial model on the gated-token (Eventful) path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--clips B] [--frames T] [--k R] [--cast bfloat16|none]

Workload (BASELINE.json configs[1]): ViViT-B factorised-encoder SPATIAL model, Kinetics shape
16 x 224^2 -> T = 16 backbone frames of 196 patch tokens + class token (N = 197), D = 768, 12
`EventfulBlock`s, top-k r = 128, reference `matmul_2_cast="bfloat16"` semantics for the A.v stage
(fp32 everywhere else).  Random-init weights (seeded normal std 0.02), synthetic token clips already
resident in HBM.  One STEP = one batch of B clips taken through all T frames the way
`FactorizedViViT._forward_view` does (vivit.py:146-147): reset(), frame 0 dense, frames 1..T-1 gated;
per frame: prepend class token, backbone, final LayerNorm, take token 0 (vivit.py:293-303).

Multi-GPU (--gpus N under torch.distributed.run): clips are independent, so every rank runs its own
B clips ("weak" scaling, no data-path collective); RCCL only broadcasts the weights from rank 0 and
reduces the timing.  value = whole-job frames/s = N * B * T * K / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the MFMA gated-linear GEMM, timed
with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle = torch-CPU port of the
reference, timed on the host cores in this same run).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "eventful-transformer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "frames/sec/GPU ViViT-B 16x224^2 r=128; gate-index bit-exact vs ref"
DIM, DEPTH, HEADS, TOKENS = 768, 12, 12, 196
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (not the 2:1-sparse figure)


def seeded_state_dict(seed=77, std=0.02):
    """Same version-stable generator as the parity tests (numpy RandomState)."""
    rs = np.random.RandomState(seed)

    def n(*shape, s=std):
        return torch.from_numpy((rs.standard_normal(shape) * s).astype(np.float32))

    sd = {"position_encoding.encoding": n(1, TOKENS + 1, DIM)}
    for i in range(DEPTH):
        p = f"blocks.{i}."
        sd[p + "input_layer_norm.weight"] = 1.0 + n(DIM, s=0.05)
        sd[p + "input_layer_norm.bias"] = n(DIM, s=0.05)
        sd[p + "qkv.weight"], sd[p + "qkv.bias"] = n(3 * DIM, DIM), n(3 * DIM)
        sd[p + "projection.weight"], sd[p + "projection.bias"] = n(DIM, DIM), n(DIM)
        sd[p + "mlp_layer_norm.weight"] = 1.0 + n(DIM, s=0.05)
        sd[p + "mlp_layer_norm.bias"] = n(DIM, s=0.05)
        sd[p + "mlp_1.weight"], sd[p + "mlp_1.bias"] = n(4 * DIM, DIM), n(4 * DIM)
        sd[p + "mlp_2.weight"], sd[p + "mlp_2.bias"] = n(DIM, 4 * DIM), n(DIM)
    extra = {"class_token": n(1, 1, DIM), "ln.weight": 1.0 + n(DIM, s=0.05), "ln.bias": n(DIM, s=0.05)}
    return sd, extra


def synthetic_clips(batch, frames, k, seed, device):
    """(T, B, 196, D) token clips: frame 0 ~ N(0,1); each later frame re-randomises exactly k patches per
    clip and jitters the rest by N(0, 0.01^2).  Generated on the device (torch.Generator)."""
    g = torch.Generator(device=device).manual_seed(seed)
    cur = torch.randn(batch, TOKENS, DIM, generator=g, device=device)
    out = [cur]
    for _ in range(1, frames):
        cur = cur + 0.01 * torch.randn(batch, TOKENS, DIM, generator=g, device=device)
        pick = torch.rand(batch, TOKENS, generator=g, device=device).argsort(dim=1)[:, :k]
        fresh = torch.randn(batch, k, DIM, generator=g, device=device)
        cur = cur.scatter(1, pick.unsqueeze(-1).expand(-1, -1, DIM), fresh)
        out.append(cur)
    return torch.stack(out)


class SpatialModel:
    """ViViTSubModel.forward of the reference (vivit.py:293-303) around our ViTBackbone."""

    def __init__(self, sd, extra, cast, k, device):
        from eventful_transformer import policies
        from eventful_transformer.backbones import ViTBackbone
        from eventful_transformer.modules import SimpleSTGTGate, TokenDeltaGate, TokenGate

        cfg = dict(dim=DIM, heads=HEADS, mlp_ratio=4)
        if cast:
            cfg["matmul_2_cast"] = cast
        bb = ViTBackbone(block_config=cfg, depth=DEPTH, position_encoding_size=(14, 14), input_size=(14, 14),
                         block_class="EventfulBlock", has_class_token=True)
        bb.load_state_dict(sd, strict=True)
        self.backbone = bb.eval().to(device)
        for cls in (SimpleSTGTGate, TokenDeltaGate, TokenGate):  # utils/misc.py:140-143
            for gate in self.backbone.modules_of_type(cls):
                gate.policy = policies.TokenNormTopK(k=k)
        self.class_token = extra["class_token"].to(device)
        self.ln_w, self.ln_b = extra["ln.weight"].to(device), extra["ln.bias"].to(device)
        self.graphs = None

    def reset(self):
        self.backbone.reset()

    def frame(self, x):
        from eventful_transformer import _native

        B = x.shape[0]
        x = torch.concat([self.class_token.expand(B, 1, DIM), x], dim=1)
        y = self.backbone(x)
        # LayerNorm is row-wise: normalising only the class-token rows equals layer_norm(y)[:, 0]
        cls_rows = y[:, 0].contiguous()
        out = torch.empty_like(cls_rows)
        _native.row_pass(cls_rows, B, DIM, ln_w=self.ln_w, ln_b=self.ln_b, eps=1e-6, c_out=out)
        return out

    def clip(self, clips):
        """vivit.py:146-147: reset, then one backbone call per time step."""
        if self.graphs is not None:  # HIP-graph replay of the same launches (eventful_transformer/graphs.py)
            self.graphs.reset()
            return torch.stack([self.graphs(clips[t]).clone() for t in range(clips.shape[0])], dim=1)
        self.reset()
        return torch.stack([self.frame(clips[t]) for t in range(clips.shape[0])], dim=1)

    def use_graphs(self):
        from eventful_transformer.graphs import FrameGraphs

        self.graphs = FrameGraphs(self.backbone, forward=self.frame)


def cpu_baseline(sd, extra, cast, k, frames, budget_s=20.0):
    """CPU column: the oracle (a torch-CPU port of the reference's op sequence, pinned bit-exact to the
    reference by tests/golden) on this box's host cores.  Bounded sample: whole single clips (B=1, T
    frames, first dense frame included) until ~budget_s, after one warm-up clip."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eventful_oracle as O

    threads = min(8, os.cpu_count() or 1)  # the reference's own CPU setting (configs/time/*/_cpu.yml: threads 8)
    torch.set_num_threads(threads)
    blocks = []
    for i in range(DEPTH):
        pre = f"blocks.{i}."
        params = {key[len(pre):]: v for key, v in sd.items() if key.startswith(pre)}
        blocks.append(O.BlockOracle("EventfulBlock", params, DIM, HEADS, (14, 14), matmul_2_cast=cast))
    bb = O.BackboneOracle(blocks, sd["position_encoding.encoding"], (14, 14), (14, 14), True)
    bb.set_policy(lambda: O.TopK(k))
    model = O.ViViTSpatialOracle(bb, extra["class_token"], extra["ln.weight"], extra["ln.bias"])
    clip = synthetic_clips(1, frames, k, 1234, torch.device("cpu"))
    with torch.inference_mode():
        def run():
            model.reset()
            for t in range(frames):
                model.forward(clip[t])
        run()
        n, t0 = 0, time.perf_counter()
        while True:
            run()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 50:
                break
    return {"value": round(n * frames / el, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} clips x {frames} frames, B=1, ViViT-B spatial k={k} cast={cast}, torch-CPU oracle, "
                      f"{threads} threads, {el:.1f}s"}


def broadcast_weights(sd, extra, device, rank):
    """Rank 0's weights -> every rank, as ONE flat-buffer broadcast (RCCL over xGMI on the GPU box; the
    same code runs on gloo/CPU in tests/test_dist_cpu.py).  Non-zero ranks' values are overwritten."""
    flat = torch.cat([v.reshape(-1) for v in list(sd.values()) + list(extra.values())]).to(device)
    if rank != 0:
        flat.zero_()
    dist.broadcast(flat, src=0)
    off = 0
    for d in (sd, extra):
        for key in d:
            n = d[key].numel()
            d[key] = flat[off:off + n].view_as(d[key]).cpu().clone()
            off += n


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def clips_for_rank(total_clips, world, rank):
    """Clip i -> rank i mod world; used when a FIXED clip set is split (strong scaling)."""
    return list(range(rank, total_clips, world))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=256, help="clips resident per GPU (B); 256 clips = ~32 GB of per-clip state")
    ap.add_argument("--frames", type=int, default=16, help="backbone frames per clip (T)")
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--cast", default="bfloat16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--graphs", action="store_true", help="replay HIP graphs of the per-frame launches (small --clips: "
                    "host-bound otherwise); implies --no-kernel-events")
    args = ap.parse_args()
    cast = None if args.cast in ("none", "fp32", "None") else args.cast

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from eventful_transformer import _native

    _native.load()  # fail loudly here if the HIP library is missing

    # Weights: rank 0 generates, RCCL broadcasts one flat buffer (the only start-up collective).
    sd, extra = seeded_state_dict()
    if world > 1:
        broadcast_weights(sd, extra, device, rank)
    model = SpatialModel(sd, extra, cast, args.k, device)
    if args.graphs:
        model.use_graphs()
        args.no_kernel_events = True
    clips = synthetic_clips(args.clips, args.frames, args.k, 1000 + rank, device)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for _ in range(args.warmup):
            model.clip(clips)
        events = None if args.no_kernel_events else []
        _native.GEMM_EVENTS = events
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.clip(clips)
        sync_all()
        elapsed = time.perf_counter() - t0
        _native.GEMM_EVENTS = None

    if world > 1:
        elapsed = max_over_ranks(elapsed, device)

    frames_total = world * args.clips * args.frames * args.steps
    value = frames_total / elapsed

    # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be
    # read in-process); profiles/*/pmc_traffic_*.json holds the per-launch figure for the workload it names.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", f"pmc_traffic_B{args.clips}.json")))
        wl = pmc["workload"]
        if (wl["clips"], wl["frames"], wl["k"], wl["cast"], wl["gemm"]) == (args.clips, args.frames, args.k, cast, _native.GEMM_MODE):
            traffic = pmc["gated_linear_hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = None
    if events:
        ms = sum(ev[0].elapsed_time(ev[1]) for ev in events)
        flops = sum(ev[2] for ev in events)
        launches = sum(ev[3] for ev in events)
        achieved = flops / (ms * 1e-3) / 1e12
        split = _native.GEMM_MODE == "split"
        # split mode: each fp32 product = 3 bf16 MFMA products (hi.hi + hi.lo + lo.hi).  `achieved` stays
        # ALGORITHMIC (2*M*K*N per launch) and is priced against the bf16 dense peak, so 1/3 is the ceiling
        # of `frac`; `mfma_issue_frac` = issued bf16 MFMA FLOP/s over the same peak (matrix-pipe utilisation).
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": ("gated_linear_split_kernel" if split else "gated_linear_kernel") +
                              " (evt_gated_linear / evt_gated_mlp)",
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes/launch from rocprofv3 PMC (2*FETCH_SIZE+WRITE_SIZE), profiles/r01/pmc_traffic_B<clips>.json",
                    "arith": "bf16x3 split MFMA, fp32 accumulate" if split else "fp32-input MFMA",
                    "mfma_issue_frac": round(achieved * (3 if split else 1) / peak, 4),
                    "launches": launches, "avg_launch_us": round(ms * 1e3 / launches, 2),
                    "share_of_step_time": round(ms * 1e-3 / elapsed, 3)}

    if rank == 0:
        line = {
            "metric": METRIC, "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("f32" + (" via bf16x3 split MFMA" if _native.GEMM_MODE == "split" else "") +
                                                  (" (A.v stage bf16 = reference matmul_2_cast)" if cast else "")),
            "data": "synthetic", "per_gpu": round(value / world, 2),
            "config": {"workload": f"ViViT-B spatial {args.frames}x224^2 (N=197, D=768, 12 EventfulBlocks) top-k r={args.k}, "
                                   f"T={args.frames} frames/clip incl. dense first frame, matmul_2_cast={cast}",
                       "clips_per_gpu": args.clips, "frames_per_step": args.clips * args.frames,
                       "parallelism": f"clip-sharded x{world}", "launch": "hip-graph replay" if args.graphs else "eager"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, extra, cast, args.k, args.frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
