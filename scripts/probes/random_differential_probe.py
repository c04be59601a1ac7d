#!/usr/bin/env python3
"""Probe: randomised differential testing of the block kinds against the CPU oracle -- random widths (head dim 64 x 1..6 heads, or head dims 16/32/48/80),
token grids, k, batch, block kind, options (class token, rel-pos, pooling, windows, gate_before_ln, stgt, casts), policy kinds.  fp32 streams with a
designed qkv-gate margin; a miss is reported with the gate sets (fork at a near-tie or a bug?)."""
import os, sys, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("eventful-transformer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import eventful_oracle as O
import helpers as H
from eventful_transformer import policies, blocks as EB
torch.set_num_threads(8)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for case in range(n_cases):
    dh = rng.choice([64, 64, 64, 16, 32, 48, 80])
    heads = rng.choice([2, 2, 3, 4, 6])   # (one head: the reference asserts, blocks.py:341)
    dim = dh * heads
    kind = rng.choice(["EventfulTokenwiseBlock", "EventfulMatmul1Block", "EventfulBlock", "EventfulBlock"])
    gh, gw = rng.randint(1, 24), rng.randint(1, 24)
    kw = {}
    opt = rng.choice(["plain", "plain", "cls", "rel", "pool", "win", "gbl", "stgt"])
    cls = opt == "cls"
    if opt == "rel":
        kw["relative_embedding_size"] = (gh, gw)
    if opt == "pool" and kind != "EventfulTokenwiseBlock":
        p0, p1 = rng.choice([1, 2, 3]), rng.choice([1, 2, 3])
        gh, gw = max(p0, gh - gh % p0), max(p1, gw - gw % p1)
        kw["pool_size"] = (p0, p1)
    if opt == "win" and kind == "EventfulTokenwiseBlock":
        kw["window_size"] = (rng.randint(2, 8), rng.randint(2, 8))
        if rng.random() < 0.5:
            kw["relative_embedding_size"] = kw["window_size"]
    if opt == "gbl" and kind == "EventfulBlock":
        kw["gate_before_ln"] = True
    if opt == "stgt" and kind == "EventfulTokenwiseBlock":
        kw["stgt"] = True
    cast = rng.choice([None, "bfloat16", "bfloat16", "float16"]) if kind != "EventfulTokenwiseBlock" else None
    if cast:
        kw["matmul_2_cast"] = cast
    n = gh * gw + int(cls)
    B = rng.choice([1, 1, 2, 3])
    if "pool_size" in kw:
        B = 1   # (the reference couples pooled clips of a batch: DESIGN section 2)
    pk = rng.choice(["topk", "topk", "topk", "thr", "frac"])
    k = rng.randint(1, n)
    desc = f"#{case} {kind} dim {dim} (dh {dh}) grid {gh}x{gw}{'+cls' if cls else ''} B {B} {kw} policy {pk} k {k}"
    try:
        rel = kw.get("relative_embedding_size")
        params = O.make_block_params(dim, 4, seed=case, std=0.05, rel_sizes=rel, head_dim=dh)
        ob = O.BlockOracle(kind, params, dim, heads, (gh, gw), **kw)
        blk = H.product_block(kind, params, dim, heads, (gh, gw), **kw)
        if pk == "topk":
            ob.set_policy(lambda: O.TopK(k)); H.set_policies(blk, policies.TokenNormTopK, k=k)
        elif pk == "frac":
            fr = k / n
            ob.set_policy(lambda: O.TopFraction(fr)); H.set_policies(blk, policies.TokenNormTopFraction, fraction=fr)
        else:
            if B != 1:
                B = 1
            ob.set_policy(lambda: O.Threshold(0.3)); H.set_policies(blk, policies.TokenNormThreshold, threshold=0.3)
        xs = O.make_token_stream(B, n, dim, 4, k, seed=case + 1000, small=0.01)
        tol = 3e-4 if cast is None else (5e-3 if cast == "float16" else 4e-2)
        errs, forks, margins = [], 0, []
        # TEACHER-FORCED decisions: the product's gates record their own selection and are handed the oracle's (a fork at a near-tie would
        # otherwise hide everything behind it); the recorded selections are compared where the oracle's margin allows
        gates = ("qkv_gate", "projection_gate", "mlp_gate")
        class Forced(torch.nn.Module):
            def __init__(self, real):
                super().__init__()
                self.real, self.force, self.mine = real, None, None
            def forward(self, e, dim=-1):
                self.mine = self.real(e, dim=dim)
                return self.force
        for gn in gates:
            getattr(blk, gn).policy = Forced(getattr(blk, gn).policy)
        with torch.inference_mode():
            for t in range(4):
                y_ref = ob.forward(xs[t])
                if t:
                    for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                        getattr(blk, gn).policy.force = ob.trace[tk].sort(dim=-1)[0].cuda()
                y = blk(xs[t].cuda()).cpu()
                errs.append(float((y - y_ref).abs().max()))
                if t:
                    for gn, tk in zip(gates, ("qkv_index", "projection_index", "mlp_index")):
                        mine = getattr(blk, gn).policy.mine
                        if mine is not None and not torch.equal(mine.sort(dim=-1)[0].cpu(), ob.trace[tk].sort(dim=-1)[0]):
                            forks += 1
                            if pk == "topk":
                                kk = ob.trace[tk].shape[-1]
                                nrm = torch.linalg.vector_norm(ob.policy[gn].last_input.double(), dim=-1).sort(dim=-1, descending=True)[0]
                                if 0 < kk < nrm.shape[-1]:
                                    margins.append(float(((nrm[..., kk - 1] - nrm[..., kk]) / nrm[..., kk - 1]).min()))
        ok = max(errs) <= tol and bool(torch.isfinite(y).all())
        if not ok:
            bad += 1
            print("MISS", desc, ["%.1e" % e for e in errs], f"(own selections differing from the oracle's: {forks})", flush=True)
        elif forks:
            print("fork(s) only", desc, forks, "oracle margins at the differing gates:", ["%.1e" % m for m in margins], flush=True)
            if margins and max(margins) > 1e-3:
                bad += 1
                print("   ^^^ a selection differs at a LARGE margin", flush=True)
    except Exception as e:
        msg = str(e)
        expected = ("head dim" in msg) or ("non-square" in msg)
        if not expected:
            bad += 1
            print("RAISED", desc, type(e).__name__, msg[:200], flush=True)
print(f"{n_cases} random cases, {bad} to look at", flush=True)
