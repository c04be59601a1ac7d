#!/usr/bin/env python3
"""Aggregate the rocprofv3 --pmc passes of scripts/collect_profiles.sh per kernel: HBM traffic (FETCH_SIZE / WRITE_SIZE with
the gfx950 correction of MI355X_MICROARCH.md) and matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES)."""
import collections, csv, glob, hashlib, json, os, re, sys

out, B = sys.argv[1], int(sys.argv[2])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["gated_linear_split_kernel<0", "gated_linear_split_kernel<1", "gated_linear_kernel<0", "gated_linear_kernel<1",
         "attn_gated_kernel", "softmax_av_gated_kernel", "qk_kernel", "qk_split_kernel", "row_pass_kernel", "v_gate_t_kernel", "v_gate_kernel", "select_kernel",
         "av_kernel", "softmax_gate_kernel", "split_weights_kernel", "attn_dense_kernel", "splitk_finish_kernel"]


BIG = re.compile(r"gated_linear_pipe_kernel<(\d), (\d+), (\d), \d>")   # <ACT, TBN, FMT, WAVES> (evt_linear_pipe.hip)
FMT = {"0": "fp32", "1": "presplit_A", "2": "hl32_out", "4": "bf16_A"}


def key(n):
    m = BIG.search(n)   # the persistent 256-row kernel: one row per (activation, tile width, operand format)
    if m:
        k = f"gated_linear_pipe_kernel<{m.group(1)}|256x{m.group(2)}|{FMT.get(m.group(3), m.group(3))}"
        if k not in NAMES:
            NAMES.insert(0, k)
        return k
    for k in NAMES:
        if k in n:
            return k


def agg(tag):
    """{counter: {kernel: [launches, sum value, sum ns]}} of one pass"""
    res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0]))
    for path in glob.glob(f"{out}/pmc_{tag}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = key(r["Kernel_Name"])
            if k:
                t = res[r["Counter_Name"]][k]
                t[0] += 1
                t[1] += float(r["Counter_Value"])
                t[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return res


f = agg("FETCH_SIZE")["FETCH_SIZE"]
w = agg("WRITE_SIZE")["WRITE_SIZE"]
sq = agg("SQ_VALU_MFMA_BUSY_CYCLES")
gr = agg("GRBM_GUI_ACTIVE")["GRBM_GUI_ACTIVE"]
rows = []
for n in list(NAMES):
    if n in f and n in w:
        c, v, t = f[n]
        wc, wv, _ = w[n]
        row = dict(kernel=n.replace("<0", "<ACT_NONE").replace("<1", "<ACT_GELU").replace("|", ", ") + (">" if "<" in n else ""), launches=c,
                   fetch_size_kb_raw=round(v / c, 1), write_size_kb=round(wv / wc, 1),
                   hbm_bytes_per_launch=int((2 * v / c + wv / wc) * 1024), avg_us_profiled=round(t / c / 1e3, 1))
        mb = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get(n)
        if mb and mb[1] > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (32 per 32x32x16 bf16 MFMA); GRBM_GUI_ACTIVE is
            # reported summed over the 8 XCDs: / 8 = shader cycles the launch was active (checks out against the profiled
            # duration at ~2.0 GHz, the clock this chip sustains under the MFMA load)
            row["mfma_busy_cycles"] = round(mb[1] / mb[0])
            ga = gr.get(n)
            if ga and ga[1] > 0:
                cyc = ga[1] / ga[0] / 8
                row["active_cycles"] = round(cyc)
                row["clock_ghz_profiled"] = round(cyc / (ga[2] / ga[0]), 3)
                row["mfma_util_pmc"] = round((mb[1] / mb[0]) / (cyc * 1024), 4)
            row["mfma_bf16_mops"] = round(sq.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", {}).get(n, [1, 0, 0])[1] / max(1, mb[0]))
        rows.append(row)
g = [r for r in rows if r["kernel"].startswith("gated_linear")]
gem = int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in g) / max(1, sum(r["launches"] for r in g)))
h = hashlib.sha256()
for src in sorted(glob.glob(os.path.join(ROOT, "eventful-transformer_amd", "csrc", "evt_linear*.hip"))):
    h.update(open(src, "rb").read())
json.dump(dict(
    command=f"rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --clips {B} --total-clips {B} --steps 1 --warmup 1 "
            "--no-cpu-baseline --no-check --no-exact --no-kernel-events  (one separate pass per counter group: FETCH_SIZE; WRITE_SIZE; "
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16; GRBM_GUI_ACTIVE)",
    correction="hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) "
               "coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected; Infinity-Cache hits are counted",
    mfma_util="mfma_util_pmc = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs), averaged per launch: the "
              "fraction of the launch's shader cycles in which a SIMD's matrix pipe is busy, at the clock the chip actually ran",
    workload=dict(clips=B, frames=16, k=128, cast="bfloat16", gemm="split"), gemm_source_sha16=h.hexdigest()[:16],
    gated_linear_hbm_bytes_per_launch=gem, kernels=rows), open(f"{out}/pmc_traffic_B{B}.json", "w"), indent=1)
print("GEMM HBM bytes/launch:", gem)
for r in rows:
    print(r)
