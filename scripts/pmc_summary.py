#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (run by scripts/collect_profiles.sh)."""
import collections, csv, json, sys

out, B = sys.argv[1], int(sys.argv[2])
NAMES = ["gated_linear_split_kernel<0", "gated_linear_split_kernel<1", "gated_linear_kernel<0", "gated_linear_kernel<1",
         "softmax_av_gated_kernel", "qk_kernel", "row_pass_kernel", "v_gate_t_kernel", "v_gate_kernel", "select_kernel",
         "av_kernel", "softmax_gate_kernel", "split_weights_kernel", "attn_dense_kernel", "splitk_finish_kernel"]


def key(n):
    for k in NAMES:
        if k in n:
            return k


def agg(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(path)):
        k = key(r["Kernel_Name"])
        if r["Counter_Name"] == counter and k:
            t = tot[k]
            t[0] += 1
            t[1] += float(r["Counter_Value"])
            t[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return tot


f = agg(f"{out}/fetch/f_counter_collection.csv", "FETCH_SIZE")
w = agg(f"{out}/write/w_counter_collection.csv", "WRITE_SIZE")
rows = []
for n in NAMES:
    if n in f and n in w:
        c, v, t = f[n]
        wc, wv, _ = w[n]
        rows.append(dict(kernel=n.replace("<0", "<ACT_NONE>").replace("<1", "<ACT_GELU>"), launches=c,
                         fetch_size_kb_raw=round(v / c, 1), write_size_kb=round(wv / wc, 1),
                         hbm_bytes_per_launch=int((2 * v / c + wv / wc) * 1024), avg_us_profiled=round(t / c / 1e3, 1)))
g = [r for r in rows if r["kernel"].startswith("gated_linear")]
gem = int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in g) / max(1, sum(r["launches"] for r in g)))
json.dump(dict(
    command=f"rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --clips {B} --steps 1 --warmup 1 --no-cpu-baseline "
            "--no-kernel-events  (second, separate pass with --pmc WRITE_SIZE)",
    correction="hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) "
               "coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected; Infinity-Cache hits are counted",
    workload=dict(clips=B, frames=16, k=128, cast="bfloat16", gemm="split"),
    gated_linear_hbm_bytes_per_launch=gem, kernels=rows), open(f"{out}/pmc_traffic_B{B}.json", "w"), indent=1)
print("GEMM HBM bytes/launch:", gem)
