# Practical dense bf16 peak + shader clock + matrix-pipe busy of a hipBLASLt GEMM on this box (separate PMC passes).
OUT=gpurun_out/dense_peak; mkdir -p $OUT; export TMPDIR=/tmp
for shape in "8192 8192 8192" "32768 2304 768" "32768 768 3072"; do
  python scripts/probes/dense_bf16_peak.py $shape 2>&1 | grep -v amdgpu.ids | tee -a $OUT/peak.txt
done
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d" " -f1)
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$n -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probes/dense_bf16_peak.py 8192 8192 8192 > $GRAFT_REPO_ROOT/$OUT/pmc_$n.log 2>&1)
done
python - <<'PY' | tee -a gpurun_out/dense_peak/peak.txt
import csv, glob, collections
root = "gpurun_out/dense_peak"
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(root + "/pmc_*/"):
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True); cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    for f in kt:
        for r in csv.DictReader(open(f)):
            if "Cijk" in r["Kernel_Name"]: dur[d].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in cc:
        for r in csv.DictReader(open(f)):
            if "Cijk" in r["Kernel_Name"]: cnt[d][r["Counter_Name"]].append(float(r["Counter_Value"]))
for d in cnt:
    us = sum(dur[d]) / max(1, len(dur[d]))
    for c, v in cnt[d].items():
        m = sum(v) / len(v)
        print(f"{c}: {m:.0f} per launch over {us:.1f} us" + (f" -> shader clock {m / 8 / us / 1e3:.3f} GHz (GRBM_GUI_ACTIVE sums the 8 XCDs)" if c == "GRBM_GUI_ACTIVE" else ""))
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
