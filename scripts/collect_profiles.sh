#!/bin/bash
# Run ON the GPU box (via gpurun): bench line + rocprofv3 kernel stats + separate PMC passes (HBM traffic, MFMA busy)
# for the default bench workload at one resident batch.  Writes under gpurun_out/prof/ (copied into profiles/rNN/).
#   gpurun -- 'bash scripts/collect_profiles.sh 256'
set -u
B=${1:-256}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
SHORT="--clips $B --total-clips $B --no-cpu-baseline --no-check --no-exact --no-other --no-kernel-events --overlap 1"
python bench.py --steps 5 --warmup 2 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
# the same command as the headline line (default steps/warmup), minus the legs that run other workloads or the CPU
# (--overlap 1: batches one after the other, so that a kernel's duration is its own -- the same serial order bench.py's roofline pass uses)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-exact --no-other --overlap 1 > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o p --output-format csv -- python3 bench.py $SHORT --steps 1 --warmup 1 > $OUT/pmc_$n.log 2>&1
done
python scripts/pmc_summary.py $OUT $B
# the raw per-dispatch CSVs are large; keep only the aggregates
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete
ls -la $OUT $OUT/stats
cat $OUT/bench_default.json
