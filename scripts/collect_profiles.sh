#!/bin/bash
# Run ON the GPU box (via gpurun): bench line + rocprofv3 kernel stats + the two PMC passes for the default
# bench workload.  Writes everything under gpurun_out/final/ (copied into profiles/r01/ afterwards).
set -u
B=${1:-256}
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --clips $B --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_B$B.json
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --clips $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 bench.py --clips $B --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- python3 bench.py --clips $B --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/write.log 2>&1
python scripts/pmc_summary.py $OUT $B
# the raw per-dispatch counter CSVs are large; keep only the aggregates
rm -f $OUT/fetch/*_kernel_trace.csv $OUT/write/*_kernel_trace.csv $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv $OUT/stats/s_kernel_trace.csv
ls -la $OUT $OUT/stats
cat $OUT/bench_B$B.json
