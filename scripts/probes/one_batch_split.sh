mkdir -p gpurun_out/p40
for cfg in "--clips 256 --total-clips 256 --overlap 1" "--clips 128 --total-clips 256 --overlap 2" "--clips 128 --total-clips 256 --overlap 1" "--clips 256 --total-clips 256 --overlap 2"; do
  for rep in 1 2; do
    echo "== $cfg (rep $rep)" | tee -a gpurun_out/p40/split.txt
    python bench.py $cfg --steps 20 --warmup 3 --no-other --no-cpu-baseline --no-check --no-exact --no-kernel-events 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/p40/split.txt
  done
done
