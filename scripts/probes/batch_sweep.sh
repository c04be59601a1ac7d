mkdir -p gpurun_out/p41
for b in 128 256 512 1024; do
  echo "== --clips $b --total-clips 2048 --overlap 2" | tee -a gpurun_out/p41/sweep.txt
  python bench.py --clips $b --total-clips 2048 --overlap 2 --steps 4 --warmup 1 --no-other --no-cpu-baseline --no-check --no-exact --no-kernel-events 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/p41/sweep.txt
done
