mkdir -p gpurun_out/p30
timeout 1500 python -m pytest tests -m gpu -q -x -k "stream or pooled or timing_configs_vitdet or abi or version" 2>&1 | tail -8 | tee gpurun_out/p30/pytest.txt
for v in 1 0; do
  echo "== EVT_STREAM_POOLED=$v" | tee -a gpurun_out/p30/bench.txt
  EVT_STREAM_POOLED=$v python bench.py --workload vitdet672_pool2 --no-cpu-baseline --no-other 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/p30/bench.txt
done
