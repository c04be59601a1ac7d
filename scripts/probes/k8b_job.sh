mkdir -p gpurun_out/k8b
for v in base ship; do
  if [ $v = base ]; then export EVT_LIB=$PWD/scripts/probes/bin/libevt_base.so; else unset EVT_LIB; fi
  python scripts/onestream_bench.py --only window > gpurun_out/k8b/onestream_$v.txt 2>&1
  python scripts/kbench.py --clips 256 --only attention_dense > gpurun_out/k8b/kb_bf16_$v.txt 2>&1
  python scripts/kbench.py --clips 256 --cast none --only attention_dense > gpurun_out/k8b/kb_f32_$v.txt 2>&1
done
unset EVT_LIB
timeout 900 python -m pytest tests -m gpu -q -x -k "dense or window or vitdet or winpool or Block_win or small_blocks" > gpurun_out/k8b/tests.txt 2>&1
tail -3 gpurun_out/k8b/tests.txt
for f in gpurun_out/k8b/*_base.txt gpurun_out/k8b/*_ship.txt; do echo == $f; tail -8 $f; done
