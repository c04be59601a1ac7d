mkdir -p gpurun_out/p72
for r in 1 2 3; do for v in ship new; do
  if [ $v = ship ]; then L=$PWD/eventful-transformer_amd/eventful_transformer/libevt_hip.so; else L=$PWD/scripts/probes/bin/libevt_$v.so; fi
  echo "== $v run $r" | tee -a gpurun_out/p72/kb.txt
  EVT_LIB=$L python scripts/kbench.py --clips 256 --only softmax_av_fused_qk_norm_noout 2>&1 | grep -v "amdgpu.ids\|^#" | tee -a gpurun_out/p72/kb.txt
done; done
