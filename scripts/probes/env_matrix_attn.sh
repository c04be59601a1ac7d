# The block / model parity tests under the switches that route through other paths of the fused attention kernel.
mkdir -p gpurun_out/envm2
for cfg in "EVT_FUSED_QK=0" "EVT_QK_SPLIT=0" "EVT_REL_TERMS=0" "EVT_FUSE_PROJ_NORM=0" "EVT_PROJ_FROM_STATE=0" "EVT_DENSE_FUSED=0"; do
  echo "== $cfg" | tee -a gpurun_out/envm2/matrix.txt
  env $cfg timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_models.py -m gpu -q \
      -k "not sharp_bf16_projection and not forced_big and not vivit_b_full_size" 2>&1 | tail -3 | tee -a gpurun_out/envm2/matrix.txt
done
