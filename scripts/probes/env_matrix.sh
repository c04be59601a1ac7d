# The block / model parity tests under each non-default switch (the alternative code paths must stay correct).
mkdir -p gpurun_out/envm
for cfg in "EVT_GEMM_BIG=0" "EVT_GEMM_SMALL=0" "EVT_STREAM_PREP=0" "EVT_STREAM_QK=0" "EVT_FUSED_QK=0" "EVT_DENSE_FUSED=0" "EVT_DENSE_TILED=1" \
           "EVT_FUSE_PROJ_NORM=0" "EVT_PROJ_FROM_STATE=0" "EVT_REL_TERMS=0" "EVT_QK_SPLIT=0" "EVT_GEMM=f32" "EVT_PREFETCH=0" "EVT_STREAM_POOLED=0" \
           "EVT_FUSE_DENSE_NORM_ROWS=0" "EVT_CHAIN_BLOCKS=0"; do
  echo "== $cfg" | tee -a gpurun_out/envm/matrix.txt
  env $cfg timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_models.py tests/test_gpu_modules.py -m gpu -q -x \
      -k "not sharp_bf16_projection and not forced_big" 2>&1 | tail -4 | tee -a gpurun_out/envm/matrix.txt
done
