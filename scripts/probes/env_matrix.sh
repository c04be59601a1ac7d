#!/bin/bash
# The block / model / module parity tests under the two arithmetic modes and the forced GEMM routings -- the only switches the
# library still reads from the environment (EVT_GEMM; EVT_GEMM_BIG / EVT_GEMM_SMALL: test routing of tests/test_gpu_big_tiles.py).
# The alternative host paths (K4 + stored score state, the K4/K5/K6 chain for pooled blocks, un-chained blocks) are module
# constants that the tests flip themselves (test_lazy_qk_state_is_exact_when_read, test_pooled_eventful_block_on_the_stream_kernel, ...).
mkdir -p gpurun_out/envm
for cfg in "EVT_GEMM_BIG=0" "EVT_GEMM_SMALL=0" "EVT_GEMM=f32"; do
  echo "== $cfg" | tee -a gpurun_out/envm/matrix.txt
  env $cfg timeout 1200 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_models.py tests/test_gpu_modules.py -m gpu -q -x \
      -k "not sharp_bf16_projection and not forced_big" 2>&1 | tail -4 | tee -a gpurun_out/envm/matrix.txt
done
