mkdir -p gpurun_out/p49
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "in_kernel_qk or attention_dense" 2>&1 | tail -6 | tee gpurun_out/p49/pytest.txt
for r in 1 2; do python scripts/kbench.py --clips 256 --only softmax_av_fused_qk_norm_noout,softmax_av_fused_qk_norm_noout_T 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/p49/kb.txt; done
