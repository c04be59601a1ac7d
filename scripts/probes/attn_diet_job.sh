mkdir -p gpurun_out/p51
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fused or softmax or attention" 2>&1 | tail -5 | tee gpurun_out/p51/pytest.txt
for r in 1 2; do python scripts/kbench.py --clips 256 --only softmax_av_fused_qk_norm_noout,softmax_av_fused_qk_norm,softmax_av_fused 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/p51/kb.txt; done
