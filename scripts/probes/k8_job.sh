mkdir -p gpurun_out/p63
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dense or in_kernel_qk" 2>&1 | tail -4 | tee gpurun_out/p63/pytest.txt
python scripts/kbench.py --clips 256 2>&1 | grep -v amdgpu.ids | grep -i "dense\|#" | tee gpurun_out/p63/kb.txt
