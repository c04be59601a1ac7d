mkdir -p gpurun_out/p38
timeout 1500 python -m pytest tests -m gpu -q -x -k "row_pass or big_tile or gated_linear or gated_mlp" 2>&1 | tail -12 | tee gpurun_out/p38/pytest.txt
python scripts/kbench.py --clips 256 --only row_pass_ln_norm,row_pass_ln_norm_planes,linear_qkv,linear_qkv_planes,mlp,mlp_planes,linear_mlp1_gelu 2>&1 | grep -v amdgpu.ids | tee gpurun_out/p38/kb.txt
