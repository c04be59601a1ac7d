#!/bin/bash
# GEMM work, run ON the GPU box: correctness of the persistent 256-row kernel (evt_linear_pipe.hip, 8 waves) under every forced
# tile, then per-launch times.   $1 = tag, rest = steps (check, kb, kbv=<lib variants>)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
KB=${KB:-"linear_qkv,linear_qkv_nop,linear_proj_bf16,linear_mlp1_gelu,linear_mlp2,mlp,linear_dense_qkv"}
for what in "$@"; do
case $what in
  check)
    for mode in 2 3 4; do
      echo "== forced tile mode $mode" | tee -a $OUT/check.txt
      EVT_GEMM_BIG=$mode EVT_GEMM=split EVT_GEMM_SMALL=0 timeout 600 python tests/big_tile_check.py 2>&1 | tail -4 | grep -v amdgpu.ids | tee -a $OUT/check.txt
    done ;;
  kb)
    python scripts/kbench.py --clips 256 --only $KB 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kb.txt ;;
  kbv=*)   # build variants under scripts/probes/bin/libevt_<v>.so
    for v in $(echo ${what#kbv=} | tr ',' ' '); do
      echo "== variant $v" | tee -a $OUT/kb.txt
      EVT_LIB=$PWD/scripts/probes/bin/libevt_$v.so python scripts/kbench.py --clips 256 --only $KB 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kb.txt
    done ;;
  prof=*)  # phase profile of the pipe kernel (variant library built with -DEVT_PROF): prof=<variant>
    for shp in qkv mlp1 mlp2 proj; do
      EVT_LIB=$PWD/scripts/probes/bin/libevt_${what#prof=}.so python scripts/gemm_prof.py --shape $shp 2>&1 | grep -v amdgpu.ids | tee -a $OUT/prof.txt
    done ;;
  pmc)     # hardware counters of ONE GEMM shape (default QKV: 32768 x 768 x 2304) per kernel generation, separate passes
    (cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters_available.txt 2>&1)
    grep -o "^\s*[A-Z][A-Za-z0-9_]*" $OUT/counters_available.txt | sort -u | tr -d ' \t' | tr '\n' ' ' > $OUT/counter_names.txt
    SHAPE=${SHAPE:-"32768 768 2304 12 0"}
    for cfg in "pipe"; do
      tag=$cfg
      i=0
      for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
               "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
               "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
               "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
               "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
               "GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_${tag}_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_only.py $SHAPE > $GRAFT_REPO_ROOT/$OUT/pmc_${tag}_$i.log 2>&1) || echo "pass $i ($c) failed" | tee -a $OUT/pmc.txt
      done
      echo "== $tag" | tee -a $OUT/pmc.txt
      python scripts/pmc_kernel.py $OUT gated_linear 2>&1 | tee -a $OUT/pmc.txt
      find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete; rm -rf $OUT/pmc_${tag}_*
    done
    ;;
  tiles)   # forced tile shapes on the headline launches (EVT_GEMM_BIG: 2 = 256x256, 4 = 256x192, 3 = 256x128)
    for mode in 2 4 3; do
      echo "== EVT_GEMM_BIG=$mode" | tee -a $OUT/tiles.txt
      EVT_GEMM_BIG=$mode python scripts/kbench.py --clips 256 --only linear_qkv,linear_mlp1_gelu,mlp 2>&1 | grep -v "amdgpu.ids\|^#" | tee -a $OUT/tiles.txt
    done ;;
  gl)      # the kernel-level GEMM tests
    timeout 900 python -m pytest tests -m gpu -q -x -k "gated_linear or gated_mlp or big_tiles or operating_point" 2>&1 | tail -5 | tee -a $OUT/gl.txt ;;
  *) echo "unknown step $what" ;;
esac
done
