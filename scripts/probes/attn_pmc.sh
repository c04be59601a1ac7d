# Hardware counters of the gated attention launch K10 (scripts/kbench.py --only gated_resident_norm_noout), separate passes.
# ($1 = kbench case, $2 = kernel-name substring: defaults below; the round-5 kernel: softmax_av_fused_qk_norm_noout softmax_av_gated)
CASE=${1:-gated_resident_norm_noout}; KERN=${2:-attn_gated}
OUT=gpurun_out/attn_pmc; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
         "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py --clips 256 --only $CASE > $GRAFT_REPO_ROOT/$OUT/pmc_$i.log 2>&1) || echo "pass $i ($c) failed" | tee -a $OUT/pmc.txt
done
python scripts/pmc_kernel.py $OUT $KERN 2>&1 | tee -a $OUT/pmc.txt
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete
