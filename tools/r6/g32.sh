mkdir -p gpurun_out/r6af
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TCC|TCP|TA|TD|SQ|GRBM|SQC)_[A-Z0-9_]+(_sum)?\b" | sort -u | tr '\n' ' ' > gpurun_out/r6af/avail.txt
wc -c gpurun_out/r6af/avail.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  RO_ONLY=1 RO_STREAMS=1 timeout 600 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r6af/pmc_$i -o p -- python $R/tools/ro_probe.py 256 2 0 > /dev/null 2> $R/gpurun_out/r6af/err_$i.txt
  python $R/tools/rocpd_summary.py $(ls $R/gpurun_out/r6af/pmc_$i/*.db | head -1) --pmc 2>/dev/null | grep -E "k_colpass_lds|k_rowpass_lds|k_ro_step" > $R/gpurun_out/r6af/set_$i.txt
  rm -rf $R/gpurun_out/r6af/pmc_$i
  cat $R/gpurun_out/r6af/set_$i.txt | cut -c1-60,100-220
done
