mkdir -p gpurun_out/r6o
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ[A-Z_0-9]*ICACHE[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_ACTIVE_INST_[A-Z_0-9]*\|SQ_INST_CYCLES_[A-Z_0-9]*\|SQ_WAIT_INST_LDS\|SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE\|SQ_WAVES\b\|SQC_INST[A-Z_0-9]*" | sort -u | tr '\n' ' '
echo
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  RO_ONLY=1 RO_STREAMS=1 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r6o/pmc_$tag -o p -- python $R/tools/ro_probe.py 64 2 0 > /dev/null 2> $R/gpurun_out/r6o/err_$tag.txt
  python $R/tools/rocpd_summary.py $(ls $R/gpurun_out/r6o/pmc_$tag/*.db | head -1) --pmc 2>/dev/null | grep "k_ro_step" | head -8
done
