cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
rocprofv3 --list-avail > $R/gpurun_out/pmc3/list_avail.txt 2>&1
for mode in f32_bf16x6; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
i=$((i+1))
timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc3/${mode}_$i -- python3 $R/tools/pmc_probe.py gemm $mode > $R/gpurun_out/pmc3/${mode}_$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3 gemm_bf16x > $R/gpurun_out/pmc3/summary_$mode.txt
done
grep -A40 "true, true, true> grid=1310720" $R/gpurun_out/pmc3/summary_f32_bf16x6.txt
tail -3 $R/gpurun_out/pmc3/*_4.log
