cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc4
mkdir -p $OUT
for mode in bf16 f32_bf16x6; do
i=0
for set in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
i=$((i+1))
timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${mode}_$i -- python3 $R/tools/pmc_probe.py gemm $mode > $OUT/${mode}_$i.log 2>&1
echo "$mode set $i rc=$?"
done
done
python3 $R/tools/pmc_summary.py $OUT gemm_bf16x > $OUT/summary.txt
cat $OUT/summary.txt
