# PMC passes of the fused warp kernel (tools/kernel_bench.py --only warp_f32_f32_epi_ws): counters only, one group per pass
mkdir -p gpurun_out/r4p
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r4p/counters_list.txt 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r4p/pmc_$tag -- python3 $R/tools/kernel_bench.py --only warp_f32_f32_epi_ws,warp_f32_u8tof32,warp_lin_only_ws,warp_nn_f32_only_ws --reps 5 "$@" > $R/gpurun_out/r4p/pmc_$tag.log 2>&1 || echo "group failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/r4p warp_ > gpurun_out/r4p/summary.json; head -c 3000 gpurun_out/r4p/summary.json
