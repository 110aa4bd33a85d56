mkdir -p gpurun_out/r2l
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r2l/pmc_$tag -- python3 $R/tools/kernel_bench.py --only zoom --reps 5 "$@" > $R/gpurun_out/r2l/pmc_$tag.log 2>&1 || echo "group failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/r2l zoom > gpurun_out/r2l/summary.json; cat gpurun_out/r2l/summary.json
