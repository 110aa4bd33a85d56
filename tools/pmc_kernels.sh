# usage (GPU box): bash tools/pmc_kernels.sh <tag> <kernel name filter> <kernel_bench args...>
# SQ instruction mix / waits / LDS conflicts of the kernels `tools/kernel_bench.py <args>` launches (counter-only passes).
tag=$1; shift; filt=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$t -- python3 $R/tools/kernel_bench.py "$@" > $out/pmc_$t.log 2>&1 || echo "group failed: $grp"
done
cd $R && python tools/pmc_summary.py $out $filt > $out/summary.json; cat $out/summary.json
