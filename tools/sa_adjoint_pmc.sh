# usage (GPU box): bash tools/sa_adjoint_pmc.sh <tag>   -- SQ instruction mix of sa_adjoint_nn_lds_kernel on one random stack (counter-only passes)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  t=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$t -- python3 $R/tools/sa_adjoint_one.py > $out/pmc_$t.log 2>&1 || echo "group failed: $grp"
done
cd $R && python tools/pmc_summary.py $out sa_adjoint_nn_lds > $out/summary.json; cat $out/summary.json; cat $out/pmc_*.log | grep pixel_taps | head -1
