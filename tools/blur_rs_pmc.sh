# FETCH_SIZE / WRITE_SIZE of the fused blur+resample kernels (separate counters-only passes), 256^3 -> gpurun_out/<tag>/blur_rs_pmc_256.json
tag=${1:-blurrspmc}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for size in 256; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_${size}_$ctr -- python3 $R/tools/kernel_bench.py --only blur_rs_x,blur_rs_yz_m --reps 5 --size $size > $R/gpurun_out/$tag/pmc_${size}_$ctr.log 2>&1 || echo "failed $size $ctr"
  done
  (cd $R && python tools/pmc_summary.py gpurun_out/$tag blur_rs > gpurun_out/$tag/blur_rs_pmc_$size.json)
done
head -c 1500 $R/gpurun_out/$tag/blur_rs_pmc_256.json
