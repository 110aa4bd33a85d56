# FETCH_SIZE / WRITE_SIZE of the blur kernels (separate counters-only passes), 256^3 and 384^3 -> gpurun_out/<tag>/blur_pmc_<size>.json
tag=${1:-blurpmc}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for size in 256 384; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/$tag/pmc_${size}_$ctr -- python3 $R/tools/kernel_bench.py --only blur --reps 5 --size $size > $R/gpurun_out/$tag/pmc_${size}_$ctr.log 2>&1 || echo "failed $size $ctr"
  done
  mkdir -p $R/gpurun_out/$tag/s$size && rm -rf $R/gpurun_out/$tag/s$size/* && cp -r $R/gpurun_out/$tag/pmc_${size}_FETCH_SIZE $R/gpurun_out/$tag/s$size/pmc_a && cp -r $R/gpurun_out/$tag/pmc_${size}_WRITE_SIZE $R/gpurun_out/$tag/s$size/pmc_b
  (cd $R && python tools/pmc_summary.py gpurun_out/$tag/s$size blur_ > gpurun_out/$tag/blur_pmc_$size.json)
done
head -c 1500 $R/gpurun_out/$tag/blur_pmc_256.json
