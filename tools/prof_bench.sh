# rocprofv3 kernel trace + stats of the bench step (no side sections), summary copied for profiles/
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-microbench --no-sr --no-config3 --no-config5 > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -20 $out/kernel_stats.csv
python bench.py --steps 300 --warmup 20 > $out/bench.json 2> $out/bench.err; cut -c1-300 $out/bench.json
