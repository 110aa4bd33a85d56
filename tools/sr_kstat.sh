# kernel statistics of SimulateMotion at 384^3 (tools/sr_stage_bench.py, 3 repetitions + 1 first call): top 25 kernels by total time
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/psr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psr -- python3 $GRAFT_REPO_ROOT/tools/sr_stage_bench.py --stages simulate_motion --reps 3 > $GRAFT_REPO_ROOT/$out/sr_stage.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/psr/*/*kernel_stats.csv | head -1); cp $f $out/sr_kernel_stats.csv
python3 - $out/sr_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 4 calls:", round(tot / 1e6, 1))
for r in rows[:25]:
    print(r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), "ms", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
cat $out/sr_stage.json | cut -c1-400
