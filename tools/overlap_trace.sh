# kernel trace of the bench step with the head overlap on: are the head and the previous sample's tail concurrent?
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-microbench --no-sr --no-config3 --no-config5 > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err
cd $GRAFT_REPO_ROOT
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
python - "$f" > $out/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
# last 40 kernels: name, queue, start (us rel), dur
t0 = int(rows[-60]["Start_Timestamp"])
for r in rows[-60:]:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    print(f'{name:40s} q={r.get("Queue_Id","?"):>3s} start={(int(r["Start_Timestamp"])-t0)/1e3:9.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}')
PY
cat $out/timeline.txt
