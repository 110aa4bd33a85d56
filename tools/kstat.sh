# average duration of the kernels whose name contains $1 (default: all top 12) in a short profiled bench run
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-microbench --no-sr --no-config3 --no-config5 > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n = r["Name"]
    if sys.argv[1] and sys.argv[1] not in n: continue
    print(n.split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
