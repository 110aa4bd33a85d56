mkdir -p gpurun_out/r4f
for s in 1 2 3; do python bench.py --no-cpu-baseline --no-sr --no-config3 --no-config5 --no-microbench --streams $s > gpurun_out/r4f/bench_s$s.json 2> gpurun_out/r4f/bench_s$s.err; echo "streams $s rc=$?"; done
python - <<'PY'
import json
for s in (1,2,3):
    d=json.loads([l for l in open(f"gpurun_out/r4f/bench_s{s}.json") if l.startswith("{")][-1])
    print(s, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline_step"]["frac"])
PY
