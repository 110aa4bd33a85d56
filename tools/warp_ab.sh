mkdir -p gpurun_out/r4h
python -m pytest tests -m gpu -x -q > gpurun_out/r4h/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4h/tests.log
python tools/kernel_bench.py --only "zoom" --reps 50 --m 171 2>/dev/null > gpurun_out/r4h/zoom.txt; cat gpurun_out/r4h/zoom.txt
python bench.py --no-cpu-baseline --no-sr > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; echo "bench rc=$?"
FSG_NO_MM_SLOTS=1 python bench.py --no-cpu-baseline --no-sr --no-config3 --no-config5 > gpurun_out/r4h/bench_noslots.json 2> gpurun_out/r4h/bench_noslots.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench","bench_noslots"):
    d=json.loads([l for l in open(f"gpurun_out/r4h/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline_step"]["frac"])
    if "config3" in d: print(d["config3"]["wall_ms"], d["config3"]["checksum"])
PY
