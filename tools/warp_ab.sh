mkdir -p gpurun_out/r4n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4n/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4n/tests.log
bash tools/prof_bench.sh r4n > gpurun_out/r4n/prof.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4n/bench_driver.json 2> gpurun_out/r4n/bench_driver.err; echo "driver rc=$?"
FSG_BENCH_SHARE_GPU0=1 python bench.py --gpus 2 --steps 100 --warmup 10 --stream-volumes 400 > gpurun_out/r4n/bench_2rank.json 2> gpurun_out/r4n/bench_2rank.err; echo "2rank rc=$?"
python tools/host_phases.py > gpurun_out/r4n/host_phases.txt 2>&1
python - <<'PY'
import json
for f in ["bench_driver","bench","bench_2rank"]:
    d=json.loads([l for l in open(f"gpurun_out/r4n/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline_step"]["frac"], d["n_gpus"])
    print("   c3", d["config3"]["wall_ms"], d["config3"]["checksum"], "c5", {k:v["volumes_per_s"] for k,v in d["config5"].items() if isinstance(v, dict)})
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d.get("gpu_over_cpu"), d.get("gpu_over_reference_anchor"))
PY
head -14 gpurun_out/r4n/kernel_stats.csv | cut -c1-170
