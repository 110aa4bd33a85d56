mkdir -p gpurun_out/r2t
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2t/bench_driver.json 2> gpurun_out/r2t/bench_driver.err; echo "driver rc=$?"
python bench.py > gpurun_out/r2t/bench_default.json 2> gpurun_out/r2t/bench_default.err; echo "default rc=$?"
FSG_BENCH_SHARE_GPU0=1 python bench.py --gpus 2 --steps 100 --warmup 10 --stream-volumes 400 > gpurun_out/r2t/bench_2rank.json 2> gpurun_out/r2t/bench_2rank.err; echo "2rank rc=$?"
python - <<'PY'
import json
for f in ["bench_driver","bench_default","bench_2rank"]:
    try:
        d=json.loads([l for l in open(f"gpurun_out/r2t/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_step"]["frac"], d["n_gpus"])
        print("   c3", d["config3"]["wall_ms"], d["config3"]["volumes_per_s"], d["config3"]["checksum"])
        print("   c5", {k:v["volumes_per_s"] for k,v in d["config5"].items() if isinstance(v, dict)})
    except Exception as e:
        print(f, "ERR", e)
PY
