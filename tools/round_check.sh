# End-of-round check on the GPU box: full parity suite, rocprofv3 kernel stats of the bench step, the bench line in its three
# forms (default, as the driver invokes it, 2 self-launched ranks sharing the one GPU), host phases.  Usage (from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/round_check.sh <tag>'      -> gpurun_out/<tag>/
tag=${1:-check}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
bash tools/prof_bench.sh $tag > $out/prof.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "driver rc=$?"
FSG_BENCH_SHARE_GPU0=1 python bench.py --gpus 2 --steps 100 --warmup 10 --stream-volumes 400 > $out/bench_2rank.json 2> $out/bench_2rank.err; echo "2rank rc=$?"
python tools/host_phases.py > $out/host_phases.txt 2>&1
python - "$out" <<'PY'
import json, sys
out = sys.argv[1]
for f in ["bench_driver", "bench", "bench_2rank"]:
    d = json.loads([l for l in open(f"{out}/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_launch"], d["roofline_step"]["frac"], d["n_gpus"])
    print("   c3", d["config3"]["wall_ms"], d["config3"]["checksum"], "c5", {k: v["volumes_per_s"] for k, v in d["config5"].items() if isinstance(v, dict)})
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d.get("gpu_over_cpu"), d.get("gpu_over_reference_anchor"))
PY
head -14 $out/kernel_stats.csv | cut -c1-170
