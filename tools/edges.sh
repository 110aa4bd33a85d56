# is the driver's 20-step figure a matter of WHICH 20 samples?  the same form over different sample windows (warm-up W = first sample index)
for w in 5 25 45 65 105 205; do echo "warmup $w"; FSG_BENCH_EDGES=1 python bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-microbench --no-sr --no-config3 --no-config5 2>&1 >/dev/null | grep edges; done
