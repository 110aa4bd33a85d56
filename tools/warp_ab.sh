mkdir -p gpurun_out/r2p
python -m pytest tests/test_hip_parity.py -x -q > gpurun_out/r2p/pytest.log 2>&1; tail -3 gpurun_out/r2p/pytest.log
for m in 171 100 256; do echo "m $m" >> gpurun_out/r2p/zoom.txt; python tools/kernel_bench.py --m $m --only zoom >> gpurun_out/r2p/zoom.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r2p/zoom.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-sr --no-config3 --no-config5 --no-microbench > gpurun_out/r2p/bench.json 2>&1; cut -c1-400 gpurun_out/r2p/bench.json
