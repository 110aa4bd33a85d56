mkdir -p gpurun_out/r3e
python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest.log 2>&1; tail -8 gpurun_out/r3e/pytest.log
python tools/host_phases.py > gpurun_out/r3e/host_phases.txt 2>&1; tail -4 gpurun_out/r3e/host_phases.txt
