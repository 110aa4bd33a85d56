mkdir -p gpurun_out/r2s
python -m pytest tests -m gpu -x -q > gpurun_out/r2s/pytest.log 2>&1; tail -5 gpurun_out/r2s/pytest.log
