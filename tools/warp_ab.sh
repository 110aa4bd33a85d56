mkdir -p gpurun_out/r2r
FSG_TEST_REPORT=1 python -m pytest tests/test_sr_stages.py tests/test_sr_known_answers.py tests/test_sr_parity.py -x -q -s -m gpu > gpurun_out/r2r/pytest_sr.log 2>&1; tail -3 gpurun_out/r2r/pytest_sr.log; grep "\[report\]" gpurun_out/r2r/pytest_sr.log
