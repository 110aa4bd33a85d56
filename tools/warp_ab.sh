mkdir -p gpurun_out/r4k
timeout -k 10 300 python tools/overlap_replay.py > gpurun_out/r4k/replay.txt 2>&1; tail -5 gpurun_out/r4k/replay.txt
