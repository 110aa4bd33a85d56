mkdir -p gpurun_out/r4o
for r in 6 12 16 20; do for v in 0 10 11 12; do echo "rot $r variant $v"; python tools/kernel_bench.py --only warp --reps 40 --variant $v --rot $r 2>/dev/null | grep "u8tof32_epi_ws"; done; done > gpurun_out/r4o/shape.txt 2>&1
cat gpurun_out/r4o/shape.txt
