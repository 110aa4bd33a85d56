mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; tail -3 gpurun_out/r2c/pytest.log
for t in 0 4096; do for rot in 12 20 0; do echo "tune $t rot $rot" >> gpurun_out/r2c/warp_lean.txt; python tools/kernel_bench.py --only warp_f32_f32_epi_ws,warp_f32_u8_epi_ws,warp_f32_u8tof32,warp_lin_only_ws,warp_nn_f32_only_ws --tune $t --rot $rot >> gpurun_out/r2c/warp_lean.txt 2>&1; done; done
grep -v amdgpu.ids gpurun_out/r2c/warp_lean.txt
