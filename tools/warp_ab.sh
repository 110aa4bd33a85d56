mkdir -p gpurun_out/r2x
for mode in default 77 78 79 80; do echo "mm_end $mode (slab, 4096 workgroups)" >> gpurun_out/r2x/zoom_end.txt; if [ $mode = default ]; then FSG_ZOOM_MM_BLOCKS=65536 python tools/kernel_bench.py --tune 8192 --m 171 --only zoom_minmax >> gpurun_out/r2x/zoom_end.txt 2>&1; else FSG_DIAG_MM_END=$mode FSG_ZOOM_MM_BLOCKS=65536 python tools/kernel_bench.py --tune 8192 --m 171 --only zoom_minmax >> gpurun_out/r2x/zoom_end.txt 2>&1; fi; done
grep -v amdgpu.ids gpurun_out/r2x/zoom_end.txt
