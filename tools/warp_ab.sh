mkdir -p gpurun_out/r3b
for rep in 1 2; do for w in A B; do echo "slab $w (A = unidiv, B = IEEE division), one tile per workgroup" >> gpurun_out/r3b/zoom.txt; FSG_LIB=$GRAFT_REPO_ROOT/tools/ab/libfsg_slab_$w.so python tools/kernel_bench.py --m 171 --only zoom_normalise >> gpurun_out/r3b/zoom.txt 2>&1; done; echo "rows" >> gpurun_out/r3b/zoom.txt; python tools/kernel_bench.py --tune 256 --m 171 --only zoom_normalise >> gpurun_out/r3b/zoom.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r3b/zoom.txt
