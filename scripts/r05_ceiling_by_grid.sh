mkdir -p gpurun_out/r05
for g in 256 128 64; do CSB_GRID=$g timeout 300 scripts/_bin/csb_ceiling 3; done > gpurun_out/r05/csb_ceiling_by_grid.txt 2>&1
cat gpurun_out/r05/csb_ceiling_by_grid.txt
