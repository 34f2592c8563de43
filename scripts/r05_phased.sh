# round 5: the lock-step (phased) sweep against the free-running one, at 256 / 128 / 64 workgroups
mkdir -p gpurun_out/r05
for g in 256; do CSB_GRID=$g timeout 400 scripts/_bin/csb_break 3 d; done > gpurun_out/r05/csb_break_d.txt 2>&1
cat gpurun_out/r05/csb_break_d.txt
