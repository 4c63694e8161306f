mkdir -p gpurun_out
timeout 300 python scripts/host_profile.py cumtime > gpurun_out/r3_hostprof_cumtime.txt 2>&1
sed -n 1,75p gpurun_out/r3_hostprof_cumtime.txt
