mkdir -p gpurun_out
timeout 300 python scripts/host_profile.py tottime > gpurun_out/r3_hostprof_tottime.txt 2>&1
timeout 300 python scripts/host_profile.py cumtime > gpurun_out/r3_hostprof_cumtime.txt 2>&1
head -60 gpurun_out/r3_hostprof_tottime.txt
