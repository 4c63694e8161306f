# A/B of runtime switches on the headline bench line (no code change): kernel arguments in device memory, SDMA on / off
for cfg in "" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_SDMA=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_SDMA=0" ""; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 100 --no-other-configs --no-aten-baseline --no-cpu-baseline --no-mode-matrix 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(d[k],4) for k in ('value','ms_per_step','host_enqueue_ms_per_step')})"
done
