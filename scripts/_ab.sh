Q="--no-cpu-baseline --no-aten-baseline --no-mode-matrix"
j() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(sys.argv[1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4))" "$1"; }
for rep in 1 2; do
TIO_EXACT_PLAN=0 python bench.py $Q 2>/dev/null | j "exact unplanned"
python bench.py $Q 2>/dev/null | j "exact default  "
TIO_FAST_KERNEL=brick python bench.py $Q --resample-precision fast 2>/dev/null | j "fast brick     "
python bench.py $Q --resample-precision fast 2>/dev/null | j "fast planned   "
done
