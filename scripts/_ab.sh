timeout 600 python -m pytest tests/test_gpu_bspline.py -x -q 2>&1 | tail -12
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
(cd tests/native/_build && timeout 300 ./resample_bench --cases parity 2>&1 | grep -v "^dtype" | grep -v "mismatch vs gather: 0 .*vs oracle: 0" | tail -3)
Q="--no-cpu-baseline --no-aten-baseline --no-mode-matrix"
j() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(sys.argv[1], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'launch_ms', round(d['roofline']['launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'host', round(d['host_enqueue_ms_per_step'],3))" "$1"; }
python bench.py $Q 2>/dev/null | j "fast  "
python bench.py $Q --resample-precision exact 2>/dev/null | j "exact "
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from torchio_amd import ops
e = ops.engine()
x = torch.rand(2, 1, 256, 256, 256, device='cuda')
for order in (2, 3):
    e.bspline_prefilter(x, order); torch.cuda.synchronize()
    t = time.perf_counter(); c = e.bspline_prefilter(x, order); torch.cuda.synchronize(); print('prefilter order', order, '2x256^3 ms', round(1e3 * (time.perf_counter() - t), 3))
    m = torch.eye(3, 4, device='cuda')[None].repeat(2, 1, 1); m[:, 0, 1] = 0.1; m[:, :, 3] = 1.5
    kw = dict(out_shape=(256, 256, 256), mapping=m, control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=['cubic' if order == 3 else 'quadratic'], fills=[None])
    e.resample3d([c], **kw); torch.cuda.synchronize()
    t = time.perf_counter(); e.resample3d([c], **kw); torch.cuda.synchronize(); print('  sampling ms', round(1e3 * (time.perf_counter() - t), 3))
PY
