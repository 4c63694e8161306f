mkdir -p gpurun_out
nproc
timeout 900 python -m pytest tests/test_host_rng.py tests/test_gpu_full_size.py tests/test_gpu_golden.py tests/test_gpu_ops_parity.py -q -x 2>&1 | tail -4
python - <<'PY'
import sys,time; sys.path.insert(0,'.')
import ctypes as C, torch
from torchio_amd import _abi,_lib,ops
f=_lib.load()[1]
n=8*256**3
out=torch.empty(n,dtype=torch.float32,pin_memory=True)
for th in (1,8,16,32,64):
    st=(C.c_uint64*(_abi.HOST_MT_STATE_BYTES//8))(); f["host_mt19937_seed"](C.addressof(st),7)
    t=time.perf_counter(); f["host_mt19937_randn"](C.addressof(st),C.c_void_p(out.data_ptr()),n,th); e=time.perf_counter()-t
    print(f"threads {th}: {e*1e3:.1f} ms for 8 x 256^3 draws")
for _ in range(3):
    s=ops.HostNormalStream(5); torch.cuda.synchronize(); t=time.perf_counter(); x=s.randn((8,1,256,256,256),'cuda'); torch.cuda.synchronize(); print("stream.randn to device ms", (time.perf_counter()-t)*1e3)
t=time.perf_counter(); y=torch.randn(n,generator=torch.Generator().manual_seed(5)); print("torch.randn ms",(time.perf_counter()-t)*1e3)
print("equal:", torch.equal(x.flatten().cpu(), y))
PY
timeout 400 python bench.py --steps 30 --no-cpu-baseline --no-aten-baseline > gpurun_out/r3_exp10_bench.json 2> gpurun_out/r3_exp10_bench.err; tail -2 gpurun_out/r3_exp10_bench.err | cut -c1-300; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_exp10_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')}, d['roofline']['frac'], d['roofline']['launch_ms'])
print({k:(round(v['volumes_per_s'],1), v.get('resample_launch_ms')) for k,v in d['mode_matrix'].items()})
for k,v in d['other_configs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
PY
