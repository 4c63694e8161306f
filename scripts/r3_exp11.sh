python - <<'PY'
import sys,time; sys.path.insert(0,'.')
import ctypes as C, torch
from torchio_amd import _abi,_lib,ops
f=_lib.load()[1]
n=8*256**3
out=torch.empty(n,dtype=torch.float32,pin_memory=True)
for th in (2,4,8,16,32):
    for rep in range(2):
        st=(C.c_uint64*(_abi.HOST_MT_STATE_BYTES//8))(); f["host_mt19937_seed"](C.addressof(st),7)
        t=time.perf_counter(); f["host_mt19937_randn"](C.addressof(st),C.c_void_p(out.data_ptr()),n,th); e=time.perf_counter()-t
    print(f"threads {th}: {e*1e3:.1f} ms for 8 x 256^3 draws")
for _ in range(4):
    s=ops.HostNormalStream(5); torch.cuda.synchronize(); t=time.perf_counter(); x=s.randn((8,1,256,256,256),'cuda'); torch.cuda.synchronize(); print("stream.randn to device ms", (time.perf_counter()-t)*1e3)
y=torch.randn(n,generator=torch.Generator().manual_seed(5))
print("equal:", torch.equal(x.flatten().cpu(), y))
PY
timeout 400 python bench.py --steps 30 --no-cpu-baseline --no-aten-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v['volumes_per_s'],1), v['ms_per_step']) for k,v in d['mode_matrix'].items()})"
