import sys, torch
sys.path.insert(0, ".")
from torchio_amd import ops
from oracle.oracle import oracle_engine
sys.path.insert(0, "tests")
from test_gpu_ops_parity import _mapping, _control_points, _data
E, O = ops.engine(), oracle_engine()
for elastic, af, interp in [(True, True, "linear"), (True, False, "linear"), (True, True, "nearest")]:
    batch, channels = 2, 2
    in_shape, out_shape = (20, 24, 70), (22, 19, 67)
    data = _data((batch, channels, *in_shape), torch.float32, 1)
    kw = dict(out_shape=out_shape, in_spacing=(1.0, 1.25, 0.8), out_spacing=(0.9, 1.1, 0.75), affine_first=af, interps=[interp])
    m, cp = _mapping(1, 2), _control_points(1, (7, 6, 5), 3)
    c = O.resample3d([data], mapping=m, control_points=cp, fills=[None], **kw)[0]
    g = E.resample3d([data.cuda()], mapping=m.cuda(), control_points=cp.cuda(), fills=[None], **kw)[0].cpu()
    d = (c - g).abs()
    print(elastic, af, interp, "mismatch", int((c != g).sum()), "of", c.numel(), "max", float(d.max()))
    idx = (c != g).nonzero()[:5]
    print(idx.tolist())
# isolate: zero mapping rotation, only elastic
ident = torch.eye(3, 4).reshape(1, 3, 4)
for spacing in [(1.0, 1.0, 1.0), (1.0, 1.25, 0.8)]:
    data = _data((1, 1, 20, 24, 70), torch.float32, 1)
    kw = dict(out_shape=(20, 24, 70), in_spacing=spacing, out_spacing=spacing, affine_first=True, interps=["linear"])
    cp = _control_points(1, (7, 6, 5), 3)
    c = O.resample3d([data], mapping=ident, control_points=cp, fills=[None], **kw)[0]
    g = E.resample3d([data.cuda()], mapping=ident.cuda(), control_points=cp.cuda(), fills=[None], **kw)[0].cpu()
    print("ident", spacing, int((c != g).sum()), float((c - g).abs().max()))
