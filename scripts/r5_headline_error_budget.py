"""Where the headline mode's rounding comes from (round 5): the five-transform Compose at 3 x 256^3 against the oracle, per
voxel (|d| / max(|ref|, 1e-3 range)), with the stencil's taps exact / fused and with / without the Noise stage."""
import copy, json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torchio_amd as tio
from oracle.oracle import oracle_engine
from parity_harness import use_engine

size, batch = 256, 3
g = torch.Generator().manual_seed(31)
subjects = [tio.Subject(t1=tio.ScalarImage(torch.rand(1, size, size, size, generator=g))) for _ in range(batch)]
kw = dict(per_instance=True)
head = lambda: [tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), **kw), tio.ElasticDeformation(**kw), tio.BiasField(**kw), tio.Blur(std=(0.5, 2), **kw)]
tio.set_noise_rng("philox")
out = {}
for with_noise in (False, True):
    transform = tio.Compose(head() + ([tio.Noise(**kw)] if with_noise else []))
    cpu = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    torch.manual_seed(32)
    with use_engine(oracle_engine()):
        want = transform(cpu).t1.data.double()
    rng = float(want.max() - want.min())
    for resample in ("tight", "exact"):
        for stencil in ("fast", "exact"):
            tio.set_resample_precision(resample); tio.set_stencil_precision(stencil)
            gpu = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)).to("cuda")
            torch.manual_seed(32)
            got = transform(gpu).t1.data.cpu().double()
            d = (want - got).abs()
            rel = d / want.abs().clamp_min(1e-3 * rng)
            out[f"noise={with_noise},resample={resample},stencil={stencil}"] = dict(range=rng, max_abs_over_range=float(d.max()) / rng, per_voxel_max=float(rel.max()),
                                                                                     beyond=int((rel > 1e-4).sum()), beyond_half=int((rel > 5e-5).sum()))
            print(f"noise={with_noise},resample={resample},stencil={stencil}", out[f"noise={with_noise},resample={resample},stencil={stencil}"], flush=True)
json.dump(out, open("gpurun_out/r5_headline_error_budget.json", "w"), indent=1)
