"""GPU: BiasField / Blur / Noise folded into the stencil passes (data/_pending.py, tio_blur_fused).

The fused launch must give the same bits as the three separate launches
(``TIO_NO_LAZY_FUSION=1``) and as the CPU oracle up to the transcendental tolerance of the
bias field and the Box-Muller draw; deferring must be invisible to readers of ``.data``.
"""
from __future__ import annotations

import copy
import os

import pytest
import torch

import torchio_amd as tio
from parity_harness import make_subjects
from parity_harness import use_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _philox_noise():
    """These tests draw the noise on the device; the process-wide mode is restored afterwards."""
    previous = tio.get_noise_rng()
    tio.set_noise_rng("philox")
    yield
    tio.set_noise_rng(previous)


class _FusedCalls:
    """Counts what reaches ``Engine.blur_fused`` (the tests must SEE the fused branch taken)."""

    def __init__(self, engine):
        self.engine, self.calls, self.with_noise, self.with_bias = engine, 0, 0, 0
        self._original = engine.blur_fused

    def __enter__(self):
        def counted(data, taps, radius, *, bias_coarse=None, noise=None):
            out = self._original(data, taps, radius, bias_coarse=bias_coarse, noise=noise)
            if out is not None:
                self.calls += 1
                self.with_noise += int(noise is not None)
                self.with_bias += int(bias_coarse is not None)
            return out

        self.engine.blur_fused = counted
        return self

    def __exit__(self, *exc):
        del self.engine.blur_fused  # back to the class attribute


def _run(transform, batch, seed, *, lazy: bool):
    previous = os.environ.get("TIO_NO_LAZY_FUSION")
    os.environ["TIO_NO_LAZY_FUSION"] = "0" if lazy else "1"
    try:
        torch.manual_seed(seed)
        out = transform(batch)
        data = out.t1.data  # reading flushes anything still queued
        torch.cuda.synchronize()
        return out, data
    finally:
        if previous is None:
            os.environ.pop("TIO_NO_LAZY_FUSION", None)
        else:
            os.environ["TIO_NO_LAZY_FUSION"] = previous


CHAINS = {
    "bias_blur_noise": lambda: [tio.BiasField(), tio.Blur(std=(0.5, 2)), tio.Noise()],
    "blur_noise": lambda: [tio.Blur(std=(0.5, 2)), tio.Noise()],
    "bias_blur": lambda: [tio.BiasField(), tio.Blur(std=(0.5, 2))],
    "bias_blur_gamma_noise": lambda: [tio.BiasField(), tio.Blur(std=(0.5, 2)), tio.Gamma(log_gamma=(-0.3, 0.3)), tio.Noise()],
    "blur_blur_noise": lambda: [tio.Blur(std=(0.5, 1.0)), tio.Blur(std=(1.0, 2.0)), tio.Noise()],
    "single_axis_blur_noise": lambda: [tio.BiasField(), tio.Blur(std=(0.0, 0.0, 0.0, 0.0, 1.0, 1.5)), tio.Noise()],
}


@pytest.mark.parametrize("name", list(CHAINS))
@pytest.mark.parametrize("size,batch", [(48, 3), (20, 1)])
def test_lazy_fusion_is_bit_identical_to_separate_launches(hip, name, size, batch):
    subjects = make_subjects(size, batch, seed=11, with_label=True)
    gpu_batch = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    transform = tio.Compose(CHAINS[name]())
    assert tio.get_noise_rng() == "philox"
    with _FusedCalls(hip) as seen:
        fused, fused_data = _run(transform, gpu_batch, 5, lazy=True)
    # the fused launch really ran: one per float image, carrying the noise / the bias field whenever the chain has them
    # adjacent to the LAST blur (a Gamma between Blur and Noise keeps the noise out of the stencil's stores)
    # (a stencil that is not active on all three axes has no fused form: tio_blur_fused answers UNSUPPORTED_CONFIG and the
    # three launches run one after the other)
    if name != "single_axis_blur_noise":
        assert seen.calls >= 1, "the deferred Blur never reached tio_blur_fused"
    if name in ("bias_blur_noise", "blur_noise", "blur_blur_noise"):
        assert seen.with_noise >= 1, "Noise did not ride on the stencil's stores (conv_march_kernel<..., POST_NOISE>)"
    if name in ("bias_blur_noise", "bias_blur", "bias_blur_gamma_noise"):
        assert seen.with_bias >= 1, "BiasField did not ride on the stencil's loads"
    with _FusedCalls(hip) as unfused:
        plain, plain_data = _run(transform, gpu_batch, 5, lazy=False)
    assert unfused.calls == 0
    assert [t.params for t in fused.applied_transforms] == [t.params for t in plain.applied_transforms]
    assert torch.equal(fused_data, plain_data)
    assert torch.equal(fused.seg.data, plain.seg.data)
    assert fused_data.data_ptr() != gpu_batch.t1.data.data_ptr()  # never aliases the caller's tensor


def test_lazy_fusion_matches_oracle_and_is_invisible(oracle, hip):
    subjects = make_subjects(32, 2, seed=13)
    transform = tio.Compose([tio.BiasField(), tio.Blur(std=(0.5, 2)), tio.Noise()])
    cpu_batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
    gpu_batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)).to("cuda")
    torch.manual_seed(3)
    with use_engine(oracle):
        expected = transform(cpu_batch)
    torch.manual_seed(3)
    with _FusedCalls(hip) as seen:
        actual = transform(gpu_batch)
        actual.t1.data
    assert seen.with_noise == 1 and seen.with_bias == 1  # ONE fused launch carried all three transforms
    torch.testing.assert_close(actual.t1.data.cpu(), expected.t1.data, rtol=1e-5, atol=2e-5)
    # copy=True (the default) returns finished tensors; with copy=False the work stays queued
    # on the batch until somebody looks at .data
    torch.manual_seed(4)
    finished = tio.Blur(std=(1.0, 1.0))(gpu_batch)
    assert finished.t1._pending is None
    working = copy.deepcopy(gpu_batch)
    before = working.t1.data.clone()
    torch.manual_seed(4)
    queued = tio.Blur(std=(1.0, 1.0), copy=False)(working)
    assert queued.t1._pending is not None and queued.t1._pending.blur is not None
    values = queued.t1.data
    assert queued.t1._pending is None and torch.equal(values, finished.t1.data)
    assert not torch.equal(values, before)


def test_fast_stencil_stays_within_float_rounding_of_the_exact_one(hip):
    """``tio.set_stencil_precision("fast")``: fused multiply-adds in the taps of the fused Blur launch — same taps, same order,
    one rounding per tap instead of two."""
    subjects = make_subjects(48, 3, seed=17, with_label=False)
    gpu_batch = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    transform = tio.Compose([tio.BiasField(), tio.Blur(std=(0.5, 2)), tio.Noise()])
    previous = tio.get_stencil_precision()
    try:
        tio.set_stencil_precision("exact")
        _, exact = _run(transform, gpu_batch, 9, lazy=True)
        tio.set_stencil_precision("fast")
        with _FusedCalls(hip) as seen:
            _, fast = _run(transform, gpu_batch, 9, lazy=True)
    finally:
        tio.set_stencil_precision(previous)
    assert seen.with_noise == 1
    assert not torch.equal(exact, fast)  # it really is another arithmetic ...
    scale = float(exact.abs().max())
    assert float((exact - fast).abs().max()) <= 2e-6 * scale  # ... within a few float32 roundings


def test_host_resident_subjects_are_staged_through_the_device(hip, oracle):
    """``tio.Compose(...)(cpu_subject)`` — the reference's everyday call (transform.py:212-254) — works on the HIP engine:
    the data takes ONE trip through the device and comes back as host tensors with the oracle's values."""
    import copy

    subjects = make_subjects(24, 2, seed=19)
    transform = tio.Compose([tio.Affine(degrees=(-8, 8), scales=(0.95, 1.05)), tio.Blur(std=(0.5, 1.0)), tio.Gamma(log_gamma=(-0.2, 0.2))])
    previous = tio.get_noise_rng()
    tio.set_noise_rng("reference")
    try:
        torch.manual_seed(21)
        with use_engine(oracle):
            expected = transform(tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects)))
        host_batch = tio.SubjectsBatch.from_subjects(copy.deepcopy(subjects))
        before = host_batch.t1.data.clone()
        torch.manual_seed(21)
        actual = transform(host_batch)
        assert actual.t1.data.device.type == "cpu" and actual.seg.data.device.type == "cpu"
        assert host_batch.t1.data.device.type == "cpu" and torch.equal(host_batch.t1.data, before)  # the caller's batch: untouched, at home
        assert torch.equal(actual.seg.data, expected.seg.data)
        torch.testing.assert_close(actual.t1.data, expected.t1.data, rtol=1e-5, atol=2e-6)
        # a single Subject goes the same way
        torch.manual_seed(22)
        with use_engine(oracle):
            want = tio.Affine(degrees=(-8, 8))(copy.deepcopy(subjects[0]))
        torch.manual_seed(22)
        got = tio.Affine(degrees=(-8, 8))(copy.deepcopy(subjects[0]))
        assert got.t1.data.device.type == "cpu" and torch.equal(got.seg.data, want.seg.data)
    finally:
        tio.set_noise_rng(previous)


# ---- the REFERENCE's noise stream on the stencil's stores (round 4: tio_blur_fused(noise_on = 2), HostNormalStream.randn_ahead) ----
def _reference_noise_expected(transform_without_noise, gpu_batch, seed_of_step, history, names):
    """blurred + (mean + std * torch.randn(shape, generator=Generator().manual_seed(seed))) — ONE generator for all images of
    the call, in dict order (noise.py:108-116) — computed with torch on the host from the recorded parameters."""
    torch.manual_seed(seed_of_step)
    blurred = transform_without_noise(gpu_batch)
    params = history[-1].params
    generator = torch.Generator().manual_seed(params["seed"])
    expected = {}
    for name in names:
        x = getattr(blurred, name).data.cpu()
        z = torch.randn(x.shape, generator=generator)
        mean = torch.tensor(params["mean"], dtype=torch.float32).reshape(-1, 1, 1, 1, 1) if isinstance(params["mean"], list) else params["mean"]
        std = torch.tensor(params["std"], dtype=torch.float32).reshape(-1, 1, 1, 1, 1) if isinstance(params["std"], list) else params["std"]
        expected[name] = x + (mean + std * z)
    return expected


@pytest.mark.parametrize("size,batch,per_instance,second", [(72, 3, True, False), (112, 1, False, False), (64, 5, True, True)])
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_reference_noise_rides_on_the_stencil_stores(hip, size, batch, per_instance, second, precision):
    """Noise mode "reference": the draws of the seeded CPU generator are made ahead on the draw stream and added by the fused
    J + K pass.  Same bits as the separate launches (which are pinned to torch.randn by tests/test_gpu_device_rng.py) and as
    torch's own stream applied on the host; two float images share ONE generator in dict order."""
    subjects = make_subjects(size, batch, seed=23, with_label=False, second_modality=second)
    gpu_batch = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
    names = ["t1", "t2"] if second else ["t1"]
    head = lambda: [tio.BiasField(per_instance=per_instance), tio.Blur(std=(0.5, 2), per_instance=per_instance)]  # noqa: E731
    transform = tio.Compose([*head(), tio.Noise(per_instance=per_instance)])
    previous, previous_stencil, previous_policy = tio.get_noise_rng(), tio.get_stencil_precision(), tio.get_draw_policy()
    tio.set_noise_rng("reference")
    tio.set_stencil_precision(precision)
    tio.set_draw_policy("free")  # (round 5: the draw stream is a POLICY — default "off"; this test is about the road with it)
    try:
        seen_draws = []
        original = hip.blur_fused

        def spying(data, taps, radius, *, bias_coarse=None, noise=None):
            out = original(data, taps, radius, bias_coarse=bias_coarse, noise=noise)
            if out is not None and noise is not None:
                seen_draws.append(isinstance(noise[2], torch.Tensor))
            return out

        hip.blur_fused = spying
        try:
            fused, _ = _run(transform, gpu_batch, 5, lazy=True)
            fused_data = {name: getattr(fused, name).data.clone() for name in names}
        finally:
            del hip.blur_fused
        assert seen_draws == [True] * len(names), seen_draws  # every float image: explicit draws on the fused launch
        plain, _ = _run(transform, gpu_batch, 5, lazy=False)
        assert [t.params for t in fused.applied_transforms] == [t.params for t in plain.applied_transforms]
        if precision == "exact":  # (the fast taps only exist in the fused launch)
            for name in names:
                assert torch.equal(fused_data[name], getattr(plain, name).data), name
        expected = _reference_noise_expected(tio.Compose(head()), gpu_batch, 5, fused.applied_transforms, names)
        for name in names:
            if precision == "exact":
                assert torch.equal(fused_data[name].cpu(), expected[name]), name
            else:
                torch.testing.assert_close(fused_data[name].cpu(), expected[name], rtol=0, atol=4e-6)
        # the generator of the global stream ends where the reference's would: the next draw is the same either way
        torch.manual_seed(5)
        transform(gpu_batch)
        probe_a = torch.rand(3)
        torch.manual_seed(5)
        os.environ["TIO_NO_DRAW_STREAM"] = "1"
        try:
            again = transform(gpu_batch)
            again_data = {name: getattr(again, name).data for name in names}
        finally:
            os.environ.pop("TIO_NO_DRAW_STREAM", None)
        probe_b = torch.rand(3)
        assert torch.equal(probe_a, probe_b)
        for name in names:  # the switch takes the previous road (draws + sum in one kernel after the stencil): same values
            if precision == "exact":
                assert torch.equal(again_data[name], fused_data[name])
    finally:
        tio.set_noise_rng(previous)
        tio.set_stencil_precision(previous_stencil)
        tio.set_draw_policy(previous_policy)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_ring_kernels_give_the_marching_kernels_fused_values(hip, monkeypatch, precision):
    """`TIO_CONV_RING=1` sends the I and J passes to the LDS-ring kernels (the road of radii the register window does not hold):
    bias on the loads, the fused K stage with its preloaded taps, Philox noise and explicit draws on the stores — the same values
    as the marching kernels, bit for bit in the exact mode and in the fast mode's fused multiply-adds alike... the ring kernels
    have no fast taps, so the fast launch is compared within float rounding."""
    import numpy as np

    from torchio_amd.transforms.blur import _stacked_gaussian_taps

    data = torch.rand(2, 1, 40, 36, 64, generator=torch.Generator().manual_seed(31)).cuda()
    taps, radius, _ = _stacked_gaussian_taps(np.array([[1.6, 1.1, 1.9], [1.2, 1.9, 0.8]]), per_element=True)
    taps = taps.cuda()
    coarse = (0.3 * torch.randn(2, 1, 4, 4, 4, generator=torch.Generator().manual_seed(32))).cuda()
    mean = torch.zeros(2, device="cuda")
    std = torch.full((2,), 0.25, device="cuda")
    draws = torch.randn(data.shape, generator=torch.Generator().manual_seed(33)).cuda()
    previous = tio.get_stencil_precision()
    tio.set_stencil_precision(precision)
    try:
        results = {}
        for ring in ("", "1"):
            if ring:
                monkeypatch.setenv("TIO_CONV_RING", ring)
            else:
                monkeypatch.delenv("TIO_CONV_RING", raising=False)
            results[ring] = [
                hip.blur_fused(data, taps, radius, bias_coarse=coarse),
                hip.blur_fused(data, taps, radius, bias_coarse=coarse, noise=(mean, std, 4321)),
                hip.blur_fused(data, taps, radius, bias_coarse=coarse, noise=(mean, std, draws)),
            ]
            torch.cuda.synchronize()
        monkeypatch.delenv("TIO_CONV_RING", raising=False)
    finally:
        tio.set_stencil_precision(previous)
    for marching, ringed in zip(results[""], results["1"], strict=True):
        assert marching is not None and ringed is not None
        if precision == "exact":
            assert torch.equal(marching, ringed)
        else:
            assert float((marching - ringed).abs().max()) <= 2e-6 * float(marching.abs().max())


def test_reference_noise_on_the_draw_stream_from_several_threads(hip):
    """`Queue`'s workers call transforms from a thread pool (queue.py:119-123): the draw stream, its back-pressure marks and the
    pinned staging rings are shared or per-thread state — three threads, each applying Blur + Noise (recorded parameters: the
    result is a function of them alone) to its own batches, get the values the same calls give one after the other."""
    import copy
    from concurrent.futures import ThreadPoolExecutor

    previous, previous_policy = tio.get_noise_rng(), tio.get_draw_policy()
    tio.set_noise_rng("reference")
    tio.set_draw_policy("free")
    try:
        subjects = make_subjects(72, 3, seed=41, with_label=False)
        base = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
        blur, noise = tio.Blur(std=(0.5, 2)), tio.Noise()
        jobs = []
        for seed in range(6):
            torch.manual_seed(100 + seed)
            jobs.append((blur.make_params(base), noise.make_params(base)))

        def run(job):
            batch = copy.deepcopy(base)
            batch = blur.apply_transform(batch, job[0])
            assert batch.t1._pending is not None  # (the Blur is queued: the Noise call below rides on its stores)
            batch = noise.apply_transform(batch, job[1])
            out = batch.t1.data
            torch.cuda.current_stream().synchronize()
            return out

        expected = [run(job) for job in jobs]
        with ThreadPoolExecutor(max_workers=3) as pool:
            for _ in range(2):
                got = list(pool.map(run, jobs))
                for want, have in zip(expected, got, strict=True):
                    assert torch.equal(want, have)
        assert not torch.equal(expected[0], expected[1])
    finally:
        tio.set_noise_rng(previous)
        tio.set_draw_policy(previous_policy)


@pytest.mark.parametrize("policy", ["off", "free", "gated"])
def test_two_noise_children_in_one_compose_keep_their_own_streams(hip, policy):
    """both Noise children of a Compose that draws ahead start the plan of their generator's stream BEFORE
    any child applies; the pinned staging buffer of the first plan had no recorded upload yet and was handed out again — two
    jobs wrote one buffer.  The result must be the sequential one, under every draw policy."""
    previous, previous_policy = tio.get_noise_rng(), tio.get_draw_policy()
    tio.set_noise_rng("reference")
    tio.set_draw_policy(policy)
    try:
        subjects = make_subjects(104, 1, seed=51, with_label=False)  # 1 124 864 voxels: the device draws the stream itself
        pipeline = tio.Compose([tio.Noise(std=(0.1, 0.2)), tio.Blur(std=(0.5, 1.5)), tio.Noise(std=(0.3, 0.4))])
        assert pipeline._may_draw_ahead()
        gpu = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
        torch.manual_seed(77)
        ahead = pipeline(gpu)
        history = ahead.applied_transforms
        seeds = [t.params["seed"] for t in history if t.name == "Noise"]
        assert len(seeds) == 2 and seeds[0] != seeds[1]
        # the same recorded parameters, child by child (no draw-ahead road)
        stepwise = tio.SubjectsBatch.from_subjects(subjects).to("cuda")
        for child, record in zip(pipeline.transforms, history, strict=True):
            stepwise = child.apply_transform(stepwise, record.params)
        torch.cuda.synchronize()
        assert torch.equal(ahead.t1.data, stepwise.t1.data)
        # and torch's own streams on the host for the first child
        x = tio.SubjectsBatch.from_subjects(subjects).t1.data
        z = torch.randn(x.shape, generator=torch.Generator().manual_seed(seeds[0]))
        first = pipeline.transforms[0].apply_transform(tio.SubjectsBatch.from_subjects(subjects).to("cuda"), history[0].params)
        std = history[0].params["std"]
        std = std[0] if isinstance(std, list) else std
        mean = history[0].params["mean"]
        mean = mean[0] if isinstance(mean, list) else mean
        torch.testing.assert_close(first.t1.data.cpu(), x + (mean + std * z), rtol=0, atol=0)
    finally:
        tio.set_noise_rng(previous)
        tio.set_draw_policy(previous_policy)
