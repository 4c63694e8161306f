"""CPU tests of the host-side mirror (sampling, gating, history, inverse, wrapping, errors).

Modelled on the reference's own behavioural tests (identity,
gated-out rows are bit-exact no-ops, seeded reproducibility, label values stay in
the input set, inverse restores geometry, dtype preservation).  The compute runs
on the CPU oracle through the test-only engine hook.
"""
from __future__ import annotations

import copy
import warnings

import numpy as np
import pytest
import torch

import torchio_amd as tio
from parity_harness import make_subjects
from parity_harness import use_engine
from torchio_amd.transforms.parameter_range import _ParameterRange
from torchio_amd.transforms.parameter_range import to_nonneg_range


@pytest.fixture(autouse=True)
def _oracle_engine(oracle):
    with use_engine(oracle), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def subject(size=12, seed=0):
    return make_subjects(size, 1, seed)[0]


def batch(n=4, size=10, seed=0):
    return tio.SubjectsBatch.from_subjects(make_subjects(size, n, seed))


# -- parameter ranges ---------------------------------------------------------------
def test_parameter_range_forms_and_draw_order():
    assert _ParameterRange(2.0).sample() == (2.0, 2.0, 2.0)
    assert _ParameterRange((1.0, 2.0, 3.0)).sample() == (1.0, 2.0, 3.0)
    torch.manual_seed(0)
    a = _ParameterRange((0.0, 1.0)).sample()
    torch.manual_seed(0)
    b = tuple(torch.empty(1).uniform_(0.0, 1.0).item() for _ in range(3))
    assert a == b  # one uniform_ draw per axis, in axis order
    torch.manual_seed(0)
    state = torch.get_rng_state()
    _ParameterRange((5.0, 5.0)).sample()  # degenerate range draws nothing
    assert torch.equal(state, torch.get_rng_state())
    six = _ParameterRange((0, 1, 2, 3, 4, 5))._ranges
    assert six == ((0.0, 1.0), (2.0, 3.0), (4.0, 5.0))
    assert _ParameterRange((0.0, 1.0)).sample(5).shape == (5, 3)
    assert _ParameterRange(tio.Choice([1, 2, 3])).sample_1d(7).shape == (7,)
    with pytest.raises(ValueError):
        to_nonneg_range((-1.0, 1.0))
    with pytest.raises(ValueError):
        _ParameterRange((1.0, 2.0, 3.0, 4.0))
    with pytest.raises(TypeError):
        _ParameterRange("x")


def test_constructor_validation_matches_reference_errors():
    with pytest.raises(ValueError, match="Probability"):
        tio.Affine(p=1.5)
    with pytest.raises(ValueError, match="strictly positive"):
        tio.Affine(scales=(0.0, 1.0))
    with pytest.raises(ValueError, match="isotropic"):
        tio.Affine(scales=(0.9, 1.1, 0.9, 1.1, 0.9, 1.1), isotropic=True)
    with pytest.raises(ValueError, match="not supported"):
        tio.Spatial(image_interpolation="bicubic")
    with pytest.raises(ValueError, match="image_interpolation cannot"):
        tio.Spatial(image_interpolation="label")
    with pytest.raises(ValueError, match="locked_borders"):
        tio.ElasticDeformation(locked_borders=3)
    with pytest.raises(ValueError, match="identity elastic field"):
        tio.ElasticDeformation(num_control_points=4)
    with pytest.raises(ValueError, match="greater than 3"):
        tio.ElasticDeformation(num_control_points=3)
    with pytest.raises(ValueError, match="scale must be"):
        tio.BiasField(scale=0.0)
    with pytest.raises(ValueError, match="non-negative"):
        tio.Noise(std=(-1.0, 1.0))
    with pytest.raises(ValueError, match="default_pad_value"):
        tio.Affine(default_pad_value="median")
    with pytest.warns(UserWarning, match="no-op"):
        tio.Affine()
    with pytest.warns(UserWarning, match="no-op"):
        tio.Gamma()


def test_every_interpolation_order_of_the_reference_is_taken():
    """Orders 0 - 7, as names and as integers (reference: spatial.py `_INTERPOLATION_ORDERS`): none falls back, none raises
    (orders >= 4 raised NotImplementedError until round 4)."""
    s = subject()
    for order, name in enumerate(["nearest", "linear", "quadratic", "cubic", "fourth", "fifth", "sixth", "seventh"]):
        torch.manual_seed(3)
        by_name = tio.Affine(degrees=(5, 5), image_interpolation=name)(s)
        torch.manual_seed(3)
        by_number = tio.Affine(degrees=(5, 5), image_interpolation=order)(s)
        assert torch.equal(by_name.t1.data, by_number.t1.data)
        assert by_name.t1.data.shape == s.t1.data.shape and torch.isfinite(by_name.t1.data).all()
    out = tio.Affine(degrees=(5, 5), label_interpolation="label", one_hot_label_interpolation="fifth")(s)
    assert set(out.seg.data.unique().tolist()) <= set(s.seg.data.unique().tolist()) | {0}


# -- "label" partial-volume mode ------------------------------------------------------
def test_label_mode_identity_pad_and_value_set():
    s = subject(size=12)
    labels = set(s.seg.data.unique().tolist())
    out = tio.Affine(degrees=(12, 12), translation=(2, 2), label_interpolation="label", default_pad_label=9)(s)
    assert out.seg.data.dtype == s.seg.data.dtype and out.seg.data.shape == s.seg.data.shape
    assert set(out.seg.data.unique().tolist()) <= labels | {9}
    # a transform with nothing to do is skipped by the envelope; a pure re-gridding onto the same grid is exact
    same = tio.Resample(target=1.0, label_interpolation="label")(s)
    assert torch.equal(same.seg.data, s.seg.data)
    # everything out of view -> the pad label everywhere
    gone = tio.Affine(translation=(500.0, 0.0, 0.0), label_interpolation="label", default_pad_label=4)(s)
    assert bool((gone.seg.data == 4).all())


def test_label_mode_fused_equals_materialised_one_hot():
    """The fused kernel against the reference's four steps with the one-hot channels built.

    ``antialias=True`` selects the materialised path; with an unchanged grid the smoothing has
    sigma 0 on every axis, so both paths compute the same pipeline.
    """
    for seed, pad in ((3, 0), (4, 2)):
        s = subject(size=14, seed=seed)
        kwargs = dict(degrees=(-20, 20), scales=(0.8, 1.2), translation=(-3, 3), max_displacement=3.0,
                      label_interpolation="label", default_pad_label=pad)
        torch.manual_seed(seed)
        fused = tio.Spatial(**kwargs)(s)
        torch.manual_seed(seed)
        materialised = tio.Spatial(antialias=True, **kwargs)(s)
        assert torch.equal(fused.seg.data, materialised.seg.data)
        assert torch.equal(fused.t1.data, materialised.t1.data)


def test_cascade_sum_restatements_match_torch_sum(oracle):
    """ATen's channel sum (multi_row_sum cascade) as restated in the oracle and on the host."""
    import ctypes

    from oracle.oracle import LIBRARY_PATH
    from torchio_amd.transforms.spatial import _cascade_sum_channels

    lib = ctypes.CDLL(LIBRARY_PATH)
    lib.tio_oracle_cascade_sum.restype = ctypes.c_float
    lib.tio_oracle_cascade_sum.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    generator = torch.Generator().manual_seed(5)
    for channels in (1, 2, 5, 15, 16, 17, 33, 100, 256, 257, 300):
        # inner extent a multiple of 32 per thread chunk: ATen's vectorised main path (its remainder columns
        # use a different, thread-partition dependent order that nothing here reproduces)
        x = torch.rand(2, channels, 4, 8, 256, generator=generator) * (torch.rand(2, channels, 4, 8, 256, generator=generator) < 0.2)
        expected = x.sum(dim=1)
        assert torch.equal(_cascade_sum_channels(x), expected), channels
        column = x[1, :, 2, 3, 77].contiguous()
        assert lib.tio_oracle_cascade_sum(column.data_ptr(), channels) == expected[1, 2, 3, 77].item()


# -- envelope: copy, wrapping, scope ------------------------------------------------
def test_copy_semantics_and_output_types():
    s = subject()
    before = s.t1.data.clone()
    out = tio.Noise(std=0.5)(s)
    assert torch.equal(s.t1.data, before) and not torch.equal(out.t1.data, before)
    assert isinstance(out, tio.Subject) and [t.name for t in out.applied_transforms] == ["Noise"]
    tensor = torch.rand(1, 8, 8, 8)
    assert isinstance(tio.Gamma(log_gamma=0.3)(tensor), torch.Tensor)
    array = np.random.rand(1, 8, 8, 8).astype(np.float32)
    assert isinstance(tio.Gamma(log_gamma=0.3)(array), np.ndarray)
    image = tio.ScalarImage(tensor)
    result = tio.Gamma(log_gamma=0.3)(image)
    assert isinstance(result, tio.ScalarImage) and result.applied_transforms[0].name == "Gamma"
    as_dict = tio.Gamma(log_gamma=0.3)({"a": tensor, "meta": 3})
    assert isinstance(as_dict["a"], torch.Tensor) and as_dict["meta"] == 3
    images = tio.ImagesBatch(torch.rand(2, 1, 6, 6, 6), [tio.AffineMatrix(), tio.AffineMatrix()])
    assert isinstance(tio.Gamma(log_gamma=0.3)(images), tio.ImagesBatch)
    with pytest.raises(TypeError, match="Expected Subject"):
        tio.Gamma(log_gamma=0.3)("not data")


def test_intensity_transforms_skip_label_maps_and_honour_include_exclude():
    s = make_subjects(10, 1, 3, second_modality=True)[0]
    seg = s.seg.data.clone()
    out = tio.Compose([tio.BiasField(), tio.Blur(std=1.0), tio.Noise(), tio.Gamma(log_gamma=0.2)])(s)
    assert torch.equal(out.seg.data, seg)
    out = tio.Noise(include=["t2"])(s)
    assert torch.equal(out.t1.data, s.t1.data) and not torch.equal(out.t2.data, s.t2.data)
    out = tio.Affine(degrees=(10, 10), exclude=["seg"])(s)
    assert torch.equal(out.seg.data, seg) and not torch.equal(out.t1.data, s.t1.data)
    assert out.applied_transforms[0].exclude == ["seg"]


def test_p_zero_and_p_one():
    s = subject()
    assert torch.equal(tio.Noise(p=0.0)(s).t1.data, s.t1.data)
    assert tio.Noise(p=0.0)(s).applied_transforms == []
    assert not torch.equal(tio.Noise(p=1.0)(s).t1.data, s.t1.data)


# -- spatial ------------------------------------------------------------------------
def test_noop_spatial_is_exact_identity():
    s = subject()
    with pytest.warns(UserWarning):
        transform = tio.Affine()
    out = transform(s)
    assert torch.equal(out.t1.data, s.t1.data) and out.t1.affine == s.t1.affine


def test_label_values_stay_in_input_set_and_seeded_reproducibility():
    s = subject(size=16)
    transform = tio.Spatial(degrees=(-20, 20), scales=(0.8, 1.2), max_displacement=4.0)
    torch.manual_seed(7)
    a = transform(s)
    torch.manual_seed(7)
    b = transform(s)
    assert torch.equal(a.t1.data, b.t1.data) and torch.equal(a.seg.data, b.seg.data)
    assert set(a.seg.data.unique().tolist()) <= set(s.seg.data.unique().tolist())
    torch.manual_seed(8)
    assert not torch.equal(transform(s).t1.data, a.t1.data)


def test_resample_bookkeeping():
    s = subject(size=12)
    out = tio.Resample(2)(s)
    assert out.t1.spatial_shape == (6, 6, 6) and out.t1.spacing == (2.0, 2.0, 2.0)
    assert out.seg.spatial_shape == (6, 6, 6)
    target = tio.ScalarImage(torch.zeros(1, 5, 7, 9), affine=tio.AffineMatrix.from_spacing((1.5, 1.0, 0.75)))
    out = tio.Resample(target)(s)
    assert out.t1.spatial_shape == (5, 7, 9) and np.allclose(out.t1.spacing, (1.5, 1.0, 0.75))
    out = tio.Resample("seg")(s)
    assert out.t1.spatial_shape == s.seg.spatial_shape
    with pytest.raises(ValueError, match="Unknown target"):
        tio.Resample("nope")(s)


def test_pad_value_semantics():
    s = subject(size=10)
    s.t1.set_data(s.t1.data + 5.0)  # minimum > 0
    far = tio.Affine(translation=(100.0, 0.0, 0.0))
    out = far(s)
    assert torch.all(out.t1.data == s.t1.data.min())  # "minimum" fill
    assert torch.all(out.seg.data == 0)  # default_pad_label
    out = tio.Affine(translation=(100.0, 0.0, 0.0), default_pad_value=-2.0, default_pad_label=9)(s)
    assert torch.all(out.t1.data == -2.0) and torch.all(out.seg.data == 9)


def test_shared_space_is_enforced():
    a = tio.ScalarImage(torch.rand(1, 8, 8, 8))
    b = tio.ScalarImage(torch.rand(1, 8, 8, 8), affine=tio.AffineMatrix.from_spacing((2, 2, 2)))
    with pytest.raises(RuntimeError, match="share the same affine"):
        tio.Affine(degrees=(10, 10))(tio.Subject(a=a, b=b))
    c = tio.ScalarImage(torch.rand(1, 8, 8, 6))
    with pytest.raises(RuntimeError, match="has shape"):
        tio.Affine(degrees=(10, 10))(tio.Subject(a=a, c=c))


# -- per-instance / per-element gating ----------------------------------------------
@pytest.mark.parametrize(
    "make",
    [
        lambda: tio.Affine(degrees=(-15, 15), p=0.5),
        lambda: tio.ElasticDeformation(p=0.5),
        lambda: tio.BiasField(p=0.5),
        lambda: tio.Blur(std=(0.5, 1.5), p=0.5),
        lambda: tio.Noise(rician=True, p=0.5),
        lambda: tio.Gamma(log_gamma=(-0.4, 0.4), p=0.5),
    ],
)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gated_out_rows_are_bit_exact_noops(make, dtype):
    data = batch(n=6, size=8, seed=11)
    data.t1.data = data.t1.data.to(dtype)
    original = data.t1.data.clone()
    for seed in range(50):  # first seed whose per-element gate keeps some rows and drops others
        torch.manual_seed(seed)
        out = make()(data)
        keep = out.applied_transforms[-1].params.get("_keep") if out.applied_transforms else None
        if keep is not None and any(keep) and not all(keep):
            break
    else:
        pytest.fail("no seed produced a mixed keep mask")
    for index, kept in enumerate(keep):
        same = torch.equal(out.t1.data[index], original[index])
        assert same == (not kept), f"element {index} kept={kept}"
    assert out.t1.data.dtype == dtype


def test_per_instance_equals_element_by_element():
    """The reference's `assert_vectorized` idea: batched result == each element with its sliced params."""
    data = batch(n=3, size=10, seed=5)
    for transform in (tio.Blur(std=(0.3, 1.5)), tio.BiasField(), tio.Gamma(log_gamma=(-0.3, 0.3)),
                      # numeric pad: the "minimum" fill is taken from the FIRST batch element (spatial.py:2054-2055)
                      tio.Spatial(degrees=(-10, 10), max_displacement=3.0, default_pad_value=0.25)):
        torch.manual_seed(1)
        out = transform(data)
        subjects = out.unbatch()
        for index, original in enumerate(data.unbatch()):
            params = subjects[index].applied_transforms[-1].params
            assert "_batched_keys" not in params
            single = tio.SubjectsBatch.from_subjects([copy.deepcopy(original)])
            redone = transform.apply_transform(single, params)
            torch.testing.assert_close(redone.t1.data[0], out.t1.data[index], rtol=1e-5, atol=1e-6)
            assert torch.equal(redone.seg.data[0], out.seg.data[index])


def test_per_instance_false_shares_parameters():
    data = batch(n=3, size=8, seed=2)
    data.t1.data = data.t1.data[:1].repeat(3, 1, 1, 1, 1)
    torch.manual_seed(0)
    out = tio.Affine(degrees=(-20, 20), per_instance=False)(data)
    assert "_batched_keys" not in out.applied_transforms[-1].params
    assert torch.equal(out.t1.data[0], out.t1.data[1]) and torch.equal(out.t1.data[1], out.t1.data[2])
    torch.manual_seed(0)
    out = tio.Affine(degrees=(-20, 20))(data)
    assert not torch.equal(out.t1.data[0], out.t1.data[1])


def test_dtype_promotion_follows_the_reference():
    half = tio.SubjectsBatch.from_subjects(make_subjects(8, 2, 0))
    half.t1.data = half.t1.data.half()
    assert tio.Affine(degrees=(5, 5))(half).t1.data.dtype == torch.float16  # resample casts back
    assert tio.Blur(std=1.0)(half).t1.data.dtype == torch.float16
    assert tio.Noise()(half).t1.data.dtype == torch.float32  # data + float32 noise promotes
    assert tio.BiasField()(half).t1.data.dtype == torch.float16  # per-instance path casts back
    assert tio.BiasField(per_instance=False)(half).t1.data.dtype == torch.float32  # shared path promotes
    assert tio.Gamma(log_gamma=(0.1, 0.2))(half).t1.data.dtype == torch.float32  # per-element exponent tensor
    assert tio.Gamma(log_gamma=0.2, per_instance=False)(half).t1.data.dtype == torch.float16


# -- history / inverse ---------------------------------------------------------------
def test_history_params_are_json_serialisable_and_replayable():
    import json

    s = subject(size=12)
    torch.manual_seed(0)
    out = tio.Compose([tio.Spatial(degrees=(-10, 10), max_displacement=3.0), tio.BiasField(), tio.Gamma(log_gamma=(-0.2, 0.2))])(s)
    for trace in out.applied_transforms:
        restored = json.loads(json.dumps(trace.params))
        assert restored == trace.params
    replay = tio.Spatial(degrees=0.0)
    redone = replay.apply_transform(tio.SubjectsBatch.from_subjects([copy.deepcopy(s)]), out.applied_transforms[0].params)
    again = tio.Spatial().apply_transform(tio.SubjectsBatch.from_subjects([copy.deepcopy(s)]), out.applied_transforms[0].params)
    assert torch.equal(redone.seg.data, again.seg.data)


def test_inverse_restores_geometry_and_intensity():
    s = subject(size=16)
    torch.manual_seed(0)
    out = tio.Compose([tio.Affine(degrees=(-10, 10), translation=(-2, 2)), tio.Gamma(log_gamma=(-0.3, 0.3)), tio.BiasField(), tio.Noise()])(s)
    with pytest.warns(UserWarning, match="Noise is not invertible"):
        restored = out.apply_inverse_transform()
    assert restored.applied_transforms == []
    assert restored.t1.spatial_shape == s.t1.spatial_shape and restored.t1.affine == s.t1.affine
    restored = out.apply_inverse_transform(warn=False, ignore_intensity=True)
    assert restored.t1.spatial_shape == s.t1.spatial_shape
    only = tio.Compose([tio.Gamma(log_gamma=0.4), tio.BiasField()])(s)
    back = only.apply_inverse_transform()
    torch.testing.assert_close(back.t1.data, s.t1.data, rtol=1e-4, atol=1e-5)


def test_batch_inverse_after_per_instance_transform():
    data = batch(n=3, size=10, seed=9)
    torch.manual_seed(4)
    out = tio.Compose([tio.Gamma(log_gamma=(-0.3, 0.3)), tio.BiasField()])(data)
    back = out.apply_inverse_transform()
    torch.testing.assert_close(back.t1.data, data.t1.data, rtol=1e-4, atol=1e-5)
    subjects = out.unbatch()
    back0 = subjects[0].apply_inverse_transform()
    torch.testing.assert_close(back0.t1.data, data.t1.data[0], rtol=1e-4, atol=1e-5)


def test_compose_operators_and_repr():
    pipeline = tio.Affine(degrees=(1, 2)) + tio.Noise(std=0.1)
    assert isinstance(pipeline, tio.Compose) and len(pipeline) == 2
    assert "Noise(std=0.1)" in repr(pipeline)
    assert repr(tio.Blur(std=(0.5, 2))) == "Blur(std=(0.5, 2))"


def test_lazy_copy_never_aliases_the_input(oracle):
    """copy=True on a batch shares tensors only until a transform replaces them (data/_lazy.py)."""
    import torchio_amd as tio
    from parity_harness import make_subjects, use_engine

    batch = tio.SubjectsBatch.from_subjects(make_subjects(12, 2, seed=3))
    original = batch.t1.data.clone()
    with use_engine(oracle):
        untouched = tio.Compose([])(batch)  # nothing replaces the data → must be cloned at scope exit
        assert untouched.t1.data.data_ptr() != batch.t1.data.data_ptr()
        assert torch.equal(untouched.t1.data, original)
        gated = tio.Gamma(log_gamma=(-0.3, 0.3), p=0.0)(batch)  # gated out as a whole
        assert gated.t1.data.data_ptr() != batch.t1.data.data_ptr()
        changed = tio.Gamma(log_gamma=(0.2, 0.3))(batch)
        assert changed.t1.data.data_ptr() != batch.t1.data.data_ptr()
        assert changed.seg.data.data_ptr() != batch.seg.data.data_ptr()  # label map untouched by Gamma → cloned
        assert not torch.equal(changed.t1.data, original)
    assert torch.equal(batch.t1.data, original)  # the caller's tensors are never modified


def test_lazy_params_reads_like_the_eager_dict(oracle):
    """Elastic control points are parked as tensors and turn into nested lists on any read."""
    import copy
    import json
    import pickle

    import torchio_amd as tio
    from parity_harness import make_subjects, use_engine
    from torchio_amd.transforms._lazy_params import LazyParams

    batch = tio.SubjectsBatch.from_subjects(make_subjects(12, 3, seed=5))
    transform = tio.ElasticDeformation()
    with use_engine(oracle):
        torch.manual_seed(7)
        out = transform(batch)
        torch.manual_seed(7)
        again = transform(batch)
    params = out.applied_transforms[-1].params
    assert isinstance(params, LazyParams) and isinstance(params, dict)
    parked, fields = params.raw("control_points")
    assert parked and len(fields) == 3 and all(isinstance(f, torch.Tensor) for f in fields)
    assert list(params)[:2] == ["selected_images", "original"]  # keys and their order are those of the eager dict
    # equality between two histories (both lazy) and against a materialised plain dict
    assert params == again.applied_transforms[-1].params
    as_text = json.dumps(params)  # the C encoder goes through .items()
    assert json.loads(as_text)["control_points"][0][1][2][3] == params["control_points"][0][1][2][3]
    assert not params.raw("control_points")[0]  # read once → stored as lists from now on
    assert isinstance(params["control_points"][0], list) and len(params["control_points"][0]) == 7
    clone = copy.deepcopy(again.applied_transforms[-1].params)
    assert type(clone) is dict and clone == params
    assert pickle.loads(pickle.dumps(params)) == dict(params)
    # the history still drives the inverse transform
    with use_engine(oracle):
        restored = out.apply_inverse_transform()
    assert restored.t1.data.shape == batch.t1.data.shape


# -- Resize / Anisotropy (F.interpolate users) ----------------------------------------------------
def test_resize_keeps_the_field_of_view_and_rescales_the_affines():
    s = subject(size=12)
    s.t1.affine = tio.AffineMatrix(np.diag([2.0, 3.0, 4.0, 1.0]))
    s.seg.affine = tio.AffineMatrix(np.diag([2.0, 3.0, 4.0, 1.0]))
    out = tio.Resize((6, 24, 12))(s)
    assert out.t1.shape == (1, 6, 24, 12) and out.seg.shape == (1, 6, 24, 12)
    assert out.t1.spacing == (4.0, 1.5, 4.0)  # old / new per axis: same physical extent
    assert s.t1.spacing == (2.0, 3.0, 4.0)  # the input is a copy: untouched
    assert set(out.seg.data.unique().tolist()) <= set(s.seg.data.unique().tolist())  # nearest for label maps
    assert tio.Resize(5)(s).t1.shape == (1, 5, 5, 5)
    same = tio.Resize(12)(s)
    assert torch.equal(same.t1.data, s.t1.data) and torch.equal(same.seg.data, s.seg.data)


def test_anisotropy_parameters_gating_and_errors():
    with pytest.raises(ValueError, match="upper bound must be >= 1"):
        tio.Anisotropy(downsampling=(0.2, 0.8))
    with pytest.warns(UserWarning, match="no-op"):
        tio.Anisotropy()
    s = subject(size=12)
    torch.manual_seed(4)
    out = tio.Anisotropy(axes=(2,), downsampling=(2, 4))(s)
    params = out.applied_transforms[-1].params
    assert params["axis"] == 2 and 2.0 <= params["factor"] <= 4.0
    assert out.t1.shape == s.t1.shape and not torch.equal(out.t1.data, s.t1.data)
    # along the untouched axes nothing is mixed: every (i, j) line is a function of the same line of the input
    assert set(out.seg.data.unique().tolist()) <= set(s.seg.data.unique().tolist())
    # per-instance: lists of axes / factors; gated-out elements are exact no-ops with factor 1
    b = batch(n=6, size=10)
    torch.manual_seed(5)
    result = tio.Anisotropy(downsampling=(1.5, 3), p=0.5)(b)
    params = result.applied_transforms[-1].params
    assert len(params["axis"]) == len(params["factor"]) == 6 and params["_batched_keys"] == ["axis", "factor"]
    for index, keep in enumerate(params["_keep"]):
        unchanged = torch.equal(result.images["t1"].data[index], b.images["t1"].data[index])
        assert unchanged == (not keep) and (keep or params["factor"][index] == 1.0)
    # shared parameters on a batch: one axis, one factor
    shared = tio.Anisotropy(downsampling=(2, 3), per_instance=False)(b)
    assert isinstance(shared.applied_transforms[-1].params["axis"], int)


# -- Pad / Crop ----------------------------------------------------------------------
def test_pad_and_crop_parse_shift_the_origin_and_invert_each_other():
    import torch.nn.functional as F

    sub = subject(10, 3)
    sub.t1.affine = tio.AffineMatrix(torch.tensor([[0.0, -2.0, 0.0, 5.0], [1.5, 0.0, 0.0, -3.0], [0.0, 0.0, 0.5, 1.0], [0, 0, 0, 1.0]], dtype=torch.float64))
    assert tio.Pad(padding=2).padding == (2,) * 6
    assert tio.Pad(padding=(1, 2, 3)).padding == (1, 1, 2, 2, 3, 3)
    assert tio.Crop(cropping=(1, 2, 3, 4, 5, 6)).cropping == (1, 2, 3, 4, 5, 6)
    with pytest.raises(ValueError, match="1, 3, or 6 values"):
        tio.Pad(padding=(1, 2))
    with pytest.raises(ValueError, match="1, 3, or 6 values"):
        tio.Crop(cropping=(1, 2, 3, 4))
    with pytest.raises(ValueError, match="padding_mode must be one of"):
        tio.Pad(padding=1, padding_mode="edge")

    padded = tio.Pad(padding=(2, 0, 1, 3, 0, 4), fill=9.0)(sub)
    assert padded.t1.spatial_shape == (12, 14, 14) and padded.seg.spatial_shape == (12, 14, 14)
    assert torch.equal(padded.t1.data, F.pad(sub.t1.data, (0, 4, 1, 3, 2, 0), value=9.0))
    # voxel (2, 1, 0) of the padded image is voxel (0, 0, 0) of the input: same world position
    world_before = sub.t1.affine.data @ torch.tensor([0, 0, 0, 1.0], dtype=torch.float64)
    world_after = padded.t1.affine.data @ torch.tensor([2, 1, 0, 1.0], dtype=torch.float64)
    torch.testing.assert_close(world_before, world_after, rtol=0, atol=1e-12)
    assert [t.name for t in padded.applied_transforms] == ["Pad"]
    restored = padded.apply_inverse_transform()
    assert torch.equal(restored.t1.data, sub.t1.data) and torch.equal(restored.seg.data, sub.seg.data)
    torch.testing.assert_close(restored.t1.affine.data, sub.t1.affine.data, rtol=0, atol=1e-12)

    cropped = tio.Crop(cropping=(2, 0, 1, 3, 0, 4))(sub)
    assert torch.equal(cropped.t1.data, sub.t1.data[:, 2:, 1:7, :6])
    back = cropped.apply_inverse_transform()  # Crop's inverse is a zero Pad of the same border
    assert back.t1.spatial_shape == sub.t1.spatial_shape
    assert torch.equal(back.t1.data[:, 2:, 1:7, :6], sub.t1.data[:, 2:, 1:7, :6]) and float(back.t1.data[:, :2].abs().max()) == 0.0
    torch.testing.assert_close(back.t1.affine.data, sub.t1.affine.data, rtol=0, atol=1e-12)


def test_pad_statistic_modes_batch_values_and_warnings():
    b = batch(3, 8, 5)
    b.t1.data = b.t1.data + torch.arange(3.0).reshape(3, 1, 1, 1, 1)  # a different statistic per element
    only_t1 = dict(include=["t1"])
    for mode, expected in (("minimum", b.t1.data.flatten(1).amin(1)), ("mean", b.t1.data.flatten(1).mean(1)), ("median", torch.quantile(b.t1.data.flatten(1), 0.5, dim=1))):
        out = tio.Pad(padding=(1, 0, 0, 2, 0, 0), padding_mode=mode, **only_t1)(b)
        assert out.t1.data.shape[2:] == (9, 10, 8) and out.seg.data.shape[2:] == (8, 8, 8)
        for index in range(3):
            border = torch.cat([out.t1.data[index, :, 0].flatten(), out.t1.data[index, :, :, 8:].flatten()])
            assert bool((border == expected[index]).all()), mode
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tio.Pad(padding=1, padding_mode="minimum")(subject(6))  # integer data + minimum: no warning
        with pytest.raises(RuntimeWarning, match="might be truncated"):
            tio.Pad(padding=1, padding_mode="mean")(subject(6))
    with pytest.raises(RuntimeError, match="doesn't take in value argument"):
        tio.Pad(padding=1, padding_mode="replicate", fill=1.0)(subject(6))
    with pytest.raises(ValueError, match="4D or 5D"):
        from torchio_amd.transforms.pad import pad_tensor

        pad_tensor(torch.zeros(4, 4, 4), (1,) * 6, "constant", 0.0)


def test_grid_sampler_padding_covers_the_border_with_full_weight():
    """The point of padding: after cropping the overlap//2 border of every patch, the ORIGINAL volume is tiled."""
    volume = torch.rand(1, 12, 10, 8)
    sub = tio.Subject(t1=tio.ScalarImage(volume))
    sampler = tio.GridSampler(sub, patch_size=6, patch_overlap=(2, 2, 4), padding_mode="replicate")
    assert sampler.subject.t1.spatial_shape == (14, 12, 12) and sub.t1.spatial_shape == (12, 10, 8)
    aggregator = tio.PatchAggregator(sampler.subject.spatial_shape, overlap_mode="crop", patch_overlap=(2, 2, 4))
    patches = [sampler[i] for i in range(len(sampler))]
    aggregator.add_batch(torch.stack([p.t1.data for p in patches]), [p.patch_location for p in patches])
    restored = tio.Crop(cropping=(1, 1, 1, 1, 2, 2))(tio.Subject(out=tio.ScalarImage(aggregator.get_output())))
    assert torch.equal(restored.out.data, volume)


# -- Motion --------------------------------------------------------------------------
def test_motion_parameters_gating_and_errors():
    with pytest.raises(ValueError, match="num_transforms"):
        tio.Motion(num_transforms=0)
    with pytest.raises(ValueError, match="num_transforms"):
        tio.Motion(num_transforms=1.5)
    sub = subject(12, 1)
    torch.manual_seed(3)
    out = tio.Motion(degrees=15, translation=4, num_transforms=3)(sub)
    (applied,) = out.applied_transforms
    assert applied.name == "Motion" and len(applied.params["transforms"]) == 3
    assert all(set(t) == {"degrees", "translation"} and len(t["degrees"]) == 3 for t in applied.params["transforms"])
    assert all(abs(v) <= 15 for t in applied.params["transforms"] for v in t["degrees"])
    assert not torch.equal(out.t1.data, sub.t1.data) and torch.equal(out.seg.data, sub.seg.data)  # label maps untouched
    assert out.t1.data.dtype == sub.t1.data.dtype and out.t1.shape == sub.t1.shape
    with pytest.raises(ValueError, match="motion segments"):
        tio.Motion(num_transforms=12)(sub)  # 13 segments on a 12-plane axis

    # zero motion: every segment is the still image and the slabs tile k-space, so the composite is the image
    still = tio.Motion(degrees=0, translation=0, num_transforms=4)(sub)
    torch.testing.assert_close(still.t1.data, sub.t1.data, rtol=0, atol=2e-6)

    b = batch(4, 10, 2)
    torch.manual_seed(5)
    out = tio.Motion(p=0.5)(b)
    lists = out.applied_transforms[-1].params["transforms"]
    assert len(lists) == 4 and any(len(entry) == 0 for entry in lists) and any(len(entry) == 2 for entry in lists)
    for index, entry in enumerate(lists):
        same = torch.equal(out.t1.data[index], b.t1.data[index])
        assert same == (len(entry) == 0)  # gated-out rows come back bit for bit
    shared = tio.Motion(per_instance=False)(b)
    assert isinstance(shared.applied_transforms[-1].params["transforms"][0], dict)
    from torchio_amd.transforms.motion import _apply_motion_per_instance

    with pytest.raises(ValueError, match="Expected 4 motion parameter lists"):
        _apply_motion_per_instance(b.t1.data, [[]])
    one = {"degrees": (1.0, 0.0, 0.0), "translation": (0.0, 0.0, 0.0)}
    with pytest.raises(ValueError, match="uniform motion transform counts"):
        _apply_motion_per_instance(b.t1.data, [[one], [one, one], [], [one]])


def test_unique_labels_on_the_oracle_engine(oracle):
    """The label table of the partial-volume mode: sorted distinct values as float64 (spatial.py:1360)."""
    g = torch.Generator().manual_seed(11)
    for dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.float32):
        data = torch.randint(-3 if dtype != torch.uint8 else 0, 9, (2, 1, 5, 6, 7), generator=g).to(dtype)
        table = oracle.unique_labels(data)
        assert table.dtype == torch.float64 and torch.equal(table, torch.unique(data).double())


# -- round-1 advisor findings ---------------------------------------------------------------
def test_crop_result_never_aliases_the_input_batch(oracle):
    """a transform that returns a VIEW of the borrowed tensor must still be cloned on the way out."""
    data = torch.arange(2 * 1 * 6 * 6 * 6, dtype=torch.float32).reshape(2, 1, 6, 6, 6)
    batch = tio.SubjectsBatch({"t1": tio.ImagesBatch(data.clone(), [tio.AffineMatrix() for _ in range(2)], image_class=tio.ScalarImage)})
    before = batch.t1.data.clone()
    with use_engine(oracle):
        out = tio.Crop(cropping=1)(batch)
    assert out.t1.data.untyped_storage().data_ptr() != batch.t1.data.untyped_storage().data_ptr()
    out.t1.data.add_(1)
    assert torch.equal(batch.t1.data, before)


def test_lazy_params_survive_plain_dict_copies():
    """dict(params) / {**params} / update() must see the materialised lists, not the placeholders."""
    from torchio_amd.transforms._lazy_params import LazyParams

    params = LazyParams(a=1)
    params.set_lazy("control_points", torch.ones(2, 2))
    assert dict(params)["control_points"] == [[1.0, 1.0], [1.0, 1.0]]
    params.set_lazy("affine_matrix", torch.eye(2))
    assert {**params}["affine_matrix"] == [[1.0, 0.0], [0.0, 1.0]]
    other = {}
    params.set_lazy("again", torch.zeros(1))
    other.update(params)
    assert other["again"] == [0.0] and list(other) == ["a", "control_points", "affine_matrix", "again"]


def test_scalar_draw_plan_follows_reassigned_ranges(oracle):
    """the draw-plan cache is keyed on the ranges' values, not on object ids."""
    transform = tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5))
    subject = tio.Subject(t1=tio.ScalarImage(torch.rand(1, 8, 8, 8)))
    with use_engine(oracle):
        transform(subject)
        transform.degrees = type(transform.degrees)(0)  # a new range object: no rotation at all
        torch.manual_seed(0)
        out = transform(subject)
    matrix = torch.as_tensor(out.applied_transforms[-1].params["affine_matrix"], dtype=torch.float64)
    off_diagonal = matrix[:3, :3] - torch.diag(torch.diagonal(matrix[:3, :3]))
    assert float(off_diagonal.abs().max()) < 1e-12, "the stale plan still rotated"


# -- round 2: packed uploads, the folded minimum's bookkeeping ------------------------------------------------
def test_h2d_packed_passes_host_tensors_through_on_a_cpu_target():
    from torchio_amd import ops

    a, b = torch.arange(6, dtype=torch.float32).view(2, 3), torch.ones(5)
    out = ops.h2d_packed([a, None, b], "cpu")
    assert out[1] is None and torch.equal(out[0], a) and torch.equal(out[2], b)


def test_folded_channel_min_record_is_dropped_by_an_in_place_write():
    from torchio_amd import ops

    data = torch.rand(2, 1, 4, 4, 4)
    assert ops.folded_channel_min(data) is None
    data._tio_channel_min = (data._version, torch.tensor([0.25]))
    assert torch.equal(ops.folded_channel_min(data), torch.tensor([0.25]))
    data.mul_(2.0)  # the values changed: the record no longer describes the tensor
    assert ops.folded_channel_min(data) is None


@pytest.mark.parametrize("count", [2, 3, 8])
@pytest.mark.parametrize(
    "kwargs",
    [
        dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5)),
        dict(degrees=(-10, 10), scales=(0.8, 1.2), isotropic=True),
        dict(degrees=(0, 0, 5, 15, -3, 3)),
        dict(max_displacement=7.5),
        dict(max_displacement=(2.0, 9.0), num_control_points=(5, 6, 7)),
        dict(max_displacement=(0.0, 4.0, 7.5, 7.5, 1.0, 2.0), degrees=(-5, 5), locked_borders=1),
        dict(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=(1.0, 6.0), num_control_points=4 + 1),
    ],
)
def test_batched_parameter_draw_equals_the_per_element_loop(kwargs, count):
    """`Spatial._sample_many`: same values, same generator state as `count` calls of `_sample_one` (the reference's order)."""
    transform = tio.Spatial(**kwargs)
    shape, affine = (16, 16, 16), tio.AffineMatrix()
    torch.manual_seed(123)
    loop = [transform._sample_one(shape, affine, build=False) for _ in range(count)]
    state_loop = torch.get_rng_state()
    torch.manual_seed(123)
    many = transform._sample_many(shape, affine, count)
    assert torch.equal(state_loop, torch.get_rng_state())
    for (fa, ca, da, ga), (fb, cb, db, gb) in zip(loop, many, strict=True):
        assert fa == fb and da == db and ga == gb
        assert (ca is None) == (cb is None) and (ca is None or torch.equal(ca, cb))


def test_gaussian_taps_from_one_block_equal_the_per_axis_form():
    """`_stacked_gaussian_taps`: the all-axes-at-once chain (per-instance draws, every sigma positive) gives the bits of the
    per-axis expressions it replaces (the reference's, blur.py:292-328)."""
    import math

    import numpy as np

    from torchio_amd.transforms.blur import _stacked_gaussian_taps

    def per_axis(sigmas):
        n = sigmas.shape[0]
        radii = np.array([[max(math.ceil(3 * v), 1) for v in row] for row in sigmas.tolist()], dtype=np.int64)
        radius = [int(radii[:, a].max()) for a in range(3)]
        taps = torch.zeros(n, 3, 2 * max(radius) + 1)
        for axis in range(3):
            r = radius[axis]
            offsets = torch.arange(2 * r + 1, dtype=torch.float32) - r
            column = torch.as_tensor(sigmas[:, axis], dtype=torch.float32)[:, None]
            kernels = torch.exp(-0.5 * (offsets[None, :] / column) ** 2)
            kernels = torch.where(offsets[None, :].abs() <= torch.as_tensor(radii[:, axis])[:, None], kernels, torch.zeros_like(kernels))
            taps[:, axis, : 2 * r + 1] = kernels / kernels.sum(dim=1, keepdim=True)
        return taps, radius

    rng = np.random.default_rng(5)
    for _ in range(400):
        sigmas = rng.uniform(0.05, 2.7, size=(int(rng.integers(2, 9)), 3))
        expected, expected_radius = per_axis(sigmas)
        taps, radius, skip = _stacked_gaussian_taps(sigmas, per_element=True)
        assert skip is None and radius == expected_radius and torch.equal(taps, expected)


# -- Compose draws ahead (round 4): gates and parameters of all children first, in order; then the children apply ----------
def _draw_ahead_pipeline(p=1.0):
    return tio.Compose([
        tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-2, 2), p=p),
        tio.ElasticDeformation(p=p),
        tio.BiasField(p=p),
        tio.Blur(std=(0.5, 1.5), p=p),
        tio.Noise(std=(0.05, 0.1), p=p),
        tio.Gamma(log_gamma=(-0.2, 0.2), p=p),
    ])


@pytest.mark.parametrize("p", [1.0, 0.6])
def test_compose_drawing_ahead_changes_nothing_observable(monkeypatch, p):
    """Same values bit for bit, same history (names and parameter dictionaries), same state of the global generator afterwards
    as the child-by-child road (`TIO_NO_DRAW_AHEAD=1`), with gates that fail and per-element keep masks in play."""
    import warnings

    subjects = [subject(size=12, seed=seed) for seed in range(3)]
    results = []
    for ahead in (True, False):
        monkeypatch.setenv("TIO_NO_DRAW_AHEAD", "0" if ahead else "1")
        pipeline = _draw_ahead_pipeline(p)
        assert pipeline._may_draw_ahead() is ahead
        outs = []
        for seed in range(4):
            torch.manual_seed(40 + seed)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = pipeline(tio.SubjectsBatch.from_subjects(subjects))
            outs.append((out.t1.data.clone(), out.seg.data.clone(), [(t.name, dict(t.params)) for t in out.applied_transforms], torch.rand(3)))
        results.append(outs)
    for with_ahead, without in zip(*results):
        assert torch.equal(with_ahead[0], without[0]) and torch.equal(with_ahead[1], without[1])
        assert with_ahead[2] == without[2]
        assert torch.equal(with_ahead[3], without[3])  # the generator stands where it stood


def test_compose_does_not_draw_ahead_across_a_change_of_geometry(monkeypatch):
    monkeypatch.delenv("TIO_NO_DRAW_AHEAD", raising=False)
    assert tio.Compose([tio.Affine(degrees=5), tio.Noise()])._may_draw_ahead()
    assert not tio.Compose([tio.Resample(target=2.0), tio.Noise()])._may_draw_ahead()      # the grid changes: later parameters may read it
    assert not tio.Compose([tio.Affine(degrees=5), tio.Flip(axes=(0,))])._may_draw_ahead()  # a child that does not say it may
    assert not tio.Compose([tio.Noise()])._may_draw_ahead()                                  # nothing to be ahead of
    hooked = tio.Noise()
    hooked.register_forward_hook(lambda module, args, output: output)
    assert not tio.Compose([tio.Affine(degrees=5), hooked])._may_draw_ahead()


def test_compose_prepares_threaded_children_first_and_applies_in_order(monkeypatch):
    """Draw-ahead road: the DRAWS and the APPLICATIONS keep the children's order (the global generator's sequence, the data's
    dependencies); only the preparations are reordered — a child whose `_prefetch` hands work to a native thread (Noise: the
    plan of its generator's stream) prepares before the others, so that their preparation runs beside that thread."""
    monkeypatch.delenv("TIO_NO_DRAW_AHEAD", raising=False)
    pipeline = tio.Compose([tio.Affine(degrees=5), tio.BiasField(), tio.Blur(std=(0.5, 1.0)), tio.Noise()])
    assert pipeline._may_draw_ahead() and tio.Noise.prefetch_is_threaded
    events = []
    for child in pipeline.transforms:
        name = type(child).__name__
        for phase in ("make_params", "_prefetch", "apply_transform"):
            original = getattr(child, phase)

            def wrapper(*args, _original=original, _name=name, _phase=phase, **kwargs):
                events.append((_phase, _name))
                return _original(*args, **kwargs)

            monkeypatch.setattr(child, phase, wrapper)
    torch.manual_seed(3)
    pipeline(tio.SubjectsBatch.from_subjects([subject(size=10, seed=1)]))
    order = ["Affine", "BiasField", "Blur", "Noise"]
    assert [name for phase, name in events if phase == "make_params"] == order
    assert [name for phase, name in events if phase == "apply_transform"] == order
    assert [name for phase, name in events if phase == "_prefetch"] == ["Noise", "Affine", "BiasField", "Blur"]
    first_apply = next(i for i, (phase, _) in enumerate(events) if phase == "apply_transform")
    assert all(phase != "_prefetch" for phase, _ in events[first_apply:])  # every preparation before the first application


def test_pending_flush_takes_explicit_draws_from_a_callable_and_keeps_the_plain_sequence_equivalent(monkeypatch):
    """`_pending.flush(noise=(mean, std, source))`: *source* may be a Philox seed word, a tensor of explicit draws (the reference's
    stream made ahead) or a callable returning one — called ONCE, after the parameter uploads; without a fused form the plain
    sequence hands the draws to `add_noise` as `base1` (a stub engine records the calls: no device involved)."""
    from torchio_amd import ops
    from torchio_amd.data import _pending

    calls = []

    class Stub:
        fused_answer = None

        def blur_fused(self, data, taps, radius, *, bias_coarse=None, noise=None):
            calls.append(("blur_fused", None if noise is None else type(noise[2]).__name__))
            return self.fused_answer

        def bias_field_apply(self, data, coarse):
            calls.append(("bias",))
            return data + 1

        def separable_conv3d(self, data, taps, radius):
            calls.append(("conv",))
            return data * 2

        def add_noise(self, data, mean, std, *, base1=None, philox_seed=0):
            calls.append(("add_noise", "base1" if base1 is not None else "philox", None if base1 is None else tuple(base1.shape)))
            return data + (0 if base1 is None else base1)

    stub = Stub()
    monkeypatch.setattr(ops, "engine", lambda: stub)
    uploads = []
    monkeypatch.setattr(ops, "h2d_packed", lambda tensors, device, **kw: (uploads.append(len(calls)), list(tensors))[1])
    data = torch.zeros(2, 1, 4, 4, 8)
    draws = torch.arange(data.numel(), dtype=torch.float32)
    asked = []

    def source():
        asked.append(len(uploads))  # (the uploads were enqueued before the draws were asked for)
        return draws

    pending = _pending.Pending(bias_coarse=torch.zeros(2, 1, 4, 4, 4), blur=(torch.ones(1, 3, 3), [1, 1, 1]))
    out = _pending.flush(data, pending, noise=(0.0, 1.0, source))
    assert asked == [1]
    assert calls == [("blur_fused", "Tensor"), ("bias",), ("conv",), ("add_noise", "base1", tuple(data.shape))]
    assert torch.equal(out, (data + 1) * 2 + draws.view(data.shape))
    # a Philox seed word keeps the in-kernel draws; a fused answer is returned as it is
    calls.clear()
    stub.fused_answer = torch.full_like(data, 7.0)
    pending = _pending.Pending(blur=(torch.ones(1, 3, 3), [1, 1, 1]))
    assert torch.equal(_pending.flush(data, pending, noise=(0.0, 1.0, 1234)), stub.fused_answer)
    assert calls == [("blur_fused", "int")]


def test_precision_and_draw_policy_switches():
    """Round 5: the third resampling precision and the draw policy of the reference's noise stream are process-wide switches
    with validated values (no GPU needed)."""
    import torchio_amd as tio
    from torchio_amd import _abi, ops

    assert ops.PRECISION_CODES == {"exact": _abi.PRECISION_EXACT, "fast": _abi.PRECISION_FAST, "tight": _abi.PRECISION_TIGHT}
    assert (_abi.PRECISION_EXACT, _abi.PRECISION_FAST, _abi.PRECISION_TIGHT) == (0, 1, 2)
    previous = tio.get_resample_precision()
    try:
        tio.set_resample_precision("tight")
        assert tio.get_resample_precision() == "tight"
        with pytest.raises(ValueError):
            tio.set_resample_precision("approximate")
        # round 6: "fast" is outside the per-voxel tolerance on noisy data and is refused without an explicit opt-in
        opted = ops._FAST_OPTED_IN
        ops._FAST_OPTED_IN = False
        try:
            with pytest.raises(ValueError, match="allow_out_of_tolerance"):
                tio.set_resample_precision("fast")
            assert tio.get_resample_precision() == "tight"
            tio.set_resample_precision("fast", allow_out_of_tolerance=True)
            assert tio.get_resample_precision() == "fast"
            tio.set_resample_precision("exact")
            tio.set_resample_precision("fast")  # (restoring a saved mode after the opt-in)
        finally:
            ops._FAST_OPTED_IN = opted
    finally:
        tio.set_resample_precision(previous)
    policy = tio.get_draw_policy()
    try:
        for name in ("gated", "free", "off"):
            tio.set_draw_policy(name)
            assert tio.get_draw_policy() == name
        with pytest.raises(ValueError):
            tio.set_draw_policy("sometimes")
        tio.set_draw_policy("off")
        assert ops.draw_stream("cuda:0") is None  # off: the draws are made on the data stream
    finally:
        tio.set_draw_policy(policy)


def test_rank_shares_are_whole_physical_cores(monkeypatch):
    """Round 5 (scripts/host_stress_ranks.py on the MI355X host): a rank's CPU share is made of WHOLE physical cores — with the
    usual numbering (logical CPU n and n + cores are siblings) a plain slice of the sorted ids gave every core to two ranks —
    and its worker budget is counted in physical cores."""
    from torchio_amd import distributed as tdist

    cores = 16
    usable = list(range(2 * cores))  # 16 cores x 2 threads: cpu n and n + 16 are siblings
    monkeypatch.setattr(tdist, "_core_groups", lambda cpus: [[c, c + cores] for c in sorted(set(c % cores for c in cpus)) if c in cpus and c + cores in cpus] or [[c] for c in cpus])
    shares = [tdist.plan_host_cpus(rank, 4, usable, [None] * 4) for rank in range(4)]
    assert sorted(c for share in shares for c in share) == usable  # a partition of the host
    for share in shares:
        assert len(share) == 8 and {c % cores for c in share} == {c % cores for c in share if c < cores}  # both siblings of each core
        assert len({c % cores for c in share}) == 4
    # four GPUs of one NUMA node (all 32 CPUs local to each): the node is split the same way
    node = [usable] * 4
    numa = [tdist.plan_host_cpus(rank, 4, usable, node) for rank in range(4)]
    assert sorted(c for share in numa for c in share) == usable
    assert all(len({c % cores for c in share}) == 4 for share in numa)
    # budget: physical cores of the rank's share minus the enqueue thread
    monkeypatch.setattr(tdist, "_usable_cpus", lambda: shares[0])
    monkeypatch.setattr(tdist, "local_world_size", lambda: 4)
    monkeypatch.setattr(tdist.os, "cpu_count", lambda: 2 * cores)
    assert tdist.host_thread_budget() == 3  # 4 physical cores - 1 (pinned: the mask already is this rank's share)


def _rotation_mapping(degrees, zoom=1.0) -> np.ndarray:
    from torchio_amd.transforms.spatial import _euler_to_rotation_matrix

    rotation = _euler_to_rotation_matrix(np.asarray(degrees, dtype=np.float64)) * zoom
    return np.concatenate([rotation, np.zeros((3, 1))], axis=1)[None].astype(np.float32)


def test_large_box_hint_follows_the_planner_arithmetic():
    """transforms/spatial.py `_expects_large_boxes` (TIO_GEOM_LARGE_BOXES / TIO_GEOM_MOSTLY_LARGE_BOXES, ABI 15): the box of a 16^3
    brick under the mapping, in floats, against the planned roads' staging tile — 0 over the bench's parameter ranges (calibrated
    against the planner's own descriptors on the GPU: scripts/r5_box_estimate.py), 1 as soon as ONE element's boxes exceed the
    tile (round 6: per element, the listed bricks are staged in passes behind the launch), 2 where most elements' do."""
    from torchio_amd.transforms.spatial import _expects_large_boxes

    shape, spacing = (256, 256, 256), (1.0, 1.0, 1.0)
    assert _expects_large_boxes(None, None, None, shape, spacing) == 0  # (the identity mapping is not even materialised)
    assert _expects_large_boxes(_rotation_mapping((0, 0, 0)), None, None, shape, spacing) == 0
    assert _expects_large_boxes(_rotation_mapping((10, 10, 10)), None, None, shape, spacing) == 0  # the bench's largest rotation (measured: staged)
    assert _expects_large_boxes(_rotation_mapping((25, 25, 25)), None, None, shape, spacing) == 2
    assert _expects_large_boxes(_rotation_mapping((0, 0, 0), 2.0), None, None, shape, spacing) == 2  # downsampling by two: 31-voxel boxes
    assert _expects_large_boxes(_rotation_mapping((0, 0, 180)), None, None, shape, spacing) == 0  # half a turn about one axis: the boxes of the identity
    # per ELEMENT: one large element in eight is enough for the list, most of them for the one-launch form
    small, large = _rotation_mapping((5, 5, 5)), _rotation_mapping((30, 30, 30))
    assert _expects_large_boxes(np.concatenate([large] + [small] * 7), None, None, shape, spacing) == 1
    assert _expects_large_boxes(np.concatenate([large] * 3 + [small] * 5), None, None, shape, spacing) == 1
    assert _expects_large_boxes(np.concatenate([large] * 4 + [small] * 4), None, None, shape, spacing) == 2
    # the displacement field's share: the bench's 7^3 control points / 7.5 mm on top of its LARGEST rotation list some bricks
    # (the fused tio.Spatial's 2 % that sampled voxel by voxel until round 5), on top of a typical one nothing
    assert _expects_large_boxes(_rotation_mapping((10, 10, 10)), [(7.5, 7.5, 7.5)], (7, 7, 7), shape, spacing) == 1
    assert _expects_large_boxes(_rotation_mapping((6, 6, 6)), [(7.5, 7.5, 7.5)], (7, 7, 7), shape, spacing) == 0
    # the bench's draws: never "most"
    transform = tio.Spatial(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-5, 5), max_displacement=7.5, per_instance=True)
    batch = tio.SubjectsBatch.from_subjects(make_subjects(16, 8, 1, with_label=False))
    from torchio_amd.transforms import spatial as sp

    affine_only = 0
    for seed in range(20):
        torch.manual_seed(seed)
        params = transform.make_params(batch)
        matrix, field, displacement, per_sample = sp._resolve_spatial_params(params)
        matrices = np.stack([np.linalg.inv(np.asarray(m, dtype=np.float64))[:3] for m in per_sample.affine_matrices]).astype(np.float32)
        assert _expects_large_boxes(matrices, per_sample.max_displacements, (7, 7, 7), shape, spacing) <= 1, seed
        affine_only += _expects_large_boxes(matrices, None, None, shape, spacing)
    assert affine_only <= 2  # (the headline's Affine launch, 16 elements per draw: a zoom-out of 1.1 on top of ~9 degrees — one draw in twenty)


def test_large_box_hint_reaches_the_geometry_struct():
    """ops.Engine: `large_boxes=True` sets TIO_GEOM_LARGE_BOXES in tio_resample_geom.flags of the launch AND of the plan query."""
    from torchio_amd import _abi
    from torchio_amd import ops

    seen = []

    class Fake(dict):
        def __missing__(self, key):
            def call(*args):
                seen.append((key, args[0]._obj.flags if hasattr(args[0], "_obj") else None))
                return 0

            return call

        def __contains__(self, key):
            return key != "last_error"

    engine = ops.Engine(Fake(), "cpu", "fake")
    data = torch.zeros(1, 1, 4, 4, 4)
    mapping = torch.eye(4)[:3].unsqueeze(0)
    for hint in (False, True):
        engine.resample3d([data], out_shape=(4, 4, 4), mapping=mapping, control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                          affine_first=True, interps=["linear"], fills=[None], large_boxes=hint)
    assert [flags for name, flags in seen if name == "resample3d"] == [0, _abi.GEOM_LARGE_BOXES]
