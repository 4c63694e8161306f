"""GPU parity: every C-ABI entry point of libtio_hip.so vs the CPU oracle.

Same seeded inputs through ``torchio_amd.ops.engine()`` (HIP, cuda tensors) and
``oracle.oracle_engine()`` (C restatement pinned against the reference, CPU
tensors).  Bars: bit-exact for resampling (nearest AND linear: the kernel follows
the oracle's float32 operation order), the stencil, noise with explicit draws and
the channel minimum; 1e-6 relative where a transcendental (exp / pow / log / cos)
is evaluated by different math libraries.
"""
from __future__ import annotations

import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _mapping(batch, seed, scale=0.15, shift=3.0):
    g = torch.Generator().manual_seed(seed)
    m = torch.eye(3, 4).repeat(batch, 1, 1)
    m[:, :, :3] += scale * torch.randn(batch, 3, 3, generator=g)
    m[:, :, 3] = shift * torch.randn(batch, 3, generator=g)
    return m.float()


def _control_points(batch, shape, seed, amplitude=4.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(batch, *shape, 3, generator=g) - 0.5) * 2 * amplitude).float()


def _data(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    if dtype.is_floating_point:
        return (torch.rand(*shape, generator=g) * 4 - 1).to(dtype)
    return torch.randint(0, 7, shape, generator=g).to(dtype)


def _both(oracle, hip, fn, tensors, **kwargs):
    """Run engine method *fn* on CPU (oracle) and GPU (hip) with the same arguments."""

    def move(obj, device):
        if isinstance(obj, torch.Tensor):
            return obj.to(device)
        if isinstance(obj, (list, tuple)):
            return type(obj)(move(o, device) for o in obj)
        return obj

    cpu = getattr(oracle, fn)(*move(tensors, "cpu"), **{k: move(v, "cpu") for k, v in kwargs.items()})
    gpu = getattr(hip, fn)(*move(tensors, DEV), **{k: move(v, DEV) for k, v in kwargs.items()})
    torch.cuda.synchronize()
    return cpu, gpu


RESAMPLE_CASES = list(
    itertools.product(
        [False, True],  # elastic
        [True, False],  # affine_first
        ["linear", "nearest"],
        [True, False],  # fill
    )
)


@pytest.mark.parametrize("elastic,affine_first,interp,with_fill", RESAMPLE_CASES)
def test_resample_matches_oracle_bit_exact(oracle, hip, elastic, affine_first, interp, with_fill):
    batch, channels = 2, 2
    in_shape, out_shape = (20, 24, 70), (22, 19, 67)
    data = _data((batch, channels, *in_shape), torch.float32, 1)
    kwargs = dict(
        out_shape=out_shape,
        mapping=_mapping(1, 2),
        control_points=_control_points(1, (7, 6, 5), 3) if elastic else None,
        in_spacing=(1.0, 1.25, 0.8),
        out_spacing=(0.9, 1.1, 0.75),
        affine_first=affine_first,
        interps=[interp],
        fills=[torch.tensor([-1.0, 0.5]) if with_fill else None],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


@pytest.mark.parametrize(
    "dtype", [torch.float64, torch.float16, torch.bfloat16, torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64]
)
@pytest.mark.parametrize("interp", ["linear", "nearest"])
def test_resample_dtypes(oracle, hip, dtype, interp):
    data = _data((1, 2, 17, 18, 40), dtype, 4)
    kwargs = dict(
        out_shape=(17, 18, 40),
        mapping=_mapping(1, 5, scale=0.1),
        control_points=_control_points(1, (7, 7, 7), 6),
        in_spacing=(1, 1, 1),
        out_spacing=(1, 1, 1),
        affine_first=True,
        interps=[interp],
        fills=[torch.tensor([2.0, 3.0])],
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert gpu[0].dtype == dtype
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_resample_multi_image_per_instance_flags(oracle, hip):
    """2 float modalities + int label map, per-element matrices/fields, skip + passthrough rows."""
    batch = 4
    shape = (16, 20, 66)
    t1 = _data((batch, 1, *shape), torch.float32, 7)
    t2 = _data((batch, 2, *shape), torch.float32, 8)
    seg = _data((batch, 1, *shape), torch.int16, 9)
    kwargs = dict(
        out_shape=shape,
        mapping=_mapping(batch, 10, scale=0.08),
        control_points=_control_points(batch, (7, 7, 7), 11),
        in_spacing=(1, 1, 1),
        out_spacing=(1, 1, 1),
        affine_first=True,
        interps=["linear", "linear", "nearest"],
        fills=[torch.tensor([0.25]), torch.tensor([-1.0, 0.0]), None],
        cp_skip=torch.tensor([0, 1, 0, 0], dtype=torch.uint8),
        passthrough=torch.tensor([0, 0, 1, 0], dtype=torch.uint8),
    )
    cpu, gpu = _both(oracle, hip, "resample3d", ([t1, t2, seg],), **kwargs)
    for c, g in zip(cpu, gpu, strict=True):
        assert torch.equal(c, g.cpu())
    assert torch.equal(gpu[2][2].cpu(), seg[2])  # passthrough row is bit-exact input


def test_resample_identity_and_half_voxel_ties(oracle, hip):
    """Identity mapping reproduces the input; a half-voxel shift exercises round-half-to-even."""
    data = _data((1, 1, 9, 10, 65), torch.int32, 12)
    ident = torch.eye(3, 4).reshape(1, 3, 4)
    common = dict(out_shape=(9, 10, 65), control_points=None, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                  affine_first=True, interps=["nearest"], fills=[None])
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), mapping=ident, **common)
    assert torch.equal(gpu[0].cpu(), data) and torch.equal(cpu[0], data)
    half = ident.clone()
    half[0, :, 3] = 0.5
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), mapping=half, **common)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_resample_2d_and_out_of_view(oracle, hip):
    data = _data((2, 1, 33, 65, 1), torch.float32, 13)  # 2-D image: K == 1
    far = torch.eye(3, 4).reshape(1, 3, 4).clone()
    far[0, 0, 3] = 500.0  # everything maps outside → all fill
    for mapping in (_mapping(1, 14, scale=0.05, shift=1.0), far):
        mapping = mapping.clone()
        mapping[:, 2, :] = torch.tensor([0.0, 0.0, 1.0, 0.0])
        kwargs = dict(out_shape=(33, 65, 1), mapping=mapping, control_points=None, in_spacing=(1, 1, 1),
                      out_spacing=(1, 1, 1), affine_first=True, interps=["linear"], fills=[torch.tensor([7.0])])
        cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
        assert torch.equal(cpu[0], gpu[0].cpu())
    assert bool((gpu[0] == 7.0).all())


def test_channel_min(oracle, hip):
    for dtype in (torch.float32, torch.float16, torch.int16, torch.float64):
        data = _data((3, 4, 11, 13, 17), dtype, 15)
        cpu, gpu = _both(oracle, hip, "channel_min", (data,))
        assert torch.equal(cpu, gpu.cpu())
        assert torch.equal(cpu, data[0].float().amin(dim=(1, 2, 3)))


def _taps(batch, sigmas, stride):
    taps = torch.zeros(batch, 3, stride)
    radius = [0, 0, 0]
    for b in range(batch):
        for axis in range(3):
            s = sigmas[b][axis]
            if s <= 0:
                continue
            r = max(int(-(-3 * s // 1)), 1)
            radius[axis] = max(radius[axis], r)
    for b in range(batch):
        for axis in range(3):
            r = radius[axis]
            if r == 0:
                continue
            s = sigmas[b][axis]
            x = torch.arange(2 * r + 1, dtype=torch.float32) - r
            if s > 0:
                k = torch.exp(-0.5 * (x / s) ** 2)
                k[(x.abs() > max(int(-(-3 * s // 1)), 1))] = 0
            else:
                k = (x == 0).float()
            taps[b, axis, : 2 * r + 1] = k / k.sum()
    return taps, radius


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.float64, torch.bfloat16])
def test_separable_conv_matches_oracle(oracle, hip, dtype):
    data = _data((2, 2, 19, 23, 70), dtype, 16)
    taps, radius = _taps(1, [(1.3, 0.6, 2.0)], 32)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius))
    assert torch.equal(cpu, gpu.cpu())


def test_separable_conv_per_element_and_skip(oracle, hip):
    data = _data((3, 1, 12, 14, 66), torch.float32, 17)
    taps, radius = _taps(3, [(1.0, 0.0, 0.7), (0.0, 0.0, 0.0), (0.4, 0.0, 1.9)], 16)
    assert radius[1] == 0
    skip = torch.tensor([0, 1, 0], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius), skip=skip)
    assert torch.equal(cpu, gpu.cpu())
    assert torch.equal(gpu[1].cpu(), data[1])
    # single-axis and no-axis corner cases
    taps1, radius1 = _taps(1, [(0.0, 0.0, 1.1)], 16)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps1, radius1))
    assert torch.equal(cpu, gpu.cpu())
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps1, [0, 0, 0]))
    assert torch.equal(gpu.cpu(), data)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
@pytest.mark.parametrize("divide", [False, True])
def test_bias_field(oracle, hip, dtype, divide):
    data = _data((3, 2, 18, 21, 68), dtype, 18)
    g = torch.Generator().manual_seed(19)
    coarse = 0.5 * torch.randn(3, 2, 4, 5, 6, generator=g)
    skip = torch.tensor([0, 0, 1], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "bias_field_apply", (data, coarse), divide=divide, skip=skip)
    rtol = {torch.float32: 2e-6, torch.float64: 2e-6, torch.float16: 2e-3}[dtype]
    torch.testing.assert_close(gpu.cpu(), cpu, rtol=rtol, atol=0)
    assert torch.equal(gpu[2].cpu(), data[2])


@pytest.mark.parametrize("rician", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_noise_with_explicit_draws_is_exact(oracle, hip, rician, dtype):
    data = _data((3, 2, 9, 11, 31), dtype, 20)
    g = torch.Generator().manual_seed(21)
    base1 = torch.randn(data.shape, generator=g)
    base2 = torch.randn(data.shape, generator=g)
    mean = torch.tensor([0.1, 0.0, -0.2])
    std = torch.tensor([0.25, 0.0, 0.5])
    keep = torch.tensor([1, 0, 1], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "add_noise", (data, mean, std), rician=rician, base1=base1, base2=base2, keep=keep)
    if rician:
        torch.testing.assert_close(gpu.cpu(), cpu, rtol=1e-6, atol=1e-7)
    else:
        assert torch.equal(cpu, gpu.cpu())
    assert torch.equal(gpu[1].cpu(), data[1])
    cpu, gpu = _both(oracle, hip, "add_noise", (data, 0.05, 0.3), rician=False, base1=base1)
    assert torch.equal(cpu, gpu.cpu())


def test_philox_stream_and_fast_noise(oracle, hip):
    n = 4 * 1000 + 3
    cpu = oracle.philox_normal((n,), 1234567890123, 0, "cpu")
    gpu = hip.philox_normal((n,), 1234567890123, 0, DEV)
    # (round 5: sin / cos by the same IEEE polynomial on both sides — what is left is the radius: hardware log2 / sqrt vs libm)
    torch.testing.assert_close(gpu.cpu(), cpu, rtol=4e-7, atol=1e-7)
    big = hip.philox_normal((1 << 22,), 42, 1, DEV)
    wide = oracle.philox_normal((1 << 22,), 42, 1, "cpu")
    assert float((big.cpu() - wide).abs().max()) <= 2e-6 and float(((big.cpu() - wide).abs() / wide.abs().clamp_min(1.0)).max()) <= 5e-7
    assert abs(float(big.mean())) < 3e-3 and abs(float(big.std()) - 1.0) < 3e-3
    data = _data((2, 1, 8, 8, 64), torch.float32, 22)
    cpu, gpu = _both(oracle, hip, "add_noise", (data, 0.0, 0.25), philox_seed=99)
    torch.testing.assert_close(gpu.cpu(), cpu, rtol=0, atol=1e-6)
    z = hip.philox_normal(data.shape, 99, 0, DEV)
    torch.testing.assert_close(gpu, data.to(DEV) + 0.25 * z, rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16])
def test_gamma(oracle, hip, dtype):
    data = _data((3, 2, 9, 11, 31), dtype, 23)
    data[0, 0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.0]).to(dtype)
    cpu, gpu = _both(oracle, hip, "gamma_pow", (data, torch.tensor([0.8, 1.0, 1.3])))
    rtol = 2e-3 if dtype == torch.float16 else 2e-6
    torch.testing.assert_close(gpu.cpu(), cpu, rtol=rtol, atol=0)
    cpu, gpu = _both(oracle, hip, "gamma_pow", (data, 1.25))
    torch.testing.assert_close(gpu.cpu(), cpu, rtol=rtol, atol=0)


def test_engine_rejects_cpu_tensors(hip):
    from torchio_amd.ops import EngineError

    with pytest.raises(EngineError):
        hip.gamma_pow(torch.rand(1, 1, 2, 2, 2), 1.5)


@pytest.mark.parametrize(
    "shape",
    [(2, 2, 19, 23, 72), (1, 1, 70, 40, 64), (1, 1, 9, 50, 520), (1, 2, 37, 5, 256), (1, 1, 16, 14, 12), (2, 1, 9, 33, 4)],
)
def test_separable_conv_float4_paths(oracle, hip, shape):
    """K % 4 == 0 float32: the marching / 16-byte kernels (several steps, segments and K tiles)."""
    data = _data(shape, torch.float32, 61)
    taps, radius = _taps(1, [(1.3, 0.6, 2.0)], 32)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius))
    assert torch.equal(cpu, gpu.cpu())


def test_separable_conv_float4_per_element_skip_and_single_axes(oracle, hip):
    data = _data((3, 1, 40, 36, 128), torch.float32, 62)
    taps, radius = _taps(3, [(1.0, 1.7, 0.7), (0.0, 0.0, 0.0), (0.4, 0.5, 1.9)], 16)
    skip = torch.tensor([0, 1, 0], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius), skip=skip)
    assert torch.equal(cpu, gpu.cpu())
    assert torch.equal(gpu[1].cpu(), data[1])
    for sigmas in [(2.0, 0.0, 0.0), (0.0, 1.2, 0.0), (0.0, 0.0, 0.9)]:
        taps1, radius1 = _taps(1, [sigmas], 16)
        cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps1, radius1))
        assert torch.equal(cpu, gpu.cpu())


@pytest.mark.parametrize("sigmas", [(0.5, 0.5, 3.5), (3.1, 4.0, 0.4), (0.5, 0.5, 6.0), (5.5, 0.3, 2.7)])
def test_separable_conv_float4_radius_classes(oracle, hip, sigmas):
    """K radius <= 8 (register window), 9..16 (LDS loop), > 16 (generic kernels); long line radii."""
    data = _data((1, 1, 45, 41, 128), torch.float32, 63)
    taps, radius = _taps(1, [sigmas], 48)
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius))
    assert torch.equal(cpu, gpu.cpu())


# -- degenerate sizes: empty batch, one-voxel axes, one-voxel volumes -------------------------
def test_empty_batch_is_a_no_op_everywhere(hip):
    """B = 0: every entry point returns an empty tensor of the right shape and launches nothing."""
    empty = torch.zeros(0, 2, 8, 8, 8, device=DEV)
    out = hip.resample3d([empty], out_shape=(6, 7, 8), mapping=torch.eye(3, 4, device=DEV)[None], control_points=None,
                         in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True, interps=["linear"], fills=[None])[0]
    assert out.shape == (0, 2, 6, 7, 8)
    taps = torch.full((1, 3, 3), 1 / 3, device=DEV)
    assert hip.separable_conv3d(empty, taps, [1, 1, 1]).shape == empty.shape
    assert hip.gamma_pow(empty, 1.5).shape == empty.shape
    torch.cuda.synchronize()


@pytest.mark.parametrize("in_shape,out_shape", [((1, 1, 1), (1, 1, 1)), ((1, 9, 1), (3, 4, 2)), ((5, 1, 7), (5, 1, 7)), ((2, 2, 2), (1, 1, 1))])
@pytest.mark.parametrize("interp", ["linear", "nearest", "label"])
def test_resample_degenerate_shapes_match_oracle(oracle, hip, in_shape, out_shape, interp):
    """Axes of length 1 (2-D slices, single voxels): max(S - 1, 1) in the normalisation, taps on the border."""
    batch = 2
    dtype = torch.float32 if interp == "linear" else torch.int16
    data = _data((batch, 1, *in_shape), dtype, 81)
    kwargs = dict(
        out_shape=out_shape, mapping=_mapping(batch, 82, scale=0.3, shift=0.4), control_points=None, in_spacing=(1, 1, 1),
        out_spacing=(1, 1, 1), affine_first=True, interps=[interp], fills=[torch.tensor([9.0]) if interp != "label" else None],
    )
    if interp == "label":
        kwargs.update(label_tables=[torch.unique(data).double()], pad_labels=[5.0])
    cpu, gpu = _both(oracle, hip, "resample3d", ([data],), **kwargs)
    assert torch.equal(cpu[0], gpu[0].cpu())


def test_blur_on_one_voxel_axes_matches_oracle(oracle, hip):
    """Replicate padding over an axis of length 1 (every tap reads the same voxel)."""
    data = _data((2, 1, 1, 6, 4), torch.float32, 83)
    taps = torch.tensor([[0.25, 0.5, 0.25]]).repeat(3, 1)[None].contiguous()
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, [1, 1, 1]))
    assert torch.equal(cpu, gpu.cpu())


# -- F.interpolate users (Resize / Anisotropy) ----------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.uint8, torch.int16, torch.int64])
@pytest.mark.parametrize("mode", ["nearest", "linear"])
def test_interpolate3d_matches_oracle_bit_exact(oracle, hip, dtype, mode):
    data = _data((2, 2, 13, 9, 20), dtype, 91)
    for out_shape in [(20, 9, 7), (5, 18, 33), (13, 9, 20), (1, 1, 1)]:
        cpu, gpu = _both(oracle, hip, "interpolate3d", (data, out_shape, mode))
        assert gpu.dtype == dtype and tuple(gpu.shape[2:]) == out_shape
        assert torch.equal(cpu, gpu.cpu())


def test_interpolate3d_equals_stock_aten_on_the_device(hip):
    """The semantic anchor: torch's own F.interpolate on the same GPU (nearest exactly; trilinear within 1 ulp of the fma order)."""
    import torch.nn.functional as F

    data = torch.rand(2, 3, 40, 36, 44, device=DEV)
    for out_shape in [(64, 20, 44), (17, 50, 90)]:
        assert torch.equal(hip.interpolate3d(data, out_shape, "nearest"), F.interpolate(data, size=out_shape, mode="nearest"))
        ours = hip.interpolate3d(data, out_shape, "linear")
        torch.testing.assert_close(ours, F.interpolate(data, size=out_shape, mode="trilinear", align_corners=True), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int16])
@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("blend", [False, True])
def test_axis_gather_lerp_matches_oracle_bit_exact(oracle, hip, dtype, axis, blend):
    data = _data((3, 2, 10, 12, 14), dtype, 92)
    length = data.shape[2 + axis]
    g = torch.Generator().manual_seed(93)
    lower = torch.randint(0, length, (3, length), generator=g, dtype=torch.int32)
    upper = torch.randint(0, length, (3, length), generator=g, dtype=torch.int32) if blend else None
    weight = torch.rand(3, length, generator=g) if blend else None
    active = torch.tensor([1, 0, 1], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "axis_gather_lerp", (data, axis, lower, upper, weight, active))
    assert torch.equal(cpu, gpu.cpu())
    assert torch.equal(gpu[1].cpu(), data[1])  # inactive element: untouched


_MOVE_DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64]


@pytest.mark.parametrize("dtype", _MOVE_DTYPES)
def test_flip3d_equals_torch_flip(oracle, hip, dtype):
    data = _data((3, 2, 7, 9, 11), dtype, 101)
    for axes in ([0], [1], [2], [0, 2], [0, 1, 2], []):
        cpu, gpu = _both(oracle, hip, "flip3d", (data, axes))
        expected = torch.flip(data, [2 + a for a in axes]) if axes else data
        assert torch.equal(cpu, expected) and torch.equal(gpu.cpu(), expected)
    flags = torch.tensor([[1, 0, 1], [0, 0, 0], [0, 1, 0]], dtype=torch.uint8)
    cpu, gpu = _both(oracle, hip, "flip3d", (data,), per_element=flags)
    expected = torch.stack([torch.flip(data[0], [1, 3]), data[1], torch.flip(data[2], [2])])
    assert torch.equal(cpu, expected) and torch.equal(gpu.cpu(), expected)


@pytest.mark.parametrize("dtype", _MOVE_DTYPES)
@pytest.mark.parametrize("mode", ["constant", "reflect", "replicate", "circular"])
def test_pad3d_matches_oracle_and_f_pad(oracle, hip, dtype, mode):
    import torch.nn.functional as F

    data = _data((2, 3, 6, 9, 11), dtype, 103)
    padding = (2, 5, 0, 3, 4, 1)
    fill = -3.0 if mode == "constant" and dtype != torch.uint8 else (3.0 if mode == "constant" else 0.0)
    cpu, gpu = _both(oracle, hip, "pad3d", (data, padding, mode), fill=fill)
    assert torch.equal(cpu, gpu.cpu())
    assert gpu.shape == (2, 3, 13, 12, 16)
    # stock ATen on the device as a second witness (where it implements the dtype)
    kwargs = {"value": fill} if mode == "constant" else {}
    try:
        aten = F.pad(data.to(DEV), (4, 1, 0, 3, 2, 5), mode=mode, **kwargs)
    except (RuntimeError, NotImplementedError):
        return
    assert torch.equal(aten, gpu)


def test_pad3d_per_element_constants_and_limits(oracle, hip):
    data = _data((3, 2, 5, 6, 7), torch.float32, 105)
    fills = torch.tensor([0.25, -7.0, 1e9])
    cpu, gpu = _both(oracle, hip, "pad3d", (data, (1, 2, 3, 0, 0, 4)), fill_per_element=fills)
    assert torch.equal(cpu, gpu.cpu())
    assert all(float(gpu[b, 0, 0, 0, 0]) == float(fills[b]) for b in range(3))
    assert torch.equal(gpu[:, :, 1:6, 3:, :7].cpu(), data)
    big = _data((1, 1, 40, 50, 300), torch.int16, 107)  # > one block per row, odd extents
    cpu, gpu = _both(oracle, hip, "pad3d", (big, (3, 3, 7, 7, 33, 31), "reflect"))
    assert torch.equal(cpu, gpu.cpu())
    from torchio_amd.ops import EngineError

    with pytest.raises(EngineError, match="reflect padding must be smaller"):
        hip.pad3d(data.to(DEV), (5, 0, 0, 0, 0, 0), "reflect")
    with pytest.raises(EngineError, match="circular padding must not exceed"):
        hip.pad3d(data.to(DEV), (0, 0, 0, 7, 0, 0), "circular")
    with pytest.raises(EngineError, match="paddings >= 0"):
        hip.pad3d(data.to(DEV), (0, 0, 0, -1, 0, 0))
    assert hip.pad3d(data[:0].to(DEV), (1, 1, 1, 1, 1, 1)).shape == (0, 2, 7, 8, 9)


def _segments(shape, n, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(*shape, generator=g) * 4 - 1) for _ in range(n)]


@pytest.mark.parametrize(
    "shape,bounds",
    [
        ((2, 1, 12, 6, 5), [0, 4, 8, 12]),           # unaligned plane, one row tile
        ((1, 2, 16, 8, 8), [0, 8, 16]),              # float4 path
        ((1, 1, 150, 7, 9), [0, 37, 74, 111, 150]),  # two row tiles per wave (I > 128), ragged rows and columns
        ((1, 1, 260, 4, 36), [0, 130, 260]),         # two blocks along the rows (I > 256)
        ((3, 1, 9, 12, 1), [0, 9]),                  # one segment: the identity map
        ((1, 1, 33, 20, 20), [0, 0, 11, 33]),        # an empty slab
    ],
)
def test_kspace_segment_mix_matches_the_fft_route(oracle, hip, shape, bounds):
    segments = _segments(shape, len(bounds) - 1, 201)
    cpu = oracle.kspace_segment_mix(segments, bounds, torch.float32)
    gpu = hip.kspace_segment_mix([s.to(DEV) for s in segments], bounds, torch.float32)
    torch.cuda.synchronize()
    # two float32 evaluations of the same sums (float64 DFT on the oracle side): <= 1e-5 of the magnitude
    assert float((cpu - gpu.cpu()).abs().max()) <= 1e-5 * 4
    spectrum = torch.fft.fftn(segments[0].double(), dim=(-3, -2, -1))
    for s in range(1, len(segments)):
        spectrum[:, :, bounds[s] : bounds[s + 1]] = torch.fft.fftn(segments[s].double(), dim=(-3, -2, -1))[:, :, bounds[s] : bounds[s + 1]]
    expected = torch.fft.ifftn(spectrum, dim=(-3, -2, -1)).real
    assert float((expected - gpu.cpu().double()).abs().max()) <= 1e-5 * 4


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16, torch.bfloat16, torch.int16, torch.uint8])
def test_kspace_segment_mix_output_dtypes_and_inactive_rows(oracle, hip, dtype):
    shape, bounds = (3, 2, 10, 6, 8), [0, 5, 10]
    segments = [s * 20 + 30 for s in _segments(shape, 2, 203)]  # positive, integer-scale values
    active = torch.tensor([1, 0, 1], dtype=torch.uint8)
    cpu = oracle.kspace_segment_mix(segments, bounds, dtype, active=active)
    gpu = hip.kspace_segment_mix([s.to(DEV) for s in segments], bounds, dtype, active=active.to(DEV))
    torch.cuda.synchronize()
    assert gpu.dtype == dtype
    diff = (cpu[[0, 2]].double() - gpu.cpu()[[0, 2]].double()).abs().max()
    # float outputs: rounding of the storage type; integer outputs truncate, so a 1e-5 difference may cross an integer
    bound = {torch.float64: 1e-3, torch.float16: 0.07, torch.bfloat16: 0.6}.get(dtype, 1.0)
    assert float(diff) <= bound
    if not dtype.is_floating_point:
        assert float((cpu[[0, 2]].double() != gpu.cpu()[[0, 2]].double()).double().mean()) < 1e-3


def test_kspace_segment_mix_properties_at_full_size(hip):
    """256^3 (the bench volume): slabs tile k-space, so equal segments give the image back; the map is linear."""
    g = torch.Generator().manual_seed(205)
    x = torch.rand(1, 1, 256, 256, 256, generator=g).to(DEV)
    y = torch.rand(1, 1, 256, 256, 256, generator=g).to(DEV)
    bounds = [0, 85, 170, 256]
    same = hip.kspace_segment_mix([x, x, x], bounds, torch.float32)
    assert float((same - x).abs().max()) <= 1e-5
    mixed = hip.kspace_segment_mix([x, y, x], bounds, torch.float32)
    other = hip.kspace_segment_mix([y, x, y], bounds, torch.float32)
    assert float((mixed + other - (x + y)).abs().max()) <= 2e-5  # W_0 + W_1 + W_2 = identity
    assert float((mixed - x).abs().max()) > 0.1
    # the FFT route on the same device, for one (j, k) column block only (a full complex 256^3 is 128 MiB per copy)
    xs, ys = x[..., :8, :8].contiguous(), y[..., :8, :8].contiguous()
    spectrum = torch.fft.fft(xs.double(), dim=2)
    spectrum[:, :, 85:170] = torch.fft.fft(ys.double(), dim=2)[:, :, 85:170]
    expected = torch.fft.ifft(spectrum, dim=2).real
    assert float((expected - mixed[..., :8, :8].double()).abs().max()) <= 1e-5


def test_kspace_segment_mix_argument_errors(hip):
    from torchio_amd.ops import EngineError

    x = torch.rand(1, 1, 8, 4, 4, device=DEV)
    with pytest.raises(EngineError, match="bounds must run from 0"):
        hip.kspace_segment_mix([x, x], [0, 4, 7], torch.float32)
    with pytest.raises(ValueError, match="float32 tensors of one shape"):
        hip.kspace_segment_mix([x, x.double()], [0, 4, 8], torch.float32)
    with pytest.raises(ValueError, match="at most"):
        hip.kspace_segment_mix([x] * 40, list(range(41)), torch.float32)
    assert hip.kspace_segment_mix([x[:0], x[:0]], [0, 4, 8], torch.float32).shape == (0, 1, 8, 4, 4)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int8, torch.int16])
def test_unique_labels_equals_torch_unique(oracle, hip, dtype):
    g = torch.Generator().manual_seed(301)
    info = torch.iinfo(dtype)
    cases = [
        torch.randint(0, 5, (2, 1, 17, 19, 23), generator=g).to(dtype),                                  # a few labels, odd size
        torch.randint(info.min, info.max + 1, (1, 1, 40, 40, 40), generator=g, dtype=torch.int64).to(dtype),  # the whole range
        torch.full((1, 1, 8, 8, 8), info.min, dtype=dtype),                                               # one label, the smallest value
        torch.tensor([info.max, info.min, 0, info.max], dtype=dtype).reshape(1, 1, 1, 1, 4),             # shorter than one vector
        torch.zeros(0, 1, 4, 4, 4, dtype=dtype),                                                         # empty
    ]
    for data in cases:
        expected = torch.unique(data).double()
        assert torch.equal(oracle.unique_labels(data), expected)
        got = hip.unique_labels(data.to(DEV))
        assert got.dtype == torch.float64 and torch.equal(got.cpu(), expected)
    view = cases[1].to(DEV).reshape(-1)[3:]  # not 16-byte aligned: the engine re-aligns
    assert torch.equal(hip.unique_labels(view).cpu(), torch.unique(cases[1].reshape(-1)[3:]).double())


def test_unique_labels_wide_dtypes_keep_the_aten_path_and_full_size(hip):
    data = torch.tensor([5, -7, 5, 100000], dtype=torch.int32, device=DEV)
    assert hip.unique_labels(data).tolist() == [-7.0, 5.0, 100000.0]
    labels = torch.randint(0, 40, (1, 1, 512, 512, 512), dtype=torch.int16, device=DEV)  # the config-5 label map size
    assert torch.equal(hip.unique_labels(labels), torch.unique(labels).double())


@pytest.mark.parametrize("rk", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("rj", [2, 6, 8])
def test_fused_jk_stage_every_k_radius(oracle, hip, rj, rk):
    """The K stage of the fused J + K pass applies its taps in tiers of two radii (<= 4 / 6 / 8; round 4): every radius of
    every tier, next to a small, the usual and the largest J radius — the exact launch bit for bit against the oracle (and the
    same with a non-finite voxel just beyond the K radius: a skipped tap must stay skipped), the fast launch (zero-padded
    taps inside a tier, fused multiply-adds) within float rounding of the exact one."""
    import torchio_amd as tio

    sigmas = (1.2, (rj - 0.5) / 3.0, (rk - 0.5) / 3.0)
    data = _data((2, 1, 21, 37, 64), torch.float32, 70 + rj * 8 + rk)
    taps, radius = _taps(1, [sigmas], 32)
    assert radius == [4, rj, rk]
    cpu, gpu = _both(oracle, hip, "separable_conv3d", (data, taps, radius))
    assert torch.equal(cpu, gpu.cpu())
    # a non-finite voxel: the outputs it reaches are the oracle's, no further (0 * inf would be NaN)
    poisoned = data.clone()
    poisoned[0, 0, 10, 18, 30] = float("inf")
    cpu_p, gpu_p = _both(oracle, hip, "separable_conv3d", (poisoned, taps, radius))
    assert torch.equal(torch.isfinite(cpu_p), torch.isfinite(gpu_p.cpu()))
    assert torch.equal(cpu_p[torch.isfinite(cpu_p)], gpu_p.cpu()[torch.isfinite(cpu_p)])
    previous = tio.get_stencil_precision()
    try:
        tio.set_stencil_precision("exact")
        exact = hip.blur_fused(data.to(DEV), taps.to(DEV), radius)
        tio.set_stencil_precision("fast")
        fast = hip.blur_fused(data.to(DEV), taps.to(DEV), radius)
    finally:
        tio.set_stencil_precision(previous)
    assert exact is not None and fast is not None  # (all three axes active, K = 64: the fused form exists)
    assert torch.equal(exact.cpu(), cpu)
    assert float((exact - fast).abs().max()) <= 2e-6 * float(exact.abs().max())
