"""``torch.ops.tio_hip.*`` (csrc/torch_ops.cpp): the PyTorch-ROCm custom-op face of the C ABI.

CPU: the library loads, every op is registered with the documented schema, CPU tensors are refused by the
dispatcher.  GPU: every op returns exactly what the ``ctypes`` engine returns (same entry points underneath), on
the current stream, without synchronising.
"""
from __future__ import annotations

import pytest
import torch

OPS = ("resample3d", "resample3d_adjoint", "separable_conv3d", "bias_field_apply", "add_noise", "gamma_pow", "channel_min", "bspline_prefilter")


def test_library_loads_and_registers_every_op():
    import torchio_amd.torch_ops  # noqa: F401

    for name in OPS:
        schema = str(getattr(torch.ops.tio_hip, name).default._schema)
        assert schema.startswith(f"tio_hip::{name}("), schema
    assert "Tensor?[] fill" in str(torch.ops.tio_hip.resample3d.default._schema)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.tio_hip.gamma_pow(torch.rand(1, 1, 2, 2, 2), torch.tensor(2.0))


@pytest.mark.gpu
def test_custom_ops_equal_the_ctypes_engine(hip):
    import torchio_amd.torch_ops as tops

    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(2, 1, 40, 36, 44, generator=g, device="cuda")
    seg = (torch.rand(2, 1, 40, 36, 44, generator=g, device="cuda") * 5).to(torch.int16)
    mapping = torch.tensor([[[0.98, 0.05, -0.02, 1.5], [-0.04, 1.03, 0.03, -2.0], [0.02, -0.03, 0.95, 0.7]]], device="cuda").repeat(2, 1, 1)
    mapping[1, :, 3] += 1.25
    cp = (torch.rand(2, 5, 5, 5, 3, generator=g, device="cuda") - 0.5) * 4
    fill = torch.tensor([-1.0], device="cuda")
    ours = torch.ops.tio_hip.resample3d([x, seg], [1, 0], mapping, cp, [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [40, 36, 44], True, [fill, None])
    ref = hip.resample3d([x, seg], interps=["linear", "nearest"], mapping=mapping, control_points=cp, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                         out_shape=(40, 36, 44), affine_first=True, fills=[fill, None])
    assert torch.equal(ours[0], ref[0]) and torch.equal(ours[1], ref[1]) and ours[1].dtype == torch.int16

    blurred = tops.gaussian_blur3d(x, [[1.5, 0.0, 0.7], [0.6, 2.0, 0.0]])
    from torchio_amd.transforms.blur import _gaussian_smooth

    assert torch.equal(blurred, _gaussian_smooth(x, [[1.5, 0.0, 0.7], [0.6, 2.0, 0.0]]))

    coarse = torch.randn(2, 1, 6, 6, 6, generator=g, device="cuda") * 0.3
    assert torch.equal(torch.ops.tio_hip.bias_field_apply(x, coarse), hip.bias_field_apply(x, coarse))
    base = torch.randn(x.shape, generator=g, device="cuda")
    mean, std = torch.tensor([0.1, -0.2], device="cuda"), torch.tensor([0.3, 0.05], device="cuda")
    assert torch.equal(torch.ops.tio_hip.add_noise(x, mean, std, False, base), hip.add_noise(x, mean, std, base1=base))
    assert torch.equal(torch.ops.tio_hip.add_noise(x, torch.tensor(0.0), torch.tensor(0.25), False, None, None, 1234),
                       hip.add_noise(x, 0.0, 0.25, philox_seed=1234))
    gamma = torch.tensor([0.8, 1.3], device="cuda")
    assert torch.equal(torch.ops.tio_hip.gamma_pow(x - 0.5, gamma), hip.gamma_pow(x - 0.5, gamma))
    assert torch.equal(torch.ops.tio_hip.channel_min(x), hip.channel_min(x))
    assert torch.equal(torch.ops.tio_hip.bspline_prefilter(x, 3), hip.bspline_prefilter(x, 3))


def test_fake_kernels_give_shapes_and_dtypes_without_a_gpu():
    """``torch.library.register_fake``: the ops trace under FakeTensorMode (what ``torch.compile`` / ``torch.export`` need)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    import torchio_amd.torch_ops  # noqa: F401

    with FakeTensorMode():
        x = torch.empty(2, 3, 8, 9, 10, device="cuda")
        seg = torch.empty(2, 1, 8, 9, 10, device="cuda", dtype=torch.int16)
        mapping = torch.empty(2, 3, 4, device="cuda")
        out = torch.ops.tio_hip.resample3d([x, seg], [1, 0], mapping, None, [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [6, 7, 8], True, [None, None])
        assert [tuple(t.shape) for t in out] == [(2, 3, 6, 7, 8), (2, 1, 6, 7, 8)] and out[1].dtype == torch.int16
        back = torch.ops.tio_hip.resample3d_adjoint(out[0], [8, 9, 10], mapping, None, [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], True)
        assert tuple(back.shape) == (2, 3, 8, 9, 10) and back.dtype == torch.float32
        taps = torch.empty(1, 3, 5, device="cuda")
        assert torch.ops.tio_hip.separable_conv3d(x, taps, [2, 2, 2]).shape == x.shape
        assert torch.ops.tio_hip.bias_field_apply(x, torch.empty(2, 3, 4, 4, 4, device="cuda")).shape == x.shape
        assert torch.ops.tio_hip.add_noise(x, torch.empty((), device="cuda"), torch.empty((), device="cuda")).shape == x.shape
        assert torch.ops.tio_hip.gamma_pow(x, torch.empty(2, device="cuda")).shape == x.shape
        assert tuple(torch.ops.tio_hip.channel_min(x).shape) == (3,)
        assert torch.ops.tio_hip.bspline_prefilter(seg, 3).dtype == torch.float32


def test_backward_passes_are_registered_with_the_dispatcher():
    import torchio_amd.torch_ops  # noqa: F401

    for name in ("resample3d", "resample3d_adjoint", "separable_conv3d", "bias_field_apply", "add_noise", "gamma_pow"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"tio_hip::{name}", "Autograd"), name


@pytest.mark.gpu
def test_custom_ops_pass_opcheck_and_backpropagate_like_the_engine(hip):
    """``torch.library.opcheck`` (schema, autograd registration, fake tensor, AOT dispatch) on the differentiable ops, and
    the gradients the dispatcher computes through ``torch.ops.tio_hip.*`` against the ``ctypes`` engine's."""
    import torchio_amd.torch_ops as tops  # noqa: F401

    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(2, 1, 20, 18, 24, generator=g, device="cuda") + 0.2
    mapping = torch.tensor([[[0.98, 0.05, -0.02, 1.5], [-0.04, 1.03, 0.03, -2.0], [0.02, -0.03, 0.95, 0.7]]], device="cuda").repeat(2, 1, 1)
    cp = (torch.rand(2, 5, 5, 5, 3, generator=g, device="cuda") - 0.5) * 3
    fill = torch.zeros(1, device="cuda")
    spacing = [1.0, 1.0, 1.0]
    leaf = x.clone().requires_grad_(True)
    torch.library.opcheck(torch.ops.tio_hip.resample3d, ([leaf], [1], mapping, cp, spacing, spacing, [20, 18, 24], True, [fill]))
    torch.library.opcheck(torch.ops.tio_hip.gamma_pow, (x.clone().requires_grad_(True), torch.tensor([0.8, 1.3], device="cuda")))
    torch.library.opcheck(torch.ops.tio_hip.bias_field_apply, (x.clone().requires_grad_(True), torch.randn(2, 1, 4, 4, 4, generator=g, device="cuda") * 0.2))
    torch.library.opcheck(torch.ops.tio_hip.add_noise, (x.clone().requires_grad_(True), torch.tensor(0.0), torch.tensor(0.1), False, None, None, 7))

    weights = torch.rand(x.shape, generator=g, device="cuda")
    taps = torch.tensor([[[0.25, 0.5, 0.25]] * 3], device="cuda")
    coarse = torch.randn(2, 1, 4, 4, 4, generator=g, device="cuda") * 0.2
    gamma = torch.tensor([0.8, 1.3], device="cuda")

    def through_ops(t):
        (y,) = torch.ops.tio_hip.resample3d([t], [1], mapping, cp, spacing, spacing, [20, 18, 24], True, [fill])
        y = torch.ops.tio_hip.bias_field_apply(y, coarse)
        y = torch.ops.tio_hip.separable_conv3d(y, taps, [1, 1, 1])
        y = torch.ops.tio_hip.add_noise(y, torch.tensor(0.0), torch.tensor(0.1), False, None, None, 7)
        return torch.ops.tio_hip.gamma_pow(y.abs() + 0.1, gamma)

    def through_engine(t):
        (y,) = hip.resample3d([t], interps=["linear"], mapping=mapping, control_points=cp, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1),
                              out_shape=(20, 18, 24), affine_first=True, fills=[fill])
        y = hip.bias_field_apply(y, coarse)
        y = hip.separable_conv3d(y, taps, [1, 1, 1])
        y = hip.add_noise(y, 0.0, 0.1, philox_seed=7)
        return hip.gamma_pow(y.abs() + 0.1, gamma)

    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    out_a, out_b = through_ops(a), through_engine(b)
    assert torch.equal(out_a.detach(), out_b.detach())
    (out_a * weights).sum().backward()
    (out_b * weights).sum().backward()
    assert a.grad is not None and torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
