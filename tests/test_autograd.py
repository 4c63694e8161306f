"""Autograd through the engine: the reference's transforms are differentiable with respect to
the image data (reference tests/test_noise.py:75-80, docs/concepts/transforms.md:289-299).

* the adjoint launch (``TIO_LINEAR_ADJOINT``) IS the transpose of the forward trilinear resampling: dot-product test
  <A x, g> == <x, A^T g> on random data, with and without elastic field / fill value;
* the transforms backpropagate: Affine, ElasticDeformation, BiasField, Blur, Noise (Gaussian and Rician), Gamma, Flip;
* build container: gradients equal the ones autograd derives through the UNMODIFIED reference.
CPU tests run the oracle; the ``gpu`` test repeats the adjoint and a Compose on the HIP engine.
"""
from __future__ import annotations

import pytest
import torch

import torchio_amd as tio
from parity_harness import use_engine


def _geometry(device, elastic: bool):
    g = torch.Generator().manual_seed(3)
    mapping = torch.tensor([[[0.97, 0.06, -0.03, 1.2], [-0.05, 1.04, 0.02, -1.7], [0.03, -0.02, 0.93, 0.9]]], device=device)
    cp = ((torch.rand(1, 5, 5, 5, 3, generator=g) - 0.5) * 3).to(device) if elastic else None
    return mapping, cp


def _adjoint_identity(engine, device, elastic, with_fill):
    g = torch.Generator().manual_seed(7)
    shape = (2, 2, 20, 18, 24)
    x = torch.rand(shape, generator=g).to(device)
    grad = torch.rand(shape, generator=g).to(device)
    mapping, cp = _geometry(device, elastic)
    fill = torch.zeros(2, device=device) if with_fill else None  # fill 0: the forward stays linear in x
    common = dict(out_shape=shape[2:], mapping=mapping, control_points=cp, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True)
    (forward,) = engine.resample3d([x], interps=["linear"], fills=[fill], **common)
    accumulator = torch.zeros_like(x)
    engine.resample3d([accumulator], interps=["linear_adjoint"], fills=[fill], _adjoint_of=[grad], **common)
    lhs = (forward.double() * grad.double()).sum().item()
    rhs = (x.double() * accumulator.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (lhs, rhs)
    # and autograd uses exactly that launch
    leaf = x.clone().requires_grad_(True)
    (out,) = engine.resample3d([leaf], interps=["linear"], fills=[fill], **common)
    (out * grad).sum().backward()
    assert torch.allclose(leaf.grad, accumulator, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("elastic", [False, True])
@pytest.mark.parametrize("with_fill", [False, True])
def test_adjoint_launch_is_the_transpose_of_the_forward(oracle, elastic, with_fill):
    _adjoint_identity(oracle, "cpu", elastic, with_fill)


def _passthrough_gradient(engine, device):
    """A gated-out element (forward = bit-exact copy) passes its gradient through unchanged, the incoming gradient is left
    untouched, and the other elements still get the adjoint launch (the pass branch used to copy the zeroed
    accumulator OVER the incoming gradient)."""
    g = torch.Generator().manual_seed(17)
    shape = (3, 1, 12, 10, 16)
    x = torch.rand(shape, generator=g).to(device)
    grad = torch.rand(shape, generator=g).to(device)
    grad_before = grad.clone()
    mapping, cp = _geometry(device, True)
    gate = torch.tensor([0, 1, 0], dtype=torch.uint8, device=device)
    common = dict(out_shape=shape[2:], mapping=mapping, control_points=cp, in_spacing=(1, 1, 1), out_spacing=(1, 1, 1), affine_first=True,
                  passthrough=gate)
    leaf = x.clone().requires_grad_(True)
    (out,) = engine.resample3d([leaf], interps=["linear"], fills=[None], **common)
    assert torch.equal(out[1].detach(), x[1])
    (out * grad).sum().backward()
    assert torch.equal(grad, grad_before), "the adjoint launch wrote into the incoming gradient"
    assert torch.equal(leaf.grad[1], grad[1]), "gated-out element: d(copy)/dx is the identity"
    # the other rows: same as a launch without the gate
    del common["passthrough"]
    leaf2 = x.clone().requires_grad_(True)
    (out2,) = engine.resample3d([leaf2], interps=["linear"], fills=[None], **common)
    (out2 * grad).sum().backward()
    assert torch.allclose(leaf.grad[[0, 2]], leaf2.grad[[0, 2]], rtol=1e-5, atol=1e-6)


def test_gated_out_element_backpropagates_the_identity(oracle):
    _passthrough_gradient(oracle, "cpu")


def test_ops_accept_tensors_that_require_grad_when_grad_is_disabled(oracle):
    """Under ``torch.no_grad()`` a tensor that requires grad is just data (`_check` used to refuse it in some ops)."""
    leaf = (torch.rand(1, 1, 8, 8, 8) + 0.2).requires_grad_(True)
    with torch.no_grad():
        assert not oracle.gamma_pow(leaf, 1.3).requires_grad
        assert not oracle.add_noise(leaf, 0.0, 0.1, philox_seed=3).requires_grad
        assert not oracle.flip3d(leaf, axes=(0,)).requires_grad
        assert not oracle.pad3d(leaf, (1, 1, 0, 0, 2, 0)).requires_grad
        assert not oracle.bias_field_apply(leaf, torch.zeros(1, 1, 2, 2, 2)).requires_grad
        taps = torch.tensor([[[0.25, 0.5, 0.25]] * 3])
        assert not oracle.separable_conv3d(leaf, taps, [1, 1, 1]).requires_grad


def _pipeline():
    return tio.Compose([
        tio.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-2, 2)), tio.ElasticDeformation(), tio.BiasField(), tio.Blur(std=(0.5, 1.5)),
        tio.Noise(std=0.1), tio.Gamma(log_gamma=(-0.3, 0.3)), tio.Flip(axes=(0, 2)),
    ])


def test_transforms_backpropagate_to_the_input(oracle):
    data = (torch.rand(1, 14, 12, 16) + 0.2).requires_grad_(True)
    with use_engine(oracle):
        torch.manual_seed(5)
        out = _pipeline()(tio.Subject(t1=tio.ScalarImage(data)))
        loss = (out.t1.data ** 2).sum()
        loss.backward()
    assert data.grad is not None and data.grad.shape == data.shape
    assert torch.isfinite(data.grad).all() and float(data.grad.abs().sum()) > 0


@pytest.mark.parametrize("name", ["noise", "rician", "gamma", "bias", "blur", "flip"])
def test_each_intensity_backward_against_finite_differences(oracle, name):
    transform = {
        "noise": tio.Noise(std=0.2), "rician": tio.Noise(std=0.2, rician=True), "gamma": tio.Gamma(log_gamma=0.4),
        "bias": tio.BiasField(std=0.4), "blur": tio.Blur(std=(1.0, 0.0, 0.7)), "flip": tio.Flip(axes=(1,)),
    }[name]
    base = torch.rand(1, 6, 5, 7, dtype=torch.float64).float() + 0.3
    weights = torch.rand(1, 6, 5, 7)

    def run(tensor):
        with use_engine(oracle):
            torch.manual_seed(11)
            return (transform(tio.Subject(t1=tio.ScalarImage(tensor))).t1.data * weights).sum()

    leaf = base.clone().requires_grad_(True)
    run(leaf).backward()
    probe = [(0, 2, 3, 4), (0, 0, 0, 0), (0, 5, 4, 6), (0, 3, 1, 2)]
    for index in probe:
        step = 1e-2
        plus, minus = base.clone(), base.clone()
        plus[index] += step
        minus[index] -= step
        numeric = float(run(plus) - run(minus)) / (2 * step)
        assert abs(numeric - float(leaf.grad[index])) <= 2e-2 * max(1.0, abs(numeric)), (name, index, numeric, float(leaf.grad[index]))


@pytest.mark.reference
def test_gradients_equal_the_references_autograd(oracle):
    import ref_import

    if not ref_import.reference_available():
        pytest.skip("/root/reference is only present in the build container")
    theirs = ref_import.import_reference()

    def grad_of(module, engine):
        data = (torch.rand(1, 16, 14, 18, generator=torch.Generator().manual_seed(2)) + 0.2).requires_grad_(True)
        pipe = module.Compose([
            module.Affine(degrees=(-10, 10), scales=(0.9, 1.1), translation=(-2, 2)), module.ElasticDeformation(), module.BiasField(),
            module.Blur(std=(0.5, 1.5)), module.Noise(std=0.1), module.Gamma(log_gamma=(-0.3, 0.3)),
        ])
        torch.manual_seed(9)
        if engine is None:
            out = pipe(module.Subject(t1=module.ScalarImage(data)))
        else:
            with use_engine(engine):
                out = pipe(module.Subject(t1=module.ScalarImage(data)))
        (out["t1"].data ** 2).sum().backward()
        return data.grad

    expected, actual = grad_of(theirs, None), grad_of(tio, oracle)
    scale = expected.abs().max().item()
    assert (expected - actual).abs().max().item() <= 1e-4 * scale


@pytest.mark.gpu
def test_adjoint_and_compose_backward_on_the_gpu(hip, oracle):
    for elastic in (False, True):
        _adjoint_identity(hip, "cuda", elastic, True)
    _passthrough_gradient(hip, "cuda")
    data = (torch.rand(1, 40, 36, 44, generator=torch.Generator().manual_seed(4)) + 0.2)
    grads = []
    for device, engine in (("cpu", oracle), ("cuda", hip)):
        leaf = data.clone().to(device).requires_grad_(True)
        with use_engine(engine):
            torch.manual_seed(6)
            out = _pipeline()(tio.Subject(t1=tio.ScalarImage(leaf)))
        (out.t1.data ** 2).sum().backward()
        grads.append(leaf.grad.cpu())
    assert (grads[0] - grads[1]).abs().max().item() <= 1e-4 * grads[0].abs().max().item()
