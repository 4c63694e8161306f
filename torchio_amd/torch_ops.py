"""``torch.ops.tio_hip.*`` — the engine as PyTorch-ROCm custom ops.

    import torchio_amd.torch_ops          # loads csrc/libtio_torch_ops.so: TORCH_LIBRARY(tio_hip, ...)
    out, = torch.ops.tio_hip.resample3d([x], [1], mapping, None, [1, 1, 1], [1, 1, 1], x.shape[2:], True, [None])

The ops are registered in C++ (``csrc/torch_ops.cpp``) on top of the C ABI of ``libtio_hip.so``: they take and
return tensors, run on the current HIP stream, allocate their outputs and never synchronise.  The dispatcher sees
them, so they can be called from TorchScript-free C++ front ends and show up in profiler traces by name.  One
composite is added here in Python: ``gaussian_blur3d(x, sigma_vox)``, which builds the reference's normalised taps
(``transforms/blur.py``) and calls ``separable_conv3d``.

The transform classes of this package keep calling the C ABI through ``ctypes`` (``ops.py``): the marshalling cost
per call is the same order (a few microseconds), and the C ABI is the boundary that also serves non-PyTorch callers.

Round 3: the ops are full citizens of the dispatcher — **fake (meta) kernels** give shapes and dtypes without a GPU
(``torch.library.register_fake``: FakeTensor / ``torch.compile`` tracing, ``torch.library.opcheck``), and the **backward
passes are registered with autograd** (``torch.library.register_autograd``) like the reference's compositions of torch
ops are differentiable (reference tests/test_noise.py:75-80): the trilinear resampling through the adjoint launch
(``resample3d_adjoint``), the bias field through the same multiply, noise through the identity (Rician: ``(x + n1) / y``),
gamma through ``g |x|^(g-1)``, the stencil through its transposed kernel (``tio_separable_conv3d_adjoint``, round 6).
"""
from __future__ import annotations

import os

import numpy as np
import torch

_LIBRARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtio_torch_ops.so")
_loaded = False


class TorchOpsError(RuntimeError):
    pass


def load() -> None:
    """Register the ``tio_hip`` op library with the dispatcher (idempotent); fails loudly when it is not built."""
    global _loaded
    if _loaded:
        return
    if not os.path.isfile(_LIBRARY):
        raise TorchOpsError(f"{_LIBRARY} is missing: run `make -C torchio_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)")
    torch.ops.load_library(_LIBRARY)
    _register_dispatcher_extras()
    _loaded = True


_LINEAR, _NEAREST = 1, 0


def _register_dispatcher_extras() -> None:
    """Fake kernels and autograd formulas of the ``tio_hip`` ops (once, after the library is loaded)."""
    lib = torch.library

    # ---- shapes / dtypes without a device -------------------------------------------------------------------------
    @lib.register_fake("tio_hip::resample3d")
    def _(images, modes, mapping, control_points, in_spacing, out_spacing, out_shape, affine_first, fill, passthrough=None, precision=0):
        return [
            image.new_empty((image.shape[0], image.shape[1], *out_shape), dtype=torch.float32 if mode in (4, 5) else image.dtype)
            for image, mode in zip(images, modes, strict=True)
        ]

    @lib.register_fake("tio_hip::resample3d_adjoint")
    def _(grad, in_shape, mapping, control_points, in_spacing, out_spacing, affine_first, fill=None, passthrough=None):
        return grad.new_empty((grad.shape[0], grad.shape[1], *in_shape), dtype=torch.float32)

    for name in ("separable_conv3d", "bias_field_apply", "add_noise", "gamma_pow"):
        lib.register_fake(f"tio_hip::{name}")(lambda x, *args, **kwargs: torch.empty_like(x))

    @lib.register_fake("tio_hip::channel_min")
    def _(x):
        return x.new_empty((x.shape[1],), dtype=torch.float32)

    @lib.register_fake("tio_hip::bspline_prefilter")
    def _(x, order):
        return torch.empty_like(x, dtype=torch.float32)

    # ---- backward passes --------------------------------------------------------------------------------------------
    def resample_setup(ctx, inputs, output):
        images, modes, mapping, control_points, in_spacing, out_spacing, out_shape, affine_first, fill = inputs[:9]
        passthrough = inputs[9] if len(inputs) > 9 else None
        ctx.n_inputs = len(inputs)  # (trailing defaults may or may not be part of the call as the dispatcher records it)
        ctx.geometry = (mapping, control_points, list(in_spacing), list(out_spacing), bool(affine_first), passthrough)
        ctx.modes, ctx.fills = list(modes), list(fill)
        ctx.in_shapes = [tuple(int(v) for v in image.shape[2:]) for image in images]
        ctx.dtypes = [image.dtype for image in images]
        ctx.needs = [image.requires_grad for image in images]

    def resample_backward(ctx, grads):
        mapping, control_points, in_spacing, out_spacing, affine_first, passthrough = ctx.geometry
        image_grads = []
        for n, grad in enumerate(grads):
            if not ctx.needs[n] or grad is None:
                image_grads.append(None)
                continue
            if ctx.modes[n] != _LINEAR or not ctx.dtypes[n].is_floating_point:
                raise RuntimeError("tio_hip::resample3d: only floating-point images resampled trilinearly (mode 1) are differentiable")
            back = torch.ops.tio_hip.resample3d_adjoint(
                grad, list(ctx.in_shapes[n]), mapping, control_points, in_spacing, out_spacing, affine_first, ctx.fills[n], passthrough)
            image_grads.append(back.to(ctx.dtypes[n]))
        # one entry per input in the structure the dispatcher recorded for THIS call (lists for the list arguments, trailing
        # defaults as it chose to pass them): built from its own spec, then the images' slot filled in
        from torch.utils import _pytree  # noqa: PLC0415

        # (`ctx._pt_metadata` is private to torch.library's autograd glue — torch 2.10 here; ADVICE r3: version fragile.  Where it
        # is missing or shaped differently the documented form is returned: one entry per positional input of the schema, the
        # list argument as a list)
        try:
            spec = ctx._pt_metadata.input_spec  # (its last child is the dispatcher's own metadata argument: its slot is added by the caller)
            structure = list(_pytree.tree_unflatten([None] * spec.num_leaves, spec))[:-1]
        except (AttributeError, TypeError, ValueError):
            structure = [None] * ctx.n_inputs
        structure[0] = image_grads
        return tuple(structure)

    lib.register_autograd("tio_hip::resample3d", resample_backward, setup_context=resample_setup)

    def adjoint_setup(ctx, inputs, output):
        grad, in_shape, mapping, control_points, in_spacing, out_spacing, affine_first = inputs[:7]
        fill = inputs[7] if len(inputs) > 7 else None
        passthrough = inputs[8] if len(inputs) > 8 else None
        ctx.n_inputs = len(inputs)
        ctx.args = (mapping, control_points, list(in_spacing), list(out_spacing), bool(affine_first), fill, passthrough)
        ctx.out_shape = [int(v) for v in grad.shape[2:]]

    def adjoint_backward(ctx, grad):  # the adjoint's adjoint is the forward resampling (of a zero-filled, fill-masked field)
        mapping, control_points, in_spacing, out_spacing, affine_first, fill, passthrough = ctx.args
        zero = None if fill is None else torch.zeros_like(fill)
        (forward,) = torch.ops.tio_hip.resample3d([grad.float().contiguous()], [_LINEAR], mapping, control_points, in_spacing, out_spacing, ctx.out_shape,
                                                  affine_first, [zero], passthrough, 0)
        return (forward, *([None] * (ctx.n_inputs - 1)))

    lib.register_autograd("tio_hip::resample3d_adjoint", adjoint_backward, setup_context=adjoint_setup)

    def conv_setup(ctx, inputs, output):
        x, taps, radius = inputs[:3]
        ctx.n_inputs = len(inputs)
        ctx.save_for_backward(x, taps)
        ctx.radius, ctx.skip = list(radius), (inputs[3] if len(inputs) > 3 else None)

    def conv_backward(ctx, grad):
        from . import ops  # noqa: PLC0415

        x, taps = ctx.saved_tensors  # (the transposed clamped stencil: tio_separable_conv3d_adjoint, ABI 16)
        adjoint = ops.hip_engine().separable_conv3d_adjoint(grad, taps.detach(), ctx.radius, skip=ctx.skip)
        return (adjoint.to(x.dtype), *([None] * (ctx.n_inputs - 1)))

    lib.register_autograd("tio_hip::separable_conv3d", conv_backward, setup_context=conv_setup)

    def bias_setup(ctx, inputs, output):
        x, coarse = inputs[:2]
        ctx.n_inputs = len(inputs)
        ctx.coarse, ctx.divide, ctx.skip, ctx.dtype = coarse, bool(inputs[2]) if len(inputs) > 2 else False, (inputs[3] if len(inputs) > 3 else None), x.dtype

    def bias_backward(ctx, grad):  # y = x * f (or x / f): the same multiply applied to the incoming gradient
        return (torch.ops.tio_hip.bias_field_apply(grad.to(ctx.dtype).contiguous(), ctx.coarse, ctx.divide, ctx.skip), *([None] * (ctx.n_inputs - 1)))

    lib.register_autograd("tio_hip::bias_field_apply", bias_backward, setup_context=bias_setup)

    def noise_setup(ctx, inputs, output):
        padded = tuple(inputs) + (None,) * (8 - len(inputs))
        x, mean, std, rician, base, base2, philox_seed, keep = padded
        ctx.n_inputs = len(inputs)
        ctx.rician = bool(rician)
        if ctx.rician:
            if base is None:
                raise RuntimeError("tio_hip::add_noise: the Rician backward needs the explicit draws (`base`); in-kernel Philox draws are not kept")
            ctx.save_for_backward(x, output, mean, std, base)
            ctx.keep = keep

    def noise_backward(ctx, grad):
        if not ctx.rician:  # additive: dy/dx = 1 (gated-out rows are copies: 1 as well)
            return (grad, *([None] * (ctx.n_inputs - 1)))
        x, y, mean, std, base = ctx.saved_tensors
        shape = (-1,) + (1,) * (x.ndim - 1)
        mean_b = mean.reshape(shape) if mean.numel() > 1 else mean
        std_b = std.reshape(shape) if std.numel() > 1 else std
        numerator = x.float() + (mean_b + std_b * base)
        slope = torch.where(y != 0, numerator / y.float(), torch.zeros_like(numerator))
        if ctx.keep is not None:
            slope = torch.where(ctx.keep.bool().reshape(shape), slope, torch.ones_like(slope))
        return ((grad.float() * slope).to(x.dtype), *([None] * (ctx.n_inputs - 1)))

    lib.register_autograd("tio_hip::add_noise", noise_backward, setup_context=noise_setup)

    def gamma_setup(ctx, inputs, output):
        x, gamma = inputs
        ctx.save_for_backward(x, gamma)

    def gamma_backward(ctx, grad):  # y = sign(x) |x|^g: dy/dx = g |x|^(g - 1)
        x, gamma = ctx.saved_tensors
        exponent = gamma.to(x.device, torch.float32).reshape((-1,) + (1,) * (x.ndim - 1)) if gamma.numel() > 1 else gamma.to(x.device, torch.float32)
        return (grad.float() * exponent * x.float().abs().pow(exponent - 1)).to(x.dtype), None

    lib.register_autograd("tio_hip::gamma_pow", gamma_backward, setup_context=gamma_setup)


def gaussian_blur3d(x: torch.Tensor, sigma_vox) -> torch.Tensor:
    """``_gaussian_smooth(x, sigmas)`` (reference blur.py:129-154) on the custom ops: *sigma_vox* is ``(3,)`` or ``(B, 3)`` voxels."""
    from .transforms.blur import _stacked_gaussian_taps  # noqa: PLC0415

    load()
    sigmas = np.asarray(torch.as_tensor(sigma_vox, dtype=torch.float64).cpu().numpy(), dtype=np.float64)
    if np.all(sigmas <= 0):
        return x
    per_element = sigmas.ndim == 2 and not np.all(sigmas == sigmas[0])
    rows = sigmas if per_element else (sigmas[0] if sigmas.ndim == 2 else sigmas)[None]
    taps, radius, skip = _stacked_gaussian_taps(rows, per_element=per_element)
    skip_dev = None if skip is None else torch.from_numpy(skip).to(x.device)
    return torch.ops.tio_hip.separable_conv3d(x, taps.to(x.device), [int(r) for r in radius], skip_dev)


load()
