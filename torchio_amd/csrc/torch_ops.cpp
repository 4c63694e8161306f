// torch_ops.cpp — PyTorch-ROCm custom-op registration of the C ABI (BASELINE.json north_star:
// "exposed through PyTorch-ROCm custom ops").
//
//   TORCH_LIBRARY(tio_hip, m): resample3d, separable_conv3d, bias_field_apply, add_noise, gamma_pow, channel_min,
//                              bspline_prefilter
//
// Each op is a thin shim: it checks device / dtype / shapes (TORCH_CHECK -> Python RuntimeError), allocates the
// outputs with the caching allocator (inputs are borrowed, nothing is written in place), takes the CURRENT HIP
// stream and calls the extern "C" entry point of libtio_hip.so (include/tio_hip.h) — no synchronisation, no
// .item(), re-entrant.  The kernels live in libtio_hip.so; this file contains no device code and is compiled by
// the host compiler.  Registered for the CUDA (= HIP on ROCm) dispatch key only: CPU tensors get the
// dispatcher's "no kernel for backend CPU" error.  Shape inference (fake / meta kernels) and the backward passes are
// registered with the dispatcher from Python (torchio_amd/torch_ops.py: torch.library.register_fake /
// register_autograd); the one backward that needs its own launch is an op of this library too: resample3d_adjoint.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <vector>

#include "../../include/tio_hip.h"

namespace {

int32_t dtype_code(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return TIO_F32;
    case at::kDouble: return TIO_F64;
    case at::kHalf: return TIO_F16;
    case at::kBFloat16: return TIO_BF16;
    case at::kByte: return TIO_U8;
    case at::kChar: return TIO_I8;
    case at::kShort: return TIO_I16;
    case at::kInt: return TIO_I32;
    case at::kLong: return TIO_I64;
    default: TORCH_CHECK(false, "tio_hip: unsupported dtype ", t); return -1;
  }
}

void* current_stream(const at::Tensor& ref) { return static_cast<void*>(at::hip::getCurrentHIPStream(ref.device().index()).stream()); }

void check_status(int status, const char* what) { TORCH_CHECK(status == TIO_OK, what, ": ", tio_last_error()); }

void check_volume(const at::Tensor& x, const char* what) {
  TORCH_CHECK(x.dim() == 5, what, ": expected a (B, C, I, J, K) tensor, got ", x.dim(), " dimensions");
  TORCH_CHECK(x.is_cuda(), what, ": tensor must live on the GPU");
}

const float* opt_f32(const c10::optional<at::Tensor>& t, std::vector<at::Tensor>& keep, const at::Device& device, const char* what) {
  if (!t.has_value() || !t->defined()) return nullptr;
  TORCH_CHECK(t->device() == device, what, ": optional tensor on another device");
  keep.push_back(t->to(at::kFloat).contiguous());
  return keep.back().data_ptr<float>();
}

const uint8_t* opt_flags(const c10::optional<at::Tensor>& t, std::vector<at::Tensor>& keep, const at::Device& device, int64_t batch, const char* what) {
  if (!t.has_value() || !t->defined()) return nullptr;
  TORCH_CHECK(t->device() == device && t->numel() == batch, what, ": flags must be B values on the data's device");
  keep.push_back((t->scalar_type() == at::kBool ? t->to(at::kByte) : t->ne(0).to(at::kByte)).contiguous());
  return keep.back().data_ptr<uint8_t>();
}

// resample3d(Tensor[] images, int[] modes, Tensor mapping(B|1,3,4), Tensor? cp(B|1,ni,nj,nk,3), float[3] in_spacing,
//            float[3] out_spacing, int[3] out_shape, bool affine_first, Tensor?[] fill (C floats each or None),
//            Tensor? passthrough(B), int precision) -> Tensor[]
std::vector<at::Tensor> resample3d(at::TensorList images, at::IntArrayRef modes, const at::Tensor& mapping,
                                   const c10::optional<at::Tensor>& control_points, at::ArrayRef<double> in_spacing,
                                   at::ArrayRef<double> out_spacing, at::IntArrayRef out_shape, bool affine_first,
                                   const c10::List<c10::optional<at::Tensor>>& fill, const c10::optional<at::Tensor>& passthrough,
                                   int64_t precision) {
  TORCH_CHECK(!images.empty() && images.size() <= TIO_MAX_IMAGES, "resample3d: 1..", TIO_MAX_IMAGES, " images per call");
  TORCH_CHECK(modes.size() == images.size() && fill.size() == images.size(), "resample3d: one mode and one fill entry per image");
  TORCH_CHECK(in_spacing.size() == 3 && out_spacing.size() == 3 && out_shape.size() == 3, "resample3d: spacings and out_shape have 3 entries");
  const at::Tensor& first = images[0];
  check_volume(first, "resample3d");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(first.device());  // kernels, scratch and the stream below belong to the DATA's device
  const at::Device device = first.device();
  std::vector<at::Tensor> keep;
  tio_resample_geom geom{};
  geom.batch = static_cast<int32_t>(first.size(0));
  for (int d = 0; d < 3; d++) {
    geom.in_shape[d] = static_cast<int32_t>(first.size(2 + d));
    geom.out_shape[d] = static_cast<int32_t>(out_shape[d]);
    geom.in_spacing[d] = static_cast<float>(in_spacing[d]);
    geom.out_spacing[d] = static_cast<float>(out_spacing[d]);
  }
  geom.affine_first = affine_first ? 1 : 0;
  TORCH_CHECK(mapping.device() == device && mapping.dim() == 3 && mapping.size(1) == 3 && mapping.size(2) == 4 &&
                  (mapping.size(0) == 1 || mapping.size(0) == geom.batch),
              "resample3d: mapping must be (B|1, 3, 4) on the data's device");
  keep.push_back(mapping.to(at::kFloat).contiguous());
  geom.mapping_dev = keep.back().data_ptr<float>();
  geom.mapping_batched = mapping.size(0) > 1 ? 1 : 0;
  if (control_points.has_value() && control_points->defined()) {
    const at::Tensor& cp = *control_points;
    TORCH_CHECK(cp.device() == device && cp.dim() == 5 && cp.size(4) == 3 && (cp.size(0) == 1 || cp.size(0) == geom.batch),
                "resample3d: control points must be (B|1, ni, nj, nk, 3) on the data's device");
    keep.push_back(cp.to(at::kFloat).contiguous());
    geom.control_points_dev = keep.back().data_ptr<float>();
    geom.cp_batched = cp.size(0) > 1 ? 1 : 0;
    for (int d = 0; d < 3; d++) geom.cp_shape[d] = static_cast<int32_t>(cp.size(1 + d));
  }
  geom.passthrough_dev = opt_flags(passthrough, keep, device, geom.batch, "resample3d");
  geom.precision = static_cast<int32_t>(precision);

  std::vector<tio_resample_image> descs(images.size());
  std::vector<at::Tensor> outputs;
  outputs.reserve(images.size());
  for (size_t i = 0; i < images.size(); i++) {
    check_volume(images[i], "resample3d");
    TORCH_CHECK(images[i].device() == device && images[i].size(0) == geom.batch, "resample3d: images must share device and batch size");
    keep.push_back(images[i].contiguous());
    const at::Tensor& in = keep.back();
    outputs.push_back(at::empty({in.size(0), in.size(1), out_shape[0], out_shape[1], out_shape[2]}, in.options()));
    tio_resample_image& d = descs[i];
    d = tio_resample_image{};
    d.in = in.data_ptr();
    d.out = outputs.back().data_ptr();
    d.channels = static_cast<int32_t>(in.size(1));
    d.dtype = dtype_code(in.scalar_type());
    d.interp = static_cast<int32_t>(modes[i]);
    TORCH_CHECK(d.interp == TIO_NEAREST || d.interp == TIO_LINEAR || TIO_BSPLINE_ORDER(d.interp) != 0,
                "resample3d: modes are 0 (nearest), 1 (linear), 4 / 5 (quadratic / cubic B-spline over bspline_prefilter's coefficients), 6 ... 9 (B-spline orders 4 ... 7)");
    const c10::optional<at::Tensor> f = fill.get(i);
    if (f.has_value() && f->defined()) TORCH_CHECK(f->numel() == in.size(1), "resample3d: a fill tensor holds one value per channel");
    d.fill_dev = opt_f32(f, keep, device, "resample3d");
  }
  check_status(tio_resample3d(&geom, static_cast<int32_t>(descs.size()), descs.data(), current_stream(first)), "tio_resample3d");
  return outputs;
}

// resample3d_adjoint(Tensor grad(B,C,*out_shape) f32, int[3] in_shape, mapping, cp, spacings, affine_first, Tensor? fill(C), passthrough)
//   -> Tensor (B, C, *in_shape) f32: the TRANSPOSE of the trilinear resample3d of one image (TIO_LINEAR_ADJOINT): every output
//   voxel adds grad * w_t to its in-bounds taps; voxels that took the fill value carry no gradient; gated-out elements pass
//   their gradient through.  The backward of resample3d (torchio_amd/torch_ops.py registers it with the dispatcher).
at::Tensor resample3d_adjoint(const at::Tensor& grad, at::IntArrayRef in_shape, const at::Tensor& mapping,
                              const c10::optional<at::Tensor>& control_points, at::ArrayRef<double> in_spacing,
                              at::ArrayRef<double> out_spacing, bool affine_first, const c10::optional<at::Tensor>& fill,
                              const c10::optional<at::Tensor>& passthrough) {
  check_volume(grad, "resample3d_adjoint");
  TORCH_CHECK(in_shape.size() == 3 && in_spacing.size() == 3 && out_spacing.size() == 3, "resample3d_adjoint: shapes and spacings have 3 entries");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(grad.device());
  const at::Device device = grad.device();
  std::vector<at::Tensor> keep;
  const at::Tensor g = grad.to(at::kFloat).contiguous();  // (a handle of its own: `keep` reallocates as it grows)
  at::Tensor accumulator = at::zeros({g.size(0), g.size(1), in_shape[0], in_shape[1], in_shape[2]}, g.options());
  tio_resample_geom geom{};
  geom.batch = static_cast<int32_t>(g.size(0));
  for (int d = 0; d < 3; d++) {
    geom.in_shape[d] = static_cast<int32_t>(in_shape[d]);
    geom.out_shape[d] = static_cast<int32_t>(g.size(2 + d));
    geom.in_spacing[d] = static_cast<float>(in_spacing[d]);
    geom.out_spacing[d] = static_cast<float>(out_spacing[d]);
  }
  geom.affine_first = affine_first ? 1 : 0;
  TORCH_CHECK(mapping.device() == device && mapping.dim() == 3 && mapping.size(1) == 3 && mapping.size(2) == 4 &&
                  (mapping.size(0) == 1 || mapping.size(0) == geom.batch),
              "resample3d_adjoint: mapping must be (B|1, 3, 4) on the data's device");
  keep.push_back(mapping.to(at::kFloat).contiguous());
  geom.mapping_dev = keep.back().data_ptr<float>();
  geom.mapping_batched = mapping.size(0) > 1 ? 1 : 0;
  if (control_points.has_value() && control_points->defined()) {
    const at::Tensor& cp = *control_points;
    TORCH_CHECK(cp.device() == device && cp.dim() == 5 && cp.size(4) == 3 && (cp.size(0) == 1 || cp.size(0) == geom.batch),
                "resample3d_adjoint: control points must be (B|1, ni, nj, nk, 3) on the data's device");
    keep.push_back(cp.to(at::kFloat).contiguous());
    geom.control_points_dev = keep.back().data_ptr<float>();
    geom.cp_batched = cp.size(0) > 1 ? 1 : 0;
    for (int d = 0; d < 3; d++) geom.cp_shape[d] = static_cast<int32_t>(cp.size(1 + d));
  }
  geom.passthrough_dev = opt_flags(passthrough, keep, device, geom.batch, "resample3d_adjoint");
  geom.precision = TIO_PRECISION_EXACT;
  tio_resample_image d{};
  d.in = accumulator.data_ptr();          // written: dL/d(input)
  d.out = const_cast<float*>(g.data_ptr<float>());  // read: the incoming gradient
  d.channels = static_cast<int32_t>(g.size(1));
  d.dtype = TIO_F32;
  d.interp = TIO_LINEAR_ADJOINT;
  if (fill.has_value() && fill->defined()) TORCH_CHECK(fill->numel() == g.size(1), "resample3d_adjoint: a fill tensor holds one value per channel");
  d.fill_dev = opt_f32(fill, keep, device, "resample3d_adjoint");
  check_status(tio_resample3d(&geom, 1, &d, current_stream(grad)), "tio_resample3d (adjoint)");
  return accumulator;
}

// separable_conv3d(Tensor x, Tensor taps(1|B,3,stride), int[3] radius, Tensor? skip(B)) -> Tensor
at::Tensor separable_conv3d(const at::Tensor& x, const at::Tensor& taps, at::IntArrayRef radius, const c10::optional<at::Tensor>& skip) {
  check_volume(x, "separable_conv3d");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  TORCH_CHECK(at::isFloatingType(x.scalar_type()), "separable_conv3d: floating dtype expected");
  TORCH_CHECK(radius.size() == 3, "separable_conv3d: radius has 3 entries");
  TORCH_CHECK(taps.device() == x.device() && taps.dim() == 3 && taps.size(1) == 3 && (taps.size(0) == 1 || taps.size(0) == x.size(0)),
              "separable_conv3d: taps must be (1|B, 3, stride) on the data's device");
  std::vector<at::Tensor> keep;
  const at::Tensor in = x.contiguous(), t = taps.to(at::kFloat).contiguous();
  at::Tensor out = at::empty_like(in);
  int active = 0;
  int32_t shape[3], rad[3];
  for (int d = 0; d < 3; d++) {
    shape[d] = static_cast<int32_t>(in.size(2 + d));
    rad[d] = static_cast<int32_t>(radius[d]);
    active += rad[d] > 0;
  }
  at::Tensor tmp;
  if (active > 1) tmp = at::empty({2, in.numel()}, in.options().dtype(at::kFloat));
  check_status(tio_separable_conv3d(in.data_ptr(), out.data_ptr(), tmp.defined() ? tmp.data_ptr() : nullptr, dtype_code(in.scalar_type()),
                                    static_cast<int32_t>(in.size(0)), static_cast<int32_t>(in.size(1)), shape, t.data_ptr<float>(),
                                    (taps.size(0) > 1) ? 1 : 0, static_cast<int32_t>(t.size(2)), rad,
                                    opt_flags(skip, keep, x.device(), x.size(0), "separable_conv3d"), current_stream(x)),
               "tio_separable_conv3d");
  return out;
}

// bias_field_apply(Tensor x, Tensor coarse(B,C,si,sj,sk), bool divide, Tensor? skip(B)) -> Tensor
at::Tensor bias_field_apply(const at::Tensor& x, const at::Tensor& coarse, bool divide, const c10::optional<at::Tensor>& skip) {
  check_volume(x, "bias_field_apply");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  TORCH_CHECK(coarse.device() == x.device() && coarse.dim() == 5 && coarse.size(0) == x.size(0) && coarse.size(1) == x.size(1),
              "bias_field_apply: coarse field must be (B, C, si, sj, sk) on the data's device");
  std::vector<at::Tensor> keep;
  const at::Tensor in = x.contiguous(), c = coarse.to(at::kFloat).contiguous();
  at::Tensor out = at::empty_like(in);
  int32_t shape[3], cshape[3];
  for (int d = 0; d < 3; d++) { shape[d] = static_cast<int32_t>(in.size(2 + d)); cshape[d] = static_cast<int32_t>(c.size(2 + d)); }
  check_status(tio_bias_field_apply(in.data_ptr(), out.data_ptr(), dtype_code(in.scalar_type()), static_cast<int32_t>(in.size(0)),
                                    static_cast<int32_t>(in.size(1)), shape, c.data_ptr<float>(), cshape, divide ? 1 : 0,
                                    opt_flags(skip, keep, x.device(), x.size(0), "bias_field_apply"), current_stream(x)),
               "tio_bias_field_apply");
  return out;
}

// add_noise(Tensor x, Tensor mean(1|B), Tensor std(1|B), bool rician, Tensor? base, Tensor? base2, int philox_seed, Tensor? keep(B)) -> Tensor
at::Tensor add_noise(const at::Tensor& x, const at::Tensor& mean, const at::Tensor& std_, bool rician, const c10::optional<at::Tensor>& base,
                     const c10::optional<at::Tensor>& base2, int64_t philox_seed, const c10::optional<at::Tensor>& keep_rows) {
  check_volume(x, "add_noise");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  const int64_t batch = x.size(0);
  TORCH_CHECK((mean.numel() == 1 || mean.numel() == batch) && (std_.numel() == 1 || std_.numel() == batch), "add_noise: mean / std hold 1 or B values");
  std::vector<at::Tensor> keep;
  const at::Tensor in = x.contiguous();
  at::Tensor out = at::empty_like(in);
  const bool batched = mean.numel() > 1 || std_.numel() > 1;
  float mean_s = 0.0f, std_s = 0.0f;
  const float *mean_dev = nullptr, *std_dev = nullptr;
  if (batched) {  // device vectors of B values (a scalar partner is expanded on the device: no host round trip)
    keep.push_back(mean.to(x.device(), at::kFloat).expand({batch}).contiguous());
    mean_dev = keep.back().data_ptr<float>();
    keep.push_back(std_.to(x.device(), at::kFloat).expand({batch}).contiguous());
    std_dev = keep.back().data_ptr<float>();
  } else {
    TORCH_CHECK(mean.is_cpu() && std_.is_cpu(), "add_noise: scalar mean / std are host tensors (a device scalar would need a sync)");
    mean_s = mean.item<float>();
    std_s = std_.item<float>();
  }
  const float* b1 = nullptr;
  const float* b2 = nullptr;
  if (base.has_value() && base->defined()) {
    TORCH_CHECK(base->numel() == in.numel(), "add_noise: base draws are shaped like x");
    b1 = opt_f32(base, keep, x.device(), "add_noise");
    if (rician) {
      TORCH_CHECK(base2.has_value() && base2->defined() && base2->numel() == in.numel(), "add_noise: rician needs a second set of draws");
      b2 = opt_f32(base2, keep, x.device(), "add_noise");
    }
  }
  check_status(tio_add_noise(in.data_ptr(), out.data_ptr(), dtype_code(in.scalar_type()), static_cast<int32_t>(batch), in.numel() / batch, mean_s,
                             std_s, mean_dev, std_dev, batched ? 1 : 0, rician ? 1 : 0, b1, b2, static_cast<uint64_t>(philox_seed),
                             opt_flags(keep_rows, keep, x.device(), batch, "add_noise"), current_stream(x)),
               "tio_add_noise");
  return out;
}

// gamma_pow(Tensor x, Tensor gamma(1|B)) -> Tensor
at::Tensor gamma_pow(const at::Tensor& x, const at::Tensor& gamma) {
  check_volume(x, "gamma_pow");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  const int64_t batch = x.size(0);
  TORCH_CHECK(gamma.numel() == 1 || gamma.numel() == batch, "gamma_pow: gamma holds 1 or B values");
  const at::Tensor in = x.contiguous();
  at::Tensor out = at::empty_like(in);
  float gamma_s = 1.0f;
  at::Tensor gamma_dev;
  if (gamma.numel() > 1) {
    gamma_dev = gamma.to(x.device(), at::kFloat).contiguous();
  } else {
    TORCH_CHECK(gamma.is_cpu(), "gamma_pow: a scalar gamma is a host tensor (a device scalar would need a sync)");
    gamma_s = gamma.item<float>();
  }
  check_status(tio_gamma_pow(in.data_ptr(), out.data_ptr(), dtype_code(in.scalar_type()), static_cast<int32_t>(batch), in.numel() / batch, gamma_s,
                             gamma_dev.defined() ? gamma_dev.data_ptr<float>() : nullptr, gamma_dev.defined() ? 1 : 0, current_stream(x)),
               "tio_gamma_pow");
  return out;
}

// channel_min(Tensor x) -> Tensor (C floats, device): per-channel minimum of the FIRST batch element
at::Tensor channel_min(const at::Tensor& x) {
  check_volume(x, "channel_min");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  const at::Tensor in = x.contiguous();
  at::Tensor out = at::empty({in.size(1)}, in.options().dtype(at::kFloat));
  check_status(tio_channel_min(in.data_ptr(), dtype_code(in.scalar_type()), static_cast<int32_t>(in.size(1)),
                               in.size(2) * in.size(3) * in.size(4), out.data_ptr<float>(), current_stream(x)),
               "tio_channel_min");
  return out;
}

// bspline_prefilter(Tensor x, int order) -> Tensor (float32): B-spline coefficients for resample3d's modes 4 / 5
at::Tensor bspline_prefilter(const at::Tensor& x, int64_t order) {
  check_volume(x, "bspline_prefilter");
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());  // kernels, scratch and the stream below belong to the DATA's device
  const at::Tensor in = x.contiguous();
  at::Tensor out = at::empty(in.sizes(), in.options().dtype(at::kFloat));
  const int32_t shape[3] = {static_cast<int32_t>(in.size(2)), static_cast<int32_t>(in.size(3)), static_cast<int32_t>(in.size(4))};
  check_status(tio_bspline_prefilter(in.data_ptr(), out.data_ptr<float>(), dtype_code(in.scalar_type()), in.size(0) * in.size(1), shape,
                                     static_cast<int32_t>(order), current_stream(x)),
               "tio_bspline_prefilter");
  return out;
}

}  // namespace

TORCH_LIBRARY(tio_hip, m) {
  m.def("resample3d(Tensor[] images, int[] modes, Tensor mapping, Tensor? control_points, float[] in_spacing, float[] out_spacing, "
        "int[] out_shape, bool affine_first, Tensor?[] fill, Tensor? passthrough=None, int precision=0) -> Tensor[]");
  m.def("resample3d_adjoint(Tensor grad, int[] in_shape, Tensor mapping, Tensor? control_points, float[] in_spacing, float[] out_spacing, "
        "bool affine_first, Tensor? fill=None, Tensor? passthrough=None) -> Tensor");
  m.def("separable_conv3d(Tensor x, Tensor taps, int[] radius, Tensor? skip=None) -> Tensor");
  m.def("bias_field_apply(Tensor x, Tensor coarse, bool divide=False, Tensor? skip=None) -> Tensor");
  m.def("add_noise(Tensor x, Tensor mean, Tensor std, bool rician=False, Tensor? base=None, Tensor? base2=None, int philox_seed=0, "
        "Tensor? keep=None) -> Tensor");
  m.def("gamma_pow(Tensor x, Tensor gamma) -> Tensor");
  m.def("channel_min(Tensor x) -> Tensor");
  m.def("bspline_prefilter(Tensor x, int order) -> Tensor");
}

TORCH_LIBRARY_IMPL(tio_hip, CUDA, m) {
  m.impl("resample3d", resample3d);
  m.impl("resample3d_adjoint", resample3d_adjoint);
  m.impl("separable_conv3d", separable_conv3d);
  m.impl("bias_field_apply", bias_field_apply);
  m.impl("add_noise", add_noise);
  m.impl("gamma_pow", gamma_pow);
  m.impl("channel_min", channel_min);
  m.impl("bspline_prefilter", bspline_prefilter);
}
