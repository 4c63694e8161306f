/*
 * tio_hip.h — C ABI of the MI355X (gfx950) 3-D augmentation engine.
 *
 * This is the drop-in boundary.  The reference (TorchIO 2.0.0a2)
 * has no FFI of its own: its transforms call torch.nn.functional directly.  Each
 * entry point below replaces one of the reference's private functional seams;
 * the file:line next to it is the reference code whose arithmetic it reproduces.
 *
 * Conventions
 *   - every pointer named *_dev (and x / y / in / out) is DEVICE memory unless
 *     the comment says HOST; nothing here allocates or frees the caller's tensors, and nothing synchronises in the
 *     steady state (two entry points keep a small scratch array per (device, stream) for themselves — the brick plan
 *     of large tio_resample3d launches, the keys of tio_channel_min — allocated on first use and grown on demand,
 *     which synchronises that one time; do not capture their first call into a graph);
 *   - tensors are dense row-major (B, C, I, J, K) — K fastest — exactly the
 *     layout of ImagesBatch.data (reference src/torchio/data/batch.py:21-50);
 *   - inputs are borrowed, outputs must not alias inputs;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns TIO_OK (0) or a negative tio_status; the failing
 *     call's message is available through tio_last_error() (thread-local);
 *   - re-entrant: no global mutable state besides the thread-local error text and those mutex-guarded scratch tables
 *     (the reference calls transforms from Queue worker threads, src/torchio/data/queue.py:119-123); calls that share
 *     a stream are ordered by it, calls on different streams use different scratch.
 *
 * The CPU restatement in oracle/ exports the same functions with the prefix
 * tio_oracle_ and HOST pointers (stream ignored); it is test infrastructure.
 */
#ifndef TIO_HIP_H
#define TIO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIO_ABI_VERSION 16
#define TIO_MAX_IMAGES 8 /* images resampled per launch with shared coordinates */

typedef enum tio_status {
  TIO_OK = 0,
  TIO_ERR_INVALID_ARGUMENT = -1,
  TIO_ERR_UNSUPPORTED_DTYPE = -2,
  TIO_ERR_LAUNCH = -3,
  TIO_ERR_NO_DEVICE = -4,
  TIO_ERR_UNSUPPORTED_CONFIG = -5 /* valid arguments, but this fused form is not available for them: nothing was launched */
} tio_status;

/* Element types of image tensors (torch dtype ↔ code is fixed here). */
typedef enum tio_dtype {
  TIO_F32 = 0,
  TIO_F64 = 1,
  TIO_F16 = 2,
  TIO_BF16 = 3,
  TIO_U8 = 4,
  TIO_I8 = 5,
  TIO_I16 = 6,
  TIO_I32 = 7,
  TIO_I64 = 8
} tio_dtype;

typedef enum tio_precision {
  TIO_PRECISION_EXACT = 0,
  TIO_PRECISION_FAST = 1,
  TIO_PRECISION_TIGHT = 2 /* ABI 13: the reference's coordinates, taps and fill decisions bit for bit; fused interpolation */
} tio_precision;

typedef enum tio_interp {
  TIO_NEAREST = 0, /* grid_sample(mode="nearest"): nearbyint, half-to-even      */
  TIO_LINEAR = 1,  /* grid_sample(mode="bilinear"): 8-tap trilinear, zero pad   */
  /* label_interpolation="label" with one_hot_label_interpolation="linear", C == 1,
   * no antialias (_resample_label_partial_volume, spatial.py:1275-1389), fused:
   * the one-hot (B, L, I, J, K) tensor is never built.  Per output voxel the value of
   * one-hot channel l is the sum, in ATen's tap order, of the trilinear weights of
   * the in-bounds taps that carry label l; the winner is torch.argmax's (first
   * maximum = smallest label); the voxel is in bounds when the float32 sum over the
   * channels (torch.sum's cascade over the sorted label table) exceeds 0.5, else it
   * receives the pad label.                                                       */
  TIO_LABEL_PV = 2,
  /* The ADJOINT of TIO_LINEAR — the backward pass of the trilinear resampling with respect to the image
   * (what autograd derives from grid_sample in the reference, whose transforms are differentiable:
   * tests/test_noise.py:75-80, docs/concepts/transforms.md:289-299).  For such an image the roles of the two
   * buffers are swapped: `out` is READ, (B, C, Io, Jo, Ko) float32 = dL/d(output); `in` is ACCUMULATED INTO
   * (despite the const), (B, C, I, J, K) float32, zero-initialised by the caller: every output voxel adds
   * g * w_t to its in-bounds taps (hardware float atomics: the order of the additions is not defined).  A
   * non-NULL fill_dev means the forward pass used a fill value: voxels whose in-bounds weight is <= 0.5 took the
   * fill and carry no gradient.  Same geometry struct, same coordinate arithmetic as the forward launch.     */
  TIO_LINEAR_ADJOINT = 3,
  /* B-spline interpolation of order 2 / 3 — the reference's orders >= 2 go to torch-interpol (spatial.py:1734-1761:
   * interpol.grid_pull(data.float(), grid, interpolation=order, bound="dct2", extrapolate=False, prefilter=True)).
   * Here `in` holds the B-spline COEFFICIENTS of the image (float32, tio_bspline_prefilter), `out` is float32 (the
   * caller casts back like `.to(data.dtype)`), and each output voxel is the (order + 1)^3-tap sum of the basis weights at
   * its voxel coordinate — the coordinate the reference hands to grid_pull, i.e. BEFORE grid_sample's normalise /
   * un-normalise round trip — with half-sample-symmetric ("dct2") index reflection, and zero wherever a coordinate lies
   * outside (-0.05, S - 1 + 0.05) (grid_pull's extrapolate=False mask); fill_dev is ignored, as the reference ignores
   * its fill value on this road.  torch-interpol is not vendored by the reference (pyproject.toml:46) and absent from
   * the build image: the arithmetic follows its published algorithm, pinned against scipy.ndimage (mode="reflect",
   * the same extension) instead of against the package itself — "parity unpinned" in DESIGN.md.                      */
  TIO_QUADRATIC = 4,
  TIO_CUBIC = 5,
  /* Orders 4 - 7 ("fourth" ... "seventh", ABI 10): the same road — prefilter (2 / 2 / 3 / 3 poles), then the (order + 1)^3
   * taps.  The basis weights of these orders come from the Cox - de Boor recursion of the uniform B-spline, evaluated in
   * float64 and rounded to float32 (every term positive: no cancellation; the truncated-power form loses four digits at
   * order 7), identically in the oracle and on the device.  Orders 4 and 5 are pinned against scipy.ndimage like 2 and 3;
   * scipy stops at 5, so 6 and 7 are held to what defines them: the interpolation property, partition of unity, exact
   * reproduction of polynomials up to the order, and a float64 numpy reference built from scipy.interpolate.BSpline
   * (tests/test_bspline.py).  Parity with the reference (torch-interpol) stays unpinned, as for orders 2 and 3.          */
  TIO_BSPLINE4 = 6,
  TIO_BSPLINE5 = 7,
  TIO_BSPLINE6 = 8,
  TIO_BSPLINE7 = 9
} tio_interp;
/* B-spline order of an interpolation code (0 for the others) */
#define TIO_BSPLINE_ORDER(interp) ((interp) == TIO_QUADRATIC ? 2 : (interp) == TIO_CUBIC ? 3 : ((interp) >= TIO_BSPLINE4 && (interp) <= TIO_BSPLINE7) ? (interp) - 2 : 0)

/* ------------------------------------------------------------------------ */
/* Fused spatial resampling                                                  */
/* ------------------------------------------------------------------------ */

/*
 * Geometry shared by every image of one resampling launch.  Replaces the
 * materialised (I,J,K,3) sampling grid of the reference:
 *   _build_sampling_grid            spatial.py:1504-1579
 *   _output_voxel_coordinates       spatial.py:1604-1613
 *   _apply_voxel_mapping            spatial.py:1616-1624   ([c,1] @ M^T, float32)
 *   _upsample_displacement_field    spatial.py:2171-2189   (trilinear, align_corners)
 *   _voxel_coordinates_to_grid      spatial.py:1627-1648   (g = 2 v / max(S-1,1) - 1)
 *   ATen grid_sampler un-normalise  ((g + 1) / 2) * (S - 1)
 *   per-instance variant            spatial.py:1881-1918   (B matrices / fields, no
 *                                                           (B,I,J,K,3) stack)
 */
typedef struct tio_resample_geom {
  int32_t batch;        /* B                                                     */
  int32_t in_shape[3];  /* input  (I, J, K)                                      */
  int32_t out_shape[3]; /* output (I, J, K)                                      */
  int32_t affine_first; /* 1: v = M c + d / in_spacing ; 0: v = M (c + d / out_spacing) */
  /* Output-voxel → input-voxel mapping, float32 rows [m00 m01 m02 m03 | m10 ...]
   * (the first three rows of spatial.py:1582-1601's matrix, already cast to f32).
   * mapping_batched = 0: one 3x4 shared by the batch; 1: B matrices.            */
  const float* mapping_dev;
  int32_t mapping_batched;
  /* Elastic control points in mm, (n, ni, nj, nk, 3) float32, n = B if
   * cp_batched else 1; NULL = no elastic component.                            */
  const float* control_points_dev;
  int32_t cp_batched;
  int32_t cp_shape[3];
  /* Optional per-element flags (B bytes each, NULL = all zero):
   *   cp_skip[b]     != 0 → element b has no elastic component
   *                         (control_points None, spatial.py:1542-1543);
   *   passthrough[b] != 0 → element b is copied bit-exactly (gated-out rows,
   *                         spatial.py:1078-1107); needs in_shape == out_shape. */
  const uint8_t* cp_skip_dev;
  const uint8_t* passthrough_dev;
  float in_spacing[3];  /* AffineMatrix.spacing of the input grid, as float32    */
  float out_spacing[3]; /* ... of the output grid                                */
  /* Shape whose (S - 1) normalises the voxel coordinates before grid_sample un-normalises
   * them with the image's own (S - 1).  The reference builds ONE grid from the first selected
   * image and samples every image with it (spatial.py:1136-1191, 1704), so an image of another
   * shape (Resample("t1") on a multi-resolution subject) sees g = 2 v / (S_first - 1) - 1 but
   * x = ((g + 1) / 2) (S_own - 1).  All zeros = in_shape (the usual case: every image shares it). */
  int32_t norm_shape[3];
  /* tio_precision.  TIO_PRECISION_EXACT (0, the default): the reference's float32 operation
   * sequence, bit for bit.  TIO_PRECISION_FAST: the float32 trilinear images of the call may skip
   * the normalise / un-normalise round trip of the coordinates and interpolate with nested fma
   * lerps — same interpolant, different rounding: within ~1e-5 absolute of the exact result on
   * unit-range data (the north_star bar for intensities is 1e-4 relative).  Nearest-neighbour
   * images WITHOUT a fill rule (label maps) are bit-identical to the reference in either mode
   * and do not hold the float images back (their own kernel: the index of a voxel only
   * depends on its coordinate's rounding, and coordinates within rounding error of a
   * half-integer are re-evaluated with the exact chain); any other image in the call — a
   * nearest image with a fill rule, TIO_LABEL_PV, another dtype — keeps the whole call exact.
   * TIO_PRECISION_TIGHT (ABI 13): every sampling coordinate is the reference's float32 value bit for bit (MKL's FMA
   * order, ATen's lerp nesting, the normalise / un-normalise round trip), hence the same eight taps, the same weights
   * and the same `mask > 0.5` decisions; only the interpolation of the float32 trilinear images is fused (three nested
   * fma lerps instead of ATen's eight weighted taps): the result differs from the reference by the rounding of seven
   * fused multiply-adds (~1e-7 of the taps) and meets |d| <= 1e-4 max(|ref|, 1e-3 range) PER VOXEL on white noise, which
   * TIO_PRECISION_FAST cannot (one ulp of a coordinate is already more).  Launches the lean exact-coordinate kernel does
   * not take (small launches, other dtypes, non-unit spacing with control points) run the exact kernels. */
  int32_t precision;
  /* Optional (NULL / 0 = the call plans for itself): a brick plan made AHEAD of the call by tio_resample3d_plan from a
   * geometry with exactly these fields (ABI 11).  Large launches of 16^3 bricks start from a plan — one descriptor per
   * brick, written by a small kernel that reads only the geometry above (mapping, control points, flags), never the
   * images — and that kernel plus the gap behind it sit on the critical path of the call (~8 - 23 us + ~5 us on the
   * bench launch).  A caller that knows the geometry before the data is ready (a Compose that has drawn every child's
   * parameters) can have the plan made on another stream while earlier work runs.  The plan must stay untouched until
   * the call's kernels have finished; a call that takes another road than the one the plan was made for ignores it. */
  const void* plan_dev;
  int64_t plan_bytes;
  /* TIO_GEOM_* bits (ABI 14; 0 = none).  Hints of a caller who holds the mappings on the host, about SPEED only: every road
   * computes the same values in the exact modes and stays within its mode's tolerance otherwise.
   * TIO_GEOM_LARGE_BOXES: the input boxes of SOME 16^3 output bricks are expected to exceed the staging tile of the planned
   * roads (an element rotated beyond ~12 degrees about all three axes, a strong zoom).  Without the hint such a brick samples
   * voxel by voxel from global memory (3.5 x the time; fine for the odd brick).  With it (ABI 15) the exact-coordinate lean
   * kernels — TIO_PRECISION_EXACT / TIGHT, large float32 launches — have the planner bound such bricks again in halves /
   * quarters of their planes and list them; a second kernel behind the first walks the list and stages each brick in two or
   * four passes (~6 - 15 us per launch when the list is short: why it is a hint).  FAST launches take the brick kernels with
   * in-kernel boxes, which split such a brick likewise (resample_tile.hpp).
   * TIO_GEOM_MOSTLY_LARGE_BOXES (ABI 15): MOST bricks are expected to — the pass logic then runs in every block of ONE launch
   * (no list, no second kernel; its one-pass bricks cost 2 - 7 % more than without the hint). */
  int32_t flags;
} tio_resample_geom;

#define TIO_GEOM_LARGE_BOXES 1
#define TIO_GEOM_MOSTLY_LARGE_BOXES 2

/* One image tensor resampled with the shared geometry
 * (_resample_image_batch, spatial.py:1194-1272; _sample_batch_grid_sample,
 * spatial.py:1695-1731). */
typedef struct tio_resample_image {
  const void* in;       /* (B, C, I, J, K)                                       */
  void* out;            /* (B, C, Io, Jo, Ko), same dtype                        */
  int32_t channels;     /* C                                                     */
  int32_t dtype;        /* tio_dtype; computed in float32, cast back like
                           `.float()` / `.to(data.dtype)` (spatial.py:1708,1731) */
  int32_t interp;       /* tio_interp                                            */
  /* Per-channel fill (C floats, device).  NULL = reference's "fill is scalar 0"
   * branch (no mask step, spatial.py:2075-2076).  Non-NULL = trilinear in-bounds
   * weight mask, out = mask > 0.5 ? sampled : fill[c]  (spatial.py:1719-1728).  */
  const float* fill_dev;
  /* TIO_LABEL_PV only (ignored otherwise; requires channels == 1).
   *   labels_dev / n_labels: torch.unique(data) of the WHOLE batch tensor as
   *     ascending float64 (spatial.py:1360).  Only the POSITION of a label in this
   *     table matters, and only through the rounding order of the reference's
   *     channel sum (ATen's cascade_sum dumps its accumulator every 16 channels);
   *     NULL / 0 = plain sequential sum in ascending label order, which is what the
   *     cascade does for fewer than 16 distinct labels.
   *   pad_label: default_pad_label, cast to the image dtype like
   *     torch.full_like(resampled, default_pad_label)  (spatial.py:1380-1384).  */
  const double* labels_dev;
  int32_t n_labels;
  double pad_label;
  /* Optional (NULL = not wanted): C floats on the device that receive the per-channel minimum of the FIRST batch
   * element of `out` — what the NEXT transform's default_pad_value="minimum" will ask for (spatial.py:2054-2060,
   * 2094-2095: `tensor[0, c].min()` of the data it is handed).  Large TIO_PRECISION_FAST launches fold it into their
   * stores (no extra pass over the volume); every other launch runs tio_channel_min on `out` before returning.
   * NaN propagates like torch.min.  Ignored for TIO_LINEAR_ADJOINT. */
  float* out_min_dev;
} tio_resample_image;

/* Bytes of the brick plan tio_resample3d would make for this geometry when every image of the call is a float32
 * trilinear image (0: such a call takes a road without a plan — small launches, K not a multiple of 4, ...);
 * geom->plan_dev / plan_bytes are ignored.  Host-only: nothing is enqueued. */
int64_t tio_resample3d_plan_bytes(const tio_resample_geom* geom);
/* Enqueue the planning kernel for this geometry on `stream`: plan_dev (>= tio_resample3d_plan_bytes(geom) bytes,
 * 16-byte aligned) is then handed to tio_resample3d in geom->plan_dev, on any stream ordered behind this one. */
int tio_resample3d_plan(const tio_resample_geom* geom, void* plan_dev, int64_t plan_bytes, void* stream);

/* Resample n_images image tensors through ONE coordinate computation. */
int tio_resample3d(const tio_resample_geom* geom, int32_t n_images,
                   const tio_resample_image* images, void* stream);

/* Per-channel minimum of the FIRST batch element, kept on the device (the
 * reference's default_pad_value="minimum": spatial.py:2054-2060,2094-2095 —
 * there a host .item() per channel; here out_dev[c] feeds fill_dev directly).
 * x is (B, C, n_spatial); out_dev receives C floats. */
int tio_channel_min(const void* x, int32_t dtype, int32_t channels,
                    int64_t n_spatial, float* out_dev, void* stream);

/* ------------------------------------------------------------------------ */
/* Intensity path                                                            */
/* ------------------------------------------------------------------------ */

/*
 * Separable 3-axis cross-correlation with replicate padding — the stencil under
 * Blur (blur.py:157-252: F.pad(mode="replicate") + F.conv3d per axis, order
 * I, J, K) and the antialias filter (spatial.py:1980-2031).
 *   taps_dev: (n, 3, tap_stride) float32, n = B if taps_batched else 1; for
 *             axis a the 2*radius[a]+1 centred taps start at offset 0 (already
 *             normalised / zero-extended / delta for sigma = 0 exactly as
 *             blur.py:179-183 and blur.py:292-328 build them);
 *   radius:   HOST int32[3]; radius[a] = 0 skips axis a (sigma <= 0, blur.py:177);
 *   skip_dev: optional B bytes; non-zero rows are copied bit-exactly
 *             (blur.py:249-251).
 *   tmp:      scratch with the size of y (may be NULL when at most one axis is
 *             active).
 * Computes in float32 (blur.py:173 `.float()`), stores dtype (blur.py:204).
 */
int tio_separable_conv3d(const void* x, void* y, void* tmp, int32_t dtype,
                         int32_t batch, int32_t channels, const int32_t shape[3],
                         const float* taps_dev, int32_t taps_batched,
                         int32_t tap_stride, const int32_t radius[3],
                         const uint8_t* skip_dev, void* stream);

/*
 * Backward of tio_separable_conv3d with respect to x (ABI 16): gx = A_I^T A_J^T A_K^T gy, where A_a is the
 * replicate-padded correlation along axis a — the transpose of a CLAMPED stencil folds the taps that were
 * clamped onto a border voxel back onto it.  What autograd computes through the reference's
 * F.pad(mode="replicate") + grouped conv3d per axis (blur.py:157-252), one gather kernel per active axis.
 *   gy, gx: (B, C, I, J, K) float32 (the reference computes the blur, hence its backward, in float32:
 *           blur.py:173); taps_dev / taps_batched / tap_stride / radius / skip_dev as in the forward call;
 *           rows with skip_dev[b] != 0 pass their gradient through (the forward copied them);
 *   tmp:    scratch with the size of gx (may be NULL when at most one axis is active).
 */
int tio_separable_conv3d_adjoint(const float* gy, float* gx, float* tmp, int32_t batch,
                                 int32_t channels, const int32_t shape[3], const float* taps_dev,
                                 int32_t taps_batched, int32_t tap_stride, const int32_t radius[3],
                                 const uint8_t* skip_dev, void* stream);

/*
 * Blur with its neighbours folded in: y = Noise(Blur(BiasField(x))) in the separable passes
 * of tio_separable_conv3d — the bias field multiplies every row on its way into the first
 * (I) pass, the noise is added to every row the last (fused J+K) pass stores, so the two
 * elementwise transforms cost no extra trip through HBM.  Values are bit-identical to
 * tio_bias_field_apply -> tio_separable_conv3d -> tio_add_noise (same float32 operations in
 * the same order; tests compare the two).  Replaces, for a Compose that holds them back to
 * back, bias_field.py:99-132 + blur.py:76-252 + noise.py:98-123.
 *   bias_coarse_dev == NULL: no bias stage;  noise_on == 0: no noise stage (fast-mode
 *   Philox draws only, as tio_add_noise with base1_dev == NULL, not rician);
 *   tmp: scratch of 2 x size(y) floats.
 * Returns TIO_ERR_UNSUPPORTED_CONFIG (and launches nothing) unless: float32, 16-byte
 * aligned, all three radii in 1..16 (K: 1..8), K <= 256 and K % 4 == 0 — the caller then
 * runs the three entry points one after the other.
 */
int tio_blur_fused(const void* x, void* y, void* tmp, int32_t dtype, int32_t batch,
                   int32_t channels, const int32_t shape[3], const float* taps_dev,
                   int32_t taps_batched, int32_t tap_stride, const int32_t radius[3],
                   const float* bias_coarse_dev, const int32_t bias_coarse_shape[3],
                   int32_t noise_on, float noise_mean, float noise_std,
                   const float* noise_mean_dev, const float* noise_std_dev,
                   int32_t noise_batched, uint64_t philox_seed, const float* noise_base_dev,
                   int32_t fast_math, void* stream);
/* noise_on (ABI 12): 0 = no noise stage; 1 = in-kernel Philox draws (philox_seed), as tio_add_noise with base1_dev == NULL;
 * 2 = explicit draws: noise_base_dev holds one float32 normal draw per element, laid out like x (16-byte aligned) — as
 * tio_add_noise with base1_dev.  This is how the REFERENCE's noise stream (noise.py:108-116: one seeded CPU generator,
 * reproduced on the device by tio_mt19937_randn_device) rides on the stencil's stores: the draws are made ahead on another
 * stream, the sum x + (mean + std z) costs no pass of its own.  Same float32 operations as tio_add_noise. */
/* fast_math (ABI 8): 0 = every tap is `acc = acc + w * v` with two roundings, the reference's accumulation (bit-identical
 * to tio_separable_conv3d); 1 = fused multiply-adds in the register-window passes (radii <= 8): one rounding per tap,
 * results within float rounding of the exact ones (~1e-7 relative; the contract for intensities is 1e-4), and a J+K pass
 * that is bound by vector instructions ~25 % shorter.  The counterpart of tio_resample_geom.precision for the stencil. */

/*
 * BiasField: y = x * exp(trilinear_upsample(coarse))   (or x / ... when divide)
 *   bias_field.py:296-341 (_generate_bias_field), :201-255 (_apply_bias_per_element),
 *   :130 / :196 (multiply / divide).  coarse_dev is (B, C, si, sj, sk) float32 —
 *   the seeded CPU-generator draw of bias_field.py:321-330 stays on the host
 *   side of this boundary.  skip_dev: optional B bytes, rows copied bit-exactly
 *   (std == 0 rows, bias_field.py:247-253).
 */
int tio_bias_field_apply(const void* x, void* y, int32_t dtype, int32_t batch,
                         int32_t channels, const int32_t shape[3],
                         const float* coarse_dev, const int32_t coarse_shape[3],
                         int32_t divide, const uint8_t* skip_dev, void* stream);

/*
 * Noise: y = x + (mean + std * z)            (noise.py:98-123, :166-178)
 *        rician: y = sqrt((x + n1)^2 + n2^2), n_i = mean + std * z_i
 *   mean_dev / std_dev: per-element (B floats) when params_batched, else the
 *   scalars mean / std are used.
 *   base1_dev / base2_dev: standard-normal draws shaped like x.  Parity mode: the
 *   caller fills them from the reference's seeded CPU generator (noise.py:177).
 *   Fast mode (base1_dev == NULL): drawn in-kernel from Philox4x32-10 keyed by
 *   (philox_seed, element index) + Box-Muller; base2 uses stream id 1.
 *   keep_dev: optional B bytes; rows with keep == 0 are copied bit-exactly
 *   (noise.py:126-146).
 */
int tio_add_noise(const void* x, void* y, int32_t dtype, int32_t batch,
                  int64_t n_per_element, float mean, float std,
                  const float* mean_dev, const float* std_dev,
                  int32_t params_batched, int32_t rician, const float* base1_dev,
                  const float* base2_dev, uint64_t philox_seed,
                  const uint8_t* keep_dev, void* stream);

/* Fill out_dev[0..n) with the same standard normals the fast noise mode draws
 * (stream_id 0 → base1, 1 → base2). */
int tio_philox_normal(float* out_dev, int64_t n, uint64_t philox_seed,
                      int32_t stream_id, void* stream);

/*
 * Gamma: y = sign(x) * |x| ^ gamma            (gamma.py:80-91, :133-142)
 *   gamma_dev: per-element exponents (B floats) when params_batched, else the
 *   scalar gamma (= exp(log_gamma), gamma.py:103-120).
 */
int tio_gamma_pow(const void* x, void* y, int32_t dtype, int32_t batch,
                  int64_t n_per_element, float gamma, const float* gamma_dev,
                  int32_t params_batched, void* stream);

/* ------------------------------------------------------------------------ */
/* F.interpolate users: Resize, Anisotropy                */
/* ------------------------------------------------------------------------ */

/*
 * B-spline coefficients of a (B * C) stack of volumes for tio_resample3d's TIO_QUADRATIC / TIO_CUBIC images: the
 * recursive prefilter of order `order` (2: pole sqrt(8) - 3, 3: pole sqrt(3) - 2; 4 and 5: two poles, 6 and 7: three — the roots
 * inside the unit circle of the order's B-spline polynomial) along I, J, K, one pole after the other, with the
 * half-sample-symmetric ("dct2") boundary — what grid_pull(prefilter=True, bound="dct2") applies before sampling
 * (spatial.py:1753-1760).  x has `dtype`, y is float32 of the same shape.
 */
int tio_bspline_prefilter(const void* x, float* y, int32_t dtype, int64_t n_batch_channels, const int32_t shape[3],
                          int32_t order, void* stream);

/*
 * F.interpolate(x.float(), size=out_shape, mode=...).to(x.dtype) on a dense
 * (N, I, J, K) stack of volumes (N = B * C):
 *   mode TIO_NEAREST: the legacy "nearest" — src = min(floor(dst * float(in) / float(out)), in - 1)
 *                     (resize.py:70-76, anisotropy.py:380-384), an element move for every dtype;
 *   mode TIO_LINEAR:  "trilinear" with align_corners=True (resize.py:70-76,
 *                     anisotropy.py:386-391): ATen's source index / lambda per axis and the
 *                     K-, J-, I-nested fma(t0, w0, t1 * w1), computed in float32.
 */
int tio_interpolate3d(const void* x, void* y, int32_t dtype, int64_t n_batch_channels,
                      const int32_t in_shape[3], const int32_t out_shape[3], int32_t mode, void* stream);

/*
 * Anisotropy with per-element parameters (_simulate_anisotropy_fixed_axis,
 * anisotropy.py:180-208): along `axis`, y[b, c, .., p, ..] = x[.., lower[b, p], ..] (upper_dev
 * NULL: nearest) or x[.., lower[b, p], ..] * (1 - w[b, p]) + x[.., upper[b, p], ..] * w[b, p] with
 * every product and the sum rounded separately like the reference's three tensor ops.  The
 * (B, length) index / weight tables are the reference's own integer arithmetic, done on the
 * host.  active_dev (B bytes or NULL): 0 = the element is copied unchanged (factor <= 1).
 */
int tio_axis_gather_lerp(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels,
                         const int32_t shape[3], int32_t axis, const int32_t* lower_dev,
                         const int32_t* upper_dev, const float* weight_dev, const uint8_t* active_dev,
                         void* stream);

/*
 * Flip (transforms/spatial/flip.py:182-236): y = torch.flip(x, axes) as one element move.
 * axes_mask: bit a = flip spatial axis a for every element; flags_dev (B x 3 bytes, or NULL)
 * overrides it per element (_flip_per_element: a flip + torch.where per axis in the reference).
 */
int tio_flip3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels,
               const int32_t shape[3], int32_t axes_mask, const uint8_t* flags_dev, void* stream);

typedef enum tio_pad_mode {
  TIO_PAD_CONSTANT = 0,  /* F.pad(mode="constant", value=fill); also the statistic modes (per-element constants) */
  TIO_PAD_REFLECT = 1,   /* F.pad(mode="reflect"): mirror without repeating the edge                              */
  TIO_PAD_REPLICATE = 2, /* F.pad(mode="replicate"): clamp                                                        */
  TIO_PAD_CIRCULAR = 3   /* F.pad(mode="circular"): wrap                                                          */
} tio_pad_mode;

/*
 * pad_tensor (transforms/spatial/_padding.py:62-104; Pad, pad.py:88-108; GridSampler's border
 * padding, data/sampler.py:127-147): y = F.pad(x, (k0, k1, j0, j1, i0, i1), mode, value) as one
 * element move.  padding = (i0, i1, j0, j1, k0, k1).  The 'mean' / 'median' / 'minimum' modes of
 * the reference are constant padding with one value per batch element: fill_per_element_dev
 * (B values of the image dtype) overrides `fill` then.
 */
int tio_pad3d(const void* x, void* y, int32_t dtype, int32_t batch, int32_t channels,
              const int32_t in_shape[3], const int32_t padding[6], int32_t mode, double fill,
              const void* fill_per_element_dev, void* stream);

/* ------------------------------------------------------------------------ */
/* Feeding side: dense-inference patch aggregation        */
/* ------------------------------------------------------------------------ */

typedef enum tio_overlap_mode {
  TIO_OVERLAP_CROP = 0,    /* out[dst] = patch[src]          (aggregator.py:166-204) */
  TIO_OVERLAP_AVERAGE = 1, /* out += patch; weights += 1     (aggregator.py:206-214) */
  TIO_OVERLAP_HANN = 2     /* out += patch * w; weights += w (aggregator.py:216-232) */
} tio_overlap_mode;

#define TIO_MAX_PATCHES 32 /* placements per call */

/* Where one patch lands: voxels [src_ini, src_ini + extent) of the patch go to
 * [dst_ini, dst_ini + extent) of the volume (PatchLocation.to_slices(), or the trimmed
 * centre that 'crop' keeps).  HOST memory, copied into the launch. */
typedef struct tio_patch_placement {
  int32_t dst_ini[3];
  int32_t src_ini[3];
  int32_t extent[3];
} tio_patch_placement;

/*
 * PatchAggregator.add_batch without the D2H copy (aggregator.py:94-99 forces
 * tensor.cpu() and accumulates with one Python slice assignment per patch).
 *   out        (C, I, J, K) device accumulator, same dtype as the patches
 *   weight_sum (C, I, J, K) device, same dtype; NULL for TIO_OVERLAP_CROP
 *   patches    (n_patches, C, pi, pj, pk) device
 *   window_*   HANN only: the three 1-D windows torch.hann_window(size + 2,
 *              periodic=False)[1:-1] as float32 (device); the 3-D weight of a voxel
 *              is (wi * wj) * wk in float32, the rounding of _build_hann_3d
 *              (aggregator.py:237-245).
 * Every volume voxel applies the patches that cover it IN PATCH ORDER, so overlaps
 * accumulate in the reference's order (float addition does not commute) and 'crop'
 * keeps the last writer.  Arithmetic: float64 for TIO_F64, otherwise float32 rounded to
 * the storage dtype after every patch - what the in-place `+=` on a half tensor does.
 * CROP copies elements and takes every dtype; AVERAGE / HANN need a floating dtype.
 */
int tio_patch_accumulate(void* out, void* weight_sum, int32_t dtype, int32_t channels,
                         const int32_t vol_shape[3], const void* patches, int32_t n_patches,
                         const int32_t patch_shape[3], const tio_patch_placement* placements_host,
                         int32_t mode, const float* window_i_dev, const float* window_j_dev,
                         const float* window_k_dev, void* stream);

/*
 * torch.unique(data) of an 8- or 16-bit integer label map: the sorted value set that sizes the
 * one-hot encoding of the partial-volume label mode (spatial.py:1360; it becomes labels_dev /
 * n_labels of a TIO_LABEL_PV image).  A presence bitmap instead of torch.unique's sort.
 *   x             device, n elements of dtype TIO_U8 / TIO_I8 / TIO_I16, 16-byte aligned
 *   table_dev     device, room for 65536 doubles (256 for the 8-bit types): the values, ascending
 *   count_dev     device, one int32: how many
 *   workspace_dev device, 8 KiB scratch (the bitmap; cleared by the call)
 * Other dtypes: TIO_ERR_UNSUPPORTED_DTYPE, nothing launched (callers keep torch.unique).
 */
int tio_unique_labels(const void* x, int32_t dtype, int64_t n, double* table_dev, int32_t* count_dev,
                      void* workspace_dev, void* stream);

/* ------------------------------------------------------------------------ */
/* Motion: k-space compositing                                               */
/* ------------------------------------------------------------------------ */
#define TIO_MAX_SEGMENTS 32

/*
 * _apply_motion_segments (transforms/intensity/motion.py:334-372).  The reference takes
 * spectrum = fftn(x_0), overwrites the planes [bounds[s], bounds[s+1]) along the FIRST
 * spatial axis with the same planes of fftn(x_s) for every motion segment s >= 1 (x_s: the
 * rigidly moved image, tio_resample3d with the matrix of _apply_rigid_transform,
 * motion.py:393-425) and returns ifftn(spectrum).real.  Only the first axis is masked, so
 * the transforms along J and K cancel and the composite is the real linear map
 *
 *     out[b, c, i, j, k] = sum_s sum_i' W[s][i'][i] * x_s[b, c, i', j, k]
 *     W[s][i'][i] = (1 / I) * sum_{f = bounds[s]}^{bounds[s+1] - 1} cos(2 pi f (i - i') / I)
 *
 * which tio_kspace_segment_mix evaluates as one float32 GEMM per (b, c) on the matrix cores
 * (v_mfma_f32_32x32x2_f32: exact float32 products and sums, no FFT, no complex volumes).
 *   segments    host array of n_segments DEVICE pointers, each a float32 (B, C, I, J, K)
 *               tensor: segments[0] = data.float(), segments[s] = moved image s
 *   bounds      host, n_segments + 1 plane indices, bounds[0] = 0, bounds[n_segments] = I
 *               (_segment_bounds, motion.py:375-390)
 *   mix_dev     device, float32 (n_segments, I, I): the table W above; fill a host buffer
 *               with tio_kspace_mix_table (float64 arithmetic) and upload it once per shape
 *   out         device (B, C, I, J, K) of `dtype` (`.to(data.dtype)`); must not alias a segment
 *   active_dev  device, B bytes or NULL: elements with 0 are skipped, their `out` rows are NOT
 *               written (the caller restores them from the input, torch.where in the reference)
 * Parity: float rounding only (the reference's complex64 FFTs and this GEMM are two float32
 * evaluations of the same sums): <= 1e-5 of the image's magnitude in the tests.
 */
int tio_kspace_segment_mix(const void* const* segments, int32_t n_segments, const int32_t* bounds,
                           const float* mix_dev, void* out, int32_t dtype, int32_t batch,
                           int32_t channels, const int32_t shape[3], const uint8_t* active_dev,
                           void* stream);
/* Host helper: fills table_host (n_segments * length * length floats) with W. */
int tio_kspace_mix_table(int32_t length, int32_t n_segments, const int32_t* bounds, float* table_host);

/* ------------------------------------------------------------------------ */
/* Host helper: torch's CPU `randn` stream on all host cores (ABI 8)         */
/* ------------------------------------------------------------------------ */
/*
 * Replaces: `torch.randn(data.shape, generator=cpu_gen)` of the reference's Noise (transforms/intensity/noise.py:108-116,
 * 166-178: ONE CPU generator seeded with params["seed"], shared by the images of the batch in dict order, one or — Rician —
 * two draws per image).  torch draws on a single thread: 0.35 s for the bench batch.  Here the mt19937 state chain runs on
 * one thread with integer vectors and every other step (tempering, the 24-bit uniform, the Box-Muller step of
 * normal_fill_16_AVX2 with avx_mathfun's log / sincos) on `n_threads - 1` more, in place in `out`.  BIT-IDENTICAL to
 * torch 2.10's CPU kernel for float32 and n >= 16 (pinned against torch.randn: tests/test_host_rng.py); n < 16 returns
 * TIO_ERR_UNSUPPORTED_CONFIG (torch takes another path there: the caller keeps using torch.randn).  Host pointers only, no
 * device work, no stream.  The state persists across calls, like the generator it stands for: a second call continues
 * the stream where the first one stopped (including the 16 extra draws torch spends when n is not a multiple of 16).
 */
#define TIO_HOST_MT_STATE_BYTES 2688
typedef struct tio_host_mt_state { uint64_t opaque[TIO_HOST_MT_STATE_BYTES / 8]; } tio_host_mt_state;
/* `torch.Generator().manual_seed(seed)` (at::mt19937 keeps the low 32 bits) */
int tio_host_mt19937_seed(tio_host_mt_state* state, uint64_t seed);
/* `torch.randn(n, generator=...)` into out (host memory, ideally pinned); n_threads <= 1: everything on the calling thread */
int tio_host_mt19937_randn(tio_host_mt_state* state, float* out, int64_t n, int32_t n_threads);

/*
 * The same stream with the draws produced ON THE DEVICE (ABI 9): the host keeps the one part that is a chain — the
 * mt19937 state twists — and hands the device a PLAN: a snapshot of the state every 128 blocks (2.5 KB per 79 872 draws:
 * 4 MB for 8 x 256^3) plus the rest of the current block and torch's 16 tail draws.  tio_mt19937_randn_device replays the
 * twists from the snapshots (one workgroup per snapshot, the state in LDS) and applies tempering, the 24-bit uniform and
 * normal_fill_16_AVX2's Box-Muller step with avx_mathfun's log / sincos — the float32 operation sequence of the host
 * restatement, every multiply-add fused where torch's build fuses it: BIT-IDENTICAL to torch.randn(n, generator=cpu_gen)
 * (tests/test_gpu_device_rng.py).  No 4-bytes-per-draw upload, no worker threads: 6.5 ms of one host core per 134 M draws.
 *
 *   words = tio_host_mt19937_plan_words(n);                 capacity (uint32 words) a plan of n draws can need
 *   tio_host_mt19937_plan(state, n, plan_host, words, &used, n_threads);   advances `state` exactly like tio_host_mt19937_randn(state, ., n, .)
 *   copy plan_host[0 : used] to plan_dev (the caller's copy, on `stream`)
 *   tio_mt19937_randn_device(plan_host, plan_dev, out_dev, stream);
 *
 * tio_host_mt19937_plan returns TIO_ERR_UNSUPPORTED_CONFIG — and leaves the state untouched — when n < 16 or when the
 * stream stands inside a group of 16 (a previous draw count that was not a multiple of 16 AND did not end a block: torch's
 * groups then straddle state blocks); the caller falls back to tio_host_mt19937_randn.
 * n_threads > 1 cuts long chains into segments: mt19937 is linear over GF(2), so a thread can JUMP to the start of its
 * segment (g(f) applied to the state, g = x^J mod the characteristic polynomial — computed and verified at first use,
 * csrc/host_rng_jump.cpp) and chain from there: 134 M draws are planned in ~1.2 ms on 8 threads (0.55 - 0.65 on 32) instead of 5.5 - 7 ms on one.
 * The snapshot at the start of a jumped segment may differ from the chained one in the 31 low bits of its first word —
 * bits that are not part of the generator's state (the recurrence never reads them).
 */
int64_t tio_host_mt19937_plan_words(int64_t n);
int tio_host_mt19937_plan(tio_host_mt_state* state, int64_t n, uint32_t* plan_host, int64_t capacity_words, int64_t* used_words,
                          int32_t n_threads);
/* The same plan started AHEAD of its launch (ABI 10): _begin hands the call to a native thread and returns a handle (0: bad
 * arguments) at once, _end waits for it and returns what tio_host_mt19937_plan would have (status, *used_words).  The state
 * and the plan buffer belong to the job until _end has returned; one job per state at a time.  For a caller that knows the
 * seed before it is ready to launch (Compose drawing its children's parameters ahead, torchio_amd/transforms/compose.py):
 * the 0.6 - 0.9 ms of the state chain then overlap the caller's enqueue work instead of sitting on its critical path. */
int64_t tio_host_mt19937_plan_begin(tio_host_mt_state* state, int64_t n, uint32_t* plan_host, int64_t capacity_words, int32_t n_threads);
int tio_host_mt19937_plan_end(int64_t handle, int64_t* used_words);
int tio_mt19937_randn_device(const uint32_t* plan_host, const uint32_t* plan_dev, float* out_dev, void* stream);
/* The plan whose snapshots the DEVICE makes (ABI 15).  With several ranks per host the state chain is what the
 * reference-identical noise mode waits for (eight ranks: 4.8 - 5.7 ms per step and rank on the 15 threads each gets, against
 * 1.8 ms of GPU time).  Every snapshot is an independent polynomial evaluation over GF(2) — a jump is the correlation of the
 * polynomial's bits with the generator's own word sequence — so:
 *   tio_host_mt19937_plan_prefix(state, n, plan_host, capacity, &prefix_words, &used_words, &total_blocks)
 *       writes only the PREFIX of the plan — header, the rest of the current block, snapshot 0 (the state the chain starts
 *       from): prefix_words (= 1 280) words to upload of the used_words the plan has on the device — and advances `state` like
 *       tio_host_mt19937_plan does, lazily: the twists are owed and settled (one host-side jump) if the state is read again.
 *       TIO_ERR_UNSUPPORTED_CONFIG (state untouched) where tio_host_mt19937_plan returns it, and when n is not a multiple of 16
 *       (torch's tail rule needs the final state at once) — the caller then makes the whole plan on the host.
 *   tio_host_mt19937_segment_polynomials(segment_blocks, count, out)
 *       the jump polynomials t * segment_blocks twists ahead, t = 1 .. count, 624 words each (bit k of word k / 32 = the
 *       coefficient of x^k) — cached per segment length; upload once per device and length.
 *   tio_mt19937_device_snapshots(plan_dev, total_blocks, segment_blocks, polys_dev, stream)
 *       one workgroup per segment of segment_blocks (a multiple of 128) twists: jumps to its first state, chains through the
 *       segment, writes the snapshots into plan_dev.  The draw kernels above then read plan_dev as if the host had made it. */
int tio_host_mt19937_plan_prefix(tio_host_mt_state* state, int64_t n, uint32_t* plan_host, int64_t capacity_words, int64_t* prefix_words,
                                 int64_t* used_words, int64_t* total_blocks);
int tio_host_mt19937_segment_polynomials(int64_t segment_blocks, int32_t count, uint32_t* out);
int tio_mt19937_device_snapshots(uint32_t* plan_dev, int64_t total_blocks, int64_t segment_blocks, const uint32_t* polys_dev, void* stream);
/* Noise.apply_transform for a float32 image in one kernel: out = x + (mean + std z) with z the plan's draws (the three
 * float32 roundings of noise.py:178, :119, as tio_add_noise) — the draws never exist in memory.  x / out: the image's
 * B * n_per_element values (the plan's n); mean_dev / std_dev: (B,) per-element parameters or NULL (then the scalars).
 * out may be x itself only when n is a multiple of 16 (otherwise TIO_ERR_UNSUPPORTED_CONFIG: torch's tail rule draws the
 * last 16 values a second time, from x). */
int tio_mt19937_add_noise_device(const uint32_t* plan_host, const uint32_t* plan_dev, const float* x_dev, float* out_dev,
                                 int64_t n_per_element, float mean, float std, const float* mean_dev, const float* std_dev,
                                 void* stream);

/* ------------------------------------------------------------------------ */
/* Introspection                                                             */
/* ------------------------------------------------------------------------ */
int tio_abi_version(void);
const char* tio_last_error(void);
/* Number of visible HIP devices (0 when none; never throws). */
int tio_device_count(void);
/* The library parses its TIO_* environment switches (A/B experiments, see torchio_amd/csrc/common.hpp: EnvSwitches)
 * ONCE, at the first call that needs one; no entry point calls getenv() afterwards.  A process that changes such a
 * variable later (tests, tests/native/resample_bench) calls this to have them parsed again.  ABI 10. */
void tio_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* TIO_HIP_H */
