// resample_lean_exact.hpp — round 5: the lean planned brick kernel with the REFERENCE'S OWN COORDINATES.
//
// Why it exists (VERDICT r4, weak #1 / next #1 / next #3d).  The north-star tolerance, read per voxel
// (|d| <= 1e-4 max(|ref|, 1e-3 range)) on the white-noise volumes SURVEY 8(d) prescribes, leaves no room for a
// coordinate that is not the reference's float32 value bit for bit: one ulp of a coordinate near 128 is 1.5e-5 voxel,
// white noise turns that into a value difference of the same size, and a voxel whose reference value is below ~0.15
// is then beyond the bar.  TIO_PRECISION_FAST (resample_fast.hpp: the coordinate as a line per control cell) therefore
// cannot pass on such data whatever its fill-decision recheck does — the cause is structural.  What CAN be cheap is
// everything else.  This kernel keeps the reference's coordinate chain, operation for operation (MKL's FMA order of the
// affine row, ATen's lerp nesting of the displacement field, the normalise / un-normalise round trip with its correctly
// rounded division: the sequence of resample_tile_kernel and resample_kernel, spatial.py:1504-1648 + ATen
// grid_sampler_unnormalize), on the lean structure of resample_fast.hpp (one planned descriptor per brick, one float32
// channel per launch, the first LDS-DMA instruction a few hundred instructions after entry), and it puts the chain
// WHERE IT COSTS NOTHING: round 3's stamps showed a wave spending ~3 700 ticks issuing its 11 - 12 DMA instructions
// (330 ticks each: the CU's vector-memory path pushes back) and another ~1 450 waiting for the box — the 16 planes x 30
// vector instructions of the exact chain are issued BETWEEN the DMA instructions and while the box lands.  What is left
// behind the barrier is floor / weights / address / 8 taps / interpolation: fewer vector instructions per voxel than the
// FAST line kernel's sampling loop (no line evaluation, no run logic).
//
// Two instantiations of the interpolation:
//   EXACT_LERP = true  — ATen's grid_sampler_3d weights and accumulation order (tile_finish): the launch is bit-identical
//                        to resample_tile_kernel / the oracle.  TIO_PRECISION_EXACT takes it for large float32 launches.
//   EXACT_LERP = false — TIO_PRECISION_TIGHT: three nested fma lerps on the SAME coordinates, taps and fill decisions
//                        (the in-bounds weight is evaluated in ATen's order on ATen's weights wherever a tap can leave
//                        the volume, so `mask > 0.5` is the reference's decision by construction — no recheck tail).
//                        Differs from the reference by the rounding of seven fused multiply-adds: ~1e-7 of the taps.
//
// Included by resample.hip after resample_fast.hpp (planner, LeanArgs, StreamBox, StageLanes) and resample_tile.hpp
// (TapSet, TileAddr, tile_issue_interior, tile_finish*, tile_mask, cp_plane, normalise_roundtrip_folded).
#pragma once

namespace tio {

// ---- the packed LDS-DMA of one box as a STEPPER: the same instruction sequence as stream_stage_packed (resample_fast.hpp),
// one instruction per call, so that the caller can put arithmetic between two of them ------------------------------------
// every vector-memory AND LDS operation of this wave done (the zero chunks of BoxDmaStepper::issue<false> are LDS stores the
// compiler does not know about)
__device__ __forceinline__ void tile_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

template <int NW>
struct BoxDmaStepper {
  typedef __attribute__((address_space(1))) const char* global_byte_ptr;
  global_byte_ptr origin;
  float* lp;
  unsigned off, step_b, wrap_b;
  int row, p, r;
  int step, step_p, step_r, total_rows, Ly, lp_step;
  int bx0, by0, I, J;
  int left;  // DMA instructions this wave has not issued yet (wave uniform)
  bool ch_ok, lane_ok;

  __device__ __forceinline__ void init(float* tile, const float* src, const StreamBox& bx, int I_, int J_, int K, int wave, int lane) {
    StageLanes sl;
    sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
    stage_lanes(sl, bx.cpr, K, lane);
    const int rpi = sl.rpi;
    total_rows = bx.Lx * bx.Ly;
    // (uniform float divisions of small integers: exact after the half-step nudge — as in stream_stage_packed)
    const int n_instr = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(total_rows + rpi - 1) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(rpi))));
    step = NW * rpi;
    const float rcp_ly = __builtin_amdgcn_rcpf(static_cast<float>(bx.Ly));
    step_p = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(step) + 0.5f) * rcp_ly));
    step_r = step - step_p * bx.Ly;
    row = wave * rpi + sl.row_l;
    p = static_cast<int>((static_cast<float>(row) + 0.5f) * rcp_ly);
    r = row - p * bx.Ly;
    const unsigned plane_b = static_cast<unsigned>(J_) * static_cast<unsigned>(K) * 4u, row_b = static_cast<unsigned>(K) * 4u;
    off = static_cast<unsigned>(p) * plane_b + static_cast<unsigned>(r) * row_b + static_cast<unsigned>(sl.gz_rel) * 4u;
    step_b = static_cast<unsigned>(step_p) * plane_b + static_cast<unsigned>(step_r) * row_b;
    wrap_b = plane_b - static_cast<unsigned>(bx.Ly) * row_b;
    origin = (global_byte_ptr)(src) + ((static_cast<int64_t>(bx.bx0) * J_ + bx.by0) * K + bx.za) * 4;
    const int dgroup = rpi * bx.cpr * 4;  // LDS floats per instruction
    lp = tile + wave * dgroup;
    lp_step = NW * dgroup;
    ch_ok = static_cast<unsigned>(bx.za + sl.gz_rel) < static_cast<unsigned>(K);
    lane_ok = sl.lane_ok;
    Ly = bx.Ly; bx0 = bx.bx0; by0 = bx.by0; I = I_; J = J_;
    left = n_instr > wave ? (n_instr - wave + NW - 1) / NW : 0;
  }

  // one DMA instruction; INTERIOR: the box lies inside the volume (no per-row checks, nothing to zero)
  template <bool INTERIOR>
  __device__ __forceinline__ void issue(int lane) {
    const bool in_box = lane_ok & (row < total_rows);
    if constexpr (INTERIOR) {
      if (in_box) __builtin_amdgcn_global_load_lds(origin + off, (fast_lds_wptr)(lp), 16, 0, 0);
    } else {
      const bool in_vol = in_box & ch_ok & (static_cast<unsigned>(bx0 + p) < static_cast<unsigned>(I)) &
                          (static_cast<unsigned>(by0 + r) < static_cast<unsigned>(J));
      if (in_vol) __builtin_amdgcn_global_load_lds(origin + off, (fast_lds_wptr)(lp), 16, 0, 0);
      else if (in_box) {
        // zeros for the chunks outside the volume — as an instruction the compiler does not see (lds_zero_chunk, resample_tile.hpp:
        // a plain LDS store here is preceded by `s_waitcnt vmcnt(0)`, and the issue phase of the third of the bricks that touch
        // the volume's surface became a chain of memory round trips); callers wait with tile_dma_wait_all() before the barrier.
        lds_zero_chunk(lp + 4 * lane);
      }
    }
    row += step; p += step_p; r += step_r; off += step_b;
    if (r >= Ly) { r -= Ly; p += 1; off += wrap_b; }
    lp += lp_step;
    left -= 1;
  }
};

// ---- the reference's coordinate chain for one voxel, by composition mode (block uniform; literals fold the branches) ----
//   MODE 0: no displacement (Affine; elements of an elastic launch whose control points are skipped)
//   MODE 1: displacement on an identity mapping (ElasticDeformation alone): c + d either way round
//   MODE 2: affine first (spatial.py:1570-1573)      MODE 3: elastic first (spatial.py:1574-1577)
// UNIT: all three spacings are 1.0f (d / 1 is the identity); SHORT: one Markstein refinement is the correctly rounded
// quotient for this launch's divisors (resample_tile.hpp: normalise_roundtrip_folded, tests/native/divtest.c)
template <int MODE, bool UNIT, bool SHORT>
__device__ __forceinline__ void lean_exact_coord(const float (&m)[12], const LeanArgs& a, float ci, float cj, float ck, float di, float dj,
                                                 float dk, float& x, float& y, float& z) {
#define TIO_LE_ROW(R, A, B, C) \
  __builtin_fmaf(1.0f, m[4 * R + 3], __builtin_fmaf(C, m[4 * R + 2], __builtin_fmaf(B, m[4 * R + 1], __fmul_rn(A, m[4 * R]))))
  float vi, vj, vk;
  if constexpr (MODE == 0) {
    vi = TIO_LE_ROW(0, ci, cj, ck); vj = TIO_LE_ROW(1, ci, cj, ck); vk = TIO_LE_ROW(2, ci, cj, ck);
  } else {
    float qi = di, qj = dj, qk = dk;
    if constexpr (!UNIT) {
      qi = exact_div(qi, a.sp[0], a.rsp[0]);
      qj = exact_div(qj, a.sp[1], a.rsp[1]);
      qk = exact_div(qk, a.sp[2], a.rsp[2]);
    }
    if constexpr (MODE == 1) {
      vi = __fadd_rn(ci, qi); vj = __fadd_rn(cj, qj); vk = __fadd_rn(ck, qk);
    } else if constexpr (MODE == 2) {
      vi = __fadd_rn(TIO_LE_ROW(0, ci, cj, ck), qi);
      vj = __fadd_rn(TIO_LE_ROW(1, ci, cj, ck), qj);
      vk = __fadd_rn(TIO_LE_ROW(2, ci, cj, ck), qk);
    } else {
      const float ei = __fadd_rn(ci, qi), ej = __fadd_rn(cj, qj), ek = __fadd_rn(ck, qk);
      vi = TIO_LE_ROW(0, ei, ej, ek); vj = TIO_LE_ROW(1, ei, ej, ek); vk = TIO_LE_ROW(2, ei, ej, ek);
    }
  }
#undef TIO_LE_ROW
  x = normalise_roundtrip_folded<SHORT>(vi, a.dh[0], a.rdh[0], a.half_h[0]);
  y = normalise_roundtrip_folded<SHORT>(vj, a.dh[1], a.rdh[1], a.half_h[1]);
  z = normalise_roundtrip_folded<SHORT>(vk, a.dh[2], a.rdh[2], a.half_h[2]);
}

// ---- round 6: what a ROW of the wave shares ---------------------------------------------------------------------------------
// A wave is 4 rows of 16 lanes (tk = lane & 15 along K, one tj per row).  Everything that depends on (plane, j) only — the
// first two terms of the affine row, fma(j, m1, i * m0) — is the same in the 16 lanes of a row, and so is what depends on
// the plane alone (the control grid's lerp along I).  Lane t of a row computes it for plane t ONCE per brick; plane t's
// value reaches the other lanes through the DPP row broadcast of gfx90a+ (v_mov_b32_dpp row_newbcast:t — one full-rate move
// instead of a multiply, a multiply-add and a conversion per row, or of a half-rate v_readlane per scalar).
__device__ __forceinline__ float row_bcast16(float v, int t) {  // (t is a literal after unrolling: the control word must be)
  const int b = __float_as_int(v);
  switch (t) {
  // (`mov_dpp`, not `update_dpp(0, ..)`: every lane of the row is written, and with an explicit old value the compiler initialises the
  // destination first — a `v_mov_b32 v, 0` (and often an `s_nop`) in front of EACH of the three to five broadcasts per plane, in the issue class of the
  // move itself: 4.6 cycles apiece on this chip against 2.9 for a multiply-add)
#define TIO_LE_BCAST(T) case T: return __int_as_float(__builtin_amdgcn_mov_dpp(b, 0x150 + T, 0xF, 0xF, false));
    TIO_LE_BCAST(0) TIO_LE_BCAST(1) TIO_LE_BCAST(2) TIO_LE_BCAST(3) TIO_LE_BCAST(4) TIO_LE_BCAST(5) TIO_LE_BCAST(6) TIO_LE_BCAST(7)
    TIO_LE_BCAST(8) TIO_LE_BCAST(9) TIO_LE_BCAST(10) TIO_LE_BCAST(11) TIO_LE_BCAST(12) TIO_LE_BCAST(13) TIO_LE_BCAST(14) TIO_LE_BCAST(15)
#undef TIO_LE_BCAST
  }
  return v;
}

// lean_exact_coord for the modes whose affine row acts on the integer voxel index (MODE 0 and 2), from the row's shared
// partial sums p_r = fma(j, m[4r+1], i * m[4r]): the remaining two terms in MKL's order — fma(k, m2, p), then
// fma(1, m3, .), which IS the rounded sum — bit for bit the same value
template <int MODE, bool SHORT>
__device__ __forceinline__ void lean_exact_coord_shared(const float (&m)[12], const LeanArgs& a, float p0, float p1, float p2, float ck, float di,
                                                        float dj, float dk, float& x, float& y, float& z) {
  static_assert(MODE == 0 || MODE == 2, "the shared partial sums are those of the integer voxel index");
  float vi = __fadd_rn(__builtin_fmaf(ck, m[2], p0), m[3]);
  float vj = __fadd_rn(__builtin_fmaf(ck, m[6], p1), m[7]);
  float vk = __fadd_rn(__builtin_fmaf(ck, m[10], p2), m[11]);
  if constexpr (MODE == 2) { vi = __fadd_rn(vi, di); vj = __fadd_rn(vj, dj); vk = __fadd_rn(vk, dk); }  // (unit spacing: d / 1)
  x = normalise_roundtrip_folded<SHORT>(vi, a.dh[0], a.rdh[0], a.half_h[0]);
  y = normalise_roundtrip_folded<SHORT>(vj, a.dh[1], a.rdh[1], a.half_h[1]);
  z = normalise_roundtrip_folded<SHORT>(vk, a.dh[2], a.rdh[2], a.half_h[2]);
}

// (j, k)-lerped control planes ia, ia + 1, ia + 2 of this thread's column, three components each.  NAMED members, not an
// array: the optimiser turns `e == 0 ? P[0] : (e == 3 ? P[3] : P[6])` into a dynamically indexed load of a stack object
// (first build of this kernel: 48 bytes of scratch per lane and a scratch_load per component per plane).
struct CtlPlanes {
  float a_i, a_j, a_k, b_i, b_j, b_k, c_i, c_j, c_k;
};

// The 16 planes of this thread's column (phase A), one DMA instruction of an interior box between two planes.
//   li_lane: lane t of the wave holds ATen's lerp of plane i_begin + t along the control grid's first axis (block uniform
//   per plane: read back as scalars)
struct LeanNoPlaneHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// `each(t)`: called once plane t's coordinates stand (resample_nearest_exact_kernel: the plane's load goes out while the next
// plane is being formed); the float kernels pass nothing
template <int MODE, bool UNIT, bool SHORT, bool INTERLEAVE, int NW, typename EACH = LeanNoPlaneHook>
__device__ __forceinline__ void lean_exact_planes(const float (&m)[12], const LeanArgs& a, int i0, int i_last, float cj, float ck,
                                                  const Lerp1D& li_lane, int ia, const CtlPlanes P, BoxDmaStepper<NW>& dma, int lane,
                                                  float (&X)[16], float (&Y)[16], float (&Z)[16], EACH each = EACH()) {
  float pa_i = 0.f, pa_j = 0.f, pa_k = 0.f, pb_i = 0.f, pb_j = 0.f, pb_k = 0.f;
  int cur0 = -1, cur1 = -1;
  // what the row shares (round 6): lane t of a row holds plane t's partial sums of the three affine rows ...
  constexpr bool SHARED = (MODE == 0 || MODE == 2) && UNIT;
  float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f;
  if constexpr (SHARED) {
    const float ci_l = static_cast<float>(min(i0 + (lane & 15), i_last));
    ps0 = __builtin_fmaf(cj, m[1], __fmul_rn(ci_l, m[0]));
    ps1 = __builtin_fmaf(cj, m[5], __fmul_rn(ci_l, m[4]));
    ps2 = __builtin_fmaf(cj, m[9], __fmul_rn(ci_l, m[8]));
  }
  // ... and which control planes plane t lerps between, as two bit masks per operand (bit t: the operand is control plane
  // ia + 1 or beyond / ia + 2): scalar compares on literals instead of two v_readlane per plane
  unsigned long long e0_ge1 = 0, e0_ge2 = 0, e1_ge1 = 0, e1_ge2 = 0;
  unsigned changes = 0u;
  if constexpr (MODE != 0) {
    e0_ge1 = __builtin_amdgcn_ballot_w64(li_lane.i0 - ia >= 1); e0_ge2 = __builtin_amdgcn_ballot_w64(li_lane.i0 - ia >= 2);
    e1_ge1 = __builtin_amdgcn_ballot_w64(li_lane.i1 - ia >= 1); e1_ge2 = __builtin_amdgcn_ballot_w64(li_lane.i1 - ia >= 2);
    // ... and at which planes either operand CHANGES (bit t: plane t lerps between other control planes than plane t - 1; bit 0: the
    // first plane) — once or twice per brick.  One scalar bit test per plane decides whether the operands are looked at at all (the
    // extraction of both operands' plane numbers and their two compares cost ten scalar instructions per plane, and a wave issues its
    // scalar and its vector instructions in order)
    const unsigned g01 = static_cast<unsigned>(e0_ge1) & 0xFFFFu, g02 = static_cast<unsigned>(e0_ge2) & 0xFFFFu;
    const unsigned g11 = static_cast<unsigned>(e1_ge1) & 0xFFFFu, g12 = static_cast<unsigned>(e1_ge2) & 0xFFFFu;
    changes = ((g01 ^ (g01 << 1)) | (g02 ^ (g02 << 1)) | (g11 ^ (g11 << 1)) | (g12 ^ (g12 << 1)) | 1u) & 0xFFFFu;
  }
#pragma unroll
  for (int t = 0; t < 16; t++) {
    if constexpr (INTERLEAVE) {
      // ONE DMA instruction, then ONE plane (fenced: left alone the scheduler hoists a group's four DMA instructions in front
      // of its four planes, and a wave that waits at the vector-memory queue issues no arithmetic; a wave issues one
      // vector instruction per ~5 clocks on this chip whatever its ILP — profiles/r02_resample_sq.md — so nothing is lost)
      if (dma.left > 0) dma.template issue<true>(lane);  // (wave-uniform branch)
      __builtin_amdgcn_sched_barrier(0);
    }
    float di = 0.0f, dj = 0.0f, dk = 0.0f;
    if constexpr (MODE != 0) {
      // the two control planes a voxel lerps between change once or twice per brick: named registers that a SCALAR branch
      // refreshes (resample_tile.hpp: indexing by the plane number costs an s_set_gpr_idx sequence per access)
      const float l0 = row_bcast16(li_lane.l0, t), l1 = row_bcast16(li_lane.l1, t);
      if ((changes >> t) & 1u) {  // (rarely taken; bit 0 is always set: the first plane loads both operands)
        const int e0 = static_cast<int>((e0_ge1 >> t) & 1ull) + static_cast<int>((e0_ge2 >> t) & 1ull);
        const int e1 = static_cast<int>((e1_ge1 >> t) & 1ull) + static_cast<int>((e1_ge2 >> t) & 1ull);
        // (ONE rarely taken scalar branch per operand plane, branch-free selects inside: the nested if / else form compiled into
        // ~45 scalar instructions per plane — and a CU has one scalar unit for its twelve resident waves)
        if (e0 != cur0) {
          const bool z0 = e0 == 0, z1 = e0 == 1;
          pa_i = z0 ? P.a_i : (z1 ? P.b_i : P.c_i);
          pa_j = z0 ? P.a_j : (z1 ? P.b_j : P.c_j);
          pa_k = z0 ? P.a_k : (z1 ? P.b_k : P.c_k);
          cur0 = e0;
        }
        if (e1 != cur1) {
          const bool z0 = e1 == 0, z1 = e1 == 1;
          pb_i = z0 ? P.a_i : (z1 ? P.b_i : P.c_i);
          pb_j = z0 ? P.a_j : (z1 ? P.b_j : P.c_j);
          pb_k = z0 ? P.a_k : (z1 ? P.b_k : P.c_k);
          cur1 = e1;
        }
      }
      di = lerp2(pa_i, l0, pb_i, l1);
      dj = lerp2(pa_j, l0, pb_j, l1);
      dk = lerp2(pa_k, l0, pb_k, l1);
    }
    if constexpr (SHARED) {
      lean_exact_coord_shared<MODE, SHORT>(m, a, row_bcast16(ps0, t), row_bcast16(ps1, t), row_bcast16(ps2, t), ck, di, dj, dk, X[t], Y[t], Z[t]);
    } else {
      const float ci = static_cast<float>(min(i0 + t, i_last));  // (scalar unit, one conversion)
      lean_exact_coord<MODE, UNIT, SHORT>(m, a, ci, cj, ck, di, dj, dk, X[t], Y[t], Z[t]);
    }
    if constexpr (INTERLEAVE) {
      // pin the plane HERE: the optimiser sinks the (pure) chain of all 16 planes behind the last DMA instruction otherwise —
      // the first build of this kernel "interleaved" nothing (its assembly: 16 DMA blocks back to back, then 16 planes)
      asm volatile("" : "+v"(X[t]), "+v"(Y[t]), "+v"(Z[t])::"memory");
    }
    each(t);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ATen's grid_sampler_3d for ONE voxel straight from global memory (bricks whose box does not fit the tile, non-finite
// geometry): per-tap bounds, zero padding, the in-bounds weight mask and the fill rule in ATen's accumulation order —
// gather_voxel's arithmetic (resample_tile.hpp) for one float32 channel.  Bit-identical to the exact kernels in both
// instantiations (these bricks are rare; the fused lerps are not worth a second copy).
__device__ __forceinline__ float lean_exact_gather(const float* __restrict__ chan, int J, int K, float x, float y, float z, bool has_fill,
                                                   float fillv, float hx, float hy, float hz) {
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
  const float wx0 = x1 - x, wx1 = x - x0, wy0 = y1 - y, wy1 = y - y0, wz0 = z1 - z, wz1 = z - z0;
  const bool bx0 = (x0 >= 0.0f) & (x0 <= hx), bx1 = (x1 >= 0.0f) & (x1 <= hx);
  const bool by0 = (y0 >= 0.0f) & (y0 <= hy), by1 = (y1 >= 0.0f) & (y1 <= hy);
  const bool bz0 = (z0 >= 0.0f) & (z0 <= hz), bz1 = (z1 >= 0.0f) & (z1 <= hz);
  const int ix0 = static_cast<int>(fminf(fmaxf(x0, 0.0f), hx)), ix1 = static_cast<int>(fminf(fmaxf(x1, 0.0f), hx));
  const int iy0 = static_cast<int>(fminf(fmaxf(y0, 0.0f), hy)), iy1 = static_cast<int>(fminf(fmaxf(y1, 0.0f), hy));
  const int iz0 = static_cast<int>(fminf(fmaxf(z0, 0.0f), hz)), iz1 = static_cast<int>(fminf(fmaxf(z1, 0.0f), hz));
  float val = 0.0f, mask = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float w = __fmul_rn(__fmul_rn((k & 1) ? wx1 : wx0, (k & 2) ? wy1 : wy0), (k & 4) ? wz1 : wz0);
    const bool ok = ((k & 1) ? bx1 : bx0) & ((k & 2) ? by1 : by0) & ((k & 4) ? bz1 : bz0);
    const int off = (((k & 1) ? ix1 : ix0) * J + ((k & 2) ? iy1 : iy0)) * K + ((k & 4) ? iz1 : iz0);
    const float v = chan[off];
    const float next_v = __fadd_rn(val, __fmul_rn(v, w));
    const float next_m = __fadd_rn(mask, w);
    val = ok ? next_v : val;
    mask = ok ? next_m : mask;
  }
  if (has_fill) val = (mask > 0.5f) ? val : fillv;
  return val;
}

// Four planes of this thread's column from the staged box: the VALUES (taps in flight for all four before the first
// interpolation starts).  MASKED: a tap of this wave's group can leave the volume and the image has a fill rule — the
// in-bounds weight on ATen's weights in ATen's order (tile_mask); `has_fill` false (partial bricks reuse the masked code
// for images without a fill rule): the sample stands.
template <bool EXACT_LERP, bool MASKED>
__device__ __forceinline__ void lean_exact_group_values(const float (&X4)[4], const float (&Y4)[4], const float (&Z4)[4], const TileAddr& ta, float hx,
                                                        float hy, float hz, bool has_fill, float fillv, float (&vals)[4]) {
  TapSet ts[4];
#pragma unroll
  for (int u = 0; u < 4; u++) tile_issue_folded(ts[u], X4[u], Y4[u], Z4[u], ta);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    float val = EXACT_LERP ? tile_finish(ts[u]) : tile_finish_fast(ts[u]);
    if constexpr (MASKED) val = (!has_fill || tile_mask(ts[u], X4[u], Y4[u], Z4[u], hx, hy, hz) > 0.5f) ? val : fillv;
    vals[u] = val;
  }
}

// ... and their stores.  GUARD: a partial brick (stores predicated).  TRACK: the folded minimum of batch element 0.
template <bool GUARD, bool TRACK>
__device__ __forceinline__ void lean_exact_group_store(const float (&vals)[4], char*& out_generic, unsigned urow, int64_t slab_b, int t0, int i_count,
                                                       bool col_active, uint32_t& kmin) {
  typedef __attribute__((address_space(1))) char* global_char_ptr;   // (typed global: a flat store would count on lgkmcnt too —
  typedef __attribute__((address_space(1))) float* global_float_ptr;  //  resample_fast.hpp: fast_sample_run)
  global_char_ptr out_t = (global_char_ptr)out_generic;
  // (the row offset re-enters the block as a 32-bit register: instruction selection works block by block, and only a zero
  // extension it can SEE lets the store take the `scalar base + 32-bit vector offset` form — otherwise one 64-bit vector add per store)
  unsigned row_off = urow;
  asm volatile("" : "+v"(row_off));
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const bool live = !GUARD || (col_active && (t0 + u) < i_count);
    if (live) {
      *(global_float_ptr)(out_t + row_off) = vals[u];
      if constexpr (TRACK) kmin = min(kmin, float_to_key(vals[u]));
    }
    out_t += slab_b;
    asm volatile("" : "+s"(out_t));  // one running pointer (2 scalar adds per plane)
  }
  out_generic = (char*)out_t;
}

template <bool EXACT_LERP, bool MASKED, bool GUARD, bool TRACK>
__device__ __forceinline__ void lean_exact_group(const float (&X4)[4], const float (&Y4)[4], const float (&Z4)[4], const TileAddr& ta, char*& out_generic,
                                                 unsigned urow, int64_t slab_b, int t0, int i_count, bool col_active, float hx, float hy,
                                                 float hz, bool has_fill, float fillv, uint32_t& kmin) {
  float vals[4];
  lean_exact_group_values<EXACT_LERP, MASKED>(X4, Y4, Z4, ta, hx, hy, hz, has_fill, fillv, vals);
  lean_exact_group_store<GUARD, TRACK>(vals, out_generic, urow, slab_b, t0, i_count, col_active, kmin);
  __builtin_amdgcn_sched_barrier(0);
}

// can a tap of these four voxels lie outside the volume?  (first taps in [0, S - 2] on every axis <=> all eight inside)
__device__ __forceinline__ bool lean_exact_group_leaves(const float (&X4)[4], const float (&Y4)[4], const float (&Z4)[4], float hx, float hy, float hz) {
  const float xl = fminf(fminf(X4[0], X4[1]), fminf(X4[2], X4[3])), xh = fmaxf(fmaxf(X4[0], X4[1]), fmaxf(X4[2], X4[3]));
  const float yl = fminf(fminf(Y4[0], Y4[1]), fminf(Y4[2], Y4[3])), yh = fmaxf(fmaxf(Y4[0], Y4[1]), fmaxf(Y4[2], Y4[3]));
  const float zl = fminf(fminf(Z4[0], Z4[1]), fminf(Z4[2], Z4[3])), zh = fmaxf(fmaxf(Z4[0], Z4[1]), fmaxf(Z4[2], Z4[3]));
  // (a NaN coordinate compares false: the group is "leaving" and takes the masked code, whose bounds tests are ATen's)
  const bool inside = (xl >= 0.0f) & (xh < hx) & (yl >= 0.0f) & (yh < hy) & (zl >= 0.0f) & (zh < hz);
  return !inside;
}


// The planes of a brick that are NOT staged, voxel by voxel: every plane of a brick the planner could not stage (box beyond the
// LDS budget, non-finite geometry; bricks over more than three control planes), or — round 6 — the planes of those passes of a
// multi-pass brick that still do not fit (kPassSlow) or see nothing of the volume (kPassOutside: the fill value).  `states`: 4
// bits per pass, a pass = 1 << span_shift planes.  Everything but the brick index is derived HERE, from the argument block
// (through a pointer the optimiser cannot see through) and the descriptor: the call that follows the staged passes of a
// multi-pass brick must not keep the mapping, the control-point pointers and the column's constants alive across the sampling
// loop (first build: 24 - 60 scalar registers spilled in every instantiation, +2 ... +7 % on launches without such a brick).
// element bits of a label channel (1 / 2 / 4 bytes, block uniform) at an element offset, and their store
__device__ __forceinline__ unsigned lean_label_load(const void* chan, int es, int64_t off) {
  if (es == 1) return static_cast<const uint8_t*>(chan)[off];
  if (es == 2) return static_cast<const uint16_t*>(chan)[off];
  return static_cast<const uint32_t*>(chan)[off];
}
__device__ __forceinline__ void lean_label_store(void* chan, int es, int64_t off, unsigned bits) {
  if (es == 1) static_cast<uint8_t*>(chan)[off] = static_cast<uint8_t>(bits);
  else if (es == 2) static_cast<uint16_t*>(chan)[off] = static_cast<uint16_t>(bits);
  else static_cast<uint32_t*>(chan)[off] = bits;
}

// LABEL (only with SECOND = false): the brick's label voxels as well, from the coordinates this road forms anyway (every plane of a
// brick that is not staged at all — the only bricks a label launch sends here)
template <bool ELASTIC_POSSIBLE, bool SECOND = false, bool LABEL = false>
__device__ __forceinline__ void lean_exact_slow_planes(unsigned brick, int states, int span_shift, uint32_t& kmin) {
  typedef __attribute__((address_space(4))) const LeanArgs* const_args_ptr;
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const_args_ptr ka = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const int b = static_cast<int>(fastdiv_exact(brick, ka->bpe_magic, ka->bricks_per_element));
  const_int_ptr d = (const_int_ptr)(ka->plan + ka->B * 16) + static_cast<size_t>(brick) * kDescInts;
  const int i_begin = d[11], j_lo = d[12], k_lo = d[13];
  const bool elastic = ELASTIC_POSSIBLE && d[14] != 0;
  const_float_ptr mp = (const_float_ptr)(ka->mapping) + (ka->mapping_batched ? b * 12 : 0);
  float m[12];
#pragma unroll
  for (int q = 0; q < 12; q++) m[q] = mp[q];
  const int tid = threadIdx.x, tk = tid & 15, tj = tid >> 4;
  const int Io = ka->Io, Jo = ka->Jo, Ko = ka->Ko;
  const int i_count = min(16, Io - i_begin), nv = min(16, Jo - j_lo), nw = min(16, Ko - k_lo);
  if (!((tj < nv) & (tk < nw))) return;
  const int col_off = (j_lo + tj) * Ko + (k_lo + tk);
  const int64_t slab_b = static_cast<int64_t>(Jo) * Ko * 4;
  const float* in_chan = SECOND ? ka->in2 + static_cast<int64_t>(b) * ka->in_stride2 : ka->in + static_cast<int64_t>(b) * ka->in_stride;
  char* out_chan = reinterpret_cast<char*>(SECOND ? ka->out2 + static_cast<int64_t>(b) * ka->out_stride2 : ka->out + static_cast<int64_t>(b) * ka->out_stride);
  const float* fill_p = SECOND ? ka->fill2 : ka->fill;
  const bool has_fill = fill_p != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)fill_p)[0] : 0.0f;
  const float hx = ka->hx, hy = ka->hy, hz = ka->hz;
  const float cj = static_cast<float>(j_lo + tj), ck = static_cast<float>(k_lo + tk);
  const float* cp = elastic ? ka->cp + (ka->cp_batched ? static_cast<int64_t>(b) * (ka->ni * ka->nj * ka->nk * 3) : 0) : nullptr;
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      lj = lerp_index(j_lo + tj, ka->nj, Jo, ka->scj);
      lk = lerp_index(k_lo + tk, ka->nk, Ko, ka->sck);
    }
  }
  ExactChainArgs ea;
  ea.ni = ka->ni; ea.nj = ka->nj; ea.nk = ka->nk; ea.Io = Io; ea.unit_spacing = ka->unit_spacing; ea.affine_first = ka->affine_first;
  ea.scale_i = ka->sci;
#pragma unroll
  for (int e = 0; e < 3; e++) { ea.sp[e] = ka->sp[e]; ea.rsp[e] = ka->rsp[e]; ea.den[e] = ka->den[e]; ea.rden[e] = ka->rden[e]; }
  ea.size_m1[0] = hx; ea.size_m1[1] = hy; ea.size_m1[2] = hz;
  for (int t = 0; t < i_count; t++) {
    const int st = (states >> (4 * (t >> span_shift))) & 0xF;  // (block uniform)
    if (st == kPassStaged) continue;
    float val = fillv;
    if (st == kPassSlow) {
      float x, y, z;
      exact_voxel_coords<ELASTIC_POSSIBLE>(ea, m, elastic, cp, lj, lk, i_begin + t, cj, ck, x, y, z);
      val = lean_exact_gather(in_chan, ka->J, ka->K, x, y, z, has_fill, fillv, hx, hy, hz);
      if constexpr (LABEL) {
        // nearbyint per axis, zero padding (resample_nearest.hpp: nearest_offset), element bits
        const float xn = rintf(x), yn = rintf(y), zn = rintf(z);
        const bool ok = (xn >= 0.0f) & (xn <= hx) & (yn >= 0.0f) & (yn <= hy) & (zn >= 0.0f) & (zn <= hz);  // NaN fails
        const int64_t n_in_l = static_cast<int64_t>(ka->I) * ka->J * ka->K, n_out_l = static_cast<int64_t>(Io) * Jo * Ko;
        const int es = ka->lab_es;
        unsigned bits = 0u;
        if (ok) bits = lean_label_load(ka->lab_in, es, b * n_in_l + (static_cast<int64_t>(static_cast<int>(xn)) * ka->J + static_cast<int>(yn)) * ka->K + static_cast<int>(zn));
        lean_label_store(ka->lab_out, es, b * n_out_l + static_cast<int64_t>(i_begin + t) * (static_cast<int64_t>(Jo) * Ko) + col_off, bits);
      }
    }
    *reinterpret_cast<float*>(out_chan + (i_begin + t) * slab_b + static_cast<unsigned>(col_off) * 4u) = val;
    kmin = min(kmin, float_to_key(val));
  }
}

// One brick.  MULTI = false: the body of resample_lean_exact_kernel, one block per brick of the launch — a multi-pass brick
// (kDescMulti) is NOT its business: it returns at once, and resample_lean_exact_multi_kernel, behind it on the stream, walks
// the list of those bricks the planner left (first build of round 6: the pass logic inside the one kernel cost every
// instantiation 24 - 75 spilled scalar registers, +2 ... +7 % on launches that have no such brick).  MULTI = true: the same body
// with the pass switches compiled in.
// PAIR (round 6; never with FOLD_MIN or MULTI): the launch carries a second channel of the same geometry (a.in2 / out2 / fill2).  The
// block samples it from the SAME sixteen planes of coordinates: every wave done with the tile, the same box of the second
// channel staged into it (requested and waited for in one go — the other resident blocks cover the wait), the sampling loop again.
// What a second launch would repeat — descriptor, control planes, the coordinate chain: about half of a block's life — is done once.
// LABEL (round 6; never with FOLD_MIN or MULTI): ONE nearest-neighbour label channel of the same geometry rides along (a.lab_in /
// lab_out / lab_es).  Its voxels depend on the SAME coordinates through three roundings, and its own kernel (resample_nearest.hpp:
// resample_nearest_exact_kernel, 86 vector instructions per voxel with control points) spends two thirds of them forming those
// coordinates again: here the tail alone — offset on the full-rate pipes, a buffer load whose range check is the zero padding, a store —
// runs behind the float channels, ~20 instructions per voxel.
// ... in two halves: the LOADS go out as soon as the sixteen planes of coordinates stand — in front of the wait for the box, which then
// leaves exactly these sixteen in flight (`tile_dma_wait_keep16`) — and land while the float channels are sampled (three resident blocks
// per CU hide little: with loads and stores together behind the float channels the tail cost as much as the label map's own kernel); the
// STORES close the block.
__device__ __forceinline__ void tile_dma_wait_keep16() { asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void lean_exact_label_issue(const float (&X)[16], const float (&Y)[16], const float (&Z)[16], int b, unsigned (&vlab)[16]) {
  typedef __attribute__((address_space(4))) const LeanArgs* const_args_ptr;
  const_args_ptr ka = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const int es = ka->lab_es;  // block uniform
  const int J = ka->J, Kes = ka->K * es;
  const unsigned Hx = static_cast<unsigned>(ka->hx), Hy = static_cast<unsigned>(ka->hy), Hz = static_cast<unsigned>(ka->hz);
  const int64_t n_in = static_cast<int64_t>(ka->I) * ka->J * ka->K;
  char* src = static_cast<char*>(const_cast<void*>(ka->lab_in)) + b * n_in * es;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(src, 0, static_cast<int>(static_cast<unsigned>(n_in) * static_cast<unsigned>(es)), 0x00020000);
  const int sh = es >> 1;  // 1 / 2 / 4 bytes: shift 0 / 1 / 2
  unsigned boffs[16];
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const float x = X[t], y = Y[t], z = Z[t];
    const int ix = static_cast<int>(rintf(x)), iy = static_cast<int>(rintf(y)), iz = static_cast<int>(rintf(z));
    const bool ok = (static_cast<unsigned>(ix) <= Hx) & (static_cast<unsigned>(iy) <= Hy) & (static_cast<unsigned>(iz) <= Hz) &
                    !__builtin_isunordered(x, y) & (z == z);
    int off;
    {  // (ix J + iy) (K es) + iz es: two 24-bit multiply-adds (the launch gate: I J <= 2^24, the channel below 2^32 bytes)
      int t1;
      asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t1) : "v"(ix), "v"(J), "v"(iy));
      const int izb = iz << sh;
      asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(t1), "v"(Kes), "v"(izb));
    }
    boffs[t] = ok ? static_cast<unsigned>(off) : 0xFFFFFFFFu;
    asm volatile("" : "+v"(boffs[t]));
    __builtin_amdgcn_sched_barrier(0);
  }
  // (exactly sixteen vector-memory instructions whatever the element size: the wait above counts them)
  if (ka->ablate & 256) {  // (measurement: TIO_TILE_ABLATE=256 — no label loads; sixteen loads of offset 0 keep the count)
#pragma unroll
    for (int t = 0; t < 16; t++) vlab[t] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, 0u, 0, 0) + boffs[t];
  } else if (es == 1) {
#pragma unroll
    for (int t = 0; t < 16; t++) vlab[t] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, boffs[t], 0, 0);
  } else if (es == 2) {
#pragma unroll
    for (int t = 0; t < 16; t++) vlab[t] = __builtin_amdgcn_raw_buffer_load_b16(rsrc, boffs[t], 0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < 16; t++) vlab[t] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, boffs[t], 0, 0);
  }
}

template <bool GUARD>
__device__ __forceinline__ void lean_exact_label_store(const unsigned (&vlab)[16], int b, int i_begin, int col_off, int i_count, bool col_active) {
  typedef __attribute__((address_space(4))) const LeanArgs* const_args_ptr;
  typedef __attribute__((address_space(1))) char* global_char_ptr;
  const_args_ptr ka = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const int es = ka->lab_es;
  const int sh = es >> 1;
  const int64_t slab = static_cast<int64_t>(ka->Jo) * ka->Ko;
  const int64_t n_out = static_cast<int64_t>(ka->Io) * slab;
  global_char_ptr out_t = (global_char_ptr)(static_cast<char*>(ka->lab_out) + (b * n_out + static_cast<int64_t>(i_begin) * slab) * es);
  const int64_t slab_b = slab * es;
  unsigned row_off = static_cast<unsigned>(col_off) << sh;
#define TIO_LE_LABEL_STORES(BITS_T)                                                             \
  {                                                                                             \
    typedef __attribute__((address_space(1))) BITS_T* global_bits_ptr;                          \
    asm volatile("" : "+v"(row_off));                                                           \
    if constexpr (!GUARD) {                                                                     \
      _Pragma("unroll") for (int t = 0; t < 16; t++) {                                          \
        *(global_bits_ptr)(out_t + row_off) = static_cast<BITS_T>(vlab[t]);                     \
        out_t += slab_b;                                                                        \
        asm volatile("" : "+s"(out_t));                                                         \
      }                                                                                         \
    } else if (col_active) {                                                                    \
      for (int t = 0; t < i_count; t++) {                                                       \
        unsigned w = vlab[0];                                                                   \
        _Pragma("unroll") for (int u = 1; u < 16; u++) w = (t == u) ? vlab[u] : w;              \
        *(global_bits_ptr)(out_t + row_off) = static_cast<BITS_T>(w);                           \
        out_t += slab_b;                                                                        \
      }                                                                                         \
    }                                                                                           \
  }
  if (ka->ablate & 512) return;  // (measurement: TIO_TILE_ABLATE=512 — no label stores)
  if (es == 1) TIO_LE_LABEL_STORES(uint8_t)
  else if (es == 2) TIO_LE_LABEL_STORES(uint16_t)
  else TIO_LE_LABEL_STORES(uint32_t)
#undef TIO_LE_LABEL_STORES
}

template <bool ELASTIC_POSSIBLE, bool EXACT_LERP, bool FOLD_MIN, bool MULTI, bool PAIR = false, bool LABEL = false>
__device__ __forceinline__ void lean_exact_brick(const LeanArgs& a, const unsigned brick, float* s_tile) {
  static_assert(!PAIR || (!FOLD_MIN && !MULTI), "the pair kernel has no folded minimum and no passes");
  static_assert(!LABEL || (!FOLD_MIN && !MULTI), "a label channel rides along launches without a folded minimum and without passes");
  constexpr int TI = 16, TJ = 16, TK = 16, NW = 4;
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;

  // every argument the road to the first DMA needs, in scalar registers NOW (resample_planned_lean_kernel)
  // (round 5: the first build of this kernel fetched mapping_batched, tile_floats, the strides and the output shape one by one,
  // each behind its own `s_waitcnt lgkmcnt(0)` — nine scalar round trips between entry and the first DMA instruction)
  // (round 6, measured and not kept: the descriptor's address from PRELOADED leading kernel arguments — kernarg preload,
  // -mllvm -amdgpu-kernarg-preload-count=9 — same-box A/B +0.7 % / 0.0 % / -0.4 % on the affine / elastic / fused launch against
  // -1.0 / -1.3 / -1.9 % without it: the preload is not free at wave launch)
  {
    const int* plan_p = a.plan; const float* in_p = a.in; const float* map_p = a.mapping; float* out_p = a.out; const float* fill_p = a.fill;
    asm volatile("" ::"s"(a.n_items), "s"(a.bricks_per_element), "s"(a.bpe_magic), "s"(plan_p), "s"(in_p), "s"(map_p), "s"(a.B), "s"(a.I), "s"(a.J), "s"(a.K),
                 "s"(a.mapping_batched), "s"(a.tile_floats), "s"(a.in_stride), "s"(a.out_stride), "s"(a.Io), "s"(a.Jo), "s"(a.Ko), "s"(a.interleave),
                 "s"(out_p), "s"(fill_p), "s"(a.hx), "s"(a.hy), "s"(a.hz), "s"(a.affine_first));
    if constexpr (ELASTIC_POSSIBLE) {  // (what the control-point prelude reads: ten more round trips in the first build)
      const float* cp_p = a.cp;
      asm volatile("" ::"s"(cp_p), "s"(a.cp_batched), "s"(a.ni), "s"(a.nj), "s"(a.nk), "s"(a.sci), "s"(a.scj), "s"(a.sck));
    }
  }
  const int b = static_cast<int>(fastdiv_exact(brick, a.bpe_magic, a.bricks_per_element));
  // descriptor and the element's UNSCALED mapping (the planner's copy is scaled by the axis ratios), requested together
  const_int_ptr d = (const_int_ptr)(a.plan + a.B * 16) + static_cast<size_t>(brick) * kDescInts;
  const_float_ptr mp = (const_float_ptr)(a.mapping) + (a.mapping_batched ? b * 12 : 0);
  const int kind_w = d[0];
  StreamBox bx;
  bx.bx0 = d[1]; bx.by0 = d[2]; bx.za = d[3]; bx.Lx = d[4]; bx.Ly = d[5]; bx.cpr = d[6];
  const int i_begin = d[11], j_lo = d[12], k_lo = d[13];
  const int elastic_w = d[14];
  int multi_w = 0, plan_tile = 0;
  if constexpr (MULTI) {
    multi_w = d[15];
    plan_tile = ((const_int_ptr)(a.plan))[b * 16 + 13];  // (the tile the planner sized the boxes for)
  }
  float m[12];
#pragma unroll
  for (int q = 0; q < 12; q++) m[q] = mp[q];
  asm volatile("" ::"s"(kind_w), "s"(i_begin), "s"(j_lo), "s"(k_lo), "s"(elastic_w));  // (the whole descriptor behind ONE wait)
  if constexpr (MULTI) asm volatile("" ::"s"(multi_w), "s"(plan_tile));
  const bool elastic = ELASTIC_POSSIBLE && elastic_w != 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid & (TK - 1), tj = tid / TK;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float* in_chan = a.in + static_cast<int64_t>(b) * a.in_stride;
  char* out_chan = reinterpret_cast<char*>(a.out + static_cast<int64_t>(b) * a.out_stride);

  int kind = kind_w & 0xFF;
  // round 6: a brick whose box exceeds the tile, bounded again by the planner in halves / quarters of its planes (kDescMulti:
  // [1..6] hold pass 0's box, [15] the number of passes and every pass's state; the other passes' boxes are fetched when
  // their turn comes).  Pass 0 is requested, and waited for, exactly like the box of a one-pass brick.  The planner has checked
  // for every pass what this kernel re-checks for a one-pass box; its word holds if the plan was sized for a tile no larger
  // than this launch's (a plan made AHEAD may have been made for another road).
  const bool multi = MULTI && kind == kDescMulti;
  if constexpr (!MULTI) {
    if (kind == kDescMulti) return;  // (resample_lean_exact_multi_kernel's)
  }
  int nsplit = 1, states = 0;  // states: 4 bits per pass (kPassStaged = 0: a one-pass brick is "all staged")
  if constexpr (MULTI) {
    if (multi) {
      nsplit = multi_w & 0xF; states = multi_w >> 8;
      kind = plan_tile <= a.tile_floats ? kDescStaged : kDescSlow;
    }
  }
  // (a plan made AHEAD may have been sized for another road's tile: a box beyond THIS launch's tile takes the per-voxel road)
  // ... and so does a box whose taps' LDS addresses cannot be formed from absolute indices in float32 (box_address_fits: a
  // volume thousands of voxels long)
  if (!multi && kind == kDescStaged && (static_cast<int64_t>(bx.Lx) * bx.Ly * (bx.cpr * 4) > static_cast<int64_t>(a.tile_floats) ||
                                        !box_address_fits(bx.bx0, bx.by0, bx.za, bx.Lx, bx.Ly, bx.cpr)))
    kind = kDescSlow;
  if (kind == kDescSlow) states = 0x2222;  // (every plane on the per-voxel road)
  const int span_shift = nsplit == 4 ? 2 : (nsplit == 2 ? 3 : 4);  // planes per pass = 1 << span_shift
  bx.kind = kind; bx.interior = kind_w >> 8;
  BoxDmaStepper<NW> dma;
  dma.left = 0;
  if (kind == kDescStaged && (states & 0xF) == kPassStaged) dma.init(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane);

  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const bool full = (i_count == TI) & (nv == TJ) & (nw == TK);  // block uniform
  // out-of-range threads shadow the last valid column (coordinates of real voxels: the planned box covers them) and never store
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
  const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
  const unsigned urow = static_cast<unsigned>(col_off) * 4u;
  bool has_fill = a.fill != nullptr;
  float fillv = has_fill ? ((const_float_ptr)a.fill)[0] : 0.0f;
  const float hx = a.hx, hy = a.hy, hz = a.hz;

  const bool track = FOLD_MIN && a.min_keys != nullptr && b == 0;  // block uniform
  uint32_t kmin = 0xFFFFFFFFu;
  auto publish_min = [&]() {  // (one returnless atomic per wave, spread over kMinSlots addresses by brick: resample_planned_kernel)
    const uint32_t wmin = wave_min_u32(kmin);
    if (lane == 0 && wmin != 0xFFFFFFFFu)
      __hip_atomic_fetch_min(a.min_keys + (brick & (kMinSlots - 1)), wmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (kind == kDescGated || kind == kDescOutside) {  // gated-out element: bit-exact copy; nothing of the volume in sight: fill (or 0)
    if (col_active) {
      for (int t = i_begin; t < i_begin + i_count; t++) {
        const float val = kind == kDescGated ? in_chan[static_cast<int64_t>(t) * slab + col_off] : fillv;
        *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = val;
        kmin = min(kmin, float_to_key(val));
      }
    }
    if constexpr (LABEL) {
      if (col_active) {  // gated-out element: the label map's bits copied; nothing of the volume in sight: zero padding
        const int es = a.lab_es;
        const int64_t n_out_l = static_cast<int64_t>(a.Io) * slab;
        for (int t = i_begin; t < i_begin + i_count; t++) {
          const int64_t e = b * n_out_l + static_cast<int64_t>(t) * slab + col_off;
          lean_label_store(a.lab_out, es, e, kind == kDescGated ? lean_label_load(a.lab_in, es, e) : 0u);
        }
      }
    }
    if constexpr (PAIR) {
      if (col_active) {
        const float* in2_chan = a.in2 + static_cast<int64_t>(b) * a.in_stride2;
        char* out2_chan = reinterpret_cast<char*>(a.out2 + static_cast<int64_t>(b) * a.out_stride2);
        const float fillv2 = a.fill2 != nullptr ? ((const_float_ptr)a.fill2)[0] : 0.0f;
        for (int t = i_begin; t < i_begin + i_count; t++)
          *reinterpret_cast<float*>(out2_chan + t * slab_b + urow) = kind == kDescGated ? in2_chan[static_cast<int64_t>(t) * slab + col_off] : fillv2;
      }
    }
    if (track) publish_min();
    return;
  }

  const float cj = static_cast<float>(j_lo + jv), ck = static_cast<float>(k_lo + kw);
  const float* cp = elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      lj = lerp_index(j_lo + jv, a.nj, a.Jo, a.scj);
      lk = lerp_index(k_lo + kw, a.nk, a.Ko, a.sck);
    }
  }
  const int i_last = i_begin + i_count - 1;
  // the plane lerps of the control grid: lane t computes plane t's (block uniform per plane)
  Lerp1D li_lane{0, 0, 1.0f, 0.0f};
  int ia = 0, ib = 0;
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      li_lane = lerp_index(min(i_begin + (lane & (TI - 1)), i_last), a.ni, a.Io, a.sci);
      ia = __builtin_amdgcn_readlane(li_lane.i0, 0);
      ib = __builtin_amdgcn_readlane(li_lane.i1, TI - 1);
    }
  }

  // Bricks the planner could not stage (box beyond the LDS budget, non-finite geometry) and — never on the planned road, whose
  // gate asks for control cells at least a brick wide — bricks over more than three control planes: every plane voxel by voxel
  if (elastic && ib - ia > 2) states = 0x2222;
  bool any_staged = states == 0;
  if constexpr (MULTI) {
    for (int pass = 0; pass < nsplit; pass++) any_staged |= ((states >> (4 * pass)) & 0xF) == kPassStaged;
  }
  if (!any_staged) {
    lean_exact_slow_planes<ELASTIC_POSSIBLE, false, LABEL>(brick, states, span_shift, kmin);
    if constexpr (PAIR) lean_exact_slow_planes<ELASTIC_POSSIBLE, true>(brick, states, span_shift, kmin);
    if (track) publish_min();
    return;
  }

  // ---- phase A: the reference's coordinates of this column's 16 planes, between the DMA instructions ------------------
  if (a.interleave == 2 && bx.interior) {  // A/B (TIO_LEAN_INTERLEAVE=2): the box first, the control points behind it
    while (dma.left > 0) dma.template issue<true>(lane);
  }
  CtlPlanes P{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {  // <= 3 control planes under the brick, lerped along J and K once per column (36 control values, cache resident)
      // (all three planes UNCONDITIONALLY — a plane beyond the last one the brick touches is clamped and never selected — and
      // every member passed through an empty asm: members written under a branch keep the struct in memory, and the optimiser
      // then turns the selects of phase A into a dynamically indexed load of that stack object: 40 bytes of scratch per lane)
      const int s_i = a.nj * a.nk * 3, s_j = a.nk * 3;
      float pa[3], pb[3], pc[3];
      cp_plane(cp, ia, s_i, s_j, lj, lk, pa);
      cp_plane(cp, min(ia + 1, a.ni - 1), s_i, s_j, lj, lk, pb);
      cp_plane(cp, min(ia + 2, a.ni - 1), s_i, s_j, lj, lk, pc);
      P.a_i = pa[0]; P.a_j = pa[1]; P.a_k = pa[2];
      P.b_i = pb[0]; P.b_j = pb[1]; P.b_k = pb[2];
      P.c_i = pc[0]; P.c_j = pc[1]; P.c_k = pc[2];
      asm volatile("" : "+v"(P.a_i), "+v"(P.a_j), "+v"(P.a_k), "+v"(P.b_i), "+v"(P.b_j), "+v"(P.b_k), "+v"(P.c_i), "+v"(P.c_j), "+v"(P.c_k));
    }
  }
  // a box that sticks out of the volume: per-row checks and zeroed chunks — all of it now, nothing interleaved (behind the
  // control-point loads: a wave's vector-memory operations return in order, and phase A needs the control planes first)
  if (!bx.interior) {
    while (dma.left > 0) dma.template issue<false>(lane);
  }
  if (!a.interleave) {  // A/B (TIO_LEAN_INTERLEAVE=0): the whole box requested before the first coordinate is formed
    while (dma.left > 0) dma.template issue<true>(lane);
  }
  float X[TI], Y[TI], Z[TI];
  const bool ident = (m[0] == 1.0f) & (m[1] == 0.0f) & (m[2] == 0.0f) & (m[3] == 0.0f) & (m[4] == 0.0f) & (m[5] == 1.0f) &
                     (m[6] == 0.0f) & (m[7] == 0.0f) & (m[8] == 0.0f) & (m[9] == 0.0f) & (m[10] == 1.0f) & (m[11] == 0.0f);
  // (the launch gate guarantees short_div, and unit spacing whenever control points are present)
  bool done = false;
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      if (ident) lean_exact_planes<1, true, true, true, NW>(m, a, i_begin, i_last, cj, ck, li_lane, ia, P, dma, lane, X, Y, Z);
      else if (a.affine_first) lean_exact_planes<2, true, true, true, NW>(m, a, i_begin, i_last, cj, ck, li_lane, ia, P, dma, lane, X, Y, Z);
      else lean_exact_planes<3, true, true, true, NW>(m, a, i_begin, i_last, cj, ck, li_lane, ia, P, dma, lane, X, Y, Z);
      done = true;
    }
  }
  if (!done) lean_exact_planes<0, true, true, true, NW>(m, a, i_begin, i_last, cj, ck, li_lane, ia, P, dma, lane, X, Y, Z);
  while (dma.left > 0) dma.template issue<true>(lane);  // (interior boxes with more instructions per wave than planes)

  TileAddr ta;
  auto set_tile_addr = [&](const StreamBox& box) {
    ta.ox = static_cast<float>(box.bx0); ta.oy = static_cast<float>(box.by0); ta.oz = static_cast<float>(box.za);
    ta.sYb = box.cpr * 16; ta.sXb = box.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
    ta.sYbf = static_cast<float>(ta.sYb); ta.sXbf = static_cast<float>(ta.sXb);
    ta.base_f = static_cast<float>(static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile)));
    ta.c_f = ta.base_f - ta.ox * ta.sXbf - ta.oy * ta.sYbf - 4.0f * ta.oz;  // (exact: box_address_fits held above / in the planner)
  };
  set_tile_addr(bx);

  unsigned vlab[16];
  (void)vlab;
  if constexpr (LABEL) {
    lean_exact_label_issue(X, Y, Z, b, vlab);
    tile_dma_wait_keep16();
  } else {
    tile_dma_wait_all();
  }
  __syncthreads();

  // ---- phase B: sample.  The fill rule only matters where a tap can leave the volume: interior boxes never, the others
  // group by group and wave by wave (the masked code runs only in waves one of whose four voxels has a tap outside)
  char* out_t = out_chan + static_cast<int64_t>(i_begin) * slab_b;
  bool may_leave = has_fill & !bx.interior;  // block uniform
  if constexpr (!MULTI) {
#define TIO_LE_COORDS4                                                          \
  const float x4[4] = {X[4 * q], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3]};     \
  const float y4[4] = {Y[4 * q], Y[4 * q + 1], Y[4 * q + 2], Y[4 * q + 3]};     \
  const float z4[4] = {Z[4 * q], Z[4 * q + 1], Z[4 * q + 2], Z[4 * q + 3]};
#define TIO_LE_GROUPS(TRACK)                                                                                                              \
  _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                                         \
    TIO_LE_COORDS4                                                                                                                        \
    bool masked = false;                                                                                                                  \
    if (may_leave) masked = __builtin_amdgcn_ballot_w64(lean_exact_group_leaves(x4, y4, z4, hx, hy, hz)) != 0ull;                         \
    if (masked)                                                                                                                           \
      lean_exact_group<EXACT_LERP, true, false, TRACK>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, true, fillv, kmin);   \
    else                                                                                                                                  \
      lean_exact_group<EXACT_LERP, false, false, TRACK>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, false, fillv, kmin); \
  }
    // partial bricks (a volume edge that is not a multiple of 16: rare): one copy, predicated stores, the mask wherever the image has a fill rule
#define TIO_LE_GROUPS_GUARDED(TRACK)                                                                                                      \
  _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                                         \
    TIO_LE_COORDS4                                                                                                                        \
    lean_exact_group<EXACT_LERP, true, true, TRACK>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, has_fill, fillv, kmin); \
  }
    if constexpr (PAIR) {
      // (the first channel's groups as the one-channel kernel's, then — everything it needs of the launch re-read from the argument
      // block through a pointer the optimiser cannot see through, as the passes of a multi-pass brick do — the second channel)
      if (full) { TIO_LE_GROUPS(false) } else { TIO_LE_GROUPS_GUARDED(false) }
      __syncthreads();
      typedef __attribute__((address_space(4))) const LeanArgs* const_args_ptr;
      const_args_ptr ka = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      const_int_ptr d2 = (const_int_ptr)(ka->plan + ka->B * 16) + static_cast<size_t>(brick) * kDescInts;
      StreamBox nb;
      nb.bx0 = d2[1]; nb.by0 = d2[2]; nb.za = d2[3]; nb.Lx = d2[4]; nb.Ly = d2[5]; nb.cpr = d2[6];
      nb.interior = d2[0] >> 8; nb.kind = kDescStaged;
      BoxDmaStepper<NW> step;
      step.init(s_tile, ka->in2 + static_cast<int64_t>(b) * ka->in_stride2, nb, ka->I, ka->J, ka->K, wave, lane);
      if (nb.interior) { while (step.left > 0) step.template issue<true>(lane); }
      else { while (step.left > 0) step.template issue<false>(lane); }
      has_fill = ka->fill2 != nullptr;
      fillv = has_fill ? ((const_float_ptr)ka->fill2)[0] : 0.0f;
      may_leave = has_fill & !nb.interior;
      out_t = reinterpret_cast<char*>(ka->out2 + static_cast<int64_t>(b) * ka->out_stride2) + static_cast<int64_t>(i_begin) * slab_b;
      tile_dma_wait_all();
      __syncthreads();
      if (full) { TIO_LE_GROUPS(false) } else { TIO_LE_GROUPS_GUARDED(false) }
      if constexpr (LABEL) {
        if (full) lean_exact_label_store<false>(vlab, b, i_begin, col_off, i_count, col_active);
        else lean_exact_label_store<true>(vlab, b, i_begin, col_off, i_count, col_active);
      }
    } else if constexpr (LABEL) {
      if (full) { TIO_LE_GROUPS(false) } else { TIO_LE_GROUPS_GUARDED(false) }
      if (full) lean_exact_label_store<false>(vlab, b, i_begin, col_off, i_count, col_active);
      else lean_exact_label_store<true>(vlab, b, i_begin, col_off, i_count, col_active);
    } else if (FOLD_MIN && track) {
      if (full) { TIO_LE_GROUPS(true) } else { TIO_LE_GROUPS_GUARDED(true) }
      publish_min();
    } else {
      if (full) { TIO_LE_GROUPS(false) } else { TIO_LE_GROUPS_GUARDED(false) }
    }
#undef TIO_LE_GROUPS_GUARDED
#undef TIO_LE_COORDS4
#undef TIO_LE_GROUPS
    return;
  }
  // the next pass of a multi-pass brick: every wave is done with the tile, then the pass's box is requested and waited for
  // (nothing of this block overlaps the wait — the other resident blocks do)
  // (what it needs of the launch's arguments is re-read from the argument block through a pointer the optimiser cannot see
  // through: kept live across the sampling groups instead, those ~20 scalars pushed 26 - 58 others out of the register file)
  auto next_pass = [&](int pass) {
    __syncthreads();
    typedef __attribute__((address_space(4))) const LeanArgs* const_args_ptr;
    const_args_ptr ka = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const_int_ptr r8 = (const_int_ptr)(ka->plan + ka->B * 16 + static_cast<size_t>(ka->n_items) * kDescInts) +
                       (static_cast<size_t>(brick) * kPassesPerBrick + pass) * kPassInts;
    StreamBox nb;
    nb.bx0 = r8[0]; nb.by0 = r8[1]; nb.za = r8[2]; nb.Lx = r8[3]; nb.Ly = r8[4]; nb.cpr = r8[5];
    nb.interior = r8[6] >> 8; nb.kind = kDescStaged;
    BoxDmaStepper<NW> step;  // (its own stepper: re-using phase A's keeps twenty fields alive across the groups)
    step.init(s_tile, ka->in + static_cast<int64_t>(b) * ka->in_stride, nb, ka->I, ka->J, ka->K, wave, lane);
    if (nb.interior) { while (step.left > 0) step.template issue<true>(lane); }
    else { while (step.left > 0) step.template issue<false>(lane); }
    set_tile_addr(nb);
    may_leave = (ka->fill != nullptr) & !nb.interior;
    tile_dma_wait_all();
    __syncthreads();
  };
  const bool track_min = FOLD_MIN && track;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (multi) {  // (block uniform; one-pass bricks in this kernel's list — none today — : states == 0, nothing of this)
      const int pass = (4 * q) >> span_shift;
      if (((states >> (4 * pass)) & 0xF) != kPassStaged) {  // this group's planes were written above
        out_t += 4 * slab_b;
        continue;
      }
      if (q > 0 && ((4 * q) & ((1 << span_shift) - 1)) == 0) next_pass(pass);
    }
    const float x4[4] = {X[4 * q], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3]};
    const float y4[4] = {Y[4 * q], Y[4 * q + 1], Y[4 * q + 2], Y[4 * q + 3]};
    const float z4[4] = {Z[4 * q], Z[4 * q + 1], Z[4 * q + 2], Z[4 * q + 3]};
    if (!full) {
      // partial bricks (a volume edge that is not a multiple of 16: rare): one copy, predicated stores, the mask wherever the image has a fill rule
      if (track_min) lean_exact_group<EXACT_LERP, true, true, FOLD_MIN>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, has_fill, fillv, kmin);
      else lean_exact_group<EXACT_LERP, true, true, false>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, has_fill, fillv, kmin);
      continue;
    }
    bool masked = false;
    if (may_leave) masked = __builtin_amdgcn_ballot_w64(lean_exact_group_leaves(x4, y4, z4, hx, hy, hz)) != 0ull;
    if (track_min) {
      if (masked) lean_exact_group<EXACT_LERP, true, false, FOLD_MIN>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, true, fillv, kmin);
      else lean_exact_group<EXACT_LERP, false, false, FOLD_MIN>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, false, fillv, kmin);
    } else {
      if (masked) lean_exact_group<EXACT_LERP, true, false, false>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, true, fillv, kmin);
      else lean_exact_group<EXACT_LERP, false, false, false>(x4, y4, z4, ta, out_t, urow, slab_b, 4 * q, i_count, col_active, hx, hy, hz, false, fillv, kmin);
    }
  }
  // the planes of the passes that were NOT staged (a multi-pass brick one of whose parts still does not fit, or sees nothing
  // of the volume): voxel by voxel, behind everything else
  if (multi && states != 0) lean_exact_slow_planes<ELASTIC_POSSIBLE>(brick, states, span_shift, kmin);
  if (track_min) publish_min();
}

template <bool ELASTIC_POSSIBLE, bool EXACT_LERP, int WAVES_PER_SIMD, bool FOLD_MIN = false>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void resample_lean_exact_kernel(const LeanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lean_exact_brick<ELASTIC_POSSIBLE, EXACT_LERP, FOLD_MIN, false>(a, xcd_remap(blockIdx.x, static_cast<unsigned>(a.n_items)), smem);
}

// ... two channels of one geometry per block (PAIR above): what a subject with several float32 images — or an image with several
// channels — launches instead of one kernel per channel when the launch has no multi-pass bricks and no folded minimum
template <bool ELASTIC_POSSIBLE, bool EXACT_LERP>
__global__ __launch_bounds__(256, 3) void resample_lean_exact_pair_kernel(const LeanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lean_exact_brick<ELASTIC_POSSIBLE, EXACT_LERP, false, false, true>(a, xcd_remap(blockIdx.x, static_cast<unsigned>(a.n_items)), smem);
}

// ... one or two float32 channels AND a nearest-neighbour label channel per block (LABEL above): config 5's whole call in one launch
template <bool ELASTIC_POSSIBLE, bool EXACT_LERP, bool PAIR>
__global__ __launch_bounds__(256, 3) void resample_lean_exact_label_kernel(const LeanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lean_exact_brick<ELASTIC_POSSIBLE, EXACT_LERP, false, false, PAIR, true>(a, xcd_remap(blockIdx.x, static_cast<unsigned>(a.n_items)), smem);
}

// ... the same with the pass switches compiled in: what a launch MOST of whose bricks need passes takes (the caller's hint
// TIO_GEOM_MOSTLY_LARGE_BOXES: rotations beyond ~15 degrees about all axes) — one block per brick, no list, no second kernel; its
// one-pass bricks pay the pass logic's registers (+2 ... +7 %, measured: the reason it is not the only kernel)
template <bool ELASTIC_POSSIBLE, bool EXACT_LERP, bool FOLD_MIN = false>
__global__ __launch_bounds__(256, 3) void resample_lean_exact_all_kernel(const LeanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lean_exact_brick<ELASTIC_POSSIBLE, EXACT_LERP, FOLD_MIN, true>(a, xcd_remap(blockIdx.x, static_cast<unsigned>(a.n_items)), smem);
}

// The multi-pass bricks of the launch (round 6), behind resample_lean_exact_kernel on the stream: a few blocks per CU are
// WALKERS of the planner's list (header in front of the plan: [0] the bricks listed, [1] the walkers' cursor, [2] the walkers
// that are done), taking one brick at a time from the cursor — an atomic per brick: the bricks cost two to four passes and some
// of them a per-voxel part, a static share leaves the launch waiting for its unluckiest walker.  At the bench's ranges the list
// holds the 1.6 % of the bricks that sampled voxel by voxel from global memory until round 5 — every walker takes one or
// none; beyond ~12 degrees about all axes it holds most bricks.  The last walker to finish zeroes cursor and done count — and
// the list's length when the host says this launch is the last one that reads this plan (a.last_use: the launches of a call's
// channels share a plan) — so the planner starts from zeros without a memset on the stream.
// (Both kinds of blocks in ONE launch — walkers behind the bricks' own blocks — were built and measured: +9 % on the bench's
// affine launch, the register allocation of the walker's body leaks into the one-brick body; profiles/r06_resample.md.)
template <bool ELASTIC_POSSIBLE, bool EXACT_LERP, bool FOLD_MIN = false>
__global__ __launch_bounds__(256, 3) void resample_lean_exact_multi_kernel(const LeanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* head = const_cast<int*>(a.plan) - kPlanHeaderInts;
  const int* list = a.plan + a.B * 16 + static_cast<size_t>(a.n_items) * (kDescInts + kPassInts * kPassesPerBrick);
  const int count = min(__builtin_nontemporal_load(head), a.n_items);
  int* mailbox = reinterpret_cast<int*>(smem);  // (the tile is nobody's between two bricks)
  while (count > 0) {
    if (threadIdx.x == 0) mailbox[0] = atomicAdd(head + 1, 1);
    __syncthreads();
    const int at = __builtin_amdgcn_readfirstlane(mailbox[0]);
    __syncthreads();
    if (at >= count) break;
    const int brick = __builtin_amdgcn_readfirstlane(list[at]);
    if (brick >= 0 && brick < a.n_items)
      lean_exact_brick<ELASTIC_POSSIBLE, EXACT_LERP, FOLD_MIN, true>(a, static_cast<unsigned>(brick), smem);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(head + 2, 1) == static_cast<int>(gridDim.x) - 1) {
      head[1] = 0; head[2] = 0;
      if (a.last_use) head[0] = 0;
    }
  }
}

}  // namespace tio
