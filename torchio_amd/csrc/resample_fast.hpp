// resample_fast.hpp — the TIO_PRECISION_FAST path of tio_resample3d: launches of float32 trilinear images whose
// caller accepts the reference's own tolerance for intensities (1e-4 relative, BASELINE.json north_star) instead of
// the bit-exact reproduction of its float32 operation sequence.
//
// What it does differently from the exact brick kernel (resample_tile.hpp):
//   * coordinates as a LINE in the plane index per control cell, box-relative (x(t) = A + t B, the constant formed in
//     float64): 3 fma per voxel, no coordinate arrays, error ~2e-6 voxel on top of the reference's own rounding;
//   * the staged box from the <= 27 VERTICES of (planes x control cells) — the coordinate map is multilinear on each
//     such sub-box, so its extremes sit on vertices — instead of per-voxel min / max tracking and a block reduction;
//   * that planning is a kernel of its own, ONE THREAD PER BRICK (plan_bricks_kernel, microseconds): 16 dwords per
//     brick (box, float64-formed line constants, kind) and the scaled mapping per batch element.  The sampling
//     kernel (resample_planned_kernel, one block per 16^3 brick) starts with one s_load_dwordx16, issues its LDS-DMA
//     (scalar-addressed: one wave instruction per group of rows of one x-plane) and forms the per-column line while
//     the brick is on its way: no decode, no vertex evaluation, no reductions, no LDS besides the tile;
//   * seven fma lerps, the fill rule's in-bounds weight in its separable form and only in waves that really have a
//     column leaving the volume (a monotone line is interior when its two end planes are).
//
// Round-2 history (profiles/r02_resample_sq.md has every number): five other structures around the same arithmetic
// were built and measured — lean bricks with in-kernel planning, persistent streaming rings, pipelined persistent
// bricks, wide (8 / 16 wave) blocks — none beat the brick kernel's FAST instantiation (0.437 ms on the bench launch);
// the shader-clock traces showed why (a block's life is kernarg -> mapping -> vertices -> reductions -> barrier -> DMA
// -> barrier -> sampling, and with LDS capping a CU at three bricks the throughput is 3 / that latency).  Moving the
// planning out of the sampling kernel is what shortened it: 0.372 ms affine / 0.42 ms elastic (36 % / 32 % of 8 TB/s).
// Those kernels are gone from the tree; commit 45474dd has them all.
#pragma once

namespace tio {

typedef __attribute__((address_space(3))) const float* fast_lds_ptr;
typedef __attribute__((address_space(3))) float* fast_lds_wptr;

// continuous position along one control axis -> cell and weight (ATen's align_corners lerp,
// extended to non-integer positions; NaN positions land in cell 0 with weight 0)
__device__ __forceinline__ void fast_axis(float src, int n, int& i0, int& i1, float& l) {
  const float c = fminf(fmaxf(floorf(src), 0.0f), static_cast<float>(n > 1 ? n - 2 : 0));
  i0 = static_cast<int>(c);
  i1 = min(i0 + 1, n - 1);
  l = fminf(fmaxf(src - c, 0.0f), 1.0f);
}

template <typename CP>
struct FastFrameT {
  float m[12];            // voxel mapping (rows scaled by the normalisation ratio of the axis)
  double c[3];            // mapping applied to (0, j_lo, k_lo)
  int j_lo, k_lo;
  bool elastic, affine_first;
  CP cp;                  // control points of this batch element (LDS copy in the streaming kernel, global in the brick kernel)
  int ni, nj, nk;
  float sci, scj, sck;    // control-grid lerp scales
  float dsc[3];           // displacement scale: 1 / spacing (times the axis ratio when affine_first)
};
typedef FastFrameT<fast_lds_ptr> FastFrame;

// displacement (already in voxels) at the volume position (i, j, k), any of them fractional
template <typename CP>
__device__ __forceinline__ void fast_displacement(const FastFrameT<CP>& f, float pi, float pj, float pk, float (&d)[3]) {
  int i0, i1, j0, j1, k0, k1;
  float li, lj, lk;
  fast_axis(f.sci * pi, f.ni, i0, i1, li);
  fast_axis(f.scj * pj, f.nj, j0, j1, lj);
  fast_axis(f.sck * pk, f.nk, k0, k1, lk);
  const int s_i = f.nj * f.nk * 3, s_j = f.nk * 3;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const CP p = f.cp + e;
    const float a00 = p[i0 * s_i + j0 * s_j + k0 * 3], a01 = p[i0 * s_i + j0 * s_j + k1 * 3];
    const float a10 = p[i0 * s_i + j1 * s_j + k0 * 3], a11 = p[i0 * s_i + j1 * s_j + k1 * 3];
    const float b00 = p[i1 * s_i + j0 * s_j + k0 * 3], b01 = p[i1 * s_i + j0 * s_j + k1 * 3];
    const float b10 = p[i1 * s_i + j1 * s_j + k0 * 3], b11 = p[i1 * s_i + j1 * s_j + k1 * 3];
    const float a0 = __builtin_fmaf(lk, a01 - a00, a00), a1 = __builtin_fmaf(lk, a11 - a10, a10);
    const float b0 = __builtin_fmaf(lk, b01 - b00, b00), b1 = __builtin_fmaf(lk, b11 - b10, b10);
    const float av = __builtin_fmaf(lj, a1 - a0, a0), bv = __builtin_fmaf(lj, b1 - b0, b0);
    d[e] = __builtin_fmaf(li, bv - av, av) * f.dsc[e];
  }
}

// sampling coordinate (volume frame) of output position (u, j_lo + v, k_lo + w), u / v / w possibly fractional
template <typename CP>
__device__ __forceinline__ void fast_coord(const FastFrameT<CP>& f, float u, float v, float w, float& x, float& y, float& z) {
  float d[3] = {0.0f, 0.0f, 0.0f};
  if (f.elastic) fast_displacement(f, u, static_cast<float>(f.j_lo) + v, static_cast<float>(f.k_lo) + w, d);
  float eu = u, ev = v, ew = w, ax = 0.0f, ay = 0.0f, az = 0.0f;
  if (f.affine_first) { ax = d[0]; ay = d[1]; az = d[2]; } else { eu += d[0]; ev += d[1]; ew += d[2]; }
  x = __builtin_fmaf(f.m[0], eu, __builtin_fmaf(f.m[1], ev, __builtin_fmaf(f.m[2], ew, static_cast<float>(f.c[0])))) + ax;
  y = __builtin_fmaf(f.m[4], eu, __builtin_fmaf(f.m[5], ev, __builtin_fmaf(f.m[6], ew, static_cast<float>(f.c[1])))) + ay;
  z = __builtin_fmaf(f.m[8], eu, __builtin_fmaf(f.m[9], ev, __builtin_fmaf(f.m[10], ew, static_cast<float>(f.c[2])))) + az;
}

// first interior control-cell boundary of the index range [lo, hi] along an axis (lo itself when
// there is none); `dense` is raised when a second one follows
__device__ __forceinline__ float fast_breakpoint(float sc, int n, int lo, int hi, bool& dense) {
  if (n <= 2 || !(sc > 0.0f)) return static_cast<float>(lo);
  const float cell_lo = fminf(fmaxf(floorf(sc * static_cast<float>(lo)), 0.0f), static_cast<float>(n - 2));
  const float cell_hi = fminf(fmaxf(floorf(sc * static_cast<float>(hi)), 0.0f), static_cast<float>(n - 2));
  if (cell_hi <= cell_lo) return static_cast<float>(lo);
  dense |= cell_hi > cell_lo + 1.0f;
  return fminf(fmaxf((cell_lo + 1.0f) / sc, static_cast<float>(lo)), static_cast<float>(hi));
}

struct FastTaps {
  float v[8];
  float fx, fy, fz;
};

struct FastAddr {
  float sXf, sYf, base_f;
  unsigned sXb, sYb, sXYb;
};

__device__ __forceinline__ void fast_issue(FastTaps& ts, float x, float y, float z, const FastAddr& ta) {
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  ts.fx = x - x0; ts.fy = y - y0; ts.fz = z - z0;
  // LDS byte address in float32 (every term an integer far below 2^24: exact), one conversion
  const float af = __builtin_fmaf(x0, ta.sXf, __builtin_fmaf(y0, ta.sYf, __builtin_fmaf(z0, 4.0f, ta.base_f)));
  const unsigned addr = static_cast<unsigned>(static_cast<int>(af));
  fast_lds_ptr q00 = reinterpret_cast<fast_lds_ptr>(static_cast<uintptr_t>(addr));
  fast_lds_ptr q10 = reinterpret_cast<fast_lds_ptr>(static_cast<uintptr_t>(addr + ta.sXb));
  fast_lds_ptr q01 = reinterpret_cast<fast_lds_ptr>(static_cast<uintptr_t>(addr + ta.sYb));
  fast_lds_ptr q11 = reinterpret_cast<fast_lds_ptr>(static_cast<uintptr_t>(addr + ta.sXYb));
  ts.v[0] = q00[0]; ts.v[4] = q00[1];
  ts.v[1] = q10[0]; ts.v[5] = q10[1];
  ts.v[2] = q01[0]; ts.v[6] = q01[1];
  ts.v[3] = q11[0]; ts.v[7] = q11[1];
}

__device__ __forceinline__ float fast_finish(const FastTaps& ts) {
  const float a00 = __builtin_fmaf(ts.fz, ts.v[4] - ts.v[0], ts.v[0]);
  const float a10 = __builtin_fmaf(ts.fz, ts.v[5] - ts.v[1], ts.v[1]);
  const float a01 = __builtin_fmaf(ts.fz, ts.v[6] - ts.v[2], ts.v[2]);
  const float a11 = __builtin_fmaf(ts.fz, ts.v[7] - ts.v[3], ts.v[3]);
  const float b0 = __builtin_fmaf(ts.fy, a01 - a00, a00);
  const float b1 = __builtin_fmaf(ts.fy, a11 - a10, a10);
  return __builtin_fmaf(ts.fx, b1 - b0, b0);
}

// separable form of the in-bounds weight sum; (x0, y0, z0) = first-tap indices in the volume's frame
__device__ __forceinline__ float fast_mask(const FastTaps& ts, float x0, float y0, float z0, float hx, float hy, float hz) {
  const float mx = (((x0 >= 0.0f) & (x0 <= hx)) ? 1.0f - ts.fx : 0.0f) + (((x0 >= -1.0f) & (x0 <= hx - 1.0f)) ? ts.fx : 0.0f);
  const float my = (((y0 >= 0.0f) & (y0 <= hy)) ? 1.0f - ts.fy : 0.0f) + (((y0 >= -1.0f) & (y0 <= hy - 1.0f)) ? ts.fy : 0.0f);
  const float mz = (((z0 >= 0.0f) & (z0 <= hz)) ? 1.0f - ts.fz : 0.0f) + (((z0 >= -1.0f) & (z0 <= hz - 1.0f)) ? ts.fz : 0.0f);
  return mx * my * mz;
}

// n planes of this thread's column: coordinate(t) = (ax, ay, az) + t (bxs, bys, bzs), box-relative.
// G voxels have their LDS reads in flight before the first interpolation starts.  The output
// address is a block-uniform running pointer (one plane = slab_b bytes) + this thread's byte offset.
// TRACK: the ordered-integer key of the smallest value stored (the folded minimum of the launch's own output)
// MASKED: the fill rule.  The FAST in-bounds weight decides; a voxel whose weight is within `margin` of 1/2 additionally
// raises bit (plane_base + its index in the run) of `unsure`: the kernel's tail re-decides those with the reference's
// exact chain (resample_exact_chain.hpp).  margin < 0: nothing is ever recorded (A/B: TIO_FAST_FILL_RECHECK=0).
template <bool MASKED, int G, bool NOSTORE = false, bool TRACK = false>
__device__ __forceinline__ void fast_sample_run(int n, float ax, float ay, float az, float bxs, float bys, float bzs, const FastAddr& ta,
                                                char* out_generic, unsigned urow, int64_t slab_b, float ox, float oy, float oz, float hx,
                                                float hy, float hz, float fillv, uint32_t& kmin, float margin, int plane_base, unsigned& unsure) {
  // The running pointer passes through an empty asm (to pin it to scalar registers), which hides its address space
  // from the compiler: it MUST be typed global here, or the stores become flat_store_dword — flat operations count
  // on lgkmcnt as well, so every wait for the LDS taps would also wait for the previous voxels' stores to be
  // acknowledged by memory (measured: the whole sampling phase then runs at the store round-trip time).
  typedef __attribute__((address_space(1))) char* global_char_ptr;
  typedef __attribute__((address_space(1))) float* global_float_ptr;
  global_char_ptr out_t = (global_char_ptr)out_generic;
  const float bxg = static_cast<float>(G) * bxs, byg = static_cast<float>(G) * bys, bzg = static_cast<float>(G) * bzs;
  // full groups store unconditionally — one basic block, so the scheduler can interleave the G voxels' dependent
  // chains (a lone wave on its SIMD pays ~9 clocks per DEPENDENT instruction) — then one guarded tail group
#pragma unroll 1
  for (int tg = 0; tg < n; tg += G) {
    FastTaps ts[G];
    float x0s[G], y0s[G], z0s[G];
#pragma unroll
    for (int q = 0; q < G; q++) {
      const float qf = static_cast<float>(q);
      const float x = q == 0 ? ax : __builtin_fmaf(qf, bxs, ax);
      const float y = q == 0 ? ay : __builtin_fmaf(qf, bys, ay);
      const float z = q == 0 ? az : __builtin_fmaf(qf, bzs, az);
      fast_issue(ts[q], x, y, z, ta);
      if constexpr (MASKED) { x0s[q] = x - ts[q].fx; y0s[q] = y - ts[q].fy; z0s[q] = z - ts[q].fz; }
    }
    __builtin_amdgcn_sched_barrier(0);
    float vals[G];
    unsigned redo = 0u;  // voxels of this group the tail will store again (their first value never reaches the folded minimum)
#pragma unroll
    for (int q = 0; q < G; q++) {
      vals[q] = fast_finish(ts[q]);
      if constexpr (MASKED) {
        const float mk = fast_mask(ts[q], x0s[q] + ox, y0s[q] + oy, z0s[q] + oz, hx, hy, hz);
        vals[q] = (mk > 0.5f) ? vals[q] : fillv;
        // (a NaN weight compares false: the FAST answer — fill — stands; planes beyond the run are never raised)
        const bool again = (fabsf(mk - 0.5f) <= margin) & (tg + q < n);
        unsure |= again ? (1u << (plane_base + tg + q)) : 0u;
        if constexpr (TRACK) redo |= again ? (1u << q) : 0u;
      }
    }
    // (the row offset re-enters the block as a 32-bit register: instruction selection works block by block, and only a zero
    // extension it can SEE lets the store take the `scalar base + 32-bit vector offset` form — one 64-bit vector add per store less)
    unsigned row_off = urow;
    asm volatile("" : "+v"(row_off));
    if (tg + G <= n) {
#pragma unroll
      for (int q = 0; q < G; q++) {
        if (!NOSTORE || vals[q] == 1.2345e37f) *(global_float_ptr)(out_t + row_off) = vals[q];
        if constexpr (TRACK) kmin = ((redo >> q) & 1u) ? kmin : min(kmin, float_to_key(vals[q]));
        out_t += slab_b;
        asm volatile("" : "+s"(out_t));  // one running pointer (2 scalar adds per plane), not G precomputed ones
      }
    } else {
#pragma unroll
      for (int q = 0; q < G; q++) {
        if (tg + q < n) {
          unsigned tail_off = row_off;  // (its own block)
          asm volatile("" : "+v"(tail_off));
          if (!NOSTORE || vals[q] == 1.2345e37f) *(global_float_ptr)(out_t + tail_off) = vals[q];
          if constexpr (TRACK) kmin = ((redo >> q) & 1u) ? kmin : min(kmin, float_to_key(vals[q]));
        }
        out_t += slab_b;
        asm volatile("" : "+s"(out_t));
      }
    }
    ax += bxg; ay += byg; az += bzg;
    __builtin_amdgcn_sched_barrier(0);
  }
}


// ---- the coordinate line of one column through a run of planes inside ONE control cell ---------------------
// x(run0 + t) = A + t B, box-relative: C3 = mapping of (u_ref, j_lo, k_lo) relative to the box origin (float64 ->
// float32, block uniform), col3 = the column's own (v, w) offset through the mapping, the elastic part as a linear
// function of the plane index inside the cell.  Returns the end of the run (the next cell boundary or `limit`).
struct ColumnPlanes {
  int cell;        // control cell whose end planes are cached (-2: none)
  float P0[3], P1[3];
};

template <typename CP>
__device__ __forceinline__ int fast_column_line(const FastFrameT<CP>& f, const Lerp1D& lj, const Lerp1D& lk, ColumnPlanes& cache, int run0,
                                                int limit, int u_ref, const float (&C3)[3], const float (&col3)[3], int lane,
                                                float (&A3)[3], float (&B3)[3]) {
  int run1 = limit;
  const float du = static_cast<float>(run0 - u_ref);
  if (f.elastic) {
    const int cmax = f.ni > 1 ? f.ni - 2 : 0;
    const int cell_l = min(max(static_cast<int>(floorf(f.sci * static_cast<float>(run0 + lane))), 0), cmax);
    const int cell = __builtin_amdgcn_readlane(cell_l, 0);
    const unsigned long long later = __builtin_amdgcn_ballot_w64((lane < run1 - run0) & (cell_l > cell));
    if (later != 0ull) run1 = run0 + __builtin_ctzll(later);
    if (cell != cache.cell) {  // (j, k)-lerped control planes at the two ends of the cell
      const int s_i = f.nj * f.nk * 3, s_j = f.nk * 3;
      const int c1 = min(cell + 1, f.ni - 1);
#pragma unroll
      for (int e = 0; e < 3; e++) {
        const CP q = f.cp + e;
        const float a00 = q[cell * s_i + lj.i0 * s_j + lk.i0 * 3], a01 = q[cell * s_i + lj.i0 * s_j + lk.i1 * 3];
        const float a10 = q[cell * s_i + lj.i1 * s_j + lk.i0 * 3], a11 = q[cell * s_i + lj.i1 * s_j + lk.i1 * 3];
        const float b00 = q[c1 * s_i + lj.i0 * s_j + lk.i0 * 3], b01 = q[c1 * s_i + lj.i0 * s_j + lk.i1 * 3];
        const float b10 = q[c1 * s_i + lj.i1 * s_j + lk.i0 * 3], b11 = q[c1 * s_i + lj.i1 * s_j + lk.i1 * 3];
        const float a0 = __builtin_fmaf(lk.l1, a01 - a00, a00), a1 = __builtin_fmaf(lk.l1, a11 - a10, a10);
        const float b0 = __builtin_fmaf(lk.l1, b01 - b00, b00), b1 = __builtin_fmaf(lk.l1, b11 - b10, b10);
        cache.P0[e] = __builtin_fmaf(lj.l1, a1 - a0, a0);
        cache.P1[e] = __builtin_fmaf(lj.l1, b1 - b0, b0);
      }
      cache.cell = cell;
    }
    // d(run0 + t) = P0 + (sci (run0 + t) - cell) (P1 - P0)
    const float l_ref = fminf(fmaxf(__builtin_fmaf(f.sci, static_cast<float>(run0), -static_cast<float>(cell)), 0.0f), 1.0f);
    float D0[3], D1[3];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const float dP = cache.P1[e] - cache.P0[e];
      D0[e] = __builtin_fmaf(l_ref, dP, cache.P0[e]) * f.dsc[e];
      D1[e] = f.sci * dP * f.dsc[e];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if (f.affine_first) {
        A3[r] = __builtin_fmaf(f.m[4 * r], du, C3[r] + col3[r]) + D0[r];
        B3[r] = f.m[4 * r] + D1[r];
      } else {
        A3[r] = __builtin_fmaf(f.m[4 * r], du + D0[0], __builtin_fmaf(f.m[4 * r + 1], D0[1], __builtin_fmaf(f.m[4 * r + 2], D0[2], C3[r] + col3[r])));
        B3[r] = __builtin_fmaf(f.m[4 * r], 1.0f + D1[0], __builtin_fmaf(f.m[4 * r + 1], D1[1], f.m[4 * r + 2] * D1[2]));
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) { A3[r] = __builtin_fmaf(f.m[4 * r], du, C3[r] + col3[r]); B3[r] = f.m[4 * r]; }
  }
  return run1;
}

// Sample one run with the cheapest loop that is correct for it: the fill rule only matters where a tap can leave
// the volume.  Each coordinate of the line is monotone, so a column whose two END planes keep all first taps in
// [0, S - 2] is interior for the whole run; the wave takes the masked loop only if one of its columns is not.
template <int GMAX, bool TRACK = false>
__device__ __forceinline__ void fast_sample_line(int len, const float (&A3)[3], const float (&B3)[3], const FastAddr& ta, char* o, unsigned urow,
                                                 int64_t slab_b, bool needs_mask, float ox, float oy, float oz, float hx, float hy, float hz,
                                                 float fillv, uint32_t& kmin, float margin, int plane_base, unsigned& unsure) {
  bool masked = false;
  if (needs_mask) {
    const float el = static_cast<float>(len - 1);
    const float xa = A3[0] + ox, xb = __builtin_fmaf(el, B3[0], A3[0]) + ox;
    const float ya = A3[1] + oy, yb = __builtin_fmaf(el, B3[1], A3[1]) + oy;
    const float za = A3[2] + oz, zb = __builtin_fmaf(el, B3[2], A3[2]) + oz;
    const bool inside = (fminf(xa, xb) >= 0.0f) & (fmaxf(xa, xb) < hx) & (fminf(ya, yb) >= 0.0f) & (fmaxf(ya, yb) < hy) &
                        (fminf(za, zb) >= 0.0f) & (fmaxf(za, zb) < hz);
    masked = __builtin_amdgcn_ballot_w64(!inside) != 0ull;
  }
  if (!masked) {
    if (GMAX == 8 && len > 4)
      fast_sample_run<false, 8, false, TRACK>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv, kmin, margin, plane_base, unsure);
    else
      fast_sample_run<false, 4, false, TRACK>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv, kmin, margin, plane_base, unsure);
  } else {  // rare (waves on the volume's surface): four voxels in flight keep the register budget of the common loop
    fast_sample_run<true, 4, false, TRACK>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, ox, oy, oz, hx, hy, hz, fillv, kmin, margin, plane_base, unsure);
  }
}


enum : int {
  kSlabStaged = 0,   // box staged in LDS: sample from it
  kSlabOutside = 1,  // the brick sees nothing of the volume: fill (or 0)
  kSlabGather = 2    // non-finite geometry / box beyond the LDS budget: per-voxel global gather
};

struct StreamBox {
  int kind, bx0, by0, za, Lx, Ly, cpr, interior;
};

// Lane constants of the LDS-DMA for one row length (cpr 16-byte chunks per row): a wave instruction
// covers rpi = 64 / cpr consecutive rows of one x-plane, lane l fetching chunk ch_l of row row_l.
struct StageLanes {
  int cpr;            // key (-1: nothing cached)
  int rpi;            // rows per wave instruction
  int row_l, gz_rel;  // this lane's row inside the group and 4 * its chunk index
  unsigned goff;      // byte offset of this lane's chunk from the group's first byte (needs K: per launch constant)
  bool lane_ok;       // row_l < rpi
};

__device__ __forceinline__ void stage_lanes(StageLanes& sl, int cpr, int K, int lane) {
  if (cpr == sl.cpr) return;
  const float rcp = __builtin_amdgcn_rcpf(static_cast<float>(cpr));
  sl.cpr = cpr;
  sl.row_l = static_cast<int>((static_cast<float>(lane) + 0.5f) * rcp);
  const int ch_l = lane - sl.row_l * cpr;
  sl.rpi = __builtin_amdgcn_readfirstlane(static_cast<int>(64.0f * rcp + 1e-3f));
  sl.lane_ok = sl.row_l < sl.rpi;
  sl.gz_rel = 4 * ch_l;
  sl.goff = static_cast<unsigned>(sl.row_l * K + 4 * ch_l) * 4u;
}

// This wave's share (x-planes wave, wave + NW, ...) of the LDS-DMA of one box (dense layout: row
// pitch = 4 cpr floats).  Interior boxes: nothing but scalar pointer increments between two DMA
// instructions; boxes that stick out of the volume check rows per lane and store zeros for the
// chunks outside.
template <int NW>
__device__ __forceinline__ int stream_stage(float* __restrict__ tile, const float* __restrict__ src, const StreamBox& bx, int I, int J,
                                             int K, int wave, int lane, StageLanes& sl) {
  typedef __attribute__((address_space(1))) const char* global_byte_ptr;
  stage_lanes(sl, bx.cpr, K, lane);
  const int rpi = sl.rpi;
  // full groups of rpi rows, then one partial group (uniform float division: tiny operands, exact after the nudge)
  const int full = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(bx.Ly) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(rpi))));
  const int rest = bx.Ly - full * rpi;
  const int64_t plane_b = static_cast<int64_t>(J) * K * 4;         // bytes between x-planes of the volume
  const int64_t group_b = static_cast<int64_t>(rpi) * K * 4;       // bytes between row groups
  const int dplane = bx.Ly * bx.cpr * 4, dgroup = rpi * bx.cpr * 4;  // the same steps in LDS floats
  global_byte_ptr gp = (global_byte_ptr)(src) + ((static_cast<int64_t>(bx.bx0 + wave) * J + bx.by0) * K + bx.za) * 4;
  float* lp = tile + wave * dplane;
  int issued = 0;  // DMA instructions of this wave (scalar)
  if (bx.interior) {
    const bool tail_ok = sl.lane_ok & (sl.row_l < rest);
    for (int xr = wave; xr < bx.Lx; xr += NW) {
      issued += full + (rest > 0 ? 1 : 0);
      global_byte_ptr g = gp;
      float* l = lp;
      for (int q = 0; q < full; q++) {
        if (sl.lane_ok) __builtin_amdgcn_global_load_lds(g + sl.goff, (fast_lds_wptr)(l), 16, 0, 0);
        g += group_b; l += dgroup;
      }
      if (rest > 0 && tail_ok) __builtin_amdgcn_global_load_lds(g + sl.goff, (fast_lds_wptr)(l), 16, 0, 0);
      gp += NW * plane_b; lp += NW * dplane;
    }
    return issued;
  }
  const bool ch_ok = static_cast<unsigned>(bx.za + sl.gz_rel) < static_cast<unsigned>(K);
  const int groups = full + (rest > 0 ? 1 : 0);
  for (int xr = wave; xr < bx.Lx; xr += NW) {
    const bool plane_ok = static_cast<unsigned>(bx.bx0 + xr) < static_cast<unsigned>(I);
    global_byte_ptr g = gp;
    float* l = lp;
    for (int q = 0; q < groups; q++) {
      const int r0 = q * rpi;
      const bool in_box = sl.lane_ok & (sl.row_l < bx.Ly - r0);
      const bool in_vol = in_box & plane_ok & ch_ok & (static_cast<unsigned>(bx.by0 + r0 + sl.row_l) < static_cast<unsigned>(J));
      if (__builtin_amdgcn_ballot_w64(in_vol) != 0ull) issued++;
      if (in_vol) __builtin_amdgcn_global_load_lds(g + sl.goff, (fast_lds_wptr)(l), 16, 0, 0);
      else if (in_box) lds_zero_chunk(l + 4 * lane);
      g += group_b; l += dgroup;
    }
    gp += NW * plane_b; lp += NW * dplane;
  }
  return issued;
}

// The same box with the DMA instructions PACKED (round 3): the rows of the box are one dense sequence in LDS (x-plane
// after x-plane), and a wave instruction simply covers the next rpi rows of that sequence, whatever x-plane they belong
// to.  stream_stage issues ceil(Ly / rpi) instructions per x-plane — for the bench geometry (Ly ~ 21, rpi = 10) three,
// the third carrying a single row — i.e. 63 instructions for a box of 44 KiB; packed, the same box takes 45.  A
// vector-memory instruction costs this kernel's CU about the same whatever it moves (profiles/r03_resample.md), so the
// count is what matters.  The price is a per-lane address (row -> x-plane and row inside it, advanced incrementally)
// instead of scalar pointer increments: a handful of vector instructions per DMA instruction.
template <int NW, int AUX = 0>
__device__ __forceinline__ int stream_stage_packed(float* __restrict__ tile, const float* __restrict__ src, const StreamBox& bx, int I, int J,
                                                    int K, int wave, int lane, StageLanes& sl, int skip_mod = 0) {
  // skip_mod (instrumented instantiation only, --ablate 256 / 512): every skip_mod-th DMA instruction of a wave is NOT issued
  // (the tile keeps stale data: timing experiment — how much of the launch is the number of line requests)
  typedef __attribute__((address_space(1))) const char* global_byte_ptr;
  stage_lanes(sl, bx.cpr, K, lane);
  const int rpi = sl.rpi;
  const int total_rows = bx.Lx * bx.Ly;
  // (uniform float divisions of small integers: exact after the half-step nudge)
  const int n_instr = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(total_rows + rpi - 1) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(rpi))));
  const int step = NW * rpi;  // rows between two instructions of this wave
  const float rcp_ly = __builtin_amdgcn_rcpf(static_cast<float>(bx.Ly));
  const int step_p = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(step) + 0.5f) * rcp_ly));
  const int step_r = step - step_p * bx.Ly;
  // this lane's first row of the sequence -> (x-plane, row)
  int row = wave * rpi + sl.row_l;
  int p = static_cast<int>((static_cast<float>(row) + 0.5f) * rcp_ly);
  int r = row - p * bx.Ly;
  const unsigned plane_b = static_cast<unsigned>(J) * static_cast<unsigned>(K) * 4u, row_b = static_cast<unsigned>(K) * 4u;
  // byte offset of this lane's chunk from the box origin (inside one channel: < 2^32, resample.hip checks n_in < 2^30)
  unsigned off = static_cast<unsigned>(p) * plane_b + static_cast<unsigned>(r) * row_b + static_cast<unsigned>(sl.gz_rel) * 4u;
  const unsigned step_b = static_cast<unsigned>(step_p) * plane_b + static_cast<unsigned>(step_r) * row_b;
  const unsigned wrap_b = plane_b - static_cast<unsigned>(bx.Ly) * row_b;  // one x-plane further, Ly rows back
  global_byte_ptr origin = (global_byte_ptr)(src) + ((static_cast<int64_t>(bx.bx0) * J + bx.by0) * K + bx.za) * 4;
  const int dgroup = rpi * bx.cpr * 4;  // LDS floats per instruction
  float* lp = tile + wave * dgroup;
  int issued = 0;
  const bool ch_ok = static_cast<unsigned>(bx.za + sl.gz_rel) < static_cast<unsigned>(K);
  int it = 0;
  for (int n = wave; n < n_instr; n += NW, it++) {
    const bool in_box = sl.lane_ok & (row < total_rows) & !(skip_mod > 0 && (it % skip_mod) == skip_mod - 1);
    if (bx.interior) {
      if (in_box) __builtin_amdgcn_global_load_lds(origin + off, (fast_lds_wptr)(lp), 16, 0, AUX);
      issued++;
    } else {
      const bool in_vol = in_box & ch_ok & (static_cast<unsigned>(bx.bx0 + p) < static_cast<unsigned>(I)) &
                          (static_cast<unsigned>(bx.by0 + r) < static_cast<unsigned>(J));
      if (__builtin_amdgcn_ballot_w64(in_vol) != 0ull) issued++;
      if (in_vol) __builtin_amdgcn_global_load_lds(origin + off, (fast_lds_wptr)(lp), 16, 0, AUX);
      else if (in_box) lds_zero_chunk(lp + 4 * lane);
    }
    row += step; p += step_p; r += step_r; off += step_b;
    if (r >= bx.Ly) { r -= bx.Ly; p += 1; off += wrap_b; }
    lp += NW * dgroup;
  }
  return issued;
}

typedef FastFrameT<const float*> FastFrameG;

struct PipePlan {
  int b, it, jt, kt;
  int i_begin, j_lo, k_lo, i_count, nv, nw;
  int fast;  // the brick's whole plane range in one pass: a staged box, or nothing of the volume in sight
  StreamBox bx;
};

template <int TI, int TJ, int TK>
__device__ __forceinline__ void pipe_decode(const ResampleArgs& a, unsigned tile, PipePlan& p) {
  const unsigned t1 = fastdiv_exact(tile, a.magic_k, a.tiles_k);
  p.kt = tile - t1 * a.tiles_k;
  const unsigned t2 = fastdiv_exact(t1, a.magic_j, a.tiles_j);
  p.jt = t1 - t2 * a.tiles_j;
  const unsigned t3 = fastdiv_exact(t2, a.magic_i, a.tiles_i);
  p.it = t2 - t3 * a.tiles_i;
  p.b = static_cast<int>(t3);
  p.i_begin = p.it * TI; p.j_lo = p.jt * TJ; p.k_lo = p.kt * TK;
  p.i_count = min(TI, a.Io - p.i_begin); p.nv = min(TJ, a.Jo - p.j_lo); p.nw = min(TK, a.Ko - p.k_lo);
  p.fast = 0;
}

__device__ __forceinline__ void pipe_brick_frame(FastFrameG& f, int j_lo, int k_lo) {
#pragma unroll
  for (int r = 0; r < 3; r++)
    f.c[r] = static_cast<double>(f.m[4 * r + 1]) * j_lo + static_cast<double>(f.m[4 * r + 2]) * k_lo + static_cast<double>(f.m[4 * r + 3]);
  f.j_lo = j_lo; f.k_lo = k_lo;
}

// vertex `vtx` of the brick framed in f (planes [u0, u0 + n)): its coordinate, clamped for the integer box
__device__ __forceinline__ void pipe_vertex(const ResampleArgs& a, const FastFrameG& f, int vtx, int u0, int n, int nv, int nw, int (&r)[6], bool& bad) {
  const int u_lo = u0, u_hi = u0 + n - 1;
  int du, dv, dw;
  if (f.elastic) { du = vtx % 3; dv = (vtx / 3) % 3; dw = vtx / 9; } else { du = vtx & 1; dv = (vtx >> 1) & 1; dw = vtx >> 2; }
  bool dense = false;
  float u = du == 0 ? static_cast<float>(u_lo) : static_cast<float>(u_hi);
  float v = dv == 0 ? 0.0f : static_cast<float>(nv - 1);
  float w = dw == 0 ? 0.0f : static_cast<float>(nw - 1);
  if (f.elastic) {
    if (du == 2) u = fast_breakpoint(f.sci, f.ni, u_lo, u_hi, dense);
    if (dv == 2) v = fast_breakpoint(f.scj, f.nj, f.j_lo, f.j_lo + nv - 1, dense) - static_cast<float>(f.j_lo);
    if (dw == 2) w = fast_breakpoint(f.sck, f.nk, f.k_lo, f.k_lo + nw - 1, dense) - static_cast<float>(f.k_lo);
  }
  float x, y, z;
  fast_coord(f, u, v, w, x, y, z);
  constexpr float kMargin = 1.0f / 64.0f;
  bad = !(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f) | dense;
  const float capx = a.size_m1[0] + 1.0f + kTileFar, capy = a.size_m1[1] + 1.0f + kTileFar, capz = a.size_m1[2] + 1.0f + kTileFar;
  r[0] = -static_cast<int>(fminf(fmaxf(floorf(x - kMargin), -kTileFar), capx));
  r[1] = static_cast<int>(fminf(fmaxf(floorf(x + kMargin), -kTileFar), capx));
  r[2] = -static_cast<int>(fminf(fmaxf(floorf(y - kMargin), -kTileFar), capy));
  r[3] = static_cast<int>(fminf(fmaxf(floorf(y + kMargin), -kTileFar), capy));
  r[4] = -static_cast<int>(fminf(fmaxf(floorf(z - kMargin), -kTileFar), capz));
  r[5] = static_cast<int>(fminf(fmaxf(floorf(z + kMargin), -kTileFar), capz));
}

// =====================================================================================================================
// The plan: 16 dwords per brick, 16 floats per batch element ahead of them
//   brick:  [0] kind | interior << 8   [1] bx0 [2] by0 [3] za [4] Lx [5] Ly [6] cpr   [7..9] C3 (float bits): the mapping
//           of (i_begin, j_lo, k_lo) relative to the box origin   [10] b [11] i_begin [12] j_lo [13] k_lo [14] elastic
//   batch:  [0..11] mapping rows scaled by the axis ratios (S_own - 1) / max(S_norm - 1, 1)   [12] margin of the fill decision
// Round 6 (a.plan_multi, the exact-coordinate lean kernel only): a brick whose box exceeds the tile is bounded again in HALVES,
// then QUARTERS, of its planes.  kind = kDescMulti, [1..6] = the box of pass 0, [15] = passes | state of pass q << (8 + 4 q), and the
// boxes of ALL passes follow the descriptors: kPassInts ints per pass, 4 passes per brick, at plan + B 16 + n_items 16 +
// brick 32:  [0] bx0 [1] by0 [2] za [3] Lx [4] Ly [5] cpr [6] state (kPassStaged / kPassOutside / kPassSlow) | interior << 8
// [7] planes.  A pass that still does not fit samples its planes voxel by voxel; the others stage their box one after the other.
// Behind the pass boxes: the LIST of these bricks, which the walker blocks of resample_lean_exact_kernel take bricks from; its
// header — [0] their number, [1] / [2] the walkers' cursor and done count — sits in FRONT of the plan (plan[-kPlanHeaderInts ...]:
// one place whatever the launch's size, zero between launches).
// =====================================================================================================================
enum : int { kPlanHeaderInts = 16, kDescMulti = 4, kPassInts = 8, kPassesPerBrick = 4, kPassStaged = 0, kPassOutside = 1, kPassSlow = 2 };

// the integer box of extremes `ext` (butterfly-reduced: identical in every lane of the group)
struct PlanBox {
  int xmin, ymin, za, Lx, Ly, cpr, interior, outside, fits;
};
__device__ __forceinline__ PlanBox plan_box(const ResampleArgs& a, const int (&ext)[7], bool weird) {
  PlanBox pb;
  const int xmin = -ext[0], xmax = ext[1], ymin = -ext[2], ymax = ext[3], zmin = -ext[4], zmax = ext[5];
  const bool wrd = weird | (ext[6] != 0);
  pb.interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
  pb.outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
  pb.za = zmin & ~3; pb.Lx = xmax + 2 - xmin; pb.Ly = ymax + 2 - ymin;
  const int Lz = ((zmax + 1 + 4) & ~3) - pb.za;
  pb.fits = !wrd && (Lz <= 256) && (pb.Lx <= 4096) && (pb.Ly <= 4096) && (static_cast<int64_t>(pb.Lx) * pb.Ly * Lz <= static_cast<int64_t>(a.tile_cap));
  pb.xmin = xmin; pb.ymin = ymin; pb.cpr = Lz >> 2;
  return pb;
}
// ... of a PASS of a multi-pass brick: the consumer (resample_lean_exact_kernel) takes the planner's word for these, so the
// conditions it re-checks for a one-pass box — the tap addresses formed from absolute indices stay exact in float32
// (box_address_fits, resample_tile.hpp) — are decided here
__device__ __forceinline__ PlanBox plan_pass_box(const ResampleArgs& a, const int (&ext)[7], bool weird) {
  PlanBox pb = plan_box(a, ext, weird);
  if (pb.fits && !box_address_fits(pb.xmin, pb.ymin, pb.za, pb.Lx, pb.Ly, pb.cpr)) pb.fits = 0;
  return pb;
}
// A group of lanes per brick, one vertex per lane: 32 lanes (27 busy) when the launch has control points — a vertex then
// reads 24 control values, and one thread walking its 27 vertices one after the other made the planner a chain of 27
// memory round trips (21.6 us per bench launch; side by side 16.4) — 8 lanes for affine-only launches (5.8 us either
// way).  The extremes meet through shuffles inside the group; its lane 0 writes the descriptor.
constexpr int plan_group(bool elastic_possible) { return elastic_possible ? 32 : 8; }

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK>
__global__ __launch_bounds__(256) void plan_bricks_kernel(const ResampleArgs a, int* __restrict__ plan, int n_items) {
  constexpr int GROUP = plan_group(ELASTIC_POSSIBLE);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};
  if (gtid < a.B) {
    const float* m = a.mapping + (a.mapping_batched ? gtid * 12 : 0);
    float* fr = reinterpret_cast<float*>(plan) + gtid * 16;
    for (int q = 0; q < 12; q++) fr[q] = m[q] * ratio[q >> 2];
    {  // [12]: the margin of the fill decision for this element (resample_exact_chain.hpp)
      float mm[12];
      for (int q = 0; q < 12; q++) mm[q] = m[q];
      fr[12] = fast_fill_margin(mm, static_cast<float>(a.Io), static_cast<float>(a.Jo), static_cast<float>(a.Ko), a.size_m1[0] + 1.0f,
                                a.size_m1[1] + 1.0f, a.size_m1[2] + 1.0f);
    }
    fr[13] = __int_as_float(a.tile_cap);  // (the tile the boxes were sized for: the consumer of a multi-pass brick compares it with its own)
    for (int q = 14; q < 16; q++) fr[q] = 0.0f;
  }
  const int t = gtid / GROUP, v = gtid % GROUP;
  if (t >= n_items) return;  // (whole groups leave together: the shuffles below stay inside a group)
  PipePlan p;
  pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(t), p);
  int* d = plan + a.B * 16 + t * kDescInts;
  if (a.passthrough != nullptr && a.passthrough[p.b] != 0) {
    for (int q = v; q < kDescInts; q += GROUP) d[q] = q == 0 ? kDescGated : (q == 10 ? p.b : (q == 11 ? p.i_begin : (q == 12 ? p.j_lo : (q == 13 ? p.k_lo : 0))));
    return;
  }
  FastFrameG f;
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.cp = nullptr; f.elastic = false;
  bool weird = false;
  {
    const float* m = a.mapping + (a.mapping_batched ? p.b * 12 : 0);
#pragma unroll
    for (int q = 0; q < 12; q++) {
      const float mv = m[q];
      weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;  // |m| > 1e30, Inf or NaN
      f.m[q] = mv * ratio[q >> 2];
    }
  }
#pragma unroll
  for (int e = 0; e < 3; e++) f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f);
  if constexpr (ELASTIC_POSSIBLE) {
    f.elastic = !(a.cp_skip != nullptr && a.cp_skip[p.b] != 0);
    f.cp = f.elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(p.b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
  }
  pipe_brick_frame(f, p.j_lo, p.k_lo);
  int ext[7] = {-0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000, 0};
  const int n_vert = f.elastic ? 27 : 8;
  if (v < n_vert) {
    int r[6]; bool bad;
    pipe_vertex(a, f, v, p.i_begin, p.i_count, p.nv, p.nw, r, bad);
#pragma unroll
    for (int q = 0; q < 6; q++) ext[q] = r[q];
    ext[6] = bad ? 1 : 0;
  }
#pragma unroll
  for (int s = GROUP / 2; s > 0; s >>= 1) {
#pragma unroll
    for (int q = 0; q < 7; q++) ext[q] = max(ext[q], __shfl_xor(ext[q], s));
  }
  // (the butterfly leaves the extremes in EVERY lane of the group: the decisions below are uniform across it)
  const PlanBox whole = plan_box(a, ext, weird);
  const bool wrd = weird | (ext[6] != 0);
  int kind = whole.outside ? kDescOutside : (whole.fits ? kDescStaged : kDescSlow);
  PlanBox first = whole;
  int d15 = 0;
  if (a.plan_multi != 0 && kind == kDescSlow && !wrd) {
    // the box exceeds the tile: halves of the planes, else quarters (pass boxes behind the descriptors)
    int* passes = plan + a.B * 16 + n_items * kDescInts + t * (kPassInts * kPassesPerBrick);
    for (int nsplit = 2; nsplit <= 4; nsplit *= 2) {
      const int span = TI / nsplit;
      bool all_ok = true;
      for (int q = 0; q < nsplit; q++) {
        const int u0 = p.i_begin + q * span, n = min(span, p.i_begin + p.i_count - u0);
        int e7[7] = {-0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000, 0};
        if (v < n_vert && n > 0) {
          int r[6]; bool bad;
          pipe_vertex(a, f, v, u0, n, p.nv, p.nw, r, bad);
#pragma unroll
          for (int c = 0; c < 6; c++) e7[c] = r[c];
          e7[6] = bad ? 1 : 0;
        }
#pragma unroll
        for (int sft = GROUP / 2; sft > 0; sft >>= 1) {
#pragma unroll
          for (int c = 0; c < 7; c++) e7[c] = max(e7[c], __shfl_xor(e7[c], sft));
        }
        PlanBox pb = plan_pass_box(a, e7, weird);
        if (n <= 0) { pb.outside = 1; pb.fits = 0; pb.interior = 0; }  // (no plane of a partial brick in this part: nothing to do)
        all_ok &= (pb.fits | pb.outside) != 0;
        const int state = pb.outside ? kPassOutside : (pb.fits ? kPassStaged : kPassSlow);
        if (q == 0) { first = pb; d15 = nsplit; }
        d15 |= state << (8 + 4 * q);
        if (v == 0) {
          int* r8 = passes + q * kPassInts;
          r8[0] = pb.xmin; r8[1] = pb.ymin; r8[2] = pb.za; r8[3] = pb.Lx; r8[4] = pb.Ly; r8[5] = pb.cpr;
          r8[6] = state | (pb.interior << 8); r8[7] = max(n, 0);
        }
      }
      kind = kDescMulti;
      if (all_ok) break;
    }
  }
  if (v != 0) return;
  if (kind == kDescMulti && a.plan_multi == 1) {  // ... and listed for the walker blocks of resample_lean_exact_kernel: [0] the count (left at zero by the last launch that read this buffer), [4 ...] the bricks
    int* list = plan + a.B * 16 + n_items * (kDescInts + kPassInts * kPassesPerBrick);
    const int at = atomicAdd(plan - kPlanHeaderInts, 1);
    if (at < n_items) list[at] = t;
  }
  d[10] = p.b; d[11] = p.i_begin; d[12] = p.j_lo; d[13] = p.k_lo; d[15] = d15;
  d[0] = kind | (first.interior << 8);
  d[1] = first.xmin; d[2] = first.ymin; d[3] = first.za; d[4] = first.Lx; d[5] = first.Ly; d[6] = first.cpr;
  const double org[3] = {static_cast<double>(first.xmin), static_cast<double>(first.ymin), static_cast<double>(first.za)};
#pragma unroll
  for (int r = 0; r < 3; r++) d[7 + r] = __float_as_int(static_cast<float>(static_cast<double>(f.m[4 * r]) * p.i_begin + f.c[r] - org[r]));
  d[14] = f.elastic ? 1 : 0;
}

constexpr int kMinSlots = 64;  // keys per channel of the folded minimum

// one 64-lane block per channel: the slots' minimum, decoded; the slots go back to all ones
constexpr int kMinChannels = 16;  // channels of one launch that can fold their minimum (more: the plain reduction)
struct MinOuts { float* p[kMinChannels]; };

__global__ __launch_bounds__(kMinSlots) void min_finish_kernel(uint32_t* __restrict__ keys, const MinOuts outs, int n_channels) {
  const int c = blockIdx.x;
  if (c >= n_channels) return;
  uint32_t* slot = keys + c * kMinSlots + threadIdx.x;
  const uint32_t k = __hip_atomic_exchange(slot, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t m = wave_min_u32(k);
  if (threadIdx.x == 0) *outs.p[c] = key_to_float(m);
}

// =====================================================================================================================
// One block per planned brick, one column of TI planes per thread; images and channels of the launch one after the
// other through the same LDS tile (the geometry, hence the box and the lines, is shared).
// =====================================================================================================================
template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int OCC>
__global__ __launch_bounds__(TJ* TK, (TJ * TK) / 256 * OCC) void resample_planned_kernel(const ResampleArgs a, const int* __restrict__ plan) {
  constexpr int NT = TJ * TK, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_tile = smem;
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;

  const unsigned brick = xcd_remap(blockIdx.x, gridDim.x);
  const_int_ptr d = (const_int_ptr)(plan + a.B * 16) + static_cast<size_t>(brick) * kDescInts;
  const int kind_w = d[0];
  const int kind = kind_w & 0xFF;
  StreamBox bx;
  bx.kind = kind; bx.interior = kind_w >> 8;
  bx.bx0 = d[1]; bx.by0 = d[2]; bx.za = d[3]; bx.Lx = d[4]; bx.Ly = d[5]; bx.cpr = d[6];
  const float C3[3] = {__int_as_float(d[7]), __int_as_float(d[8]), __int_as_float(d[9])};
  const int b = d[10], i_begin = d[11], j_lo = d[12], k_lo = d[13];
  const bool elastic = ELASTIC_POSSIBLE && d[14] != 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid % TK, tj = tid / TK;
  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
  const unsigned urow = static_cast<unsigned>(col_off) * 4u;
  const int u0 = i_begin, u1 = i_begin + i_count;

  // The folded minimum (ImgArgs::out_min): the waves of the FIRST batch element fold the keys of what they store into one
  // of kMinSlots keys per channel (by brick, so that a few hundred — not sixteen thousand — atomics meet on an address;
  // returnless: nobody waits for them); min_finish_kernel, launched behind this kernel, folds the slots, decodes the
  // result into out_min[c] and restores the keys.  It replaces the three launches and the re-read of tio_channel_min
  // for the next transform's "minimum" fill.  (A first version decoded in the last wave to finish, through a ticket
  // counter and returning atomics: 0.40 -> 0.43 - 0.45 ms per launch, more than the reduction it saved.)
  auto publish_min = [&](const ImgArgs& g, int c, uint32_t kmin) {
    const uint32_t wmin = wave_min_u32(kmin);
    if (lane == 0 && wmin != 0xFFFFFFFFu)
      __hip_atomic_fetch_min(g.min_keys + c * kMinSlots + (brick & (kMinSlots - 1)), wmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  if (kind == kDescGated || kind == kDescOutside) {  // gated-out element: bit-exact copy; nothing of the volume in sight: fill (or 0)
    for (int im = 0; im < a.n_images; im++) {
      const ImgArgs& g = a.img[im];
      for (int c = 0; c < g.channels; c++) {
        const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
        char* out_chan = static_cast<char*>(g.out) + bc * n_out * 4;
        const float* in_chan = static_cast<const float*>(g.in) + bc * n_in;
        const float fillv = g.fill != nullptr ? ((const_float_ptr)g.fill)[c] : 0.0f;
        uint32_t kmin = 0xFFFFFFFFu;
        if (col_active) {
          for (int t = u0; t < u1; t++) {
            const float val = kind == kDescGated ? in_chan[static_cast<int64_t>(t) * slab + col_off] : fillv;
            *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = val;
            kmin = min(kmin, float_to_key(val));
          }
        }
        if (g.out_min != nullptr && b == 0) publish_min(g, c, kmin);
      }
    }
    return;
  }

  // the scaled mapping of this batch element: scalar loads, no arithmetic
  FastFrameG f;
  {
    const_float_ptr fm = (const_float_ptr)(plan) + b * 16;
#pragma unroll
    for (int q = 0; q < 12; q++) f.m[q] = fm[q];
  }
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.elastic = elastic;
  f.cp = elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
  f.j_lo = j_lo; f.k_lo = k_lo;
  {
    const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};
#pragma unroll
    for (int e = 0; e < 3; e++) { f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f); f.c[e] = 0.0; }
  }

  if (kind == kDescSlow) {  // box beyond the LDS budget / non-finite geometry: per-voxel evaluation, global gathers (rare)
    // (with the reference's exact coordinate chain: gather_voxel applies the fill rule to the coordinate it is handed)
    const float* mp = a.mapping + (a.mapping_batched ? b * 12 : 0);
    float mm[12];
#pragma unroll
    for (int q = 0; q < 12; q++) mm[q] = mp[q];
    Lerp1D lj_e{0, 0, 1.0f, 0.0f}, lk_e{0, 0, 1.0f, 0.0f};
    if (elastic) {
      lj_e = lerp_index(j_lo + jv, a.nj, a.Jo, a.scale_j);
      lk_e = lerp_index(k_lo + kw, a.nk, a.Ko, a.scale_k);
    }
    for (int im = 0; im < a.n_images; im++) {
      const ImgArgs& g = a.img[im];
      if (col_active) {
        for (int t = u0; t < u1; t++) {
          float x, y, z;
          exact_voxel_coords<ELASTIC_POSSIBLE>(a, mm, elastic, f.cp, lj_e, lk_e, t, static_cast<float>(j_lo + jv), static_cast<float>(k_lo + kw), x, y, z);
          gather_voxel<0>(g, a, b, n_in, n_out, t * slab + col_off, x, y, z, false);
        }
      }
      if (g.out_min != nullptr && b == 0) {  // (this thread reads back what it has just stored)
        for (int c = 0; c < g.channels; c++) {
          uint32_t kmin = 0xFFFFFFFFu;
          if (col_active) {
            const float* out_chan = static_cast<const float*>(g.out) + (static_cast<int64_t>(b) * g.channels + c) * n_out;
            for (int t = u0; t < u1; t++) kmin = min(kmin, float_to_key(out_chan[static_cast<int64_t>(t) * slab + col_off]));
          }
          publish_min(g, c, kmin);
        }
      }
    }
    return;
  }

  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
  // per-column constants, formed while the first brick is on its way
  const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
  float col3[3];
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  ColumnPlanes planes;
  planes.cell = -2;
#pragma unroll
  for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
  FastAddr ta;
  ta.sYb = bx.cpr * 16; ta.sXb = bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
  ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
  ta.base_f = static_cast<float>(static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile)));
  const float ox = static_cast<float>(bx.bx0), oy = static_cast<float>(bx.by0), oz = static_cast<float>(bx.za);
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];

  bool first = true;
  for (int im = 0; im < a.n_images; im++) {
    const ImgArgs& g = a.img[im];
    for (int c = 0; c < g.channels; c++) {
      const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
      char* out_chan = static_cast<char*>(g.out) + bc * n_out * 4;
      const float* in_chan = static_cast<const float*>(g.in) + bc * n_in;
      const bool has_fill = g.fill != nullptr;
      const float fillv = has_fill ? ((const_float_ptr)g.fill)[c] : 0.0f;
      if (!first) __syncthreads();  // the previous channel's taps are read
      if (a.dma_packed) stream_stage_packed<NW>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl);
      else stream_stage<NW>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl);
      if (first) {
#pragma unroll
        for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
        if constexpr (ELASTIC_POSSIBLE) {
          if (elastic) {
            lj = lerp_index(j_lo + jv, a.nj, a.Jo, a.scale_j);
            lk = lerp_index(k_lo + kw, a.nk, a.Ko, a.scale_k);
          }
        }
        first = false;
      }
      float A3[3], B3[3];
      int run0 = u0;
      int run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
      tile_dma_wait();
      __syncthreads();
      const bool track = g.out_min != nullptr && b == 0;  // block uniform
      uint32_t kmin = 0xFFFFFFFFu;
      unsigned unsure = 0u;  // planes of this column whose in-bounds weight came within the margin of 1/2
      const float margin = (a.fill_recheck != 0 && has_fill) ? ((const_float_ptr)(plan) + b * 16)[12] : -1.0f;
      if (col_active) {
        for (;;) {
          char* o_run = out_chan + static_cast<int64_t>(run0) * slab_b;
          if (track)
            fast_sample_line<4, true>(run1 - run0, A3, B3, ta, o_run, urow, slab_b, has_fill & !bx.interior, ox, oy, oz, hx, hy, hz, fillv, kmin, margin, run0 - u0, unsure);
          else
            fast_sample_line<4, false>(run1 - run0, A3, B3, ta, o_run, urow, slab_b, has_fill & !bx.interior, ox, oy, oz, hx, hy, hz, fillv, kmin, margin, run0 - u0, unsure);
          run0 = run1;
          if (run0 >= u1) break;
          run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
        }
      }
      if (has_fill & !bx.interior)  // (block uniform) re-decide the recorded voxels with the exact chain
        fast_fill_tail<ELASTIC_POSSIBLE>(unsure, a, a.mapping + (a.mapping_batched ? b * 12 : 0), elastic, f.cp, lj, lk, u0, static_cast<float>(j_lo + jv),
                                         static_cast<float>(k_lo + kw), in_chan, a.I, a.J, a.K, hx, hy, hz, fillv, out_chan, slab_b, urow, track ? &kmin : nullptr);
      if (track) publish_min(g, c, kmin);
    }
  }
}



// =====================================================================================================================
// Round 3: the LEAN planned kernel — ONE channel of one float32 trilinear image per launch; a FAST call with several images
// or channels is one plan and one lean launch per channel (resample_planned_kernel above, with its image / channel loops,
// remains for launches that fold the minimum of their output).
//
// Shader-clock stamps in the general kernel (profiles/r03_planned2_stamps.log) showed where a block's life goes:
// 900 ticks to its descriptor, **6 400 from the descriptor to the last DMA instruction issued**, 2 400 until the box has
// landed, 5 900 of sampling, 600 until the stores are acknowledged.  The 6 400 are not the DMA: halving the number of
// vector-memory instructions (packed DMA rows, 16-byte stores) changed nothing.  They are the PROLOGUE — ~550
// instructions and five dependent scalar-load round trips (grid size -> plan pointer -> descriptor -> mapping ->
// image table -> fill value) generated from a 900-byte argument struct with image and channel loops, 64 spilled scalar
// registers — executed by waves that are nearly alone on their SIMDs (~9 clocks per dependent instruction).  Same
// arithmetic here, organised for the shortest road to the first DMA instruction: a 150-byte argument block that arrives
// in the first scalar loads, the descriptor and the element's mapping requested TOGETHER (the element index comes from
// the brick index by a multiply-high, not from the descriptor), lane constants formed in the shadow of those loads,
// no loops over images or channels, the per-column constants after the DMA has been issued.
// =====================================================================================================================
struct LeanArgs {
  const float* in;    // channel c of a (B, C, I, J, K) image: base pointer of (0, c), batch elements in_stride apart
  float* out;         // the same channel of the (B, C, Io, Jo, Ko) output, batch elements out_stride apart
  int64_t in_stride, out_stride;
  const float* fill;  // one float, or nullptr (no fill rule)
  const int* plan;    // plan_bricks_kernel's output
  const float* cp;    // control points (ELASTIC_POSSIBLE) or nullptr
  int I, J, K, Io, Jo, Ko;
  int B, n_items;
  unsigned bricks_per_element, bpe_magic;
  int ni, nj, nk, cp_batched;
  float sci, scj, sck;
  float dsc[3];
  float hx, hy, hz;
  int affine_first, ablate;
  // what the exact chain of the fill recheck reads (resample_exact_chain.hpp; only voxels whose in-bounds weight is within
  // a margin of 1/2 get there): the unscaled mapping, the spacings, the normalisation divisors
  const float* mapping;
  int mapping_batched, unit_spacing, fill_recheck;
  float sp[3], rsp[3], den[3], rden[3];
  // resample_lean_exact.hpp: the folded normalise round trip (den / 2, its reciprocal, (S - 1) / 2) and the A/B switch of its
  // interleaved DMA issue
  float dh[3], rdh[3], half_h[3];
  int interleave;
  int tile_floats;  // floats of ONE staging tile of this launch (the planner's tile_cap)
  int last_use;     // resample_lean_exact_multi_kernel: this launch is the last one that reads the plan (the last walker zeroes the list's length)
  // the folded minimum (tio_resample_image.out_min_dev): kMinSlots keys of this channel, or nullptr.  Only the bricks of batch
  // element 0 track what they store (a block-uniform branch into the TRACK instantiation of the sampling loop).
  uint32_t* min_keys;
  // resample_lean_exact_pair_kernel (round 6): a SECOND channel of the same geometry — another image of the subject or another channel
  // of this one — sampled by the same block from the same coordinates (the box is staged again into the same tile)
  const float* in2;
  float* out2;
  int64_t in_stride2, out_stride2;
  const float* fill2;
  // resample_lean_exact_label_kernel (round 6): ONE nearest-neighbour label channel of the same geometry riding along — element bits of
  // lab_es bytes (1, 2 or 4), no fill rule (zero padding) — sampled from the same coordinates behind the float channels
  const void* lab_in;
  void* lab_out;
  int lab_es;
};

// FAST trilinear sample straight from global memory (bricks whose box does not fit the tile, non-finite geometry): zero
// padding, the staged path's lerp nest and separable fill rule
__device__ __forceinline__ float lean_gather(const float* __restrict__ chan, int I, int J, int K, float x, float y, float z, bool has_fill,
                                             float fillv, float hx, float hy, float hz, float margin, int plane, unsigned& unsure) {
  if (!(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f)) return has_fill ? fillv : 0.0f;  // NaN / Inf: nothing in bounds
  const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
  FastTaps ts;
  ts.fx = x - x0; ts.fy = y - y0; ts.fz = z - z0;
  const float cx = fminf(fmaxf(x0, -2.0f), hx + 1.0f), cy = fminf(fmaxf(y0, -2.0f), hy + 1.0f), cz = fminf(fmaxf(z0, -2.0f), hz + 1.0f);
  const int ix = static_cast<int>(cx), iy = static_cast<int>(cy), iz = static_cast<int>(cz);
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int px = ix + (t & 1), py = iy + ((t >> 1) & 1), pz = iz + (t >> 2);
    const bool ok = (static_cast<unsigned>(px) < static_cast<unsigned>(I)) & (static_cast<unsigned>(py) < static_cast<unsigned>(J)) &
                    (static_cast<unsigned>(pz) < static_cast<unsigned>(K)) & (cx == x0) & (cy == y0) & (cz == z0);
    ts.v[t] = ok ? chan[(static_cast<int64_t>(px) * J + py) * K + pz] : 0.0f;
  }
  float val = fast_finish(ts);
  if (has_fill) {
    const float mk = fast_mask(ts, x0, y0, z0, hx, hy, hz);
    val = (mk > 0.5f) ? val : fillv;
    unsure |= (fabsf(mk - 0.5f) <= margin) ? (1u << plane) : 0u;  // re-decided by the kernel's tail (exact chain)
  }
  return val;
}

// Brick shape TI x TJ x TK (planes x rows x lanes along K, TJ * TK = 256 threads); the planner is instantiated for the same
// shape.  Tried and dropped (profiles/r03_resample.md): eight waves per 16-plane brick (two threads per column) — no
// faster for affine launches, much slower for elastic ones; 8-plane bricks with five or six blocks per CU — no faster;
// 16 x 8 x 32 bricks (longer rows: fewer, fuller cache lines) — boxes outgrow the tile, slower; 16-byte stores through a
// quad transpose — a quarter of the store instructions, no faster (the transpose costs what the stores saved).  Later in
// round 3 (profiles/r03_resample.md section 5): wave priorities (s_setprio around the DMA issue or the sampling),
// nontemporal stores, a wave footprint of 8 x 8 columns (conflict-free LDS reads, 32-byte store segments), the box
// through registers (global_load_dwordx4 + ds_write_b128) instead of the LDS-DMA, 12- and 14-plane bricks with a fourth
// block per CU, 512-thread blocks on 16 x 16 x 32 and 8 x 16 x 32 bricks — none faster on the affine launch (the
// 16 x 16 x 32 bricks gain 6 % on an elastic-only launch and lose 2 x on rotated boxes that outgrow 80 KB).
// INSTR = true is the INSTRUMENTED instantiation (launched only when TIO_TILE_ABLATE is set: tests/native/resample_bench
// --ablate): a.ablate 1 no DMA, 2 no sampling, 64 shader-clock stamps written over the brick's first output row, 128
// every lane samples one LDS address.  The production instantiation (INSTR = false) carries none of it.
// FOLD_MIN = true is the instantiation launched when the caller asked for the folded minimum (LeanArgs::min_keys): the
// bricks of batch element 0 run the TRACK form of the sampling loop.  Its own instantiation because the second copy of the
// loop costs scalar registers (14 / 26 more spilled) that launches without the request should not pay for.
template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int WAVES_PER_SIMD, bool INSTR = false, bool FOLD_MIN = false>
__global__ __launch_bounds__(TJ* TK, WAVES_PER_SIMD) void resample_planned_lean_kernel(const LeanArgs a) {
  const int ablate = INSTR ? a.ablate : 0;
  constexpr int NW = TJ * TK / 64;
  static_assert((TJ * TK) % 64 == 0 && (TK & (TK - 1)) == 0, "one thread per column of the brick");
  constexpr int PLANES = TI;  // output planes per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_tile = smem;
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;

  // every argument the road to the first DMA needs, in scalar registers NOW: the loads go out back to back and share one
  // wait (left to itself the compiler fetches each kernel argument right before its first use: five round trips)
  {
    const int* plan_p = a.plan; const float* in_p = a.in;
    asm volatile("" ::"s"(a.n_items), "s"(a.bricks_per_element), "s"(a.bpe_magic), "s"(plan_p), "s"(in_p), "s"(a.B), "s"(a.I), "s"(a.J), "s"(a.K));
    if constexpr (INSTR) asm volatile("" ::"s"(a.ablate));
  }
  const unsigned long long t_entry = (ablate & 64) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned brick = xcd_remap(blockIdx.x, static_cast<unsigned>(a.n_items));
  const int b = static_cast<int>(fastdiv_exact(brick, a.bpe_magic, a.bricks_per_element));
  // the two scalar loads everything waits for, requested together
  const_int_ptr d = (const_int_ptr)(a.plan + a.B * 16) + static_cast<size_t>(brick) * kDescInts;
  const_float_ptr fm = (const_float_ptr)(a.plan) + b * 16;
  const int kind_w = d[0];
  StreamBox bx;
  bx.bx0 = d[1]; bx.by0 = d[2]; bx.za = d[3]; bx.Lx = d[4]; bx.Ly = d[5]; bx.cpr = d[6];
  const float C3[3] = {__int_as_float(d[7]), __int_as_float(d[8]), __int_as_float(d[9])};
  const int i_begin = d[11], j_lo = d[12], k_lo = d[13];
  const bool elastic = ELASTIC_POSSIBLE && d[14] != 0;
  FastFrameG f;
#pragma unroll
  for (int q = 0; q < 12; q++) f.m[q] = fm[q];

  // lane constants (no dependence on the loads above)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid & (TK - 1), tj = tid / TK;
  constexpr int part = 0;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float* in_chan = a.in + static_cast<int64_t>(b) * a.in_stride;
  char* out_chan = reinterpret_cast<char*>(a.out + static_cast<int64_t>(b) * a.out_stride);

  const int kind = kind_w & 0xFF;
  bx.kind = kind; bx.interior = kind_w >> 8;
  unsigned long long t_desc = 0ull, t_issued = 0ull;
  if (ablate & 64) { asm volatile("" ::"s"(kind)); t_desc = __builtin_amdgcn_s_memtime(); }
  if (kind == kDescStaged && !(ablate & 1)) {  // the road to the first DMA instruction ends here
    StageLanes sl;
    sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
    if (INSTR && (ablate & 2048))  // experiment: the box requested with the nontemporal hint
      stream_stage_packed<NW, 2>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl);
    else
      stream_stage_packed<NW>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl, INSTR ? ((ablate & 256) ? 3 : ((ablate & 512) ? 2 : 0)) : 0);
  }
  if (ablate & 64) t_issued = __builtin_amdgcn_s_memtime();

  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
  const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
  const unsigned urow = static_cast<unsigned>(col_off) * 4u;
  const int u_ref = i_begin;  // plane the descriptor's line constants refer to
  const int u0 = min(i_begin + part * PLANES, i_begin + i_count), u1 = min(u0 + PLANES, i_begin + i_count);
  const bool has_fill = a.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)a.fill)[0] : 0.0f;
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
      for (int t = u0; t < u1; t++) {
        const float val = kind == kDescGated ? in_chan[static_cast<int64_t>(t) * slab + col_off] : fillv;
        *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = val;
        kmin = min(kmin, float_to_key(val));
      }
    }
    if (track) publish_min();
    return;
  }

  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.sci; f.scj = a.scj; f.sck = a.sck;
  f.elastic = elastic;
  f.cp = elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
  f.j_lo = j_lo; f.k_lo = k_lo;
#pragma unroll
  for (int e = 0; e < 3; e++) { f.dsc[e] = a.dsc[e]; f.c[e] = 0.0; }

  // The fill decision of voxels whose FAST in-bounds weight comes within `margin` of 1/2 (a few hundred per 256^3 volume, all
  // in bricks on the volume's surface) is made by the reference's exact chain, in a tail behind the sampling loops
  // (resample_exact_chain.hpp); the loops only raise the plane's bit in `unsure`.
  unsigned unsure = 0u;
  const float margin = (a.fill_recheck != 0 && has_fill) ? fm[12] : -1.0f;
  auto lean_tail = [&]() {
    if (__builtin_amdgcn_ballot_w64(unsure != 0u) == 0ull) return;
    ExactChainArgs ea;
    ea.ni = a.ni; ea.nj = a.nj; ea.nk = a.nk; ea.Io = a.Io; ea.unit_spacing = a.unit_spacing; ea.affine_first = a.affine_first;
    ea.scale_i = a.sci;
#pragma unroll
    for (int e = 0; e < 3; e++) { ea.sp[e] = a.sp[e]; ea.rsp[e] = a.rsp[e]; ea.den[e] = a.den[e]; ea.rden[e] = a.rden[e]; }
    ea.size_m1[0] = hx; ea.size_m1[1] = hy; ea.size_m1[2] = hz;
    Lerp1D lj_e{0, 0, 1.0f, 0.0f}, lk_e{0, 0, 1.0f, 0.0f};
    if (elastic) {
      lj_e = lerp_index(j_lo + jv, a.nj, a.Jo, a.scj);
      lk_e = lerp_index(k_lo + kw, a.nk, a.Ko, a.sck);
    }
    fast_fill_tail<ELASTIC_POSSIBLE>(unsure, ea, a.mapping + (a.mapping_batched ? b * 12 : 0), elastic, f.cp, lj_e, lk_e, u0, static_cast<float>(j_lo + jv),
                                     static_cast<float>(k_lo + kw), in_chan, a.I, a.J, a.K, hx, hy, hz, fillv, out_chan, slab_b, urow, track ? &kmin : nullptr);
  };

  if (kind == kDescSlow) {  // box beyond the LDS budget / non-finite geometry: per-voxel evaluation, global gathers (rare)
    pipe_brick_frame(f, j_lo, k_lo);
    if (col_active) {
      for (int t = u0; t < u1; t++) {
        float x, y, z;
        fast_coord(f, static_cast<float>(t), static_cast<float>(jv), static_cast<float>(kw), x, y, z);
        const float val = lean_gather(in_chan, a.I, a.J, a.K, x, y, z, has_fill, fillv, hx, hy, hz, margin, t - u0, unsure);
        *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = val;
        if (!((unsure >> (t - u0)) & 1u)) kmin = min(kmin, float_to_key(val));  // (a re-decided voxel is counted by the tail)
      }
    }
    if (has_fill) lean_tail();
    if (track) publish_min();
    return;
  }

  // per-column constants, formed while the box is on its way
  const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
  float col3[3];
#pragma unroll
  for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      lj = lerp_index(j_lo + jv, a.nj, a.Jo, a.scj);
      lk = lerp_index(k_lo + kw, a.nk, a.Ko, a.sck);
    }
  }
  ColumnPlanes planes;
  planes.cell = -2;
#pragma unroll
  for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
  FastAddr ta;
  ta.sYb = bx.cpr * 16; ta.sXb = bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
  ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
  ta.base_f = static_cast<float>(static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile)));
  const float ox = static_cast<float>(bx.bx0), oy = static_cast<float>(bx.by0), oz = static_cast<float>(bx.za);

  float A3[3], B3[3];
  int run0 = u0;
  int run1 = fast_column_line(f, lj, lk, planes, run0, u1, u_ref, C3, col3, lane, A3, B3);
  if (ablate & 128) {  // experiment: every lane samples the box origin (same instructions, one LDS address per wave)
#pragma unroll
    for (int r = 0; r < 3; r++) { A3[r] = 0.25f; B3[r] = 0.0f; }
  }
  unsigned long long t_ready = 0ull, t_own = 0ull, t_landed = 0ull;
  if (ablate & 64) t_ready = __builtin_amdgcn_s_memtime();
  tile_dma_wait();
  if (ablate & 64) t_own = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (ablate & 64) t_landed = __builtin_amdgcn_s_memtime();
  if (col_active && u0 < u1 && !(ablate & 2)) {
    const bool needs_mask = has_fill & !bx.interior;
    for (;;) {
      char* o_run = out_chan + static_cast<int64_t>(run0) * slab_b;
      if (track)
        fast_sample_line<4, true>(run1 - run0, A3, B3, ta, o_run, urow, slab_b, needs_mask, ox, oy, oz, hx, hy, hz, fillv, kmin, margin, run0 - u0, unsure);
      else
        fast_sample_line<4, false>(run1 - run0, A3, B3, ta, o_run, urow, slab_b, needs_mask, ox, oy, oz, hx, hy, hz, fillv, kmin, margin, run0 - u0, unsure);
      run0 = run1;
      if (run0 >= u1) break;
      run1 = fast_column_line(f, lj, lk, planes, run0, u1, u_ref, C3, col3, lane, A3, B3);
    }
  }
  if (has_fill & !bx.interior) lean_tail();  // (block uniform)
  if (track) publish_min();
  if (ablate & 64) {  // instrumentation: the block's shader-clock stamps over the first row of its own output
    const unsigned long long t_sampled = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t_drained = __builtin_amdgcn_s_memtime();
    if (tid == 0 && nw >= 12 && i_count >= 1) {
      unsigned* w = reinterpret_cast<unsigned*>(out_chan + static_cast<int64_t>(i_begin) * slab_b) + (j_lo * a.Ko + k_lo);
      w[0] = 0x53544D50u;
      w[1] = static_cast<unsigned>(t_entry); w[2] = static_cast<unsigned>(t_entry >> 32);
      w[3] = static_cast<unsigned>(t_desc - t_entry); w[4] = static_cast<unsigned>(t_issued - t_entry);
      w[5] = static_cast<unsigned>(t_ready - t_entry); w[6] = static_cast<unsigned>(t_own - t_entry);
      w[7] = static_cast<unsigned>(t_landed - t_entry); w[8] = static_cast<unsigned>(t_sampled - t_entry);
      w[9] = static_cast<unsigned>(t_drained - t_entry); w[10] = static_cast<unsigned>(kind);
      w[11] = static_cast<unsigned>(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   // HW_ID
      w[12] = static_cast<unsigned>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));  // XCC_ID
    }
  }
}

}  // namespace tio
