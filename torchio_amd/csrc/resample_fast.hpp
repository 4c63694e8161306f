// resample_fast.hpp — EXPERIMENTAL kernels for TIO_PRECISION_FAST launches (float32 trilinear images only),
// opt-in through TIO_FAST_KERNEL=lean | stream | stream8.  The product path for FAST launches is the brick kernel's
// FAST instantiation in resample_tile.hpp; nothing here runs unless the variable is set.
//
// Why they exist.  The exact brick kernel spends ~114 vector instructions per voxel on reproducing the reference's
// float32 operation sequence and is ~80 % vector-ALU bound (profiles/r02_resample_sq.md: 239 M VALU wave-instructions
// per bench launch, 0.39 ms of issue time at the measured gfx950 rates of tests/native/valu_rates.cpp, 0.496 ms
// measured).  Intensities only owe the reference 1e-4 relative (BASELINE.json north_star), so the question of round 2
// was how far a launch of float32 trilinear images can go once the arithmetic is cut to the bone:
//   * coordinates as a LINE in the plane index per control cell, box-relative (x(t) = A + t B, the constant formed in
//     float64): 3 fma per voxel, no coordinate arrays, error ~2e-6 voxel on top of the reference's own rounding;
//   * the staged box from the <= 27 VERTICES of (planes x control cells) — the coordinate map is multilinear on each
//     such sub-box, so its extremes sit on vertices — instead of per-voxel min / max tracking;
//   * LDS-DMA addressed on the scalar unit (one wave instruction per group of rows of one x-plane, lane constants
//     cached per row length);
//   * seven fma lerps, the fill rule's in-bounds weight in its separable form and only in waves that really have a
//     column leaving the volume (a monotone line is interior when its two end planes are).
// That is 46 vector instructions per voxel all in (97 M per launch, 0.18 ms of issue time).
//
// What was measured (8 x 256^3 f32, affine launch; ablations in profiles/r02_resample_sq.md).  Two structures:
//   1. `resample_stream_kernel` — persistent blocks, one per (batch element, channel, 16 x 16 tile column), walking the
//      column along i in slabs of 8 planes through a two-buffer LDS ring: every wave issues its share of slab n + 1's
//      DMA, samples slab n (8 voxels of LDS reads in flight per lane), waits with a COUNTED vmcnt (its stores may stay
//      in flight), one s_barrier per slab; the slab table of an item (boxes, float64 constants) is computed up front,
//      one vertex per thread + LDS atomics.  0.49 - 0.50 ms: set-up alone 0.09, + DMA 0.25, + sampling 0.39 (alone).
//      With two blocks of four waves per CU (LDS: 2 x 2 x 28 KB) the sampling runs at two waves per SIMD — 6 cycles
//      per vector instruction on this chip — and the one-slab look-ahead leaves the ~2 us DMA latency exposed; eight
//      waves per block (`stream8`) fix the first and expose the second; a third buffer only fits with 4-plane slabs.
//   2. `resample_fastbrick_kernel` — the brick structure (independent 16^3 bricks, three resident blocks per CU) with
//      the lean arithmetic: 0.436 ms, the same as the product path (0.437): set-up 0.146 + DMA 0.10 + sampling 0.17,
//      which ADD UP: with three blocks per CU the throughput is 3 / (latency of one brick), and the latency chain
//      kernarg -> mapping -> box -> barrier -> DMA -> barrier -> sampling is what is left once the ALU work is gone.
//   3. (removed after measurement) persistent bricks with a helper wave preparing the next pass's descriptor in LDS:
//      0.47 - 0.54 ms — the helper's ~1.3 us (affine) / ~3.4 us (elastic) of single-wave work per pass became the chain.
// Neither beats the product path, so neither is the default; they stay as the reproducible A/B behind those numbers
// (tests/native/resample_bench --path fast) and as the starting point for the next attempt.
#pragma once

namespace tio {

typedef __attribute__((address_space(3))) const float* fast_lds_ptr;
typedef __attribute__((address_space(3))) float* fast_lds_wptr;

struct StreamItem {  // one work item: (batch element, image, channel, tile column)
  int b, im, c, jt, kt;
};

__device__ __forceinline__ StreamItem stream_decode(const ResampleArgs& a, int item) {
  StreamItem it;
  const int tiles = a.tiles_j * a.tiles_k;
  const int t = item % tiles;
  const int rest = item / tiles;
  it.kt = t % a.tiles_k;
  it.jt = t / a.tiles_k;
  int nch = 0;
  for (int i = 0; i < a.n_images; i++) nch += a.img[i].channels;
  int ch = rest % nch;
  it.b = rest / nch;
  it.im = 0;
  while (ch >= a.img[it.im].channels) { ch -= a.img[it.im].channels; it.im++; }
  it.c = ch;
  return it;
}

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
template <bool MASKED, int G, bool NOSTORE = false>
__device__ __forceinline__ void fast_sample_run(int n, float ax, float ay, float az, float bxs, float bys, float bzs, const FastAddr& ta,
                                                char* out_generic, unsigned urow, int64_t slab_b, float ox, float oy, float oz, float hx,
                                                float hy, float hz, float fillv) {
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
#pragma unroll
    for (int q = 0; q < G; q++) {
      vals[q] = fast_finish(ts[q]);
      if constexpr (MASKED) vals[q] = (fast_mask(ts[q], x0s[q] + ox, y0s[q] + oy, z0s[q] + oz, hx, hy, hz) > 0.5f) ? vals[q] : fillv;
    }
    if (tg + G <= n) {
#pragma unroll
      for (int q = 0; q < G; q++) {
        if (!NOSTORE || vals[q] == 1.2345e37f) *(global_float_ptr)(out_t + urow) = vals[q];
        out_t += slab_b;
        asm volatile("" : "+s"(out_t));  // one running pointer (2 scalar adds per plane), not G precomputed ones
      }
    } else {
#pragma unroll
      for (int q = 0; q < G; q++) {
        if (tg + q < n) {
          if (!NOSTORE || vals[q] == 1.2345e37f) *(global_float_ptr)(out_t + urow) = vals[q];
        }
        out_t += slab_b;
        asm volatile("" : "+s"(out_t));
      }
    }
    ax += bxg; ay += byg; az += bzg;
    __builtin_amdgcn_sched_barrier(0);
  }
}


// profiling only (TIO_PIPE_TRACE): the unmasked G = 4 loop with shader-clock stamps around its three parts —
// address arithmetic + LDS issue, the wait for the taps, interpolation + stores.  stamps[3 * iteration + {0, 1, 2}]
__device__ __forceinline__ void fast_sample_run_traced(int n, float ax, float ay, float az, float bxs, float bys, float bzs, const FastAddr& ta,
                                                       char* out_generic, unsigned urow, int64_t slab_b, unsigned long long* stamps, bool writer) {
  constexpr int G = 4;
  typedef __attribute__((address_space(1))) char* global_char_ptr;
  typedef __attribute__((address_space(1))) float* global_float_ptr;
  global_char_ptr out_t = (global_char_ptr)out_generic;
  const float bxg = static_cast<float>(G) * bxs, byg = static_cast<float>(G) * bys, bzg = static_cast<float>(G) * bzs;
  int k = 0;
#pragma unroll 1
  for (int tg = 0; tg < n; tg += G) {
    FastTaps ts[G];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int q = 0; q < G; q++) {
      const float qf = static_cast<float>(q);
      fast_issue(ts[q], __builtin_fmaf(qf, bxs, ax), __builtin_fmaf(qf, bys, ay), __builtin_fmaf(qf, bzs, az), ta);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < G; q++) {
      const float val = fast_finish(ts[q]);
      if (tg + q < n) *(global_float_ptr)(out_t + urow) = val;
      out_t += slab_b;
      asm volatile("" : "+s"(out_t));
    }
    ax += bxg; ay += byg; az += bzg;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = __builtin_readcyclecounter();
    if (writer && k < 4) { stamps[4 * k] = t0; stamps[4 * k + 1] = t1; stamps[4 * k + 2] = t2; stamps[4 * k + 3] = t3; }
    k++;
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
template <int GMAX>
__device__ __forceinline__ void fast_sample_line(int len, const float (&A3)[3], const float (&B3)[3], const FastAddr& ta, char* o, unsigned urow,
                                                 int64_t slab_b, bool needs_mask, float ox, float oy, float oz, float hx, float hy, float hz,
                                                 float fillv) {
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
    if (GMAX == 8 && len > 4) fast_sample_run<false, 8>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
    else fast_sample_run<false, 4>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
  } else {  // rare (waves on the volume's surface): four voxels in flight keep the register budget of the common loop
    fast_sample_run<true, 4>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, ox, oy, oz, hx, hy, hz, fillv);
  }
}


// ---- slab table of one work item (LDS): 16 dwords per slab ----------------------------------
enum : int {
  kSlabStaged = 0,   // box staged in LDS: sample from it
  kSlabOutside = 1,  // the slab sees nothing of the volume: fill (or 0)
  kSlabGather = 2    // non-finite geometry / box beyond the LDS budget: per-voxel global gather
};
enum : int {
  // raw extremes, filled by the vertex pass with LDS atomics (stored negated for the minima)
  kTNegXmin = 0, kTXmax, kTNegYmin, kTYmax, kTNegZmin, kTZmax, kTBad,
  // finalised (the raw slots are reused: they are dead once the box is known)
  kTCx = 0, kTCy = 1, kTCz = 2,  // float: mapping of (u0, j_lo, k_lo) relative to the box origin
  kTKind = 7, kTBx0, kTBy0, kTZa, kTLx, kTLy, kTCpr, kTInterior, kTPad,
  kTableInts = 16
};
constexpr int kStreamMaxSlabs = 128;  // slabs per item (Io <= 1024 at S = 8; larger volumes shrink nothing: they use the brick kernel)

// block barrier without the vmcnt(0) that __syncthreads() implies: LDS traffic of this wave done,
// vector-memory operations (the DMA ring, the output stores) stay in flight
__device__ __forceinline__ void stream_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void stream_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most `keep` of this wave's vector-memory operations are outstanding (in issue order);
// waiting for fewer than allowed is always safe
__device__ __forceinline__ void stream_vm_wait(int keep) {
#define TIO_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (keep < 0 ? 0 : (keep > 31 ? 31 : keep)) {
    TIO_VMW(1) TIO_VMW(2) TIO_VMW(3) TIO_VMW(4) TIO_VMW(5) TIO_VMW(6) TIO_VMW(7) TIO_VMW(8) TIO_VMW(9) TIO_VMW(10) TIO_VMW(11)
    TIO_VMW(12) TIO_VMW(13) TIO_VMW(14) TIO_VMW(15) TIO_VMW(16) TIO_VMW(17) TIO_VMW(18) TIO_VMW(19) TIO_VMW(20) TIO_VMW(21)
    TIO_VMW(22) TIO_VMW(23) TIO_VMW(24) TIO_VMW(25) TIO_VMW(26) TIO_VMW(27) TIO_VMW(28) TIO_VMW(29) TIO_VMW(30) TIO_VMW(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TIO_VMW
}

struct StreamBox {
  int kind, bx0, by0, za, Lx, Ly, cpr, interior;
};

__device__ __forceinline__ StreamBox stream_load_box(const int* t) {
  StreamBox b;
  b.kind = __builtin_amdgcn_readfirstlane(t[kTKind]); b.bx0 = __builtin_amdgcn_readfirstlane(t[kTBx0]);
  b.by0 = __builtin_amdgcn_readfirstlane(t[kTBy0]); b.za = __builtin_amdgcn_readfirstlane(t[kTZa]);
  b.Lx = __builtin_amdgcn_readfirstlane(t[kTLx]); b.Ly = __builtin_amdgcn_readfirstlane(t[kTLy]);
  b.cpr = __builtin_amdgcn_readfirstlane(t[kTCpr]); b.interior = __builtin_amdgcn_readfirstlane(t[kTInterior]);
  return b;
}

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
      else if (in_box) *reinterpret_cast<float4*>(l + 4 * lane) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      g += group_b; l += dgroup;
    }
    gp += NW * plane_b; lp += NW * dplane;
  }
  return issued;
}

// NPW: waves per 64-column group; the planes of a slab are split between them (more waves per
// LDS byte: the occupancy of this kernel is bounded by LDS, not by registers)
// NB: slab buffers in the ring = DMA look-ahead + 1
template <bool ELASTIC_POSSIBLE, int TJ, int TK, int NPW, int NB>
__global__ __launch_bounds__(TJ* TK* NPW, NPW == 1 ? 2 : 4) void resample_stream_kernel(const ResampleArgs a, int n_items) {
  constexpr int NT = TJ * TK * NPW, NW = NT / 64;
  static_assert(NB == 2, "ring depth (a third buffer forces 4-plane slabs into the LDS budget and measured slower)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_table = reinterpret_cast<int*>(smem);                 // kStreamMaxSlabs x kTableInts
  int* s_flags = s_table + kStreamMaxSlabs * kTableInts;       // [0]: some slab of the item does not fit
  float* s_cp = smem + kStreamMaxSlabs * kTableInts + 16;
  const int cp_slot_floats = a.cp_lds;  // floats reserved for the control points (multiple of 4; 0 without elastic)
  float* s_buf = s_cp + cp_slot_floats;
  const int cap = a.tile_cap;  // floats per slab buffer (two of them)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ctid = tid % (TJ * TK);
  const int pg = __builtin_amdgcn_readfirstlane(tid / (TJ * TK));  // which share of a slab's planes this wave samples
  const int tk = ctid % TK, tj = ctid / TK;

  // ---- the block's share of the work items: XCD x takes a contiguous range, its blocks interleave ----
  const int nblk = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = (nblk + 7 - xcd) >> 3;  // blocks that live on this XCD
  const int q_items = n_items / 8, r_items = n_items % 8;
  const int range0 = xcd * q_items + min(xcd, r_items), range1 = range0 + q_items + (xcd < r_items ? 1 : 0);

  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};  // (S_own - 1) / max(S_norm - 1, 1)
  const unsigned buf0_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_buf));
  int cp_b = -1;  // batch element whose control points sit in LDS

  for (int item = range0 + slot; item < range1; item += per_xcd) {
    // ================= item set-up: constants, control points, slab table =================
    const StreamItem it = stream_decode(a, item);
    FastFrame f;
    f.j_lo = it.jt * TJ; f.k_lo = it.kt * TK;
    const int nv = min(TJ, a.Jo - f.j_lo), nw = min(TK, a.Ko - f.k_lo);
    const bool col_active = (tj < nv) & (tk < nw);
    const bool wave_stores = __builtin_amdgcn_ballot_w64(col_active) != 0ull;
    const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
    const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
    const ImgArgs& g = a.img[it.im];
    const int64_t bc = static_cast<int64_t>(it.b) * g.channels + it.c;
    char* out_tile = static_cast<char*>(g.out) + bc * n_out * 4;
    const float* in_chan = static_cast<const float*>(g.in) + bc * n_in;
    const int col_off = (f.j_lo + jv) * a.Ko + (f.k_lo + kw);
    const unsigned urow = static_cast<unsigned>(col_off) * 4u;
    const bool has_fill = g.fill != nullptr;
    typedef __attribute__((address_space(4))) const float* const_float_ptr;
    const float fillv = has_fill ? ((const_float_ptr)g.fill)[it.c] : 0.0f;
    const float* fill_ptr = has_fill ? g.fill + it.c : nullptr;

    stream_barrier();  // the previous item is fully consumed: table, control points and buffers are free
    if (a.passthrough != nullptr && a.passthrough[it.b] != 0) {  // gated-out element: bit-exact copy
      if (col_active)
        for (int t = pg; t < a.Io; t += NPW)
          *reinterpret_cast<float*>(out_tile + t * slab_b + urow) = in_chan[static_cast<int64_t>(t) * slab + col_off];
      continue;
    }
    bool weird = false;
    {
      const float* m = a.mapping + (a.mapping_batched ? it.b * 12 : 0);
#pragma unroll
      for (int q = 0; q < 12; q++) {
        const float mv = m[q];
        weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;
        f.m[q] = mv * ratio[q >> 2];
      }
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
      f.c[r] = static_cast<double>(f.m[4 * r + 1]) * f.j_lo + static_cast<double>(f.m[4 * r + 2]) * f.k_lo + static_cast<double>(f.m[4 * r + 3]);
    f.affine_first = a.affine_first != 0;
    f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
    f.cp = (fast_lds_ptr)s_cp;
    f.elastic = false;
#pragma unroll
    for (int e = 0; e < 3; e++) f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f);
    Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
    if constexpr (ELASTIC_POSSIBLE) {
      f.elastic = !(a.cp_skip != nullptr && a.cp_skip[it.b] != 0);
      if (f.elastic) {
        if (cp_b != it.b) {
          const int n_cp = a.ni * a.nj * a.nk * 3;
          const float* cpg = a.cp + (a.cp_batched ? static_cast<int64_t>(it.b) * n_cp : 0);
          for (int t = tid; t < n_cp; t += NT) s_cp[t] = cpg[t];
          cp_b = it.b;
        }
        lj = lerp_index(f.j_lo + jv, a.nj, a.Jo, a.scale_j);
        lk = lerp_index(f.k_lo + kw, a.nk, a.Ko, a.scale_k);
      }
    }

    // ---- slab table: S planes per slab, the largest of 8 / 4 / 2 / 1 whose boxes all fit ----
    const int n_vert = f.elastic ? 27 : 8;
    int S = 8, n_slabs = 0;
    for (;;) {
      n_slabs = (a.Io + S - 1) / S;
      for (int t = tid; t < n_slabs * kTableInts; t += NT) {
        const int fld = t & (kTableInts - 1);
        s_table[t] = fld < 6 ? -0x40000000 : 0;  // maxima of (negated) minima / maxima start at -inf; bad = 0
      }
      if (tid == 0) s_flags[0] = 0;
      stream_vm_drain();  // the control-point copy (when there was one)
      stream_barrier();
      // one vertex of (slab x control cells) per thread: extremes through LDS atomics
      for (int gidx = tid; gidx < n_slabs * n_vert; gidx += NT) {
        const int sl = gidx / n_vert, vtx = gidx - sl * n_vert;
        const int u_lo = sl * S, u_hi = min(u_lo + S, a.Io) - 1;
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
        constexpr float kMargin = 1.0f / 64.0f;  // the per-voxel lines differ from these vertex values by rounding only
        const bool bad = !(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f) | dense;
        const float capx = hx + 1.0f + kTileFar, capy = hy + 1.0f + kTileFar, capz = hz + 1.0f + kTileFar;
        int* t = s_table + sl * kTableInts;
        atomicMax(&t[kTNegXmin], -static_cast<int>(fminf(fmaxf(floorf(x - kMargin), -kTileFar), capx)));
        atomicMax(&t[kTXmax], static_cast<int>(fminf(fmaxf(floorf(x + kMargin), -kTileFar), capx)));
        atomicMax(&t[kTNegYmin], -static_cast<int>(fminf(fmaxf(floorf(y - kMargin), -kTileFar), capy)));
        atomicMax(&t[kTYmax], static_cast<int>(fminf(fmaxf(floorf(y + kMargin), -kTileFar), capy)));
        atomicMax(&t[kTNegZmin], -static_cast<int>(fminf(fmaxf(floorf(z - kMargin), -kTileFar), capz)));
        atomicMax(&t[kTZmax], static_cast<int>(fminf(fmaxf(floorf(z + kMargin), -kTileFar), capz)));
        if (bad) atomicOr(&t[kTBad], 1);
      }
      stream_barrier();
      for (int sl = tid; sl < n_slabs; sl += NT) {
        int* t = s_table + sl * kTableInts;
        const int xmin = -t[kTNegXmin], xmax = t[kTXmax], ymin = -t[kTNegYmin], ymax = t[kTYmax], zmin = -t[kTNegZmin], zmax = t[kTZmax];
        const bool wrd = weird | (t[kTBad] != 0);
        const int interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
        const int outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
        const int za = zmin & ~3, Lx = xmax + 2 - xmin, Ly = ymax + 2 - ymin, Lz = ((zmax + 1 + 4) & ~3) - za;
        const bool fits = !wrd && (Lz <= 256) && (Lx <= 4096) && (Ly <= 4096) && (static_cast<int64_t>(Lx) * Ly * Lz <= static_cast<int64_t>(cap));
        // mapping of (u0, j_lo, k_lo) relative to the box origin, in float64: what the per-voxel float32 lines start from
        const double org[3] = {static_cast<double>(xmin), static_cast<double>(ymin), static_cast<double>(za)};
#pragma unroll
        for (int r = 0; r < 3; r++) t[kTCx + r] = __float_as_int(static_cast<float>(static_cast<double>(f.m[4 * r]) * (sl * S) + f.c[r] - org[r]));
        t[kTKind] = outside ? kSlabOutside : (fits ? kSlabStaged : kSlabGather);
        t[kTBx0] = xmin; t[kTBy0] = ymin; t[kTZa] = za; t[kTLx] = Lx; t[kTLy] = Ly; t[kTCpr] = Lz >> 2; t[kTInterior] = interior;
        if (!fits && !outside && !wrd && S > 1) atomicOr(&s_flags[0], 1);
      }
      stream_barrier();
      // (halving S again must not overflow the table either)
      if (__builtin_amdgcn_readfirstlane(s_flags[0]) == 0 || S == 1 || (a.Io + (S >> 1) - 1) / (S >> 1) > kStreamMaxSlabs) break;
      S >>= 1;
      stream_barrier();  // everybody has read the flag before the table is reset
    }

    // ================= the pipeline: DMA of slab n + 1 in flight while slab n is sampled =================
    // per-column constants of the item: the column's own offset inside the tile, through the mapping
    float col3[3];
#pragma unroll
    for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
    StageLanes sl;
    sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
    // prologue: the first NB - 1 slabs are requested, the first one has landed
    StreamBox cur = stream_load_box(s_table);
    int ahead = 0;  // DMA instructions of this wave still allowed in flight at the end of an iteration (NB == 3: slab n + 2's)
    if (cur.kind == kSlabStaged && !(a.ablate & 1)) stream_stage<NW>(s_buf, in_chan, cur, a.I, a.J, a.K, wave, lane, sl);
    stream_vm_drain();
    StreamBox mid{};  // NB == 3: slab n + 1 (requested one iteration ago)
    if constexpr (NB == 3) {
      if (n_slabs > 1) {
        mid = stream_load_box(s_table + kTableInts);
        if (mid.kind == kSlabStaged && !(a.ablate & 1)) ahead = stream_stage<NW>(s_buf + cap, in_chan, mid, a.I, a.J, a.K, wave, lane, sl);
      }
    }
    stream_barrier();
    int cached_cell = -2;
    float P0[3] = {0.f, 0.f, 0.f}, P1[3] = {0.f, 0.f, 0.f};
    for (int n = 0; n < n_slabs; n++) {
      const int u0 = n * S, cnt = min(S, a.Io - u0);
      const int* tab = s_table + n * kTableInts;
      StreamBox nxt{};
      const bool have_next = n + NB - 1 < n_slabs;
      int issued_now = 0;
      if (have_next) {
        nxt = stream_load_box(tab + (NB - 1) * kTableInts);
        if (nxt.kind == kSlabStaged && !(a.ablate & 1))
          issued_now = stream_stage<NW>(s_buf + ((n + NB - 1) % NB) * cap, in_chan, nxt, a.I, a.J, a.K, wave, lane, sl);
      }
      char* out_t = out_tile + static_cast<int64_t>(u0) * slab_b;  // block uniform
      int stores = -1;  // vector-memory operations issued after the DMA share (-1: unknown, drain)
      // this wave's planes of the slab
      const int per = (cnt + NPW - 1) / NPW;
      const int my0 = min(u0 + pg * per, u0 + cnt), my1 = min(my0 + per, u0 + cnt);
      if (cur.kind == kSlabStaged) {
        stores = wave_stores ? my1 - my0 : 0;
        if (col_active && !(a.ablate & 2)) {
          FastAddr ta;
          ta.sYb = cur.cpr * 16; ta.sXb = cur.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
          ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
          ta.base_f = static_cast<float>(buf0_addr + static_cast<unsigned>((n & 1) * cap) * 4u);
          const float C3[3] = {__int_as_float(tab[kTCx]), __int_as_float(tab[kTCy]), __int_as_float(tab[kTCz])};
          // runs of planes inside one control cell (one run without elastic; at most two with: cells are
          // at least S planes deep or the launcher does not pick this kernel)
          int run0 = my0;
          while (run0 < my1) {
            int run1 = my1;
            float A3[3], B3[3];
            const float du = static_cast<float>(run0 - u0);
            bool lines_done = false;
            if constexpr (ELASTIC_POSSIBLE) {
              if (f.elastic) {
                const int cmax = a.ni > 1 ? a.ni - 2 : 0;
                const int cell_l = min(max(static_cast<int>(floorf(f.sci * static_cast<float>(run0 + lane))), 0), cmax);
                const int cell = __builtin_amdgcn_readlane(cell_l, 0);
                const unsigned long long later = __builtin_amdgcn_ballot_w64((lane < run1 - run0) & (cell_l > cell));
                if (later != 0ull) run1 = run0 + __builtin_ctzll(later);
                if (cell != cached_cell) {  // (j, k)-lerped control planes at the two ends of the cell, from the LDS copy
                  const int s_i = a.nj * a.nk * 3, s_j = a.nk * 3;
                  const int c1 = min(cell + 1, a.ni - 1);
#pragma unroll
                  for (int e = 0; e < 3; e++) {
                    fast_lds_ptr q = f.cp + e;
                    const float a00 = q[cell * s_i + lj.i0 * s_j + lk.i0 * 3], a01 = q[cell * s_i + lj.i0 * s_j + lk.i1 * 3];
                    const float a10 = q[cell * s_i + lj.i1 * s_j + lk.i0 * 3], a11 = q[cell * s_i + lj.i1 * s_j + lk.i1 * 3];
                    const float b00 = q[c1 * s_i + lj.i0 * s_j + lk.i0 * 3], b01 = q[c1 * s_i + lj.i0 * s_j + lk.i1 * 3];
                    const float b10 = q[c1 * s_i + lj.i1 * s_j + lk.i0 * 3], b11 = q[c1 * s_i + lj.i1 * s_j + lk.i1 * 3];
                    const float a0 = __builtin_fmaf(lk.l1, a01 - a00, a00), a1 = __builtin_fmaf(lk.l1, a11 - a10, a10);
                    const float b0 = __builtin_fmaf(lk.l1, b01 - b00, b00), b1 = __builtin_fmaf(lk.l1, b11 - b10, b10);
                    P0[e] = __builtin_fmaf(lj.l1, a1 - a0, a0);
                    P1[e] = __builtin_fmaf(lj.l1, b1 - b0, b0);
                  }
                  cached_cell = cell;
                }
                // d(run0 + t) = P0 + (sci (run0 + t) - cell) (P1 - P0)
                const float l_ref = fminf(fmaxf(__builtin_fmaf(f.sci, static_cast<float>(run0), -static_cast<float>(cell)), 0.0f), 1.0f);
                float D0[3], D1[3];
#pragma unroll
                for (int e = 0; e < 3; e++) {
                  const float dP = P1[e] - P0[e];
                  D0[e] = __builtin_fmaf(l_ref, dP, P0[e]) * f.dsc[e];
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
                lines_done = true;
              }
            }
            if (!lines_done) {
#pragma unroll
              for (int r = 0; r < 3; r++) { A3[r] = __builtin_fmaf(f.m[4 * r], du, C3[r] + col3[r]); B3[r] = f.m[4 * r]; }
            }
            char* o = out_t + static_cast<int64_t>(run0 - u0) * slab_b;
            const int len = run1 - run0;
            // The fill rule only matters where a tap can leave the volume.  Each coordinate of the line is
            // monotone, so a column whose two END planes keep all first taps in [0, S - 2] is interior for the
            // whole run; the wave takes the masked path only if one of its columns is not.
            bool masked = false;
            if (has_fill && !cur.interior) {
              const float el = static_cast<float>(len - 1);
              const float ox = static_cast<float>(cur.bx0), oy = static_cast<float>(cur.by0), oz = static_cast<float>(cur.za);
              const float xa = A3[0] + ox, xb = __builtin_fmaf(el, B3[0], A3[0]) + ox;
              const float ya = A3[1] + oy, yb = __builtin_fmaf(el, B3[1], A3[1]) + oy;
              const float za_ = A3[2] + oz, zb = __builtin_fmaf(el, B3[2], A3[2]) + oz;
              const bool inside = (fminf(xa, xb) >= 0.0f) & (fmaxf(xa, xb) < hx) & (fminf(ya, yb) >= 0.0f) & (fmaxf(ya, yb) < hy) &
                                  (fminf(za_, zb) >= 0.0f) & (fmaxf(za_, zb) < hz);
              masked = __builtin_amdgcn_ballot_w64(!inside) != 0ull;
            }
            if (a.ablate & 4) {  // profiling only: sample, never store
              fast_sample_run<false, 4, true>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
              if (len > 4) fast_sample_run<false, 4, true>(len - 4, A3[0] + 4.f * B3[0], A3[1] + 4.f * B3[1], A3[2] + 4.f * B3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
            } else if (!masked) {
              if (NPW == 1 && len > 4) fast_sample_run<false, 8>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
              else fast_sample_run<false, 4>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, 0.f, 0.f, 0.f, hx, hy, hz, fillv);
            } else {
              const float ox = static_cast<float>(cur.bx0), oy = static_cast<float>(cur.by0), oz = static_cast<float>(cur.za);
              if (NPW == 1 && len > 4) fast_sample_run<true, 8>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, ox, oy, oz, hx, hy, hz, fillv);
              else fast_sample_run<true, 4>(len, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, o, urow, slab_b, ox, oy, oz, hx, hy, hz, fillv);
            }
            run0 = run1;
          }
        }
        if (a.ablate & 6) stores = 0;
      } else if (cur.kind == kSlabOutside) {
        stores = wave_stores ? my1 - my0 : 0;
        if (col_active)
          for (int t = my0 - u0; t < my1 - u0; t++) *reinterpret_cast<float*>(out_t + t * slab_b + urow) = fillv;
      } else {  // kSlabGather: rare — full per-voxel evaluation, per-tap bounds, global gathers
        if (col_active) {
          ImgArgs g1 = g;  // this item's single channel
          g1.in = in_chan; g1.out = out_tile; g1.channels = 1; g1.fill = fill_ptr;
          for (int t = my0 - u0; t < my1 - u0; t++) {
            float x, y, z;
            fast_coord(f, static_cast<float>(u0 + t), fv, fw, x, y, z);
            gather_voxel<0>(g1, a, 0, n_in, n_out, (u0 + t) * slab + col_off, x, y, z, false);
          }
        }
      }
      // Slab n + 1's share must have landed before the barrier.  In issue order it is followed by this
      // iteration's request (NB == 3 only: slab n + 2) and this iteration's stores: that many
      // vector-memory operations may stay in flight.
      if (NB == 2) {
        if (have_next) { if (stores >= 0) stream_vm_wait(stores); else stream_vm_drain(); }
        cur = nxt;
      } else {
        if (n + 1 < n_slabs) { if (stores >= 0) stream_vm_wait(stores + issued_now); else stream_vm_drain(); }
        cur = mid; mid = nxt;
      }
      stream_barrier();
    }
  }
}

// =====================================================================================================================
// The FAST brick kernel: the brick structure of resample_tile.hpp (independent 16^3 bricks, three resident blocks per
// CU, the DMA in flight while the per-column constants are formed) with this file's lean arithmetic:
//   * the box of the brick from <= 27 vertices evaluated by ONE wave (a DPP wave reduction, 7 ints through LDS, one
//     barrier) instead of per-voxel min / max tracking and a block reduction;
//   * no coordinate arrays: each column's coordinates are a line per control cell, 3 fma per voxel;
//   * scalar-addressed LDS-DMA (stream_stage), 8 voxels of LDS reads in flight per lane, masked loop only in waves
//     that really have a column leaving the volume.
// =====================================================================================================================
typedef FastFrameT<const float*> FastFrameG;

// One brick, start to finish (every block-level step is uniform): the body of the lean brick kernel, also the
// fall-back of the pipelined kernel for bricks whose box needs several passes.
template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK>
__device__ __forceinline__ void fastbrick_process(const ResampleArgs& a, float* smem, int b, int it, int jt, int kt, bool first_barrier) {
  constexpr int NT = TJ * TK, NW = NT / 64;
  int* s_box = reinterpret_cast<int*>(smem);  // 7 raw extremes of the pass (wave 0 -> everybody)
  float* s_tile = smem + 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid % TK, tj = tid / TK;
  const int i_begin = it * TI, j_lo = jt * TJ, k_lo = kt * TK;
  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
  const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
  const unsigned urow = static_cast<unsigned>(col_off) * 4u;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];

  if (a.passthrough != nullptr && a.passthrough[b] != 0) {  // gated-out element: bit-exact copy
    if (col_active) {
      for (int im = 0; im < a.n_images; im++) {
        const ImgArgs& g = a.img[im];
        for (int c = 0; c < g.channels; c++) {
          const int64_t off = (static_cast<int64_t>(b) * g.channels + c) * n_out + col_off;
          for (int t = 0; t < i_count; t++)
            static_cast<float*>(g.out)[off + static_cast<int64_t>(i_begin + t) * slab] = static_cast<const float*>(g.in)[off + static_cast<int64_t>(i_begin + t) * slab];
        }
      }
    }
    return;
  }

  // ---- the frame of this brick's tile column ----
  FastFrameG f;
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};  // (S_own - 1) / max(S_norm - 1, 1)
  bool weird = false;
  {
    const float* m = a.mapping + (a.mapping_batched ? b * 12 : 0);
#pragma unroll
    for (int q = 0; q < 12; q++) {
      const float mv = m[q];
      weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;
      f.m[q] = mv * ratio[q >> 2];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
    f.c[r] = static_cast<double>(f.m[4 * r + 1]) * j_lo + static_cast<double>(f.m[4 * r + 2]) * k_lo + static_cast<double>(f.m[4 * r + 3]);
  f.j_lo = j_lo; f.k_lo = k_lo;
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.cp = nullptr;
  f.elastic = false;
#pragma unroll
  for (int e = 0; e < 3; e++) f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f);
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    f.elastic = !(a.cp_skip != nullptr && a.cp_skip[b] != 0);
    if (f.elastic) {
      f.cp = a.cp + (a.cp_batched ? static_cast<int64_t>(b) * (a.ni * a.nj * a.nk * 3) : 0);
      lj = lerp_index(j_lo + jv, a.nj, a.Jo, a.scale_j);
      lk = lerp_index(k_lo + kw, a.nk, a.Ko, a.scale_k);
    }
  }
  float col3[3];
#pragma unroll
  for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
  const int n_vert = f.elastic ? 27 : 8;
  const unsigned tile_lds_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile));
  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
  ColumnPlanes planes;
  planes.cell = -2;
#pragma unroll
  for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }

  // ---- passes over the planes: the largest leading range whose box fits the LDS budget ----
  int u0 = i_begin;
  const int u_end = i_begin + i_count;
  while (u0 < u_end) {
    int n = u_end - u0;
    StreamBox bx{};
    bool fits = false, wrd = false;
    for (;;) {
      if (!first_barrier) __syncthreads();  // s_box (and the brick area) are free again
      first_barrier = false;
      if (wave == 0) {  // one vertex of (planes x control cells) per lane, extremes by a DPP wave reduction
        const int vtx = lane < n_vert ? lane : 0;
        const int u_lo = u0, u_hi = u0 + n - 1;
        int du, dv, dw;
        if (f.elastic) { du = vtx % 3; dv = (vtx / 3) % 3; dw = vtx / 9; } else { du = vtx & 1; dv = (vtx >> 1) & 1; dw = vtx >> 2; }
        bool dense = false;
        float u = du == 0 ? static_cast<float>(u_lo) : static_cast<float>(u_hi);
        float v = dv == 0 ? 0.0f : static_cast<float>(nv - 1);
        float w = dw == 0 ? 0.0f : static_cast<float>(nw - 1);
        if (f.elastic) {
          if (du == 2) u = fast_breakpoint(f.sci, f.ni, u_lo, u_hi, dense);
          if (dv == 2) v = fast_breakpoint(f.scj, f.nj, j_lo, j_lo + nv - 1, dense) - static_cast<float>(j_lo);
          if (dw == 2) w = fast_breakpoint(f.sck, f.nk, k_lo, k_lo + nw - 1, dense) - static_cast<float>(k_lo);
        }
        float x, y, z;
        fast_coord(f, u, v, w, x, y, z);
        constexpr float kMargin = 1.0f / 64.0f;  // the per-voxel lines differ from these vertex values by rounding only
        const bool bad = !(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f) | dense;
        const float capx = hx + 1.0f + kTileFar, capy = hy + 1.0f + kTileFar, capz = hz + 1.0f + kTileFar;
        int r[7];
        r[0] = -static_cast<int>(fminf(fmaxf(floorf(x - kMargin), -kTileFar), capx));
        r[1] = static_cast<int>(fminf(fmaxf(floorf(x + kMargin), -kTileFar), capx));
        r[2] = -static_cast<int>(fminf(fmaxf(floorf(y - kMargin), -kTileFar), capy));
        r[3] = static_cast<int>(fminf(fmaxf(floorf(y + kMargin), -kTileFar), capy));
        r[4] = -static_cast<int>(fminf(fmaxf(floorf(z - kMargin), -kTileFar), capz));
        r[5] = static_cast<int>(fminf(fmaxf(floorf(z + kMargin), -kTileFar), capz));
        r[6] = bad ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 7; q++) r[q] = wave_max_i32(r[q]);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 7; q++) s_box[q] = r[q];
        }
      }
      __syncthreads();
      const int xmin = -__builtin_amdgcn_readfirstlane(s_box[0]), xmax = __builtin_amdgcn_readfirstlane(s_box[1]);
      const int ymin = -__builtin_amdgcn_readfirstlane(s_box[2]), ymax = __builtin_amdgcn_readfirstlane(s_box[3]);
      const int zmin = -__builtin_amdgcn_readfirstlane(s_box[4]), zmax = __builtin_amdgcn_readfirstlane(s_box[5]);
      wrd = weird | (__builtin_amdgcn_readfirstlane(s_box[6]) != 0);
      bx.interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
      const int outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
      bx.bx0 = xmin; bx.by0 = ymin; bx.za = zmin & ~3;
      bx.Lx = xmax + 2 - xmin; bx.Ly = ymax + 2 - ymin;
      const int Lz = ((zmax + 1 + 4) & ~3) - bx.za;
      bx.cpr = Lz >> 2;
      fits = !wrd && (Lz <= 256) && (bx.Lx <= 4096) && (bx.Ly <= 4096) && (static_cast<int64_t>(bx.Lx) * bx.Ly * Lz <= static_cast<int64_t>(a.tile_cap));
      bx.kind = outside ? kSlabOutside : (fits ? kSlabStaged : kSlabGather);
      if (fits | (outside != 0) | wrd | (n <= 1)) break;
      n = (n + 1) >> 1;
    }
    const int u1 = u0 + n;

    for (int im = 0; im < a.n_images; im++) {
      const ImgArgs& g = a.img[im];
      typedef __attribute__((address_space(4))) const float* const_float_ptr;
      for (int c = 0; c < g.channels; c++) {
        const int64_t bc = static_cast<int64_t>(b) * g.channels + c;
        const bool has_fill = g.fill != nullptr;
        const float fillv = has_fill ? ((const_float_ptr)g.fill)[c] : 0.0f;
        char* out_chan = static_cast<char*>(g.out) + bc * n_out * 4;
        const float* in_chan = static_cast<const float*>(g.in) + bc * n_in;
        if (bx.kind == kSlabOutside) {
          if (col_active)
            for (int t = u0; t < u1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = fillv;
          continue;
        }
        if (bx.kind == kSlabGather) {  // rare: full per-voxel evaluation, per-tap bounds, global gathers
          if (col_active) {
            ImgArgs g1 = g;
            g1.in = in_chan; g1.out = out_chan; g1.channels = 1; g1.fill = has_fill ? g.fill + c : nullptr;
            for (int t = u0; t < u1; t++) {
              float x, y, z;
              fast_coord(f, static_cast<float>(t), fv, fw, x, y, z);
              gather_voxel<0>(g1, a, 0, n_in, n_out, t * slab + col_off, x, y, z, false);
            }
          }
          continue;
        }
        if (im + c > 0) __syncthreads();  // the previous channel's taps are read
        if (!(a.ablate & 1)) stream_stage<NW>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl);
        // while the brick is on its way: the first line of this column
        float C3[3];
        {
          const double org[3] = {static_cast<double>(bx.bx0), static_cast<double>(bx.by0), static_cast<double>(bx.za)};
#pragma unroll
          for (int r = 0; r < 3; r++) C3[r] = static_cast<float>(static_cast<double>(f.m[4 * r]) * u0 + f.c[r] - org[r]);
        }
        float A3[3], B3[3];
        int run0 = u0;
        int run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
        FastAddr ta;
        ta.sYb = bx.cpr * 16; ta.sXb = bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
        ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
        ta.base_f = static_cast<float>(tile_lds_addr);
        const float ox = static_cast<float>(bx.bx0), oy = static_cast<float>(bx.by0), oz = static_cast<float>(bx.za);
        tile_dma_wait();
        __syncthreads();
        if (col_active && !(a.ablate & 2)) {
          for (;;) {
            fast_sample_line<4>(run1 - run0, A3, B3, ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b, has_fill & !bx.interior, ox, oy,
                                oz, hx, hy, hz, fillv);
            run0 = run1;
            if (run0 >= u1) break;
            run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
          }
        }
      }
    }
    u0 = u1;
  }
}

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int OCC>
__global__ __launch_bounds__(TJ* TK, (TJ * TK) / 256 * OCC) void resample_fastbrick_kernel(const ResampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned t1 = fastdiv(tile, a.magic_k, a.tiles_k);
  const int kt = tile - t1 * a.tiles_k;
  const unsigned t2 = fastdiv(t1, a.magic_j, a.tiles_j);
  const int jt = t1 - t2 * a.tiles_j;
  const unsigned t3 = fastdiv(t2, a.magic_i, a.tiles_i);
  const int it = t2 - t3 * a.tiles_i;
  fastbrick_process<ELASTIC_POSSIBLE, TI, TJ, TK>(a, smem, static_cast<int>(t3), it, jt, kt, true);
}

// =====================================================================================================================
// The PIPELINED brick kernel (TIO_FAST_KERNEL=pipe): persistent blocks walking a list of 16^3 bricks.  The brick
// kernels above pay their three phases one after the other — set-up (mapping, box), DMA round trip, sampling — and
// only three blocks fit a CU's LDS, so the chain's latency IS the throughput.  Here every wave works out the NEXT
// brick's box (registers only: one vertex per lane, DPP reductions, no LDS, no barrier) while the current brick's
// DMA is in flight, and the next DMA is issued the moment the sampling of the current brick has been left behind
// by every wave: per brick max(DMA, set-up) + sampling instead of their sum.  Launches of one single-channel image;
// bricks whose box needs several passes (or a non-finite geometry) drop out of the pipeline into fastbrick_process.
// =====================================================================================================================
// out-of-line copy for the pipelined kernel's rare paths: keeps their registers out of the pipeline's budget
// (arguments of a real call travel in vector registers: the launch arguments are re-read from the kernel's own
// argument segment — ResampleArgs is the kernel's first parameter — and the brick indices made scalar again)
template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK>
__device__ __attribute__((noinline)) void fastbrick_process_cold(int b, int it, int jt, int kt, int first_barrier) {
  typedef __attribute__((address_space(4))) const ResampleArgs* const_args_ptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const_args_ptr ap = (const_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  fastbrick_process<ELASTIC_POSSIBLE, TI, TJ, TK>(*(const ResampleArgs*)ap, smem, __builtin_amdgcn_readfirstlane(b), __builtin_amdgcn_readfirstlane(it),
                                                  __builtin_amdgcn_readfirstlane(jt), __builtin_amdgcn_readfirstlane(kt),
                                                  __builtin_amdgcn_readfirstlane(first_barrier) != 0);
}

struct PipePlan {
  int b, it, jt, kt;
  int i_begin, j_lo, k_lo, i_count, nv, nw;
  int fast;  // the brick's whole plane range in one pass: a staged box, or nothing of the volume in sight
  StreamBox bx;
};

template <int TI, int TJ, int TK>
__device__ __forceinline__ void pipe_decode(const ResampleArgs& a, unsigned tile, PipePlan& p) {
  const unsigned t1 = fastdiv(tile, a.magic_k, a.tiles_k);
  p.kt = tile - t1 * a.tiles_k;
  const unsigned t2 = fastdiv(t1, a.magic_j, a.tiles_j);
  p.jt = t1 - t2 * a.tiles_j;
  const unsigned t3 = fastdiv(t2, a.magic_i, a.tiles_i);
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

// box of the planes [u0, u0 + n) of the brick framed in f, by THIS wave alone (block uniform result)
__device__ __forceinline__ void pipe_box(const ResampleArgs& a, const FastFrameG& f, int u0, int n, int nv, int nw, int lane, bool weird,
                                         PipePlan& p) {
  const int n_vert = f.elastic ? 27 : 8;
  const int vtx = lane < n_vert ? lane : 0;
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
  constexpr float kMargin = 1.0f / 64.0f;  // the per-voxel lines differ from these vertex values by rounding only
  const bool bad = !(fabsf(x) <= 1e30f) | !(fabsf(y) <= 1e30f) | !(fabsf(z) <= 1e30f) | dense;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const float capx = hx + 1.0f + kTileFar, capy = hy + 1.0f + kTileFar, capz = hz + 1.0f + kTileFar;
  const int xmin = -wave_max_i32(-static_cast<int>(fminf(fmaxf(floorf(x - kMargin), -kTileFar), capx)));
  const int xmax = wave_max_i32(static_cast<int>(fminf(fmaxf(floorf(x + kMargin), -kTileFar), capx)));
  const int ymin = -wave_max_i32(-static_cast<int>(fminf(fmaxf(floorf(y - kMargin), -kTileFar), capy)));
  const int ymax = wave_max_i32(static_cast<int>(fminf(fmaxf(floorf(y + kMargin), -kTileFar), capy)));
  const int zmin = -wave_max_i32(-static_cast<int>(fminf(fmaxf(floorf(z - kMargin), -kTileFar), capz)));
  const int zmax = wave_max_i32(static_cast<int>(fminf(fmaxf(floorf(z + kMargin), -kTileFar), capz)));
  const bool wrd = weird | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
  StreamBox& bx = p.bx;
  bx.interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
  const int outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
  bx.bx0 = xmin; bx.by0 = ymin; bx.za = zmin & ~3;
  bx.Lx = xmax + 2 - xmin; bx.Ly = ymax + 2 - ymin;
  const int Lz = ((zmax + 1 + 4) & ~3) - bx.za;
  bx.cpr = Lz >> 2;
  const bool fits = !wrd && (Lz <= 256) && (bx.Lx <= 4096) && (bx.Ly <= 4096) && (static_cast<int64_t>(bx.Lx) * bx.Ly * Lz <= static_cast<int64_t>(a.tile_cap));
  bx.kind = outside ? kSlabOutside : (fits ? kSlabStaged : kSlabGather);
  p.fast = (outside != 0) | fits;
}

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int OCC>
__global__ __launch_bounds__(TJ* TK, (TJ * TK) / 256 * OCC) void resample_pipe_kernel(const ResampleArgs a, int n_items) {
  constexpr int NT = TJ * TK, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_tile = smem + 16;  // (the first 16 floats: fastbrick_process' box slots)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tk = tid % TK, tj = tid / TK;

  // the block's share of the bricks: XCD x takes a contiguous range, its blocks interleave inside it
  const int nblk = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd = (nblk + 7 - xcd) >> 3;
  const int q_items = n_items / 8, r_items = n_items % 8;
  const int range0 = xcd * q_items + min(xcd, r_items), range1 = range0 + q_items + (xcd < r_items ? 1 : 0);

  const ImgArgs& g = a.img[0];  // one image, one channel (the launcher checks)
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};
  const unsigned tile_lds_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile));
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const bool has_fill = g.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)g.fill)[0] : 0.0f;

  FastFrameG f;
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.cp = nullptr; f.elastic = false; f.j_lo = 0; f.k_lo = 0;
#pragma unroll
  for (int q = 0; q < 12; q++) f.m[q] = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; r++) { f.c[r] = 0.0; f.dsc[r] = a.rsp[r] * (f.affine_first ? ratio[r] : 1.0f); }
  bool weird = false, gated = false;
  int fb = -1;  // batch element the frame belongs to
  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;

  PipePlan P{};
  bool haveP = false;
  bool lds_busy = false;  // a pass may still be reading the tile: a barrier is owed before the next DMA
  int item = range0 + slot;
  int iter = 0;
  for (;;) {
    if (!haveP) {  // ---- (re)start the pipeline: nothing in flight ----
      if (item >= range1) break;
      pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(item), P);
      item += per_xcd;
      if (P.b != fb) {
        fb = P.b;
        gated = a.passthrough != nullptr && a.passthrough[fb] != 0;
        const float* m = a.mapping + (a.mapping_batched ? fb * 12 : 0);
        weird = false;
#pragma unroll
        for (int q = 0; q < 12; q++) {
          const float mv = m[q];
          weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;
          f.m[q] = mv * ratio[q >> 2];
        }
        if constexpr (ELASTIC_POSSIBLE) {
          f.elastic = !(a.cp_skip != nullptr && a.cp_skip[fb] != 0);
          f.cp = f.elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(fb) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
        }
      }
      if (gated) {  // gated-out element: the lean brick's bit-exact copy
        fastbrick_process_cold<ELASTIC_POSSIBLE, TI, TJ, TK>(P.b, P.it, P.jt, P.kt, 1);
        continue;
      }
      pipe_brick_frame(f, P.j_lo, P.k_lo);
      pipe_box(a, f, P.i_begin, P.i_count, P.nv, P.nw, lane, weird, P);
      if (!P.fast) {
        fastbrick_process_cold<ELASTIC_POSSIBLE, TI, TJ, TK>(P.b, P.it, P.jt, P.kt, lds_busy ? 0 : 1);
        lds_busy = true;
        continue;
      }
      if (P.bx.kind == kSlabStaged) {
        if (lds_busy) { __syncthreads(); lds_busy = false; }
        if (!(a.ablate & 1))
          stream_stage<NW>(s_tile, static_cast<const float*>(g.in) + static_cast<int64_t>(P.b) * n_in, P.bx, a.I, a.J, a.K, wave, lane, sl);
      }
      haveP = true;
    }

    // ---- look ahead: the next brick's box while this brick's DMA is in flight ----
    const bool tracing = a.trace != nullptr && blockIdx.x < kTraceBlocks && tid == 0 && iter < kTraceIters;
    unsigned long long* tr = a.trace + (static_cast<size_t>(blockIdx.x) * kTraceIters + iter) * kTraceStamps;
    if (tracing) { tr[0] = __builtin_readcyclecounter(); tr[6] = __builtin_amdgcn_s_memrealtime(); }
    iter++;
    PipePlan N{};
    bool haveN = false;
    if (item < range1 && !(a.ablate & 8)) {
      pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(item), N);
      if (N.b == fb) {
        FastFrameG fn = f;
        pipe_brick_frame(fn, N.j_lo, N.k_lo);
        pipe_box(a, fn, N.i_begin, N.i_count, N.nv, N.nw, lane, weird, N);
        if (N.fast) { haveN = true; item += per_xcd; }
      }
    }

    // ---- this brick ----
    if (tracing) tr[1] = __builtin_readcyclecounter();
    {
      const bool col_active = (tj < P.nv) & (tk < P.nw);
      const int jv = min(tj, P.nv - 1), kw = min(tk, P.nw - 1);
      const int col_off = (P.j_lo + jv) * a.Ko + (P.k_lo + kw);
      const unsigned urow = static_cast<unsigned>(col_off) * 4u;
      char* out_chan = static_cast<char*>(g.out) + static_cast<int64_t>(P.b) * n_out * 4;
      const int u0 = P.i_begin, u1 = P.i_begin + P.i_count;
      if (P.bx.kind == kSlabOutside) {
        if (col_active)
          for (int t = u0; t < u1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = fillv;
      } else {
        const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
        float col3[3];
#pragma unroll
        for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
        Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
        if constexpr (ELASTIC_POSSIBLE) {
          if (f.elastic) {
            lj = lerp_index(P.j_lo + jv, a.nj, a.Jo, a.scale_j);
            lk = lerp_index(P.k_lo + kw, a.nk, a.Ko, a.scale_k);
          }
        }
        ColumnPlanes planes;
        planes.cell = -2;
#pragma unroll
        for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
        float C3[3];
        {
          const double org[3] = {static_cast<double>(P.bx.bx0), static_cast<double>(P.bx.by0), static_cast<double>(P.bx.za)};
#pragma unroll
          for (int r = 0; r < 3; r++) C3[r] = static_cast<float>(static_cast<double>(f.m[4 * r]) * u0 + f.c[r] - org[r]);
        }
        float A3[3], B3[3];
        int run0 = u0;
        int run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
        FastAddr ta;
        ta.sYb = P.bx.cpr * 16; ta.sXb = P.bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
        ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
        ta.base_f = static_cast<float>(tile_lds_addr);
        const float ox = static_cast<float>(P.bx.bx0), oy = static_cast<float>(P.bx.by0), oz = static_cast<float>(P.bx.za);
        if (tracing) tr[2] = __builtin_readcyclecounter();
        tile_dma_wait();
        __syncthreads();
        if (tracing) tr[3] = __builtin_readcyclecounter();
        lds_busy = true;
        if (a.trace != nullptr && (a.ablate & 16)) {  // profiling only: the stamped loop (affine interior bricks are what it is read for)
          unsigned long long* st = a.trace + static_cast<size_t>(kTraceBlocks) * kTraceIters * kTraceStamps +
                                   (static_cast<size_t>(min(static_cast<int>(blockIdx.x), kTraceBlocks - 1)) * kTraceIters + min(iter - 1, kTraceIters - 1)) * 16;
          if (col_active)
            fast_sample_run_traced(run1 - run0, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b,
                                   st, tracing);
        } else if (col_active && !(a.ablate & 2)) {
          for (;;) {
            fast_sample_line<4>(run1 - run0, A3, B3, ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b, has_fill & !P.bx.interior, ox,
                                oy, oz, hx, hy, hz, fillv);
            run0 = run1;
            if (run0 >= u1) break;
            run1 = fast_column_line(f, lj, lk, planes, run0, u1, u0, C3, col3, lane, A3, B3);
          }
        }
      }
    }

    // ---- hand over ----
    if (tracing) tr[4] = __builtin_readcyclecounter();
    if (haveN) {
      P = N;
      pipe_brick_frame(f, P.j_lo, P.k_lo);
      if (P.bx.kind == kSlabStaged) {
        if (lds_busy) { __syncthreads(); lds_busy = false; }
        if (!(a.ablate & 1))
          stream_stage<NW>(s_tile, static_cast<const float*>(g.in) + static_cast<int64_t>(P.b) * n_in, P.bx, a.I, a.J, a.K, wave, lane, sl);
      }
    } else {
      haveP = false;
    }
    if (tracing) tr[5] = __builtin_readcyclecounter();
  }
}


// =====================================================================================================================
// The pipelined brick kernel, wide blocks (TIO_FAST_KERNEL=pipe8 | pipe16): the shader-clock trace of the kernel above
// (profiles/r02_resample_sq.md section 4) shows a lone wave per SIMD running EVERYTHING at ~9 clocks per instruction
// — the dependent-issue latency of this chip — so with LDS capping the resident bricks at three, the lever left is
// waves per brick.  Here NPW waves share each group of 64 columns (the planes of the brick split between them), the
// next brick's box is planned by ALL waves together (one vertex per lane of a few lanes per wave, extremes through
// LDS atomics into one of two alternating slots) between the two barriers of the current brick, and its DMA is issued
// by all waves right after the second one.
// =====================================================================================================================
enum : int { kPlanInts = 16, kPlanBase = 16, kPipeTile = kPlanBase + 2 * kPlanInts };  // LDS floats ahead of the tile

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

// the box from a plan slot's six extremes (+ the non-finite flag)
__device__ __forceinline__ void pipe_box_from_slot(const ResampleArgs& a, const int* slot, bool weird, PipePlan& p) {
  const int xmin = -__builtin_amdgcn_readfirstlane(slot[0]), xmax = __builtin_amdgcn_readfirstlane(slot[1]);
  const int ymin = -__builtin_amdgcn_readfirstlane(slot[2]), ymax = __builtin_amdgcn_readfirstlane(slot[3]);
  const int zmin = -__builtin_amdgcn_readfirstlane(slot[4]), zmax = __builtin_amdgcn_readfirstlane(slot[5]);
  const bool wrd = weird | (__builtin_amdgcn_readfirstlane(slot[6]) != 0);
  StreamBox& bx = p.bx;
  bx.interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
  const int outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
  bx.bx0 = xmin; bx.by0 = ymin; bx.za = zmin & ~3;
  bx.Lx = xmax + 2 - xmin; bx.Ly = ymax + 2 - ymin;
  const int Lz = ((zmax + 1 + 4) & ~3) - bx.za;
  bx.cpr = Lz >> 2;
  const bool fits = !wrd && (Lz <= 256) && (bx.Lx <= 4096) && (bx.Ly <= 4096) && (static_cast<int64_t>(bx.Lx) * bx.Ly * Lz <= static_cast<int64_t>(a.tile_cap));
  bx.kind = outside ? kSlabOutside : (fits ? kSlabStaged : kSlabGather);
  p.fast = (outside != 0) | fits;
}

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int NPW, int WPE>
__global__ __launch_bounds__(TJ* TK* NPW, WPE) void resample_pipew_kernel(const ResampleArgs a, int n_items) {
  constexpr int NC = TJ * TK, NT = NC * NPW, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_plan = reinterpret_cast<int*>(smem) + kPlanBase;  // two slots of kPlanInts
  float* s_tile = smem + kPipeTile;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ctid = tid % NC;
  const int pg = __builtin_amdgcn_readfirstlane(tid / NC);  // which share of the brick's planes this wave samples
  const int tk = ctid % TK, tj = ctid / TK;
  // the vertex this thread evaluates when a brick is planned: vertex v sits in wave v % NW, lane v / NW
  const int my_vertex = lane * NW + wave;

  const int nblk = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
  const int per_xcd = (nblk + 7 - xcd) >> 3;
  const int q_items = n_items / 8, r_items = n_items % 8;
  const int range0 = xcd * q_items + min(xcd, r_items), range1 = range0 + q_items + (xcd < r_items ? 1 : 0);

  const ImgArgs& g = a.img[0];  // one image, one channel (the launcher checks)
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};
  const unsigned tile_lds_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile));
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const bool has_fill = g.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)g.fill)[0] : 0.0f;

  FastFrameG f;
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.cp = nullptr; f.elastic = false; f.j_lo = 0; f.k_lo = 0;
#pragma unroll
  for (int q = 0; q < 12; q++) f.m[q] = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; r++) { f.c[r] = 0.0; f.dsc[r] = a.rsp[r] * (f.affine_first ? ratio[r] : 1.0f); }
  bool weird = false, gated = false;
  int fb = -1;
  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;

  PipePlan P{};
  bool haveP = false;
  bool lds_busy = false;  // a pass may still be reading the tile (or a plan slot): a barrier is owed before they are written
  int par = 0;            // plan slot the NEXT look-ahead accumulates into (it is initialised)
  int item = range0 + slot_in_xcd;
  int iter = 0;
  for (;;) {
    if (!haveP) {  // ---- (re)start the pipeline: nothing in flight ----
      if (item >= range1) break;
      pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(item), P);
      item += per_xcd;
      if (P.b != fb) {
        fb = P.b;
        gated = a.passthrough != nullptr && a.passthrough[fb] != 0;
        const float* m = a.mapping + (a.mapping_batched ? fb * 12 : 0);
        weird = false;
#pragma unroll
        for (int q = 0; q < 12; q++) {
          const float mv = m[q];
          weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;
          f.m[q] = mv * ratio[q >> 2];
        }
        if constexpr (ELASTIC_POSSIBLE) {
          f.elastic = !(a.cp_skip != nullptr && a.cp_skip[fb] != 0);
          f.cp = f.elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(fb) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
        }
      }
      if (gated) {
        if (NPW == 1 || pg == 0) {}  // (the copy below is per column; every plane share does its own planes)
        const bool col_ok = (tj < P.nv) & (tk < P.nw);
        if (col_ok) {
          const int64_t off = static_cast<int64_t>(P.b) * n_out + (P.j_lo + tj) * a.Ko + (P.k_lo + tk);
          for (int t = pg; t < P.i_count; t += NPW)
            static_cast<float*>(g.out)[off + static_cast<int64_t>(P.i_begin + t) * slab] = static_cast<const float*>(g.in)[off + static_cast<int64_t>(P.i_begin + t) * slab];
        }
        continue;
      }
      pipe_brick_frame(f, P.j_lo, P.k_lo);
      // plan P on the spot, in slot `par` (initialised); the other slot is initialised for the first look-ahead
      if (lds_busy) { __syncthreads(); lds_busy = false; }
      if (tid < kPlanInts) { s_plan[par * kPlanInts + tid] = tid < 6 ? -0x40000000 : 0; s_plan[(par ^ 1) * kPlanInts + tid] = tid < 6 ? -0x40000000 : 0; }
      __syncthreads();
      if (my_vertex < (f.elastic ? 27 : 8)) {
        int r[6]; bool bad;
        pipe_vertex(a, f, my_vertex, P.i_begin, P.i_count, P.nv, P.nw, r, bad);
        int* sp = s_plan + par * kPlanInts;
#pragma unroll
        for (int q = 0; q < 6; q++) atomicMax(&sp[q], r[q]);
        if (bad) atomicOr(&sp[6], 1);
      }
      __syncthreads();
      pipe_box_from_slot(a, s_plan + par * kPlanInts, weird, P);
      par ^= 1;
      lds_busy = true;  // (the slot just read is re-initialised only after the next barrier)
      if (!P.fast) {  // several passes / non-finite geometry: per-voxel evaluation with global gathers (rare)
        const bool col_ok = (tj < P.nv) & (tk < P.nw);
        if (col_ok) {
          ImgArgs g1 = g;
          g1.in = static_cast<const float*>(g.in) + static_cast<int64_t>(P.b) * n_in;
          g1.out = static_cast<float*>(g.out) + static_cast<int64_t>(P.b) * n_out;
          g1.channels = 1;
          const int col_off = (P.j_lo + tj) * a.Ko + (P.k_lo + tk);
          for (int t = P.i_begin + pg; t < P.i_begin + P.i_count; t += NPW) {
            float x, y, z;
            fast_coord(f, static_cast<float>(t), static_cast<float>(tj), static_cast<float>(tk), x, y, z);
            gather_voxel<0>(g1, a, 0, n_in, n_out, t * slab + col_off, x, y, z, false);
          }
        }
        continue;
      }
      if (P.bx.kind == kSlabStaged) {
        __syncthreads(); lds_busy = false;
        if (tid < kPlanInts) s_plan[(par ^ 1) * kPlanInts + tid] = tid < 6 ? -0x40000000 : 0;  // P's slot, free again
        if (!(a.ablate & 1))
          stream_stage<NW>(s_tile, static_cast<const float*>(g.in) + static_cast<int64_t>(P.b) * n_in, P.bx, a.I, a.J, a.K, wave, lane, sl);
      }
      haveP = true;
    }

    const bool tracing = a.trace != nullptr && blockIdx.x < kTraceBlocks && tid == 0 && iter < kTraceIters;
    unsigned long long* tr = a.trace + (static_cast<size_t>(blockIdx.x) * kTraceIters + iter) * kTraceStamps;
    if (tracing) { tr[0] = __builtin_readcyclecounter(); tr[6] = __builtin_amdgcn_s_memrealtime(); }
    iter++;

    // ---- look ahead, part 1 (registers only, overlaps the DMA): this thread's vertex of the next brick ----
    PipePlan N{};
    bool candN = false;
    int nr[6] = {0, 0, 0, 0, 0, 0};
    bool nbad = false;
    if (item < range1 && !(a.ablate & 8)) {
      pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(item), N);
      if (N.b == fb) {
        candN = true;
        if (my_vertex < (f.elastic ? 27 : 8)) {
          FastFrameG fn = f;
          pipe_brick_frame(fn, N.j_lo, N.k_lo);
          pipe_vertex(a, fn, my_vertex, N.i_begin, N.i_count, N.nv, N.nw, nr, nbad);
        }
      }
    }
    if (tracing) tr[1] = __builtin_readcyclecounter();

    // ---- this brick ----
    {
      const bool col_active = (tj < P.nv) & (tk < P.nw);
      const int jv = min(tj, P.nv - 1), kw = min(tk, P.nw - 1);
      const int col_off = (P.j_lo + jv) * a.Ko + (P.k_lo + kw);
      const unsigned urow = static_cast<unsigned>(col_off) * 4u;
      char* out_chan = static_cast<char*>(g.out) + static_cast<int64_t>(P.b) * n_out * 4;
      const int u0 = P.i_begin, u1 = P.i_begin + P.i_count;
      const int per = (P.i_count + NPW - 1) / NPW;
      const int my0 = min(u0 + pg * per, u1), my1 = min(my0 + per, u1);
      if (P.bx.kind == kSlabOutside) {
        if (lds_busy | candN) { __syncthreads(); lds_busy = false; }  // (the plan slot's initialisation is ordered by a barrier)
        if (candN && my_vertex < (f.elastic ? 27 : 8)) {
          int* sp = s_plan + par * kPlanInts;
#pragma unroll
          for (int q = 0; q < 6; q++) atomicMax(&sp[q], nr[q]);
          if (nbad) atomicOr(&sp[6], 1);
        }
        if (col_active)
          for (int t = my0; t < my1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = fillv;
      } else {
        const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
        float col3[3];
#pragma unroll
        for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
        Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
        if constexpr (ELASTIC_POSSIBLE) {
          if (f.elastic) {
            lj = lerp_index(P.j_lo + jv, a.nj, a.Jo, a.scale_j);
            lk = lerp_index(P.k_lo + kw, a.nk, a.Ko, a.scale_k);
          }
        }
        ColumnPlanes planes;
        planes.cell = -2;
#pragma unroll
        for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
        float C3[3];
        {
          const double org[3] = {static_cast<double>(P.bx.bx0), static_cast<double>(P.bx.by0), static_cast<double>(P.bx.za)};
#pragma unroll
          for (int r = 0; r < 3; r++) C3[r] = static_cast<float>(static_cast<double>(f.m[4 * r]) * u0 + f.c[r] - org[r]);
        }
        float A3[3] = {0.f, 0.f, 0.f}, B3[3] = {0.f, 0.f, 0.f};
        int run0 = my0, run1 = my0;
        if (my0 < my1) run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
        FastAddr ta;
        ta.sYb = P.bx.cpr * 16; ta.sXb = P.bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
        ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
        ta.base_f = static_cast<float>(tile_lds_addr);
        const float ox = static_cast<float>(P.bx.bx0), oy = static_cast<float>(P.bx.by0), oz = static_cast<float>(P.bx.za);
        if (tracing) tr[2] = __builtin_readcyclecounter();
        tile_dma_wait();
        __syncthreads();
        lds_busy = true;
        if (tracing) tr[3] = __builtin_readcyclecounter();
        // look ahead, part 2: the vertex into the (initialised) plan slot
        if (candN && my_vertex < (f.elastic ? 27 : 8)) {
          int* sp = s_plan + par * kPlanInts;
#pragma unroll
          for (int q = 0; q < 6; q++) atomicMax(&sp[q], nr[q]);
          if (nbad) atomicOr(&sp[6], 1);
        }
        if (col_active && !(a.ablate & 2) && my0 < my1) {
          for (;;) {
            fast_sample_line<4>(run1 - run0, A3, B3, ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b, has_fill & !P.bx.interior, ox,
                                oy, oz, hx, hy, hz, fillv);
            run0 = run1;
            if (run0 >= my1) break;
            run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
          }
        }
      }
    }

    // ---- hand over ----
    if (tracing) tr[4] = __builtin_readcyclecounter();
    if (candN) {
      __syncthreads();  // every wave has left the tile and added its vertices
      lds_busy = false;
      pipe_box_from_slot(a, s_plan + par * kPlanInts, weird, N);
      par ^= 1;  // the next look-ahead uses the other slot; it is re-initialised here, two barriers before it is read
      if (tid < kPlanInts) s_plan[par * kPlanInts + tid] = tid < 6 ? -0x40000000 : 0;
      if (N.fast) {
        item += per_xcd;
        P = N;
        pipe_brick_frame(f, P.j_lo, P.k_lo);
        if (P.bx.kind == kSlabStaged && !(a.ablate & 1))
          stream_stage<NW>(s_tile, static_cast<const float*>(g.in) + static_cast<int64_t>(P.b) * n_in, P.bx, a.I, a.J, a.K, wave, lane, sl);
      } else {
        haveP = false;  // the restart plans it again and takes the slow road
        lds_busy = true;
      }
    } else {
      haveP = false;
    }
    if (tracing) tr[5] = __builtin_readcyclecounter();
  }
}


// =====================================================================================================================
// Planned bricks (TIO_FAST_KERNEL=desc4 | desc8 | desc16).  What the shader-clock traces of the kernels above say
// (profiles/r02_resample_sq.md section 4): a wave that is alone on its SIMD runs EVERYTHING at ~9 clocks per
// instruction, so the per-brick chain kernel arguments -> mapping -> vertices -> reductions -> barrier -> DMA ->
// sampling is long whatever its instruction count, and making the blocks wider only multiplies the redundant part of
// it (pipe8: 1.0 ms).  So the planning leaves the sampling kernel altogether:
//   * plan_bricks_kernel — one THREAD per brick: the box from the <= 27 vertices, the float64-formed line constants,
//     the kind of the brick, 16 dwords per brick; one thread per batch element: the scaled mapping.  ~0.4 M threads
//     of a few hundred instructions: microseconds.
//   * resample_desc_kernel — one block per brick, NPW waves per 64 columns: one s_load_dwordx16 for the brick, one
//     for the mapping, the DMA, the per-column line, the sampling.  No decode, no reductions, no LDS besides the
//     tile, two barriers.
// =====================================================================================================================
enum : int { kDescInts = 16, kDescStaged = 0, kDescOutside = 1, kDescSlow = 2, kDescGated = 3 };
// desc: [0] kind | interior << 8   [1] bx0 [2] by0 [3] za [4] Lx [5] Ly [6] cpr   [7..9] C3 (float bits)
//       [10] b [11] i_begin [12] j_lo [13] k_lo [14] elastic [15] -
// batch frame (16 floats per element, ahead of the bricks): [0..11] mapping rows scaled by the axis ratios

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK>
__global__ __launch_bounds__(256) void plan_bricks_kernel(const ResampleArgs a, int* __restrict__ plan, int n_items) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};
  if (t < a.B) {
    const float* m = a.mapping + (a.mapping_batched ? t * 12 : 0);
    float* fr = reinterpret_cast<float*>(plan) + t * 16;
    for (int q = 0; q < 12; q++) fr[q] = m[q] * ratio[q >> 2];
    for (int q = 12; q < 16; q++) fr[q] = 0.0f;
  }
  if (t >= n_items) return;
  PipePlan p;
  pipe_decode<TI, TJ, TK>(a, static_cast<unsigned>(t), p);
  int* d = plan + a.B * 16 + t * kDescInts;
  d[10] = p.b; d[11] = p.i_begin; d[12] = p.j_lo; d[13] = p.k_lo; d[15] = 0;
  if (a.passthrough != nullptr && a.passthrough[p.b] != 0) {
    d[0] = kDescGated;
    for (int q = 1; q < 10; q++) d[q] = 0;
    d[14] = 0;
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
      weird |= (__float_as_uint(mv) & 0x7FFFFFFFu) > 0x7149F2CAu;
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
  int ext[6] = {-0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000, -0x40000000};
  bool any_bad = false;
  const int n_vert = f.elastic ? 27 : 8;
  for (int v = 0; v < n_vert; v++) {
    int r[6]; bool bad;
    pipe_vertex(a, f, v, p.i_begin, p.i_count, p.nv, p.nw, r, bad);
#pragma unroll
    for (int q = 0; q < 6; q++) ext[q] = max(ext[q], r[q]);
    any_bad |= bad;
  }
  const int xmin = -ext[0], xmax = ext[1], ymin = -ext[2], ymax = ext[3], zmin = -ext[4], zmax = ext[5];
  const bool wrd = weird | any_bad;
  const int interior = (xmin >= 0) & (xmax + 1 <= a.I - 1) & (ymin >= 0) & (ymax + 1 <= a.J - 1) & (zmin >= 0) & (zmax + 1 <= a.K - 1) & !wrd;
  const int outside = ((xmax + 1 < 0) | (xmin > a.I - 1) | (ymax + 1 < 0) | (ymin > a.J - 1) | (zmax + 1 < 0) | (zmin > a.K - 1)) & !wrd;
  const int za = zmin & ~3, Lx = xmax + 2 - xmin, Ly = ymax + 2 - ymin, Lz = ((zmax + 1 + 4) & ~3) - za;
  const bool fits = !wrd && (Lz <= 256) && (Lx <= 4096) && (Ly <= 4096) && (static_cast<int64_t>(Lx) * Ly * Lz <= static_cast<int64_t>(a.tile_cap));
  d[0] = (outside ? kDescOutside : (fits ? kDescStaged : kDescSlow)) | (interior << 8);
  d[1] = xmin; d[2] = ymin; d[3] = za; d[4] = Lx; d[5] = Ly; d[6] = Lz >> 2;
  const double org[3] = {static_cast<double>(xmin), static_cast<double>(ymin), static_cast<double>(za)};
#pragma unroll
  for (int r = 0; r < 3; r++) d[7 + r] = __float_as_int(static_cast<float>(static_cast<double>(f.m[4 * r]) * p.i_begin + f.c[r] - org[r]));
  d[14] = f.elastic ? 1 : 0;
}

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int NPW, int WPE, int GMAX>
__global__ __launch_bounds__(TJ* TK* NPW, WPE) void resample_desc_kernel(const ResampleArgs a, const int* __restrict__ plan) {
  constexpr int NC = TJ * TK, NT = NC * NPW, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_tile = smem;
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  typedef __attribute__((address_space(4))) const float* const_float_ptr;

  // experiment (TIO_DESC_STAGGER, carried in bits 8.. of `ablate`): the first resident generation of blocks starts in
  // three different phases, so that fetch, sampling and store phases of different blocks meet instead of coinciding
  if ((a.ablate >> 8) != 0 && blockIdx.x < 768u * 2u) {
    const int steps = static_cast<int>((blockIdx.x >> 3) % 3u) * (a.ablate >> 8);
    for (int q = 0; q < steps; q++) __builtin_amdgcn_s_sleep(16);
  }
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
  const int ctid = tid % NC;
  const int pg = __builtin_amdgcn_readfirstlane(tid / NC);
  const int tk = ctid % TK, tj = ctid / TK;
  const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
  const bool col_active = (tj < nv) & (tk < nw);
  const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
  const ImgArgs& g = a.img[0];
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
  const unsigned urow = static_cast<unsigned>(col_off) * 4u;
  char* out_chan = static_cast<char*>(g.out) + static_cast<int64_t>(b) * n_out * 4;
  const float* in_chan = static_cast<const float*>(g.in) + static_cast<int64_t>(b) * n_in;
  const int u0 = i_begin, u1 = i_begin + i_count;
  const int per = (i_count + NPW - 1) / NPW;
  const int my0 = min(u0 + pg * per, u1), my1 = min(my0 + per, u1);
  const bool has_fill = g.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)g.fill)[0] : 0.0f;

  if (kind == kDescGated) {
    if (col_active)
      for (int t = my0; t < my1; t++)
        *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = in_chan[static_cast<int64_t>(t) * slab + col_off];
    return;
  }
  if (kind == kDescOutside) {
    if (col_active)
      for (int t = my0; t < my1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = fillv;
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
    for (int e = 0; e < 3; e++) f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f);
  }
#pragma unroll
  for (int r = 0; r < 3; r++) f.c[r] = 0.0;

  if (kind == kDescSlow) {  // several passes / non-finite geometry: per-voxel evaluation with global gathers (rare)
    pipe_brick_frame(f, j_lo, k_lo);
    if (col_active) {
      ImgArgs g1 = g;
      g1.in = in_chan; g1.out = out_chan; g1.channels = 1;
      for (int t = my0; t < my1; t++) {
        float x, y, z;
        fast_coord(f, static_cast<float>(t), static_cast<float>(jv), static_cast<float>(kw), x, y, z);
        gather_voxel<0>(g1, a, 0, n_in, n_out, t * slab + col_off, x, y, z, false);
      }
    }
    return;
  }

  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;
  if (!(a.ablate & 1)) stream_stage<NW>(s_tile, in_chan, bx, a.I, a.J, a.K, wave, lane, sl);

  // while the brick is on its way: this column's line
  const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
  float col3[3];
#pragma unroll
  for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
  Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
  if constexpr (ELASTIC_POSSIBLE) {
    if (elastic) {
      lj = lerp_index(j_lo + jv, a.nj, a.Jo, a.scale_j);
      lk = lerp_index(k_lo + kw, a.nk, a.Ko, a.scale_k);
    }
  }
  ColumnPlanes planes;
  planes.cell = -2;
#pragma unroll
  for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
  float A3[3] = {0.f, 0.f, 0.f}, B3[3] = {0.f, 0.f, 0.f};
  int run0 = my0, run1 = my0;
  if (my0 < my1) run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
  FastAddr ta;
  ta.sYb = bx.cpr * 16; ta.sXb = bx.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
  ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
  ta.base_f = static_cast<float>(static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)s_tile)));
  const float ox = static_cast<float>(bx.bx0), oy = static_cast<float>(bx.by0), oz = static_cast<float>(bx.za);
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  tile_dma_wait();
  __syncthreads();
  if (a.ablate & 4) {  // profiling only: sample, never store
    if (col_active && my0 < my1)
      fast_sample_run<false, 4, true>(run1 - run0, A3[0], A3[1], A3[2], B3[0], B3[1], B3[2], ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b,
                                      0.f, 0.f, 0.f, hx, hy, hz, fillv);
  } else if (col_active && !(a.ablate & 2) && my0 < my1) {
    for (;;) {
      fast_sample_line<GMAX>(run1 - run0, A3, B3, ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b, has_fill & !bx.interior, ox, oy, oz,
                             hx, hy, hz, fillv);
      run0 = run1;
      if (run0 >= my1) break;
      run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
    }
  }
}


// =====================================================================================================================
// Planned bricks through a two-buffer LDS ring (TIO_FAST_KERNEL=ring4 | ring8).  One-brick blocks of the kernel above
// synchronise on the shared HBM: all resident blocks fetch together, then all sample together, and the two phases add
// (profiles/r02_resample_sq.md section 4: 0.05 set-up + 0.13 DMA + 0.19 sampling = 0.37 ms).  Here a persistent
// block keeps brick n + 1's DMA in flight while it samples brick n, so every CU reads HBM and samples all the time;
// the descriptors make the look-ahead free (one scalar load per brick, prefetched an iteration early).
// =====================================================================================================================
struct RingDesc {
  int kind, interior, bx0, by0, za, Lx, Ly, cpr;
  float c3x, c3y, c3z;
  int b, i_begin, j_lo, k_lo, elastic;
};

__device__ __forceinline__ RingDesc ring_load_desc(const int* plan, int n_batch, int brick) {
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  const_int_ptr d = (const_int_ptr)(plan + n_batch * 16) + static_cast<size_t>(brick) * kDescInts;
  RingDesc r;
  const int kw = d[0];
  r.kind = kw & 0xFF; r.interior = kw >> 8;
  r.bx0 = d[1]; r.by0 = d[2]; r.za = d[3]; r.Lx = d[4]; r.Ly = d[5]; r.cpr = d[6];
  r.c3x = __int_as_float(d[7]); r.c3y = __int_as_float(d[8]); r.c3z = __int_as_float(d[9]);
  r.b = d[10]; r.i_begin = d[11]; r.j_lo = d[12]; r.k_lo = d[13]; r.elastic = d[14];
  return r;
}

template <bool ELASTIC_POSSIBLE, int TI, int TJ, int TK, int NPW, int WPE, int GMAX>
__global__ __launch_bounds__(TJ* TK* NPW, WPE) void resample_ring_kernel(const ResampleArgs a, const int* __restrict__ plan, int n_items) {
  constexpr int NC = TJ * TK, NT = NC * NPW, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const int cap = a.tile_cap;  // floats per buffer (two of them)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ctid = tid % NC;
  const int pg = __builtin_amdgcn_readfirstlane(tid / NC);
  const int tk = ctid % TK, tj = ctid / TK;

  const int nblk = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
  const int per_xcd = (nblk + 7 - xcd) >> 3;
  const int q_items = n_items / 8, r_items = n_items % 8;
  const int range0 = xcd * q_items + min(xcd, r_items), range1 = range0 + q_items + (xcd < r_items ? 1 : 0);

  const ImgArgs& g = a.img[0];
  const int64_t n_in = static_cast<int64_t>(a.I) * a.J * a.K;
  const int64_t n_out = static_cast<int64_t>(a.Io) * a.Jo * a.Ko;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const float hx = a.size_m1[0], hy = a.size_m1[1], hz = a.size_m1[2];
  const bool has_fill = g.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)g.fill)[0] : 0.0f;
  const unsigned buf0_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)smem));
  const float ratio[3] = {a.half_h[0] / a.dh[0], a.half_h[1] / a.dh[1], a.half_h[2] / a.dh[2]};

  StageLanes sl;
  sl.cpr = -1; sl.rpi = 1; sl.row_l = 0; sl.gz_rel = 0; sl.goff = 0; sl.lane_ok = false;

  FastFrameG f;
  f.affine_first = a.affine_first != 0;
  f.ni = a.ni; f.nj = a.nj; f.nk = a.nk; f.sci = a.scale_i; f.scj = a.scale_j; f.sck = a.scale_k;
  f.cp = nullptr; f.elastic = false; f.j_lo = 0; f.k_lo = 0;
#pragma unroll
  for (int q = 0; q < 12; q++) f.m[q] = 0.0f;
  int fb = -1;
  int item = range0 + slot_in_xcd;
  if (item >= range1) return;
  RingDesc cur = ring_load_desc(plan, a.B, item);
  RingDesc nxt = item + per_xcd < range1 ? ring_load_desc(plan, a.B, item + per_xcd) : cur;  // descriptors run two bricks ahead
  auto box_of = [](const RingDesc& d) {
    StreamBox bx;
    bx.kind = d.kind; bx.interior = d.interior; bx.bx0 = d.bx0; bx.by0 = d.by0; bx.za = d.za; bx.Lx = d.Lx; bx.Ly = d.Ly; bx.cpr = d.cpr;
    return bx;
  };
  if (cur.kind == kDescStaged && !(a.ablate & 1))
    stream_stage<NW>(smem, static_cast<const float*>(g.in) + static_cast<int64_t>(cur.b) * n_in, box_of(cur), a.I, a.J, a.K, wave, lane, sl);
  tile_dma_wait();
  __syncthreads();

  for (int n = 0;; n++) {
    // ---- the next brick: descriptor, DMA into the other buffer ----
    const int next_item = item + per_xcd;
    const bool have_next = next_item < range1;
    if (have_next && nxt.kind == kDescStaged && !(a.ablate & 1))
      stream_stage<NW>(smem + ((n + 1) & 1) * cap, static_cast<const float*>(g.in) + static_cast<int64_t>(nxt.b) * n_in, box_of(nxt), a.I, a.J, a.K, wave,
                       lane, sl);
    // the descriptor after that one: scalar loads in flight while this brick is sampled
    const RingDesc nn = next_item + per_xcd < range1 ? ring_load_desc(plan, a.B, next_item + per_xcd) : nxt;

    // ---- this brick ----
    {
      const int i_count = min(TI, a.Io - cur.i_begin), nv = min(TJ, a.Jo - cur.j_lo), nw = min(TK, a.Ko - cur.k_lo);
      const bool col_active = (tj < nv) & (tk < nw);
      const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
      const int col_off = (cur.j_lo + jv) * a.Ko + (cur.k_lo + kw);
      const unsigned urow = static_cast<unsigned>(col_off) * 4u;
      char* out_chan = static_cast<char*>(g.out) + static_cast<int64_t>(cur.b) * n_out * 4;
      const float* in_chan = static_cast<const float*>(g.in) + static_cast<int64_t>(cur.b) * n_in;
      const int u0 = cur.i_begin, u1 = cur.i_begin + i_count;
      const int per = (i_count + NPW - 1) / NPW;
      const int my0 = min(u0 + pg * per, u1), my1 = min(my0 + per, u1);
      if (cur.kind == kDescGated) {
        if (col_active)
          for (int t = my0; t < my1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = in_chan[static_cast<int64_t>(t) * slab + col_off];
      } else if (cur.kind == kDescOutside) {
        if (col_active)
          for (int t = my0; t < my1; t++) *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = fillv;
      } else {
        if (cur.b != fb) {  // the scaled mapping of the batch element: scalar loads, once per element
          const_float_ptr fm = (const_float_ptr)(plan) + cur.b * 16;
#pragma unroll
          for (int q = 0; q < 12; q++) f.m[q] = fm[q];
          fb = cur.b;
        }
        f.elastic = ELASTIC_POSSIBLE && cur.elastic != 0;
        f.cp = f.elastic ? a.cp + (a.cp_batched ? static_cast<int64_t>(cur.b) * (a.ni * a.nj * a.nk * 3) : 0) : nullptr;
        f.j_lo = cur.j_lo; f.k_lo = cur.k_lo;
#pragma unroll
        for (int e = 0; e < 3; e++) { f.dsc[e] = a.rsp[e] * (f.affine_first ? ratio[e] : 1.0f); f.c[e] = 0.0; }
        if (cur.kind == kDescSlow) {
          pipe_brick_frame(f, cur.j_lo, cur.k_lo);
          if (col_active) {
            ImgArgs g1 = g;
            g1.in = in_chan; g1.out = out_chan; g1.channels = 1;
            for (int t = my0; t < my1; t++) {
              float x, y, z;
              fast_coord(f, static_cast<float>(t), static_cast<float>(jv), static_cast<float>(kw), x, y, z);
              gather_voxel<0>(g1, a, 0, n_in, n_out, t * slab + col_off, x, y, z, false);
            }
          }
        } else {
          const float fv = static_cast<float>(jv), fw = static_cast<float>(kw);
          float col3[3];
#pragma unroll
          for (int r = 0; r < 3; r++) col3[r] = __builtin_fmaf(f.m[4 * r + 1], fv, f.m[4 * r + 2] * fw);
          Lerp1D lj{0, 0, 1.0f, 0.0f}, lk{0, 0, 1.0f, 0.0f};
          if constexpr (ELASTIC_POSSIBLE) {
            if (f.elastic) {
              lj = lerp_index(cur.j_lo + jv, a.nj, a.Jo, a.scale_j);
              lk = lerp_index(cur.k_lo + kw, a.nk, a.Ko, a.scale_k);
            }
          }
          ColumnPlanes planes;
          planes.cell = -2;
#pragma unroll
          for (int e = 0; e < 3; e++) { planes.P0[e] = 0.0f; planes.P1[e] = 0.0f; }
          const float C3[3] = {cur.c3x, cur.c3y, cur.c3z};
          float A3[3] = {0.f, 0.f, 0.f}, B3[3] = {0.f, 0.f, 0.f};
          int run0 = my0, run1 = my0;
          if (my0 < my1) run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
          FastAddr ta;
          ta.sYb = cur.cpr * 16; ta.sXb = cur.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
          ta.sYf = static_cast<float>(ta.sYb); ta.sXf = static_cast<float>(ta.sXb);
          ta.base_f = static_cast<float>(buf0_addr + static_cast<unsigned>((n & 1) * cap) * 4u);
          const float ox = static_cast<float>(cur.bx0), oy = static_cast<float>(cur.by0), oz = static_cast<float>(cur.za);
          if (col_active && !(a.ablate & 2) && my0 < my1) {
            for (;;) {
              fast_sample_line<GMAX>(run1 - run0, A3, B3, ta, out_chan + static_cast<int64_t>(run0) * slab_b, urow, slab_b, has_fill & !cur.interior, ox,
                                     oy, oz, hx, hy, hz, fillv);
              run0 = run1;
              if (run0 >= my1) break;
              run1 = fast_column_line(f, lj, lk, planes, run0, my1, u0, C3, col3, lane, A3, B3);
            }
          }
        }
      }
    }
    if (!have_next) break;
    // brick n + 1 has landed and everybody has left buffer n & 1 (the one brick n + 2 goes into)
    tile_dma_wait();
    __syncthreads();
    cur = nxt;
    nxt = nn;
    item = next_item;
  }
}

}  // namespace tio
