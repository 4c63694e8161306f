// resample_fast.hpp — the TIO_PRECISION_FAST resampler: float32 trilinear images only.
//
// Why a second kernel.  The exact brick kernel (resample_tile.hpp) spends ~130 vector
// instructions per voxel on reproducing the reference's float32 operation sequence bit for
// bit and saturates the vector ALU (profiles/r02_resample_sq.md).  Intensities only owe the
// reference 1e-4 relative (BASELINE.json north_star), so launches made only of float32
// trilinear images may run this kernel; label maps never do.  Same interpolant and fill rule
// (spatial.py:1504-1648, 1695-1731), ~32 vector instructions per voxel.
//
// Why it is a persistent, software-pipelined kernel.  With independent bricks the three phases
// of a block — box + constants (vector ALU), staging (memory), sampling (LDS + vector ALU) —
// measured 135 + 124 + 172 us for the bench launch and simply ADD UP (0.45 ms): the resident
// blocks start together and stay in lock-step, so nothing overlaps.  Here one block owns a
// column of TJ x TK output columns and walks it along i in slabs of <= SMAX planes:
//   * wave NWC (the producer) runs ahead: for slab s + NB - 1 it evaluates the input bounding
//     box from the <= 27 vertices of (slab x control cells) — the coordinate map is
//     multilinear on each such sub-box, so its extremes sit on vertices; one lane per vertex, a
//     DPP wave reduction — writes a descriptor to LDS and issues the LDS-DMA of the box into a
//     ring of NB buffers (addressing on the scalar unit: one wave instruction per group of rows
//     of one x-plane, lane constants computed once, no per-chunk division);
//   * waves 0 .. NWC-1 (the consumers) sample slab s meanwhile: coordinates are a line in the
//     plane index, x(t) = A + t B, in a box-relative frame (|x| < ~40, so one ulp is 4e-6 voxel;
//     the slab constant is formed in float64 by the producer) — 3 fma per voxel, no coordinate
//     arrays; 8 taps as four ds_read2_b32, seven fma lerps; the in-bounds weight mask of the
//     fill rule in its separable form (boundary slabs with a fill value only);
//   * ONE s_barrier per slab, no vector-memory wait in the consumers (their only VMEM
//     operations are the output stores), control points served from an LDS copy.
// Work items (batch element, image channel, tile column) are dealt to the persistent blocks in
// XCD-contiguous ranges, neighbouring tile columns to CUs of the same XCD at the same time, so
// the halo they share is served by that XCD's L2.
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

struct FastFrame {
  float m[12];            // voxel mapping (rows scaled by the normalisation ratio of the axis)
  double c[3];            // mapping applied to (0, j_lo, k_lo)
  int j_lo, k_lo;
  bool elastic, affine_first;
  fast_lds_ptr cp;        // control points of this batch element (LDS copy)
  int ni, nj, nk;
  float sci, scj, sck;    // control-grid lerp scales
  float dsc[3];           // displacement scale: 1 / spacing (times the axis ratio when affine_first)
};

// displacement (already in voxels) at the volume position (i, j, k), any of them fractional
__device__ __forceinline__ void fast_displacement(const FastFrame& f, float pi, float pj, float pk, float (&d)[3]) {
  int i0, i1, j0, j1, k0, k1;
  float li, lj, lk;
  fast_axis(f.sci * pi, f.ni, i0, i1, li);
  fast_axis(f.scj * pj, f.nj, j0, j1, lj);
  fast_axis(f.sck * pk, f.nk, k0, k1, lk);
  const int s_i = f.nj * f.nk * 3, s_j = f.nk * 3;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    fast_lds_ptr p = f.cp + e;
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
__device__ __forceinline__ void fast_coord(const FastFrame& f, float u, float v, float w, float& x, float& y, float& z) {
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
                                                char* out_t, unsigned urow, int64_t slab_b, float ox, float oy, float oz, float hx,
                                                float hy, float hz, float fillv) {
  const float bxg = static_cast<float>(G) * bxs, byg = static_cast<float>(G) * bys, bzg = static_cast<float>(G) * bzs;
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
#pragma unroll
    for (int q = 0; q < G; q++) {
      float val = fast_finish(ts[q]);
      if constexpr (MASKED) val = (fast_mask(ts[q], x0s[q] + ox, y0s[q] + oy, z0s[q] + oz, hx, hy, hz) > 0.5f) ? val : fillv;
      if (tg + q < n) {
        if (!NOSTORE || val == 1.2345e37f) *reinterpret_cast<float*>(out_t + urow) = val;
      }
      out_t += slab_b;
      asm volatile("" : "+s"(out_t));  // one running pointer (2 scalar adds per plane), not G precomputed ones
    }
    ax += bxg; ay += byg; az += bzg;
    __builtin_amdgcn_sched_barrier(0);
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

}  // namespace tio
