// resample_lean_persist.hpp — round 5: the exact-coordinate brick kernel as a PERSISTENT, DOUBLE-BUFFERED block per CU.
//
// Every brick structure of rounds 2 - 5 (one block per brick, three blocks per CU) measures launch time = memory time +
// arithmetic time: a block fetches its box, waits, then samples, and the three blocks of a CU (and, through the shared
// HBM, the blocks of the whole chip) fall into step — everybody fetches while nobody computes and the other way round.
// For the exact-coordinate kernel both parts are about the same size (0.19 - 0.27 ms of traffic, ~0.2 ms of vector
// issue at the chip's peak rate), so the launch (0.43 ms) is ~twice what either needs.  The overlap has to be BUILT:
//   * one block of 1 024 threads (16 waves, 4 per SIMD) per CU walks its share of the planned bricks;
//   * THREE tiles in LDS: while the block samples brick n from one, the box of brick n + 1 has landed or is landing in the
//     second and the LDS-DMA of brick n + 2 — issued at the TOP of the iteration, from a descriptor loaded an iteration
//     earlier — fills the third: a box has two bricks' time to land, and a CU always has a box in flight (the first
//     version, two tiles, measured 0.665 ms against 0.424: one box in flight per CU, every CU requesting at the same
//     moment — the chip's memory system saw bursts and idled in between);
//   * ONE barrier per brick; each wave waits until only the DMA instructions of the box BEHIND the next one are outstanding
//     (`s_waitcnt vmcnt(n)`, n = what it issued at the top of this iteration) BEFORE it issues this brick's stores — the
//     four values wait in registers — so the wait never includes this brick's stores (round 3's two-bricks-per-block
//     experiment waited behind them), only the previous brick's, which are a whole brick old;
//   * sixteen waves share a brick: four planes per thread (one group of the sampling loop), a sixteenth of the DMA
//     instructions per wave — the per-brick chain of a wave is ~450 instructions instead of ~1 600.
// Rounds 2 / 3 measured persistent structures and dropped them for their per-iteration overhead (in-kernel box search,
// helper waves, two barriers).  What is different here: the planner (a descriptor per brick, nothing to search), the
// constant-count wait, one barrier, and a kernel whose arithmetic is large enough to be worth hiding.
//
// MEASURED (round 5, 8 x 256^3 affine launch, profiles/r05_resample.md): correct on the first run (every parity case
// of tests/native/resample_bench bit for bit / inside the per-voxel bar) and SLOWER — two tiles 0.665 ms, three tiles
// 0.705 ms against 0.424 - 0.430 for one block per brick; the exact and the tight instantiation take the same time, so it
// is neither the interpolation's arithmetic nor (three tiles: a box has two bricks' time) the box's latency.  What its
// assembly shows instead: every one of the 16 waves runs the per-brick UNIFORM work — descriptor rotation, the DMA
// stepper's set-up, the walk's bookkeeping, ~350 - 400 scalar instructions and 50 spilled scalar registers per brick —
// and a CU has ONE scalar unit: 16 x 400 scalar instructions per brick is more than the ~4 000 cycles a brick may cost.
// OFF by default (TIO_LEAN_PERSIST=1 enables it; the harness paths "tight-pdb" / "lean-exact-pdb" keep it honest).  What
// would have to change for it to pay: the per-brick scalar work done ONCE (by one wave, handed over through LDS) or
// precomputed by the planner, and DMA issued by one wave group per brick in turn.
//
// Same arithmetic, same results as resample_lean_exact_kernel (resample_lean_exact.hpp): the reference's coordinate chain
// per voxel, ATen's interpolation order (EXACT_LERP) or fused lerps (TIO_PRECISION_TIGHT).  Affine launches (no control
// points) in this first version.
#pragma once

namespace tio {

struct BrickDesc {
  int kind_w, bx0, by0, za, Lx, Ly, cpr, b, i_begin, j_lo, k_lo;
};

__device__ __forceinline__ BrickDesc load_brick_desc(const int* plan_bricks, unsigned brick) {
  typedef __attribute__((address_space(4))) const int* const_int_ptr;
  const_int_ptr d = (const_int_ptr)(plan_bricks) + static_cast<size_t>(brick) * kDescInts;
  BrickDesc r;
  r.kind_w = d[0]; r.bx0 = d[1]; r.by0 = d[2]; r.za = d[3]; r.Lx = d[4]; r.Ly = d[5]; r.cpr = d[6];
  r.b = d[10]; r.i_begin = d[11]; r.j_lo = d[12]; r.k_lo = d[13];
  return r;
}

// this block's bricks: XCD x (= blockIdx % 8: the hardware deals blocks round-robin) owns one contiguous eighth of the brick
// space (neighbouring bricks share box lines: that XCD's L2), its blocks walk it interleaved
// kind of a planned brick for a launch whose tiles hold `cap` floats (a plan made ahead may have been sized for another road)
__device__ __forceinline__ int brick_kind(const BrickDesc& d, int cap) {
  const int kind = d.kind_w & 0xFF;
  const bool staged_ok = static_cast<int64_t>(d.Lx) * d.Ly * (d.cpr * 4) <= static_cast<int64_t>(cap) && box_address_fits(d.bx0, d.by0, d.za, d.Lx, d.Ly, d.cpr);
  return (kind == kDescStaged && !staged_ok) ? static_cast<int>(kDescSlow) : kind;
}

struct BrickWalk {
  unsigned first, stride, count;
};
__device__ __forceinline__ BrickWalk brick_walk(unsigned n_items, unsigned block, unsigned n_blocks) {
  constexpr unsigned NX = 8;
  const unsigned xcd = block % NX, slot = block / NX, per_xcd = n_blocks / NX;  // (n_blocks is a multiple of 8)
  const unsigned q = n_items / NX, r = n_items % NX;
  const unsigned start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const unsigned len = q + (xcd < r ? 1u : 0u);
  BrickWalk w;
  w.first = start + slot; w.stride = per_xcd;
  w.count = slot < len ? (len - slot + per_xcd - 1) / per_xcd : 0u;
  return w;
}

// wait until at most `n` of this wave's vector-memory operations are outstanding (n wave uniform, small)
__device__ __forceinline__ void wait_vmcnt_at_most(int n) {
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // 0, or more than the cases above: everything
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the zero chunks of boxes that stick out of the volume are LDS stores)
  {
  }
}

// issue this wave's share of a brick's box into `tile`; returns the number of DMA instructions issued (wave uniform)
template <int NW>
__device__ __forceinline__ int persist_issue_box(const BrickDesc& d, int cap, float* tile, const LeanArgs& a, int wave, int lane) {
  if (brick_kind(d, cap) != kDescStaged) return 0;
  StreamBox bx;
  bx.kind = kDescStaged; bx.interior = d.kind_w >> 8; bx.bx0 = d.bx0; bx.by0 = d.by0; bx.za = d.za; bx.Lx = d.Lx; bx.Ly = d.Ly; bx.cpr = d.cpr;
  BoxDmaStepper<NW> dma;
  dma.init(tile, a.in + static_cast<int64_t>(d.b) * a.in_stride, bx, a.I, a.J, a.K, wave, lane);
  const int issued = dma.left;
  if (bx.interior) { while (dma.left > 0) dma.template issue<true>(lane); }
  else { while (dma.left > 0) dma.template issue<false>(lane); }
  // (a box that sticks out of the volume zeroes some chunks with plain LDS stores: ordered by the barrier like the DMA)
  return issued;
}

template <bool EXACT_LERP, bool FOLD_MIN = false>
__global__ __launch_bounds__(1024) void resample_lean_exact_persistent_kernel(const LeanArgs a) {
  constexpr int TI = 16, TJ = 16, TK = 16, NW = 16, PL = 4;  // PL planes per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef __attribute__((address_space(4))) const float* const_float_ptr;
  const int cap = a.tile_floats;  // floats per tile (the planner's tile_cap); THREE tiles: sampled / landed or landing / landing

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 2;                      // plane group of this wave: planes 4 pg .. 4 pg + 3 of every brick
  const int col = tid & 255;                     // column inside the brick (four waves per plane group, as in the brick kernels)
  const int tk = col & (TK - 1), tj = col / TK;
  const int* plan_bricks = a.plan + a.B * 16;
  const int slab = a.Jo * a.Ko;
  const int64_t slab_b = static_cast<int64_t>(slab) * 4;
  const bool has_fill = a.fill != nullptr;
  const float fillv = has_fill ? ((const_float_ptr)a.fill)[0] : 0.0f;
  const float hx = a.hx, hy = a.hy, hz = a.hz;
  const int64_t n_in_b = a.in_stride, n_out_b = a.out_stride;

  const BrickWalk walk = brick_walk(static_cast<unsigned>(a.n_items), blockIdx.x, gridDim.x);
  if (walk.count == 0) return;
  uint32_t kmin = 0xFFFFFFFFu;

  // prologue: the boxes of bricks 0 and 1 requested, the descriptors of bricks 0 .. 2 loaded; brick 0's box waited for
  BrickDesc dc = load_brick_desc(plan_bricks, walk.first);
  BrickDesc d1 = dc, d2 = dc;
  if (walk.count > 1) d1 = load_brick_desc(plan_bricks, walk.first + walk.stride);
  if (walk.count > 2) d2 = load_brick_desc(plan_bricks, walk.first + 2 * walk.stride);
  persist_issue_box<NW>(dc, cap, smem, a, wave, lane);
  int in_flight = 0;  // DMA instructions of this wave for the box BEHIND the one the next wait is about
  if (walk.count > 1) in_flight = persist_issue_box<NW>(d1, cap, smem + cap, a, wave, lane);
  wait_vmcnt_at_most(in_flight);
  __syncthreads();

  unsigned slot = 0;  // tile of the current brick (it % 3)
  for (unsigned it = 0; it < walk.count; it++) {
    float* const cur = smem + slot * cap;
    const unsigned slot2 = slot == 0 ? 2u : slot - 1u;  // (it + 2) % 3: the tile brick it - 1 was sampled from (free since the last barrier)
    const bool have_next = it + 1 < walk.count;
    // ---- the box of brick it + 2: requested now, two bricks ahead of its use --------------------------------------------
    int issued2 = 0;
    if (it + 2 < walk.count) issued2 = persist_issue_box<NW>(d2, cap, smem + slot2 * cap, a, wave, lane);
    // the descriptor after that (scalar loads: in flight during this brick's arithmetic)
    BrickDesc d3 = d2;
    if (it + 3 < walk.count) d3 = load_brick_desc(plan_bricks, walk.first + (it + 3) * walk.stride);

    // ---- this brick ----------------------------------------------------------------------------------------------------
    const int kind = brick_kind(dc, cap);
    const bool interior = (dc.kind_w >> 8) != 0;
    const int b = dc.b, i_begin = dc.i_begin, j_lo = dc.j_lo, k_lo = dc.k_lo;
    const int i_count = min(TI, a.Io - i_begin), nv = min(TJ, a.Jo - j_lo), nw = min(TK, a.Ko - k_lo);
    const bool col_active = (tj < nv) & (tk < nw);
    const bool full = (i_count == TI) & (nv == TJ) & (nw == TK);  // block uniform
    const int jv = min(tj, nv - 1), kw = min(tk, nw - 1);
    const int col_off = (j_lo + jv) * a.Ko + (k_lo + kw);
    const unsigned urow = static_cast<unsigned>(col_off) * 4u;
    const float* in_chan = a.in + static_cast<int64_t>(b) * n_in_b;
    char* out_chan = reinterpret_cast<char*>(a.out + static_cast<int64_t>(b) * n_out_b);
    const int p0 = i_begin + PL * pg;            // this thread's first plane
    const bool track = FOLD_MIN && a.min_keys != nullptr && b == 0;  // block uniform
    bool waited = false;                         // this wave has waited for its share of the next box's DMA

    if (kind == kDescGated || kind == kDescOutside) {
      if (col_active) {
        for (int t = p0; t < min(p0 + PL, i_begin + i_count); t++) {
          const float val = kind == kDescGated ? in_chan[static_cast<int64_t>(t) * slab + col_off] : fillv;
          *reinterpret_cast<float*>(out_chan + t * slab_b + urow) = val;
          if (track) kmin = min(kmin, float_to_key(val));
        }
      }
    } else {
      const_float_ptr mp = (const_float_ptr)(a.mapping) + (a.mapping_batched ? b * 12 : 0);
      float m[12];
#pragma unroll
      for (int q = 0; q < 12; q++) m[q] = mp[q];
      const float cj = static_cast<float>(j_lo + jv), ck = static_cast<float>(k_lo + kw);
      const int i_last = i_begin + i_count - 1;
      float X[PL], Y[PL], Z[PL];
#pragma unroll
      for (int t = 0; t < PL; t++) {
        const float ci = static_cast<float>(min(p0 + t, i_last));
        lean_exact_coord<0, true, true>(m, a, ci, cj, ck, 0.0f, 0.0f, 0.0f, X[t], Y[t], Z[t]);
      }
      if (kind == kDescSlow) {  // box beyond the tile / non-finite geometry: per-voxel global gathers (rare)
        if (col_active) {
#pragma unroll
          for (int t = 0; t < PL; t++) {
            if (p0 + t <= i_last) {
              const float val = lean_exact_gather(in_chan, a.J, a.K, X[t], Y[t], Z[t], has_fill, fillv, hx, hy, hz);
              *reinterpret_cast<float*>(out_chan + static_cast<int64_t>(p0 + t) * slab_b + urow) = val;
              if (track) kmin = min(kmin, float_to_key(val));
            }
          }
        }
      } else {
        TileAddr ta;
        ta.ox = static_cast<float>(dc.bx0); ta.oy = static_cast<float>(dc.by0); ta.oz = static_cast<float>(dc.za);
        ta.sYb = dc.cpr * 16; ta.sXb = dc.Ly * ta.sYb; ta.sXYb = ta.sXb + ta.sYb;
        ta.sYbf = static_cast<float>(ta.sYb); ta.sXbf = static_cast<float>(ta.sXb);
        ta.base_f = static_cast<float>(static_cast<unsigned>(reinterpret_cast<uintptr_t>((fast_lds_wptr)cur)));
        ta.c_f = ta.base_f - ta.ox * ta.sXbf - ta.oy * ta.sYbf - 4.0f * ta.oz;  // (brick_kind demotes the boxes this is not exact for)
        char* out_t = out_chan + static_cast<int64_t>(p0) * slab_b;
        const bool may_leave = has_fill & !interior;
        const int t0 = PL * pg;
        float vals[PL];
        if (full) {
          bool masked = false;
          if (may_leave) masked = __builtin_amdgcn_ballot_w64(lean_exact_group_leaves(X, Y, Z, hx, hy, hz)) != 0ull;
          if (masked) lean_exact_group_values<EXACT_LERP, true>(X, Y, Z, ta, hx, hy, hz, true, fillv, vals);
          else lean_exact_group_values<EXACT_LERP, false>(X, Y, Z, ta, hx, hy, hz, false, fillv, vals);
        } else {  // a partial brick (a volume edge that is not a multiple of 16): the mask wherever there is a fill rule
          lean_exact_group_values<EXACT_LERP, true>(X, Y, Z, ta, hx, hy, hz, has_fill, fillv, vals);
        }
        // The NEXT brick's box must have landed before the roles move on; the box behind it (requested at the top of this
        // iteration: `issued2` instructions of this wave) may stay in flight.  The wait comes BEFORE this brick's stores are
        // issued — the four values wait in registers — so it never includes them, and it holds whatever the order in which a
        // wave's loads and stores complete: loads complete in order, and "at most issued2 operations outstanding" leaves
        // room for nothing older than the last box.  (The previous brick's stores, a whole brick old, are waited for too.)
        if (have_next) wait_vmcnt_at_most(issued2);
        waited = true;
        if (full) {
          if (FOLD_MIN && track) lean_exact_group_store<false, true>(vals, out_t, urow, slab_b, t0, i_count, col_active, kmin);
          else lean_exact_group_store<false, false>(vals, out_t, urow, slab_b, t0, i_count, col_active, kmin);
        } else {
          if (FOLD_MIN && track) lean_exact_group_store<true, true>(vals, out_t, urow, slab_b, t0, i_count, col_active, kmin);
          else lean_exact_group_store<true, false>(vals, out_t, urow, slab_b, t0, i_count, col_active, kmin);
        }
      }
    }

    // ---- the next box has landed (every wave waited for its own DMA instructions) and every wave is done with this tile ----
    if (have_next) {
      if (!waited) tile_dma_wait_all();  // (paths with loads / stores of their own: everything)
      __syncthreads();
    }
    dc = d1; d1 = d2; d2 = d3;
    slot = slot == 2 ? 0u : slot + 1u;
  }
  if (FOLD_MIN && a.min_keys != nullptr) {  // one returnless atomic per wave for the whole walk
    const uint32_t wmin = wave_min_u32(kmin);
    if (lane == 0 && wmin != 0xFFFFFFFFu)
      __hip_atomic_fetch_min(a.min_keys + (blockIdx.x & (kMinSlots - 1)), wmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace tio
