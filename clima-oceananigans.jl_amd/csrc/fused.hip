// fused.hip -- the tiled kernels of the time_step! hot path (upwind-biased 5th-order advection: WENO5 Z / JS, U5).
//
//   k_tend4     : calculate_G{u,v,w}! + ab2_step_field! / rk3_substep_field! in ONE pass
//                 (calculate_nonhydrostatic_tendencies.jl:155-170, quasi_adams_bashforth_2.jl:158-166,
//                  runge_kutta_3.jl:204-218).  G^- <- G^n becomes a pointer rotation (store_tendencies.jl).
//                 Template flags: VISC (ScalarDiffusivity as face fluxes), ZB (Bounded z), REST (G^n arrives holding
//                 the non-advective terms computed by the general kernels), XT (rows wider than a workgroup or walls
//                 in x: x-tiles with a ghost column), DMA (slab staged by global_load_lds).
//   k_tracer_step: calculate_Gc! + update for passive tracers (periodic path)
//   k_rhs_wrap  : calculate_pressure_source_term_fft_based_solver! reading the predictor with periodic
//                 wrap indexing (no halo fill of U* needed)           (solve_for_pressure.jl:15-18)
//   k_project   : copy_real_component! + _pressure_correct_velocities! + the periodic halo fills of
//                 update_state! / calculate_pressure_correction!, all in one pass: each thread writes its
//                 interior value and its periodic images             (pressure_correction.jl:34-40,
//                 fill_halo_regions_periodic.jl:37-65)
//
// Design of the tendency kernel (why it looks the way it does):
//   * it is FP64-VALU bound, not HBM bound: one WENO reconstruction is ~60 DP operations and the reference
//     evaluates 36 per cell.  Here every face flux is evaluated once (9 per cell + ghost-row overhead) and only
//     on the upwind side (bitwise equal to upwind_biased_product for finite input).
//   * a workgroup owns complete x rows (x wrap stays inside the workgroup) of BY-1 output rows plus one
//     "ghost" row of threads that only produces the y-fluxes of the next row.
//   * each thread computes the fluxes through the WEST / SOUTH / BOTTOM faces of its three velocity cells.
//     EAST and NORTH fluxes come from the neighbouring threads through LDS; TOP fluxes are the next level's
//     BOTTOM fluxes, carried while the workgroup marches up its share of levels; z stencils live in a 6-deep
//     register window, x / y stencils in an LDS slab of the level.
#include "internal.h"

#define FUSED_MAX_THREADS 1024

struct FusedArgs {
  const double *u, *v, *w;        // PARENT-array base pointers (halos valid); all fields share one layout
  const double *gmu, *gmv, *gmw;  // G^-
  double *gnu, *gnv, *gnw;        // G^n
  double *us, *vs, *ws;           // predictor U*
  unsigned org;                   // byte offset of the first interior cell inside a parent array
  double dt, cn, cm;
  int use_m;
  int BYo;                        // output rows per workgroup (= blockDim.y - 1)
  int ntiles;                     // y-tiles (v3: segment decomposition); x-tiled variant: ntx * nty
  int ntx, BXo;                   // x-tiled variant: tiles along x, output columns per tile
  int prio;                       // wave-priority scheme of the tendency kernels (see PRIO_* below)
  double nu;                      // ScalarDiffusivity viscosity (0: none); see the viscous-flux note in k_tend_step3
  int zlo, zhi;                   // levels [zlo, zhi) of this launch (the whole column, or the part of a slab whose z halos are
  int zlo2, zhi2;                 // already / not yet there: api.hip fused_substep), optionally followed by a second run
  int gran;                       // [zlo2, zhi2); segments start on multiples of `gran` levels of that concatenated space
};

// Wave priorities (s_setprio).  The VALU of a SIMD is handed out by priority, then by age, so the four waves a SIMD holds
// (one per thread row) do not advance together: in-kernel stamps of round 2 show the ghost-row wave -- a third of the
// work, but the youngest -- starved until the others are done, and every wave then waiting for it at the barrier.
// s_setprio is scalar state: it must sit behind SCALAR branches (on a readfirstlane'd row index) -- behind a "divergent"
// branch on threadIdx.y it executes whatever EXEC says, and the last one in program order wins for every wave.
#ifndef OCN_HOST_EMU
#define OCN_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define OCN_SETPRIO(n) ((void)0)
#endif
// a.prio packs eight 2-bit priorities: bits [2r+1:2r] = priority of thread row class r at the start of a level, bits
// [8+2r+1:8+2r] from the middle of its flux stage on; classes: 0 = first output row, 1 = rows between, 2 = last output
// row, 3 = ghost row.  0 = leave the hardware default (age order).  All switches are scalar branches (sty is an SGPR).
OCN_DEVFN void prio_set(int v) {
  if (v == 0) OCN_SETPRIO(0); else if (v == 1) OCN_SETPRIO(1); else if (v == 2) OCN_SETPRIO(2); else OCN_SETPRIO(3);
}
OCN_DEVFN int prio_class(int sty, int BY) { return sty == BY - 1 ? 3 : sty == 0 ? 0 : sty == BY - 2 ? 2 : 1; }
OCN_DEVFN void prio_start(int code, int sty, int BY) {
  if (code) prio_set((code >> (2 * prio_class(sty, BY))) & 3);
}
OCN_DEVFN void prio_mid(int code, int sty, int BY) {
  if (code) prio_set((code >> (8 + 2 * prio_class(sty, BY))) & 3);
}

OCN_DEVFN void sto(double* base, unsigned boff, double v) { *(double*)((char*)base + boff) = v; }

// ---- k_tend4: tendencies of u, v, w + time-stepper update, ONE barrier per level --------------------------------------
// A workgroup owns BY-1 output rows (+ one ghost row of threads that only produces the south-face fluxes of the row above
// the tile) of either complete x rows (XT = false: Nx <= BX, the periodic wrap stays inside the workgroup) or of an
// x-tile of up to BX-7 output columns plus a ghost COLUMN that only produces west-face fluxes (XT = true: rows wider
// than a workgroup, or walls in x).  It marches up its share of levels; every thread forms the fluxes through the
// WEST / SOUTH / BOTTOM faces of its u, v, w cells (9 reconstructions per cell instead of the reference's 36), EAST
// and NORTH fluxes come from the neighbours, TOP fluxes are the next level's BOTTOM fluxes.
//
// What round 2 changed against the two-barrier kernel of round 1, and why (rocprofv3: VALU busy 58 %, every wave
// parked 46 % of its cycles, most of it correlated -- all 16 waves of the CU's only workgroup met at two barriers per
// level; in-kernel stamps: output-row waves waited 19 % of their life at the second barrier for the starved ghost row):
//   * the slab of the NEXT level goes straight from HBM into the other half of a double-buffered LDS slab with
//     global_load_lds_dwordx4 (DMA = true): a slab row is a contiguous run of a parent-array row, x halos included
//     (the projection / halo fills keep them current), so one wave-instruction moves 64 x 16 bytes of it; no VGPR
//     round trip, no ds_write, no image selects.  DMA = false (odd Nx, halo != 3) stages through registers.
//   * west-face fluxes reach the east neighbour by a lane shift inside the wave (v_mov_b32_dpp wave_shl:1); only the
//     first lane of each wave also drops its three values into a small LDS table for the last lane of the wave before
//     it.  South-face fluxes still go through LDS (the north neighbour is another wave), double buffered.
//   * the horizontal divergence and the bottom fluxes of the previous level are carried in registers (they lived in
//     LDS before: that space now holds the second slab buffer).
//   * with every LDS buffer double buffered by level parity a level needs one barrier: after it every thread has
//     finished the flux stage of level k-1, so that level's neighbour fluxes are complete, slab[k&1] has landed, and
//     slab[(k+1)&1] (last read at level k-1) is free to be refilled.  The update of level k-1 runs in the same
//     interval as the flux stage of level k.
//   * wave priorities (prio_start / prio_mid): the VALU goes to the highest priority, then to the oldest wave; without
//     them the youngest wave of each SIMD (the ghost row) starves and everybody waits for it.
// Measured at 256^3 (MI355X, profiles/r02_*): 0.739 -> 0.551 ms together with the leaner WENO algebra of stencils.h.
//
// ScalarDiffusivity (VISC; closure_kernel_operators.jl:22-41 with constant nu): div(2 nu Sigma)_i = nu (lap u_i + d_i div U),
// exactly (centred differences commute on a uniform grid).  Both parts are face fluxes at the very places of the advective
// ones: -nu d(u_i)/dn through every face, plus -nu div U through the centre-located face of the normal component.
// ZB: Bounded z (regular or stretched): per-level spacings and the 2nd-order fallback of every z stencil inside the
// boundary buffer (topologically_conditional_interpolation.jl:46-79).  REST: G^n arrives holding everything but advection
// (closure, Coriolis, pressure gradient, boundary fluxes from the general kernels) and leaves as the full tendency;
// walls in x / y are runtime flags of the REST variants (the same fallbacks on the x / y stencils).
template <int ADV, int BX, int BY, bool XT, int DMA, bool VISC, bool ZB, bool REST>
__global__ void __launch_bounds__(BX* BY) k_tend4(GridDev g, FusedArgs a) {
  constexpr int T = BX * BY, NR = BY + 5, SX = BX + 6;
  constexpr int WV = BX < OCN_WAVE ? BX : OCN_WAVE;   // lanes of a wave that lie in one row
  constexpr int NW = BX / WV;                         // waves per row
  constexpr int NWV = T / WV;                         // waves per workgroup
  constexpr int SLAB = 3 * NR * SX;                   // doubles per slab buffer; (f, r, s) <-> (row j0 - 3 + r, column i0 - 3 + s)
  constexpr int NG = (NR + BY - 1) / BY;              // row groups of the register-staged slab load (DMA = false)
  static_assert(SX % 2 == 0 && BX % WV == 0 && T % WV == 0, "slab rows must be whole 16-byte pieces");
  static_assert((2 * SLAB + 6 * T + 6 * BY * NW) * sizeof(double) <= OCN_LDS_BYTES, "k_tend4: LDS footprint over 160 KiB (gfx950)");
  OCN_SHARED double lds[2 * SLAB + 6 * T + 6 * BY * NW] __attribute__((aligned(16)));
  double* const fyb = lds + 2 * SLAB;                 // [parity][field][thread]: south-face fluxes
  double* const fxe = fyb + 6 * T;                    // [parity][field][row * NW + wave]: west-face fluxes of each wave's first lane
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * BX + tx;
  const int lane = tid % WV;
  const int wave = OCN_UNIFORM(tid / WV);            // scalar: the staging loops and the priority switches branch on it
  const int sty = OCN_UNIFORM(ty);                   // (BX is a multiple of the wave width: a wave lies in one row)
  const unsigned sxb = 8u, syb = (unsigned)g.sy * 8u, szb = (unsigned)g.sz * 8u;
  const double rdx = g.rdx, rdy = g.rdy, rdz = 1.0 / g.dz;
  const int nbz = (ADV == ADV_C4) ? 1 : 2;
  const bool ghost = (ty == BY - 1);
  const int nid_n = (ty + 1 < BY ? ty + 1 : ty) * BX + tx;
  // Work decomposition: the (tile, level) space is cut into gridDim.x equal segments of consecutive levels (tile-major),
  // so every workgroup marches the same number of levels -- no partial last round.  XCD-aware: workgroups b and b+8
  // share an L2, so each XCD gets a contiguous band of segments.
  const int nseg = gridDim.x, per = nseg / 8;
  const long seg = (nseg % 8 == 0) ? (long)(blockIdx.x % 8) * per + blockIdx.x / 8 : (long)blockIdx.x;
  const int n1 = a.zhi - a.zlo, NzR = n1 + (a.zhi2 - a.zlo2);
  const long total = (long)a.ntiles * NzR, units = total / a.gran;
  long lo = (seg * units / nseg) * a.gran;
  const long hi = ((seg + 1) * units / nseg) * a.gran;
  while (lo < hi) {
  const int tile = (int)(lo / NzR);
  const int vl = (int)(lo - (long)tile * NzR);                  // level index inside the (possibly two-piece) run of this tile
  const int k0 = vl < n1 ? a.zlo + vl : a.zlo2 + (vl - n1);
  const int kend = vl < n1 ? a.zhi : a.zhi2;
  const int k1 = (k0 + (hi - lo) < kend) ? (int)(k0 + (hi - lo)) : kend;
  lo += k1 - k0;
  const int ytile = XT ? tile / a.ntx : tile, xt = XT ? tile - ytile * a.ntx : 0;
  const int i0 = XT ? xt * a.BXo : 0;
  const int nout = XT ? ((g.Nx - i0 < a.BXo) ? g.Nx - i0 : a.BXo) : g.Nx;   // output columns; XT: column `nout` is the ghost column
  const int i = i0 + tx;
  const bool ocol = tx < nout;
  const bool col_ok = XT ? tx <= nout : ocol;                               // forms west-face fluxes
  const bool ldcol = XT ? tx < nout + 7 : ocol;                             // register staging: loads a slab column
  const int j0 = ytile * (BY - 1);
  const int j = j0 + ty;
  const bool row_ok = j < g.Ny;
  const bool do_y = ocol && j <= g.Ny;
  const bool full = ocol && row_ok && !ghost;
  const bool do_x = col_ok && row_ok && !ghost;
  // east neighbour: the next lane, except for the last lane of a wave and (complete rows) the periodic wrap at Nx - 1
  const bool xedge = (tx % WV == WV - 1) || (!XT && tx + 1 >= g.Nx);
  const int txe = (!XT && tx + 1 >= g.Nx) ? 0 : (tx + 1 < BX ? tx + 1 : tx);
  const int eidx = ty * NW + txe / WV;
  const unsigned cxy = a.org + (unsigned)(col_ok ? i : i0) * sxb + (unsigned)(j <= g.Ny ? j : 0) * syb;

  // ---- slab staging ------------------------------------------------------------------------------------------------
  // DMA == 1: slab row r of a field = columns i0-3 ... of parent row j0-3+r, contiguous in memory; PR 16-byte pieces of it
  // are needed (to the end of the parent row at most).  Wave w takes the (field, row) pairs w, w + NWV, ...
  const int ncols = (g.Nx + 6 - i0 < SX) ? g.Nx + 6 - i0 : SX;
  const int PR = ncols / 2;
  constexpr int NP = NR * SX / 2, NPR = (NP + T - 1) / T;   // DMA == 2: 16-byte pieces of a field's whole slab
  auto dma = [&](int k, int buf) {
    const long src0 = (long)a.org + ((long)(j0 - 3) * g.sy + (long)k * g.sz + (i0 - 3)) * 8;
    if (DMA == 2) {
      // complete rows whose pitch equals the slab's (Nx == BX): the NR rows of a field are ONE contiguous chunk
#pragma unroll
      for (int r = 0; r < NPR; ++r) {
        const int p = tid + r * T;
        if (p < NP) {
          const unsigned so = (unsigned)src0 + 16u * (unsigned)p;
          char* dst = (char*)(lds + buf * SLAB) + 16 * (p - lane);
          ocn_glds16((const char*)a.u + so, dst, lane);
          ocn_glds16((const char*)a.v + so, dst + NR * SX * 8, lane);
          ocn_glds16((const char*)a.w + so, dst + 2 * NR * SX * 8, lane);
        }
      }
      return;
    }
    for (int fr = wave; fr < 3 * NR; fr += NWV) {
      const int f = fr / NR, r = fr - f * NR;
      const double* base = f == 0 ? a.u : f == 1 ? a.v : a.w;
      const unsigned so = (unsigned)(src0 + (long)r * g.sy * 8);
      char* dst = (char*)(lds + buf * SLAB + fr * SX);
      for (int q0 = 0; q0 < PR; q0 += WV)
        if (q0 + lane < PR) ocn_glds16((const char*)base + (so + 16u * (unsigned)(q0 + lane)), dst + 16 * q0, lane);
    }
  };
  const unsigned grow = a.org + (unsigned)((XT ? (ldcol ? i0 + tx : i0 + 3) - 3 : (ocol ? tx : 0))) * sxb;   // XT: slab column tx <-> global column i0 - 3 + tx
  double pf[DMA ? 1 : 3][DMA ? 1 : NG];
  auto prefetch = [&](int k) {
    if (DMA) return;
#pragma unroll
    for (int gq = 0; gq < (DMA ? 0 : NG); ++gq) {
      int r = ty + BY * gq;
      if (r < NR) {
        int jg = j0 - 3 + r;
        if (jg > g.Ny + 2) jg = g.Ny + 2;              // rows past the halo are never used
        unsigned o = grow + (unsigned)(jg + 3) * syb - 3u * syb + (unsigned)k * szb;
        pf[0][gq] = ldo(a.u, o);
        pf[1 % (DMA ? 1 : 3)][gq] = ldo(a.v, o);
        pf[2 % (DMA ? 1 : 3)][gq] = ldo(a.w, o);
      }
    }
  };
  const bool img_e = tx < 3, img_w = tx >= g.Nx - 3;
  auto commit = [&](int buf) {
    if (DMA) return;
#pragma unroll
    for (int gq = 0; gq < (DMA ? 0 : NG); ++gq) {
      int r = ty + BY * gq;
      if (r < NR && ldcol) {
#pragma unroll
        for (int fl = 0; fl < 3; ++fl) {
          double* row = lds + buf * SLAB + (fl * NR + r) * SX;
          double val = pf[fl % (DMA ? 1 : 3)][gq];
          if (XT) row[tx] = val;                       // x halos come from the arrays' own halo columns
          else {
            row[tx + 3] = val;
            if (img_e) row[tx + 3 + g.Nx] = val;       // periodic images inside LDS
            if (img_w) row[tx + 3 - g.Nx] = val;
          }
        }
      }
    }
  };
#define SLB(f, d, e) S[(f) * NR * SX + (d) * SX + (e)]

  double zu[6], zv[6], zw[6];
  {
    const unsigned c = cxy + (unsigned)k0 * szb;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      zu[q] = ldo(a.u, c + (unsigned)(q - 3) * szb);
      zv[q] = ldo(a.v, c + (unsigned)(q - 3) * szb);
      zw[q] = ldo(a.w, c + (unsigned)(q - 3) * szb);
    }
  }
  double dprev = 0.0;
  if (VISC) {   // div U of the cell below the first level of this march
    const unsigned cb = cxy + (unsigned)k0 * szb - szb;
    dprev = (ldo(a.u, cb + sxb) - zu[2]) * rdx + (ldo(a.v, cb + syb) - zv[2]) * rdy + (zw[3] - zw[2]) * rdz;
  }
  // carried from the previous level: horizontal divergence without the north (and, on x-edge lanes, east) fluxes, bottom fluxes
  double hu = 0, hv = 0, hw = 0, bu = 0, bv = 0, bw = 0;
  __syncthreads();                          // a previous segment's readers are done with every LDS buffer
  // buffer parity is counted from the first level of the march.  (A two-level form of the loop with compile-time parity
  // was tried: 744 instead of 761 VALU instructions per level -- the register windows still rotate by v_mov -- at 128
  // VGPRs with spills; not kept.)
  if (DMA) dma(k0, 0);
  else {
    prefetch(k0);
    commit(0);
    prefetch(k0 + 1);                       // k1 > k0, and level k1 is staged too: its w feeds the last bottom fluxes
  }
  // Order of a level (round 3).  vmcnt counts loads and stores together and, once both kinds are outstanding, can only be
  // waited to zero; the level of round 2 ended with the z-window loads of the next level and the six stores of the update,
  // so every wave sat out a full memory round trip in front of the barrier -- all of them at the same time (rocprofv3:
  // VALU busy 79 %).  Now: every load a level needs (G^-, rest terms, the NEXT level's window entries) is issued right after
  // the barrier; the z reconstructions come first and the update of level k-1 -- the only consumer of those loads -- follows,
  // with its stores issued BEFORE the slab DMA of level k+1 and the long x / y flux stage, under which both complete; the
  // window shift at the end is register moves only.
  for (int k = k0; k <= k1; ++k) {
    const int kb = (k - k0) & 1;
    const unsigned c = cxy + (unsigned)k * szb;
    const bool last = (k == k1);
    __syncthreads();                        // the one barrier of the level (with DMA in flight it also waits for vmcnt(0))
    if (!last && !DMA) {                    // register staging: level k+1 into the buffer level k-1 was read from
      commit(kb ^ 1);
      if (k + 2 <= k1) prefetch(k + 2);
    }
    // Loads are unconditional inside their region and every loaded value is used on every path (clamped addresses where
    // the value is not needed: first level of a march, steps without G^-, last level): a register that may still have a
    // load in flight when it is next written makes the compiler drain vmcnt to zero in the middle of the level.
    const unsigned cnx = last ? c : c + 3 * szb;       // the window entries level k+1 will need on top
    double znu, znv, znw;
    constexpr bool visc = VISC;   // compile-time: the inviscid kernel must not pay registers for these terms
    double wxm = 0, wym = 0;
    if (visc && !last) {
      wxm = ldo(a.w, c + szb - sxb);
      wym = ldo(a.w, c + szb - syb);
    }
    prio_start(a.prio, sty, BY);
    const double* S = lds + kb * SLAB + ty * SX + tx;     // lowest corner of this thread's stencil footprint; own cell at (3, 3)
    // idx: the reference's 1-based index of the evaluation point along z (face k+1 for the bottom face of level k,
    // centre k for the w flux below face k); all conditions are uniform over the workgroup
    auto symz_at = [&](const double* z, int idx) {
      if (ZB && !(idx > nbz && idx < g.Nz + 1 - nbz)) return 0.5 * (z[2] + z[3]);
      return sym4_v(z[1], z[2], z[3], z[4]);
    };
    auto symz = [&](const double* z) { return symz_at(z, k + 1); };
    auto reconz_at = [&](const double* z, double ut, int idx) {
      bool pos = ut > 0.0;
      if (ZB) {
        const bool ok = pos ? (idx > nbz && idx < g.Nz + 1 - (nbz - 1)) : (idx > nbz - 1 && idx < g.Nz + 1 - nbz);
        if (!ok) return 0.5 * (z[2] + z[3]);
      }
      return recon5<ADV>(pos ? z[0] : z[5], pos ? z[1] : z[4], pos ? z[2] : z[3], pos ? z[3] : z[2],
                         pos ? z[4] : z[1], pos);
    };
    auto reconz = [&](const double* z, double ut) { return reconz_at(z, ut, k + 1); };
    auto sym_v = [&](double m2, double m1, double c0, double c1) { return sym4_v(m2, m1, c0, c1); };  // midway m1|c0
    auto rec_v = [&](double m3, double m2, double m1, double c0, double c1, double c2, double ut) {
      bool pos = ut > 0.0;                                            // face between m1 and c0
      return recon5<ADV>(pos ? m3 : c2, pos ? m2 : c1, pos ? m1 : c0, pos ? c0 : m1, pos ? c1 : m2, pos);
    };
    // Walls in x / y (REST variants only; g.xb / g.yb are runtime flags): inside the boundary buffer every stencil falls
    // back to 2nd order, exactly as adv_flux_b / sym_b of the general kernels (idx: 1-based index along the stencil).
    auto in_sym = [&](int idx, int N) { return idx > nbz && idx < N + 1 - nbz; };
    auto in_rec = [&](bool pos, int idx, int N) {
      return pos ? (idx > nbz && idx < N + 1 - (nbz - 1)) : (idx > nbz - 1 && idx < N + 1 - nbz);
    };
    const bool wx = XT && REST && g.xb != 0, wy = REST && g.yb != 0;
    const int ix = i + 1, jy = j + 1;                 // 1-based face indices of this thread's u / v cells
#define XSYM(f, idx) ((wx && !in_sym(idx, g.Nx)) ? 0.5 * (SLB(f, 3, 2) + SLB(f, 3, 3)) \
                                               : sym_v(SLB(f, 3, 1), SLB(f, 3, 2), SLB(f, 3, 3), SLB(f, 3, 4)))
#define YSYM(f, idx) ((wy && !in_sym(idx, g.Ny)) ? 0.5 * (SLB(f, 2, 3) + SLB(f, 3, 3)) \
                                               : sym_v(SLB(f, 1, 3), SLB(f, 2, 3), SLB(f, 3, 3), SLB(f, 4, 3)))
#define XREC(f, ut, idx) ((wx && !in_rec((ut) > 0.0, idx, g.Nx)) ? 0.5 * (SLB(f, 3, 2) + SLB(f, 3, 3)) \
                              : rec_v(SLB(f, 3, 0), SLB(f, 3, 1), SLB(f, 3, 2), SLB(f, 3, 3), SLB(f, 3, 4), SLB(f, 3, 5), ut))
#define YREC(f, ut, idx) ((wy && !in_rec((ut) > 0.0, idx, g.Ny)) ? 0.5 * (SLB(f, 2, 3) + SLB(f, 3, 3)) \
                              : rec_v(SLB(f, 0, 3), SLB(f, 1, 3), SLB(f, 2, 3), SLB(f, 3, 3), SLB(f, 4, 3), SLB(f, 5, 3), ut))
    // ---- bottom-face fluxes of level k (= top-face fluxes of level k-1), then the update of level k-1 --------------------
    if (full) {
      const bool upd = k > k0;
      const unsigned cm1 = upd ? c - szb : c;         // level k-1 (first level of a march: nothing to update, loads unused)
      double rs0 = 0, rs1 = 0, rs2 = 0;
      if (REST) {
        rs0 = ldo(a.gnu, cm1);
        rs1 = ldo(a.gnv, cm1);
        rs2 = ldo(a.gnw, cm1);
      }
      const unsigned cgm = a.use_m ? cm1 : a.org;     // without G^- (Euler start, first RK3 stage): one cached element
      double gm0 = ldo(a.gmu, cgm), gm1 = ldo(a.gmv, cgm), gm2 = ldo(a.gmw, cgm);
      znu = ldo(a.u, cnx);
      znv = ldo(a.v, cnx);
      znw = ldo(a.w, cnx);
      double wtu = XSYM(2, ix);
      double Fwu = wtu * reconz(zu, wtu);
      double wtv = YSYM(2, jy);
      double Fwv = wtv * reconz(zv, wtv);
      double wtw = symz_at(zw, k);                    // centre below face k
      double Fww = wtw * reconz_at(zw, wtw, k);
      if (visc) {
        Fwu -= a.nu * (zu[3] - zu[2]) * rdz;
        Fwv -= a.nu * (zv[3] - zv[2]) * rdz;
        const double dwz = (zw[3] - zw[2]) * rdz;
        Fww -= a.nu * (dwz + dprev);                    // dprev = div U at (i, j, k-1)
        if (!last) dprev = (SLB(0, 3, 4) - SLB(0, 3, 3)) * rdx + (SLB(1, 4, 3) - SLB(1, 3, 3)) * rdy + (zw[4] - zw[3]) * rdz;
      }
      // complete the horizontal divergence of level k-1 with the neighbours' fluxes (complete since the barrier)
      const double* fyp = fyb + (kb ^ 1) * 3 * T;
      hu = fma(fyp[0 * T + nid_n], rdy, hu);
      hv = fma(fyp[1 * T + nid_n], rdy, hv);
      hw = fma(fyp[2 * T + nid_n], rdy, hw);
      if (xedge) {
        const double* fxp = fxe + (kb ^ 1) * 3 * BY * NW;
        hu = fma(fxp[0 * BY * NW + eidx], rdx, hu);
        hv = fma(fxp[1 * BY * NW + eidx], rdx, hv);
        hw = fma(fxp[2 * BY * NW + eidx], rdx, hw);
      }
      const int km = upd ? k - 1 : k;
      const double rzc = ZB ? g_rdzc(g, km) : rdz, rzf = ZB ? g_rdzf(g, km) : rdz;
      const double Gu = rs0 - (hu + (Fwu - bu) * rzc);
      const double Gv = rs1 - (hv + (Fwv - bv) * rzc);
      const double Gw = rs2 - (hw + (Fww - bw) * rzf);
      OCN_TOUCH3(gm0, gm1, gm2);                      // the loads land HERE on every path (see above), G^- used or not
      double iu, iv, iw;
      if (a.use_m) {
        iu = a.dt * (a.cn * Gu + a.cm * gm0);
        iv = a.dt * (a.cn * Gv + a.cm * gm1);
        iw = a.dt * (a.cn * Gw + a.cm * gm2);
      } else {
        iu = a.dt * a.cn * Gu;
        iv = a.dt * a.cn * Gv;
        iw = a.dt * a.cn * Gw;
      }
      if (upd) {                                      // all six stores together, after the last use of a loaded value
        sto(a.gnu, cm1, Gu);
        sto(a.gnv, cm1, Gv);
        sto(a.gnw, cm1, Gw);
        sto(a.us, cm1, zu[2] + iu);
        sto(a.vs, cm1, zv[2] + iv);
        sto(a.ws, cm1, zw[2] + iw);
      }
      bu = Fwu;
      bv = Fwv;
      bw = Fww;
    } else {
      znu = ldo(a.u, cnx);
      znv = ldo(a.v, cnx);
      znw = ldo(a.w, cnx);
    }
    if (DMA && !last) dma(k + 1, kb ^ 1);    // level k+1 (up to k1) into the buffer level k-1 was read from; lands under the flux stage
    // ---- west- and south-face fluxes of level k --------------------------------------------------------------------------
    double f0 = 0, f1 = 0, f2 = 0;            // fluxes through the west faces of this thread's u, v, w cells (level k)
    double s0 = 0, s1 = 0, s2 = 0;            // ... through the south faces
    if (!last) {
      if (do_x) {
        double utu = XSYM(0, ix - 1);                  // centre i-1
        f0 = utu * XREC(0, utu, ix - 1);
        if (visc) {
          const double dux = (SLB(0, 3, 3) - SLB(0, 3, 2)) * rdx;
          const double divw = dux + (SLB(1, 4, 2) - SLB(1, 3, 2)) * rdy + (wxm - SLB(2, 3, 2)) * rdz;   // div U at (i-1, j, k)
          f0 -= a.nu * (dux + divw);
        }
        double utv = YSYM(0, jy);                      // u interpolated in y to the v row
        f1 = utv * XREC(1, utv, ix) - (visc ? a.nu * (SLB(1, 3, 3) - SLB(1, 3, 2)) * rdx : 0.0);
        double utw = symz(zu);                         // u interpolated in z to the w level
        f2 = utw * XREC(2, utw, ix) - (visc ? a.nu * (SLB(2, 3, 3) - SLB(2, 3, 2)) * rdx : 0.0);
      }
      prio_mid(a.prio, sty, BY);
      if (do_y) {
        double vtu = XSYM(1, ix);                      // v interpolated in x to the u column
        s0 = vtu * YREC(0, vtu, jy) - (visc ? a.nu * (SLB(0, 3, 3) - SLB(0, 2, 3)) * rdy : 0.0);
        double vtv = YSYM(1, jy - 1);                  // centre j-1
        s1 = vtv * YREC(1, vtv, jy - 1);
        if (visc) {
          const double dvy = (SLB(1, 3, 3) - SLB(1, 2, 3)) * rdy;
          const double divs = (SLB(0, 2, 4) - SLB(0, 2, 3)) * rdx + dvy + (wym - SLB(2, 2, 3)) * rdz;   // div U at (i, j-1, k)
          s1 -= a.nu * (dvy + divs);
        }
        double vtw = symz(zv);
        s2 = vtw * YREC(2, vtw, jy) - (visc ? a.nu * (SLB(2, 3, 3) - SLB(2, 2, 3)) * rdy : 0.0);
        double* fyn = fyb + kb * 3 * T;
        fyn[0 * T + tid] = s0;
        fyn[1 * T + tid] = s1;
        fyn[2 * T + tid] = s2;
      }
    }
#undef XSYM
#undef YSYM
#undef XREC
#undef YREC
    if (!last) {
      // east fluxes: the next lane's west fluxes (uniform control flow: every lane of every wave takes part)
      const double e0 = ocn_shfl_next(f0), e1 = ocn_shfl_next(f1), e2 = ocn_shfl_next(f2);
      if (tx % WV == 0 && do_x) {
        double* fxn = fxe + kb * 3 * BY * NW + ty * NW + tx / WV;
        fxn[0 * BY * NW] = f0;
        fxn[1 * BY * NW] = f1;
        fxn[2 * BY * NW] = f2;
      }
      if (full) {
        // (east - west) / dx - south / dy now; north / dy (and east / dx on x-edge lanes) after the next barrier
        hu = fma(xedge ? -f0 : e0 - f0, rdx, -s0 * rdy);
        hv = fma(xedge ? -f1 : e1 - f1, rdx, -s1 * rdy);
        hw = fma(xedge ? -f2 : e2 - f2, rdx, -s2 * rdy);
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        zu[q] = zu[q + 1];
        zv[q] = zv[q + 1];
        zw[q] = zw[q + 1];
      }
      zu[5] = znu;
      zv[5] = znv;
      zw[5] = znw;
    }
  }
  }  // segments
#undef SLB
}

// ---- k_rest4: the non-advective momentum terms of the "rest + tiled advection" models, tiled like k_tend4 ------------------
// closure stress divergence  2 d_j (nu Sigma_ij)  with a constant nu (ScalarDiffusivity) or the eddy viscosity field of
// AnisotropicMinimumDissipation interpolated to the stress locations (closure_kernel_operators.jl:22-41,72-90,
// velocity_tracer_gradients.jl), Coriolis on the f-plane (f_plane.jl:42-44) and the hydrostatic pressure gradient
// (nonhydrostatic_tendency_kernel_functions.jl:44-106).  k_tend_uvw<ADV_NONE> evaluates every stress twice (once per cell
// on either side) from ~60 cached loads per thread; here every thread forms the six stresses at the WEST / SOUTH / BOTTOM
// of its cell once -- tau11 at centre i-1, tau22 at centre j-1, tau33 at centre k-1, tau12 at the (x-face, y-face) corner,
// tau13 at (x-face, z-face), tau23 at (y-face, z-face); tau12 serves u's south and v's west flux, tau13 u's bottom and w's
// west, tau23 v's bottom and w's south -- east values come by lane shift, north values through LDS, top values are the next
// level's bottom values.  Slab: u, v, w, nu_e with one halo row / column, both buffers, staged by LDS DMA (complete rows
// only: the x-tiled and wall-in-x models keep the general kernel).  Output: G^n = these terms (the flux-boundary kernels and
// k_tend4<REST> add theirs afterwards).
struct RestArgs {
  const double *u, *v, *w, *nue, *pH;   // PARENT bases (nue / pH may be null)
  double *gu, *gv, *gw;
  unsigned org;
  double nu, f;                         // constant viscosity (nue == null), Coriolis parameter
  int closure, coriolis, ntiles;
};

template <int BX, int BY, bool ZB>
__global__ void __launch_bounds__(BX* BY) k_rest4(GridDev g, RestArgs a) {
  constexpr int T = BX * BY, NR = BY + 1, SX = BX + 6;      // rows j0-1 .. j0+BY-1; columns -3 .. Nx+2 (the parent row)
  constexpr int WV = BX < OCN_WAVE ? BX : OCN_WAVE, NW = BX / WV, NWV = T / WV;
  constexpr int SLAB = 4 * NR * SX;
  static_assert((2 * SLAB + 6 * T + 6 * BY * NW) * sizeof(double) <= OCN_LDS_BYTES, "k_rest4: LDS footprint over 160 KiB (gfx950)");
  OCN_SHARED double lds[2 * SLAB + 6 * T + 6 * BY * NW] __attribute__((aligned(16)));
  double* const fyb = lds + 2 * SLAB;
  double* const fxe = fyb + 6 * T;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * BX + tx;
  const int lane = tid % WV;
  const int wave = OCN_UNIFORM(tid / WV);
  const unsigned sxb = 8u, syb = (unsigned)g.sy * 8u, szb = (unsigned)g.sz * 8u;
  const double rdx = g.rdx, rdy = g.rdy;
  const bool ghost = (ty == BY - 1);
  const int nid_n = (ty + 1 < BY ? ty + 1 : ty) * BX + tx;
  const bool xedge = (tx % WV == WV - 1) || (tx + 1 >= g.Nx);
  const int txe = (tx + 1 >= g.Nx) ? 0 : tx + 1;
  const int eidx = ty * NW + txe / WV;
  const bool amd = a.nue != nullptr;
  const int nseg = gridDim.x, per = nseg / 8;
  const long seg = (nseg % 8 == 0) ? (long)(blockIdx.x % 8) * per + blockIdx.x / 8 : (long)blockIdx.x;
  const long total = (long)a.ntiles * g.Nz;
  long lo = seg * total / nseg;
  const long hi = (seg + 1) * total / nseg;
  const int PR = (g.Nx + 6) / 2;
  while (lo < hi) {
    const int tile = (int)(lo / g.Nz);
    const int k0 = (int)(lo - (long)tile * g.Nz);
    const int k1 = (k0 + (hi - lo) < g.Nz) ? (int)(k0 + (hi - lo)) : g.Nz;
    lo += k1 - k0;
    const int j0 = tile * (BY - 1);
    const int j = j0 + ty;
    const bool ocol = tx < g.Nx;
    const bool do_y = ocol && j <= g.Ny;              // forms the south-face stresses
    const bool full = ocol && j < g.Ny && !ghost;
    const unsigned cxy = a.org + (unsigned)(ocol ? tx : 0) * sxb + (unsigned)(j <= g.Ny ? j : 0) * syb;
    auto dma = [&](int k, int buf) {                  // rows j0-1 .. of level k, every field; wave w takes (field, row) pairs
      const long src0 = (long)a.org + ((long)(j0 - 1) * g.sy + (long)k * g.sz - 3) * 8;
      const int nf = amd ? 4 : 3;
      for (int fr = wave; fr < nf * NR; fr += NWV) {
        const int f = fr / NR, r = fr - f * NR;
        const double* base = f == 0 ? a.u : f == 1 ? a.v : f == 2 ? a.w : a.nue;
        const unsigned so = (unsigned)(src0 + (long)r * g.sy * 8);
        char* dst = (char*)(lds + buf * SLAB + fr * SX);
        for (int q0 = 0; q0 < PR; q0 += WV)
          if (q0 + lane < PR) ocn_glds16((const char*)base + (so + 16u * (unsigned)(q0 + lane)), dst + 16 * q0, lane);
      }
    };
    // element (field f, row offset d in {-1, 0, +1}, column offset e) of the level's slab, relative to this thread's cell
#define RS(f, d, e) S[(f) * NR * SX + ((d) + 1) * SX + (e) + 3]
    // carried from the level below: own u, v, w, nu and nu of the west / south neighbours; horizontal part + bottom stresses
    const unsigned cb = cxy + (unsigned)k0 * szb - szb;
    double up = ldo(a.u, cb), vp = ldo(a.v, cb), wp = ldo(a.w, cb);
    double ncp = amd ? ldo(a.nue, cb) : a.nu, nwp = amd ? ldo(a.nue, cb - sxb) : a.nu, nsp = amd ? ldo(a.nue, cb - syb) : a.nu;
    double hu = 0, hv = 0, hw = 0, bu = 0, bv = 0, bw = 0;
    __syncthreads();
    dma(k0, 0);
    for (int k = k0; k <= k1; ++k) {
      const int kb = (k - k0) & 1;
      const bool last = (k == k1);
      const unsigned c = cxy + (unsigned)k * szb;
      __syncthreads();
      if (!last) dma(k + 1, kb ^ 1);
      const double* S = lds + kb * SLAB + ty * SX + tx;
      // hydrostatic pressure of this cell and its west / south neighbours (level k), issued early
      double p0 = 0, pw = 0, ps = 0;
      if (a.pH && full && !last) {
        p0 = ldo(a.pH, c);
        pw = ldo(a.pH, c - sxb);
        ps = ldo(a.pH, c - syb);
      }
      if (full && k > k0) {                           // complete level k-1's horizontal part with the north / x-edge values
        const double* fyp = fyb + (kb ^ 1) * 3 * T;
        const double rdy2 = rdy + rdy, rdx2 = rdx + rdx;       // the stress divergence carries a factor 2
        hu = fma(fyp[0 * T + nid_n], rdy2, hu);
        hv = fma(fyp[1 * T + nid_n], rdy2, hv);
        hw = fma(fyp[2 * T + nid_n], rdy2, hw);
        if (xedge) {
          const double* fxp = fxe + (kb ^ 1) * 3 * BY * NW;
          hu = fma(fxp[0 * BY * NW + eidx], rdx2, hu);
          hv = fma(fxp[1 * BY * NW + eidx], rdx2, hv);
          hw = fma(fxp[2 * BY * NW + eidx], rdx2, hw);
        }
      }
      const double rzf = ZB ? g_rdzf(g, k) : g.rdz;   // 1 / dz at face k
      const double uc = RS(0, 0, 0), vc = RS(1, 0, 0), wc = RS(2, 0, 0);
      const double nc = amd ? RS(3, 0, 0) : a.nu, nw = amd ? RS(3, 0, -1) : a.nu, ns = amd ? RS(3, -1, 0) : a.nu;
      // stresses nu * S at the west / south / bottom of this cell (level k)
      double t11 = 0, t12 = 0, t13 = 0, t22 = 0, t23 = 0, t33 = 0;
      if (a.closure != OCN_CLOSURE_NONE && do_y) {
        const double nsw = amd ? RS(3, -1, -1) : a.nu;
        t12 = 0.25 * ((nsw + ns) + (nw + nc)) * (0.5 * ((uc - RS(0, -1, 0)) * rdy + (vc - RS(1, 0, -1)) * rdx));     // (x-face i, y-face j)
        t22 = ns * ((vc - RS(1, -1, 0)) * rdy);                                                                   // centre j-1
        t23 = 0.25 * ((nsp + ncp) + (ns + nc)) * (0.5 * ((vc - vp) * rzf + (wc - RS(2, -1, 0)) * rdy));             // (y-face j, z-face k)
        if (full) {
          t11 = nw * ((uc - RS(0, 0, -1)) * rdx);                                                                  // centre i-1
          t13 = 0.25 * ((nwp + ncp) + (nw + nc)) * (0.5 * ((uc - up) * rzf + (wc - RS(2, 0, -1)) * rdx));           // (x-face i, z-face k)
          t33 = ncp * ((wc - wp) * (ZB ? g_rdzc(g, k - 1) : g.rdz));                                               // centre k-1
        }
      }
      if (!last && do_y) {
        double* fyn = fyb + kb * 3 * T;
        fyn[0 * T + tid] = t12;
        fyn[1 * T + tid] = t22;
        fyn[2 * T + tid] = t23;
      }
      double e0 = 0, e1 = 0, e2 = 0;
      if (!last) {
        e0 = ocn_shfl_next(t11);
        e1 = ocn_shfl_next(t12);
        e2 = ocn_shfl_next(t13);
        if (tx % WV == 0 && full) {
          double* fxn = fxe + kb * 3 * BY * NW + ty * NW + tx / WV;
          fxn[0 * BY * NW] = t11;
          fxn[1 * BY * NW] = t12;
          fxn[2 * BY * NW] = t13;
        }
      }
      if (full) {
        if (k > k0) {                                 // level k-1: z differences with this level's bottom stresses on top
          const unsigned cm1 = c - szb;
          const double rzc = ZB ? g_rdzc(g, k - 1) : g.rdz, rzfm = ZB ? g_rdzf(g, k - 1) : g.rdz;
          sto(a.gu, cm1, hu + 2.0 * ((t13 - bu) * rzc));
          sto(a.gv, cm1, hv + 2.0 * ((t23 - bv) * rzc));
          sto(a.gw, cm1, hw + 2.0 * ((t33 - bw) * rzfm));   // w at face k-1: tau33 at centre k-1 (here) minus centre k-2 (carried)
        }
        if (!last) {
          // level k: 2 [(east - west) / dx - south / dy] now (north / dy, x-edge east / dx after the next barrier), plus the
          // pointwise terms
          double cu = 0, cv = 0;
          if (a.coriolis) {
            cu = a.f * (0.5 * (0.5 * (RS(1, 0, -1) + vc) + 0.5 * (RS(1, 1, -1) + RS(1, 1, 0))));
            cv = -a.f * (0.5 * (0.5 * (RS(0, -1, 0) + RS(0, -1, 1)) + 0.5 * (uc + RS(0, 0, 1))));
          }
          if (a.pH) {
            cu -= (p0 - pw) * rdx;
            cv -= (p0 - ps) * rdy;
          }
          hu = cu + 2.0 * fma(xedge ? -t11 : e0 - t11, rdx, -t12 * rdy);
          hv = cv + 2.0 * fma(xedge ? -t12 : e1 - t12, rdx, -t22 * rdy);
          hw = 2.0 * fma(xedge ? -t13 : e2 - t13, rdx, -t23 * rdy);
          bu = t13;
          bv = t23;
          bw = t33;
        }
      }
      up = uc; vp = vc; wp = wc;
      ncp = nc; nwp = nw; nsp = ns;
    }
  }
#undef RS
}

// ---- Poisson right-hand side with wrap indexing (no halo fill of the predictor) ----------------------
__global__ void k_rhs_wrap(GridDev g, const double* __restrict__ us, const double* __restrict__ vs,
                           const double* __restrict__ ws, double rdt, int zwrap, double* __restrict__ rhs) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long sy = g.sy, sz = g.sz;
  const long c = i + j * sy + k * sz;
  const long ce = (i + 1 == g.Nx) ? c + 1 - g.Nx : c + 1;
  const long cn = (j + 1 == g.Ny) ? c + sy - g.Ny * sy : c + sy;
  const long ct = (zwrap && k + 1 == g.Nz) ? c + sz - g.Nz * sz : c + sz;
  double div = (us[ce] - us[c]) * g.rdx + (vs[cn] - vs[c]) * g.rdy + (ws[ct] - ws[c]) / g.dz;
  rhs[i + (long)g.Nx * (j + (long)g.Ny * k)] = div * rdt;
}

// ---- projection + pressure copy + all periodic halo images --------------------------------------------
struct ProjArgs {
  const double* phi;              // compact (Nx,Ny,Nz) solver output
  const double *us, *vs, *ws;     // predictor (interior-origin pointers)
  double *u, *v, *w, *p;          // destination fields (interior-origin pointers)
  const double* phi_below;        // (Nx,Ny) pressure of the level below the slab (slab runs), else null
  double dt;
  int zwrap;                      // 1: z Periodic on this rank (write z images too)
};

OCN_DEVFN void store_images(const GridDev& g, double* f, int i, int j, int k, double val, int zwrap) {
  // offsets of the periodic images of an interior point along each direction
  long ox[3], oy[3], oz[3];
  int nx = 0, ny = 0, nz = 0;
  ox[nx++] = 0;
  if (i < g.Hx) ox[nx++] = g.Nx;
  if (i >= g.Nx - g.Hx) ox[nx++] = -g.Nx;
  oy[ny++] = 0;
  if (j < g.Hy) oy[ny++] = (long)g.Ny * g.sy;
  if (j >= g.Ny - g.Hy) oy[ny++] = -(long)g.Ny * g.sy;
  oz[nz++] = 0;
  if (zwrap) {
    if (k < g.Hz) oz[nz++] = (long)g.Nz * g.sz;
    if (k >= g.Nz - g.Hz) oz[nz++] = -(long)g.Nz * g.sz;
  }
  const long c = i + j * g.sy + k * g.sz;
  for (int c3 = 0; c3 < nz; ++c3)
    for (int c2 = 0; c2 < ny; ++c2)
      for (int c1 = 0; c1 < nx; ++c1) f[c + ox[c1] + oy[c2] + oz[c3]] = val;
}

__global__ void k_project(GridDev g, ProjArgs a) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long c = i + j * g.sy + k * g.sz;
  const long Nx = g.Nx, Ny = g.Ny;
  const long pc = i + Nx * (j + Ny * k);
  const long pw_ = (i == 0) ? pc + Nx - 1 : pc - 1;
  const long ps_ = (j == 0) ? pc + Nx * (Ny - 1) : pc - Nx;
  // below: z wrap on a single rank; on a slab the level below the first one comes from the exchanged halo of p
  double p0 = a.phi[pc];
  double pb;
  if (k > 0) pb = a.phi[pc - Nx * Ny];
  else if (a.zwrap) pb = a.phi[pc + Nx * Ny * (g.Nz - 1)];
  else pb = a.phi_below[i + Nx * j];
  double un = a.us[c] - (p0 - a.phi[pw_]) * g.rdx * a.dt;
  double vn = a.vs[c] - (p0 - a.phi[ps_]) * g.rdy * a.dt;
  double wn = a.ws[c] - (p0 - pb) / g.dz * a.dt;
  store_images(g, a.u, i, j, k, un, a.zwrap);
  store_images(g, a.v, i, j, k, vn, a.zwrap);
  store_images(g, a.w, i, j, k, wn, a.zwrap);
  store_images(g, a.p, i, j, k, p0, a.zwrap);
}

// ---- passive tracers on the fast path: calculate_Gc! + time-stepper update in one pass ---------------------------
// (calculate_nonhydrostatic_tendencies.jl:177-180, tracer_advection_operators.jl:31-35, quasi_adams_bashforth_2.jl:158-166).
// The updated tracer goes to a second buffer (neighbours still read the old one) together with its periodic
// images; the two buffers are then swapped.  One thread per (i, j) column of a z-chunk: the bottom flux of the next
// level is this level's top flux (register carry), z stencils live in a 6-deep register window; x / y face fluxes
// are evaluated on both faces from cached loads (a tracer adds 5 reconstructions per cell).
struct TracerArgs {
  const double *u, *v, *w, *c, *gm;   // interior-origin pointers
  double *gn, *cnew;
  double dt, cn, cm;
  int use_m, KZ, zwrap;
  double kappa;                        // ScalarDiffusivity kappa of this tracer (0: none): -kappa dc/dn rides in every face flux
};

template <int ADV>
__global__ void __launch_bounds__(256) k_tracer_step(GridDev g, TracerArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int k0 = blockIdx.z * a.KZ;
  const int k1 = (k0 + a.KZ < g.Nz) ? k0 + a.KZ : g.Nz;
  const long sy = g.sy, sz = g.sz;
  const double rdx = g.rdx, rdy = g.rdy, rdz = 1.0 / g.dz;
  long c = i + j * sy + (long)k0 * sz;
  double zc[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) zc[q] = a.c[c + (q - 3) * sz];
  auto fz = [&](long p) {   // flux through the bottom face of the cell at p (window centred there)
    const double wt = a.w[p];
    const bool pos = wt > 0.0;
    return wt * recon5<ADV>(pos ? zc[0] : zc[5], pos ? zc[1] : zc[4], pos ? zc[2] : zc[3], pos ? zc[3] : zc[2],
                            pos ? zc[4] : zc[1], pos) - a.kappa * (zc[3] - zc[2]) * rdz;
  };
  double fb = fz(c);
  for (int k = k0; k < k1; ++k, c += sz) {
    // west / east and south / north fluxes (advecting velocity un-interpolated: upwind_biased_advective_fluxes.jl:103-119)
    const double uw = a.u[c], ue = a.u[c + 1], vs_ = a.v[c], vn = a.v[c + sy];
    const double cc = zc[3];
    // div(kappa grad c) with constant kappa (closure_kernel_operators.jl:43-48) as diffusive face fluxes
    const double fxw = uw * recon_mem<ADV>(a.c + c, 1, uw) - a.kappa * (cc - a.c[c - 1]) * rdx;
    const double fxe = ue * recon_mem<ADV>(a.c + c + 1, 1, ue) - a.kappa * (a.c[c + 1] - cc) * rdx;
    const double fys = vs_ * recon_mem<ADV>(a.c + c, sy, vs_) - a.kappa * (cc - a.c[c - sy]) * rdy;
    const double fyn = vn * recon_mem<ADV>(a.c + c + sy, sy, vn) - a.kappa * (a.c[c + sy] - cc) * rdy;
    // advance the window to level k+1 and take its bottom flux = this level's top flux
#pragma unroll
    for (int q = 0; q < 5; ++q) zc[q] = zc[q + 1];
    zc[5] = a.c[c + 3 * sz];
    const double ft = fz(c + sz);
    const double G = -((fxe - fxw) * rdx + (fyn - fys) * rdy + (ft - fb) * rdz);
    fb = ft;
    a.gn[c] = G;
    const double inc = a.use_m ? a.dt * (a.cn * G + a.cm * a.gm[c]) : a.dt * a.cn * G;
    store_images(g, a.cnew, i, j, k, cc + inc, a.zwrap);
  }
}

// ---- tiled tracer kernel: three reconstructions per cell instead of five / six --------------------------------------
// Same scheme as k_tend_step3 for one Center field: a workgroup owns complete x rows of BY-1 output rows (+ ghost row),
// stages the level's slab of c in LDS, every thread forms the WEST / SOUTH / BOTTOM face fluxes of its cell (advecting
// velocity un-interpolated, upwind_biased_advective_fluxes.jl:103-128), EAST / NORTH come from the neighbours through LDS,
// TOP is the next level's BOTTOM.  kappa != 0: ScalarDiffusivity as -kappa dc/dn in every face flux.  REST: G^n arrives
// with the non-advective terms (variable diffusivity, boundary fluxes) from the general kernels.  ZB and the runtime
// wall flags g.yb (REST only) select the 2nd-order fallbacks of the boundary buffer.  IMG: write the periodic images of
// the updated tracer (all-in-one periodic path); otherwise the caller fills halos.
struct Tracer3Args {
  const double *u, *v, *w, *c, *gm;   // PARENT base pointers
  const double* kap;                  // KV: the tracer's eddy diffusivity kappa_e (Center field, halos filled)
  double *gn, *cnew;
  unsigned org;
  double dt, cn, cm, kappa;
  int use_m, ntiles, zwrap;
  int rest_shell;   // REST: 1 = only the first and last level of G^n hold anything (boundary fluxes of a Bounded z; no walls in
                    // x / y): the other levels are neither zeroed by the caller nor read here
};

// KV: variable diffusivity (AnisotropicMinimumDissipation): -kappa_face dc/dn with kappa_face the two-point average of kappa_e
// across the face (closure_kernel_operators.jl:43-48, 82-90) rides in the three face fluxes; kappa_e's slab sits in LDS next
// to the tracer's, its value one level down in a register.
template <int ADV, int BX, int BY, bool ZB, bool REST, bool IMG, bool KV>
__global__ void __launch_bounds__(BX* BY) k_tracer_step3(GridDev g, Tracer3Args a) {
  constexpr int T = BX * BY, NR = BY + 5, SX = BX + 6;
  constexpr int NG = (NR + BY - 1) / BY;
  OCN_SHARED double slab2[(KV ? 4 : 2) * NR * SX];   // slab and flux exchange are both double buffered by level parity:
  OCN_SHARED double fx[2 * 2 * T];                   // ONE barrier per level (a level is only three reconstructions of work)
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * BX + tx;
  const int i = tx;
  const unsigned sxb = 8u, syb = (unsigned)g.sy * 8u, szb = (unsigned)g.sz * 8u;
  const double rdx = g.rdx, rdy = g.rdy;
  const int nbz = 2;
  const bool col_ok = i < g.Nx;
  const bool ghost = (ty == BY - 1);
  const int txe = (tx + 1 == g.Nx) ? 0 : tx + 1;
  const int nid_e = ty * BX + (txe < BX ? txe : tx);
  const int nid_n = (ty + 1 < BY ? ty + 1 : ty) * BX + tx;
  const bool wy = REST && g.yb != 0;
  const int nseg = gridDim.x, per = nseg / 8;
  const long seg = (nseg % 8 == 0) ? (long)(blockIdx.x % 8) * per + blockIdx.x / 8 : (long)blockIdx.x;
  const long total = (long)a.ntiles * g.Nz;
  long lo = seg * total / nseg;
  const long hi = (seg + 1) * total / nseg;
  auto in_rec = [&](bool pos, int idx, int N) {
    return pos ? (idx > nbz && idx < N + 1 - (nbz - 1)) : (idx > nbz - 1 && idx < N + 1 - nbz);
  };
  while (lo < hi) {
    const int ytile = (int)(lo / g.Nz);
    const int k0 = (int)(lo - (long)ytile * g.Nz);
    const int k1 = (k0 + (hi - lo) < g.Nz) ? (int)(k0 + (hi - lo)) : g.Nz;
    lo += k1 - k0;
    const int j0 = ytile * (BY - 1);
    const int j = j0 + ty;
    const bool row_ok = j < g.Ny;
    const bool do_y = col_ok && j <= g.Ny;
    const bool full = col_ok && row_ok && !ghost;
    const unsigned cxy = a.org + (unsigned)(col_ok ? i : 0) * sxb + (unsigned)(j <= g.Ny ? j : 0) * syb;
    const unsigned grow = a.org + (unsigned)(col_ok ? i : 0) * sxb;
    double pf[NG], pfk[KV ? NG : 1];
    auto prefetch = [&](int k) {
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        int r = ty + BY * gq;
        if (r < NR) {
          int jg = j0 - 3 + r;
          if (jg > g.Ny + 2) jg = g.Ny + 2;
          const unsigned o = grow + (unsigned)(jg + 3) * syb - 3u * syb + (unsigned)k * szb;
          pf[gq] = ldo(a.c, o);
          if (KV) pfk[KV ? gq : 0] = ldo(a.kap, o);
        }
      }
    };
    const bool img_e = tx < 3, img_w = tx >= g.Nx - 3;
    auto commit = [&](int k) {
      double* slab = slab2 + (k & 1) * NR * SX;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        int r = ty + BY * gq;
        if (r < NR && col_ok) {
          double* row = slab + r * SX;
          row[tx + 3] = pf[gq];
          if (img_e) row[tx + 3 + g.Nx] = pf[gq];
          if (img_w) row[tx + 3 - g.Nx] = pf[gq];
          if (KV) {
            double* rk = row + 2 * NR * SX;
            const double kv = pfk[KV ? gq : 0];
            rk[tx + 3] = kv;
            if (img_e) rk[tx + 3 + g.Nx] = kv;
            if (img_w) rk[tx + 3 - g.Nx] = kv;
          }
        }
      }
    };
#define CS(d, e) S[(d) * SX + (e)]
#define KS(d, e) S[2 * NR * SX + (d) * SX + (e)]
    double zc[6];
    {
      const unsigned c = cxy + (unsigned)k0 * szb;
#pragma unroll
      for (int q = 0; q < 6; ++q) zc[q] = ldo(a.c, c + (unsigned)(q - 3) * szb);
    }
    double un = ldo(a.u, cxy + (unsigned)k0 * szb), vn = ldo(a.v, cxy + (unsigned)k0 * szb), wn = ldo(a.w, cxy + (unsigned)k0 * szb);
    double own_h = 0, own_b = 0;
    double kdn = KV ? ldo(a.kap, cxy + (unsigned)k0 * szb - szb) : 0.0;   // kappa_e one level down
    prefetch(k0);
    commit(k0);
    __syncthreads();
    for (int k = k0; k <= k1; ++k) {
      const unsigned c = cxy + (unsigned)k * szb;
      const bool last = (k == k1);
      double* fxx = fx + (k & 1) * 2 * T;
      double* fxy = fxx + T;
      const double* S = slab2 + (k & 1) * NR * SX + ty * SX + tx;      // own cell at S[3 * SX + 3]
      if (!last) prefetch(k + 1);                                     // lands while the fluxes are formed
      const double uw = un, vs_ = vn, wb = wn;          // advecting velocities at the west / south / bottom faces
      if (!last) {
        un = ldo(a.u, c + szb);
        vn = ldo(a.v, c + szb);
        wn = ldo(a.w, c + szb);
      }
      double gmv = 0, rest = 0;
      if (full && k > k0) {
        if (a.use_m) gmv = ldo(a.gm, c - szb);
        if (REST && (!a.rest_shell || k - 1 == 0 || k == g.Nz)) rest = ldo(a.gn, c - szb);   // uniform over the workgroup
      }
      auto rec = [&](double m3, double m2, double m1, double c0, double c1, double c2, double ut) {
        bool pos = ut > 0.0;
        return recon5<ADV>(pos ? m3 : c2, pos ? m2 : c1, pos ? m1 : c0, pos ? c0 : m1, pos ? c1 : m2, pos);
      };
      if (!last) {
        if (full) {
          double f = uw * rec(CS(3, 0), CS(3, 1), CS(3, 2), CS(3, 3), CS(3, 4), CS(3, 5), uw);
          if (KV) f -= 0.5 * (KS(3, 2) + KS(3, 3)) * (CS(3, 3) - CS(3, 2)) * rdx;
          else if (a.kappa != 0.0) f -= a.kappa * (CS(3, 3) - CS(3, 2)) * rdx;
          fxx[tid] = f;
        }
        if (do_y) {
          double r;
          if (wy && !in_rec(vs_ > 0.0, j + 1, g.Ny)) r = 0.5 * (CS(2, 3) + CS(3, 3));
          else r = rec(CS(0, 3), CS(1, 3), CS(2, 3), CS(3, 3), CS(4, 3), CS(5, 3), vs_);
          double f = vs_ * r;
          if (KV) f -= 0.5 * (KS(2, 3) + KS(3, 3)) * (CS(3, 3) - CS(2, 3)) * rdy;
          else if (a.kappa != 0.0) f -= a.kappa * (CS(3, 3) - CS(2, 3)) * rdy;
          fxy[tid] = f;
        }
      }
      double fb = 0;
      if (full) {
        const bool pos = wb > 0.0;
        double r;
        if (ZB && !in_rec(pos, k + 1, g.Nz)) r = 0.5 * (zc[2] + zc[3]);
        else r = recon5<ADV>(pos ? zc[0] : zc[5], pos ? zc[1] : zc[4], pos ? zc[2] : zc[3], pos ? zc[3] : zc[2], pos ? zc[4] : zc[1], pos);
        fb = wb * r;
        if (KV) {
          const double kc = KS(3, 3);
          fb -= 0.5 * (kdn + kc) * (zc[3] - zc[2]) * (ZB ? g_rdzf(g, k) : g.rdz);
          kdn = kc;
        } else if (a.kappa != 0.0) fb -= a.kappa * (zc[3] - zc[2]) * (ZB ? g_rdzf(g, k) : g.rdz);
      }
      if (!last) commit(k + 1);            // the other slab buffer: last read in the previous level's flux stage
      __syncthreads();
      if (full) {
        if (k > k0) {
          const unsigned cm1 = c - szb;
          const double rzc = ZB ? g_rdzc(g, k - 1) : g.rdz;
          const double G = rest - (own_h + (fb - own_b) * rzc);
          sto(a.gn, cm1, G);
          const double inc = a.use_m ? a.dt * (a.cn * G + a.cm * gmv) : a.dt * a.cn * G;
          const double val = zc[2] + inc;
          if (IMG) store_images(g, (double*)((char*)a.cnew + a.org), i, j, k - 1, val, a.zwrap);
          else sto(a.cnew, cm1, val);
        }
        if (!last) {
          own_h = (fxx[nid_e] - fxx[tid]) * rdx + (fxy[nid_n] - fxy[tid]) * rdy;
          own_b = fb;
        }
      }
      if (!last) {
#pragma unroll
        for (int q = 0; q < 5; ++q) zc[q] = zc[q + 1];
        zc[5] = ldo(a.c, c + 3 * szb);
      }
    }
  }
#undef CS
#undef KS
}

// ---- host side -------------------------------------------------------------------------------------------------
// why (not): the reason is kept on the model for ocn_model_path()
static const char* tiled_blocker(const ocn_model* m) {
  const ocn_grid* g = m->g;
  const int adv = m->d.advection;
  if (adv != ADV_WENO_Z && adv != ADV_WENO_JS && adv != ADV_U5) return "the advection scheme is not an upwind-biased 5th-order one";
  for (int d = 0; d < 3; ++d)
    if (g->topo[d] != OCN_FLAT && (g->H[d] < 3 || g->N[d] < 2 * g->H[d])) return "a direction has fewer than 6 cells";
  if (m->u.n * sizeof(double) >= (1ull << 31) || m->w.n * sizeof(double) >= (1ull << 31))
    return "parent arrays of 2 GiB or more exceed the tiled kernels' 32-bit byte offsets";
  return nullptr;
}

bool fused_available(const ocn_model* m) {
  const ocn_grid* g = m->g;
  if (g->topo[0] != OCN_PERIODIC || g->topo[1] != OCN_PERIODIC || g->topo[2] != OCN_PERIODIC) return false;
  if (!g->z_regular) return false;
  // ScalarDiffusivity (constant nu, kappa) rides in the face fluxes of the fused kernels; other closures, Coriolis and
  // buoyancy take the general path
  if (m->d.closure == OCN_CLOSURE_AMD || m->d.coriolis_fplane || m->d.buoyancy != OCN_BUOYANCY_NONE) return false;
  if (tiled_blocker(m)) return false;
  if (getenv("OCNHIP_NO_FUSED")) return false;   // model creation only
  return true;
}

bool rest4_ok(const ocn_model* m);
void fused_describe(const ocn_model* m, char* buf, size_t n) {
  const char* why = tiled_blocker(m);
  if (m->fast_path) {
    const size_t halo_bytes = (size_t)(3 + m->nt) * m->gd.Hz * m->u.sz * sizeof(double);
    const bool ov = m->g->dist && m->gd.Nz > 2 * m->gd.Hz + 2 &&
                    (m->knob_overlap >= 0 ? m->knob_overlap != 0 : (m->ctx->nranks > 1 && halo_bytes >= ((size_t)8 << 20)));
    snprintf(buf, n, "all-in-one periodic path: k_tend4 + fused Poisson passes + k_project%s",
             !m->g->dist ? "" : ov ? "; z-slabs, halo planes travel under the next interior tendency launch"
                                   : "; z-slabs, halo exchange on the model's stream");
  }
  else if (m->bz_fast) snprintf(buf, n, "tiled advection + update (k_tend4, REST) on top of the %s other terms",
                                rest4_ok(m) ? "tiled kernel's (k_rest4)" : "general kernels'");
  else if (why) snprintf(buf, n, "general kernels: %s", why);
  else if (m->g->topo[0] == OCN_FLAT || m->g->topo[1] == OCN_FLAT) snprintf(buf, n, "general kernels: Flat x / y slices are outside the tiled kernels");
  else snprintf(buf, n, "general kernels (tiled path disabled or not applicable to this topology / decomposition)");
}

static int fused_cu_count(const ocn_model* m) {
  static int ncu = 0;
  if (!ncu) {
#ifndef OCN_HOST_EMU
    hipDeviceProp_t prop;
    ncu = (hipGetDeviceProperties(&prop, m->ctx->device) == hipSuccess) ? prop.multiProcessorCount : 256;
#else
    ncu = 8;
#endif
  }
  return ncu;
}

// Workgroup shape and work decomposition shared by all variants: complete rows up to 256 columns (64 x 8, 128 x 8,
// 256 x 4 threads), x-tiles of 192 x 5 threads beyond that; one equal segment of the (tile, level) space per CU.
struct FusedShape {
  int bx, by;
  bool wide, small;
  int dma;       // slab staged by global_load_lds: 1 row by row, 2 one chunk per field (rows with the slab's pitch); 0: registers
  dim3 blk, grd;
};

// Tuning / test knobs of the tiled kernels: read ONCE, when the model is created, never on a launch path.
//   OCNHIP_FUSED_XT=1     force the x-tiled variant on small grids (tests)
//   OCNHIP_NO_LDS_DMA=1   stage the slab through registers even where the LDS-DMA path is legal (tests); =2: never the
//                         one-chunk form
//   OCNHIP_NO_TRACER3=1   column tracer kernel instead of the tiled one (tests)
//   OCNHIP_PRIO=code      wave-priority code (see prio_start; 0 = hardware default)
//   OCNHIP_NO_GRAPH=1     general path: issue every launch of a step from the host instead of replaying a hipGraph
void fused_read_knobs(ocn_model* m) {
  auto env = [](const char* n, int def) { const char* e = getenv(n); return e ? atoi(e) : def; };
  m->knob_fused_xt = env("OCNHIP_FUSED_XT", 0);
  m->knob_no_dma = env("OCNHIP_NO_LDS_DMA", 0);
  m->knob_no_tracer3 = env("OCNHIP_NO_TRACER3", 0);
  m->knob_xfft_team = env("OCNHIP_XFFT_TEAM", 0);
  m->knob_overlap = env("OCNHIP_OVERLAP", -1);   // -1: on when there is more than one rank
  m->knob_overlap_cus = env("OCNHIP_OVERLAP_CUS", 16);
  m->knob_graph = env("OCNHIP_NO_GRAPH", 0) ? 0 : 1;   // whole-step hipGraphs of the general path (api.hip step_graphed)
  // default 0x20FF: every row at priority 3 until the middle of its flux stage, then only the last output row keeps 2
  // (256^3: 0.575 ms against 0.590 with the hardware's age order; ten codes tried, all within 0.575 - 0.613)
  { const char* e = getenv("OCNHIP_PRIO"); m->knob_prio = e ? (int)strtol(e, nullptr, 0) : 0x20FF; }
}

static FusedShape fused_shape(const ocn_model* m, FusedArgs& a) {
  const GridDev& gd = m->gd;
  FusedShape f;
  f.small = m->knob_fused_xt == 1 && gd.Nx <= 57 * 4;   // test shape: 64 x 4 threads, up to 56 output columns, >= 2 tiles
#ifdef OCN_HOST_EMU
  if (gd.xb && gd.Nx <= 57 * 4) f.small = true;   // the emulation spawns one OS thread per GPU thread: keep workgroups small
#endif
  f.wide = gd.Nx > 256 || f.small || gd.xb;          // walls in x: no periodic wrap inside LDS -> x-tiles
  f.bx = f.small ? 64 : f.wide ? 192 : gd.Nx <= 64 ? 64 : gd.Nx <= 128 ? 128 : 256;
  f.by = f.small ? 4 : f.wide ? 5 : f.bx == 256 ? 4 : 8;
#ifdef OCN_HOST_EMU
  // the emulation runs one OS thread per GPU thread and every barrier wakes all of them: 16 x 4 workgroups for the tiny
  // test grids (same kernel source, another template shape)
  if ((f.small && gd.Nx <= 36) || (!f.wide && gd.Nx <= 16)) {
    f.bx = 16;
    f.by = 4;
  }
#endif
  a.BYo = f.by - 1;
  a.ntiles = (gd.Ny + f.by - 2) / (f.by - 1);
  a.ntx = 1;
  a.BXo = 0;
  if (f.wide) {
    const int cap = (f.bx - 7) & ~1;              // outputs + ghost column + 6 halo columns <= bx; even, so that every
    a.ntx = (gd.Nx + cap - 1) / cap;              // tile's slab rows start on a 16-byte boundary (LDS-DMA)
    if (f.small && a.ntx < 2) a.ntx = 2;
    a.BXo = (gd.Nx + a.ntx - 1) / a.ntx;
    a.BXo += a.BXo & 1;
    a.ntx = (gd.Nx + a.BXo - 1) / a.BXo;
    a.ntiles *= a.ntx;
  }
  int nseg = fused_cu_count(m);                   // one workgroup is resident per CU
  const long total = (long)a.ntiles * ((a.zhi - a.zlo) + (a.zhi2 - a.zlo2));
  a.gran = 1;
  if (a.zhi2 > a.zlo2) {
    // the boundary levels of a slab (their z halos arrived late), two short runs per tile: whole runs per workgroup -- a
    // segment that ends inside a run pays the start-up of the march (register windows, first slab) a second time
    a.gran = a.zhi - a.zlo;
    const long runs = total / a.gran;
    if (nseg > runs) nseg = (int)runs;
  } else {
    // the interior levels of a slab run while the halo planes travel: this kernel keeps one workgroup resident on every CU
    // for its whole duration (LDS, 4 x 128 VGPRs per SIMD), so the communication kernels would find no CU to start on --
    // or, once they hold a few, the workgroups that lost theirs would run as a second round.  Leave them some.
    if (a.zhi - a.zlo < gd.Nz && nseg > 4 * m->knob_overlap_cus) nseg -= m->knob_overlap_cus;
    if (nseg > total / 4) nseg = (int)(total / 4 > 8 ? total / 4 : 8);
    nseg = ((nseg + 7) / 8) * 8;                  // XCD-aware remap inside the kernel wants a multiple of 8
  }
#ifdef OCN_HOST_EMU
  nseg = total >= 3 ? 3 : 1;                      // one OS thread per emulated GPU thread: few workgroups, still several segments
#endif
  f.blk = dim3(f.bx, f.by, 1);
  f.grd = dim3(nseg, 1, 1);
  // 16-byte pieces: a slab row starts at parent column Hx - 3 + i0 (i0 = x-tile origin, even), rows are sy doubles apart,
  // planes sz doubles; the arrays come from hipMalloc (256-byte aligned)
  f.dma = (m->knob_no_dma != 1 && gd.Hx == 3 && gd.Nx % 2 == 0 && gd.sy % 2 == 0 && gd.sz % 2 == 0) ? 1 : 0;
  if (f.dma && !f.wide && gd.Nx == f.bx && gd.sy == gd.Nx + 6 && m->knob_no_dma != 2) f.dma = 2;
  return f;
}

// launch one of the instantiations: VISCV / ZBV / RESTV are compile-time constants at the call site
#define FUSED_T4(ADVV, BXV, BYV, XTV, VISCV, ZBV, RESTV)                                                                   \
  { if (f.dma == 2 && !XTV) ocn_launch_sync(k_tend4<ADVV, BXV, BYV, XTV, XTV ? 1 : 2, VISCV, ZBV, RESTV>, f.grd, f.blk, s, m->gd, a); \
    else if (f.dma) ocn_launch_sync(k_tend4<ADVV, BXV, BYV, XTV, 1, VISCV, ZBV, RESTV>, f.grd, f.blk, s, m->gd, a);       \
    else ocn_launch_sync(k_tend4<ADVV, BXV, BYV, XTV, 0, VISCV, ZBV, RESTV>, f.grd, f.blk, s, m->gd, a); }
#ifdef OCN_HOST_EMU
#define FUSED_EMU16(ADVV, VISCV, ZBV, RESTV)                                                                               \
  if (f.bx == 16 && f.small) FUSED_T4(ADVV, 16, 4, true, VISCV, ZBV, RESTV)                                                \
  else if (f.bx == 16) FUSED_T4(ADVV, 16, 4, false, VISCV, ZBV, RESTV)                                                     \
  else
#else
#define FUSED_EMU16(ADVV, VISCV, ZBV, RESTV)
#endif
#define FUSED_LAUNCH(ADVV, VISCV, ZBV, RESTV)                                                                              \
  FUSED_EMU16(ADVV, VISCV, ZBV, RESTV)                                                                                     \
  if (f.small) FUSED_T4(ADVV, 64, 4, true, VISCV, ZBV, RESTV)                                                              \
  else if (f.wide) FUSED_T4(ADVV, 192, 5, true, VISCV, ZBV, RESTV)                                                         \
  else if (f.bx == 256) FUSED_T4(ADVV, 256, 4, false, VISCV, ZBV, RESTV)                                                   \
  else if (f.bx == 128) FUSED_T4(ADVV, 128, 8, false, VISCV, ZBV, RESTV)                                                   \
  else FUSED_T4(ADVV, 64, 8, false, VISCV, ZBV, RESTV)
#define FUSED_BY_SCHEME(VISCV, ZBV, RESTV)                            \
  switch (m->d.advection) {                                           \
    case ADV_WENO_Z: { FUSED_LAUNCH(ADV_WENO_Z, VISCV, ZBV, RESTV) } break;   \
    case ADV_WENO_JS: { FUSED_LAUNCH(ADV_WENO_JS, VISCV, ZBV, RESTV) } break; \
    default: { FUSED_LAUNCH(ADV_U5, VISCV, ZBV, RESTV) } break;               \
  }

static void fused_fill_args(ocn_model* m, FusedArgs& a, double dt, double cn, double cm, int use_m) {
  a.u = m->u.d; a.v = m->v.d; a.w = m->w.d;
  a.gmu = m->Gm[0].d; a.gmv = m->Gm[1].d; a.gmw = m->Gm[2].d;
  a.gnu = m->Gn[0].d; a.gnv = m->Gn[1].d; a.gnw = m->Gn[2].d;
  a.us = m->us.d; a.vs = m->vs.d; a.ws = m->ws.d;
  a.org = (unsigned)((m->u.Hx + m->u.Hy * m->u.sy + m->u.Hz * m->u.sz) * sizeof(double));
  a.dt = dt; a.cn = cn; a.cm = cm; a.use_m = use_m;
  a.prio = m->knob_prio;
  a.nu = 0.0;
  a.zlo = 0;
  a.zhi = m->gd.Nz;
  a.zlo2 = a.zhi2 = 0;
  a.gran = 1;
}

// all-in-one path: triply periodic, no closure or ScalarDiffusivity
void launch_fused_tend_step(ocn_model* m, double dt, double cn, double cm, int use_m, int zlo, int zhi, int zlo2, int zhi2) {
  ProfScope ps(m->ctx, "fused_tendency_step");
  FusedArgs a;
  fused_fill_args(m, a, dt, cn, cm, use_m);
  a.zlo = zlo;
  a.zhi = zhi < 0 ? m->gd.Nz : zhi;
  a.zlo2 = zlo2;
  a.zhi2 = zhi2;
  if (a.zhi <= a.zlo) return;
  a.nu = m->d.closure == OCN_CLOSURE_SCALAR ? m->d.nu : 0.0;
  const FusedShape f = fused_shape(m, a);
  hipStream_t s = m->ctx->stream;
  if (a.nu != 0.0) {
    FUSED_BY_SCHEME(true, false, false)
  } else {
    FUSED_BY_SCHEME(false, false, false)
  }
}

// ---- Bounded z: advection + time-stepper update of u, v, w on top of the general kernels' other terms ----------------
bool fused_bz_available(const ocn_model* m) {
  const ocn_grid* g = m->g;
  if (g->topo[0] == OCN_FLAT || g->topo[1] == OCN_FLAT) return false;      // walls in x / y are fine (runtime flags of the REST variants)
  if (g->dist_y && (g->topo[0] != OCN_PERIODIC || g->topo[1] != OCN_PERIODIC)) return false;
  if (g->topo[2] == OCN_FLAT || (g->topo[2] == OCN_PERIODIC && (!g->z_regular || g->dist))) return false;
  if (tiled_blocker(m)) return false;
  if (getenv("OCNHIP_NO_FUSED") || getenv("OCNHIP_NO_FUSED_BZ")) return false;   // model creation only
  return true;
}

// G^n(u, v, w) must hold the non-advective terms (k_tend_uvw<ADV_NONE> + boundary fluxes); on return it holds the full
// tendencies and us / vs / ws the stepped velocities (interior cells; the caller swaps buffers and fills halos).
void launch_fused_bz(ocn_model* m, double dt, double cn, double cm, int use_m) {
  ProfScope ps(m->ctx, "fused_tendency_step");
  FusedArgs a;
  fused_fill_args(m, a, dt, cn, cm, use_m);
  const FusedShape f = fused_shape(m, a);
  hipStream_t s = m->ctx->stream;
  if (m->g->topo[2] == OCN_BOUNDED) {
    FUSED_BY_SCHEME(false, true, true)
  } else {
    FUSED_BY_SCHEME(false, false, true)
  }
}
#undef FUSED_BY_SCHEME
#undef FUSED_LAUNCH

// the non-advective momentum terms of the tiled "rest" models by k_rest4 (complete rows, LDS-DMA layout); false: not applicable
bool rest4_ok(const ocn_model* m) {
  const GridDev& gd = m->gd;
  if (!m->bz_fast || gd.Nx > 256 || gd.xb || m->knob_no_dma) return false;
  return gd.Hx == 3 && gd.Nx % 2 == 0 && gd.sy % 2 == 0 && gd.sz % 2 == 0 && gd.sy == gd.Nx + 6;
}

bool launch_rest4(ocn_model* m) {
  const GridDev& gd = m->gd;
  if (!rest4_ok(m)) return false;
  ProfScope ps(m->ctx, "rest_terms");
  RestArgs a;
  a.u = m->u.d; a.v = m->v.d; a.w = m->w.d;
  a.nue = m->nu_e.present ? m->nu_e.d : nullptr;
  a.pH = (m->pHY.present) ? m->pHY.d : nullptr;
  a.gu = m->Gn[0].d; a.gv = m->Gn[1].d; a.gw = m->Gn[2].d;
  a.org = (unsigned)((m->u.Hx + m->u.Hy * m->u.sy + m->u.Hz * m->u.sz) * sizeof(double));
  a.nu = m->d.nu;
  a.f = m->d.f;
  a.closure = m->d.closure;
  a.coriolis = m->d.coriolis_fplane;
  int bx = gd.Nx <= 64 ? 64 : gd.Nx <= 128 ? 128 : 256;
  int by = bx == 256 ? 4 : 8;
#ifdef OCN_HOST_EMU
  if (gd.Nx <= 16) {
    bx = 16;
    by = 4;
  }
#endif
  a.ntiles = (gd.Ny + by - 2) / (by - 1);
  int nseg = fused_cu_count(m);
  const long total = (long)a.ntiles * gd.Nz;
  if (nseg > total / 4) nseg = (int)(total / 4 > 8 ? total / 4 : 8);
  nseg = ((nseg + 7) / 8) * 8;
#ifdef OCN_HOST_EMU
  nseg = total >= 3 ? 3 : 1;
#endif
  const dim3 blk(bx, by, 1), grd(nseg, 1, 1);
  hipStream_t s = m->ctx->stream;
  const bool zb = m->g->topo[2] == OCN_BOUNDED;
#define REST4(BXV, BYV) { if (zb) ocn_launch_sync(k_rest4<BXV, BYV, true>, grd, blk, s, gd, a); else ocn_launch_sync(k_rest4<BXV, BYV, false>, grd, blk, s, gd, a); }
#ifdef OCN_HOST_EMU
  if (bx == 16) REST4(16, 4) else
#endif
  if (bx == 256) REST4(256, 4) else if (bx == 128) REST4(128, 8) else REST4(64, 8)
#undef REST4
  return true;
}

void launch_rhs_wrap(ocn_model* m, double dt, double* rhs) {
  ProfScope ps(m->ctx, "rhs");
  const GridDev& g = m->gd;
  dim3 b(256, 1, 1), gr((g.Nx + b.x - 1) / b.x, (g.Ny + b.y - 1) / b.y, g.Nz);
  ocn_launch(k_rhs_wrap, gr, b, m->ctx->stream, g, (const double*)m->us.interior(), (const double*)m->vs.interior(),
             (const double*)m->ws.interior(), 1.0 / dt, m->g->dist ? 0 : 1, rhs);
}

void launch_project(ocn_model* m, double dt, const double* phi) {
  ProfScope ps(m->ctx, "pcorrect");
  const GridDev& g = m->gd;
  ProjArgs a;
  a.phi = phi;
  a.us = m->us.interior(); a.vs = m->vs.interior(); a.ws = m->ws.interior();
  a.u = m->u.interior(); a.v = m->v.interior(); a.w = m->w.interior(); a.p = m->pNHS.interior();
  a.dt = dt;
  a.zwrap = m->g->dist ? 0 : 1;
  a.phi_below = poisson_local_phi_below(m) ? poisson_phi_below(m) : m->phi_below;
  dim3 b(256, 1, 1), gr((g.Nx + b.x - 1) / b.x, (g.Ny + b.y - 1) / b.y, g.Nz);
  ocn_launch(k_project, gr, b, m->ctx->stream, g, a);
}

// ---- slab runs: the two one-plane exchanges the fused path needs besides the z-halo exchange -----------------
// (a) w* of the first level above the slab (for div U* at the top level)
int fused_exchange_ws(ocn_model* m) {
  ocn_ctx* c = m->ctx;
  ProfScope ps(c, "halo_exchange");
  const int R = c->nranks, r = c->rank, up = (r + 1) % R, dn = (r + R - 1) % R;
  Field& f = m->ws;
  const size_t plane = (size_t)f.sz * sizeof(double);
  char* base = (char*)f.d;
  std::vector<CommOp> s{{base + plane * (size_t)f.Hz, plane, dn, 7}};
  std::vector<CommOp> q{{base + plane * (size_t)(m->gd.Nz + f.Hz), plane, up, 7}};
  return comm_exchange(c, s, q);
}
// (b) pressure of the last level below the slab (for dp/dz at the first level)
int fused_exchange_phi(ocn_model* m, const double* phi) {
  ocn_ctx* c = m->ctx;
  ProfScope ps(c, "halo_exchange");
  const int R = c->nranks, r = c->rank, up = (r + 1) % R, dn = (r + R - 1) % R;
  const size_t plane = (size_t)m->gd.Nx * m->gd.Ny * sizeof(double);
  std::vector<CommOp> s{{(char*)phi + plane * (size_t)(m->gd.Nz - 1), plane, up, 8}};
  std::vector<CommOp> q{{(void*)m->phi_below, plane, dn, 8}};
  return comm_exchange(c, s, q);
}

// G^n of a tracer arrives holding only boundary fluxes when the tiled kernel carries advection and closure; without walls in
// x / y those live in the first and last level, so only these two planes are cleared and read (kernels.hip launch_tendencies)
bool tracer_rest_shell(const ocn_model* m) { return !m->gd.xb && !m->gd.yb; }

bool fused_tracer3_ok(const ocn_model* m) {
  return m->gd.Nx <= 256 && !m->gd.xb && !m->knob_no_tracer3;
}

// tiled tracer kernel for every tracer; rest: G^n(tracers) holds the non-advective terms and halos are filled by the caller
void launch_tracer3(ocn_model* m, double dt, double cn, double cm, int use_m, bool rest) {
  ProfScope ps(m->ctx, "fused_tracer_step");
  const GridDev& gd = m->gd;
  int bx = gd.Nx <= 64 ? 64 : gd.Nx <= 128 ? 128 : 256;
  int by = bx == 256 ? 4 : 8;
#ifdef OCN_HOST_EMU
  if (gd.Nx <= 16) {
    bx = 16;
    by = 4;
  }
#endif
  Tracer3Args a;
  a.u = m->u.d; a.v = m->v.d; a.w = m->w.d;
  a.org = (unsigned)((m->u.Hx + m->u.Hy * m->u.sy + m->u.Hz * m->u.sz) * sizeof(double));
  a.dt = dt; a.cn = cn; a.cm = cm; a.use_m = use_m;
  a.ntiles = (gd.Ny + by - 2) / (by - 1);
  a.zwrap = (m->g->dist || m->g->topo[2] != OCN_PERIODIC) ? 0 : 1;
  a.rest_shell = (rest && tracer_rest_shell(m)) ? 1 : 0;
  int nseg = fused_cu_count(m);
  const long total = (long)a.ntiles * gd.Nz;
  if (nseg > total / 4) nseg = (int)(total / 4 > 8 ? total / 4 : 8);
  nseg = ((nseg + 7) / 8) * 8;
#ifdef OCN_HOST_EMU
  nseg = total >= 3 ? 3 : 1;
#endif
  const dim3 blk(bx, by, 1), grd(nseg, 1, 1);
  hipStream_t s = m->ctx->stream;
  const bool zb = m->g->topo[2] == OCN_BOUNDED;
  for (int t = 0; t < m->nt; ++t) {
    a.c = m->tr[t].d;
    a.gm = m->Gm[3 + t].d;
    a.gn = m->Gn[3 + t].d;
    a.cnew = m->trs[t].d;
    // the closure's tracer flux rides in the face fluxes: constant kappa (ScalarDiffusivity) or the eddy diffusivity field
    const bool kv = m->d.closure == OCN_CLOSURE_AMD;
    a.kappa = m->d.closure == OCN_CLOSURE_SCALAR ? m->d.kappa[t] : 0.0;
    a.kap = kv ? m->kappa_e[t].d : nullptr;
#ifdef OCN_HOST_EMU
#define TR3_EMU16(ADVV, ZBV, RESTV, IMGV) \
    if (bx == 16) { if (kv) ocn_launch_sync(k_tracer_step3<ADVV, 16, 4, ZBV, RESTV, IMGV, true>, grd, blk, s, m->gd, a);  \
                    else ocn_launch_sync(k_tracer_step3<ADVV, 16, 4, ZBV, RESTV, IMGV, false>, grd, blk, s, m->gd, a); } else
#else
#define TR3_EMU16(ADVV, ZBV, RESTV, IMGV)
#endif
#define TR3_BX(ADVV, ZBV, RESTV, IMGV, KVV)                                                                      \
    if (bx == 256) ocn_launch_sync(k_tracer_step3<ADVV, 256, 4, ZBV, RESTV, IMGV, KVV>, grd, blk, s, m->gd, a);     \
    else if (bx == 128) ocn_launch_sync(k_tracer_step3<ADVV, 128, 8, ZBV, RESTV, IMGV, KVV>, grd, blk, s, m->gd, a); \
    else ocn_launch_sync(k_tracer_step3<ADVV, 64, 8, ZBV, RESTV, IMGV, KVV>, grd, blk, s, m->gd, a);
#define TR3_SHAPES(ADVV, ZBV, RESTV, IMGV)                                                                 \
    TR3_EMU16(ADVV, ZBV, RESTV, IMGV)                                                                      \
    if (kv) { TR3_BX(ADVV, ZBV, RESTV, IMGV, true) } else { TR3_BX(ADVV, ZBV, RESTV, IMGV, false) }
#define TR3_MODE(ADVV)                                 \
    if (!rest) { TR3_SHAPES(ADVV, false, false, true) } \
    else if (zb) { TR3_SHAPES(ADVV, true, true, false) } \
    else { TR3_SHAPES(ADVV, false, true, false) }
    switch (m->d.advection) {
      case ADV_WENO_Z: TR3_MODE(ADV_WENO_Z) break;
      case ADV_WENO_JS: TR3_MODE(ADV_WENO_JS) break;
      default: TR3_MODE(ADV_U5) break;
    }
#undef TR3_MODE
#undef TR3_SHAPES
#undef TR3_BX
  }
  for (int t = 0; t < m->nt; ++t) std::swap(m->tr[t].d, m->trs[t].d);
}

void launch_tracer_steps(ocn_model* m, double dt, double cn, double cm, int use_m) {
  if (m->nt == 0) return;
  if (fused_tracer3_ok(m)) {
    launch_tracer3(m, dt, cn, cm, use_m, false);
    return;
  }
  ProfScope ps(m->ctx, "fused_tracer_step");
  const GridDev& g = m->gd;
  for (int t = 0; t < m->nt; ++t) {
    TracerArgs a;
    a.u = m->u.interior(); a.v = m->v.interior(); a.w = m->w.interior();
    a.c = m->tr[t].interior();
    a.gm = m->Gm[3 + t].interior();
    a.gn = m->Gn[3 + t].interior();
    a.cnew = m->trs[t].interior();
    a.dt = dt; a.cn = cn; a.cm = cm; a.use_m = use_m;
    a.KZ = 32;
    a.zwrap = m->g->dist ? 0 : 1;
    a.kappa = m->d.closure == OCN_CLOSURE_SCALAR ? m->d.kappa[t] : 0.0;
    dim3 b(64, 4, 1), gr((g.Nx + 63) / 64, (g.Ny + 3) / 4, (g.Nz + a.KZ - 1) / a.KZ);
    switch (m->d.advection) {
      case ADV_WENO_Z: ocn_launch(k_tracer_step<ADV_WENO_Z>, gr, b, m->ctx->stream, g, a); break;
      case ADV_WENO_JS: ocn_launch(k_tracer_step<ADV_WENO_JS>, gr, b, m->ctx->stream, g, a); break;
      default: ocn_launch(k_tracer_step<ADV_U5>, gr, b, m->ctx->stream, g, a); break;
    }
  }
  // the freshly written buffers become the tracers (pointer swap; boundary conditions stay with the field)
  for (int t = 0; t < m->nt; ++t) std::swap(m->tr[t].d, m->trs[t].d);
}
