// stencils.h -- device-side staggered-grid stencils shared by every kernel.
//
// Restated from the reference (paths relative to /root/reference/src):
//   Advection/weno_fifth_order.jl:12-19,266-272,311-317,380-403,518-524   (WENO5, Z and JS weights)
//   Advection/upwind_biased_fifth_order.jl:24-46                           (U5)
//   Advection/centered_fourth_order.jl:17-33                               (4th-order symmetric)
//   Advection/upwind_biased_advective_fluxes.jl:10                         (upwind_biased_product)
//   Advection/topologically_conditional_interpolation.jl:19-21             (boundary buffers)
//
// Memory model: every field is the reference's parent array (x fastest).  Kernels receive a pointer
// to the *first interior cell* so that p[i + j*sy + k*sz] with i,j,k possibly negative reaches halos.
#pragma once
#include "compat.h"

enum { ADV_NONE = 0, ADV_C2 = 1, ADV_C4 = 2, ADV_U5 = 3, ADV_WENO_Z = 4, ADV_WENO_JS = 5, ADV_U1 = 6, ADV_U3 = 7 };

struct GridDev {
  int Nx, Ny, Nz, Hx, Hy, Hz;
  long sy, sz;              // strides of the parent array (sx = 1)
  double dx, dy, dz;        // dz: regular spacing (1 for Flat z)
  double rdx, rdy;
  const double* dzc;        // stretched z: spacing at centres, entry [k + Hz]   (k = 0-based centre index)
  const double* dzf;        //              spacing at faces,   entry [k + Hz + 1] (k = 0-based face index)
  const double *rdzc, *rdzf;  // their reciprocals (same entries), so kernels never divide by a per-level constant
  double rdz;               // 1 / dz
  int xb, yb, zb;           // Bounded directions
  int zflat;                // z Flat (Flat x / y are stored with broadcast halos and need no flag)
  int nb;                   // boundary buffer of the advection scheme
};

OCN_DEVFN double g_dzc(const GridDev& g, int k) { return g.dzc ? g.dzc[k + g.Hz] : g.dz; }
OCN_DEVFN double g_dzf(const GridDev& g, int k) { return g.dzf ? g.dzf[k + g.Hz + 1] : g.dz; }
OCN_DEVFN double g_rdzc(const GridDev& g, int k) { return g.rdzc ? g.rdzc[k + g.Hz] : g.rdz; }
OCN_DEVFN double g_rdzf(const GridDev& g, int k) { return g.rdzf ? g.rdzf[k + g.Hz + 1] : g.rdz; }

// ---- second / fourth order symmetric interpolation -------------------------------------------------
// value midway between p[0] and p[s]
OCN_DEVFN double sym2(const double* p, long s) { return 0.5 * (p[0] + p[s]); }
// I3 of centered_fourth_order.jl:17-24:  f - delta(delta f)/6
OCN_DEVFN double i3(const double* p, long s) { return p[0] - ((p[s] - p[0]) - (p[0] - p[-s])) * (1.0 / 6.0); }
// 0.5 (I3(p) + I3(p+s)) == (7 (p[0] + p[s]) - (p[-s] + p[2s])) / 12
OCN_DEVFN double sym4_v(double m2, double m1, double c0, double c1) {
  return fma(7.0 / 12.0, m1 + c0, (-1.0 / 12.0) * (m2 + c1));
}
OCN_DEVFN double sym4(const double* p, long s) { return sym4_v(p[-s], p[0], p[s], p[2 * s]); }

// ---- one-thread-per-cell kernels: (i, j) of this thread, the blocks of a level handed out XCD-major ---------------------------
// Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2) in the order of their linear index, so blocks
// that are neighbours in y -- they share the stencil rows between them -- sit on different L2s and every shared row is
// fetched from HBM twice (k_amd_all: 1.38x its algorithmic reads; the y transform pass: 65 -> 48-57 us with this order).
// Block b of a level takes the (bx, by) position (b % 8) * (n / 8) + b / 8: an XCD then owns a band of consecutive rows.
OCN_DEVFN void ocn_cell_ij(int& i, int& j) {
  const unsigned gx = gridDim.x, n = gx * gridDim.y;
  unsigned bl = blockIdx.x + gx * blockIdx.y;
  if (n % 8 == 0) bl = (bl % 8) * (n / 8) + bl / 8;
  i = (int)((bl % gx) * blockDim.x + threadIdx.x);
  j = (int)((bl / gx) * blockDim.y + threadIdx.y);
}

// ---- fast reciprocal for the WENO weights: one hardware rcp + 1 Newton step ------------------------
// Measured on MI355X (tools/micro_checks.hip, 2^20 random arguments over 80 binades): v_rcp_f64 alone 4.6e-8 relative
// error, one Newton step 2.2e-15, two steps 1.1e-16.  The reciprocal multiplies the weighted sum of candidate
// DIFFERENCES (reconstruction minus the upwind cell value), so 2e-15 of that is far below the 1e-12 parity bound.
OCN_DEVFN double fast_rcp(double x) {
#ifndef OCN_HOST_EMU
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
#if defined(OCN_RCP_NR) && OCN_RCP_NR > 1
  r = fma(fma(-x, r, 1.0), r, r);
#endif
  return r;
#else
  return 1.0 / x;
#endif
}

// ---- upwind-biased reconstruction at the face between p[-s] and p[0] ---------------------------------
// `pos` selects the left-biased (u > 0) or right-biased (u <= 0) member.  Inputs are mirrored so that
// (A3,A2,A1,A0,B1) = (p[-3s],p[-2s],p[-s],p[0],p[s]) for the left form and
//                    (p[2s], p[s], p[0], p[-s],p[-2s]) for the right form.  The candidate
// polynomials and optimal weights of the right-biased form are exact mirrors of the left ones
// (weno_fifth_order.jl:368,518-524); the 1/4-terms of its outer smoothness indicators are NOT
// (weno_fifth_order.jl:315,317 as written) and are reproduced through the (cL, cR) coefficients.
template <int ADV>
OCN_DEVFN double recon5(double A3, double A2, double A1, double A0, double B1, bool pos) {
  if (ADV == ADV_U5) {
    return (2.0 * A3 - 13.0 * A2 + 47.0 * A1 + 27.0 * A0 - 3.0 * B1) * (1.0 / 60.0);
  } else {
    // Everything is expressed through the four first differences of the stencil (fewer FP64 operations than
    // the textbook form; the kernel that uses this is FP64-issue bound):
    //   second differences        t_k  (the 13/12 terms)
    //   biased first differences  u_k  (the 1/4 terms; the as-written right-biased variants differ from the
    //                                   mirrored left-biased ones only in which end carries the factor 3)
    //   candidate values          p_k = A1 + (linear combination of differences)
    const double e1 = A2 - A3, e2 = A1 - A2, e3 = A0 - A1, e4 = B1 - A0;
    const double t0 = e4 - e3, t1 = e3 - e2, t2 = e2 - e1;
    const double s2 = pos ? 2.0 : -2.0;
    const double u0 = fma(-s2, pos ? e3 : e4, t0);   // pos: e4 - 3 e3 = 3A1-4A0+B1 ; neg: 3 e4 - e3 = A1-4A0+3B1
    const double u1 = e2 + e3;                       // (A0 - A2); enters squared
    const double u2 = fma(s2, pos ? e2 : e1, t2);    // pos: 3 e2 - e1 = A3-4A2+3A1 ; neg: e2 - 3 e1 = 3A3-4A2+A1
    // d_k = 12/13 (beta_k + eps) = t^2 + 3/13 u^2 + 12/13 eps.  Only ratios of the (beta_k + eps) enter the
    // normalised weights (Z: tau / (beta_k + eps); JS: their squares against each other), so the common factor
    // drops out and each d_k is one product and two fused multiply-adds.
    const double c3 = 3.0 / 13.0, eps = 1e-6 * (12.0 / 13.0);
    const double d0 = fma(t0, t0, fma(u0 * c3, u0, eps));
    const double d1 = fma(t1, t1, fma(u1 * c3, u1, eps));
    const double d2 = fma(t2, t2, fma(u2 * c3, u2, eps));
    // candidate values minus A1, times the optimal weights (3, 6, 1) [/10 cancels]:
    //   3 p0 = 2 e3 - e4/2,  6 p1 = 2 e3 + e2,  p2 = 5/6 e2 - 1/3 e1
    const double x3 = e3 + e3;
    const double r0 = fma(-0.5, e4, x3);
    const double r1 = x3 + e2;
    const double r2 = fma(5.0 / 6.0, e2, (-1.0 / 3.0) * e1);
    const double q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2;
    const double P0 = q1 * q2, P1 = q0 * q2, P2 = q0 * q1;
    double a0, a1, a2;
    if (ADV == ADV_WENO_Z) {
      // alpha_k / C_k = 1 + (tau/d_k)^2 = (q_k + tau^2) / q_k  ~  Q + tau^2 P_k   with Q = q0 q1 q2, P_k = Q / q_k
      const double tau = d2 - d0, tt = tau * tau, Q = q0 * P0;
      a0 = fma(tt, P0, Q);
      a1 = fma(tt, P1, Q);
      a2 = fma(tt, P2, Q);
    } else {
      // alpha_k / C_k = 1 / d_k^2 ~ P_k
      a0 = P0;
      a1 = P1;
      a2 = P2;
    }
    const double num = fma(a0, r0, fma(a1, r1, a2 * r2));
    const double den = fma(3.0, a0, fma(6.0, a1, a2));
    return fma(num, fast_rcp(den), A1);
  }
}

// upwind reconstruction from memory: face between p[-s] and p[0], advecting velocity ut
template <int ADV>
OCN_DEVFN double recon_mem(const double* p, long s, double ut) {
  bool pos = ut > 0.0;
  double m3 = p[-3 * s], m2 = p[-2 * s], m1 = p[-s], c0 = p[0], c1 = p[s], c2 = p[2 * s];
  double A3 = pos ? m3 : c2, A2 = pos ? m2 : c1, A1 = pos ? m1 : c0, A0 = pos ? c0 : m1, B1 = pos ? c1 : m2;
  return recon5<ADV>(A3, A2, A1, A0, B1, pos);
}

// advective flux (per unit area) through the face between p[-s] and p[0]; ut = advecting velocity there
// UpwindBiasedFirstOrder / ThirdOrder (upwind_biased_first_order.jl:21-35, upwind_biased_third_order.jl:21-35): value at the
// face between p[-s] and p[0] from the upwind side
template <int ADV>
OCN_DEVFN double recon_low(const double* p, long s, bool pos) {
  if (ADV == ADV_U1) return pos ? p[-s] : p[0];
  return pos ? (2.0 * p[0] + 5.0 * p[-s] - p[-2 * s]) / 6.0 : (-p[s] + 5.0 * p[0] + 2.0 * p[-s]) / 6.0;
}

template <int ADV>
OCN_DEVFN double adv_flux(const double* p, long s, double ut) {
  if (ADV == ADV_U1 || ADV == ADV_U3) return ut * recon_low<ADV>(p, s, ut > 0.0);
  if (ADV == ADV_C4) return ut * sym4(p - s, s);
  if (ADV == ADV_C2) return ut * sym2(p - s, s);
  return ut * recon_mem<ADV>(p, s, ut);
}

// ---- Bounded-direction versions (topologically_conditional_interpolation.jl:19-21,46-79) ------------
// idx: 1-based index of the evaluation point along the bounded direction, N its size, nb the buffer.
OCN_DEVFN bool outside_sym(int idx, int N, int nb) { return idx > nb && idx < N + 1 - nb; }
OCN_DEVFN bool outside_left(int idx, int N, int nb) { return idx > nb && idx < N + 1 - (nb - 1); }
OCN_DEVFN bool outside_right(int idx, int N, int nb) { return idx > nb - 1 && idx < N + 1 - nb; }

// symmetric interpolation midway between p[0] and p[s] along a possibly Bounded direction
template <int ADV>
OCN_DEVFN double sym_b(const double* p, long s, bool bounded, int idx, int N, int nb) {
  if (ADV == ADV_C2 || ADV == ADV_U1 || ADV == ADV_U3) return sym2(p, s);   // U1 / U3: symmetric interpolation is the two-point one
  if (bounded && !outside_sym(idx, N, nb)) return sym2(p, s);
  return sym4(p, s);
}

template <int ADV>
OCN_DEVFN double adv_flux_b(const double* p, long s, double ut, bool bounded, int idx, int N, int nb) {
  if (ADV == ADV_C2) return ut * sym2(p - s, s);
  if (ADV == ADV_C4) {
    if (bounded && !outside_sym(idx, N, nb)) return ut * sym2(p - s, s);
    return ut * sym4(p - s, s);
  }
  bool pos = ut > 0.0;
  if (bounded) {
    bool ok = pos ? outside_left(idx, N, nb) : outside_right(idx, N, nb);
    if (!ok) return ut * sym2(p - s, s);
  }
  if (ADV == ADV_U1 || ADV == ADV_U3) return ut * recon_low<ADV>(p, s, pos);
  return ut * recon_mem<ADV>(p, s, ut);
}

// ---- byte-offset addressing (fused kernels): base pointer in SGPRs + one 32-bit VGPR byte offset -------
// Keeps addresses out of 64-bit VGPR pairs (global_load ... v_off, s[base:base+1]); arrays are < 2 GiB.
OCN_DEVFN double ldo(const double* base, unsigned boff) { return *(const double*)((const char*)base + boff); }
OCN_DEVFN double i3_o(const double* b, unsigned o, unsigned s) {
  double c = ldo(b, o);
  return c - ((ldo(b, o + s) - c) - (c - ldo(b, o - s))) * (1.0 / 6.0);
}
// midway between elements at byte offsets o and o + s
OCN_DEVFN double sym4_o(const double* b, unsigned o, unsigned s) {
  return sym4_v(ldo(b, o - s), ldo(b, o), ldo(b, o + s), ldo(b, o + 2 * s));
}
// upwind reconstruction at the face between elements o - s and o
template <int ADV>
OCN_DEVFN double recon_o(const double* b, unsigned o, unsigned s, double ut) {
  bool pos = ut > 0.0;
  double m3 = ldo(b, o - 3 * s), m2 = ldo(b, o - 2 * s), m1 = ldo(b, o - s), c0 = ldo(b, o), c1 = ldo(b, o + s),
         c2 = ldo(b, o + 2 * s);
  double A3 = pos ? m3 : c2, A2 = pos ? m2 : c1, A1 = pos ? m1 : c0, A0 = pos ? c0 : m1, B1 = pos ? c1 : m2;
  return recon5<ADV>(A3, A2, A1, A0, B1, pos);
}
