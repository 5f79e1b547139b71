// kernels.hip -- general (any supported configuration) kernels of the time_step! path.
// One thread per cell / column; no LDS.  The fused, tiled kernels for the headline configuration
// live in fused.hip; these are the correctness baseline and the path for every other configuration.
//
// Reference kernels restated (paths relative to /root/reference/src):
//   Models/NonhydrostaticModels/calculate_nonhydrostatic_tendencies.jl:155-180 + nonhydrostatic_tendency_kernel_functions.jl:44-232
//   Advection/momentum_advection_operators.jl:52-86, tracer_advection_operators.jl:31-35
//   TurbulenceClosures/closure_kernel_operators.jl:22-48, abstract_scalar_diffusivity_closure.jl:172-207
//   Coriolis/f_plane.jl:42-44, BuoyancyModels/{buoyancy_tracer.jl:12, linear_equation_of_state.jl:69-77}
//   TimeSteppers/quasi_adams_bashforth_2.jl:158-166, runge_kutta_3.jl:204-218, store_tendencies.jl:8-11
//   BoundaryConditions/fill_halo_regions_periodic.jl:37-65, fill_halo_regions_flux.jl:16-35,
//   fill_halo_regions_value_gradient.jl:7-99, fill_halo_regions_open.jl:34-39, apply_flux_bcs.jl:111-160
//   Models/NonhydrostaticModels/solve_for_pressure.jl:15-33, pressure_correction.jl:34-40,
//   update_hydrostatic_pressure.jl:10-18
#include "internal.h"

struct Phys {
  int closure;          // OCN_CLOSURE_*
  double nu;
  int coriolis;
  double f;
  const double* pH;     // interior pointer or null
  const double* nu_e;   // AMD
  int buoyancy, bi, Ti, Si;
  double g, alpha, beta;
};

// workgroup shape of the one-thread-per-cell kernels; OCNHIP_<NAME>_BLOCK=bx,by,bz overrides (tuning)
static dim3 tuned_block(const char* env, dim3 def) {
  const char* e = getenv(env);
  int x, y, z;
  if (e && sscanf(e, "%d,%d,%d", &x, &y, &z) == 3 && x > 0 && y > 0 && z == 1 && x * y <= 1024) return dim3(x, y, 1);
  return def;
}

static inline dim3 grid3(const GridDev& g, dim3 b) {   // workgroups are one level thick (b.z == 1)
  return dim3((g.Nx + b.x - 1) / b.x, (g.Ny + b.y - 1) / b.y, g.Nz);
}

// viscosity at the four stress locations (closure_kernel_operators.jl:72-90)
OCN_DEVFN double nu_ccc(const Phys& ph, long p) { return ph.nu_e ? ph.nu_e[p] : ph.nu; }
OCN_DEVFN double nu_ffc(const Phys& ph, long p, long sy) {
  return ph.nu_e ? 0.25 * ((ph.nu_e[p - 1 - sy] + ph.nu_e[p - sy]) + (ph.nu_e[p - 1] + ph.nu_e[p])) : ph.nu;
}
OCN_DEVFN double nu_fcf(const Phys& ph, long p, long sz) {
  return ph.nu_e ? 0.25 * ((ph.nu_e[p - 1 - sz] + ph.nu_e[p - sz]) + (ph.nu_e[p - 1] + ph.nu_e[p])) : ph.nu;
}
OCN_DEVFN double nu_cff(const Phys& ph, long p, long sy, long sz) {
  return ph.nu_e ? 0.25 * ((ph.nu_e[p - sy - sz] + ph.nu_e[p - sz]) + (ph.nu_e[p - sy] + ph.nu_e[p])) : ph.nu;
}

template <int ADV, bool WALLS>
__global__ void k_tend_uvw(GridDev g, Phys ph, const double* __restrict__ u, const double* __restrict__ v,
                           const double* __restrict__ w, double* __restrict__ Gu, double* __restrict__ Gv,
                           double* __restrict__ Gw) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long sy = g.sy, sz = g.sz;
  const long c = i + j * sy + k * sz;
  const int ii = i + 1, jj = j + 1, kk = k + 1;  // 1-based, as in the reference's boundary-buffer tests
  const bool xb = WALLS && g.xb != 0, yb = WALLS && g.yb != 0, zb = g.zb != 0, zf = g.zflat != 0;   // WALLS: Bounded x or y
  const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz, nb = g.nb;
  const double rdx = g.rdx, rdy = g.rdy;
  const double rdzc = zf ? 0.0 : g_rdzc(g, k);
  const double rdzf = zf ? 0.0 : g_rdzf(g, k);
  double gu = 0, gv = 0, gw = 0;

  if (ADV != ADV_NONE) {
    // Indices passed to the *_b helpers are the reference's 1-based index of the evaluation point along the
    // stencil direction (topologically_conditional_interpolation.jl:46-79); ii/jj/kk are those of this cell.
    // ---- div_vu at fcc (momentum_advection_operators.jl:52-56) ----
    auto Fuu = [&](long p, int i1) {   // ccc
      return adv_flux_b<ADV>(u + p + 1, 1, sym_b<ADV>(u + p, 1, xb, i1, Nx, nb), xb, i1, Nx, nb);
    };
    auto Fvu = [&](long p, int j1) {   // ffc
      return adv_flux_b<ADV>(u + p, sy, sym_b<ADV>(v + p - 1, 1, xb, ii, Nx, nb), yb, j1, Ny, nb);
    };
    auto Fwu = [&](long p, int k1) {   // fcf
      return adv_flux_b<ADV>(u + p, sz, sym_b<ADV>(w + p - 1, 1, xb, ii, Nx, nb), zb, k1, Nz, nb);
    };
    gu -= (Fuu(c, ii) - Fuu(c - 1, ii - 1)) * rdx + (Fvu(c + sy, jj + 1) - Fvu(c, jj)) * rdy;
    if (!zf) gu -= (Fwu(c + sz, kk + 1) - Fwu(c, kk)) * rdzc;
    // ---- div_vv at cfc (:68-72) ----
    auto Fuv = [&](long p, int i1) {   // ffc
      return adv_flux_b<ADV>(v + p, 1, sym_b<ADV>(u + p - sy, sy, yb, jj, Ny, nb), xb, i1, Nx, nb);
    };
    auto Fvv = [&](long p, int j1) {   // ccc
      return adv_flux_b<ADV>(v + p + sy, sy, sym_b<ADV>(v + p, sy, yb, j1, Ny, nb), yb, j1, Ny, nb);
    };
    auto Fwv = [&](long p, int k1) {   // cff
      return adv_flux_b<ADV>(v + p, sz, sym_b<ADV>(w + p - sy, sy, yb, jj, Ny, nb), zb, k1, Nz, nb);
    };
    gv -= (Fuv(c + 1, ii + 1) - Fuv(c, ii)) * rdx + (Fvv(c, jj) - Fvv(c - sy, jj - 1)) * rdy;
    if (!zf) gv -= (Fwv(c + sz, kk + 1) - Fwv(c, kk)) * rdzc;
    // ---- div_vw at ccf (:82-86) ----
    // advecting u, v interpolated in z to the w level.  CenteredSecondOrder interpolates the
    // *area-weighted* velocity (centered_second_order.jl:24-25): Ax = dy*dz_c varies with k on stretched grids.
    auto uz = [&](const double* q, long p) -> double {
      if (zf) return q[p];
      if (ADV == ADV_C2 && g.dzc) return 0.5 * (g_dzc(g, k - 1) * q[p - sz] + g_dzc(g, k) * q[p]) * rdzf;
      return sym_b<ADV>(q + p - sz, sz, zb, kk, Nz, nb);
    };
    auto Fuw = [&](long p, int i1) { return adv_flux_b<ADV>(w + p, 1, uz(u, p), xb, i1, Nx, nb); };    // fcf
    auto Fvw = [&](long p, int j1) { return adv_flux_b<ADV>(w + p, sy, uz(v, p), yb, j1, Ny, nb); };   // cff
    gw -= (Fuw(c + 1, ii + 1) - Fuw(c, ii)) * rdx + (Fvw(c + sy, jj + 1) - Fvw(c, jj)) * rdy;
    if (!zf) {
      auto Fww = [&](long p, int k1) {
        return adv_flux_b<ADV>(w + p + sz, sz, sym_b<ADV>(w + p, sz, zb, k1, Nz, nb), zb, k1, Nz, nb);
      };
      gw -= (Fww(c, kk) - Fww(c - sz, kk - 1)) * rdzf;
    }
  }
  // ---- Coriolis (f_plane.jl:42-44) ----
  if (ph.coriolis) {
    gu += ph.f * (0.5 * (0.5 * (v[c - 1] + v[c]) + 0.5 * (v[c - 1 + sy] + v[c + sy])));
    gv -= ph.f * (0.5 * (0.5 * (u[c - sy] + u[c + 1 - sy]) + 0.5 * (u[c] + u[c + 1])));
  }
  // ---- hydrostatic pressure gradient (nonhydrostatic_tendency_kernel_functions.jl:10-15) ----
  if (ph.pH) {
    gu -= (ph.pH[c] - ph.pH[c - 1]) * rdx;
    gv -= (ph.pH[c] - ph.pH[c - sy]) * rdy;
  }
  // ---- viscous stress divergence (closure_kernel_operators.jl:22-41; fluxes -2 nu Sigma) ----
  if (ph.closure != OCN_CLOSURE_NONE) {
    auto rzf = [&](int kf) { return zf ? 0.0 : g_rdzf(g, kf); };
    auto S11 = [&](long p) { return (u[p + 1] - u[p]) * rdx; };
    auto S22 = [&](long p) { return (v[p + sy] - v[p]) * rdy; };
    auto S33 = [&](long p, int kc) { return zf ? 0.0 : (w[p + sz] - w[p]) / g_dzc(g, kc); };
    auto S12 = [&](long p) { return 0.5 * ((u[p] - u[p - sy]) * rdy + (v[p] - v[p - 1]) * rdx); };
    auto S13 = [&](long p, int kf) { return 0.5 * ((zf ? 0.0 : (u[p] - u[p - sz]) * rzf(kf)) + (w[p] - w[p - 1]) * rdx); };
    auto S23 = [&](long p, int kf) { return 0.5 * ((zf ? 0.0 : (v[p] - v[p - sz]) * rzf(kf)) + (w[p] - w[p - sy]) * rdy); };
    double tu = (nu_ccc(ph, c) * S11(c) - nu_ccc(ph, c - 1) * S11(c - 1)) * rdx +
                (nu_ffc(ph, c + sy, sy) * S12(c + sy) - nu_ffc(ph, c, sy) * S12(c)) * rdy;
    double tv = (nu_ffc(ph, c + 1, sy) * S12(c + 1) - nu_ffc(ph, c, sy) * S12(c)) * rdx +
                (nu_ccc(ph, c) * S22(c) - nu_ccc(ph, c - sy) * S22(c - sy)) * rdy;
    double tw = (nu_fcf(ph, c + 1, sz) * S13(c + 1, k) - nu_fcf(ph, c, sz) * S13(c, k)) * rdx +
                (nu_cff(ph, c + sy, sy, sz) * S23(c + sy, k) - nu_cff(ph, c, sy, sz) * S23(c, k)) * rdy;
    if (!zf) {
      tu += (nu_fcf(ph, c + sz, sz) * S13(c + sz, k + 1) - nu_fcf(ph, c, sz) * S13(c, k)) * rdzc;
      tv += (nu_cff(ph, c + sz, sy, sz) * S23(c + sz, k + 1) - nu_cff(ph, c, sy, sz) * S23(c, k)) * rdzc;
      tw += (nu_ccc(ph, c) * S33(c, k) - nu_ccc(ph, c - sz) * S33(c - sz, k - 1)) * rdzf;
    }
    gu += 2.0 * tu;
    gv += 2.0 * tv;
    gw += 2.0 * tw;
  }
  Gu[c] = gu;
  Gv[c] = gv;
  Gw[c] = gw;
}

template <int ADV, bool WALLS>
__global__ void k_tend_c(GridDev g, const double* __restrict__ u, const double* __restrict__ v,
                         const double* __restrict__ w, const double* __restrict__ q, double kap,
                         const double* __restrict__ kap_e, int closure, double* __restrict__ Gc) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long sy = g.sy, sz = g.sz;
  const long c = i + j * sy + k * sz;
  const int kk = k + 1;
  const bool zb = g.zb != 0, zf = g.zflat != 0;
  const double rdx = g.rdx, rdy = g.rdy;
  const double rdzc = zf ? 0.0 : g_rdzc(g, k);
  double gc = 0;
  if (ADV != ADV_NONE) {
    // tracer_advection_operators.jl:31-35; advecting velocity un-interpolated
    auto Fx = [&](long p, int i1) { return adv_flux_b<ADV>(q + p, 1, u[p], WALLS && g.xb != 0, i1, g.Nx, g.nb); };
    auto Fy = [&](long p, int j1) { return adv_flux_b<ADV>(q + p, sy, v[p], WALLS && g.yb != 0, j1, g.Ny, g.nb); };
    gc -= (Fx(c + 1, i + 2) - Fx(c, i + 1)) * rdx + (Fy(c + sy, j + 2) - Fy(c, j + 1)) * rdy;
    if (!zf) {
      auto Fz = [&](long p, int k1) { return adv_flux_b<ADV>(q + p, sz, w[p], zb, k1, g.Nz, g.nb); };
      gc -= (Fz(c + sz, kk + 1) - Fz(c, kk)) * rdzc;
    }
  }
  if (closure != OCN_CLOSURE_NONE) {
    // div q = div(-kappa grad c)  (closure_kernel_operators.jl:43-48)
    auto kx = [&](long p) { return kap_e ? 0.5 * (kap_e[p - 1] + kap_e[p]) : kap; };
    auto ky = [&](long p) { return kap_e ? 0.5 * (kap_e[p - sy] + kap_e[p]) : kap; };
    auto kz = [&](long p) { return kap_e ? 0.5 * (kap_e[p - sz] + kap_e[p]) : kap; };
    double d = (kx(c + 1) * (q[c + 1] - q[c]) * rdx - kx(c) * (q[c] - q[c - 1]) * rdx) * rdx +
               (ky(c + sy) * (q[c + sy] - q[c]) * rdy - ky(c) * (q[c] - q[c - sy]) * rdy) * rdy;
    if (!zf)
      d += (kz(c + sz) * (q[c + sz] - q[c]) / g_dzf(g, k + 1) - kz(c) * (q[c] - q[c - sz]) / g_dzf(g, k)) * rdzc;
    gc += d;
  }
  Gc[c] = gc;
}

// ---- flux boundary conditions (apply_flux_bcs.jl:111-160): G[1] += flux * A / V, G[N] -= flux * A / V ------------
// dim: direction normal to the boundary; threads span the two other directions' interior (:xy / :xz / :yz).
__global__ void k_apply_flux(GridDev g, double* __restrict__ G, int dim, int zloc_face, BCdev lo, BCdev hi) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  const int Na = dim == 0 ? g.Ny : g.Nx, Nb = dim == 2 ? g.Ny : g.Nz;
  if (a >= Na || b >= Nb) return;
  long c, st;
  int N;
  if (dim == 0) { c = a * g.sy + b * g.sz; st = 1; N = g.Nx; }
  else if (dim == 1) { c = a + b * g.sz; st = g.sy; N = g.Ny; }
  else { c = a + b * g.sy; st = g.sz; N = g.Nz; }
  auto rspacing = [&](int idx) -> double {   // A / V at the first / last interior point
    if (dim == 0) return g.rdx;
    if (dim == 1) return g.rdy;
    return 1.0 / (zloc_face ? g_dzf(g, idx) : g_dzc(g, idx));
  };
  if (lo.kind == OCN_BC_FLUX) {
    double val = lo.arr ? lo.arr[a + (long)b * Na] : lo.value;
    G[c] += val * rspacing(0);
  }
  if (hi.kind == OCN_BC_FLUX) {
    double val = hi.arr ? hi.arr[a + (long)b * Na] : hi.value;
    G[c + (N - 1) * st] -= val * rspacing(N - 1);
  }
}

// G^n of tracer t before the boundary fluxes are added, when everything else comes from the tiled tracer kernel: zero where
// that kernel will read it -- the whole array with walls in x / y, else the first and last level only (fused.hip rest_shell)
static void tracer_gn_clear(ocn_model* m, int t) {
  if (g_ocn_dry) return;
  Field& G = m->Gn[3 + t];
  hipStream_t s = m->ctx->stream;
  if (!tracer_rest_shell(m)) {
    OCN_ASYNC(hipMemsetAsync(G.d, 0, G.n * sizeof(double), s));
    return;
  }
  const size_t plane = (size_t)G.sz * sizeof(double);
  OCN_ASYNC(hipMemsetAsync(G.d + (size_t)G.Hz * G.sz, 0, plane, s));
  if (m->gd.Nz > 1) OCN_ASYNC(hipMemsetAsync(G.d + (size_t)(G.Hz + m->gd.Nz - 1) * G.sz, 0, plane, s));
}

void launch_tendencies(ocn_model* m, bool skip_momentum_advection, bool skip_tracer_advection) {
  ProfScope ps(m->ctx, "tendencies");
  const GridDev& g = m->gd;
  hipStream_t s = m->ctx->stream;
  Phys ph;
  memset(&ph, 0, sizeof(ph));
  ph.closure = m->d.closure;
  ph.nu = m->d.nu;
  ph.coriolis = m->d.coriolis_fplane;
  ph.f = m->d.f;
  ph.pH = m->pHY.present ? m->pHY.interior() : nullptr;
  ph.nu_e = m->nu_e.present ? m->nu_e.interior() : nullptr;
  static const dim3 b = tuned_block("OCNHIP_TEND_BLOCK", dim3(64, 4, 1));
  const dim3 gr = grid3(g, b);
  const double *u = m->u.interior(), *v = m->v.interior(), *w = m->w.interior();
  double *Gu = m->Gn[0].interior(), *Gv = m->Gn[1].interior(), *Gw = m->Gn[2].interior();
#define TEND_TRACER(A, W, t)                                                                               \
  ocn_launch(k_tend_c<A, W>, gr, b, s, g, u, v, w, (const double*)m->tr[t].interior(), m->d.kappa[t],     \
             (const double*)(m->kappa_e[t].present ? m->kappa_e[t].interior() : nullptr), m->d.closure, \
             m->Gn[3 + t].interior());
#define TEND_LAUNCH(A, W)                                                       \
  if (skip_momentum_advection) { if (!launch_rest4(m)) ocn_launch(k_tend_uvw<ADV_NONE, W>, gr, b, s, g, ph, u, v, w, Gu, Gv, Gw); }  \
  else ocn_launch(k_tend_uvw<A, W>, gr, b, s, g, ph, u, v, w, Gu, Gv, Gw);      \
  for (int t = 0; t < m->nt; ++t) {                                             \
    if (skip_tracer_advection) { tracer_gn_clear(m, t); }  /* advection AND closure flux come from the tiled tracer kernel; boundary fluxes are added below */ \
    else { TEND_TRACER(A, W, t) }                                               \
  }
#define TEND_CASE(A)                \
  case A:                           \
    if (g.xb || g.yb) {             \
      TEND_LAUNCH(A, true)          \
    } else {                        \
      TEND_LAUNCH(A, false)         \
    }                               \
    break;
  switch (m->d.advection) {
    TEND_CASE(ADV_NONE)
    TEND_CASE(ADV_C2)
    TEND_CASE(ADV_C4)
    TEND_CASE(ADV_U5)
    TEND_CASE(ADV_WENO_Z)
    TEND_CASE(ADV_WENO_JS)
    TEND_CASE(ADV_U1)
    TEND_CASE(ADV_U3)
  }
#undef TEND_CASE
#undef TEND_LAUNCH
#undef TEND_TRACER
  // boundary contributions in every Bounded direction
  for (int dim = 0; dim < 3; ++dim) {
    if (m->g->topo[dim] != OCN_BOUNDED) continue;
    const int Na = dim == 0 ? g.Ny : g.Nx, Nb = dim == 2 ? g.Ny : g.Nz;
    dim3 b2(64, 4, 1), g2((Na + 63) / 64, (Nb + 3) / 4, 1);
    for (int f = 0; f < 3 + m->nt; ++f) {
      Field* fld = f == 0 ? &m->u : f == 1 ? &m->v : f == 2 ? &m->w : &m->tr[f - 3];
      const BCdev &lo = fld->bc[2 * dim], &hi = fld->bc[2 * dim + 1];
      bool nl = lo.kind == OCN_BC_FLUX && (lo.arr || lo.value != 0.0);
      bool nh = hi.kind == OCN_BC_FLUX && (hi.arr || hi.value != 0.0);
      if (!nl && !nh) continue;
      BCdev l_ = lo, h_ = hi;
      if (!nl) l_.kind = OCN_BC_NOFLUX;
      if (!nh) h_.kind = OCN_BC_NOFLUX;
      ocn_launch(k_apply_flux, g2, b2, s, g, m->Gn[f].interior(), dim, fld->loc[2], l_, h_);
    }
  }
}

// ---- time stepping (ab2_step_field!, rk3_substep_field!, store_field_tendencies!) ------------------
struct StepPtrs {
  double* U[OCN_NF];
  const double* Gn[OCN_NF];
  const double* Gm[OCN_NF];
  int n;
};

__global__ void k_step(GridDev g, StepPtrs P, double dt, double cn, double cm, int use_m) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long c = i + j * g.sy + k * g.sz;
  for (int f = 0; f < P.n; ++f) {
    // quasi_adams_bashforth_2.jl:165 / runge_kutta_3.jl:208,216 operation order
    double inc = use_m ? dt * (cn * P.Gn[f][c] + cm * P.Gm[f][c]) : dt * cn * P.Gn[f][c];
    P.U[f][c] += inc;
  }
}

// U += dt * (cn * Gn + cm * Gm)   [AB2: cn = 1.5+chi, cm = -(0.5+chi);  RK3: cn = gamma, cm = zeta]
void launch_step(ocn_model* m, double dt, double cn, double cm, int use_m, bool tracers_only) {
  ProfScope ps(m->ctx, "step");
  const GridDev& g = m->gd;
  StepPtrs P;
  P.n = 0;
  for (int f = tracers_only ? 3 : 0; f < 3 + m->nt; ++f) {
    Field* fld = f == 0 ? &m->u : f == 1 ? &m->v : f == 2 ? &m->w : &m->tr[f - 3];
    P.U[P.n] = fld->interior();
    P.Gn[P.n] = m->Gn[f].interior();
    P.Gm[P.n] = m->Gm[f].interior();
    ++P.n;
  }
  if (P.n == 0) return;
  dim3 b(64, 4, 1);
  ocn_launch(k_step, grid3(g, b), b, m->ctx->stream, g, P, dt, cn, cm, use_m);
}

struct CopyPtrs {
  double* dst[OCN_NF];
  const double* src[OCN_NF];
  int n;
};
__global__ void k_copy_interior(GridDev g, CopyPtrs P) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long c = i + j * g.sy + k * g.sz;
  for (int f = 0; f < P.n; ++f) P.dst[f][c] = P.src[f][c];
}

void launch_store(ocn_model* m) {
  ProfScope ps(m->ctx, "store");
  CopyPtrs P;
  P.n = 3 + m->nt;
  for (int f = 0; f < P.n; ++f) {
    P.dst[f] = m->Gm[f].interior();
    P.src[f] = m->Gn[f].interior();
  }
  dim3 b(64, 4, 1);
  ocn_launch(k_copy_interior, grid3(m->gd, b), b, m->ctx->stream, m->gd, P);
}

// ---- halo fills ------------------------------------------------------------------------------------------
// Periodic: exact restatement of fill_halo_regions_periodic.jl:37-65 on the *parent* array, sequential in
// the halo index (matters when N < H), launched over the full parent extent of the other two dims.
__global__ void k_fill_periodic(FieldPtrs F, int dim, int N, int H, long sy, long sz) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  const int f = blockIdx.z;
  const int Tx = F.Tx[f], Ty = F.Ty[f], Tz = F.Tz[f];
  double* p = F.p[f];
  long base, st;
  if (dim == 0) {  // a: y, b: z
    if (a >= Ty || b >= Tz) return;
    base = a * sy + b * sz;
    st = 1;
  } else if (dim == 1) {  // a: x, b: z
    if (a >= Tx || b >= Tz) return;
    base = a + b * sz;
    st = sy;
  } else {  // a: x, b: y
    if (a >= Tx || b >= Ty) return;
    base = a + b * sy;
    st = sz;
  }
  for (int i = 0; i < H; ++i) {
    p[base + i * st] = p[base + (N + i) * st];
    p[base + (N + H + i) * st] = p[base + (H + i) * st];
  }
}

static void fill_launch_shape(const FieldPtrs& F, int dim, dim3& b, dim3& gr) {
  int Tx = 0, Ty = 0, Tz = 0;
  for (int f = 0; f < F.n; ++f) {
    Tx = F.Tx[f] > Tx ? F.Tx[f] : Tx;
    Ty = F.Ty[f] > Ty ? F.Ty[f] : Ty;
    Tz = F.Tz[f] > Tz ? F.Tz[f] : Tz;
  }
  int na = dim == 0 ? Ty : Tx;
  int nbb = dim == 2 ? Ty : Tz;
  b = dim3(dim == 0 ? 8 : 64, dim == 0 ? 32 : 4, 1);
  gr = dim3((na + b.x - 1) / b.x, (nbb + b.y - 1) / b.y, F.n);
}

void launch_fill_periodic(ocn_model* m, const FieldPtrs& F, int dim) {
  const GridDev& g = m->gd;
  int N = dim == 0 ? g.Nx : dim == 1 ? g.Ny : g.Nz;
  int H = dim == 0 ? g.Hx : dim == 1 ? g.Hy : g.Hz;
  if (H == 0) return;
  dim3 b, gr;
  fill_launch_shape(F, dim, b, gr);
  ocn_launch(k_fill_periodic, gr, b, m->ctx->stream, F, dim, N, H, g.sy, g.sz);
}

// Periodic x AND y in one pass: every halo cell of the frame (rows, columns, corners) takes its periodic image straight
// from the interior -- identical to the y pass followed by the x pass (fill_halo_regions.jl order) whenever N >= H.
__global__ void k_fill_periodic_xy(FieldPtrs F, int Nx, int Ny, int H, long sy, long sz) {
  const int f = blockIdx.z;
  const int Tx = Nx + 2 * H;
  const int nrow = 2 * H * Tx, ncol = 2 * H * Ny;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;      // index inside the frame of one plane
  const int k = blockIdx.y;
  if (q >= nrow + ncol || k >= F.Tz[f]) return;
  int a, b;
  if (q < nrow) {                                           // the 2H halo rows, full width
    const int r = q / Tx;
    a = q - r * Tx;
    b = r < H ? r : Ny + r;                                 // r in [H, 2H) -> rows Ny+H .. Ny+2H-1
  } else {                                                  // the 2H halo columns of the interior rows
    const int t = q - nrow, r = t / (2 * H), cidx = t - r * (2 * H);
    b = H + r;
    a = cidx < H ? cidx : Nx + cidx;
  }
  int as = a - H, bs = b - H;                               // interior image
  as = as < 0 ? as + Nx : (as >= Nx ? as - Nx : as);
  bs = bs < 0 ? bs + Ny : (bs >= Ny ? bs - Ny : bs);
  double* p = F.p[f] + (long)k * sz;
  p[a + (long)b * sy] = p[(as + H) + (long)(bs + H) * sy];
}

bool launch_fill_periodic_xy(ocn_model* m, const FieldPtrs& F) {
  const GridDev& g = m->gd;
  if (g.Hx != g.Hy || g.Hx == 0 || g.Nx < g.Hx || g.Ny < g.Hy || g.xb || g.yb) return false;
  for (int f = 0; f < F.n; ++f)
    if (F.Tx[f] != g.Nx + 2 * g.Hx || F.Ty[f] != g.Ny + 2 * g.Hy) return false;
  int Tz = 0;
  for (int f = 0; f < F.n; ++f) Tz = F.Tz[f] > Tz ? F.Tz[f] : Tz;
  const int frame = 2 * g.Hx * (g.Nx + 2 * g.Hx) + 2 * g.Hx * g.Ny;
  dim3 b(256, 1, 1), gr((frame + 255) / 256, Tz, F.n);
  ocn_launch(k_fill_periodic_xy, gr, b, m->ctx->stream, F, g.Nx, g.Ny, g.Hx, g.sy, g.sz);
  return true;
}

// Flat x / y: every physical slot along the direction holds the single logical value
__global__ void k_fill_flat(FieldPtrs F, int dim, int H, long sy, long sz) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  const int f = blockIdx.z;
  const int Tx = F.Tx[f], Ty = F.Ty[f], Tz = F.Tz[f];
  double* p = F.p[f];
  long base, st;
  if (dim == 0) {
    if (a >= Ty || b >= Tz) return;
    base = a * sy + b * sz;
    st = 1;
  } else {
    if (a >= Tx || b >= Tz) return;
    base = a + b * sz;
    st = sy;
  }
  const double val = p[base + H * st];
  for (int i = 0; i < 2 * H + 1; ++i)
    if (i != H) p[base + i * st] = val;
}

void launch_fill_flat(ocn_model* m, const FieldPtrs& F, int dim) {
  const GridDev& g = m->gd;
  int H = dim == 0 ? g.Hx : g.Hy;
  if (H == 0) return;
  dim3 b, gr;
  fill_launch_shape(F, dim, b, gr);
  ocn_launch(k_fill_flat, gr, b, m->ctx->stream, F, dim, H, g.sy, g.sz);
}

// Bounded directions: one halo cell per side (fill_halo_regions_flux.jl:16-35, ..value_gradient.jl:7-99,
// ..open.jl:34-39), launched over the two other directions' *centre* sizes (:yz / :xz / :xy).
struct BoundedFill {          // one launch fills the same direction of several fields
  double* p[OCN_NF + 2];
  int face[OCN_NF + 2];
  BCdev lo[OCN_NF + 2], hi[OCN_NF + 2];
};
__global__ void k_fill_bounded(GridDev g, BoundedFill F, int dim) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  double* __restrict__ p = F.p[blockIdx.z];
  const int face = F.face[blockIdx.z];
  const BCdev lo = F.lo[blockIdx.z], hi = F.hi[blockIdx.z];
  const int Na = dim == 0 ? g.Ny : g.Nx, Nb = dim == 2 ? g.Ny : g.Nz;
  if (a >= Na || b >= Nb) return;
  long c, st;
  int N;
  if (dim == 0) { c = a * g.sy + b * g.sz; st = 1; N = g.Nx; }
  else if (dim == 1) { c = a + b * g.sz; st = g.sy; N = g.Ny; }
  else { c = a + b * g.sy; st = g.sz; N = g.Nz; }
  // spacing between the first interior and the first halo point: at flip(loc), index 1 / N+1 (1-based)
  auto spacing = [&](int idx) -> double {
    if (dim == 0) return g.dx;
    if (dim == 1) return g.dy;
    return face ? g_dzc(g, idx) : g_dzf(g, idx);
  };
  {
    double val = lo.arr ? lo.arr[a + (long)b * Na] : lo.value;
    if (lo.kind == OCN_BC_NOFLUX || lo.kind == OCN_BC_FLUX) p[c - st] = p[c];
    else if (lo.kind == OCN_BC_IMPENETRABLE) p[c] = val;
    else if (lo.kind == OCN_BC_VALUE || lo.kind == OCN_BC_GRADIENT) {
      double D = spacing(0);
      double cI = p[c];
      double grad = lo.kind == OCN_BC_GRADIENT ? val : (cI - val) / (D / 2);
      p[c - st] = cI + grad * (-D);
    }
  }
  {
    double val = hi.arr ? hi.arr[a + (long)b * Na] : hi.value;
    if (hi.kind == OCN_BC_NOFLUX || hi.kind == OCN_BC_FLUX) p[c + N * st] = p[c + (N - 1) * st];
    else if (hi.kind == OCN_BC_IMPENETRABLE) p[c + N * st] = val;
    else if (hi.kind == OCN_BC_VALUE || hi.kind == OCN_BC_GRADIENT) {
      double D = spacing(N);
      double cI = p[c + (N - 1) * st];
      double grad = hi.kind == OCN_BC_GRADIENT ? val : (val - cI) / (D / 2);
      p[c + N * st] = cI + grad * D;
    }
  }
}

void launch_fill_bounded(ocn_model* m, Field** fs, int n, int dim) {
  const GridDev& g = m->gd;
  const int Na = dim == 0 ? g.Ny : g.Nx, Nb = dim == 2 ? g.Ny : g.Nz;
  BoundedFill F;
  for (int i = 0; i < n; ++i) {
    F.p[i] = fs[i]->interior();
    F.face[i] = fs[i]->loc[dim];
    F.lo[i] = fs[i]->bc[2 * dim];
    F.hi[i] = fs[i]->bc[2 * dim + 1];
  }
  dim3 b(64, 4, 1), gr((Na + 63) / 64, (Nb + 3) / 4, n);
  ocn_launch(k_fill_bounded, gr, b, m->ctx->stream, g, F, dim);
}

// ---- Poisson right-hand side (solve_for_pressure.jl:15-18,30-33) ------------------------------------------
__global__ void k_rhs(GridDev g, const double* __restrict__ u, const double* __restrict__ v,
                      const double* __restrict__ w, double rdt, int mult_dz, double* __restrict__ rhs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // plain block order: the XCD-major one (ocn_cell_ij) measured 4-5 % slower here
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long c = i + j * g.sy + k * g.sz;
  double dzc = g.zflat ? 1.0 : g_dzc(g, k);
  double div = (u[c + 1] - u[c]) * g.rdx + (v[c + g.sy] - v[c]) * g.rdy;
  if (!g.zflat) div += (w[c + g.sz] - w[c]) / dzc;
  double r = div * rdt;
  if (mult_dz) r *= dzc;
  rhs[i + (long)g.Nx * (j + (long)g.Ny * k)] = r;
}

void launch_rhs(ocn_model* m, double dt, double* rhs, int mult_dz) {
  ProfScope ps(m->ctx, "rhs");
  dim3 b(64, 4, 1);
  ocn_launch(k_rhs, grid3(m->gd, b), b, m->ctx->stream, m->gd, (const double*)pred_u(m).interior(),
             (const double*)pred_v(m).interior(), (const double*)pred_w(m).interior(), 1.0 / dt, mult_dz, rhs);
}

// ---- projection (pressure_correction.jl:34-40) -----------------------------------------------------------------
// us, vs, ws: the predictor (the same arrays as u, v, w, or the separate predictor buffers of the tiled Bounded-z path)
__global__ void k_pcorrect(GridDev g, const double* __restrict__ p, double dt, const double* us, const double* vs,
                           const double* ws, double* u, double* v, double* w) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long c = i + j * g.sy + k * g.sz;
  double pc = p[c];
  u[c] = us[c] - (pc - p[c - 1]) * g.rdx * dt;
  v[c] = vs[c] - (pc - p[c - g.sy]) * g.rdy * dt;
  if (!g.zflat) w[c] = ws[c] - (pc - p[c - g.sz]) / g_dzf(g, k) * dt;
}

void launch_pcorrect(ocn_model* m, double dt) {
  ProfScope ps(m->ctx, "pcorrect");
  dim3 b(64, 4, 1);
  ocn_launch(k_pcorrect, grid3(m->gd, b), b, m->ctx->stream, m->gd, (const double*)m->pNHS.interior(), dt,
             (const double*)pred_u(m).interior(), (const double*)pred_v(m).interior(), (const double*)pred_w(m).interior(),
             m->u.interior(), m->v.interior(), m->w.interior());
  m->pred_active = false;   // u, v, w hold the corrected velocities again
}

// ---- hydrostatic pressure anomaly (update_hydrostatic_pressure.jl:10-18) ---------------------------------------
__global__ void k_hydrostatic(GridDev g, Phys ph, const double* __restrict__ b0, const double* __restrict__ T,
                              const double* __restrict__ S, double* __restrict__ pH) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const long c = i + j * g.sy, sz = g.sz;
  auto bz = [&](long p) -> double {
    if (ph.buoyancy == OCN_BUOYANCY_TRACER) return b0[p];
    if (ph.buoyancy == OCN_BUOYANCY_LINEAR_TS) return ph.g * (ph.alpha * T[p] - ph.beta * S[p]);
    return 0.0;
  };
  const int Nz = g.Nz;
  double bup = bz(c + Nz * sz);
  double acc = 0;
  // the recurrence is serial in k, the loads are not: eight levels' buoyancies are fetched before they are summed, so a
  // column pays one memory latency per eight levels instead of one per level (0.080 -> see profiles at 256x256x128)
  constexpr int CH = 8;   // 16 / 32 levels in flight: 0.091 / 0.121 ms against 0.085 (256 x 256 x 128)
  for (int k1 = Nz - 1; k1 >= 0; k1 -= CH) {
    double bk[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) bk[q] = (k1 - q >= 0) ? bz(c + (long)(k1 - q) * sz) : 0.0;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int k = k1 - q;
      if (k < 0) break;
      const double term = 0.5 * (bk[q] + bup) * g_dzf(g, k + 1);
      acc = (k == Nz - 1) ? -term : acc - term;
      pH[c + (long)k * sz] = acc;
      bup = bk[q];
    }
  }
}

// The same downward recurrence cut into SEG segments per column: one thread per (column, segment) sums its levels from the
// segment's top with a zero carry, the segment totals meet in LDS, and every thread adds the sum of the segments above its own
// before it stores.  The serial kernel keeps one wave per SIMD busy for 128 dependent steps (2.4 TB/s at 256 x 256 x 128); this
// one has SEG times the waves and 1 / SEG of the chain.  The association differs from the reference's top-to-bottom sum by the
// one addition of the carry (a few ulp of pHY'): used for columns of 64 levels or more, the serial kernel below that.
template <int SEG, int LPS>
__global__ void __launch_bounds__(64 * SEG) k_hydrostatic_seg(GridDev g, Phys ph, const double* __restrict__ b0, const double* __restrict__ T,
                                                               const double* __restrict__ S, double* __restrict__ pH) {
  OCN_SHARED double tot[SEG][64];
  const int tx = threadIdx.x, s = threadIdx.y;                 // s = 0: the top segment
  const int col = blockIdx.x * 64 + tx;                        // flattened (i, j): i fastest
  const int ncol = g.Nx * g.Ny;
  const bool ok = col < ncol;
  const int j = ok ? col / g.Nx : 0, i = ok ? col - j * g.Nx : 0;
  const long c = i + (long)j * g.sy, sz = g.sz;
  auto bz = [&](long p) -> double {
    if (ph.buoyancy == OCN_BUOYANCY_TRACER) return b0[p];
    if (ph.buoyancy == OCN_BUOYANCY_LINEAR_TS) return ph.g * (ph.alpha * T[p] - ph.beta * S[p]);
    return 0.0;
  };
  const int Nz = g.Nz;
  const int ktop = Nz - 1 - s * LPS;                           // this segment: levels ktop, ktop - 1, ..., ktop - LPS + 1 (>= 0)
  double bk[LPS + 1], loc[LPS];
#pragma unroll
  for (int q = 0; q <= LPS; ++q) {
    const int k = ktop + 1 - q;                                // the level above the segment first
    bk[q] = (ok && k >= 0 && k <= Nz) ? bz(c + (long)k * sz) : 0.0;
  }
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < LPS; ++q) {
    const int k = ktop - q;
    double term = 0.0;
    if (k >= 0) term = 0.5 * (bk[q + 1] + bk[q]) * g_dzf(g, k + 1);
    acc = (q == 0) ? -term : acc - term;
    loc[q] = acc;
  }
  tot[s][tx] = acc;
  __syncthreads();
  double carry = 0.0;
  for (int r = 0; r < s; ++r) carry = (r == 0) ? tot[0][tx] : carry + tot[r][tx];
#pragma unroll
  for (int q = 0; q < LPS; ++q) {
    const int k = ktop - q;
    if (ok && k >= 0) pH[c + (long)k * sz] = s == 0 ? loc[q] : carry + loc[q];
  }
}

void launch_hydrostatic(ocn_model* m) {
  if (!m->pHY.present || m->d.buoyancy == OCN_BUOYANCY_NONE) return;  // pHY' stays identically zero
  ProfScope ps(m->ctx, "hydrostatic");
  const GridDev& g = m->gd;
  Phys ph;
  memset(&ph, 0, sizeof(ph));
  ph.buoyancy = m->d.buoyancy;
  ph.g = m->d.g;
  ph.alpha = m->d.alpha;
  ph.beta = m->d.beta;
  const double* b0 = m->d.b_index >= 0 ? m->tr[m->d.b_index].interior() : nullptr;
  const double* T = m->d.T_index >= 0 ? m->tr[m->d.T_index].interior() : nullptr;
  const double* S = m->d.S_index >= 0 ? m->tr[m->d.S_index].interior() : nullptr;
  if (g.Nz >= 64 && g.Nz <= 8 * 32) {   // segmented form: 8 segments of 8 / 16 / 32 levels
    const int ncol = g.Nx * g.Ny;
    dim3 bs(64, 8, 1), gs((ncol + 63) / 64, 1, 1);
    if (g.Nz <= 64) ocn_launch_sync(k_hydrostatic_seg<8, 8>, gs, bs, m->ctx->stream, g, ph, b0, T, S, m->pHY.interior());
    else if (g.Nz <= 128) ocn_launch_sync(k_hydrostatic_seg<8, 16>, gs, bs, m->ctx->stream, g, ph, b0, T, S, m->pHY.interior());
    else ocn_launch_sync(k_hydrostatic_seg<8, 32>, gs, bs, m->ctx->stream, g, ph, b0, T, S, m->pHY.interior());
    return;
  }
  dim3 b(64, 4, 1), gr((g.Nx + 63) / 64, (g.Ny + 3) / 4, 1);
  ocn_launch(k_hydrostatic, gr, b, m->ctx->stream, g, ph, b0, T, S, m->pHY.interior());
}

// ---- compact (Nx,Ny,Nz) array -> field interior (copy_real_component!, fft_based_poisson_solver.jl:122-125) --
__global__ void k_copy_to_field(GridDev g, const double* __restrict__ src, double* __restrict__ dst) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  dst[i + j * g.sy + k * g.sz] = src[i + (long)g.Nx * (j + (long)g.Ny * k)];
}
void launch_copy_to_field(ocn_model* m, const double* src, Field& f) {
  dim3 b(64, 4, 1);
  ocn_launch(k_copy_to_field, grid3(m->gd, b), b, m->ctx->stream, m->gd, src, f.interior());
}

// ---- max |div U| (test helper) -------------------------------------------------------------------------------------
__global__ void k_maxdiv(GridDev g, const double* __restrict__ u, const double* __restrict__ v,
                         const double* __restrict__ w, double* out) {
  // one thread per (i,j) column, atomic max on the bit pattern of a non-negative double
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  double mx = 0;
  for (int k = 0; k < g.Nz; ++k) {
    const long c = i + j * g.sy + k * g.sz;
    double div = (u[c + 1] - u[c]) * g.rdx + (v[c + g.sy] - v[c]) * g.rdy;
    if (!g.zflat) div += (w[c + g.sz] - w[c]) / g_dzc(g, k);
    mx = fmax(mx, fabs(div));
  }
#ifndef OCN_HOST_EMU
  atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(mx));
#else
  if (mx > *out) *out = mx;
#endif
}
void launch_maxdiv(ocn_model* m, double* out_dev) {
  const GridDev& g = m->gd;
  dim3 b(64, 4, 1), gr((g.Nx + 63) / 64, (g.Ny + 3) / 4, 1);
  ocn_launch(k_maxdiv, gr, b, m->ctx->stream, g, (const double*)m->u.interior(), (const double*)m->v.interior(),
             (const double*)m->w.interior(), out_dev);
}

// ---- AnisotropicMinimumDissipation predictors (anisotropic_minimum_dissipation.jl:138-178,229-344) --------
// One thread per cell; every quantity is the reference's index function evaluated at the calling index
// (filter widths are always 2 x the *centre* spacing at that index, :213-226; norm_dx_u etc. are the plain
// gradients, velocity_tracer_gradients.jl:126-128; cy_uy uses I_xz on norm_dy_w as written, :326).
struct AmdCtx {
  GridDev g;
  const double *u, *v, *w;
  long sy, sz;
  double rdx, rdy, Dx, Dy;
  double xy, yx;            // Dx / Dy, Dy / Dx
  // per-level factors, entry [k + Hz] for k in [-Hz, Nz + Hz] (built once on the host, see amd_build_table): the
  // kernel used to spend two thirds of its instructions on FP64 divisions whose operands are the same for every
  // thread of a level.
  const double *Dz, *xz, *zx, *yz, *zy, *rdzf, *rdzc, *d2;
  int Hz;
};
OCN_DEVFN double amd_Dz(const AmdCtx& a, int k) { return a.Dz[k + a.Hz]; }
OCN_DEVFN double amd_dxu(const AmdCtx& a, long p) { return (a.u[p + 1] - a.u[p]) * a.rdx; }
OCN_DEVFN double amd_dyv(const AmdCtx& a, long p) { return (a.v[p + a.sy] - a.v[p]) * a.rdy; }
OCN_DEVFN double amd_dzw(const AmdCtx& a, long p, int k) { return (a.w[p + a.sz] - a.w[p]) * a.rdzc[k + a.Hz]; }
OCN_DEVFN double amd_ndxv(const AmdCtx& a, long p) { return a.xy * ((a.v[p] - a.v[p - 1]) * a.rdx); }
OCN_DEVFN double amd_ndyu(const AmdCtx& a, long p) { return a.yx * ((a.u[p] - a.u[p - a.sy]) * a.rdy); }
OCN_DEVFN double amd_ndxw(const AmdCtx& a, long p, int k) { return a.xz[k + a.Hz] * ((a.w[p] - a.w[p - 1]) * a.rdx); }
OCN_DEVFN double amd_ndzu(const AmdCtx& a, long p, int k) { return a.zx[k + a.Hz] * ((a.u[p] - a.u[p - a.sz]) * a.rdzf[k + a.Hz]); }
OCN_DEVFN double amd_ndyw(const AmdCtx& a, long p, int k) { return a.yz[k + a.Hz] * ((a.w[p] - a.w[p - a.sy]) * a.rdy); }
OCN_DEVFN double amd_ndzv(const AmdCtx& a, long p, int k) { return a.zy[k + a.Hz] * ((a.v[p] - a.v[p - a.sz]) * a.rdzf[k + a.Hz]); }
OCN_DEVFN double amd_S12(const AmdCtx& a, long p) { return 0.5 * (amd_ndyu(a, p) + amd_ndxv(a, p)); }
OCN_DEVFN double amd_S13(const AmdCtx& a, long p, int k) { return 0.5 * (amd_ndzu(a, p, k) + amd_ndxw(a, p, k)); }
OCN_DEVFN double amd_S23(const AmdCtx& a, long p, int k) { return 0.5 * (amd_ndzv(a, p, k) + amd_ndyw(a, p, k)); }
// double interpolations to ccc of a function f(p, k): I_y(I_x f), I_z(I_x f), I_z(I_y f)
template <class F> OCN_DEVFN double amd_Ixy(const AmdCtx& a, long p, int k, F f) {
  return 0.5 * (0.5 * (f(p, k) + f(p + 1, k)) + 0.5 * (f(p + a.sy, k) + f(p + 1 + a.sy, k)));
}
template <class F> OCN_DEVFN double amd_Ixz(const AmdCtx& a, long p, int k, F f) {
  return 0.5 * (0.5 * (f(p, k) + f(p + 1, k)) + 0.5 * (f(p + a.sz, k + 1) + f(p + 1 + a.sz, k + 1)));
}
template <class F> OCN_DEVFN double amd_Iyz(const AmdCtx& a, long p, int k, F f) {
  return 0.5 * (0.5 * (f(p, k) + f(p + a.sy, k)) + 0.5 * (f(p + a.sz, k + 1) + f(p + a.sy + a.sz, k + 1)));
}

struct AmdTracers {
  const double* q[OCN_MAX_TRACERS];
  double* kap[OCN_MAX_TRACERS];
  double Ck[OCN_MAX_TRACERS];
  int n;
  // buoyancy modification of nu_e (:142-154,299-312): b = q[b0] (BuoyancyTracer) or Cg (cb0 q[b0] - cb1 q[b1])
  // (LinearEquationOfState: g (alpha T - beta S)); has_Cb = 0 leaves the term out (Cb = nothing)
  int has_Cb, lin, b0, b1;
  double Cb, Cg, cb0, cb1;
};

// nu_e and every kappa_e in one pass: the interpolated velocity gradients are shared by all predictors
// (calc_nu / calc_kappa, anisotropic_minimum_dissipation.jl:138-178).
template <int NT>   // NT >= 0: tracer count known at compile time (the loop unrolls, its loads can be issued early); -1: any
__global__ void k_amd_all(AmdCtx a, double Cnu, double* __restrict__ nu, AmdTracers T) {
  const GridDev& g = a.g;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // plain block order: the XCD-major one (ocn_cell_ij) measured 4-5 % slower here
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;   // one level per workgroup: k is wave-uniform, so spacings are scalar loads and 1/dz is computed once per wave
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  const long sy = a.sy, sz = a.sz;
  const long c = i + j * sy + k * sz;
  auto sq = [](double x) { return x * x; };
  auto ndxv = [&](long p, int) { return amd_ndxv(a, p); };
  auto ndyu = [&](long p, int) { return amd_ndyu(a, p); };
  auto ndxw = [&](long p, int kk) { return amd_ndxw(a, p, kk); };
  auto ndzu = [&](long p, int kk) { return amd_ndzu(a, p, kk); };
  auto ndyw = [&](long p, int kk) { return amd_ndyw(a, p, kk); };
  auto ndzv = [&](long p, int kk) { return amd_ndzv(a, p, kk); };
  auto S12 = [&](long p, int) { return amd_S12(a, p); };
  auto S13 = [&](long p, int kk) { return amd_S13(a, p, kk); };
  auto S23 = [&](long p, int kk) { return amd_S23(a, p, kk); };
  const double dxu = amd_dxu(a, c), dyv = amd_dyv(a, c), dzw = amd_dzw(a, c, k);
  const double S11 = dxu, S22 = dyv, S33 = dzw;
  // gradients interpolated to ccc (shared by nu_e and kappa_e)
  const double xy_dxv = amd_Ixy(a, c, k, ndxv), xy_dyu = amd_Ixy(a, c, k, ndyu);
  const double xz_dxw = amd_Ixz(a, c, k, ndxw), xz_dzu = amd_Ixz(a, c, k, ndzu);
  const double yz_dyw = amd_Iyz(a, c, k, ndyw), yz_dzv = amd_Iyz(a, c, k, ndzv);
  // squares interpolated to ccc
  const double xy_dxv2 = amd_Ixy(a, c, k, [&](long p, int kk) { return sq(ndxv(p, kk)); });
  const double xy_dyu2 = amd_Ixy(a, c, k, [&](long p, int kk) { return sq(ndyu(p, kk)); });
  const double xz_dxw2 = amd_Ixz(a, c, k, [&](long p, int kk) { return sq(ndxw(p, kk)); });
  const double xz_dzu2 = amd_Ixz(a, c, k, [&](long p, int kk) { return sq(ndzu(p, kk)); });
  const double yz_dyw2 = amd_Iyz(a, c, k, [&](long p, int kk) { return sq(ndyw(p, kk)); });
  const double yz_dzv2 = amd_Iyz(a, c, k, [&](long p, int kk) { return sq(ndzv(p, kk)); });
  const double q = sq(dxu) + sq(dyv) + sq(dzw) + xy_dxv2 + xy_dyu2 + xz_dxw2 + xz_dzu2 + yz_dyw2 + yz_dzv2;
  const double d2 = a.d2[k + a.Hz];
  double nus = 0.0;
  if (q != 0.0) {
    const double r1 = S11 * sq(dxu) + S22 * xy_dxv2 + S33 * xz_dxw2 +
                      2 * dxu * amd_Ixy(a, c, k, [&](long p, int kk) { return ndxv(p, kk) * S12(p, kk); }) +
                      2 * dxu * amd_Ixz(a, c, k, [&](long p, int kk) { return ndxw(p, kk) * S13(p, kk); }) +
                      2 * xy_dxv * xz_dxw * amd_Iyz(a, c, k, S23);
    const double r2 = S11 * xy_dyu2 + S22 * sq(dyv) + S33 * yz_dyw2 +
                      2 * dyv * amd_Ixy(a, c, k, [&](long p, int kk) { return ndyu(p, kk) * S12(p, kk); }) +
                      2 * xy_dyu * yz_dyw * amd_Ixz(a, c, k, S13) +
                      2 * dyv * amd_Iyz(a, c, k, [&](long p, int kk) { return ndyw(p, kk) * S23(p, kk); });
    const double r3 = S11 * xz_dzu2 + S22 * yz_dzv2 + S33 * sq(dzw) +
                      2 * xz_dzu * yz_dzv * amd_Ixy(a, c, k, S12) +
                      2 * dzw * amd_Ixz(a, c, k, [&](long p, int kk) { return ndzu(p, kk) * S13(p, kk); }) +
                      2 * dzw * amd_Iyz(a, c, k, [&](long p, int kk) { return ndzv(p, kk) * S23(p, kk); });
    double Cb_zeta = 0.0;
    if (T.has_Cb) {
      const double* __restrict__ qa = T.q[T.b0];
      const double* __restrict__ qb = T.q[T.b1];
      // the reference's association: g * (alpha T - beta S)  (linear_equation_of_state.jl:69-71); lin = 0: b itself
      auto bp = [&](long p) { return T.lin ? T.Cg * (T.cb0 * qa[p] - T.cb1 * qb[p]) : qa[p]; };
      const double bc = bp(c);
      const double bxm = (bc - bp(c - 1)) * a.rdx, bxp = (bp(c + 1) - bc) * a.rdx;            // d_x b at fcc i, i+1
      const double bym = (bc - bp(c - sy)) * a.rdy, byp = (bp(c + sy) - bc) * a.rdy;          // d_y b at cfc j, j+1
      const double bzm = (bc - bp(c - sz)) * a.rdzf[k + a.Hz], bzp = (bp(c + sz) - bc) * a.rdzf[k + 1 + a.Hz];   // ccf k, k+1
      const double wx_bx = xz_dxw * a.Dx * (0.5 * (bxm + bxp));
      const double wy_by = yz_dyw * a.Dy * (0.5 * (bym + byp));
      const double wz_bz = dzw * amd_Dz(a, k) * (0.5 * (bzm + bzp));
      Cb_zeta = T.Cb * (wx_bx + wy_by + wz_bz) / amd_Dz(a, k);
    }
    nus = -Cnu * d2 * ((r1 + r2 + r3) - Cb_zeta) / q;
  }
  nu[c] = fmax(0.0, nus);
  const int nt = NT >= 0 ? NT : T.n;
  if (nt == 0) return;
  const double xz_dyw = amd_Ixz(a, c, k, ndyw);   // cy_uy interpolates norm_dy_w with I_xz, as written (:326)
#pragma unroll
  for (int t = 0; t < nt; ++t) {
    const double* __restrict__ q_ = T.q[t];
    // normalised tracer gradients at fcc / cfc / ccf
    auto nx = [&](long p) { return a.Dx * ((q_[p] - q_[p - 1]) * a.rdx); };
    auto ny = [&](long p) { return a.Dy * ((q_[p] - q_[p - sy]) * a.rdy); };
    auto nz = [&](long p, int kk) { return amd_Dz(a, kk) * ((q_[p] - q_[p - sz]) * a.rdzf[kk + a.Hz]); };
    const double x0 = nx(c), x1 = nx(c + 1), y0 = ny(c), y1 = ny(c + sy), z0 = nz(c, k), z1 = nz(c + sz, k + 1);
    const double Ix_c = 0.5 * (x0 + x1), Iy_c = 0.5 * (y0 + y1), Iz_c = 0.5 * (z0 + z1);
    const double Ix_c2 = 0.5 * (sq(x0) + sq(x1)), Iy_c2 = 0.5 * (sq(y0) + sq(y1)), Iz_c2 = 0.5 * (sq(z0) + sq(z1));
    const double sigma = Ix_c2 + Iy_c2 + Iz_c2;
    double ks = 0.0;
    if (sigma != 0.0) {
      const double cx = dxu * Ix_c2 + xy_dxv * Ix_c * Iy_c + xz_dxw * Ix_c * Iz_c;
      const double cy = xy_dyu * Iy_c * Ix_c + dyv * Iy_c2 + xz_dyw * Iy_c * Iz_c;
      const double cz = xz_dzu * Iz_c * Ix_c + yz_dzv * Iz_c * Iy_c + dzw * Iz_c2;
      ks = -T.Ck[t] * d2 * (cx + cy + cz) / sigma;
    }
    T.kap[t][c] = fmax(0.0, ks);
  }
}

// calculate_diffusivities!(diffusivity_fields, closure::AMD, model)  (anisotropic_minimum_dissipation.jl:180-205)
// per-level factors of the AMD predictors (filter widths 2 dz_c(k) :213-226, their ratios, 1/dz, delta^2 :150)
int amd_build_table(ocn_model* m) {
  const ocn_grid* g = m->g;
  const GridDev& gd = m->gd;
  const int H = gd.Hz, L = gd.Nz + 2 * H + 1;
  auto dzc = [&](int k) { return g->z_regular ? gd.dz : g->h_dzc[k + H]; };
  auto dzf = [&](int k) { return g->z_regular ? gd.dz : g->h_dzf[k + H + 1]; };
  const double Dx = 2.0 * gd.dx, Dy = 2.0 * gd.dy;
  std::vector<double> t(8 * (size_t)L);
  for (int k = -H; k <= gd.Nz + H; ++k) {
    const int e = k + H;
    const double Dz = 2.0 * dzc(k);
    t[e] = Dz;
    t[L + e] = Dx / Dz;
    t[2 * L + e] = Dz / Dx;
    t[3 * L + e] = Dy / Dz;
    t[4 * L + e] = Dz / Dy;
    t[5 * L + e] = 1.0 / dzf(k);
    t[6 * L + e] = 1.0 / dzc(k);
    const double ix = 1.0 / (Dx * Dx), iy = 1.0 / (Dy * Dy), iz = 1.0 / (Dz * Dz);
    t[7 * L + e] = 3.0 / (ix + iy + iz);
  }
  if (hipMalloc((void**)&m->amd_tab, t.size() * sizeof(double)) != hipSuccess) return OCN_ENOMEM;
  hipMemcpy(m->amd_tab, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice);
  return OCN_OK;
}

void launch_amd(ocn_model* m) {
  ProfScope ps(m->ctx, "amd_diffusivities");
  const GridDev& g = m->gd;
  AmdCtx a;
  a.g = g;
  a.u = m->u.interior(); a.v = m->v.interior(); a.w = m->w.interior();
  a.sy = g.sy; a.sz = g.sz; a.rdx = g.rdx; a.rdy = g.rdy;
  a.Dx = 2.0 * g.dx; a.Dy = 2.0 * g.dy;
  a.xy = a.Dx / a.Dy; a.yx = a.Dy / a.Dx;
  const int L = g.Nz + 2 * g.Hz + 1;
  const double* tab = m->amd_tab;
  a.Dz = tab; a.xz = tab + L; a.zx = tab + 2 * L; a.yz = tab + 3 * L; a.zy = tab + 4 * L; a.rdzf = tab + 5 * L;
  a.rdzc = tab + 6 * L; a.d2 = tab + 7 * L;
  a.Hz = g.Hz;
  static const dim3 b = tuned_block("OCNHIP_AMD_BLOCK", dim3(64, 4, 1));
  const dim3 gr = grid3(g, b);
  AmdTracers T;
  T.n = m->nt;
  for (int t = 0; t < m->nt; ++t) {
    T.q[t] = m->tr[t].interior();
    T.kap[t] = m->kappa_e[t].interior();
    T.Ck[t] = m->d.amd_Ckappa[t];
  }
  T.has_Cb = T.lin = 0; T.b0 = T.b1 = 0; T.Cb = T.Cg = T.cb0 = T.cb1 = 0.0;
  if (m->d.amd_has_Cb && m->d.buoyancy != OCN_BUOYANCY_NONE) {
    T.has_Cb = 1;
    T.Cb = m->d.amd_Cb;
    if (m->d.buoyancy == OCN_BUOYANCY_TRACER) {
      T.b0 = T.b1 = m->d.b_index;
    } else {
      T.lin = 1;
      T.b0 = m->d.T_index;
      T.b1 = m->d.S_index;
      T.Cg = m->d.g;
      T.cb0 = m->d.alpha;
      T.cb1 = m->d.beta;
    }
  }
  if (m->nt == 1) ocn_launch(k_amd_all<1>, gr, b, m->ctx->stream, a, m->d.amd_Cnu, m->nu_e.interior(), T);
  else if (m->nt == 2) ocn_launch(k_amd_all<2>, gr, b, m->ctx->stream, a, m->d.amd_Cnu, m->nu_e.interior(), T);
  else ocn_launch(k_amd_all<-1>, gr, b, m->ctx->stream, a, m->d.amd_Cnu, m->nu_e.interior(), T);
}
