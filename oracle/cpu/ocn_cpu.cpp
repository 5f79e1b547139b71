// ocn_cpu.cpp -- C++/OpenMP restatement of the reference's CPU() path for BASELINE config 2
// (triply periodic RectilinearGrid, NonhydrostaticModel, WENO5 with Z weights, QuasiAdamsBashforth2, FFT Poisson).
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (oracle/): it is the `cpu_baseline` leg of bench.py and a second, independent
// restatement the NumPy oracle is cross-checked against (tests/test_oracle_cpu.py).  The product never links or calls it.
//
// It keeps the reference's kernel structure -- one loop nest per tendency, every cell evaluating its own six face
// fluxes with both biased reconstructions, the textbook weight formulas with their divisions, a complex-to-complex
// 3-D FFT on a complex work array, separate store / fill / correct passes -- so that its timing says what that
// structure costs on this host.  Paths relative to /root/reference/src:
//   TimeSteppers/quasi_adams_bashforth_2.jl:70-104,116-166; store_tendencies.jl:14-36
//   Models/NonhydrostaticModels/calculate_nonhydrostatic_tendencies.jl:155-170; nonhydrostatic_tendency_kernel_functions.jl:44-181
//   Advection/momentum_advection_operators.jl:52-86; upwind_biased_advective_fluxes.jl:10-97
//   Advection/weno_fifth_order.jl:12-19,240-272,311-317,380-403,518-524; centered_fourth_order.jl:17-33
//   Models/NonhydrostaticModels/solve_for_pressure.jl:15-18,55-71; pressure_correction.jl:10-40
//   Solvers/fft_based_poisson_solver.jl:93-125; poisson_eigenvalues.jl:8-16
//   BoundaryConditions/fill_halo_regions_periodic.jl:15-65; set_nonhydrostatic_model.jl:45-58
#include <omp.h>

#include <chrono>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {
constexpr int H = 3;
typedef std::complex<double> cplx;

struct Grid {
  int Nx, Ny, Nz;
  long sy, sz, n;
  double dx, dy, dz;
  long id(int i, int j, int k) const { return (i + H) + sy * (j + H) + sz * (k + H); }
};

// ---- fill_halo_regions! for (Periodic, Periodic, Periodic): x, then y, then z over the full parent extent -------------
void fill_halos(const Grid& g, double* f) {
  const int Tx = g.Nx + 2 * H, Ty = g.Ny + 2 * H, Tz = g.Nz + 2 * H;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < Tz; ++k)
    for (int j = 0; j < Ty; ++j) {
      double* r = f + g.sy * j + g.sz * k;
      for (int h = 0; h < H; ++h) {
        r[h] = r[g.Nx + h];
        r[g.Nx + H + h] = r[H + h];
      }
    }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < Tz; ++k)
    for (int h = 0; h < H; ++h) {
      double* a = f + g.sz * k;
      memcpy(a + g.sy * h, a + g.sy * (g.Ny + h), sizeof(double) * Tx);
      memcpy(a + g.sy * (g.Ny + H + h), a + g.sy * (H + h), sizeof(double) * Tx);
    }
  for (int h = 0; h < H; ++h) {
    memcpy(f + g.sz * h, f + g.sz * (g.Nz + h), sizeof(double) * g.sz);
    memcpy(f + g.sz * (g.Nz + H + h), f + g.sz * (H + h), sizeof(double) * g.sz);
  }
  (void)Ty;
}

// ---- stencils ---------------------------------------------------------------------------------------------------------
inline double sq(double x) { return x * x; }
// f - delta(delta f)/6 (centered_fourth_order.jl:17-24)
inline double i3(const double* p, long s) { return p[0] - ((p[s] - p[0]) - (p[0] - p[-s])) / 6; }
// symmetric 4th-order interpolation midway between p[0] and p[s]  (:26-33)
inline double sym4(const double* p, long s) { return 0.5 * (i3(p, s) + i3(p + s, s)); }

inline void weights(double b0, double b1, double b2, double C0, double C1, double C2, double& w0, double& w1, double& w2) {
  const double eps = 1e-6;                                   // weno_fifth_order.jl:19
  const double tau = std::fabs(b2 - b0);                     // Z weights (:167 default, :380-391)
  double a0 = C0 * (1 + sq(tau / (b0 + eps)));
  double a1 = C1 * (1 + sq(tau / (b1 + eps)));
  double a2 = C2 * (1 + sq(tau / (b2 + eps)));
  const double sa = a0 + a1 + a2;
  w0 = a0 / sa;
  w1 = a1 / sa;
  w2 = a2 / sa;
}
// left-biased reconstruction at the face between p[-s] and p[0]  (:266-268,311-313,518-520)
inline double weno_left(const double* p, long s) {
  const double a3 = p[-3 * s], a2 = p[-2 * s], a1 = p[-s], a0 = p[0], b1 = p[s];
  const double B0 = 13.0 / 12 * sq(a1 - 2 * a0 + b1) + 0.25 * sq(3 * a1 - 4 * a0 + b1);
  const double B1 = 13.0 / 12 * sq(a2 - 2 * a1 + a0) + 0.25 * sq(a2 - a0);
  const double B2 = 13.0 / 12 * sq(a3 - 2 * a2 + a1) + 0.25 * sq(a3 - 4 * a2 + 3 * a1);
  double w0, w1, w2;
  weights(B0, B1, B2, 3.0 / 10, 3.0 / 5, 1.0 / 10, w0, w1, w2);
  const double p0 = 1.0 / 3 * a1 + 5.0 / 6 * a0 - 1.0 / 6 * b1;
  const double p1 = -1.0 / 6 * a2 + 5.0 / 6 * a1 + 1.0 / 3 * a0;
  const double p2 = 1.0 / 3 * a3 - 7.0 / 6 * a2 + 11.0 / 6 * a1;
  return w0 * p0 + w1 * p1 + w2 * p2;
}
// right-biased, smoothness indicators as written (:270-272,315-317,368,522-524)
inline double weno_right(const double* p, long s) {
  const double a2 = p[-2 * s], a1 = p[-s], a0 = p[0], b1 = p[s], b2 = p[2 * s];
  const double B0 = 13.0 / 12 * sq(a0 - 2 * b1 + b2) + 0.25 * sq(a0 - 4 * b1 + 3 * b2);
  const double B1 = 13.0 / 12 * sq(a1 - 2 * a0 + b1) + 0.25 * sq(a1 - b1);
  const double B2 = 13.0 / 12 * sq(a2 - 2 * a1 + a0) + 0.25 * sq(3 * a2 - 4 * a1 + a0);
  double w0, w1, w2;
  weights(B0, B1, B2, 1.0 / 10, 3.0 / 5, 3.0 / 10, w0, w1, w2);
  const double p0 = 11.0 / 6 * a0 - 7.0 / 6 * b1 + 1.0 / 3 * b2;
  const double p1 = 1.0 / 3 * a1 + 5.0 / 6 * a0 - 1.0 / 6 * b1;
  const double p2 = -1.0 / 6 * a2 + 5.0 / 6 * a1 + 1.0 / 3 * a0;
  return w0 * p0 + w1 * p1 + w2 * p2;
}
// upwind_biased_product (upwind_biased_advective_fluxes.jl:10)
inline double upw(double ut, double L, double R) { return ((ut + std::fabs(ut)) * L + (ut - std::fabs(ut)) * R) / 2; }
// advective flux per unit area (upwind_biased_product already carries the advecting velocity) through the face between q[-s] and q[0]
inline double flux_face(double ut, const double* q, long s) { return upw(ut, weno_left(q, s), weno_right(q, s)); }
// centre form: the reconstruction point is the centre above q[0] (face form shifted by +1, :248-263)
inline double flux_center(double ut, const double* q, long s) { return upw(ut, weno_left(q + s, s), weno_right(q + s, s)); }

// ---- calculate_Gu! / Gv! / Gw!: -div_Uu etc., every cell evaluates its own fluxes --------------------------------------
void tendencies(const Grid& g, const double* u, const double* v, const double* w, double* Gu, double* Gv, double* Gw) {
  const long sx = 1, sy = g.sy, sz = g.sz;
  const double Ax = g.dy * g.dz, Ay = g.dx * g.dz, Az = g.dx * g.dy, rV = 1.0 / (g.dx * g.dy * g.dz);
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < g.Nz; ++k)
    for (int j = 0; j < g.Ny; ++j)
      for (int i = 0; i < g.Nx; ++i) {
        const long c = g.id(i, j, k);
        {  // u at (f, c, c)
          auto FUu = [&](long p) { return flux_center(sym4(u + p, sx), u + p, sx); };   // at centre of p
          auto FVu = [&](long p) { double vt = sym4(v + p - sx, sx); return flux_face(vt, u + p, sy); };   // (f, f, c)
          auto FWu = [&](long p) { double wt = sym4(w + p - sx, sx); return flux_face(wt, u + p, sz); };   // (f, c, f)
          Gu[c] = -rV * (Ax * (FUu(c) - FUu(c - sx)) + Ay * (FVu(c + sy) - FVu(c)) + Az * (FWu(c + sz) - FWu(c)));
        }
        {  // v at (c, f, c)
          auto FUv = [&](long p) { double ut = sym4(u + p - sy, sy); return flux_face(ut, v + p, sx); };
          auto FVv = [&](long p) { return flux_center(sym4(v + p, sy), v + p, sy); };
          auto FWv = [&](long p) { double wt = sym4(w + p - sy, sy); return flux_face(wt, v + p, sz); };
          Gv[c] = -rV * (Ax * (FUv(c + sx) - FUv(c)) + Ay * (FVv(c) - FVv(c - sy)) + Az * (FWv(c + sz) - FWv(c)));
        }
        {  // w at (c, c, f)
          auto FUw = [&](long p) { double ut = sym4(u + p - sz, sz); return flux_face(ut, w + p, sx); };
          auto FVw = [&](long p) { double vt = sym4(v + p - sz, sz); return flux_face(vt, w + p, sy); };
          auto FWw = [&](long p) { return flux_center(sym4(w + p, sz), w + p, sz); };
          Gw[c] = -rV * (Ax * (FUw(c + sx) - FUw(c)) + Ay * (FVw(c + sy) - FVw(c)) + Az * (FWw(c) - FWw(c - sz)));
        }
      }
}

// ---- complex radix-2 FFT (sizes are powers of two), unnormalised forward, 1/N on the inverse ---------------------------
struct FFT1 {
  int n;
  std::vector<cplx> tw;
  std::vector<int> rev;
  explicit FFT1(int n_) : n(n_), tw(n_ / 2 > 0 ? n_ / 2 : 1), rev(n_) {
    for (int i = 0; i < n / 2; ++i) tw[i] = std::polar(1.0, -2 * M_PI * i / n);
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    for (int i = 0; i < n; ++i) {
      int r = 0;
      for (int b = 0; b < lg; ++b)
        if (i & (1 << b)) r |= 1 << (lg - 1 - b);
      rev[i] = r;
    }
  }
  void run(cplx* a, bool inverse) const {
    for (int i = 0; i < n; ++i)
      if (i < rev[i]) std::swap(a[i], a[rev[i]]);
    for (int len = 2; len <= n; len <<= 1) {
      const int half = len / 2, step = n / len;
      for (int s = 0; s < n; s += len)
        for (int q = 0; q < half; ++q) {
          cplx t = inverse ? std::conj(tw[q * step]) : tw[q * step];
          cplx x = a[s + q], y = a[s + q + half] * t;
          a[s + q] = x + y;
          a[s + q + half] = x - y;
        }
    }
    if (inverse)
      for (int i = 0; i < n; ++i) a[i] /= n;
  }
};

struct Poisson {
  Grid g;
  FFT1 fx, fy, fz;
  std::vector<double> lx, ly, lz;
  std::vector<cplx> st;   // `storage`: complex Nx x Ny x Nz (fft_based_poisson_solver.jl:63)
  explicit Poisson(const Grid& g_) : g(g_), fx(g_.Nx), fy(g_.Ny), fz(g_.Nz), st((size_t)g_.Nx * g_.Ny * g_.Nz) {
    auto eig = [](int N, double d, std::vector<double>& l) {   // poisson_eigenvalues.jl:8-11
      l.resize(N);
      for (int i = 0; i < N; ++i) l[i] = sq(2 * std::sin(i * M_PI / N) / d);
    };
    eig(g.Nx, g.dx, lx);
    eig(g.Ny, g.dy, ly);
    eig(g.Nz, g.dz, lz);
  }
  void transform(bool inverse) {
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
#pragma omp parallel
    {
      std::vector<cplx> line(std::max(Ny, Nz));
#pragma omp for collapse(2) schedule(static)
      for (int k = 0; k < Nz; ++k)
        for (int j = 0; j < Ny; ++j) fx.run(&st[(size_t)Nx * (j + (size_t)Ny * k)], inverse);
#pragma omp for collapse(2) schedule(static)
      for (int k = 0; k < Nz; ++k)
        for (int i = 0; i < Nx; ++i) {
          for (int j = 0; j < Ny; ++j) line[j] = st[i + (size_t)Nx * (j + (size_t)Ny * k)];
          fy.run(line.data(), inverse);
          for (int j = 0; j < Ny; ++j) st[i + (size_t)Nx * (j + (size_t)Ny * k)] = line[j];
        }
#pragma omp for collapse(2) schedule(static)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          for (int k = 0; k < Nz; ++k) line[k] = st[i + (size_t)Nx * (j + (size_t)Ny * k)];
          fz.run(line.data(), inverse);
          for (int k = 0; k < Nz; ++k) st[i + (size_t)Nx * (j + (size_t)Ny * k)] = line[k];
        }
    }
  }
  // solve!(phi, solver, b): fft_based_poisson_solver.jl:93-120
  void solve() {
    transform(false);
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          cplx& x = st[i + (size_t)Nx * (j + (size_t)Ny * k)];
          x = -x / (lx[i] + ly[j] + lz[k]);
        }
    st[0] = 0;
    transform(true);
  }
};

struct Model {
  Grid g;
  std::vector<double> u, v, w, p, Gu, Gv, Gw, Mu, Mv, Mw;
  Poisson ps;
  double previous_dt = INFINITY;
  explicit Model(const Grid& g_) : g(g_), u(g_.n), v(g_.n), w(g_.n), p(g_.n), Gu(g_.n), Gv(g_.n), Gw(g_.n), Mu(g_.n), Mv(g_.n), Mw(g_.n), ps(g_) {}
  void fill_uvw() {
    fill_halos(g, u.data());
    fill_halos(g, v.data());
    fill_halos(g, w.data());
  }
  // calculate_pressure_correction! + pressure_correct_velocities!  (pressure_correction.jl:10-40)
  void pressure_correct(double dt) {
    fill_uvw();
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          const long c = g.id(i, j, k);
          const double div = (u[c + 1] - u[c]) / g.dx + (v[c + g.sy] - v[c]) / g.dy + (w[c + g.sz] - w[c]) / g.dz;
          ps.st[i + (size_t)Nx * (j + (size_t)Ny * k)] = div / dt;     // solve_for_pressure.jl:15-18
        }
    ps.solve();
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) p[g.id(i, j, k)] = ps.st[i + (size_t)Nx * (j + (size_t)Ny * k)].real();   // copy_real_component!
    fill_halos(g, p.data());
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          const long c = g.id(i, j, k);
          u[c] -= (p[c] - p[c - 1]) / g.dx * dt;
          v[c] -= (p[c] - p[c - g.sy]) / g.dy * dt;
          w[c] -= (p[c] - p[c - g.sz]) / g.dz * dt;
        }
  }
  // time_step!(model, dt)  (quasi_adams_bashforth_2.jl:70-104)
  void time_step(double dt) {
    const bool euler = dt != previous_dt;
    const double chi = euler ? -0.5 : 0.1;
    if (euler) {
      std::fill(Mu.begin(), Mu.end(), 0.0);
      std::fill(Mv.begin(), Mv.end(), 0.0);
      std::fill(Mw.begin(), Mw.end(), 0.0);
    }
    previous_dt = dt;
    tendencies(g, u.data(), v.data(), w.data(), Gu.data(), Gv.data(), Gw.data());
    const double cn = 1.5 + chi, cm = 0.5 + chi;
    const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)               // ab2_step_field! (:158-166)
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          const long c = g.id(i, j, k);
          u[c] += dt * (cn * Gu[c] - cm * Mu[c]);
          v[c] += dt * (cn * Gv[c] - cm * Mv[c]);
          w[c] += dt * (cn * Gw[c] - cm * Mw[c]);
        }
    pressure_correct(dt);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Nz; ++k)               // store_tendencies!
      for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
          const long c = g.id(i, j, k);
          Mu[c] = Gu[c];
          Mv[c] = Gv[c];
          Mw[c] = Gw[c];
        }
    fill_uvw();                                // update_state!
  }
};

void put(const Grid& g, const double* in, std::vector<double>& f) {
  for (int k = 0; k < g.Nz; ++k)
    for (int j = 0; j < g.Ny; ++j)
      for (int i = 0; i < g.Nx; ++i) f[g.id(i, j, k)] = in[i + (size_t)g.Nx * (j + (size_t)g.Ny * k)];
}
void get(const Grid& g, const std::vector<double>& f, double* out) {
  for (int k = 0; k < g.Nz; ++k)
    for (int j = 0; j < g.Ny; ++j)
      for (int i = 0; i < g.Nx; ++i) out[i + (size_t)g.Nx * (j + (size_t)g.Ny * k)] = f[g.id(i, j, k)];
}
}  // namespace

extern "C" {
// u, v, w: interior arrays (Nx, Ny, Nz), x fastest, in / out.  set!(model; u, v, w) with its projection, then `nsteps`
// time_step!(model, dt); *seconds = wall time of the steps only.  Sizes must be powers of two.  Returns 0 on success.
int ocncpu_run(int Nx, int Ny, int Nz, double Lx, double Ly, double Lz, double* u, double* v, double* w, double* p,
               double dt, int nsteps, int threads, double* seconds) {
  for (int n : {Nx, Ny, Nz})
    if (n < 8 || (n & (n - 1))) return -1;
  if (threads > 0) omp_set_num_threads(threads);
  Grid g;
  g.Nx = Nx; g.Ny = Ny; g.Nz = Nz;
  g.sy = Nx + 2 * H;
  g.sz = g.sy * (Ny + 2 * H);
  g.n = g.sz * (Nz + 2 * H);
  g.dx = Lx / Nx; g.dy = Ly / Ny; g.dz = Lz / Nz;
  Model m(g);
  put(g, u, m.u);
  put(g, v, m.v);
  put(g, w, m.w);
  m.fill_uvw();               // set!: update_state!, projection with dt = 1, update_state! (set_nonhydrostatic_model.jl:45-58)
  m.pressure_correct(1.0);
  m.fill_uvw();
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < nsteps; ++s) m.time_step(dt);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  get(g, m.u, u);
  get(g, m.v, v);
  get(g, m.w, w);
  if (p) get(g, m.p, p);
  return 0;
}
// calculate_G{u,v,w}! alone (tests): interior arrays in, interior tendencies out
int ocncpu_tendencies(int Nx, int Ny, int Nz, double Lx, double Ly, double Lz, const double* u, const double* v, const double* w,
                      double* Gu, double* Gv, double* Gw) {
  Grid g;
  g.Nx = Nx; g.Ny = Ny; g.Nz = Nz;
  g.sy = Nx + 2 * H;
  g.sz = g.sy * (Ny + 2 * H);
  g.n = g.sz * (Nz + 2 * H);
  g.dx = Lx / Nx; g.dy = Ly / Ny; g.dz = Lz / Nz;
  std::vector<double> a(g.n), b(g.n), c(g.n), ga(g.n), gb(g.n), gc(g.n);
  put(g, u, a);
  put(g, v, b);
  put(g, w, c);
  fill_halos(g, a.data());
  fill_halos(g, b.data());
  fill_halos(g, c.data());
  tendencies(g, a.data(), b.data(), c.data(), ga.data(), gb.data(), gc.data());
  get(g, ga, Gu);
  get(g, gb, Gv);
  get(g, gc, Gw);
  return 0;
}
int ocncpu_max_threads(void) { return omp_get_max_threads(); }
}
