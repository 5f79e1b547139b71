// poisson.hip -- pressure Poisson solvers on the device.
//
// Reference algorithm (paths relative to /root/reference/src):
//   Solvers/fft_based_poisson_solver.jl:93-120   : forward transforms, phi^ = -b^/(lx+ly+lz), zero mode := 0, backward
//   Solvers/poisson_eigenvalues.jl:8-31          : lambda = (2 sin(pi (i-1)/N) / dx)^2  (Periodic)
//   Solvers/fourier_tridiagonal_poisson_solver.jl:16-101 + batched_tridiagonal_solver.jl:91-122
//                                                : x,y transforms, Thomas solve in z per (i,j), mean removal
//
// MI355X design: the right-hand side is real, so real-to-complex transforms on the half spectrum
// (Nx/2+1) x Ny x Nz replace the reference's complex-to-complex transforms on complex storage
// (half the bytes per pass).  A Bounded z direction -- regular or stretched -- always uses the
// Fourier-x/y + tridiagonal-z algorithm: on a regular grid it solves exactly the same discrete system
// as the reference's cosine-transform path (same operator, same zero-mean gauge), without a DCT.
#include "internal.h"
#ifndef OCN_HOST_EMU
// transform launches are skipped while a step is replayed from a hipGraph (compat.h g_ocn_dry)
#define dry_hipfftExecD2Z(...) (g_ocn_dry ? HIPFFT_SUCCESS : hipfftExecD2Z(__VA_ARGS__))
#define dry_hipfftExecZ2D(...) (g_ocn_dry ? HIPFFT_SUCCESS : hipfftExecZ2D(__VA_ARGS__))
#define dry_hipfftExecZ2Z(...) (g_ocn_dry ? HIPFFT_SUCCESS : hipfftExecZ2Z(__VA_ARGS__))
#endif


#ifndef OCN_HOST_EMU
#include <rocblas/rocblas.h>
#endif
#include <complex>
typedef std::complex<double> cplx;

struct double2_ {
  double x, y;
};

struct PoissonSolver {
  int kind;        // 5: y-slabs with a Bounded z (run_yslab)
                   // 4: Bounded / Flat x or y (dense cosine / Fourier transforms, see run_general)
                   // 0: 3-D FFT (z Periodic)   1: 2-D FFT (+ tridiagonal if z Bounded, plain divide if z Flat)
                   // 2: z-slabs: 2-D FFT per plane, all-to-all to ky-slabs, 1-D FFT along z, and back
  int Nx, Ny, Nz, Nxh;
  int R = 1, rank = 0, Nzg = 0, Nyl = 0;   // slab decomposition
  double2_ *ta = nullptr, *tb = nullptr;   // transpose buffers (same size as spec)
  void* zs = nullptr;                      // fused z-transform + eigenvalue division + inverse (zfft.hip), Nz == 256
  void* zsl = nullptr;                     // slab runs: transpose-free z stage (zslab.hip)
  double dz2 = 0;
  bool cxy = false;                        // custom x / y passes (zfft.hip): rhs fused into the x transform
  bool cxy_bz = false;                     // Bounded z: the same passes around the tridiagonal sweeps (poisson_run_from_predictor_bz)
  // z-slabs, Green's-function z stage, custom x / y passes: the w* plane of this rank's first level is transformed here and
  // enters the convolution as a source one level below the slab (zslab.hip `bel`) -- the lower neighbour never needs the plane
  // Both planes live in "plane Nz" of the solver's arrays (spec and rhs hold Nz + 1 planes on slab runs): the x / y passes
  // simply run over one plane more, k_zslab_below overwrites the source plane with the solution plane once the sweeps are done.
  double2_* bplane = nullptr;              // spec + ncol Nz: spectrum of w*[level 0] / (dz dt), later of the solution below the slab
  bool local_phi = false;                  // that second use is on (OCNHIP_PHI_EXCHANGE=1 keeps the exchange)
  void* tw = nullptr;                      // twiddle holder for the custom passes
  double* rhs = nullptr;      // real (Nx,Ny,Nz)
  double2_* spec = nullptr;   // complex (Nxh,Ny,Nz)
  double* tscr = nullptr;     // Thomas factors: t (ncol, Nz) then 1/beta (ncol, Nz), built once (k_tridiag_setup)
  int* tri_kbr = nullptr;     // per column: level of the reference's early break (Nz: none)
  bool tri_ready = false;
  double *lx = nullptr, *ly = nullptr, *lz = nullptr;  // eigenvalues on the device
  // kind 4 (a Bounded or Flat x / y direction), see run_walls
  bool tr = false;                 // solved on the x <-> y transposed array (x Bounded / Flat with y Periodic)
  int gNx = 0, gNy = 0, gR = 0;    // sizes after the swap; gR: x extent of the complex array (Nx/2+1 if x is Periodic)
  int gtopo[2] = {0, 0};
  double *ra = nullptr, *rb = nullptr;      // real (gNx, gNy, Nz) ping-pong
  double2_ *ga = nullptr, *gb = nullptr;    // complex (gR, gNy, Nz) ping-pong
  double* gm[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // cosine-transform matrices [x / y][forward / inverse]
  void* blas = nullptr;                     // rocblas_handle
  // kind 5 (y-slabs, Bounded z), see run_yslab
  int yw = 0, Nyg = 0;                      // kx columns per rank (padded), global Ny
  double2_ *ysend = nullptr, *yrecv = nullptr, *yT = nullptr;
#ifndef OCN_HOST_EMU
  hipfftHandle fwd = 0, inv = 0, zplan = 0, xinv = 0;
  hipfftHandle wxf = 0, wxi = 0, wz = 0;    // kind 4: batched 1-D x (R2C / C2R) and z (C2C) plans
  hipfftHandle yfft = 0;                    // kind 5: contiguous 1-D complex transforms along the global y
#endif
};

double* poisson_rhs_buffer(PoissonSolver* s) { return s->rhs; }

static std::vector<double> eigenvalues_periodic(int N, double L) {
  // poisson_eigenvalues.jl:8-11
  std::vector<double> l(N);
  for (int i = 0; i < N; ++i) {
    double s = 2.0 * sin(i * M_PI / N) / (L / N);
    l[i] = s * s;
  }
  return l;
}

static std::vector<double> eigenvalues(int topo, int N, double L) {
  // poisson_eigenvalues.jl:8-31: Periodic (2 sin(pi i / N) / d)^2, Bounded (2 sin(pi i / 2N) / d)^2, Flat 0
  if (topo == OCN_PERIODIC) return eigenvalues_periodic(N, L);
  std::vector<double> l(N, 0.0);
  if (topo == OCN_BOUNDED)
    for (int i = 0; i < N; ++i) {
      double s = 2.0 * sin(i * M_PI / (2.0 * N)) / (L / N);
      l[i] = s * s;
    }
  return l;
}

// Cosine-transform matrices of a Bounded direction, column-major N x N (discrete_transforms.jl:140-161 uses
// FFTW's REDFT10 / REDFT01, which are 2x / 1x these and normalised by 1 / 2N; the composition is the same):
//   forward  F(k, n) = cos(pi (n + 1/2) k / N)                       (DCT-II)
//   inverse  I(n, k) = (k == 0 ? 1 : 2) cos(pi (n + 1/2) k / N) / N  (DCT-III), I F = identity
static void cosine_matrices(int N, std::vector<double>& F, std::vector<double>& I) {
  F.assign((size_t)N * N, 0.0);
  I.assign((size_t)N * N, 0.0);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < N; ++k) {
      double c = cos(M_PI * (n + 0.5) * k / N);
      F[k + (size_t)n * N] = c;
      I[n + (size_t)k * N] = (k == 0 ? 1.0 : 2.0) * c / N;
    }
}

static double* upload(const std::vector<double>& v) {
  double* d = nullptr;
  if (hipMalloc((void**)&d, v.size() * sizeof(double)) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
  return d;
}

static int walls_create(ocn_model* m, PoissonSolver* s);
static int yslab_create(ocn_model* m, PoissonSolver* s);

PoissonSolver* poisson_create(ocn_model* m) {
  ocn_grid* g = m->g;
  PoissonSolver* s = new PoissonSolver;
  s->Nx = g->N[0];
  s->Ny = g->N[1];
  s->Nz = g->N[2];
  s->Nxh = s->Nx / 2 + 1;
  s->kind = (g->topo[2] == OCN_PERIODIC) ? 0 : 1;
  if (g->dist_y) {
    s->kind = 5;
    int rc5 = yslab_create(m, s);
    if (rc5) {
      poisson_destroy(s);
      return nullptr;
    }
    return s;
  }
  if (g->topo[0] != OCN_PERIODIC || g->topo[1] != OCN_PERIODIC) {
    s->kind = 4;
    int rc4 = walls_create(m, s);
    if (rc4) {
      poisson_destroy(s);
      return nullptr;
    }
    return s;
  }
  if (g->dist) {
    s->kind = 2;
    s->R = m->ctx->nranks;
    s->rank = m->ctx->rank;
    s->Nzg = g->Nzg;
    s->Nyl = s->Ny / s->R;
  }
  size_t nr = (size_t)s->Nx * s->Ny * s->Nz, nc = (size_t)s->Nxh * s->Ny * s->Nz;
  const size_t xr = g->dist ? (size_t)s->Nx * s->Ny : 0, xc = g->dist ? (size_t)s->Nxh * s->Ny : 0;   // one plane more on z-slabs
  if (hipMalloc((void**)&s->rhs, (nr + xr) * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&s->spec, (nc + xc) * sizeof(double2_)) != hipSuccess) {
    poisson_destroy(s);
    return nullptr;
  }
  if (g->topo[2] == OCN_BOUNDED) {
    if (hipMalloc((void**)&s->tscr, 2 * (size_t)s->Nxh * s->Ny * s->Nz * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&s->tri_kbr, (size_t)s->Nxh * s->Ny * sizeof(int)) != hipSuccess) {
      poisson_destroy(s);
      return nullptr;
    }
  }
  const char* dsolver = getenv("OCNHIP_DIST_SOLVER");
  const bool use_slab = g->dist && !(dsolver && strcmp(dsolver, "transpose") == 0);
  if (use_slab) {
    std::vector<double> lxh = eigenvalues_periodic(s->Nx, g->L[0]);
    lxh.resize(s->Nxh);
    s->zsl = zslab_create(m->ctx, lxh, eigenvalues_periodic(s->Ny, g->L[1]), s->Nz, s->R, s->rank);
    if (!s->zsl) {
      poisson_destroy(s);
      return nullptr;
    }
    double dz = g->L[2] / g->Nzg;
    s->dz2 = dz * dz;
  }
  const bool want_zs = !use_slab && g->topo[2] == OCN_PERIODIC && zsolve_size_ok(g->dist ? g->Nzg : s->Nz) &&
                       !(getenv("OCNHIP_NO_ZSOLVE") && atoi(getenv("OCNHIP_NO_ZSOLVE")) != 0);
  if (want_zs) {
    std::vector<double> lxh = eigenvalues_periodic(s->Nx, g->L[0]);
    lxh.resize(s->Nxh);
    std::vector<double> lyv = eigenvalues_periodic(s->Ny, g->L[1]);
    if (g->dist) lyv = std::vector<double>(lyv.begin() + s->rank * s->Nyl, lyv.begin() + (s->rank + 1) * s->Nyl);
    s->zs = zsolve_create(m->ctx, lxh, lyv);
    if (!s->zs) {
      poisson_destroy(s);
      return nullptr;
    }
    if (s->kind == 0) s->kind = 3;   // 2-D transforms per plane + fused z stage
  }
  // custom x / y passes: 128-, 256- or 512-point transforms in x and y, triply periodic, any z stage
  if (fft_size_ok(s->Nx) && fft_size_ok(s->Ny) && g->topo[2] == OCN_PERIODIC && g->z_regular &&
      !(getenv("OCNHIP_NO_CUSTOM_XY") && atoi(getenv("OCNHIP_NO_CUSTOM_XY")) != 0)) {
    if (!s->zs && !s->zsl) {
      // no fused z kernel for this Nz: the Green's-function z stage works for any Nz (one "slab")
      std::vector<double> lxh = eigenvalues_periodic(s->Nx, g->L[0]);
      lxh.resize(s->Nxh);
      s->zsl = zslab_create(m->ctx, lxh, eigenvalues_periodic(s->Ny, g->L[1]), s->Nz, 1, 0);
      double dz = g->L[2] / g->Nzg;
      s->dz2 = dz * dz;
    }
    std::vector<double> one(1, 0.0);
    s->tw = zsolve_create(m->ctx, one, one);
    s->cxy = s->tw != nullptr && (s->zs || s->zsl);
    if (s->cxy && g->dist && s->zsl && !(getenv("OCNHIP_WSTAR_EXCHANGE") && atoi(getenv("OCNHIP_WSTAR_EXCHANGE")) != 0)) {
      s->bplane = s->spec + nc;
      s->local_phi = !(getenv("OCNHIP_PHI_EXCHANGE") && atoi(getenv("OCNHIP_PHI_EXCHANGE")) != 0);
    }
  }
  // Bounded z (Fourier-tridiagonal solver) with 128-, 256- or 512-point periodic x and y: the same fused right-hand side + x
  // pass and custom y passes in front of and behind the batched Thomas sweeps (round 3: config 3 spent 0.22 ms per stage in
  // k_rhs + four rocFFT kernels where these passes take 0.15)
  if (g->topo[2] == OCN_BOUNDED && s->kind == 1 && fft_size_ok(s->Nx) && fft_size_ok(s->Ny) && g->topo[0] == OCN_PERIODIC &&
      g->topo[1] == OCN_PERIODIC && s->Nx + 2 * g->PH[0] == (int)m->gd.sy &&
      !(getenv("OCNHIP_NO_CUSTOM_XY") && atoi(getenv("OCNHIP_NO_CUSTOM_XY")) != 0)) {
    std::vector<double> one(1, 0.0);
    s->tw = zsolve_create(m->ctx, one, one);
    s->cxy_bz = s->tw != nullptr;
  }
  s->lx = upload(eigenvalues_periodic(s->Nx, g->L[0]));
  s->ly = upload(eigenvalues_periodic(s->Ny, g->L[1]));
  if (g->topo[2] == OCN_PERIODIC) s->lz = upload(eigenvalues_periodic(g->dist ? g->Nzg : s->Nz, g->L[2]));
  if (s->kind == 2 && !s->zsl) {
    if (hipMalloc((void**)&s->ta, nc * sizeof(double2_)) != hipSuccess ||
        hipMalloc((void**)&s->tb, nc * sizeof(double2_)) != hipSuccess) {
      poisson_destroy(s);
      return nullptr;
    }
  }
#ifndef OCN_HOST_EMU
  hipfftResult r1, r2;
  hipfftResult r3 = HIPFFT_SUCCESS;
  if (s->kind == 2 && !s->zs && !s->zsl) {
    // batched 1-D transforms along z of the ky-slab (Nxh, Nyl, Nzg): stride Nxh*Nyl, consecutive batches 1 apart
    int nz[1] = {s->Nzg};
    int st = s->Nxh * s->Nyl;
    r3 = hipfftPlanMany(&s->zplan, 1, nz, nz, st, 1, nz, st, 1, HIPFFT_Z2Z, st);
    if (r3 == HIPFFT_SUCCESS) hipfftSetStream(s->zplan, m->ctx->stream);
  }
  if (r3 != HIPFFT_SUCCESS) {
    ocn_set_error(m->ctx, "hipfft z-plan creation failed (%d)", (int)r3);
    poisson_destroy(s);
    return nullptr;
  }
  if (s->kind == 0) {
    r1 = hipfftPlan3d(&s->fwd, s->Nz, s->Ny, s->Nx, HIPFFT_D2Z);
    r2 = hipfftPlan3d(&s->inv, s->Nz, s->Ny, s->Nx, HIPFFT_Z2D);
  } else {
    int n[2] = {s->Ny, s->Nx};
    r1 = hipfftPlanMany(&s->fwd, 2, n, nullptr, 1, s->Nx * s->Ny, nullptr, 1, s->Nxh * s->Ny, HIPFFT_D2Z, s->Nz);
    r2 = hipfftPlanMany(&s->inv, 2, n, nullptr, 1, s->Nxh * s->Ny, nullptr, 1, s->Nx * s->Ny, HIPFFT_Z2D, s->Nz);
  }
  if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
    ocn_set_error(m->ctx, "hipfft plan creation failed (%d, %d)", (int)r1, (int)r2);
    poisson_destroy(s);
    return nullptr;
  }
  hipfftSetStream(s->fwd, m->ctx->stream);
  hipfftSetStream(s->inv, m->ctx->stream);
  if (s->cxy || s->cxy_bz) {
    int nx[1] = {s->Nx}, ie[1] = {s->Nxh}, oe[1] = {s->Nx};
    const int planes = s->Nz + (s->bplane && s->local_phi ? 1 : 0);    // the plane below the slab rides along
    if (hipfftPlanMany(&s->xinv, 1, nx, ie, 1, s->Nxh, oe, 1, s->Nx, HIPFFT_Z2D, s->Ny * planes) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfft x-inverse plan creation failed");
      poisson_destroy(s);
      return nullptr;
    }
    hipfftSetStream(s->xinv, m->ctx->stream);
  }
#endif
  return s;
}

void poisson_destroy(PoissonSolver* s) {
  if (!s) return;
#ifndef OCN_HOST_EMU
  if (s->fwd) hipfftDestroy(s->fwd);
  if (s->inv) hipfftDestroy(s->inv);
  if (s->zplan) hipfftDestroy(s->zplan);
  if (s->xinv) hipfftDestroy(s->xinv);
#endif
  zsolve_destroy(s->tw);
  hipFree(s->ta);
  hipFree(s->tb);
  hipFree(s->ga);
  hipFree(s->gb);
  hipFree(s->ra);
  hipFree(s->rb);
  hipFree(s->ysend);
  hipFree(s->yrecv);
  hipFree(s->yT);
  for (int d = 0; d < 2; ++d)
    for (int q = 0; q < 2; ++q) hipFree(s->gm[d][q]);
#ifndef OCN_HOST_EMU
  if (s->wxf) hipfftDestroy(s->wxf);
  if (s->wxi) hipfftDestroy(s->wxi);
  if (s->wz) hipfftDestroy(s->wz);
  if (s->yfft) hipfftDestroy(s->yfft);
  if (s->blas) rocblas_destroy_handle((rocblas_handle)s->blas);
#endif
  zsolve_destroy(s->zs);
  zslab_destroy(s->zsl);
  hipFree(s->rhs);
  hipFree(s->spec);
  hipFree(s->tscr);
  hipFree(s->tri_kbr);
  hipFree(s->lx);
  hipFree(s->ly);
  hipFree(s->lz);
  delete s;
}

// ---- host-emulation transforms (naive DFT; tiny grids only) ----------------------------------------------
#ifdef OCN_HOST_EMU
static void emu_dft_axis(std::vector<cplx>& a, int n0, int n1, int n2, int axis, int sign) {
  int n[3] = {n0, n1, n2};
  long st[3] = {1, n0, (long)n0 * n1};
  int N = n[axis];
  std::vector<cplx> line(N), out(N);
  int o1 = (axis + 1) % 3, o2 = (axis + 2) % 3;
  for (int b = 0; b < n[o2]; ++b)
    for (int a_ = 0; a_ < n[o1]; ++a_) {
      long base = a_ * st[o1] + b * st[o2];
      for (int i = 0; i < N; ++i) line[i] = a[base + i * st[axis]];
      for (int k = 0; k < N; ++k) {
        cplx acc = 0;
        for (int i = 0; i < N; ++i) {
          double ang = sign * 2.0 * M_PI * ((long)i * k % N) / N;
          acc += line[i] * cplx(cos(ang), sin(ang));
        }
        out[k] = acc;
      }
      for (int i = 0; i < N; ++i) a[base + i * st[axis]] = out[i];
    }
}
static void emu_forward(PoissonSolver* s) {
  int Nx = s->Nx, Ny = s->Ny, Nz = s->Nz, Nxh = s->Nxh;
  std::vector<cplx> a((size_t)Nx * Ny * Nz);
  for (size_t i = 0; i < a.size(); ++i) a[i] = s->rhs[i];
  emu_dft_axis(a, Nx, Ny, Nz, 0, -1);
  emu_dft_axis(a, Nx, Ny, Nz, 1, -1);
  if (s->kind == 0) emu_dft_axis(a, Nx, Ny, Nz, 2, -1);
  for (int k = 0; k < Nz; ++k)
    for (int j = 0; j < Ny; ++j)
      for (int i = 0; i < Nxh; ++i) {
        cplx v = a[i + (size_t)Nx * (j + (size_t)Ny * k)];
        s->spec[i + (size_t)Nxh * (j + (size_t)Ny * k)] = {v.real(), v.imag()};
      }
}
static void emu_backward(PoissonSolver* s) {
  int Nx = s->Nx, Ny = s->Ny, Nz = s->Nz, Nxh = s->Nxh;
  std::vector<cplx> a((size_t)Nx * Ny * Nz);
  // rebuild the full spectrum from Hermitian symmetry: X[-i,-j,-k] = conj(X[i,j,k])
  for (int k = 0; k < Nz; ++k)
    for (int j = 0; j < Ny; ++j)
      for (int i = 0; i < Nx; ++i) {
        cplx v;
        if (i < Nxh) {
          double2_ q = s->spec[i + (size_t)Nxh * (j + (size_t)Ny * k)];
          v = cplx(q.x, q.y);
        } else {
          int ii = Nx - i, jj = (Ny - j) % Ny, kk = (s->kind == 0) ? (Nz - k) % Nz : k;
          double2_ q = s->spec[ii + (size_t)Nxh * (jj + (size_t)Ny * kk)];
          v = cplx(q.x, -q.y);
        }
        a[i + (size_t)Nx * (j + (size_t)Ny * k)] = v;
      }
  emu_dft_axis(a, Nx, Ny, Nz, 0, +1);
  emu_dft_axis(a, Nx, Ny, Nz, 1, +1);
  if (s->kind == 0) emu_dft_axis(a, Nx, Ny, Nz, 2, +1);
  for (size_t i = 0; i < a.size(); ++i) s->rhs[i] = a[i].real();
}
#endif

#ifdef OCN_HOST_EMU
static void emu_zfft(PoissonSolver* s, int sign) {
  int n0 = s->Nxh, n1 = s->Nyl, n2 = s->Nzg;
  std::vector<cplx> a((size_t)n0 * n1 * n2);
  for (size_t i = 0; i < a.size(); ++i) a[i] = cplx(s->tb[i].x, s->tb[i].y);
  emu_dft_axis(a, n0, n1, n2, 2, sign);
  for (size_t i = 0; i < a.size(); ++i) s->tb[i] = {a[i].real(), a[i].imag()};
}
#endif

// ---- slab <-> ky-slab reordering around the all-to-all ------------------------------------------------------
// spec [zl][ky][kx]  <->  t [q][zl][kyl][kx]   with ky = q*Nyl + kyl
__global__ void k_pack_slab(int Nxh, int Ny, int Nzl, int Nyl, const double2_* __restrict__ spec,
                            double2_* __restrict__ t, int unpack) {
  const int kx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ky = blockIdx.y * blockDim.y + threadIdx.y;
  const int zl = blockIdx.z;
  if (kx >= Nxh || ky >= Ny || zl >= Nzl) return;
  const int q = ky / Nyl, kyl = ky - q * Nyl;
  const size_t is = kx + (size_t)Nxh * (ky + (size_t)Ny * zl);
  const size_t it = kx + (size_t)Nxh * (kyl + (size_t)Nyl * (zl + (size_t)Nzl * q));
  if (unpack) const_cast<double2_*>(spec)[is] = t[it];
  else t[it] = spec[is];
}

// ky-slab (Nxh, Nyl, Nzg): phi^ = -b^ / (lx + ly + lz) * norm, zero mode on the rank that owns ky = 0
__global__ void k_scale_spectrum_slab(int Nxh, int Nyl, int Nzg, int ky0, const double* __restrict__ lx,
                                      const double* __restrict__ ly, const double* __restrict__ lz, double norm,
                                      double2_* __restrict__ a) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;
  if (i >= Nxh || j >= Nyl || k >= Nzg) return;
  const size_t c = i + (size_t)Nxh * (j + (size_t)Nyl * k);
  double lam = lx[i] + ly[ky0 + j] + lz[k];
  double2_ v = a[c];
  if (i == 0 && ky0 + j == 0 && k == 0) {
    v.x = 0;
    v.y = 0;
  } else {
    double f = -norm / lam;
    v.x *= f;
    v.y *= f;
  }
  a[c] = v;
}

// ---- spectral kernels ------------------------------------------------------------------------------------------
// phi^ = -b^ / (lx + ly + lz) * norm ; zero mode := 0   (fft_based_poisson_solver.jl:106-111)
__global__ void k_scale_spectrum(int Nxh, int Ny, int Nz, const double* __restrict__ lx, const double* __restrict__ ly,
                                 const double* __restrict__ lz, double norm, double2_* __restrict__ a) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;
  if (i >= Nxh || j >= Ny || k >= Nz) return;
  const size_t c = i + (size_t)Nxh * (j + (size_t)Ny * k);
  double lam = lx[i] + ly[j] + (lz ? lz[k] : 0.0);
  double2_ v = a[c];
  if (i == 0 && j == 0 && (k == 0 || !lz)) {
    // zero mode: with no z transform (Flat z) every k-plane has its own undetermined constant
    v.x = 0;
    v.y = 0;
  } else {
    double f = -norm / lam;
    v.x *= f;
    v.y *= f;
  }
  a[c] = v;
}

// Thomas algorithm down z for each (i,j) of the half spectrum (batched_tridiagonal_solver.jl:91-122) with
// the diagonal of fourier_tridiagonal_poisson_solver.jl:16-28 computed on the fly; in place on `a`.
// The elimination factors of the Thomas algorithm (batched_tridiagonal_solver.jl:91-122 with the diagonal of
// fourier_tridiagonal_poisson_solver.jl:16-28) depend on the grid and the eigenvalues only -- not on the right-hand side.
// They are computed ONCE per solver: t_k = c_{k-1} / beta_{k-1}, 1 / beta_k, and the level at which the reference's
// early `break` (:113-114) stops a singular column.  A solve is then two sweeps whose dependent chain per level is one
// fused multiply-add and one multiply instead of two divisions -- the sweeps are latency bound (only Nxh*Ny threads).
__global__ void k_tridiag_setup(GridDev g, int Nxh, int Ny, int Nz, const double* __restrict__ lx, const double* __restrict__ ly,
                                double* __restrict__ t, double* __restrict__ rb, int* __restrict__ kbr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nxh || j >= Ny) return;
  const size_t ncol = (size_t)Nxh * Ny, col = i + (size_t)Nxh * j;
  const double lam = lx[i] + ly[j];
  auto diag = [&](int k) -> double {  // k: 0-based centre index
    double up = (k < Nz - 1) ? g_rdzf(g, k + 1) : 0.0;
    double lo = (k > 0) ? g_rdzf(g, k) : 0.0;
    return -(up + lo) - g_dzc(g, k) * lam;
  };
  double beta = diag(0);
  t[col] = 0.0;
  int kbreak = Nz;
  // A singular first pivot only happens for the horizontal-mean mode of a one-level column (lam = 0, Nz = 1: both faces are
  // walls).  The FFT-based solver the reference uses on such a (regular) grid returns its gauge value 0 there
  // (fft_based_poisson_solver.jl:113-114); so does this one: the level holds the stand-in 0 like every level past a break.
  if (!(fabs(beta) > 10.0 * 2.220446049250313e-16)) kbreak = 0;
  else rb[col] = 1.0 / beta;
  for (int k = 1; k < Nz && kbreak > 0; ++k) {
    double off = g_rdzf(g, k);  // a^{k-1} = c^{k-1} = 1/dzf(k) (1-based face k+1 -> 0-based face k)
    double tk = off / beta;
    t[col + ncol * k] = tk;
    beta = diag(k) - off * tk;
    if (!(fabs(beta) > 10.0 * 2.220446049250313e-16)) {  // reference `break`
      kbreak = k;
      break;
    }
    rb[col + ncol * k] = 1.0 / beta;
  }
  for (int k = kbreak; k < Nz; ++k) rb[col + ncol * k] = 0.0;       // levels past the break stay at the stand-in value 0
  for (int k = kbreak + 1; k < Nz; ++k) t[col + ncol * k] = 0.0;
  kbr[col] = kbreak;
}

// Thomas sweeps down z for each (i,j) column of the spectrum with the precomputed factors; in place on `a`.
__global__ void k_tridiag(GridDev g, int Nxh, int Ny, int Nz, double norm, double2_* __restrict__ a,
                          const double* __restrict__ t, const double* __restrict__ rb, const int* __restrict__ kbr,
                          int owns_mean, double* __restrict__ mean_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nxh || j >= Ny) return;
  const size_t ncol = (size_t)Nxh * Ny, col = i + (size_t)Nxh * j;
  // Both sweeps are serial in k and there are only Nxh Ny / 64 waves: the kernel is bound by memory latency.  Loads run in
  // chunks of PF levels, and the NEXT chunk is requested before the current one is consumed (the array is updated in place,
  // so the order is written out by hand: the compiler will not move a load of `a` above a store to it).
  constexpr int PF = 8;
  const int kbreak = kbr[col];
  double2_ f = a[col];
  double r0 = rb[col];
  double2_ prev = {f.x * norm * r0, f.y * norm * r0};
  a[col] = prev;
  {
    double2_ fA[PF], fB[PF];
    double rA[PF], rB[PF];
    auto fetch = [&](double2_* fb, double* rbb, int k0) {
#pragma unroll
      for (int q = 0; q < PF; ++q)
        if (k0 + q < Nz) {
          fb[q] = a[col + ncol * (k0 + q)];
          rbb[q] = rb[col + ncol * (k0 + q)];
        }
    };
    auto sweep = [&](const double2_* fb, const double* rbb, int k0) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int k = k0 + q;
        if (k >= Nz) break;
        const double off = g_rdzf(g, k);
        // rb is 0 from the break level on: those levels hold the deterministic stand-in 0, like the stale storage they replace
        double2_ cur = {(fb[q].x * norm - off * prev.x) * rbb[q], (fb[q].y * norm - off * prev.y) * rbb[q]};
        a[col + ncol * k] = cur;
        prev = cur;
      }
    };
    fetch(fA, rA, 1);
    for (int k0 = 1; k0 < Nz; k0 += 2 * PF) {
      fetch(fB, rB, k0 + PF);
      sweep(fA, rA, k0);
      fetch(fA, rA, k0 + 2 * PF);
      sweep(fB, rB, k0 + PF);
    }
  }
  (void)kbreak;
  double2_ nxt = prev;                 // level Nz - 1, as stored
  double sx = nxt.x, sy_ = nxt.y;
  {
    double tA[PF], tB[PF];
    double2_ aA[PF], aB[PF];
    auto fetch = [&](double* tb, double2_* ab, int k0) {
#pragma unroll
      for (int q = 0; q < PF; ++q)
        if (k0 - q >= 0) {
          tb[q] = t[col + ncol * (k0 - q + 1)];
          ab[q] = a[col + ncol * (k0 - q)];
        }
    };
    auto sweep = [&](const double* tb, const double2_* ab, int k0) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int k = k0 - q;
        if (k < 0) break;
        double2_ cur = ab[q];
        cur.x -= tb[q] * nxt.x;
        cur.y -= tb[q] * nxt.y;
        a[col + ncol * k] = cur;
        nxt = cur;
        sx += cur.x;
        sy_ += cur.y;
      }
    };
    // the forward sweep's last stores and these loads touch the same array: levels Nz-2 ... are read back after they were written
    fetch(tA, aA, Nz - 2);
    for (int k0 = Nz - 2; k0 >= 0; k0 -= 2 * PF) {
      fetch(tB, aB, k0 - PF);
      sweep(tA, aA, k0);
      fetch(tA, aA, k0 - 2 * PF);
      sweep(tB, aB, k0 - PF);
    }
  }
  // phi .-= mean(phi) (fourier_tridiagonal_poisson_solver.jl:95) acts on the horizontal-mean mode only: its column mean is
  // handed to k_tridiag_mean (one thread walking Nz levels here kept the whole GPU waiting for it)
  if (owns_mean && i == 0 && j == 0) {
    mean_out[0] = sx / Nz;
    mean_out[1] = sy_ / Nz;
  }
}

__global__ void k_tridiag_mean(double2_* __restrict__ a, size_t ncol, int Nz, const double* __restrict__ mean) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Nz) return;
  double2_ cur = a[ncol * k];
  cur.x -= mean[0];
  cur.y -= mean[1];
  a[ncol * k] = cur;
}

// launch: factors are built on first use (the model's GridDev -- halo, spacings -- is final by then)
static void tridiag_run(ocn_model* m, PoissonSolver* s, int Nc0, int Nc1, const double* l0, const double* l1, double norm,
                        double2_* data, int owns_mean) {
  hipStream_t st = m->ctx->stream;
  const size_t nc = (size_t)Nc0 * Nc1 * s->Nz;
  // one wave per workgroup: only Nc0*Nc1/64 waves exist, so they are spread over as many CUs as possible
  static const int by = getenv("OCNHIP_TRI_BY") ? atoi(getenv("OCNHIP_TRI_BY")) : 1;
  const dim3 b(64, by, 1), gr((Nc0 + 63) / 64, (Nc1 + by - 1) / by, 1);
  if (!s->tri_ready) {
    ocn_launch(k_tridiag_setup, gr, b, st, m->gd, Nc0, Nc1, s->Nz, l0, l1, s->tscr, s->tscr + nc, s->tri_kbr);
    s->tri_ready = true;
  }
  ocn_launch(k_tridiag, gr, b, st, m->gd, Nc0, Nc1, s->Nz, norm, data, (const double*)s->tscr, (const double*)(s->tscr + nc),
             (const int*)s->tri_kbr, owns_mean, m->d_red + 2);
  if (owns_mean) ocn_launch(k_tridiag_mean, dim3((s->Nz + 63) / 64, 1, 1), dim3(64, 1, 1), st, data, (size_t)Nc0 * Nc1, s->Nz, (const double*)(m->d_red + 2));
}

// ---- a Bounded or Flat x / y direction (kind 4) ----------------------------------------------------------------
// fft_based_poisson_solver.jl:93-120 / fourier_tridiagonal_poisson_solver.jl:67-101 for the topologies beyond
// (Periodic, Periodic, *).  The reference builds cosine transforms from permuted FFTs + twiddles
// (discrete_transforms.jl:125-175); here a cosine transform is what it is -- a small dense real matrix applied
// along one direction of a large array -- and runs as ONE FP64 GEMM on the matrix cores (rocBLAS): N x N
// times N x (everything else), 2 N flops per element, ~0.15 ms per direction at 256^3, no permutation passes,
// any N.  Periodic directions keep their FFTs: x as a batched real-to-complex transform (contiguous), z as a
// strided complex one; a cosine transform in y then acts on the interleaved half spectrum as on a real array
// with twice as many rows.  x Bounded / Flat with y Periodic is solved on the x <-> y transposed array.
__global__ void k_transpose_xy(int n0, int n1, int n2, const double* __restrict__ in, double* __restrict__ out) {
  // out(j, i, k) = in(i, j, k); in is (n0, n1, n2)
  OCN_SHARED double tile[32][33];
  const int k = blockIdx.z;
  int i = blockIdx.x * 32 + threadIdx.x, j = blockIdx.y * 32 + threadIdx.y;
  for (int r = 0; r < 32; r += 8)
    if (i < n0 && j + r < n1) tile[threadIdx.y + r][threadIdx.x] = in[i + (size_t)n0 * (j + r + (size_t)n1 * k)];
  __syncthreads();
  i = blockIdx.x * 32 + threadIdx.y;
  j = blockIdx.y * 32 + threadIdx.x;
  for (int r = 0; r < 32; r += 8)
    if (i + r < n0 && j < n1) out[j + (size_t)n1 * (i + r + (size_t)n0 * k)] = tile[threadIdx.x][threadIdx.y + r];
}
__global__ void k_real_to_complex(size_t n, const double* __restrict__ r, double2_* __restrict__ c) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c[i] = {r[i], 0.0};
}
__global__ void k_complex_to_real(size_t n, const double2_* __restrict__ c, double* __restrict__ r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = c[i].x;
}

#ifdef OCN_HOST_EMU
static void emu_x_r2c(const double* in, double2_* out, int Nx, int Nxh, size_t lines) {
  for (size_t L = 0; L < lines; ++L)
    for (int k = 0; k < Nxh; ++k) {
      cplx acc = 0;
      for (int n = 0; n < Nx; ++n) {
        double ang = -2.0 * M_PI * ((long)n * k % Nx) / Nx;
        acc += in[n + Nx * L] * cplx(cos(ang), sin(ang));
      }
      out[k + Nxh * L] = {acc.real(), acc.imag()};
    }
}
static void emu_x_c2r(const double2_* in, double* out, int Nx, int Nxh, size_t lines) {
  std::vector<cplx> full(Nx);
  for (size_t L = 0; L < lines; ++L) {
    for (int i = 0; i < Nx; ++i) {
      double2_ q = in[(i < Nxh ? i : Nx - i) + Nxh * L];
      full[i] = cplx(q.x, i < Nxh ? q.y : -q.y);
    }
    for (int n = 0; n < Nx; ++n) {
      cplx acc = 0;
      for (int i = 0; i < Nx; ++i) {
        double ang = 2.0 * M_PI * ((long)i * n % Nx) / Nx;
        acc += full[i] * cplx(cos(ang), sin(ang));
      }
      out[n + Nx * L] = acc.real();
    }
  }
}
#endif

// C(m x n) = A(m x k) * op(B), column-major, batched over `batch` with strides (0 = shared operand)
static int walls_gemm(ocn_model* m, bool transB, int M, int N, int K, const double* A, int lda, long sA, const double* B,
                      int ldb, double* C, int ldc, long sC, int batch) {
#ifndef OCN_HOST_EMU
  if (g_ocn_capturing) {   // the BLAS call is kept out of stream capture: the step that wanted a graph is run again without one
    g_ocn_capture_poison = 1;
    return OCN_OK;
  }
  const double one = 1.0, zero = 0.0;
  rocblas_status st = rocblas_dgemm_strided_batched(
      (rocblas_handle)m->solver->blas, rocblas_operation_none, transB ? rocblas_operation_transpose : rocblas_operation_none,
      M, N, K, &one, A, lda, sA, B, ldb, 0, &zero, C, ldc, sC, batch);
  if (st != rocblas_status_success) {
    ocn_set_error(m->ctx, "rocblas_dgemm_strided_batched failed (%d)", (int)st);
    return OCN_EHIP;
  }
#else
  for (int b = 0; b < batch; ++b)
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < M; ++i) {
        double acc = 0;
        for (int q = 0; q < K; ++q)
          acc += A[i + (size_t)lda * q + (size_t)sA * b] * (transB ? B[j + (size_t)ldb * q] : B[q + (size_t)ldb * j]);
        C[i + (size_t)ldc * j + (size_t)sC * b] = acc;
      }
#endif
  return OCN_OK;
}

static int walls_create(ocn_model* m, PoissonSolver* s) {
  ocn_grid* g = m->g;
  // x Bounded / Flat with y Periodic: swap the horizontal directions so that the Periodic one is contiguous
  s->tr = g->topo[0] != OCN_PERIODIC && g->topo[1] == OCN_PERIODIC;
  const int a = s->tr ? 1 : 0, b = s->tr ? 0 : 1;
  s->gNx = g->N[a];
  s->gNy = g->N[b];
  s->gtopo[0] = g->topo[a];
  s->gtopo[1] = g->topo[b];
  const bool xper = s->gtopo[0] == OCN_PERIODIC && s->gNx > 1;
  s->gR = xper ? s->gNx / 2 + 1 : s->gNx;
  s->Nxh = s->gR;
  const size_t nr = (size_t)s->gNx * s->gNy * s->Nz, nc = (size_t)s->gR * s->gNy * s->Nz;
  if (hipMalloc((void**)&s->rhs, nr * sizeof(double)) != hipSuccess || hipMalloc((void**)&s->ra, nr * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&s->rb, nr * sizeof(double)) != hipSuccess || hipMalloc((void**)&s->ga, nc * sizeof(double2_)) != hipSuccess ||
      hipMalloc((void**)&s->gb, nc * sizeof(double2_)) != hipSuccess)
    return OCN_ENOMEM;
  if (g->topo[2] == OCN_BOUNDED && (hipMalloc((void**)&s->tscr, 2 * nc * sizeof(double)) != hipSuccess ||
                                    hipMalloc((void**)&s->tri_kbr, (size_t)s->gR * s->gNy * sizeof(int)) != hipSuccess))
    return OCN_ENOMEM;
  for (int d = 0; d < 2; ++d) {
    const int N = d == 0 ? s->gNx : s->gNy;
    if (s->gtopo[d] != OCN_BOUNDED || N == 1) continue;
    std::vector<double> F, I;
    cosine_matrices(N, F, I);
    s->gm[d][0] = upload(F);
    s->gm[d][1] = upload(I);
    if (!s->gm[d][0] || !s->gm[d][1]) return OCN_ENOMEM;
  }
  std::vector<double> lx = eigenvalues(s->gtopo[0], s->gNx, g->L[a]);
  lx.resize(s->gR);
  s->lx = upload(lx);
  s->ly = upload(eigenvalues(s->gtopo[1], s->gNy, g->L[b]));
  if (g->topo[2] == OCN_PERIODIC) s->lz = upload(eigenvalues(OCN_PERIODIC, s->Nz, g->L[2]));
#ifndef OCN_HOST_EMU
  rocblas_handle h = nullptr;
  if (rocblas_create_handle(&h) != rocblas_status_success) {
    ocn_set_error(m->ctx, "rocblas_create_handle failed");
    return OCN_EHIP;
  }
  s->blas = h;
  rocblas_set_stream(h, m->ctx->stream);
  const int lines = s->gNy * s->Nz;
  if (xper) {
    int nx[1] = {s->gNx};
    if (hipfftPlanMany(&s->wxf, 1, nx, nullptr, 1, s->gNx, nullptr, 1, s->gR, HIPFFT_D2Z, lines) != HIPFFT_SUCCESS ||
        hipfftPlanMany(&s->wxi, 1, nx, nullptr, 1, s->gR, nullptr, 1, s->gNx, HIPFFT_Z2D, lines) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfft x-plan creation failed");
      return OCN_EHIP;
    }
    hipfftSetStream(s->wxf, m->ctx->stream);
    hipfftSetStream(s->wxi, m->ctx->stream);
  }
  if (g->topo[2] == OCN_PERIODIC && s->Nz > 1) {
    int nz[1] = {s->Nz};
    int st = s->gR * s->gNy;
    if (hipfftPlanMany(&s->wz, 1, nz, nz, st, 1, nz, st, 1, HIPFFT_Z2Z, st) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfft z-plan creation failed");
      return OCN_EHIP;
    }
    hipfftSetStream(s->wz, m->ctx->stream);
  }
#endif
  return OCN_OK;
}

static int run_walls(ocn_model* m) {
  PoissonSolver* s = m->solver;
  hipStream_t st = m->ctx->stream;
  const int Nx = s->gNx, Ny = s->gNy, Nz = s->Nz, R = s->gR;
  const size_t nr = (size_t)Nx * Ny * Nz, nc = (size_t)R * Ny * Nz;
  const bool xper = s->gtopo[0] == OCN_PERIODIC && Nx > 1, zper = m->g->topo[2] == OCN_PERIODIC && Nz > 1;
  const dim3 b1(256, 1, 1);
  auto g1 = [&](size_t n) { return dim3((unsigned)((n + 255) / 256), 1, 1); };
  const dim3 tb(32, 8, 1);
  int rc = OCN_OK;
  double* real = s->rhs;          // where the real array currently lives
  double2_ *cur = s->ga, *oth = s->gb;
  {
    ProfScope ps(m->ctx, "fft_forward");
    if (s->tr) {   // (Ny_orig = gNx ...): rhs is (gNy, gNx, Nz) in memory -> (gNx, gNy, Nz)
      ocn_launch_sync(k_transpose_xy, dim3((Ny + 31) / 32, (Nx + 31) / 32, Nz), tb, st, Ny, Nx, Nz, (const double*)real, s->ra);
      real = s->ra;
    }
    if (s->gm[0][0]) {   // cosine transform along x: F (Nx x Nx) * A (Nx x Ny Nz)
      double* out = real == s->ra ? s->rb : s->ra;
      if ((rc = walls_gemm(m, false, Nx, Ny * Nz, Nx, s->gm[0][0], Nx, 0, real, Nx, out, Nx, 0, 1))) return rc;
      real = out;
    }
    if (xper) {
#ifndef OCN_HOST_EMU
      if (dry_hipfftExecD2Z(s->wxf, real, (hipfftDoubleComplex*)cur) != HIPFFT_SUCCESS) {
        ocn_set_error(m->ctx, "hipfftExecD2Z failed");
        return OCN_EHIP;
      }
#else
      emu_x_r2c(real, cur, Nx, R, (size_t)Ny * Nz);
#endif
    } else {
      ocn_launch(k_real_to_complex, g1(nr), b1, st, nr, (const double*)real, cur);
    }
    if (s->gm[1][0]) {   // cosine transform along y on (2R x Ny) planes: A * F^T
      if ((rc = walls_gemm(m, true, 2 * R, Ny, Ny, (const double*)cur, 2 * R, (long)2 * R * Ny, s->gm[1][0], Ny, (double*)oth,
                           2 * R, (long)2 * R * Ny, Nz)))
        return rc;
      std::swap(cur, oth);
    }
  }
  {
    ProfScope ps(m->ctx, "spectral_solve");
    const dim3 b(64, 4, 1);
    const double norm = 1.0 / ((xper ? (double)Nx : 1.0) * (zper ? (double)Nz : 1.0));
    if (m->g->topo[2] == OCN_BOUNDED) {
      tridiag_run(m, s, R, Ny, s->lx, s->ly, norm, cur, 1);
    } else {
      if (zper) {
#ifndef OCN_HOST_EMU
        if (dry_hipfftExecZ2Z(s->wz, (hipfftDoubleComplex*)cur, (hipfftDoubleComplex*)cur, HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
          ocn_set_error(m->ctx, "hipfftExecZ2Z failed");
          return OCN_EHIP;
        }
#else
        {
          std::vector<cplx> a(nc);
          for (size_t i = 0; i < nc; ++i) a[i] = cplx(cur[i].x, cur[i].y);
          emu_dft_axis(a, R, Ny, Nz, 2, -1);
          for (size_t i = 0; i < nc; ++i) cur[i] = {a[i].real(), a[i].imag()};
        }
#endif
      }
      ocn_launch(k_scale_spectrum, dim3((R + 63) / 64, (Ny + 3) / 4, Nz), b, st, R, Ny, Nz, (const double*)s->lx,
                 (const double*)s->ly, (const double*)(zper ? s->lz : nullptr), norm, cur);
      if (zper) {
#ifndef OCN_HOST_EMU
        if (dry_hipfftExecZ2Z(s->wz, (hipfftDoubleComplex*)cur, (hipfftDoubleComplex*)cur, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) {
          ocn_set_error(m->ctx, "hipfftExecZ2Z failed");
          return OCN_EHIP;
        }
#else
        {
          std::vector<cplx> a(nc);
          for (size_t i = 0; i < nc; ++i) a[i] = cplx(cur[i].x, cur[i].y);
          emu_dft_axis(a, R, Ny, Nz, 2, +1);
          for (size_t i = 0; i < nc; ++i) cur[i] = {a[i].real(), a[i].imag()};
        }
#endif
      }
    }
  }
  {
    ProfScope ps(m->ctx, "fft_backward");
    if (s->gm[1][1]) {
      if ((rc = walls_gemm(m, true, 2 * R, Ny, Ny, (const double*)cur, 2 * R, (long)2 * R * Ny, s->gm[1][1], Ny, (double*)oth,
                           2 * R, (long)2 * R * Ny, Nz)))
        return rc;
      std::swap(cur, oth);
    }
    // back to a real array; the last stage writes s->rhs directly when no further pass follows
    const bool more = s->gm[0][1] != nullptr || s->tr;
    real = more ? s->ra : s->rhs;
    if (xper) {
#ifndef OCN_HOST_EMU
      if (dry_hipfftExecZ2D(s->wxi, (hipfftDoubleComplex*)cur, real) != HIPFFT_SUCCESS) {
        ocn_set_error(m->ctx, "hipfftExecZ2D failed");
        return OCN_EHIP;
      }
#else
      emu_x_c2r(cur, real, Nx, R, (size_t)Ny * Nz);
#endif
    } else {
      ocn_launch(k_complex_to_real, g1(nr), b1, st, nr, (const double2_*)cur, real);
    }
    if (s->gm[0][1]) {
      double* out = s->tr ? s->rb : s->rhs;
      if ((rc = walls_gemm(m, false, Nx, Ny * Nz, Nx, s->gm[0][1], Nx, 0, real, Nx, out, Nx, 0, 1))) return rc;
      real = out;
    }
    if (s->tr) ocn_launch_sync(k_transpose_xy, dim3((Nx + 31) / 32, (Ny + 31) / 32, Nz), tb, st, Nx, Ny, Nz, (const double*)real, s->rhs);
  }
  return OCN_OK;
}

// ---- y-slabs with a Bounded z (kind 5) -----------------------------------------------------------------------------
// The reference has no distributed Fourier-tridiagonal solver (NonhydrostaticModels.jl:18-21).  Here every rank
// owns Ny/R rows of whole columns: the x transform and the tridiagonal solve down z are local, and the y
// transform needs the rows of all ranks -- one all-to-all that hands each rank a band of kx for ALL y (the
// x <-> y transposition is folded into the pack / unpack kernels, so the y transform is contiguous), and one back.
__global__ void k_yslab_pack(int Nxh, int Nyl, int Nz, int w, int R, double2_* __restrict__ spec, double2_* __restrict__ blocks,
                             int unpack) {
  // blocks[q][jl + Nyl (kxl + w z)]  <->  spec[kx + Nxh (jl + Nyl z)],  kx = q w + kxl (zero padded past Nxh)
  const int kx = blockIdx.x * blockDim.x + threadIdx.x;
  const int jl = blockIdx.y * blockDim.y + threadIdx.y;
  const int z = blockIdx.z;
  if (kx >= w * R || jl >= Nyl || z >= Nz) return;
  const int q = kx / w, kxl = kx - q * w;
  const size_t ib = (size_t)q * Nyl * w * Nz + jl + (size_t)Nyl * (kxl + (size_t)w * z);
  const size_t is = kx + (size_t)Nxh * (jl + (size_t)Nyl * z);
  if (unpack) {
    if (kx < Nxh) spec[is] = blocks[ib];
  } else {
    blocks[ib] = kx < Nxh ? spec[is] : double2_{0.0, 0.0};
  }
}
__global__ void k_yslab_gather(int Nyl, int Nyg, int Nz, int w, int R, double2_* __restrict__ blocks, double2_* __restrict__ T,
                               int scatter) {
  // T[y + Nyg (kxl + w z)]  <->  blocks[p][jl + Nyl (kxl + w z)],  y = p Nyl + jl
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  const int kxl = blockIdx.y;
  const int z = blockIdx.z;
  if (y >= Nyg || kxl >= w || z >= Nz) return;
  const int p = y / Nyl, jl = y - p * Nyl;
  const size_t ib = (size_t)p * Nyl * w * Nz + jl + (size_t)Nyl * (kxl + (size_t)w * z);
  const size_t it = y + (size_t)Nyg * (kxl + (size_t)w * z);
  if (scatter) blocks[ib] = T[it];
  else T[it] = blocks[ib];
}

static int yslab_create(ocn_model* m, PoissonSolver* s) {
  ocn_grid* g = m->g;
  const int R = m->ctx->nranks;
  s->R = R;
  s->rank = m->ctx->rank;
  s->Nyg = g->Nyg;
  s->yw = (s->Nxh + R - 1) / R;
  const size_t nr = (size_t)s->Nx * s->Ny * s->Nz, nc = (size_t)s->Nxh * s->Ny * s->Nz;
  const size_t nt = (size_t)s->Nyg * s->yw * s->Nz;
  if (hipMalloc((void**)&s->rhs, nr * sizeof(double)) != hipSuccess || hipMalloc((void**)&s->spec, nc * sizeof(double2_)) != hipSuccess ||
      hipMalloc((void**)&s->ysend, nt * sizeof(double2_)) != hipSuccess || hipMalloc((void**)&s->yrecv, nt * sizeof(double2_)) != hipSuccess ||
      hipMalloc((void**)&s->yT, nt * sizeof(double2_)) != hipSuccess || hipMalloc((void**)&s->tscr, 2 * nt * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&s->tri_kbr, (size_t)s->Nyg * s->yw * sizeof(int)) != hipSuccess)
    return OCN_ENOMEM;
  std::vector<double> lxg = eigenvalues_periodic(s->Nx, g->L[0]), lxl(s->yw);
  for (int i = 0; i < s->yw; ++i) {
    int kx = s->rank * s->yw + i;
    lxl[i] = lxg[kx < s->Nxh ? kx : s->Nxh - 1];     // padding columns carry zeros; any non-singular value will do
  }
  s->lx = upload(lxl);
  s->ly = upload(eigenvalues_periodic(s->Nyg, g->L[1]));
#ifndef OCN_HOST_EMU
  int nx[1] = {s->Nx}, ny[1] = {s->Nyg};
  if (hipfftPlanMany(&s->wxf, 1, nx, nullptr, 1, s->Nx, nullptr, 1, s->Nxh, HIPFFT_D2Z, s->Ny * s->Nz) != HIPFFT_SUCCESS ||
      hipfftPlanMany(&s->wxi, 1, nx, nullptr, 1, s->Nxh, nullptr, 1, s->Nx, HIPFFT_Z2D, s->Ny * s->Nz) != HIPFFT_SUCCESS ||
      hipfftPlanMany(&s->yfft, 1, ny, nullptr, 1, s->Nyg, nullptr, 1, s->Nyg, HIPFFT_Z2Z, s->yw * s->Nz) != HIPFFT_SUCCESS) {
    ocn_set_error(m->ctx, "hipfft plan creation failed (y-slab solver)");
    return OCN_EHIP;
  }
  hipfftSetStream(s->wxf, m->ctx->stream);
  hipfftSetStream(s->wxi, m->ctx->stream);
  hipfftSetStream(s->yfft, m->ctx->stream);
#endif
  return OCN_OK;
}

static int yslab_yfft(ocn_model* m, int sign) {
  PoissonSolver* s = m->solver;
#ifndef OCN_HOST_EMU
  if (dry_hipfftExecZ2Z(s->yfft, (hipfftDoubleComplex*)s->yT, (hipfftDoubleComplex*)s->yT, sign < 0 ? HIPFFT_FORWARD : HIPFFT_BACKWARD) !=
      HIPFFT_SUCCESS) {
    ocn_set_error(m->ctx, "hipfftExecZ2Z failed");
    return OCN_EHIP;
  }
#else
  const size_t nt = (size_t)s->Nyg * s->yw * s->Nz;
  std::vector<cplx> a(nt);
  for (size_t i = 0; i < nt; ++i) a[i] = cplx(s->yT[i].x, s->yT[i].y);
  emu_dft_axis(a, s->Nyg, s->yw, s->Nz, 0, sign);
  for (size_t i = 0; i < nt; ++i) s->yT[i] = {a[i].real(), a[i].imag()};
#endif
  return OCN_OK;
}

static int run_yslab(ocn_model* m) {
  PoissonSolver* s = m->solver;
  hipStream_t st = m->ctx->stream;
  const int Nx = s->Nx, Nxh = s->Nxh, Nyl = s->Ny, Nyg = s->Nyg, Nz = s->Nz, w = s->yw, R = s->R;
  const size_t blk = (size_t)Nyl * w * Nz * sizeof(double2_);
  const dim3 b(64, 4, 1), gp((w * R + 63) / 64, (Nyl + 3) / 4, Nz), bg(64, 1, 1), gg((Nyg + 63) / 64, w, Nz);
  int rc = OCN_OK;
  {
    ProfScope ps(m->ctx, "fft_forward");
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecD2Z(s->wxf, s->rhs, (hipfftDoubleComplex*)s->spec) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecD2Z failed");
      return OCN_EHIP;
    }
#else
    emu_x_r2c(s->rhs, s->spec, Nx, Nxh, (size_t)Nyl * Nz);
#endif
    ocn_launch(k_yslab_pack, gp, b, st, Nxh, Nyl, Nz, w, R, s->spec, s->ysend, 0);
  }
  if ((rc = comm_alltoall(m->ctx, s->ysend, s->yrecv, blk))) return rc;
  {
    ProfScope ps(m->ctx, "fft_forward");
    ocn_launch(k_yslab_gather, gg, bg, st, Nyl, Nyg, Nz, w, R, s->yrecv, s->yT, 0);
    if ((rc = yslab_yfft(m, -1))) return rc;
  }
  {
    ProfScope ps(m->ctx, "spectral_solve");
    // columns are (ky, kx_local): ky runs fastest, so the roles of lx / ly in the kernel are swapped
    tridiag_run(m, s, Nyg, w, s->ly, s->lx, 1.0 / ((double)Nx * Nyg), s->yT, s->rank == 0 ? 1 : 0);
  }
  {
    ProfScope ps(m->ctx, "fft_backward");
    if ((rc = yslab_yfft(m, +1))) return rc;
    ocn_launch(k_yslab_gather, gg, bg, st, Nyl, Nyg, Nz, w, R, s->ysend, s->yT, 1);
  }
  if ((rc = comm_alltoall(m->ctx, s->ysend, s->yrecv, blk))) return rc;
  {
    ProfScope ps(m->ctx, "fft_backward");
    ocn_launch(k_yslab_pack, gp, b, st, Nxh, Nyl, Nz, w, R, s->spec, s->yrecv, 1);
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecZ2D(s->wxi, (hipfftDoubleComplex*)s->spec, s->rhs) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecZ2D failed");
      return OCN_EHIP;
    }
#else
    emu_x_c2r(s->spec, s->rhs, Nx, Nxh, (size_t)Nyl * Nz);
#endif
  }
  return OCN_OK;
}

static int run_solver(ocn_model* m) {
  PoissonSolver* s = m->solver;
  hipStream_t st = m->ctx->stream;
  if (s->kind == 4) return run_walls(m);
  if (s->kind == 5) return run_yslab(m);
  {
    ProfScope ps(m->ctx, "fft_forward");
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecD2Z(s->fwd, s->rhs, (hipfftDoubleComplex*)s->spec) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecD2Z failed");
      return OCN_EHIP;
    }
#else
    emu_forward(s);
#endif
  }
  if (s->kind == 2 && s->zsl) {
    int rc = zslab_run(m->ctx, s->zsl, s->spec, s->dz2, 1.0 / ((double)s->Nx * s->Ny));
    if (rc) return rc;
  } else if (s->kind == 2) {
    dim3 b(64, 4, 1);
    const size_t blk = (size_t)s->Nxh * s->Nyl * s->Nz * sizeof(double2_);
    {
      ProfScope ps(m->ctx, "spectral_solve");
      dim3 gr((s->Nxh + 63) / 64, (s->Ny + 3) / 4, s->Nz);
      ocn_launch(k_pack_slab, gr, b, st, s->Nxh, s->Ny, s->Nz, s->Nyl, (const double2_*)s->spec, s->ta, 0);
    }
    int rc = comm_alltoall(m->ctx, s->ta, s->tb, blk);
    if (rc) return rc;
    if (s->zs) {
      ProfScope ps(m->ctx, "spectral_solve");
      zsolve_run(m->ctx, s->zs, s->tb, s->Nzg, s->lz, 1.0 / ((double)s->Nx * s->Ny * s->Nzg), s->rank == 0 ? 0 : -1);
    } else
    {
      ProfScope ps(m->ctx, "spectral_solve");
#ifndef OCN_HOST_EMU
      if (dry_hipfftExecZ2Z(s->zplan, (hipfftDoubleComplex*)s->tb, (hipfftDoubleComplex*)s->tb, HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
        ocn_set_error(m->ctx, "hipfftExecZ2Z failed");
        return OCN_EHIP;
      }
#else
      emu_zfft(s, -1);
#endif
      dim3 gr((s->Nxh + 63) / 64, (s->Nyl + 3) / 4, s->Nzg);
      double norm = 1.0 / ((double)s->Nx * s->Ny * s->Nzg);
      ocn_launch(k_scale_spectrum_slab, gr, b, st, s->Nxh, s->Nyl, s->Nzg, s->rank * s->Nyl, (const double*)s->lx,
                 (const double*)s->ly, (const double*)s->lz, norm, s->tb);
#ifndef OCN_HOST_EMU
      if (dry_hipfftExecZ2Z(s->zplan, (hipfftDoubleComplex*)s->tb, (hipfftDoubleComplex*)s->tb, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) {
        ocn_set_error(m->ctx, "hipfftExecZ2Z failed");
        return OCN_EHIP;
      }
#else
      emu_zfft(s, +1);
#endif
    }
    rc = comm_alltoall(m->ctx, s->tb, s->ta, blk);
    if (rc) return rc;
    {
      ProfScope ps(m->ctx, "spectral_solve");
      dim3 gr((s->Nxh + 63) / 64, (s->Ny + 3) / 4, s->Nz);
      ocn_launch(k_pack_slab, gr, b, st, s->Nxh, s->Ny, s->Nz, s->Nyl, (const double2_*)s->spec, s->ta, 1);
    }
  } else if (s->kind == 3) {
    ProfScope ps(m->ctx, "spectral_solve");
    zsolve_run(m->ctx, s->zs, s->spec, s->Nz, s->lz, 1.0 / ((double)s->Nx * s->Ny * s->Nz), 0);
  } else {
    ProfScope ps(m->ctx, "spectral_solve");
    dim3 b(64, 4, 1);
    if (m->g->topo[2] == OCN_BOUNDED) {
      dim3 gr((s->Nxh + 63) / 64, (s->Ny + 3) / 4, 1);
      double norm = 1.0 / ((double)s->Nx * s->Ny);
      tridiag_run(m, s, s->Nxh, s->Ny, s->lx, s->ly, norm, s->spec, 1);
    } else {
      dim3 gr((s->Nxh + 63) / 64, (s->Ny + 3) / 4, s->Nz);
      double norm = 1.0 / ((double)s->Nx * s->Ny * (s->kind == 0 ? s->Nz : 1));
      ocn_launch(k_scale_spectrum, gr, b, st, s->Nxh, s->Ny, s->Nz, (const double*)s->lx, (const double*)s->ly,
                 (const double*)s->lz, norm, s->spec);
    }
  }
  {
    ProfScope ps(m->ctx, "fft_backward");
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecZ2D(s->inv, (hipfftDoubleComplex*)s->spec, s->rhs) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecZ2D failed");
      return OCN_EHIP;
    }
#else
    emu_backward(s);
#endif
  }
  return OCN_OK;
}

int poisson_run(ocn_model* m) { return run_solver(m); }

bool poisson_custom_xy(const ocn_model* m) { return m->solver && m->solver->cxy; }
bool poisson_local_wstar(const ocn_model* m) { return m->solver && m->solver->bplane != nullptr; }
bool poisson_local_phi_below(const ocn_model* m) { return m->solver && m->solver->bplane != nullptr && m->solver->local_phi; }
const double* poisson_phi_below(const ocn_model* m) { return m->solver->rhs + (size_t)m->solver->Nx * m->solver->Ny * m->solver->Nz; }

#ifdef OCN_HOST_EMU
// x-inverse of the half spectrum, line by line (emulation of the batched 1-D Z2D plan)
static void emu_xinv(PoissonSolver* s, const double2_* spec = nullptr, double* out = nullptr, size_t nlines = 0) {
  const int Nx = s->Nx, Nxh = s->Nxh;
  const size_t lines = nlines ? nlines : (size_t)s->Ny * s->Nz;
  if (!spec) spec = s->spec;
  if (!out) out = s->rhs;
  std::vector<cplx> full(Nx);
  for (size_t L = 0; L < lines; ++L) {
    for (int i = 0; i < Nx; ++i) {
      double2_ q = spec[(i < Nxh ? i : Nx - i) + Nxh * L];
      full[i] = cplx(q.x, i < Nxh ? q.y : -q.y);
    }
    for (int n = 0; n < Nx; ++n) {
      cplx acc = 0;
      for (int i = 0; i < Nx; ++i) {
        double ang = 2.0 * M_PI * ((long)i * n % Nx) / Nx;
        acc += full[i] * cplx(cos(ang), sin(ang));
      }
      out[n + Nx * L] = acc.real();
    }
  }
}
#endif

// fast path: rhs + x transform fused, custom y transform, z stage, custom inverse y, library inverse x.
// Leaves the solution in the solver's real buffer (like run_solver).
int poisson_run_from_predictor(ocn_model* m, double dt) {
  PoissonSolver* s = m->solver;
  const int xp = s->bplane ? 1 : 0, ip = (s->bplane && s->local_phi) ? 1 : 0;   // planes more in the forward / inverse passes
  {
    ProfScope ps(m->ctx, "fft_forward");
    xfft_rhs_run(m, s->tw, s->spec, dt, xp);
    yfft_run(m->ctx, s->tw, s->spec, s->Nxh, s->Ny, s->Nz + xp, 0);
  }
  if (s->zs) {
    ProfScope ps(m->ctx, "spectral_solve");
    zsolve_run(m->ctx, s->zs, s->spec, s->Nz, s->lz, 1.0 / ((double)s->Nx * s->Ny * s->Nz), 0);
  } else {
    int rc = zslab_run(m->ctx, s->zsl, s->spec, s->dz2, 1.0 / ((double)s->Nx * s->Ny), s->bplane, ip ? s->bplane : nullptr);
    if (rc) return rc;
  }
  {
    ProfScope ps(m->ctx, "fft_backward");
    yfft_run(m->ctx, s->tw, s->spec, s->Nxh, s->Ny, s->Nz + ip, 1);
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecZ2D(s->xinv, (hipfftDoubleComplex*)s->spec, s->rhs) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecZ2D (x inverse) failed");
      return OCN_EHIP;
    }
#else
    emu_xinv(s, s->spec, s->rhs, (size_t)s->Ny * (s->Nz + ip));
#endif
  }
  return OCN_OK;
}

// Bounded z: rhs (times dz) + x transform fused, custom y transform, batched Thomas sweeps, custom inverse y, library inverse x
static int poisson_run_from_predictor_bz(ocn_model* m, double dt) {
  PoissonSolver* s = m->solver;
  {
    ProfScope ps(m->ctx, "fft_forward");
    xfft_rhs_run(m, s->tw, s->spec, dt);
    yfft_run(m->ctx, s->tw, s->spec, s->Nxh, s->Ny, s->Nz, 0);
  }
  {
    ProfScope ps(m->ctx, "spectral_solve");
    tridiag_run(m, s, s->Nxh, s->Ny, s->lx, s->ly, 1.0 / ((double)s->Nx * s->Ny), s->spec, 1);
  }
  {
    ProfScope ps(m->ctx, "fft_backward");
    yfft_run(m->ctx, s->tw, s->spec, s->Nxh, s->Ny, s->Nz, 1);
#ifndef OCN_HOST_EMU
    if (dry_hipfftExecZ2D(s->xinv, (hipfftDoubleComplex*)s->spec, s->rhs) != HIPFFT_SUCCESS) {
      ocn_set_error(m->ctx, "hipfftExecZ2D (x inverse) failed");
      return OCN_EHIP;
    }
#else
    emu_xinv(s);
#endif
  }
  return OCN_OK;
}

// solve_for_pressure!(pNHS, solver, dt, U*)  (solve_for_pressure.jl:55-89)
int poisson_solve(ocn_model* m, double dt) {
  PoissonSolver* s = m->solver;
  int rc;
  if (s->cxy_bz) rc = poisson_run_from_predictor_bz(m, dt);
  else {
    launch_rhs(m, dt, s->rhs, m->g->topo[2] == OCN_BOUNDED ? 1 : 0);
    rc = run_solver(m);
  }
  if (rc) return rc;
  ProfScope ps(m->ctx, "copy_pressure");
  launch_copy_to_field(m, s->rhs, m->pNHS);
  return OCN_OK;
}

__global__ void k_mult_dz(GridDev g, double* __restrict__ r) {
  int i, j;
  ocn_cell_ij(i, j);
  const int k = blockIdx.z;
  if (i >= g.Nx || j >= g.Ny || k >= g.Nz) return;
  r[i + (size_t)g.Nx * (j + (size_t)g.Ny * k)] *= g_dzc(g, k);
}

// solve!(phi, solver, b) for a given source term (tests): rhs_dev, phi_dev compact (Nx,Ny,Nz) device arrays
int poisson_solve_rhs(ocn_model* m, const double* rhs_dev, double* phi_dev) {
  PoissonSolver* s = m->solver;
  size_t nr = (size_t)s->Nx * s->Ny * s->Nz;
  OCN_ASYNC(hipMemcpyAsync(s->rhs, rhs_dev, nr * sizeof(double), hipMemcpyDeviceToDevice, m->ctx->stream));
  if (m->g->topo[2] == OCN_BOUNDED) {  // set_source_term!: multiply by dz_c (fourier_tridiagonal_poisson_solver.jl:109-123)
    dim3 b(64, 4, 1), gr((s->Nx + 63) / 64, (s->Ny + 3) / 4, s->Nz);
    ocn_launch(k_mult_dz, gr, b, m->ctx->stream, m->gd, s->rhs);
  }
  int rc = run_solver(m);
  if (rc) return rc;
  OCN_ASYNC(hipMemcpyAsync(phi_dev, s->rhs, nr * sizeof(double), hipMemcpyDeviceToDevice, m->ctx->stream));
  return OCN_OK;
}
