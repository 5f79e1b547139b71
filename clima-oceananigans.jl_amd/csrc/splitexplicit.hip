// splitexplicit.hip -- the HydrostaticFreeSurfaceModel with a SplitExplicitFreeSurface (BASELINE config 5): grids and fields, the
// free surface's barotropic sub-cycle, and the model's whole AB2 time step, on one GPU or on latitude bands.
//
//   reference (paths relative to /root/reference/src)                                         here
//   Grids/latitude_longitude_grid.jl:174-213,418-445 (regular longitude / latitude, precomputed metrics)   ocn_hgrid_create (+ bands)
//   BoundaryConditions/fill_halo_regions*.jl, default conditions, z / x / y                      hfield_fill, k_h_fill_*, hfield_exchange_y
//   Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface.jl:78-117,137-154            ocn_sefs_create
//   .../split_explicit_free_surface_kernels.jl:14-19  kernel 1                                    k_se_uv
//                                              :21-29  kernel 2                                   k_se_eta
//                                              :31-58  substep!                                   sefs_substep_plain / k_se_uv_fused /
//                                                                                                 k_se_substep1 / k_se_multi
//                                              :63-81  barotropic_mode!                           k_se_vsum (+ fills)
//                                              :83-87  set_average_to_zero!
//                                              :89-113 corrector                                  k_se_correct
//                                              :124-171 split_explicit_free_surface_step!         ocn_sefs_step, sefs_step_tail
//   .../calculate_hydrostatic_free_surface_tendencies.jl, hydrostatic_free_surface_tendency_kernel_functions.jl,
//   Advection/vector_invariant_advection.jl, Coriolis/hydrostatic_spherical_coriolis.jl, Advection/tracer_advection_operators.jl
//                                                                                                 k_hy_Guv, k_hy_Gc, k_hy_Gc_hi
//   .../hydrostatic_free_surface_ab2_step.jl:15-48, TimeSteppers/quasi_adams_bashforth_2.jl:70-166  ocn_hydro_ab2_step, k_hy_ab2, k_hy_momentum,
//                                                                                                 k_hy_tracers, ocn_hydro_time_step
//   .../compute_w_from_continuity.jl:31-36, NonhydrostaticModels/update_hydrostatic_pressure.jl:10-18,
//   .../update_hydrostatic_free_surface_model_state.jl:21-48                                     k_hy_w, k_hy_pressure, hydro_update_state
//   Distributed/ (Partition by latitude)                                                          bands: hfield_exchange_y, hydro_allgather_rows,
//                                                                                                 band_refresh
//
// The loop over the substeps is the hot part of the free surface: 200 substeps of five tiny launches each (fill eta, kernel 1, fill U,
// fill V, kernel 2) in the reference -- 1000 launches of a few microseconds per time step, all latency.  ocn_sefs_substeps runs the same
// arithmetic as two launches per substep, as one (eta, U, V double buffered), or as four substeps per launch on tiles with ghost rings
// (k_se_multi), and replays the whole train from a hipGraph; every form leaves exactly the bits the launch-by-launch path leaves, halos
// included (tests/test_reference_split_explicit.py).  The kernels of this file are compiled without contraction into FMAs and divide
// through correctly rounded reciprocals (hy_div), so they also agree with the NumPy oracle bit for bit.
#include "internal.h"
#include "stencils.h"

// kernels whose launch-by-launch and fused forms (and the NumPy oracle) must round identically: no contraction into FMAs
// OCN_UNIFORM (compat.h): a value every lane of the wave holds alike -- a row index with 64-wide rows of threads (blockDim.x == 64).
// Told to the compiler, the per-row metric loads become scalar loads (one per wave through the scalar cache) instead of 64-lane
// vector loads.
#if defined(__clang__)
#define OCN_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define OCN_NO_CONTRACT
#endif

// x / d, correctly rounded, from the correctly rounded reciprocal r = RN(1 / d) of a divisor that is a per-row or per-level constant:
// q = RN(x r) is a faithful quotient, the remainder x - q d is exact in one fused multiply-add, and RN(q + rem r) is then the
// correctly rounded quotient (Markstein's theorem) -- the bits of the IEEE division the oracle performs, at three instructions
// instead of the ~15 of a double-precision division (k_hy_Guv holds 17 of them per cell).
__device__ inline double hy_div(double x, double d, double r) {
  const double q = x * r;
  return fma(fma(-q, d, x), r, q);
}

enum { HG_RECT = 0, HG_LATLON = 1 };

struct ocn_hgrid {
  ocn_ctx* ctx;
  // Fields and free surfaces keep using their grid (its stream, its metrics) until they are destroyed themselves, and a host
  // language with a garbage collector finalises a dropped object graph in no particular order: the grid is reference counted,
  // ocn_hgrid_destroy only gives up the caller's reference.
  int refs = 1;
  int kind;
  int N[3], H[3], topo[3];
  // latitude bands (y-slabs) over the context's ranks: N[1] is the LOCAL number of rows, rows j0 + 1 .. j0 + N[1] of gNy; a band
  // keeps the Bounded shape of its fields (Face-y fields hold N[1] + 1 rows: the last one is the upper neighbour's first, or the wall)
  bool slab = false;
  int gNy = 0, j0 = 0;
  // an EXTENDED band (ocn_hgrid_desc.band_overlap = W): a stand-alone Bounded grid of the band's rows plus ext_lo / ext_hi rows of the
  // neighbouring bands; rank / nranks remembered for the refresh of those rows (slab stays false: its fills are a plain grid's)
  int ext_lo = 0, ext_hi = 0, own = 0, overlap = 0;
  bool wall_lo = true, wall_hi = true;      // the band touches the southern / northern wall (Bounded y) -- else a neighbour
  double *pack_s = nullptr, *pack_r = nullptr;
  size_t pack_n = 0;
  double x0[3], L[3], radius;
  bool z_regular;
  std::vector<double> nodeF[3], nodeC[3];          // incl. halos, entry [i - 1 + H] for reference index i
  std::vector<double> h_dxfc, h_dxcf, h_dyfc, h_dycf, h_azcc, h_dzc;   // per row j (entry [j - 1 + Hy]) / per level k (entry [k - 1])
  std::vector<double> h_azff, h_phif;                                  // Az^ff per row; latitude of the rows of faces [degrees] (lat-lon only)
  std::vector<double> h_dzf;                                           // dz^f[k] = zC[k] - zC[k-1], k = 1..Nz+1 (entry [k - 1]); needs Hz >= 1
  double *dxfc = nullptr, *dxcf = nullptr, *dyfc = nullptr, *dycf = nullptr, *azcc = nullptr, *dzc = nullptr, *dzf = nullptr, *azff = nullptr;   // device copies
  // correctly rounded reciprocals of the divisors of the tendency kernel (hy_div below)
  double *r_dxfc = nullptr, *r_dycf = nullptr, *r_azcc = nullptr, *r_azff = nullptr, *r_dzf = nullptr;
};

struct ocn_hfield {
  ocn_hgrid* g;
  int loc[3];          // OCN_CENTER / OCN_FACE; loc[2] == OCN_NOTHING: reduced in z
  int T[3], S[3];      // parent size, interior size
  double* d = nullptr;
  size_t n = 0;
  bool owned = true;
};

struct ocn_sefs {
  ocn_hgrid* g;
  double grav;
  int substeps;
  std::vector<double> wv, wf;
  ocn_hfield *eta, *U, *V, *etabar, *Ubar, *Vbar, *GU, *GV, *Hfc, *Hcf, *Hcc;
  double *eta2 = nullptr, *U2 = nullptr, *V2 = nullptr;   // second copies of eta, U, V: the one-launch substep reads one set and writes the other
  // hipGraph of a train of substeps: key = (dtau bits, first index, count)
  struct Train { uint64_t dtau_bits; int first, count, mode; void* exec; };
  std::vector<Train> trains;
  int64_t graph_replays = 0;
};

static int total_len(int loc, int topo, int N, int H) {
  if (loc == OCN_NOTHING) return 1;
  return (loc == OCN_FACE && topo == OCN_BOUNDED) ? N + 1 + 2 * H : N + 2 * H;
}
static int interior_len(int loc, int topo, int N) {
  if (loc == OCN_NOTHING) return 1;
  return (loc == OCN_FACE && topo == OCN_BOUNDED) ? N + 1 : N;
}

// regular axis: Grids/grid_generation.jl:77-107 (L / N once, nodes as an evenly spaced range incl. halos)
static void regular_axis(double c1, double L, int N, int H, int topo, std::vector<double>& F, std::vector<double>& C, double& d) {
  d = (double)((long double)L / N);
  const int TF = total_len(OCN_FACE, topo, N, H), TC = total_len(OCN_CENTER, topo, N, H);
  const long double dl = (long double)L / N;
  const double Fm = (double)((long double)c1 - H * dl);
  const double Fp = (double)((long double)c1 - H * dl + (long double)L + (topo == OCN_BOUNDED ? 2 * H : 2 * H - 1) * dl);
  const double Cm = (double)((long double)c1 - H * dl + dl / 2);
  const double Cp = (double)((long double)c1 - H * dl + dl / 2 + (long double)L + dl * (2 * H - 1));
  F.resize(TF);
  C.resize(TC);
  for (int i = 0; i < TF; ++i) F[i] = TF > 1 ? Fm + i * ((Fp - Fm) / (TF - 1)) : Fm;
  if (TF > 1) F[TF - 1] = Fp;
  for (int i = 0; i < TC; ++i) C[i] = TC > 1 ? Cm + i * ((Cp - Cm) / (TC - 1)) : Cm;
  if (TC > 1) C[TC - 1] = Cp;
}

static int upload(ocn_ctx* ctx, const std::vector<double>& h, double** d) {
  OCN_HIP_CHECK(ctx, hipMalloc((void**)d, (h.size() ? h.size() : 1) * sizeof(double)));
  if (!h.empty()) OCN_HIP_CHECK(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  return OCN_OK;
}

// ---- halo fills in x and y (every level of a 3-D field; one level of a reduced field) -----------------------------------------
// periodic: fill_halo_regions_periodic.jl:37-65 -- one thread per line, the H copies in sequence, whole extent of the other dims
__global__ void k_h_fill_periodic(double* p, int dim, int N, int H, int Tx, int Ty, int Tz) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  const long sy = Tx, sz = (long)Tx * Ty;
  if (k >= Tz) return;
  long base, st;
  if (dim == 0) {
    if (a >= Ty) return;
    base = (long)a * sy + k * sz;
    st = 1;
  } else {
    if (a >= Tx) return;
    base = a + k * sz;
    st = sy;
  }
  for (int i = 0; i < H; ++i) {
    p[base + i * st] = p[base + (N + i) * st];
    p[base + (N + H + i) * st] = p[base + (H + i) * st];
  }
}
// Bounded: Center -> no-flux (fill_halo_regions_flux.jl:16-35, first halo cell only); Face -> impenetrable
// (fill_halo_regions_open.jl:34-39: the two boundary faces are zeroed).  Launched over the INTERIOR cells of the other direction.
__global__ void k_h_fill_bounded(double* p, int dim, int face, int N, int H, int No, int Ho, int Tx, int Ty, int k0, int nk, int sides) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, k = k0 + blockIdx.y;      // interior levels only (the `:xz` / `:yz` launch)
  if (a >= No || (int)blockIdx.y >= nk) return;
  const long sy = Tx, sz = (long)Tx * Ty;
  const long st = dim == 0 ? 1 : sy;
  const long base = (dim == 0 ? (long)(a + Ho) * sy : (long)(a + Ho)) + k * sz;
  // sides: bit 0 the lower boundary, bit 1 the upper one (a latitude band fills only the walls it touches)
  if (face) {
    if (sides & 1) p[base + (long)H * st] = 0.0;            // face 1
    if (sides & 2) p[base + (long)(H + N) * st] = 0.0;      // face N + 1
  } else {
    if (sides & 1) p[base + (long)(H - 1) * st] = p[base + (long)H * st];               // c[0] = c[1]
    if (sides & 2) p[base + (long)(H + N) * st] = p[base + (long)(H + N - 1) * st];     // c[N+1] = c[N]
  }
}

// rows of a latitude band travelling to / from the neighbouring bands: whole parent rows (x halos included), every parent level.
// pack: side 0 = the top H interior rows (for the upper neighbour's southern halo), side 1 = the bottom H (+1 for Face-y fields:
// the neighbour's extra row) interior rows (for the lower neighbour's northern rows); unpack: side 0 = rows below row 1 (from the
// lower neighbour), side 1 = rows above row N (from the upper one)
__global__ void k_h_pack_rows(double* p, double* buf, int Tx, int Ty, int Tz, int N, int H, int extra, int side, int unpack) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y, k = blockIdx.z;
  const int nrow = side == 0 ? H : H + extra;
  if (x >= Tx || h >= nrow || k >= Tz) return;
  int row;
  if (!unpack) row = side == 0 ? N + h : H + h;                     // parent rows [N, N + H) / [H, 2H + extra)
  else row = side == 0 ? h : N + H + h;                             // parent rows [0, H) / [N + H, N + 2H + extra)
  const long ip = x + (long)row * Tx + (long)k * Tx * Ty, ib = x + (long)Tx * (h + (long)nrow * k);
  if (unpack) p[ip] = buf[ib];
  else buf[ib] = p[ip];
}

// z (always Bounded here), over i = 1..Nx, j = 1..Ny of the grid (the `:xy` launch): Center -> no-flux, Face -> the default
// impenetrable condition of a ZFaceField zeroes faces 1 and Nz + 1
__global__ void k_h_fill_z(double* p, int face, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, int Tx, int Ty) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nx || j >= Ny) return;
  const long sz = (long)Tx * Ty, c = (i + Hx) + (long)(j + Hy) * Tx;
  if (face) {
    p[c + (long)Hz * sz] = 0.0;
    p[c + (long)(Hz + Nz) * sz] = 0.0;
  } else {
    p[c + (long)(Hz - 1) * sz] = p[c + (long)Hz * sz];
    p[c + (long)(Hz + Nz) * sz] = p[c + (long)(Hz + Nz - 1) * sz];
  }
}

static int hfield_exchange_y(ocn_hfield* f);

static void hfield_fill(ocn_hfield* f) {
  ocn_hgrid* g = f->g;
  hipStream_t s = g->ctx->stream;
  if (f->loc[2] != OCN_NOTHING && (g->H[2] > 0 || f->loc[2] == OCN_FACE))
    ocn_launch(k_h_fill_z, dim3((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1), dim3(64, 4, 1), s, f->d, f->loc[2] == OCN_FACE ? 1 : 0, g->N[0], g->N[1],
               g->N[2], g->H[0], g->H[1], g->H[2], f->T[0], f->T[1]);
  auto bounded = [&](int d, int sides) {
    const int o = 1 - d;
    if (!sides || (g->H[d] == 0 && f->loc[d] == OCN_CENTER)) return;
    const int k0 = f->loc[2] == OCN_NOTHING ? 0 : g->H[2], nk = f->loc[2] == OCN_NOTHING ? 1 : g->N[2];
    ocn_launch(k_h_fill_bounded, dim3((g->N[o] + 127) / 128, nk, 1), dim3(128, 1, 1), s, f->d, d, f->loc[d] == OCN_FACE ? 1 : 0, g->N[d], g->H[d],
               g->N[o], g->H[o], f->T[0], f->T[1], k0, nk, sides);
  };
  auto periodic = [&](int d) {
    if (g->H[d] == 0) return;
    const int na = d == 0 ? f->T[1] : f->T[0];
    ocn_launch(k_h_fill_periodic, dim3((na + 127) / 128, f->T[2], 1), dim3(128, 1, 1), s, f->d, d, g->N[d], g->H[d], f->T[0], f->T[1], f->T[2]);
  };
  if (g->slab) {
    // a latitude band: Bounded x first (interior rows), then y -- the walls this band touches and whole rows from the neighbouring
    // bands (their x halos included: fresh where x is Bounded) --, Periodic x last over the whole parent, received rows included
    if (g->topo[0] != OCN_PERIODIC) bounded(0, 3);
    bounded(1, (g->wall_lo ? 1 : 0) | (g->wall_hi ? 2 : 0));
    if (int rc = hfield_exchange_y(f))
      if (!g->ctx->sticky_rc) g->ctx->sticky_rc = rc;
    if (g->topo[0] == OCN_PERIODIC) periodic(0);
    return;
  }
  // non-periodic directions first (fill_halo_regions.jl:76-99)
  int order[2] = {0, 1};
  if (g->topo[0] == OCN_PERIODIC && g->topo[1] != OCN_PERIODIC) { order[0] = 1; order[1] = 0; }
  for (int t = 0; t < 2; ++t) {
    const int d = order[t];
    if (g->topo[d] == OCN_PERIODIC) periodic(d);
    else bounded(d, 3);
  }
}

// halo rows of a latitude band: one grouped exchange with the two neighbouring bands (collective: every rank of the context calls it)
static int hfield_exchange_y(ocn_hfield* f) {
  ocn_hgrid* g = f->g;
  ocn_ctx* c = g->ctx;
  const int H = g->H[1], N = g->N[1];
  if (H == 0 && f->loc[1] != OCN_FACE) return OCN_OK;
  const int R = c->nranks, r = c->rank;
  const bool per = g->topo[1] == OCN_PERIODIC;
  const int up = (r + 1 < R) ? r + 1 : (per ? 0 : -1), dn = (r > 0) ? r - 1 : (per ? R - 1 : -1);
  const int extra = (f->loc[1] == OCN_FACE && !per) ? 1 : 0;
  const size_t row = (size_t)f->T[0] * f->T[2];
  const size_t nA = (size_t)H * row, nB = (size_t)(H + extra) * row;      // block sent upwards / downwards
  const size_t need = nA + nB;
  if (need > g->pack_n) {
    hipStreamSynchronize(c->stream);
    hipFree(g->pack_s);
    hipFree(g->pack_r);
    g->pack_s = g->pack_r = nullptr;
    g->pack_n = 0;
    if (hipMalloc((void**)&g->pack_s, need * sizeof(double)) != hipSuccess || hipMalloc((void**)&g->pack_r, need * sizeof(double)) != hipSuccess) {
      ocn_set_error(c, "band halo staging allocation failed");
      return OCN_ENOMEM;
    }
    g->pack_n = need;
  }
  const dim3 b(64, 1, 1);
  auto grid = [&](int nrow) { return dim3((f->T[0] + 63) / 64, nrow, f->T[2]); };
  std::vector<CommOp> sends, recvs;
  if (up >= 0 && H > 0) {
    ocn_launch(k_h_pack_rows, grid(H), b, c->stream, f->d, g->pack_s, f->T[0], f->T[1], f->T[2], N, H, extra, 0, 0);
    sends.push_back({g->pack_s, nA * sizeof(double), up, 0});
  }
  if (dn >= 0 && H + extra > 0) {
    ocn_launch(k_h_pack_rows, grid(H + extra), b, c->stream, f->d, g->pack_s + nA, f->T[0], f->T[1], f->T[2], N, H, extra, 1, 0);
    sends.push_back({g->pack_s + nA, nB * sizeof(double), dn, 1});
  }
  if (dn >= 0 && H > 0) recvs.push_back({g->pack_r, nA * sizeof(double), dn, 0});
  if (up >= 0 && H + extra > 0) recvs.push_back({g->pack_r + nA, nB * sizeof(double), up, 1});
  if (int rc = comm_exchange(c, sends, recvs)) return rc;
  if (dn >= 0 && H > 0) ocn_launch(k_h_pack_rows, grid(H), b, c->stream, f->d, g->pack_r, f->T[0], f->T[1], f->T[2], N, H, extra, 0, 1);
  if (up >= 0 && H + extra > 0)
    ocn_launch(k_h_pack_rows, grid(H + extra), b, c->stream, f->d, g->pack_r + nA, f->T[0], f->T[1], f->T[2], N, H, extra, 1, 1);
  return OCN_OK;
}

// ---- the two substep kernels ---------------------------------------------------------------------------------------------
struct SeArgs {
  double *eta, *U, *V, *etabar, *Ubar, *Vbar;
  const double *GU, *GV, *Hfc, *Hcf;
  const double *dxfc, *dycf, *dyfc, *dxcf, *azcc;   // per row, entry [j + Hy] for the 0-based row j
  const double *r_dxfc, *r_dycf;                    // correctly rounded reciprocals of dxfc, dycf (hy_div)
  int Nx, Ny, Hx, Hy;
  long se, su, sv;                                   // row pitch of the Center-Center, Face-Center and Center-Face parents
  double g, dtau, wv, wf;
  int xper, yper;
};

// kernel 1 (:14-19):  U += dtau (-g H^fc d_x eta + G^U),  V += dtau (-g H^cf d_y eta + G^V)   over i = 1..Nx, j = 1..Ny
__global__ void k_se_uv(SeArgs a) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= a.Nx || j >= a.Ny) return;
  const long ce = (i + a.Hx) + (long)(j + a.Hy) * a.se, cu = (i + a.Hx) + (long)(j + a.Hy) * a.su, cv = (i + a.Hx) + (long)(j + a.Hy) * a.sv;
  const double e0 = a.eta[ce];
  const double ddx = (e0 - a.eta[ce - 1]) / a.dxfc[j + a.Hy];
  const double ddy = (e0 - a.eta[ce - a.se]) / a.dycf[j + a.Hy];
  a.U[cu] += a.dtau * (-a.g * a.Hfc[cu] * ddx + a.GU[cu]);
  a.V[cv] += a.dtau * (-a.g * a.Hcf[cv] * ddy + a.GV[cv]);
}

// kernel 2 (:21-29):  eta -= dtau div_xy(U, V);  averages
__global__ void k_se_eta(SeArgs a) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= a.Nx || j >= a.Ny) return;
  const long ce = (i + a.Hx) + (long)(j + a.Hy) * a.se, cu = (i + a.Hx) + (long)(j + a.Hy) * a.su, cv = (i + a.Hx) + (long)(j + a.Hy) * a.sv;
  const int r = OCN_UNIFORM(j + a.Hy);
  const double u0 = a.U[cu], v0 = a.V[cv];
  const double div = 1.0 / a.azcc[r] * ((a.dyfc[r] * a.U[cu + 1] - a.dyfc[r] * u0) + (a.dxcf[r + 1] * a.V[cv + a.sv] - a.dxcf[r] * v0));
  const double e1 = a.eta[ce] - a.dtau * div;
  a.eta[ce] = e1;
  a.Ubar[cu] += a.wv * u0;
  a.Vbar[cv] += a.wv * v0;
  a.etabar[ce] += a.wf * e1;
}

// The eta fill, kernel 1 and the U / V fills of one substep in ONE pass (Periodic x; y Periodic or Bounded; N >= H in the
// periodic directions).  A thread owns cell (i, j), i = 0..Nx-1, j = 0..Ny-1 (+ the row j = Ny when y is Bounded: the north
// boundary face of V and the no-flux halo row of eta / U are written by the threads of row Ny-1):
//   * eta of the west / south neighbour is read with the periodic wrap (what the eta fill would have put into the halo) or,
//     at a wall, is the no-flux copy (d_y eta = 0 across the wall: V at face 1 is then zeroed by its impenetrable fill anyway);
//   * it writes eta's halo images of its own cell (the fill of eta that precedes kernel 1 in the reference), its new U and
//     V, and their halo images (the fills that follow kernel 1): value-for-value what the five launches leave.
__global__ void k_se_uv_fused(SeArgs a) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= a.Nx || j >= a.Ny) return;
  const int Nx = a.Nx, Ny = a.Ny, Hx = a.Hx, Hy = a.Hy;
  const int iw = i > 0 ? i - 1 : Nx - 1;                         // x is Periodic
  const int js = j > 0 ? j - 1 : (a.yper ? Ny - 1 : 0);          // y wall: eta[0] = eta[1] (no-flux fill)
  const long ce = (i + Hx) + (long)(j + Hy) * a.se, cu = (i + Hx) + (long)(j + Hy) * a.su, cv = (i + Hx) + (long)(j + Hy) * a.sv;
  const double e0 = a.eta[ce];
  const double ew = a.eta[(iw + Hx) + (long)(j + Hy) * a.se], es = a.eta[(i + Hx) + (long)(js + Hy) * a.se];
  const double ddx = (e0 - ew) / a.dxfc[j + Hy];
  const double ddy = (e0 - es) / a.dycf[j + Hy];
  double un = a.U[cu] + a.dtau * (-a.g * a.Hfc[cu] * ddx + a.GU[cu]);
  double vn = a.V[cv] + a.dtau * (-a.g * a.Hcf[cv] * ddy + a.GV[cv]);
  if (!a.yper && j == 0) vn = 0.0;                               // impenetrable south face (fill after kernel 1)
  // own values and their images: the x images live in the columns i - Nx (if i >= Nx - Hx) and i + Nx (if i < Hx); the y images
  // likewise when y is Periodic; with a wall in y the first halo row copies the edge row (eta, U: no-flux) / the north face of V is 0.
  const bool xw = i >= Nx - Hx, xe = i < Hx;
  auto put = [&](double* p, long pitch, double val, int jj, bool own = true) {
    const long row = (long)(jj + Hy) * pitch;
    if (own) p[(i + Hx) + row] = val;
    if (xw) p[(i - Nx + Hx) + row] = val;
    if (xe) p[(i + Nx + Hx) + row] = val;
  };
  put(a.eta, a.se, e0, j, false);       // interior cells of eta are read by the neighbours in this pass: images only
  put(a.U, a.su, un, j);
  put(a.V, a.sv, vn, j);
  if (a.yper) {
    if (j >= Ny - Hy) { put(a.eta, a.se, e0, j - Ny); put(a.U, a.su, un, j - Ny); put(a.V, a.sv, vn, j - Ny); }
    if (j < Hy) { put(a.eta, a.se, e0, j + Ny); put(a.U, a.su, un, j + Ny); put(a.V, a.sv, vn, j + Ny); }
  } else {
    // no-flux rows of eta and U over the interior columns only (the launch range of the Bounded fill), then their periodic
    // x images come from the x fill that runs AFTER the Bounded one: over the whole y extent, halo rows included
    if (j == 0) { put(a.eta, a.se, e0, -1); put(a.U, a.su, un, -1); }
    if (j == Ny - 1) { put(a.eta, a.se, e0, Ny); put(a.U, a.su, un, Ny); put(a.V, a.sv, 0.0, Ny); }
  }
}

// One substep in ONE launch (Periodic x; y Periodic or Bounded): kernel 1 at the thread's own faces AND at its east / north
// neighbours' (the very expressions those threads evaluate: same operands, same bits), kernel 2 from the four values, the fills
// as halo images.  Reading and writing the same arrays would race (a thread's new eta against its neighbours' reads of the old
// one), so eta, U, V exist twice and a substep reads one set and writes the other: 128 B per cell instead of 152, one launch
// instead of two.  The new set's eta halo takes the images of the OLD eta -- what the reference's eta fill (before kernel 1) left
// there -- so every parent array keeps the bits of the five-launch substep at every substep.
struct SeArgs1 {
  SeArgs a;
  const double *etaI, *UI, *VI;   // read
  double *etaO, *UO, *VO;         // written
};
__global__ void k_se_substep1(SeArgs1 b) {
  OCN_NO_CONTRACT
  const SeArgs& a = b.a;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= a.Nx || j >= a.Ny) return;
  const int Nx = a.Nx, Ny = a.Ny, Hx = a.Hx, Hy = a.Hy;
  const int iw = i > 0 ? i - 1 : Nx - 1, ie = i + 1 < Nx ? i + 1 : 0;
  const int js = j > 0 ? j - 1 : (a.yper ? Ny - 1 : 0);
  const bool north_wall = !a.yper && j == Ny - 1;
  const int jn = j + 1 < Ny ? j + 1 : (a.yper ? 0 : j);          // unused at a north wall
  const int r = OCN_UNIFORM(j + Hy), rn = OCN_UNIFORM(jn + Hy);
  auto E = [&](int ii, int jj) { return b.etaI[(ii + Hx) + (long)(jj + Hy) * a.se]; };
  const long cu = (i + Hx) + (long)r * a.su, cv = (i + Hx) + (long)r * a.sv, ce = (i + Hx) + (long)r * a.se;
  const long cue = (ie + Hx) + (long)r * a.su, cvn = (i + Hx) + (long)rn * a.sv;
  const double e0 = E(i, j), ew = E(iw, j), es = E(i, js), ee = E(ie, j);
  const double u0 = b.UI[cu] + a.dtau * (-a.g * a.Hfc[cu] * hy_div(e0 - ew, a.dxfc[r], a.r_dxfc[r]) + a.GU[cu]);
  const double u1 = b.UI[cue] + a.dtau * (-a.g * a.Hfc[cue] * hy_div(ee - e0, a.dxfc[r], a.r_dxfc[r]) + a.GU[cue]);
  double v0 = b.VI[cv] + a.dtau * (-a.g * a.Hcf[cv] * hy_div(e0 - es, a.dycf[r], a.r_dycf[r]) + a.GV[cv]);
  if (!a.yper && j == 0) v0 = 0.0;                               // impenetrable south face
  double v1 = 0.0;                                               // impenetrable north face
  if (!north_wall) {
    const double en = E(i, jn);
    v1 = b.VI[cvn] + a.dtau * (-a.g * a.Hcf[cvn] * hy_div(en - e0, a.dycf[rn], a.r_dycf[rn]) + a.GV[cvn]);
  }
  const double div = 1.0 / a.azcc[r] * ((a.dyfc[r] * u1 - a.dyfc[r] * u0) + (a.dxcf[r + 1] * v1 - a.dxcf[r] * v0));
  const double e1 = e0 - a.dtau * div;
  a.Ubar[cu] += a.wv * u0;
  a.Vbar[cv] += a.wv * v0;
  a.etabar[ce] += a.wf * e1;
  const bool xw = i >= Nx - Hx, xe = i < Hx;
  auto put = [&](double* p, long pitch, double val, int jj, bool own = true) {
    const long row = (long)(jj + Hy) * pitch;
    if (own) p[(i + Hx) + row] = val;
    if (xw) p[(i - Nx + Hx) + row] = val;
    if (xe) p[(i + Nx + Hx) + row] = val;
  };
  b.etaO[ce] = e1;
  put(b.etaO, a.se, e0, j, false);                               // halo images of the OLD eta (see above)
  put(b.UO, a.su, u0, j);
  put(b.VO, a.sv, v0, j);
  if (a.yper) {
    if (j >= Ny - Hy) { put(b.etaO, a.se, e0, j - Ny); put(b.UO, a.su, u0, j - Ny); put(b.VO, a.sv, v0, j - Ny); }
    if (j < Hy) { put(b.etaO, a.se, e0, j + Ny); put(b.UO, a.su, u0, j + Ny); put(b.VO, a.sv, v0, j + Ny); }
  } else {
    if (j == 0) { put(b.etaO, a.se, e0, -1); put(b.UO, a.su, u0, -1); }
    if (j == Ny - 1) { put(b.etaO, a.se, e0, Ny); put(b.UO, a.su, u0, Ny); put(b.VO, a.sv, 0.0, Ny); }
  }
}


// Up to G substeps in ONE launch, interior cells only (Periodic x; y Periodic or Bounded).  A workgroup of 64 x 16 threads, RPT rows per
// thread, owns a tile of (64 - 2 G) x (16 RPT - 2 G) cells and carries a ring of G ghost cells around it: every thread keeps eta, U, V
// (and the constants of its cells) in registers, the neighbours' values pass through LDS, and each substep spoils one more ring of ghosts
// from the outside in -- after n <= G substeps the tile itself still holds exactly what n launches of k_se_substep1 would have left (the
// same expressions on the same operands).  <4, 1>: 56 x 8 cells, 50 B of HBM traffic per cell and substep instead of 128, a quarter of
// the launches (grids of 64 x 16 cells and more); <8, 2>: 48 x 16 cells, 28 B, an eighth (64 x 32 and more).  Halos are not touched: a
// train ends with one k_se_substep1, which writes every halo image the reference's fills leave.
#define SE_MS 8                      // the most substeps a launch takes
struct SeArgsM {
  SeArgs1 b;
  int nsub;
  double wv[SE_MS], wf[SE_MS];
};
template <int G, int RPT>
__global__ void __launch_bounds__(1024) k_se_multi(SeArgsM m) {
  OCN_NO_CONTRACT
  constexpr int BX = 64, BY = 16 * RPT, TX = BX - 2 * G, TY = BY - 2 * G, LS = BX + 1;
  OCN_SHARED double sE[BY * LS], sU[BY * LS], sV[BY * LS];
  const SeArgs& a = m.b.a;
  const int tx = threadIdx.x;
  const int Nx = a.Nx, Ny = a.Ny, Hx = a.Hx, Hy = a.Hy;
  const int i = (int)blockIdx.x * TX + tx - G;
  const int gi = i < 0 ? i + Nx : (i >= Nx ? i - Nx : i);           // Periodic x (Nx >= BX: one wrap is enough)
  double e[RPT], u[RPT], v[RPT], hfc[RPT], hcf[RPT], gu[RPT], gv[RPT], ub[RPT], vb[RPT], eb[RPT];
  double dxfc[RPT], dycf[RPT], dyfc[RPT], dxcf0[RPT], dxcf1[RPT], azcc[RPT], rdxfc[RPT], rdycf[RPT];
  long cu[RPT], cv[RPT], ce[RPT];
  bool own[RPT], rowok[RPT], south_face[RPT];
  int me[RPT], west[RPT], south[RPT], east[RPT], north[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int ty = threadIdx.y + 16 * q;
    const int j = (int)blockIdx.y * TY + ty - G;
    int gj = j;
    rowok[q] = true;
    if (a.yper) gj = j < 0 ? j + Ny : (j >= Ny ? j - Ny : j);
    else rowok[q] = j >= 0 && j < Ny;
    if (!rowok[q]) gj = j < 0 ? 0 : Ny - 1;                         // any valid row: the loads below stay in bounds, the values are dropped
    own[q] = tx >= G && tx < G + TX && ty >= G && ty < G + TY && i < Nx && j < Ny;
    const int r = OCN_UNIFORM(gj + Hy);
    cu[q] = (gi + Hx) + (long)r * a.su; cv[q] = (gi + Hx) + (long)r * a.sv; ce[q] = (gi + Hx) + (long)r * a.se;
    e[q] = m.b.etaI[ce[q]]; u[q] = m.b.UI[cu[q]]; v[q] = m.b.VI[cv[q]];
    hfc[q] = a.Hfc[cu[q]]; hcf[q] = a.Hcf[cv[q]]; gu[q] = a.GU[cu[q]]; gv[q] = a.GV[cv[q]];
    dxfc[q] = a.dxfc[r]; dycf[q] = a.dycf[r]; dyfc[q] = a.dyfc[r]; dxcf0[q] = a.dxcf[r]; dxcf1[q] = a.dxcf[r + 1]; azcc[q] = a.azcc[r];
    rdxfc[q] = a.r_dxfc[r]; rdycf[q] = a.r_dycf[r];
    ub[q] = vb[q] = eb[q] = 0.0;
    if (own[q]) { ub[q] = a.Ubar[cu[q]]; vb[q] = a.Vbar[cv[q]]; eb[q] = a.etabar[ce[q]]; }
    if (!rowok[q]) { e[q] = 0.0; u[q] = 0.0; v[q] = 0.0; }
    south_face[q] = !a.yper && j == 0;                              // the impenetrable face of the southern wall
    me[q] = ty * LS + tx;
    west[q] = tx > 0 ? me[q] - 1 : me[q];
    south[q] = ty > 0 ? me[q] - LS : me[q];
    east[q] = tx + 1 < BX ? me[q] + 1 : me[q];
    north[q] = ty + 1 < BY ? me[q] + LS : me[q];
  }
  for (int n = 0; n < m.nsub; ++n) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) sE[me[q]] = e[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const double ew = sE[west[q]], es = sE[south[q]];
      u[q] = u[q] + a.dtau * (-a.g * hfc[q] * hy_div(e[q] - ew, dxfc[q], rdxfc[q]) + gu[q]);
      v[q] = v[q] + a.dtau * (-a.g * hcf[q] * hy_div(e[q] - es, dycf[q], rdycf[q]) + gv[q]);
      if (south_face[q] || !rowok[q]) v[q] = 0.0;                   // rows beyond a wall hold V = 0: the wall's north / south face
      sU[me[q]] = u[q];
      sV[me[q]] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const double ue = sU[east[q]], vn = sV[north[q]];
      const double div = 1.0 / azcc[q] * ((dyfc[q] * ue - dyfc[q] * u[q]) + (dxcf1[q] * vn - dxcf0[q] * v[q]));
      e[q] = e[q] - a.dtau * div;
      ub[q] += m.wv[n] * u[q];
      vb[q] += m.wv[n] * v[q];
      eb[q] += m.wf[n] * e[q];
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q)
    if (own[q]) {
      m.b.etaO[ce[q]] = e[q];
      m.b.UO[cu[q]] = u[q];
      m.b.VO[cv[q]] = v[q];
      a.Ubar[cu[q]] = ub[q];
      a.Vbar[cv[q]] = vb[q];
      a.etabar[ce[q]] = eb[q];
    }
}

// ---- vertical integrals and the corrector ----------------------------------------------------------------------------------
// sum!(U, u * dz): level 1 first, then level by level (a sequential sum per column; coalesced across i).  With cm != 0 the
// summand is the AB2 combination (cn G^n + cm G^-) dz of calc_ab2_tendencies (:115) formed on the fly.
__global__ void k_se_vsum(double* out, const double* a, const double* b, double cn, double cm, const double* dzc, int Sx, int Sy, int Nz,
                          int Hx, int Hy, int Hz, long sy3, long sz3, long sy2) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Sx || j >= Sy) return;
  const long c3 = (i + Hx) + (long)(j + Hy) * sy3 + (long)Hz * sz3;
  double acc = 0.0;
  for (int k = 0; k < Nz; ++k) {
    const double q = b ? (cn * a[c3 + k * sz3] - cm * b[c3 + k * sz3]) : a[c3 + k * sz3];
    acc = k == 0 ? q * dzc[0] : acc + q * dzc[k];
  }
  out[(i + Hx) + (long)(j + Hy) * sy2] = acc;
}

// barotropic_split_explicit_corrector_kernel! (:89-95) over i = 1..Nx, j = 1..Ny, k = 1..Nz: u += (-U + U-bar) / H^fc, v likewise.
// One thread per column: the two quotients are those of every level of the column, formed once (the same operands give the same bits).
__global__ void k_se_correct(double* u, double* v, const double* U, const double* V, const double* Ub, const double* Vb, const double* Hfc,
                             const double* Hcf, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, long su3, long szu, long sv3, long szv,
                             long su2, long sv2) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nx || j >= Ny) return;
  const long cu2 = (i + Hx) + (long)(j + Hy) * su2, cv2 = (i + Hx) + (long)(j + Hy) * sv2;
  long cu3 = (i + Hx) + (long)(j + Hy) * su3 + (long)Hz * szu, cv3 = (i + Hx) + (long)(j + Hy) * sv3 + (long)Hz * szv;
  const double du = (-U[cu2] + Ub[cu2]) / Hfc[cu2], dv = (-V[cv2] + Vb[cv2]) / Hcf[cv2];
  for (int k = 0; k < Nz; ++k, cu3 += szu, cv3 += szv) {
    u[cu3] = u[cu3] + du;
    v[cv3] = v[cv3] + dv;
  }
}

__global__ void k_se_copy(double* dst, const double* src, size_t n) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) dst[q] = src[q];
}


// ---- second slice of config 5: the AB2 time step of the hydrostatic model around its tendency evaluation ---------------------
//   TimeSteppers/quasi_adams_bashforth_2.jl:158-166  ab2_step_field!                                   k_hy_ab2
//   hydrostatic_free_surface_ab2_step.jl:15-48       ab2_step! (barotropic mode, velocities, tracers, free surface)
//   compute_w_from_continuity.jl:31-36               _compute_w_from_continuity!                       k_hy_w
//   update_hydrostatic_pressure.jl:10-18             _update_hydrostatic_pressure!                     k_hy_pressure
//   TimeSteppers/store_tendencies.jl:14-36           G^- <- G^n                                        k_se_copy / fused
//   update_hydrostatic_free_surface_model_state.jl:21-48  update_state!                                hydro_update_state
// The arithmetic of these kernels is kept free of contraction so that the launch-by-launch sequence, the fused passes and
// the NumPy oracle round identically (they are bound by HBM, the FMAs buy nothing).

struct HyBuoy {          // buoyancy_perturbation: 0 none, 1 BuoyancyTracer (b = T), 2 SeawaterBuoyancy with a LinearEquationOfState
  int kind;
  double g, alpha, beta;
};
__device__ inline double hy_b(const HyBuoy& q, double T, double S) {
  OCN_NO_CONTRACT
  if (q.kind == 1) return T;
  if (q.kind == 2) return q.g * (q.alpha * T - q.beta * S);
  return 0.0;
}
__device__ inline double hy_ab2(double f, double gn, double gm, double dt, double cn, double cm) {
  OCN_NO_CONTRACT
  return f + dt * (cn * gn - cm * gm);
}

// ab2_step_field! over i = 1..Nx, j = 1..Ny, k = 1..Nz of the grid (the boundary face of a Bounded direction is not stepped)
__global__ void k_hy_ab2(double* f, const double* gn, const double* gm, double dt, double cn, double cm, int Nx, int Ny, int Nz, int Hx, int Hy,
                         int Hz, long sy, long sz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
  if (i >= Nx || j >= Ny || k >= Nz) return;
  const long c = (i + Hx) + (long)(j + Hy) * sy + (long)(k + Hz) * sz;
  f[c] = hy_ab2(f[c], gn[c], gm[c], dt, cn, cm);
}

struct HyGrid {
  const double *dxcf, *dyfc, *azcc, *dzc, *dzf;
  int Nx, Ny, Nz, Hx, Hy, Hz;
};

// w[1] = 0, w[k] = w[k-1] - dz^c[k-1] div_xy(k-1): one thread per column, coalesced along x, marching upwards
__global__ void k_hy_w(HyGrid g, const double* u, const double* v, double* w, long syu, long szu, long syv, long szv, long syw, long szw) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int r = OCN_UNIFORM(j + g.Hy);       // blockDim.x == 64: one row per wave
  const double dy = g.dyfc[r], dxs = g.dxcf[r], dxn = g.dxcf[r + 1], ra = 1 / g.azcc[r];
  long cu = (i + g.Hx) + (long)r * syu + (long)g.Hz * szu, cv = (i + g.Hx) + (long)r * syv + (long)g.Hz * szv;
  long cw = (i + g.Hx) + (long)r * syw + (long)g.Hz * szw;
  double acc = 0.0;
  w[cw] = acc;
  for (int k = 0; k < g.Nz; ++k) {
    const double div = ra * ((dy * u[cu + 1] - dy * u[cu]) + (dxn * v[cv + syv] - dxs * v[cv]));
    acc = acc - g.dzc[k] * div;
    cw += szw;
    w[cw] = acc;
    cu += szu;
    cv += szv;
  }
}

// pHY'[Nz] = -I_z(b)[Nz+1] dz^f[Nz+1]; pHY'[k] = pHY'[k+1] - I_z(b)[k+1] dz^f[k+1]: one thread per column, marching downwards; the
// buoyancy of level Nz + 1 is read from the tracers' (filled) top halo
__global__ void k_hy_pressure(HyGrid g, HyBuoy q, const double* T, const double* S, double* p, long sy, long sz) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  long c = (i + g.Hx) + (long)(j + g.Hy) * sy + (long)(g.Hz + g.Nz) * sz;       // level Nz + 1
  double bup = hy_b(q, T ? T[c] : 0.0, S ? S[c] : 0.0), acc = 0.0;
  for (int k = g.Nz; k >= 1; --k) {
    c -= sz;
    const double b = hy_b(q, T ? T[c] : 0.0, S ? S[c] : 0.0);
    const double bf = (bup + b) / 2;
    acc = k == g.Nz ? -bf * g.dzf[k] : acc - bf * g.dzf[k];                        // dzf entry [k] = face k + 1
    p[c] = acc;
    bup = b;
  }
}

// per-level tables of the vertically implicit diffusion solve (k_hy_implicit below explains them)
struct HyImp {
  const double *a, *beta, *rbeta, *t;      // a[k]: lower diagonal below level k + 1 (0-based k = 0..Nz-2); beta[k], t[k]: level k
};

// fused pass over one velocity component: the barotropic mode of the velocity before the step (-> U), the vertical integral of
// the AB2 tendency (-> G^U), the AB2 step, the barotropic mode of the stepped velocity (-> Un, for the corrector) and G^- <- G^n;
// sums over the field's interior (incl. the boundary face of a Bounded direction), the step over the grid's cells
__global__ void k_hy_momentum(double* u, const double* gn, double* gm, double* U, double* GU, double* Un, double dt, double cn, double cm,
                              const double* dzc, int Sx, int Sy, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, long sy3, long sz3, long sy2,
                              HyImp imp, int implicit) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Sx || j >= Sy) return;
  const bool step = i < Nx && j < Ny;
  const long c0 = (i + Hx) + (long)(j + Hy) * sy3 + (long)Hz * sz3;
  long c = c0;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, phi = 0.0;
  // implicit (a vertically implicit viscosity): the stepped value is the right-hand side of the column's tridiagonal system, so the
  // forward elimination rides in this pass (phi_k = (u*_k - a_{k-1} phi_{k-1}) / beta_k), the back substitution and the barotropic
  // mode of the solved column follow below -- two sweeps fewer than stepping, solving and summing in three kernels
  const bool imp_on = implicit && step;
  for (int k = 0; k < Nz; ++k, c += sz3) {
    const double uo = u[c], n = gn[c], m = gm[c], dz = dzc[k];
    const double G = cn * n - cm * m;
    const double un = step ? hy_ab2(uo, n, m, dt, cn, cm) : uo;
    a0 = k == 0 ? uo * dz : a0 + uo * dz;
    a1 = k == 0 ? G * dz : a1 + G * dz;
    a2 = k == 0 ? un * dz : a2 + un * dz;
    if (step) {
      if (imp_on) {
        phi = k == 0 ? hy_div(un, imp.beta[0], imp.rbeta[0]) : hy_div(un - imp.a[k - 1] * phi, imp.beta[k], imp.rbeta[k]);
        u[c] = phi;
      } else {
        u[c] = un;
      }
      gm[c] = n;
    }
  }
  if (imp_on) {
    c = c0 + (long)(Nz - 1) * sz3;
    for (int k = Nz - 2; k >= 0; --k) {
      c -= sz3;
      phi = u[c] - imp.t[k + 1] * phi;
      u[c] = phi;
    }
    c = c0;
    for (int k = 0; k < Nz; ++k, c += sz3) {
      const double q = u[c] * dzc[k];
      a2 = k == 0 ? q : a2 + q;
    }
  }
  const long c2 = (i + Hx) + (long)(j + Hy) * sy2;
  U[c2] = a0;
  GU[c2] = a1;
  Un[c2] = a2;
}

// AB2 step, G^- <- G^n and the vertically implicit diffusion solve of one field in one kernel: up the column the stepped value feeds the
// forward elimination directly, down the column the back substitution (seven sweeps instead of nine for k_hy_ab2_store + k_hy_implicit)
__global__ void k_hy_ab2_implicit(double* f, const double* gn, double* gm, double dt, double cn, double cm, HyImp imp, int Nx, int Ny, int Nz,
                                  int Hx, int Hy, int Hz, long sy, long sz) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nx || j >= Ny) return;
  long c = (i + Hx) + (long)(j + Hy) * sy + (long)Hz * sz;
  double phi = 0.0;
  for (int k = 0; k < Nz; ++k, c += sz) {
    const double n = gn[c];
    const double cs = hy_ab2(f[c], n, gm[c], dt, cn, cm);
    gm[c] = n;
    phi = k == 0 ? hy_div(cs, imp.beta[0], imp.rbeta[0]) : hy_div(cs - imp.a[k - 1] * phi, imp.beta[k], imp.rbeta[k]);
    f[c] = phi;
  }
  c -= sz;
  for (int k = Nz - 2; k >= 0; --k) {
    c -= sz;
    phi = f[c] - imp.t[k + 1] * phi;
    f[c] = phi;
  }
}

// fused pass over the tracers that make the buoyancy (one or two), marching downwards: AB2 step, G^- <- G^n, and the hydrostatic
// pressure from the stepped values (the no-flux top halo equals level Nz)
__global__ void k_hy_tracers(HyGrid g, HyBuoy q, double* T, const double* gnT, double* gmT, double* S, const double* gnS, double* gmS, double* p,
                             double dt, double cn, double cm, long sy, long sz) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  long c = (i + g.Hx) + (long)(j + g.Hy) * sy + (long)(g.Hz + g.Nz) * sz;
  double bup = 0.0, acc = 0.0;
  for (int k = g.Nz; k >= 1; --k) {
    c -= sz;
    const double nT = gnT[c], t = hy_ab2(T[c], nT, gmT[c], dt, cn, cm);
    T[c] = t;
    gmT[c] = nT;
    double s = 0.0;
    if (S) {
      const double nS = gnS[c];
      s = hy_ab2(S[c], nS, gmS[c], dt, cn, cm);
      S[c] = s;
      gmS[c] = nS;
    }
    const double b = hy_b(q, t, s);
    if (k == g.Nz) bup = b;
    const double bf = (bup + b) / 2;
    acc = k == g.Nz ? -bf * g.dzf[k] : acc - bf * g.dzf[k];
    p[c] = acc;
    bup = b;
  }
}

// store_field_tendencies! (TimeSteppers/store_tendencies.jl:8-11): G^-[i, j, k] = G^n[i, j, k] over the grid's cells
__global__ void k_hy_store(double* gm, const double* gn, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, long sy, long sz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
  if (i >= Nx || j >= Ny || k >= Nz) return;
  const long c = (i + Hx) + (long)(j + Hy) * sy + (long)(k + Hz) * sz;
  gm[c] = gn[c];
}

// a tracer that does not enter the buoyancy: AB2 step and G^- <- G^n in one pass
__global__ void k_hy_ab2_store(double* f, const double* gn, double* gm, double dt, double cn, double cm, int Nx, int Ny, int Nz, int Hx, int Hy,
                               int Hz, long sy, long sz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
  if (i >= Nx || j >= Ny || k >= Nz) return;
  const long c = (i + Hx) + (long)(j + Hy) * sy + (long)(k + Hz) * sz;
  const double n = gn[c];
  f[c] = hy_ab2(f[c], n, gm[c], dt, cn, cm);
  gm[c] = n;
}

// ---- third slice: calculate_tendencies! (no closure, no forcing, no immersed boundary) ----------------------------------------
//   hydrostatic_free_surface_tendency_kernel_functions.jl:24-125   G_u, G_v, G_c
//   Advection/vector_invariant_advection.jl:25-80                  VectorInvariant (enstrophy- / energy-conserving)
//   Operators/vorticity_operators.jl:2-5                           zeta_3 at (Face, Face, Center)
//   Coriolis/hydrostatic_spherical_coriolis.jl:29-66, f_plane.jl:42-43
//   Advection/tracer_advection_operators.jl:33-37, centered_advective_fluxes.jl:31-33   flux-form CenteredSecondOrder
// One thread per cell; every operator keeps the reference's operand order, without contraction (see OCN_NO_CONTRACT).
struct HyPhys {
  int madv;          // 0 none, 1 VectorInvariant enstrophy-conserving, 2 energy-conserving, 3 WENO5(vector_invariant = VorticityStencil())
  int xb, yb, jrow0, gNy;   // madv 3: Bounded x / y (boundary buffer), global row of the band's first row and global row count
  int cor;           // 0 none, 1 HydrostaticSphericalCoriolis enstrophy-conserving, 2 energy-conserving, 3 FPlane
  int tadv;          // 0 none, 1 CenteredSecondOrder, 2 CenteredFourthOrder, 3 UpwindBiasedFifthOrder, 4 WENO5 (Z weights)
  double f0;
  const double* frow;   // f at the rows of (Face, Face) points
};
struct HyMetric {
  const double *dxfc, *dxcf, *dyfc, *dycf, *azcc, *azff, *dzc, *dzf;
  const double *r_dxfc, *r_dycf, *r_azcc, *r_azff, *r_dzf;      // correctly rounded reciprocals (host: 1.0 / x)
  int Nx, Ny, Nz, Hx, Hy, Hz;
};

__global__ void __launch_bounds__(256) k_hy_Guv(HyMetric g, HyPhys ph, const double* __restrict__ u, const double* __restrict__ v,
                                                const double* __restrict__ w, const double* __restrict__ p, double* __restrict__ Gu,
                                                double* __restrict__ Gv, long syu, long szu, long syv, long szv, long syc, long szc) {
  OCN_NO_CONTRACT
  // one thread per column marching upwards: the vertical-advection products of a level's upper face are kept for the next level (the
  // upper face of level k IS the lower face of level k + 1: same operands, same bits), so along z every 3-D value comes from HBM once
  // -- the one-thread-per-cell form pulled 2.2 times its algorithmic bytes through L2 (profiles/r03_pmc_config5.json)
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int r = OCN_UNIFORM(j + g.Hy);       // blockDim.x == 64: one row per wave
  long cu = (i + g.Hx) + (long)r * syu + (long)g.Hz * szu, cv = (i + g.Hx) + (long)r * syv + (long)g.Hz * szv;
  long cc = (i + g.Hx) + (long)r * syc + (long)g.Hz * szc;       // w and pHY' share the (Center, Center) row pitch
  const long szw = szc;
  int k = 0;
  auto U = [&](int di, int dj, int dk) { return u[cu + di + dj * syu + dk * szu]; };
  auto V = [&](int di, int dj, int dk) { return v[cv + di + dj * syv + dk * szv]; };
  auto W = [&](int di, int dj, int dk) { return w[cc + di + dj * syc + dk * szw]; };
  auto zeta = [&](int di, int dj) {
    const double circ = (g.dycf[r + dj] * V(di, dj, 0) - g.dycf[r + dj] * V(di - 1, dj, 0)) -
                        (g.dxfc[r + dj] * U(di, dj, 0) - g.dxfc[r + dj - 1] * U(di, dj - 1, 0));
    return hy_div(circ, g.azff[r + dj], g.r_azff[r + dj]);
  };
  auto sq = [](double x) { return x * x; };
  auto Kh = [&](int di, int dj) {
    return (0.5 * (sq(U(di, dj, 0)) + sq(U(di + 1, dj, 0))) + 0.5 * (sq(V(di, dj, 0)) + sq(V(di, dj + 1, 0)))) / 2;
  };
  auto Iy_dxv = [&](int di) { return 0.5 * (g.dxcf[r] * V(di, 0, 0) + g.dxcf[r + 1] * V(di, 1, 0)); };
  auto Ix_dyu = [&](int dj) { return 0.5 * (g.dyfc[r + dj] * U(0, dj, 0) + g.dyfc[r + dj] * U(1, dj, 0)); };
  auto Ix_dxv = [&](int dj) { return 0.5 * (g.dxcf[r + dj] * V(-1, dj, 0) + g.dxcf[r + dj] * V(0, dj, 0)); };
  auto Iy_dyu = [&](int di) { return 0.5 * (g.dyfc[r - 1] * U(di, -1, 0) + g.dyfc[r] * U(di, 0, 0)); };
  const double dxfc = g.dxfc[r], dycf = g.dycf[r], rdxfc = g.r_dxfc[r], rdycf = g.r_dycf[r];
  auto z2w = [&](int dk) {
    return (0.5 * (g.azcc[r] * W(-1, 0, dk) + g.azcc[r] * W(0, 0, dk))) * hy_div(U(0, 0, dk) - U(0, 0, dk - 1), g.dzf[k + dk], g.r_dzf[k + dk]);
  };
  auto z1w = [&](int dk) {
    return (0.5 * (g.azcc[r - 1] * W(0, -1, dk) + g.azcc[r] * W(0, 0, dk))) * hy_div(V(0, 0, dk) - V(0, 0, dk - 1), g.dzf[k + dk], g.r_dzf[k + dk]);
  };
  double z2w_lo = 0.0, z1w_lo = 0.0;         // zeta_2 w and zeta_1 w at the lower face of the current level
  if (ph.madv) {
    z2w_lo = z2w(0);
    z1w_lo = z1w(0);
  }
  for (; k < g.Nz; ++k, cu += szu, cv += szv, cc += szc) {
  double Au = 0.0, Av = 0.0;
  if (ph.madv) {
    double vvU, vvV;
    if (ph.madv == 1) {
      const double z00 = zeta(0, 0);
      vvU = hy_div(-(0.5 * (z00 + zeta(0, 1))) * (0.5 * (Iy_dxv(-1) + Iy_dxv(0))), dxfc, rdxfc);
      vvV = hy_div(+(0.5 * (z00 + zeta(1, 0))) * (0.5 * (Ix_dyu(-1) + Ix_dyu(0))), dycf, rdycf);
    } else if (ph.madv == 2) {
      const double z00 = zeta(0, 0);
      vvU = hy_div(-(0.5 * (z00 * Ix_dxv(0) + zeta(0, 1) * Ix_dxv(1))), dxfc, rdxfc);
      vvV = hy_div(+(0.5 * (z00 * Iy_dyu(0) + zeta(1, 0) * Iy_dyu(1))), dycf, rdycf);
    } else {
      // WENO5(vector_invariant = VorticityStencil()) (vector_invariant_advection.jl:54-66): transporting velocity times the upwind-biased
      // WENO5 interpolation of zeta to the velocity point (stencils.h recon5: parity with the oracle to round-off, like the WENO tracer
      // kernel), second order inside the boundary buffer of a Bounded direction (topologically_conditional_interpolation.jl:49-62)
      const double vhat = hy_div(0.5 * (Iy_dxv(-1) + Iy_dxv(0)), dxfc, rdxfc), uhat = hy_div(0.5 * (Ix_dyu(-1) + Ix_dyu(0)), dycf, rdycf);
      const int jg = ph.jrow0 + j + 1, ig = i + 1;                  // 1-based global indices of the buffer test
      {
        const bool pos = vhat > 0.0;
        double zi;
        if (ph.yb && !(pos ? outside_left(jg, ph.gNy, 2) : outside_right(jg, ph.gNy, 2))) zi = 0.5 * (zeta(0, 0) + zeta(0, 1));
        else zi = pos ? recon5<ADV_WENO_Z>(zeta(0, -2), zeta(0, -1), zeta(0, 0), zeta(0, 1), zeta(0, 2), true)
                      : recon5<ADV_WENO_Z>(zeta(0, 3), zeta(0, 2), zeta(0, 1), zeta(0, 0), zeta(0, -1), false);
        vvU = -(vhat * zi);
      }
      {
        const bool pos = uhat > 0.0;
        double zi;
        if (ph.xb && !(pos ? outside_left(ig, g.Nx, 2) : outside_right(ig, g.Nx, 2))) zi = 0.5 * (zeta(0, 0) + zeta(1, 0));
        else zi = pos ? recon5<ADV_WENO_Z>(zeta(-2, 0), zeta(-1, 0), zeta(0, 0), zeta(1, 0), zeta(2, 0), true)
                      : recon5<ADV_WENO_Z>(zeta(3, 0), zeta(2, 0), zeta(1, 0), zeta(0, 0), zeta(-1, 0), false);
        vvV = +(uhat * zi);
      }
    }
    const double z2w_hi = z2w(1), z1w_hi = z1w(1);
    const double vaU = hy_div(0.5 * (z2w_lo + z2w_hi), g.azcc[r], g.r_azcc[r]);          // Az^fcc = Az^cc (regular x)
    const double vaV = hy_div(0.5 * (z1w_lo + z1w_hi), g.azff[r], g.r_azff[r]);          // Az^cfc = Az^ff
    z2w_lo = z2w_hi;
    z1w_lo = z1w_hi;
    const double k00 = Kh(0, 0);
    const double bhU = hy_div(k00 - Kh(-1, 0), dxfc, rdxfc), bhV = hy_div(k00 - Kh(0, -1), dycf, rdycf);
    Au = (vvU + vaU) + bhU;
    Av = (vvV + vaV) + bhV;
  }
  double Cu = 0.0, Cv = 0.0;
  if (ph.cor == 3) {
    Cu = -ph.f0 * (0.5 * (0.5 * (V(-1, 0, 0) + V(0, 0, 0)) + 0.5 * (V(-1, 1, 0) + V(0, 1, 0))));
    Cv = ph.f0 * (0.5 * (0.5 * (U(0, -1, 0) + U(1, -1, 0)) + 0.5 * (U(0, 0, 0) + U(1, 0, 0))));
  } else if (ph.cor == 1) {
    const double f0 = ph.frow[r], f1 = ph.frow[r + 1];
    Cu = hy_div(-(0.5 * (f0 + f1)) * (0.5 * (Iy_dxv(-1) + Iy_dxv(0))), dxfc, rdxfc);
    Cv = hy_div(+(0.5 * (f0 + f0)) * (0.5 * (Ix_dyu(-1) + Ix_dyu(0))), dycf, rdycf);
  } else if (ph.cor == 2) {
    const double f0 = ph.frow[r], f1 = ph.frow[r + 1];
    Cu = hy_div(-(0.5 * (f0 * Ix_dxv(0) + f1 * Ix_dxv(1))), dxfc, rdxfc);
    Cv = hy_div(+(0.5 * (f0 * Iy_dyu(0) + f0 * Iy_dyu(1))), dycf, rdycf);
  }
  const double px = hy_div(p[cc] - p[cc - 1], dxfc, rdxfc), py = hy_div(p[cc] - p[cc - syc], dycf, rdycf);
  Gu[cu] = ((-Au - 0.0) - Cu) - px;
  Gv[cv] = ((-Av - 0.0) - Cv) - py;
  }
}

// NT tracers in one launch, one thread per column marching upwards: the area-weighted velocities of the faces are formed once and
// shared by the tracers, a level's own values and its top flux stay in registers for the next level (the top flux of level k IS the
// bottom flux of level k + 1: same operands, same bits), so every 3-D value is fetched once along z -- the one-thread-per-cell form
// of this kernel pulled 2.1 times its algorithmic bytes through L2 (profiles/r03_pmc_config5.json), the levels above and below a
// level being fetched again by other workgroups long after.
template <int NT>
__global__ void __launch_bounds__(256) k_hy_Gc(HyMetric g, const double* __restrict__ u, const double* __restrict__ v,
                                               const double* __restrict__ w, const double* __restrict__ c0, const double* __restrict__ c1,
                                               double* __restrict__ G0, double* __restrict__ G1, int tadv, long syu, long szu, long syv, long szv,
                                               long syc, long szc) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int r = OCN_UNIFORM(j + g.Hy);       // blockDim.x == 64: one row per wave
  long cu = (i + g.Hx) + (long)r * syu + (long)g.Hz * szu, cv = (i + g.Hx) + (long)r * syv + (long)g.Hz * szv;
  long cc = (i + g.Hx) + (long)r * syc + (long)g.Hz * szc;
  if (!tadv) {
    for (int k = 0; k < g.Nz; ++k, cc += szc) {
      G0[cc] = 0.0;
      if (NT > 1) G1[cc] = 0.0;
    }
    return;
  }
  const double dyfc = g.dyfc[r], dxcf0 = g.dxcf[r], dxcf1 = g.dxcf[r + 1], azcc = g.azcc[r];
  double cm[NT], cc_[NT], fz0[NT];
  double az0 = azcc * w[cc];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const double* c = t ? c1 : c0;
    cm[t] = c[cc - szc];
    cc_[t] = c[cc];
    fz0[t] = az0 * (0.5 * (cm[t] + cc_[t]));
  }
  for (int k = 0; k < g.Nz; ++k, cu += szu, cv += szv, cc += szc) {
    const double dz = g.dzc[k];
    const double ax0 = (dyfc * dz) * u[cu], ax1 = (dyfc * dz) * u[cu + 1];                    // Ax_q^fcc u at faces i, i + 1
    const double ay0 = (dxcf0 * dz) * v[cv], ay1 = (dxcf1 * dz) * v[cv + syv];                // Ay_q^cfc v at faces j, j + 1
    const double az1 = azcc * w[cc + szc];                                                    // Az_q^ccf w at face k + 1
    const double rv = 1 / (azcc * dz);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double* c = t ? c1 : c0;
      const double cen = cc_[t], cup = c[cc + szc];
      const double fx0 = ax0 * (0.5 * (c[cc - 1] + cen)), fx1 = ax1 * (0.5 * (cen + c[cc + 1]));
      const double fy0 = ay0 * (0.5 * (c[cc - syc] + cen)), fy1 = ay1 * (0.5 * (cen + c[cc + syc]));
      const double fz1 = az1 * (0.5 * (cen + cup));
      const double div = rv * (((fx1 - fx0) + (fy1 - fy0)) + (fz1 - fz0[t]));
      (t ? G1 : G0)[cc] = -div;
      fz0[t] = fz1;
      cc_[t] = cup;
    }
  }
}

// higher-order flux-form tracer advection on these grids (CenteredFourthOrder, UpwindBiasedFifthOrder, WENO5 with Z weights): the
// reconstructions of stencils.h (the Nonhydrostatic kernels' own, fast reciprocal and contraction included: parity with the oracle to
// round-off, not to the bit) times the area-weighted face velocities; upwind_biased_product(Ax u, c^L, c^R) = (Ax u) c^upwind;
// inside the boundary buffer of a Bounded direction the second-order fallback (topologically_conditional_interpolation.jl:19-83).
// On a latitude band the buffer test uses the GLOBAL row (jrow0 + j) and row count.  One thread per column, the vertical flux reused
// from level to level; the x and y face fluxes are still formed twice (by the two cells they separate) -- not a tiled kernel.
template <int ADV, int NT>
__global__ void __launch_bounds__(256) k_hy_Gc_hi(HyMetric g, const double* __restrict__ u, const double* __restrict__ v,
                                                  const double* __restrict__ w, const double* __restrict__ c0, const double* __restrict__ c1,
                                                  double* __restrict__ G0, double* __restrict__ G1, int xb, int yb, int jrow0, int gNy, long syu,
                                                  long szu, long syv, long szv, long syc, long szc) {
  // one thread per column marching upwards: the flux through a level's upper face is kept for the next level
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  constexpr int NB = ADV == ADV_C4 ? 1 : 2;                       // boundary_buffer of the scheme
  const int r = OCN_UNIFORM(j + g.Hy);       // blockDim.x == 64: one row per wave
  long cu = (i + g.Hx) + (long)r * syu + (long)g.Hz * szu, cv = (i + g.Hx) + (long)r * syv + (long)g.Hz * szv;
  long cc = (i + g.Hx) + (long)r * syc + (long)g.Hz * szc;
  const double dyfc = g.dyfc[r], dxcf0 = g.dxcf[r], dxcf1 = g.dxcf[r + 1], azcc = g.azcc[r];
  const int jg = jrow0 + j;
  double fz0[NT];
  {
    const double az0 = azcc * w[cc];
#pragma unroll
    for (int t = 0; t < NT; ++t) fz0[t] = adv_flux_b<ADV>((t ? c1 : c0) + cc, szc, az0, true, 1, g.Nz, NB);
  }
  for (int k = 0; k < g.Nz; ++k, cu += szu, cv += szv, cc += szc) {
    const double dz = g.dzc[k];
    const double ax0 = (dyfc * dz) * u[cu], ax1 = (dyfc * dz) * u[cu + 1];
    const double ay0 = (dxcf0 * dz) * v[cv], ay1 = (dxcf1 * dz) * v[cv + syv];
    const double az1 = azcc * w[cc + szc];
    const double rv = 1 / (azcc * dz);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double* c = (t ? c1 : c0) + cc;
      const double fx0 = adv_flux_b<ADV>(c, 1, ax0, xb != 0, i + 1, g.Nx, NB), fx1 = adv_flux_b<ADV>(c + 1, 1, ax1, xb != 0, i + 2, g.Nx, NB);
      const double fy0 = adv_flux_b<ADV>(c, syc, ay0, yb != 0, jg + 1, gNy, NB), fy1 = adv_flux_b<ADV>(c + syc, syc, ay1, yb != 0, jg + 2, gNy, NB);
      const double fz1 = adv_flux_b<ADV>(c + szc, szc, az1, true, k + 2, g.Nz, NB);
      (t ? G1 : G0)[cc] = -(rv * (((fx1 - fx0) + (fy1 - fy0)) + (fz1 - fz0[t])));
      fz0[t] = fz1;
    }
  }
}

// implicit_step! for VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa) with constant coefficients
// (vertically_implicit_diffusion_solver.jl:46-100, Solvers/batched_tridiagonal_solver.jl:89-121): the tridiagonal coefficients depend on
// the level only, so the pivots beta_k and the multipliers t_k of the modified Thomas algorithm are tabulated once per (kappa, dt) on
// the host -- the very numbers every column of the reference's solver computes -- and a thread only substitutes: up the column
// phi_k = (f_k - a_{k-1} phi_{k-1}) / beta_k (the division through beta's correctly rounded reciprocal), down phi_k -= t_{k+1} phi_{k+1}.
__global__ void k_hy_implicit(double* f, HyImp c, int Nx, int Ny, int Nz, int Hx, int Hy, int Hz, long sy, long sz) {
  OCN_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= Nx || j >= Ny) return;
  double* p = f + (i + Hx) + (long)(j + Hy) * sy + (long)Hz * sz;
  double phi = hy_div(p[0], c.beta[0], c.rbeta[0]);
  p[0] = phi;
  for (int k = 1; k < Nz; ++k) {
    phi = hy_div(p[k * sz] - c.a[k - 1] * phi, c.beta[k], c.rbeta[k]);
    p[k * sz] = phi;
  }
  for (int k = Nz - 2; k >= 0; --k) {
    phi = p[k * sz] - c.t[k + 1] * phi;
    p[k * sz] = phi;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int api_done(ocn_ctx* ctx, int rc) {
  if (ctx->sticky_rc) {
    if (rc == OCN_OK) rc = ctx->sticky_rc;
    ctx->sticky_rc = 0;
  }
#ifndef OCN_HOST_EMU
  if (g_ocn_launch_err.err != hipSuccess) {
    if (rc == OCN_OK) {
      ocn_set_error(ctx, "launch failed: %s [%s]", hipGetErrorString(g_ocn_launch_err.err), g_ocn_launch_err.what);
      rc = OCN_EHIP;
    }
    g_ocn_launch_err.err = hipSuccess;
  }
#endif
  return rc;
}

static SeArgs se_args(const ocn_sefs* s, double dtau, int index) {
  const ocn_hgrid* g = s->g;
  SeArgs a;
  a.eta = s->eta->d; a.U = s->U->d; a.V = s->V->d; a.etabar = s->etabar->d; a.Ubar = s->Ubar->d; a.Vbar = s->Vbar->d;
  a.GU = s->GU->d; a.GV = s->GV->d; a.Hfc = s->Hfc->d; a.Hcf = s->Hcf->d;
  a.dxfc = g->dxfc; a.dycf = g->dycf; a.dyfc = g->dyfc; a.dxcf = g->dxcf; a.azcc = g->azcc;
  a.r_dxfc = g->r_dxfc; a.r_dycf = g->r_dycf;
  a.Nx = g->N[0]; a.Ny = g->N[1]; a.Hx = g->H[0]; a.Hy = g->H[1];
  a.se = s->eta->T[0]; a.su = s->U->T[0]; a.sv = s->V->T[0];
  a.g = s->grav; a.dtau = dtau;
  a.wv = s->wv[index - 1]; a.wf = s->wf[index - 1];
  a.xper = g->topo[0] == OCN_PERIODIC; a.yper = g->topo[1] == OCN_PERIODIC;
  return a;
}

static void se_shape(const ocn_hgrid* g, dim3& b, dim3& gr) {
  b = dim3(64, 4, 1);
  gr = dim3((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1);
}

// split_explicit_free_surface_substep! as the reference issues it
static void sefs_substep_plain(ocn_sefs* s, double dtau, int index) {
  dim3 b, gr;
  se_shape(s->g, b, gr);
  hipStream_t st = s->g->ctx->stream;
  const SeArgs a = se_args(s, dtau, index);
  hfield_fill(s->eta);
  ocn_launch(k_se_uv, gr, b, st, a);
  hfield_fill(s->U);
  hfield_fill(s->V);
  ocn_launch(k_se_eta, gr, b, st, a);
}

static bool sefs_fusable(const ocn_sefs* s) {
  const ocn_hgrid* g = s->g;
  if (g->topo[0] != OCN_PERIODIC || g->N[0] < g->H[0] || g->H[0] < 1 || g->H[1] < 1) return false;
  if (g->topo[1] == OCN_PERIODIC && g->N[1] < g->H[1]) return false;
  return true;
}

static void sefs_substep_fused(ocn_sefs* s, double dtau, int index) {
  dim3 b, gr;
  se_shape(s->g, b, gr);
  hipStream_t st = s->g->ctx->stream;
  const SeArgs a = se_args(s, dtau, index);
  ocn_launch(k_se_uv_fused, gr, b, st, a);
  ocn_launch(k_se_eta, gr, b, st, a);
}

// substep `index` in one launch; flip: 0 reads the fields' own arrays and writes the second set, 1 the other way round
static void sefs_substep_one(ocn_sefs* s, double dtau, int index, int flip) {
  dim3 b, gr;
  se_shape(s->g, b, gr);
  SeArgs1 q;
  q.a = se_args(s, dtau, index);
  q.etaI = flip ? s->eta2 : s->eta->d; q.UI = flip ? s->U2 : s->U->d; q.VI = flip ? s->V2 : s->V->d;
  q.etaO = flip ? s->eta->d : s->eta2; q.UO = flip ? s->U->d : s->U2; q.VO = flip ? s->V->d : s->V2;
  ocn_launch(k_se_substep1, gr, b, s->g->ctx->stream, q);
}
// the most substeps one launch takes on this grid: 8 (tiles of 48 x 16, two rows per thread) from 64 x 32 cells up, else 4 (56 x 8)
static int sefs_multi_width(const ocn_sefs* s) {
  static const int forced = getenv("OCNHIP_SE_MULTI") ? atoi(getenv("OCNHIP_SE_MULTI")) : 0;     // 4 or 8: measurement knob
  if (forced == 4 || s->g->N[1] < 32) return 4;
  return 8;
}
// substeps index .. index + n - 1 (n <= sefs_multi_width) in one launch, interior cells only
static void sefs_substep_multi(ocn_sefs* s, double dtau, int index, int n, int flip) {
  const ocn_hgrid* g = s->g;
  SeArgsM m;
  m.b.a = se_args(s, dtau, index);
  m.b.etaI = flip ? s->eta2 : s->eta->d; m.b.UI = flip ? s->U2 : s->U->d; m.b.VI = flip ? s->V2 : s->V->d;
  m.b.etaO = flip ? s->eta->d : s->eta2; m.b.UO = flip ? s->U->d : s->U2; m.b.VO = flip ? s->V->d : s->V2;
  m.nsub = n;
  for (int q = 0; q < SE_MS; ++q) {
    m.wv[q] = q < n ? s->wv[index - 1 + q] : 0.0;
    m.wf[q] = q < n ? s->wf[index - 1 + q] : 0.0;
  }
  if (sefs_multi_width(s) == 8) {
    constexpr int TX = 64 - 16, TY = 32 - 16;
    ocn_launch_sync(k_se_multi<8, 2>, dim3((g->N[0] + TX - 1) / TX, (g->N[1] + TY - 1) / TY, 1), dim3(64, 16, 1), g->ctx->stream, m);
  } else {
    constexpr int TX = 64 - 8, TY = 16 - 8;
    ocn_launch_sync(k_se_multi<4, 1>, dim3((g->N[0] + TX - 1) / TX, (g->N[1] + TY - 1) / TY, 1), dim3(64, 16, 1), g->ctx->stream, m);
  }
}
static bool sefs_multi_ok(const ocn_sefs* s) { return sefs_fusable(s) && s->g->N[0] >= 64 && s->g->N[1] >= 16; }
static void se_copy(ocn_ctx* ctx, double* dst, const double* src, size_t n) {
  ocn_launch(k_se_copy, dim3((unsigned)((n + 255) / 256), 1, 1), dim3(256, 1, 1), ctx->stream, dst, src, n);
}

static int hfield_new(ocn_hgrid* g, int lx, int ly, int lz, ocn_hfield** out) {
  ocn_hfield* f = new ocn_hfield;
  f->g = g;
  const int loc[3] = {lx, ly, lz};
  f->n = 1;
  for (int d = 0; d < 3; ++d) {
    f->loc[d] = loc[d];
    f->T[d] = total_len(loc[d], g->topo[d], g->N[d], g->H[d]);
    f->S[d] = interior_len(loc[d], g->topo[d], g->N[d]);
    f->n *= (size_t)f->T[d];
  }
  if (hipMalloc((void**)&f->d, f->n * sizeof(double)) != hipSuccess) {
    delete f;
    ocn_set_error(g->ctx, "allocation of %zu bytes failed", f->n * sizeof(double));
    return OCN_ENOMEM;
  }
  OCN_ASYNC(hipMemsetAsync(f->d, 0, f->n * sizeof(double), g->ctx->stream));
  g->refs += 1;
  *out = f;
  return OCN_OK;
}

static void vsum(ocn_sefs* s, ocn_hfield* out, const ocn_hfield* a, const ocn_hfield* b, double cn, double cm) {
  const ocn_hgrid* g = s->g;
  dim3 blk(64, 4, 1), gr((out->S[0] + 63) / 64, (out->S[1] + 3) / 4, 1);
  ocn_launch(k_se_vsum, gr, blk, g->ctx->stream, out->d, (const double*)a->d, b ? (const double*)b->d : (const double*)nullptr, cn, cm,
             (const double*)g->dzc, out->S[0], out->S[1], g->N[2], g->H[0], g->H[1], g->H[2], (long)a->T[0], (long)a->T[0] * a->T[1], (long)out->T[0]);
}


// ---- the hydrostatic step (second slice) ----------------------------------------------------------------------------------------
struct ocn_hydro {
  ocn_sefs* fs;
  ocn_hgrid* lg;                             // the grid of the 3-D fields: the free surface's, or a latitude band of it (the free surface is then
                                             // replicated: every rank sub-cycles the whole barotropic problem after one all-gather of U, V, G^U, G^V)
  long offU = 0, offV = 0;                   // first element of the band's rows inside the free surface's (Face, Center) / (Center, Face) arrays
  ocn_hfield *u, *v, *w, *pHY;
  std::vector<ocn_hfield*> c, gn, gm;        // tracers and the tendencies: entries 0, 1 of gn / gm are u, v; 2.. the tracers
  HyBuoy buoy;
  int bT, bS;                                // indices (into c) of the tracers the buoyancy reads, -1: none
  double *Un = nullptr, *Vn = nullptr;       // barotropic mode of the stepped velocities, kept for the corrector
  HyPhys phys{1, 0, 0, 0, 0, 0, 1, 0.0, nullptr};        // the model's defaults: VectorInvariant(), no Coriolis, CenteredSecondOrder tracers
  double* frow = nullptr;
  double chi = 0.1;                          // QuasiAdamsBashforth2TimeStepper's default
  // VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa): entry 0 the viscosity, 1 + q the diffusivity of tracer q
  std::vector<double> kap;
  struct ImpTab { double kappa, dt; double* d = nullptr; };      // device table of 4 Nz doubles: a, beta, 1 / beta, t
  std::vector<ImpTab> imptab;
};

static HyGrid hy_grid(const ocn_hgrid* g) {
  HyGrid q;
  q.dxcf = g->dxcf; q.dyfc = g->dyfc; q.azcc = g->azcc; q.dzc = g->dzc; q.dzf = g->dzf;
  q.Nx = g->N[0]; q.Ny = g->N[1]; q.Nz = g->N[2]; q.Hx = g->H[0]; q.Hy = g->H[1]; q.Hz = g->H[2];
  return q;
}
static void hy_cols(const ocn_hgrid* g, dim3& b, dim3& gr) {
  b = dim3(64, 4, 1);
  gr = dim3((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1);
}
static void hy_ab2_launch(ocn_hfield* f, const ocn_hfield* gn, ocn_hfield* gm, double dt, double chi, bool store) {
  const ocn_hgrid* g = f->g;
  dim3 b(64, 4, 1), gr((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, g->N[2]);
  if (store)
    ocn_launch(k_hy_ab2_store, gr, b, g->ctx->stream, f->d, (const double*)gn->d, gm->d, dt, 1.5 + chi, 0.5 + chi, g->N[0], g->N[1], g->N[2], g->H[0],
               g->H[1], g->H[2], (long)f->T[0], (long)f->T[0] * f->T[1]);
  else
    ocn_launch(k_hy_ab2, gr, b, g->ctx->stream, f->d, (const double*)gn->d, (const double*)gm->d, dt, 1.5 + chi, 0.5 + chi, g->N[0], g->N[1], g->N[2],
               g->H[0], g->H[1], g->H[2], (long)f->T[0], (long)f->T[0] * f->T[1]);
}
static void hy_store_launch(ocn_hfield* gm, const ocn_hfield* gn) {
  const ocn_hgrid* g = gm->g;
  dim3 b(64, 4, 1), gr((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, g->N[2]);
  ocn_launch(k_hy_store, gr, b, g->ctx->stream, gm->d, (const double*)gn->d, g->N[0], g->N[1], g->N[2], g->H[0], g->H[1], g->H[2], (long)gm->T[0],
             (long)gm->T[0] * gm->T[1]);
}
static void hy_w_launch(const ocn_hfield* u, const ocn_hfield* v, ocn_hfield* w) {
  const ocn_hgrid* g = w->g;
  dim3 b, gr;
  hy_cols(g, b, gr);
  ocn_launch(k_hy_w, gr, b, g->ctx->stream, hy_grid(g), (const double*)u->d, (const double*)v->d, w->d, (long)u->T[0], (long)u->T[0] * u->T[1],
             (long)v->T[0], (long)v->T[0] * v->T[1], (long)w->T[0], (long)w->T[0] * w->T[1]);
}
static void hy_pressure_launch(ocn_hfield* p, const HyBuoy& q, const ocn_hfield* T, const ocn_hfield* S) {
  const ocn_hgrid* g = p->g;
  dim3 b, gr;
  hy_cols(g, b, gr);
  ocn_launch(k_hy_pressure, gr, b, g->ctx->stream, hy_grid(g), q, T ? (const double*)T->d : (const double*)nullptr,
             S ? (const double*)S->d : (const double*)nullptr, p->d, (long)p->T[0], (long)p->T[0] * p->T[1]);
}
static bool is_loc(const ocn_hfield* f, const ocn_hgrid* g, int lx, int ly, int lz) {
  return f && f->g == g && f->loc[0] == lx && f->loc[1] == ly && f->loc[2] == lz;
}

// banded free surface: the overlap rows of the listed 2-D fields from the neighbouring bands' own rows -- whole parent rows are
// contiguous, so the blocks travel straight out of and into the arrays; one grouped exchange (collective over the context's ranks)
static int band_refresh(ocn_sefs* s, std::initializer_list<ocn_hfield*> fields) {
  ocn_hgrid* g = s->g;
  ocn_ctx* c = g->ctx;
  const int W = g->overlap, Hy = g->H[1];
  if (W <= 0) return OCN_OK;
  std::vector<CommOp> sends, recvs;
  int tag = 20;
  for (ocn_hfield* f : fields) {
    const size_t T0 = f->T[0];
    const int extra = f->loc[1] == OCN_FACE ? 1 : 0;
    auto at = [&](int row) { return f->d + (size_t)(Hy + row) * T0; };
    if (g->ext_hi > 0) {    // upper neighbour: my top W own rows up, its bottom W (+1) own rows down into my upper overlap
      sends.push_back({at(g->ext_lo + g->own - W), (size_t)W * T0 * sizeof(double), c->rank + 1, tag});
      recvs.push_back({at(g->ext_lo + g->own), (size_t)(W + extra) * T0 * sizeof(double), c->rank + 1, tag + 1});
    }
    if (g->ext_lo > 0) {    // lower neighbour
      sends.push_back({at(g->ext_lo), (size_t)(W + extra) * T0 * sizeof(double), c->rank - 1, tag + 1});
      recvs.push_back({at(0), (size_t)W * T0 * sizeof(double), c->rank - 1, tag});
    }
    tag += 2;
  }
  return comm_exchange(c, sends, recvs);
}
// train form of the time step: four substeps per launch; the host emulation (one OS thread per GPU thread, 1024 per workgroup here)
// keeps the one-launch form for speed -- the four-substep kernel is compared with the launch-by-launch substeps in its own test
#ifndef OCN_HOST_EMU
#define SE_STEP_MODE 3
#else
#define SE_STEP_MODE 2
#endif
// everything of ocn_sefs_step after the vertical integrals of the tendencies
static int sefs_step_tail(ocn_sefs* s, double dt) {
  ocn_ctx* ctx = s->g->ctx;
  const double dtau = 2 * dt / s->substeps;                        // "we evolve for two times the dt" (:137)
  if (s->g->overlap > 0) {
    // banded: the overlap rows of the state and the forcing from the neighbours, then blocks of W substeps on the extended band (the
    // artificial walls at its ends spoil one row per substep, from the outside in: after W substeps the band's own rows are still
    // those of the whole-domain run) with a refresh of eta, U, V between blocks
    const int W = s->g->overlap;
    if (int rc = band_refresh(s, {s->eta, s->U, s->V, s->GU, s->GV})) return rc;
    for (ocn_hfield* f : {s->U, s->V, s->GU, s->GV}) hfield_fill(f);
    for (int first = 1; first <= s->substeps; first += W) {
      const int count = s->substeps - first + 1 < W ? s->substeps - first + 1 : W;
      if (int rc = ocn_sefs_substeps(s, dtau, first, count, SE_STEP_MODE)) return rc;
      if (first + count <= s->substeps)
        if (int rc = band_refresh(s, {s->eta, s->U, s->V})) return rc;
    }
    se_copy(ctx, s->eta->d, s->etabar->d, s->eta->n);
    hfield_fill(s->eta);
    return OCN_OK;
  }
  hfield_fill(s->GU);
  hfield_fill(s->GV);
  if (int rc = ocn_sefs_substeps(s, dtau, 1, s->substeps, SE_STEP_MODE)) return rc;
  // set!(eta, etabar) copies the parent array (Fields/set!.jl:41-44); then fill_halo_regions!(eta)
  se_copy(ctx, s->eta->d, s->etabar->d, s->eta->n);
  hfield_fill(s->eta);
  return OCN_OK;
}
// offU / offV: first element of the rows of u's grid inside the free surface's 2-D arrays (a latitude band; 0 otherwise)
static void sefs_correct_launch(ocn_sefs* s, ocn_hfield* u, ocn_hfield* v, long offU = 0, long offV = 0) {
  const ocn_hgrid* g = u->g;
  dim3 blk(64, 4, 1), gr((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1);
  ocn_launch(k_se_correct, gr, blk, g->ctx->stream, u->d, v->d, (const double*)s->U->d + offU, (const double*)s->V->d + offV,
             (const double*)s->Ubar->d + offU, (const double*)s->Vbar->d + offV, (const double*)s->Hfc->d + offU, (const double*)s->Hcf->d + offV, g->N[0],
             g->N[1], g->N[2], g->H[0], g->H[1], g->H[2], (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1],
             (long)s->U->T[0], (long)s->V->T[0]);
}

// every band's rows of the listed (Face, Center) / (Center, Face) arrays of the replicated free surface to every other rank: rows are
// contiguous (whole parent rows), so the blocks travel straight out of and into the arrays -- one grouped exchange
static int hydro_allgather_rows(ocn_hydro* h) {
  ocn_sefs* s = h->fs;
  ocn_hgrid* lg = h->lg;
  ocn_ctx* c = lg->ctx;
  if (!lg->slab || s->g->overlap > 0) return OCN_OK;
  const int R = c->nranks, nl = lg->N[1], Hy = lg->H[1];
  const bool per = lg->topo[1] == OCN_PERIODIC;
  ocn_hfield* fl[4] = {s->U, s->GU, s->V, s->GV};
  std::vector<CommOp> sends, recvs;
  for (int p = 0; p < R; ++p) {
    if (p == c->rank) continue;
    for (int q = 0; q < 4; ++q) {
      ocn_hfield* f = fl[q];
      const bool facey = q >= 2 && !per;
      const size_t T0 = f->T[0];
      auto rows = [&](int rank) { return (size_t)(nl + ((facey && rank == R - 1) ? 1 : 0)); };   // the wall row belongs to the last band
      sends.push_back({f->d + (size_t)(Hy + nl * c->rank) * T0, rows(c->rank) * T0 * sizeof(double), p, 10 + q});
      recvs.push_back({f->d + (size_t)(Hy + nl * p) * T0, rows(p) * T0 * sizeof(double), p, 10 + q});
    }
  }
  return comm_exchange(c, sends, recvs);
}
// update_state!: fills of the prognostic fields, w from continuity, the hydrostatic pressure, fills of w and pHY'
static void hydro_update_state(ocn_hydro* h, bool pressure_done) {
  hfield_fill(h->u);
  hfield_fill(h->v);
  hfield_fill(h->fs->eta);
  for (ocn_hfield* c : h->c) hfield_fill(c);
  hy_w_launch(h->u, h->v, h->w);
  if (!pressure_done) hy_pressure_launch(h->pHY, h->buoy, h->bT >= 0 ? h->c[h->bT] : nullptr, h->bS >= 0 ? h->c[h->bS] : nullptr);
  hfield_fill(h->w);
  hfield_fill(h->pHY);
}

static HyMetric hy_metric(const ocn_hgrid* g) {
  HyMetric q;
  q.dxfc = g->dxfc; q.dxcf = g->dxcf; q.dyfc = g->dyfc; q.dycf = g->dycf; q.azcc = g->azcc; q.azff = g->azff; q.dzc = g->dzc; q.dzf = g->dzf;
  q.r_dxfc = g->r_dxfc; q.r_dycf = g->r_dycf; q.r_azcc = g->r_azcc; q.r_azff = g->r_azff; q.r_dzf = g->r_dzf;
  q.Nx = g->N[0]; q.Ny = g->N[1]; q.Nz = g->N[2]; q.Hx = g->H[0]; q.Hy = g->H[1]; q.Hz = g->H[2];
  return q;
}
static void hydro_tendencies(ocn_hydro* h) {
  const ocn_hgrid* g = h->lg;
  dim3 b(64, 4, 1), gr((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, g->N[2]);
  const ocn_hfield *u = h->u, *v = h->v, *p = h->pHY;
  HyPhys ph = h->phys;
  ph.frow = h->frow;
  ph.xb = g->topo[0] != OCN_PERIODIC;
  ph.yb = g->topo[1] != OCN_PERIODIC;
  ph.jrow0 = g->j0;
  ph.gNy = g->gNy;
  ocn_launch(k_hy_Guv, dim3((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1), b, g->ctx->stream, hy_metric(g), ph, (const double*)u->d, (const double*)v->d, (const double*)h->w->d, (const double*)p->d,
             h->gn[0]->d, h->gn[1]->d, (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1], (long)p->T[0],
             (long)p->T[0] * p->T[1]);
  for (size_t q = 0; q < h->c.size(); q += 2) {
    const bool two = q + 1 < h->c.size();
    const double *c0 = h->c[q]->d, *c1 = two ? h->c[q + 1]->d : nullptr;
    double *G0 = h->gn[2 + q]->d, *G1 = two ? h->gn[3 + q]->d : nullptr;
    const int tadv = h->phys.tadv;
    if (tadv >= 2) {
      // CenteredFourthOrder / UpwindBiasedFifthOrder / WENO5
      const int xb = g->topo[0] != OCN_PERIODIC, yb = g->topo[1] != OCN_PERIODIC;
      const dim3 grh((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1);          // one thread per column
#define HY_GC_HI(ADVV)                                                                                                                     \
  if (two)                                                                                                                                 \
    ocn_launch(k_hy_Gc_hi<ADVV, 2>, grh, b, g->ctx->stream, hy_metric(g), (const double*)u->d, (const double*)v->d, (const double*)h->w->d,  \
               c0, c1, G0, G1, xb, yb, g->j0, g->gNy, (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1],        \
               (long)p->T[0], (long)p->T[0] * p->T[1]);                                                                                    \
  else                                                                                                                                     \
    ocn_launch(k_hy_Gc_hi<ADVV, 1>, grh, b, g->ctx->stream, hy_metric(g), (const double*)u->d, (const double*)v->d, (const double*)h->w->d,  \
               c0, c1, G0, G1, xb, yb, g->j0, g->gNy, (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1],        \
               (long)p->T[0], (long)p->T[0] * p->T[1]);
      if (tadv == 2) { HY_GC_HI(ADV_C4) } else if (tadv == 3) { HY_GC_HI(ADV_U5) } else { HY_GC_HI(ADV_WENO_Z) }
#undef HY_GC_HI
      continue;
    }
    const dim3 grc((g->N[0] + 63) / 64, (g->N[1] + 3) / 4, 1);          // one thread per column
    if (two)
      ocn_launch(k_hy_Gc<2>, grc, b, g->ctx->stream, hy_metric(g), (const double*)u->d, (const double*)v->d, (const double*)h->w->d, c0, c1, G0, G1,
                 h->phys.tadv, (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1], (long)p->T[0], (long)p->T[0] * p->T[1]);
    else
      ocn_launch(k_hy_Gc<1>, grc, b, g->ctx->stream, hy_metric(g), (const double*)u->d, (const double*)v->d, (const double*)h->w->d, c0, c1, G0, G1,
                 h->phys.tadv, (long)u->T[0], (long)u->T[0] * u->T[1], (long)v->T[0], (long)v->T[0] * v->T[1], (long)p->T[0], (long)p->T[0] * p->T[1]);
  }
}

// the per-level tables of the implicit solve for (kappa, dt); cached on the handle
static int hydro_imp_table(ocn_hydro* h, double kappa, double dt, HyImp* out) {
  const ocn_hgrid* g = h->lg;
  const int Nz = g->N[2];
  for (auto& e : h->imptab)
    if (e.kappa == kappa && e.dt == dt) {
      *out = HyImp{e.d, e.d + Nz, e.d + 2 * Nz, e.d + 3 * Nz};
      return OCN_OK;
    }
  auto dzc = [&](int k) { return g->h_dzc[k - 1]; };          // 1-based level
  auto dzf = [&](int k) { return g->h_dzf[k - 1]; };          // 1-based face
  auto upper = [&](int k) { return k > Nz - 1 ? 0.0 : -dt * (kappa / dzc(k) / dzf(k + 1)); };
  auto lower = [&](int k) { return k < 1 ? 0.0 : -dt * (kappa / dzc(k + 1) / dzf(k + 1)); };
  auto diag = [&](int k) { return (1.0 - dt * 0.0 - upper(k)) - lower(k - 1); };
  std::vector<double> tab(4 * (size_t)Nz, 0.0);
  double* a = tab.data();
  double *beta = a + Nz, *rbeta = a + 2 * Nz, *t = a + 3 * Nz;
  double b = diag(1);
  beta[0] = b;
  for (int k = 2; k <= Nz; ++k) {
    t[k - 1] = upper(k - 1) / b;
    b = diag(k) - lower(k - 1) * t[k - 1];
    if (!(fabs(b) > 10 * 2.220446049250313e-16)) {
      ocn_set_error(g->ctx, "implicit vertical diffusion: the tridiagonal system is not diagonally dominant at level %d", k);
      return OCN_EINVAL;
    }
    beta[k - 1] = b;
    a[k - 2] = lower(k - 1);
  }
  for (int k = 0; k < Nz; ++k) rbeta[k] = 1.0 / beta[k];
  if (h->imptab.size() >= 16) {
    hipStreamSynchronize(g->ctx->stream);
    for (auto& e : h->imptab) hipFree(e.d);
    h->imptab.clear();
  }
  ocn_hydro::ImpTab e;
  e.kappa = kappa;
  e.dt = dt;
  if (int rc = upload(g->ctx, tab, &e.d)) return rc;
  h->imptab.push_back(e);
  *out = HyImp{e.d, e.d + Nz, e.d + 2 * Nz, e.d + 3 * Nz};
  return OCN_OK;
}
// implicit_step!(field, ...) for entry q of h->kap (0: u and v, 1 + n: tracer n); nothing when that diffusivity is zero
static int hydro_implicit(ocn_hydro* h, ocn_hfield* f, int q, double dt) {
  if ((size_t)q >= h->kap.size() || h->kap[q] == 0.0) return OCN_OK;
  HyImp c;
  if (int rc = hydro_imp_table(h, h->kap[q], dt, &c)) return rc;
  const ocn_hgrid* g = h->lg;
  dim3 b, gr;
  hy_cols(g, b, gr);
  ocn_launch(k_hy_implicit, gr, b, g->ctx->stream, f->d, c, g->N[0], g->N[1], g->N[2], g->H[0], g->H[1], g->H[2], (long)f->T[0], (long)f->T[0] * f->T[1]);
  return OCN_OK;
}
static bool hydro_has_implicit(const ocn_hydro* h) {
  for (double k : h->kap)
    if (k != 0.0) return true;
  return false;
}

static bool same_shape(const ocn_hfield* a, const ocn_hfield* b) {
  return a && b && a->g == b->g && a->T[0] == b->T[0] && a->T[1] == b->T[1] && a->T[2] == b->T[2] && a->loc[0] == b->loc[0] &&
         a->loc[1] == b->loc[1] && a->loc[2] == b->loc[2];
}

extern "C" {

int ocn_hgrid_create(ocn_ctx* ctx, const ocn_hgrid_desc* d, ocn_hgrid** out) {
  if (!ctx || !d || !out) return OCN_EINVAL;
  if (d->kind != HG_RECT && d->kind != HG_LATLON) return OCN_EINVAL;
  for (int q = 0; q < 3; ++q)
    if (d->N[q] < 1 || d->H[q] < 0 || (q < 2 && !(d->L[q] > 0))) {
      ocn_set_error(ctx, "ocn_hgrid_create: invalid size / halo / extent in direction %d", q);
      return OCN_EINVAL;
    }
  if ((d->topology[0] != OCN_PERIODIC && d->topology[0] != OCN_BOUNDED) || (d->topology[1] != OCN_PERIODIC && d->topology[1] != OCN_BOUNDED) ||
      d->topology[2] != OCN_BOUNDED) {
    ocn_set_error(ctx, "ocn_hgrid_create: x and y must be Periodic or Bounded and z Bounded (hydrostatic_free_surface_model.jl:118-119)");
    return OCN_EUNSUPPORTED;
  }
  if (d->kind == HG_LATLON && (d->L[0] > 360 || d->x0[1] < -90 || d->x0[1] + d->L[1] > 90 || d->topology[1] != OCN_BOUNDED)) {
    ocn_set_error(ctx, "ocn_hgrid_create: longitude must span at most 360 degrees, latitude must lie in [-90, 90] and be Bounded");
    return OCN_EINVAL;
  }
  if ((d->partition != 0 && d->partition != 1) || d->band_overlap < 0) return OCN_EINVAL;
  const bool slab = d->partition == 1 && ctx->nranks > 1;
  if (slab && d->band_overlap > d->N[1] / ctx->nranks) {
    ocn_set_error(ctx, "ocn_hgrid_create: an overlap of %d rows is wider than a band of %d", d->band_overlap, d->N[1] / ctx->nranks);
    return OCN_EINVAL;
  }
  if (slab && (d->N[1] % ctx->nranks != 0 || d->N[1] / ctx->nranks < d->H[1] + 1)) {
    ocn_set_error(ctx, "ocn_hgrid_create: %d rows do not split into %d bands of more than H = %d rows", d->N[1], ctx->nranks, d->H[1]);
    return OCN_EINVAL;
  }
  ocn_hgrid* g = new ocn_hgrid;
  g->ctx = ctx;
  g->kind = d->kind;
  for (int q = 0; q < 3; ++q) { g->N[q] = d->N[q]; g->H[q] = d->H[q]; g->topo[q] = d->topology[q]; g->x0[q] = d->x0[q]; g->L[q] = d->L[q]; }
  g->radius = d->radius > 0 ? d->radius : 6371.0e3;
  g->gNy = d->N[1];
  double dx, dy, dz = 0;
  regular_axis(g->x0[0], g->L[0], g->N[0], g->H[0], g->topo[0], g->nodeF[0], g->nodeC[0], dx);
  regular_axis(g->x0[1], g->L[1], g->N[1], g->H[1], g->topo[1], g->nodeF[1], g->nodeC[1], dy);
  if (slab) {
    // the band's rows of the GLOBAL node arrays (every metric below is then the global one of the same row, bit for bit)
    int nl = d->N[1] / ctx->nranks, j0 = nl * ctx->rank;
    g->wall_lo = g->topo[1] != OCN_PERIODIC && ctx->rank == 0;
    g->wall_hi = g->topo[1] != OCN_PERIODIC && ctx->rank == ctx->nranks - 1;
    g->own = nl;
    if (d->band_overlap > 0) {
      if (g->topo[1] == OCN_PERIODIC) {
        delete g;
        ocn_set_error(ctx, "ocn_hgrid_create: extended bands need a Bounded y (the global node arrays do not wrap)");
        return OCN_EUNSUPPORTED;
      }
      g->overlap = d->band_overlap;
      g->ext_lo = g->wall_lo ? 0 : d->band_overlap;
      g->ext_hi = g->wall_hi ? 0 : d->band_overlap;
      j0 -= g->ext_lo;
      nl += g->ext_lo + g->ext_hi;
    } else {
      g->slab = true;
    }
    g->j0 = j0;
    g->N[1] = nl;
    auto band = [&](std::vector<double>& v, int want) {
      std::vector<double> w(want);
      for (int q = 0; q < want; ++q) w[q] = (j0 + q >= 0 && j0 + q < (int)v.size()) ? v[j0 + q] : NAN;
      v.swap(w);
    };
    band(g->nodeF[1], nl + 1 + 2 * g->H[1]);        // Bounded shape for every band
    band(g->nodeC[1], nl + 2 * g->H[1]);
  }
  g->z_regular = d->z_faces == nullptr;
  g->h_dzc.assign(g->N[2], 0.0);
  if (g->z_regular) {
    if (!(d->L[2] > 0)) { delete g; return OCN_EINVAL; }
    regular_axis(g->x0[2], g->L[2], g->N[2], g->H[2], OCN_BOUNDED, g->nodeF[2], g->nodeC[2], dz);
    for (int k = 0; k < g->N[2]; ++k) g->h_dzc[k] = dz;
  } else {
    for (int k = 0; k < g->N[2]; ++k) {
      g->h_dzc[k] = d->z_faces[k + 1] - d->z_faces[k];       // grid_generation.jl:53: dz^c[k] = F[k+1] - F[k]
      if (!(g->h_dzc[k] > 0)) { delete g; ocn_set_error(ctx, "z_faces must be strictly increasing"); return OCN_EINVAL; }
    }
    g->nodeF[2].assign(d->z_faces, d->z_faces + g->N[2] + 1);
    g->nodeC[2].resize(g->N[2]);
    for (int k = 0; k < g->N[2]; ++k) g->nodeC[2][k] = (d->z_faces[k + 1] + d->z_faces[k]) / 2;
    g->L[2] = d->z_faces[g->N[2]] - d->z_faces[0];
  }
  // dz^f: regular -> dz; stretched -> differences of the centres, the halo centres built from faces extended by the boundary
  // cells' widths (grid_generation.jl:28-75)
  g->h_dzf.assign(g->N[2] + 1, dz);
  if (!g->z_regular) {
    const int n = g->N[2];
    const double* F = d->z_faces;
    const double cb = ((F[0] - (F[1] - F[0])) + F[0]) / 2, ct = ((F[n] + (F[n] - F[n - 1])) + F[n]) / 2;
    auto C = [&](int k) { return k < 1 ? cb : k > n ? ct : g->nodeC[2][k - 1]; };          // reference index k
    for (int k = 1; k <= n + 1; ++k) g->h_dzf[k - 1] = C(k) - C(k - 1);
  }
  const int ny = g->N[1] + 2 * g->H[1] + 1;
  g->h_dxfc.assign(ny, 0); g->h_dxcf.assign(ny, 0); g->h_dyfc.assign(ny, 0); g->h_dycf.assign(ny, 0); g->h_azcc.assign(ny, 0);
  g->h_azff.assign(ny, 0);
  if (g->kind == HG_RECT) {
    for (int r = 0; r < ny; ++r) { g->h_dxfc[r] = dx; g->h_dxcf[r] = dx; g->h_dyfc[r] = dy; g->h_dycf[r] = dy; g->h_azcc[r] = dx * dy; g->h_azff[r] = dx * dy; }
  } else {
    // latitude_longitude_grid.jl:418-445 (regular longitude and latitude)
    const double R = g->radius, dlam = dx * (M_PI / 180.0), dphi = dy * (M_PI / 180.0);
    auto cosd = [](double p) { return cos(M_PI * p / 180.0); };
    auto sind = [](double p) { return sin(M_PI * p / 180.0); };
    const std::vector<double>&Fy = g->nodeF[1], &Cy = g->nodeC[1];
    for (int r = 0; r < ny; ++r) {
      g->h_dxfc[r] = r < (int)Cy.size() ? R * cosd(Cy[r]) * dlam : NAN;
      g->h_dxcf[r] = r < (int)Fy.size() ? R * cosd(Fy[r]) * dlam : NAN;
      g->h_dyfc[r] = R * dphi;
      g->h_dycf[r] = R * dphi;
      if (r < (int)Fy.size()) g->h_phif.resize(r + 1), g->h_phif[r] = Fy[r];
      g->h_azcc[r] = r + 1 < (int)Fy.size() ? R * R * dlam * (sind(Fy[r + 1]) - sind(Fy[r])) : NAN;
      g->h_azff[r] = (r >= 1 && r < (int)Cy.size()) ? R * R * dlam * (sind(Cy[r]) - sind(Cy[r - 1])) : NAN;   // :444 (regular longitude: = Az^cf)
    }
  }
  int rc = upload(ctx, g->h_dxfc, &g->dxfc);
  if (!rc) rc = upload(ctx, g->h_dxcf, &g->dxcf);
  if (!rc) rc = upload(ctx, g->h_dyfc, &g->dyfc);
  if (!rc) rc = upload(ctx, g->h_dycf, &g->dycf);
  if (!rc) rc = upload(ctx, g->h_azcc, &g->azcc);
  if (!rc) rc = upload(ctx, g->h_dzc, &g->dzc);
  if (!rc) rc = upload(ctx, g->h_dzf, &g->dzf);
  if (!rc) rc = upload(ctx, g->h_azff, &g->azff);
  {
    auto recip = [](const std::vector<double>& v) {
      std::vector<double> r(v.size());
      for (size_t q = 0; q < v.size(); ++q) r[q] = 1.0 / v[q];
      return r;
    };
    if (!rc) rc = upload(ctx, recip(g->h_dxfc), &g->r_dxfc);
    if (!rc) rc = upload(ctx, recip(g->h_dycf), &g->r_dycf);
    if (!rc) rc = upload(ctx, recip(g->h_azcc), &g->r_azcc);
    if (!rc) rc = upload(ctx, recip(g->h_azff), &g->r_azff);
    if (!rc) rc = upload(ctx, recip(g->h_dzf), &g->r_dzf);
  }
  if (rc) { ocn_hgrid_destroy(g); return rc; }
  *out = g;
  return OCN_OK;
}

static void hgrid_release(ocn_hgrid* g);
void ocn_hgrid_destroy(ocn_hgrid* g) { hgrid_release(g); }
static void hgrid_release(ocn_hgrid* g) {
  if (!g || --g->refs > 0) return;
  hipFree(g->pack_s);
  hipFree(g->pack_r);
  hipFree(g->dxfc); hipFree(g->dxcf); hipFree(g->dyfc); hipFree(g->dycf); hipFree(g->azcc); hipFree(g->dzc); hipFree(g->dzf); hipFree(g->azff);
  hipFree(g->r_dxfc); hipFree(g->r_dycf); hipFree(g->r_azcc); hipFree(g->r_azff); hipFree(g->r_dzf);
  delete g;
}

int ocn_hgrid_band(const ocn_hgrid* g, int32_t* j0, int32_t* ny_local, int32_t* ny_global) {
  if (!g) return OCN_EINVAL;
  if (j0) *j0 = g->j0;
  if (ny_local) *ny_local = g->N[1];
  if (ny_global) *ny_global = g->gNy;
  return OCN_OK;
}

/* which: 0 dx^fc, 1 dx^cf, 2 dy^fc, 3 dy^cf, 4 Az^cc (rows j = 1 - Hy ...), 5 dz^c (levels 1..Nz),
 * 6 / 7 x nodes Face / Center, 8 / 9 y nodes Face / Center (incl. halos), 10 dz^f (faces 1..Nz+1), 11 Az^ff (rows); returns the number of entries, copies min(n, entries) */
int ocn_hgrid_metric(const ocn_hgrid* g, int which, double* host, int n) {
  if (!g || !host || n < 0) return OCN_EINVAL;
  const std::vector<double>* v = nullptr;
  switch (which) {
    case 0: v = &g->h_dxfc; break;
    case 1: v = &g->h_dxcf; break;
    case 2: v = &g->h_dyfc; break;
    case 3: v = &g->h_dycf; break;
    case 4: v = &g->h_azcc; break;
    case 5: v = &g->h_dzc; break;
    case 6: v = &g->nodeF[0]; break;
    case 7: v = &g->nodeC[0]; break;
    case 8: v = &g->nodeF[1]; break;
    case 9: v = &g->nodeC[1]; break;
    case 10: v = &g->h_dzf; break;
    case 11: v = &g->h_azff; break;
    default: return OCN_EINVAL;
  }
  const int m = (int)v->size() < n ? (int)v->size() : n;
  for (int q = 0; q < m; ++q) host[q] = (*v)[q];
  return (int)v->size();
}

int ocn_hfield_create(ocn_hgrid* g, int locx, int locy, int locz, ocn_hfield** out) {
  if (!g || !out) return OCN_EINVAL;
  if ((locx != OCN_CENTER && locx != OCN_FACE) || (locy != OCN_CENTER && locy != OCN_FACE) ||
      (locz != OCN_CENTER && locz != OCN_FACE && locz != OCN_NOTHING)) {
    ocn_set_error(g->ctx, "ocn_hfield_create: locations must be Center / Face in x and y, Center, Face or Nothing in z");
    return OCN_EINVAL;
  }
  return hfield_new(g, locx, locy, locz, out);
}

void ocn_hfield_destroy(ocn_hfield* f) {
  if (!f) return;
  hipStreamSynchronize(f->g->ctx->stream);
  if (f->owned) hipFree(f->d);
  ocn_hgrid* g = f->g;
  delete f;
  hgrid_release(g);
}

int ocn_hfield_shape(const ocn_hfield* f, int32_t total[3], int32_t interior[3], int32_t halo[3]) {
  if (!f) return OCN_EINVAL;
  for (int d = 0; d < 3; ++d) {
    if (total) total[d] = f->T[d];
    if (interior) interior[d] = f->S[d];
    if (halo) halo[d] = f->loc[d] == OCN_NOTHING ? 0 : f->g->H[d];
  }
  return OCN_OK;
}

void* ocn_hfield_ptr(ocn_hfield* f) { return f ? (void*)f->d : nullptr; }

int ocn_hfield_upload(ocn_hfield* f, const double* host_parent) {
  if (!f || !host_parent) return OCN_EINVAL;
  OCN_HIP_CHECK(f->g->ctx, hipStreamSynchronize(f->g->ctx->stream));
  OCN_HIP_CHECK(f->g->ctx, hipMemcpy(f->d, host_parent, f->n * sizeof(double), hipMemcpyHostToDevice));
  return OCN_OK;
}

int ocn_hfield_download(const ocn_hfield* f, double* host_parent) {
  if (!f || !host_parent) return OCN_EINVAL;
  OCN_HIP_CHECK(f->g->ctx, hipStreamSynchronize(f->g->ctx->stream));
  OCN_HIP_CHECK(f->g->ctx, hipMemcpy(host_parent, f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
  return api_done(f->g->ctx, OCN_OK);
}

int ocn_hfield_fill_halos(ocn_hfield* f) {
  if (!f) return OCN_EINVAL;
  hfield_fill(f);
  return api_done(f->g->ctx, OCN_OK);
}

int ocn_sefs_create(ocn_hgrid* g, double gravitational_acceleration, int substeps, ocn_sefs** out) {
  if (!g || !out || substeps < 1) return OCN_EINVAL;
  if (g->slab) {
    ocn_set_error(g->ctx, "ocn_sefs_create: the free surface lives on the whole grid (it is replicated on every rank); pass the unpartitioned grid");
    return OCN_EINVAL;
  }
  ocn_sefs* s = new ocn_sefs;
  s->g = g;
  g->refs += 1;
  s->grav = gravitational_acceleration;
  s->substeps = substeps;
  s->wv.assign(substeps, 1.0 / substeps);       // SplitExplicitSettings: ones(substeps) ./ substeps
  s->wf.assign(substeps, 1.0 / substeps);
  struct { ocn_hfield** f; int lx, ly; } tab[] = {
      {&s->eta, OCN_CENTER, OCN_CENTER}, {&s->U, OCN_FACE, OCN_CENTER},    {&s->V, OCN_CENTER, OCN_FACE},   {&s->etabar, OCN_CENTER, OCN_CENTER},
      {&s->Ubar, OCN_FACE, OCN_CENTER},  {&s->Vbar, OCN_CENTER, OCN_FACE}, {&s->GU, OCN_FACE, OCN_CENTER},  {&s->GV, OCN_CENTER, OCN_FACE},
      {&s->Hfc, OCN_FACE, OCN_CENTER},   {&s->Hcf, OCN_CENTER, OCN_FACE},  {&s->Hcc, OCN_CENTER, OCN_CENTER}};
  for (auto& t : tab) *t.f = nullptr;
  for (auto& t : tab)
    if (int rc = hfield_new(g, t.lx, t.ly, OCN_NOTHING, t.f)) {
      ocn_sefs_destroy(s);
      return rc;
    }
  if (hipMalloc((void**)&s->eta2, s->eta->n * sizeof(double)) != hipSuccess || hipMalloc((void**)&s->U2, s->U->n * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&s->V2, s->V->n * sizeof(double)) != hipSuccess) {
    ocn_sefs_destroy(s);
    return OCN_ENOMEM;
  }
  // H = sum!(H, dz) over the interior of each depth field (split_explicit_free_surface.jl:103-110)
  double H = 0.0;
  for (int k = 0; k < g->N[2]; ++k) H = k == 0 ? g->h_dzc[0] : H + g->h_dzc[k];
  for (ocn_hfield* f : {s->Hfc, s->Hcf, s->Hcc}) {
    std::vector<double> h(f->n, 0.0);
    for (int j = 0; j < f->S[1]; ++j)
      for (int i = 0; i < f->S[0]; ++i) h[(i + g->H[0]) + (size_t)(j + g->H[1]) * f->T[0]] = H;
    if (hipMemcpy(f->d, h.data(), f->n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      ocn_sefs_destroy(s);
      return OCN_EHIP;
    }
  }
  *out = s;
  return OCN_OK;
}

void ocn_sefs_destroy(ocn_sefs* s) {
  if (!s) return;
  hipStreamSynchronize(s->g->ctx->stream);
#ifndef OCN_HOST_EMU
  for (auto& t : s->trains)
    if (t.exec) hipGraphExecDestroy((hipGraphExec_t)t.exec);
#endif
  ocn_hgrid* g = s->g;
  for (ocn_hfield* f : {s->eta, s->U, s->V, s->etabar, s->Ubar, s->Vbar, s->GU, s->GV, s->Hfc, s->Hcf, s->Hcc})
    if (f) {
      hipFree(f->d);
      delete f;
      hgrid_release(g);          // the reference hfield_new took
    }
  hipFree(s->eta2); hipFree(s->U2); hipFree(s->V2);
  delete s;
  hgrid_release(g);
}

ocn_hfield* ocn_sefs_field(ocn_sefs* s, int which) {
  if (!s) return nullptr;
  ocn_hfield* tab[] = {s->eta, s->U, s->V, s->etabar, s->Ubar, s->Vbar, s->GU, s->GV, s->Hfc, s->Hcf, s->Hcc};
  return (which >= 0 && which < 11) ? tab[which] : nullptr;
}

int ocn_sefs_set_weights(ocn_sefs* s, int n, const double* velocity_weights, const double* free_surface_weights) {
  if (!s || n < 1 || !velocity_weights || !free_surface_weights) return OCN_EINVAL;
  s->substeps = n;
  s->wv.assign(velocity_weights, velocity_weights + n);
  s->wf.assign(free_surface_weights, free_surface_weights + n);
#ifndef OCN_HOST_EMU
  hipStreamSynchronize(s->g->ctx->stream);
  for (auto& t : s->trains)
    if (t.exec) hipGraphExecDestroy((hipGraphExec_t)t.exec);
#endif
  s->trains.clear();                              // the weights are baked into recorded launches
  return OCN_OK;
}

int ocn_sefs_substep(ocn_sefs* s, double dtau, int substep_index) {
  if (!s || substep_index < 1 || substep_index > s->substeps) return OCN_EINVAL;
  sefs_substep_plain(s, dtau, substep_index);
  return api_done(s->g->ctx, OCN_OK);
}

int ocn_sefs_substeps(ocn_sefs* s, double dtau, int first_index, int count, int fused) {
  if (!s || first_index < 1 || count < 0 || first_index + count - 1 > s->substeps) return OCN_EINVAL;
  ocn_ctx* ctx = s->g->ctx;
  if (count == 0) return OCN_OK;
  const bool fuse = fused && sefs_fusable(s);
  const bool one = fuse && fused >= 2;
  const bool multi = one && fused >= 3 && sefs_multi_ok(s) && count >= 2;
  auto issue = [&]() {
    if (one) {
      // cells neither set ever writes (halo rows behind a wall's first one) must agree between the two sets
      se_copy(ctx, s->eta2, s->eta->d, s->eta->n);
      se_copy(ctx, s->U2, s->U->d, s->U->n);
      se_copy(ctx, s->V2, s->V->d, s->V->n);
      int launches = 0;
      if (multi) {
        // all but the last substep SE_MS at a time (interior cells), the last one by the one-substep kernel, which also writes the halos
        int q = 0;
        const int width = sefs_multi_width(s);
        while (q < count - 1) {
          const int n = count - 1 - q < width ? count - 1 - q : width;
          sefs_substep_multi(s, dtau, first_index + q, n, launches & 1);
          q += n;
          ++launches;
        }
        sefs_substep_one(s, dtau, first_index + count - 1, launches & 1);
        ++launches;
      } else {
        for (int q = 0; q < count; ++q) sefs_substep_one(s, dtau, first_index + q, q & 1);
        launches = count;
      }
      if (launches & 1) {                              // an odd train ends in the second set: bring it home (the handles' pointers never change)
        se_copy(ctx, s->eta->d, s->eta2, s->eta->n);
        se_copy(ctx, s->U->d, s->U2, s->U->n);
        se_copy(ctx, s->V->d, s->V2, s->V->n);
      }
      return;
    }
    for (int q = 0; q < count; ++q) {
      if (fuse) sefs_substep_fused(s, dtau, first_index + q);
      else sefs_substep_plain(s, dtau, first_index + q);
    }
  };
#ifndef OCN_HOST_EMU
  static const bool no_graph = getenv("OCNHIP_NO_GRAPH") && atoi(getenv("OCNHIP_NO_GRAPH")) != 0;
  if (fuse && count >= 4 && !no_graph && !ctx->profiling) {
    uint64_t bits;
    memcpy(&bits, &dtau, 8);
    // with uniform weights (the default) a recorded train serves any first index: the blocks of a banded sub-cycle share one graph
    bool uniform = true;
    for (size_t q = 1; q < s->wv.size(); ++q) uniform = uniform && s->wv[q] == s->wv[0] && s->wf[q] == s->wf[0];
    const int kfirst = uniform ? 1 : first_index;
    for (auto& t : s->trains)
      if (t.dtau_bits == bits && t.first == kfirst && t.count == count && t.mode == (multi ? 3 : one ? 2 : 1) && t.exec) {
        OCN_HIP_CHECK(ctx, hipGraphLaunch((hipGraphExec_t)t.exec, ctx->stream));
        s->graph_replays += 1;
        return api_done(ctx, OCN_OK);
      }
    // record the train (kernel arguments are captured by value: field pointers, dtau and the weights of each substep)
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      issue();
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
      if (e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec) {
        hipGraphDestroy(graph);
        if (s->trains.size() >= 8) {
          hipGraphExecDestroy((hipGraphExec_t)s->trains.front().exec);
          s->trains.erase(s->trains.begin());
        }
        s->trains.push_back({bits, kfirst, count, multi ? 3 : one ? 2 : 1, (void*)exec});
        OCN_HIP_CHECK(ctx, hipGraphLaunch(exec, ctx->stream));
        s->graph_replays += 1;
        return api_done(ctx, OCN_OK);
      }
      if (graph) hipGraphDestroy(graph);
      (void)hipGetLastError();
      g_ocn_launch_err.err = hipSuccess;
    }
  }
#endif
  issue();
  return api_done(ctx, OCN_OK);
}

int ocn_sefs_graph_replays(const ocn_sefs* s, int64_t* replays) {
  if (!s || !replays) return OCN_EINVAL;
  *replays = s->graph_replays;
  return OCN_OK;
}

int ocn_sefs_barotropic_mode(ocn_sefs* s, const ocn_hfield* u, const ocn_hfield* v, int into_forcing) {
  if (!s || !u || !v) return OCN_EINVAL;
  ocn_hfield *U = into_forcing ? s->GU : s->U, *V = into_forcing ? s->GV : s->V;
  if (u->g != s->g || v->g != s->g || u->loc[2] != OCN_CENTER || v->loc[2] != OCN_CENTER || u->T[0] != U->T[0] || u->T[1] != U->T[1] ||
      v->T[0] != V->T[0] || v->T[1] != V->T[1]) {
    ocn_set_error(s->g->ctx, "ocn_sefs_barotropic_mode: u must be a (Face, Center, Center) and v a (Center, Face, Center) field of the free surface's grid");
    return OCN_EINVAL;
  }
  vsum(s, U, u, nullptr, 0, 0);
  vsum(s, V, v, nullptr, 0, 0);
  hfield_fill(U);
  hfield_fill(V);
  return api_done(s->g->ctx, OCN_OK);
}

int ocn_sefs_set_average_to_zero(ocn_sefs* s) {
  if (!s) return OCN_EINVAL;
  for (ocn_hfield* f : {s->etabar, s->Ubar, s->Vbar}) OCN_ASYNC(hipMemsetAsync(f->d, 0, f->n * sizeof(double), s->g->ctx->stream));
  return api_done(s->g->ctx, OCN_OK);
}

int ocn_sefs_corrector(ocn_sefs* s, ocn_hfield* u, ocn_hfield* v) {
  if (!s || !u || !v) return OCN_EINVAL;
  int rc = ocn_sefs_barotropic_mode(s, u, v, 0);
  if (rc) return rc;
  sefs_correct_launch(s, u, v);
  return api_done(s->g->ctx, OCN_OK);
}

int ocn_sefs_step(ocn_sefs* s, const ocn_hfield* Gnu, const ocn_hfield* Gnv, const ocn_hfield* Gmu, const ocn_hfield* Gmv, double dt, double chi) {
  if (!s || !Gnu || !Gnv || !Gmu || !Gmv) return OCN_EINVAL;
  if (!same_shape(Gnu, Gmu) || !same_shape(Gnv, Gmv) || Gnu->g != s->g || Gnu->loc[2] != OCN_CENTER || Gnu->T[0] != s->GU->T[0] ||
      Gnu->T[1] != s->GU->T[1] || Gnv->T[0] != s->GV->T[0] || Gnv->T[1] != s->GV->T[1])
    return OCN_EINVAL;
  ocn_ctx* ctx = s->g->ctx;
  int rc = ocn_sefs_set_average_to_zero(s);
  if (rc) return rc;
  // barotropic_mode!(G^U, G^V, grid, Gu, Gv) with Gu = (1.5 + chi) G^n - (0.5 + chi) G^- formed inside the sum
  vsum(s, s->GU, Gnu, Gmu, 1.5 + chi, 0.5 + chi);
  vsum(s, s->GV, Gnv, Gmv, 1.5 + chi, 0.5 + chi);
  return api_done(ctx, sefs_step_tail(s, dt));
}


/* ---- second slice: the hydrostatic AB2 step around the tendencies ------------------------------------------------------------ */
int ocn_hfield_ab2_step(ocn_hfield* f, const ocn_hfield* Gn, const ocn_hfield* Gm, double dt, double chi) {
  if (!f || !Gn || !Gm) return OCN_EINVAL;
  if (f->loc[2] != OCN_CENTER || !same_shape(f, Gn) || !same_shape(f, Gm)) {
    ocn_set_error(f->g->ctx, "ocn_hfield_ab2_step: the field and its two tendencies must be 3-D fields of one location on one grid");
    return OCN_EINVAL;
  }
  hy_ab2_launch(f, Gn, const_cast<ocn_hfield*>(Gm), dt, chi, false);
  return api_done(f->g->ctx, OCN_OK);
}

int ocn_hfield_store_tendency(ocn_hfield* Gm, const ocn_hfield* Gn) {
  if (!Gm || !Gn || !same_shape(Gm, Gn) || Gm->loc[2] != OCN_CENTER) return OCN_EINVAL;
  hy_store_launch(Gm, Gn);
  return api_done(Gm->g->ctx, OCN_OK);
}

int ocn_hydro_compute_w(const ocn_hfield* u, const ocn_hfield* v, ocn_hfield* w) {
  if (!u || !v || !w) return OCN_EINVAL;
  const ocn_hgrid* g = w->g;
  if (!is_loc(u, g, OCN_FACE, OCN_CENTER, OCN_CENTER) || !is_loc(v, g, OCN_CENTER, OCN_FACE, OCN_CENTER) || !is_loc(w, g, OCN_CENTER, OCN_CENTER, OCN_FACE)) {
    ocn_set_error(g->ctx, "ocn_hydro_compute_w: u, v, w must be (Face, Center, Center), (Center, Face, Center), (Center, Center, Face) fields of one grid");
    return OCN_EINVAL;
  }
  if (g->H[0] < 1 || g->H[1] < 1) {
    ocn_set_error(g->ctx, "ocn_hydro_compute_w: needs one halo cell in x and y");
    return OCN_EINVAL;
  }
  hy_w_launch(u, v, w);
  return api_done(g->ctx, OCN_OK);
}

static int hy_check_buoyancy(const ocn_hgrid* g, int kind, const ocn_hfield* T, const ocn_hfield* S) {
  if (kind < 0 || kind > 2 || (kind >= 1 && !is_loc(T, g, OCN_CENTER, OCN_CENTER, OCN_CENTER)) || (kind == 2 && !is_loc(S, g, OCN_CENTER, OCN_CENTER, OCN_CENTER))) {
    ocn_set_error(g->ctx, "buoyancy: kind 0 (none), 1 (BuoyancyTracer: T = b) or 2 (linear equation of state: T and S), on (Center, Center, Center) fields");
    return OCN_EINVAL;
  }
  if (g->H[2] < 1) {
    ocn_set_error(g->ctx, "the hydrostatic pressure integral reads one halo level in z");
    return OCN_EINVAL;
  }
  return OCN_OK;
}

int ocn_hydro_pressure(ocn_hfield* pHY, int kind, double g, double alpha, double beta, const ocn_hfield* T, const ocn_hfield* S) {
  if (!pHY) return OCN_EINVAL;
  const ocn_hgrid* hg = pHY->g;
  if (!is_loc(pHY, hg, OCN_CENTER, OCN_CENTER, OCN_CENTER)) return OCN_EINVAL;
  if (int rc = hy_check_buoyancy(hg, kind, T, S)) return rc;
  HyBuoy q{kind, g, alpha, beta};
  hy_pressure_launch(pHY, q, kind >= 1 ? T : nullptr, kind == 2 ? S : nullptr);
  return api_done(hg->ctx, OCN_OK);
}

int ocn_hydro_create(const ocn_hydro_desc* d, ocn_hydro** out) {
  if (!d || !out || !d->free_surface || !d->u || !d->v || !d->w || !d->pHY || d->ntracers < 0 || (d->ntracers > 0 && !d->tracers) || !d->Gn || !d->Gm)
    return OCN_EINVAL;
  ocn_hgrid* fg = d->free_surface->g;
  ocn_hgrid* g = d->u->g;                        // the grid of the 3-D fields
  ocn_ctx* ctx = fg->ctx;
  if (g != fg) {
    // the free surface of a model on latitude bands: replicated on the whole grid, or banded on the extended band (band_overlap)
    const bool banded = fg->overlap > 0;
    bool ok = g->slab && !fg->slab && g->ctx == fg->ctx && g->kind == fg->kind && g->N[0] == fg->N[0] && g->N[2] == fg->N[2] && g->radius == fg->radius &&
              g->gNy == (banded ? fg->gNy : fg->N[1]) && (!banded || (fg->own == g->N[1] && fg->j0 + fg->ext_lo == g->j0));
    for (int q = 0; q < 3; ++q)
      ok = ok && (g->H[q] == fg->H[q] || q == 1) && g->topo[q] == fg->topo[q] && g->x0[q] == fg->x0[q] && g->L[q] == fg->L[q];
    if (!ok) {
      ocn_set_error(ctx, "ocn_hydro_create: the 3-D fields must live on the free surface's grid or on a latitude band (partition = 1) of that very grid");
      return OCN_EINVAL;
    }
  }
  if (!is_loc(d->u, g, OCN_FACE, OCN_CENTER, OCN_CENTER) || !is_loc(d->v, g, OCN_CENTER, OCN_FACE, OCN_CENTER) ||
      !is_loc(d->w, g, OCN_CENTER, OCN_CENTER, OCN_FACE) || !is_loc(d->pHY, g, OCN_CENTER, OCN_CENTER, OCN_CENTER)) {
    ocn_set_error(ctx, "ocn_hydro_create: u, v, w, pHY must sit at their staggered locations on the free surface's grid");
    return OCN_EINVAL;
  }
  for (int q = 0; q < d->ntracers; ++q)
    if (!is_loc(d->tracers[q], g, OCN_CENTER, OCN_CENTER, OCN_CENTER)) {
      ocn_set_error(ctx, "ocn_hydro_create: tracer %d is not a (Center, Center, Center) field of the grid", q);
      return OCN_EINVAL;
    }
  for (int q = 0; q < 2 + d->ntracers; ++q) {
    const ocn_hfield* f = q == 0 ? d->u : q == 1 ? d->v : d->tracers[q - 2];
    if (!d->Gn[q] || !d->Gm[q] || !same_shape(f, d->Gn[q]) || !same_shape(f, d->Gm[q]) || d->Gn[q] == d->Gm[q]) {
      ocn_set_error(ctx, "ocn_hydro_create: tendency %d does not match its field", q);
      return OCN_EINVAL;
    }
  }
  const int kind = d->buoyancy_kind;
  if ((kind >= 1 && (d->T_index < 0 || d->T_index >= d->ntracers)) || (kind == 2 && (d->S_index < 0 || d->S_index >= d->ntracers || d->S_index == d->T_index))) {
    ocn_set_error(ctx, "ocn_hydro_create: buoyancy tracer indices out of range");
    return OCN_EINVAL;
  }
  if (int rc = hy_check_buoyancy(g, kind, kind >= 1 ? d->tracers[d->T_index] : nullptr, kind == 2 ? d->tracers[d->S_index] : nullptr)) return rc;
  if (g->H[0] < 1 || g->H[1] < 1) {
    ocn_set_error(ctx, "ocn_hydro_create: needs one halo cell in x and y");
    return OCN_EINVAL;
  }
  ocn_hydro* h = new ocn_hydro;
  h->fs = d->free_surface;
  h->lg = g;
  const long drow = (long)(g->j0 - fg->j0) + fg->H[1] - g->H[1];      // parent row of the free surface's arrays holding the model's parent row 0
  h->offU = drow * h->fs->U->T[0];
  h->offV = drow * h->fs->V->T[0];
  h->u = d->u; h->v = d->v; h->w = d->w; h->pHY = d->pHY;
  h->c.assign(d->tracers, d->tracers + d->ntracers);
  h->gn.assign(d->Gn, d->Gn + 2 + d->ntracers);
  h->gm.assign(d->Gm, d->Gm + 2 + d->ntracers);
  h->buoy = HyBuoy{kind, d->gravitational_acceleration, d->thermal_expansion, d->haline_contraction};
  h->bT = kind >= 1 ? d->T_index : -1;
  h->bS = kind == 2 ? d->S_index : -1;
  if (hipMalloc((void**)&h->Un, h->fs->U->n * sizeof(double)) != hipSuccess || hipMalloc((void**)&h->Vn, h->fs->V->n * sizeof(double)) != hipSuccess) {
    hipFree(h->Un);
    delete h;
    return OCN_ENOMEM;
  }
  OCN_ASYNC(hipMemsetAsync(h->Un, 0, h->fs->U->n * sizeof(double), ctx->stream));
  OCN_ASYNC(hipMemsetAsync(h->Vn, 0, h->fs->V->n * sizeof(double), ctx->stream));
  g->refs += 1;
  *out = h;
  return OCN_OK;
}

void ocn_hydro_destroy(ocn_hydro* h) {
  if (!h) return;
  ocn_hgrid* g = h->lg;
  hipStreamSynchronize(g->ctx->stream);
  hipFree(h->Un);
  hipFree(h->Vn);
  hipFree(h->frow);
  for (auto& e : h->imptab) hipFree(e.d);
  delete h;
  hgrid_release(g);
}

int ocn_hydro_update_state(ocn_hydro* h) {
  if (!h) return OCN_EINVAL;
  hydro_update_state(h, false);
  return api_done(h->fs->g->ctx, OCN_OK);
}

/* ab2_step!(model, dt, chi) as the reference issues it: barotropic mode of the velocities, AB2 steps of u, v and the tracers, then
 * the split-explicit free-surface step */
int ocn_hydro_ab2_step(ocn_hydro* h, double dt, double chi) {
  if (!h) return OCN_EINVAL;
  if (h->lg->slab) {
    ocn_set_error(h->lg->ctx, "a model on latitude bands steps through ocn_hydro_step_after_tendencies(fused = 1) / ocn_hydro_time_step only");
    return OCN_EUNSUPPORTED;
  }
  int rc = ocn_sefs_barotropic_mode(h->fs, h->u, h->v, 0);
  if (rc) return rc;
  hy_ab2_launch(h->u, h->gn[0], h->gm[0], dt, chi, false);
  hy_ab2_launch(h->v, h->gn[1], h->gm[1], dt, chi, false);
  if ((rc = hydro_implicit(h, h->u, 0, dt)) || (rc = hydro_implicit(h, h->v, 0, dt))) return rc;
  for (size_t q = 0; q < h->c.size(); ++q) hy_ab2_launch(h->c[q], h->gn[2 + q], h->gm[2 + q], dt, chi, false);
  for (size_t q = 0; q < h->c.size(); ++q)
    if ((rc = hydro_implicit(h, h->c[q], 1 + (int)q, dt))) return rc;
  return ocn_sefs_step(h->fs, h->gn[0], h->gn[1], h->gm[0], h->gm[1], dt, chi);
}

/* time_step! from ab2_step! on (quasi_adams_bashforth_2.jl:94-100): the step, the barotropic correction of the velocities,
 * G^- <- G^n, update_state!.  fused = 0: kernel by kernel as the reference; 1: the passes over the 3-D fields merged
 * (k_hy_momentum, k_hy_tracers), which leaves the same bits in every field, halos included. */
int ocn_hydro_step_after_tendencies(ocn_hydro* h, double dt, double chi, int fused) {
  if (!h) return OCN_EINVAL;
  ocn_sefs* s = h->fs;
  ocn_hgrid* g = h->lg;
  ocn_ctx* ctx = g->ctx;
  int rc;
  if (!fused && g->slab) fused = 1;
  if (!fused) {
    if ((rc = ocn_hydro_ab2_step(h, dt, chi))) return rc;
    if ((rc = ocn_sefs_corrector(s, h->u, h->v))) return rc;
    for (size_t q = 0; q < h->gn.size(); ++q) hy_store_launch(h->gm[q], h->gn[q]);
    hydro_update_state(h, false);
    return api_done(ctx, OCN_OK);
  }
  const double cn = 1.5 + chi, cm = 0.5 + chi;
  dim3 blk(64, 4, 1);
  const bool implicit = hydro_has_implicit(h);
  HyImp impv{nullptr, nullptr, nullptr, nullptr};
  const int visc = implicit && h->kap[0] != 0.0;
  if (visc && (rc = hydro_imp_table(h, h->kap[0], dt, &impv))) return api_done(ctx, rc);
  for (int q = 0; q < 2; ++q) {
    ocn_hfield *f = q ? h->v : h->u, *U = q ? s->V : s->U, *GU = q ? s->GV : s->GU;
    const long off = q ? h->offV : h->offU;
    ocn_launch(k_hy_momentum, dim3((f->S[0] + 63) / 64, (f->S[1] + 3) / 4, 1), blk, ctx->stream, f->d, (const double*)h->gn[q]->d, h->gm[q]->d, U->d + off,
               GU->d + off, (q ? h->Vn : h->Un) + off, dt, cn, cm, (const double*)g->dzc, f->S[0], f->S[1], g->N[0], g->N[1], g->N[2], g->H[0], g->H[1],
               g->H[2], (long)f->T[0], (long)f->T[0] * f->T[1], (long)U->T[0], impv, visc);
  }
  if ((rc = hydro_allgather_rows(h))) return api_done(ctx, rc);
  hfield_fill(s->U);
  hfield_fill(s->V);
  bool pressure_done = false;
  for (size_t q = 0; q < h->c.size(); ++q) {
    if ((int)q == h->bS && h->bT >= 0 && !implicit) continue;      // stepped together with T
    if ((int)q == h->bT && !implicit) {
      dim3 b, gr;
      hy_cols(g, b, gr);
      ocn_hfield *T = h->c[q], *S = h->bS >= 0 ? h->c[h->bS] : nullptr;
      ocn_launch(k_hy_tracers, gr, b, ctx->stream, hy_grid(g), h->buoy, T->d, (const double*)h->gn[2 + q]->d, h->gm[2 + q]->d, S ? S->d : (double*)nullptr,
                 S ? (const double*)h->gn[2 + h->bS]->d : (const double*)nullptr, S ? h->gm[2 + h->bS]->d : (double*)nullptr, h->pHY->d, dt, cn, cm,
                 (long)T->T[0], (long)T->T[0] * T->T[1]);
      pressure_done = true;
    } else {
      if (implicit && h->kap[1 + q] != 0.0) {
        // explicit step, G^- <- G^n and the implicit solve in one kernel; the hydrostatic pressure then comes from update_state!'s kernel
        HyImp it;
        if ((rc = hydro_imp_table(h, h->kap[1 + q], dt, &it))) return api_done(ctx, rc);
        dim3 b2, g2;
        hy_cols(g, b2, g2);
        ocn_hfield* f = h->c[q];
        ocn_launch(k_hy_ab2_implicit, g2, b2, ctx->stream, f->d, (const double*)h->gn[2 + q]->d, h->gm[2 + q]->d, dt, cn, cm, it, g->N[0], g->N[1], g->N[2],
                   g->H[0], g->H[1], g->H[2], (long)f->T[0], (long)f->T[0] * f->T[1]);
      } else {
        hy_ab2_launch(h->c[q], h->gn[2 + q], h->gm[2 + q], dt, chi, true);
      }
    }
  }
  // the free surface: G^U, G^V are in place
  for (ocn_hfield* f : {s->etabar, s->Ubar, s->Vbar}) OCN_ASYNC(hipMemsetAsync(f->d, 0, f->n * sizeof(double), ctx->stream));
  if ((rc = sefs_step_tail(s, dt))) return api_done(ctx, rc);
  // corrector: the barotropic mode of the stepped velocities was summed in the first pass
  se_copy(ctx, s->U->d, h->Un, s->U->n);
  se_copy(ctx, s->V->d, h->Vn, s->V->n);
  hfield_fill(s->U);
  hfield_fill(s->V);
  sefs_correct_launch(s, h->u, h->v, h->offU, h->offV);
  hydro_update_state(h, pressure_done);
  return api_done(ctx, OCN_OK);
}



/* closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa = (kappa_1, ...)) with constant coefficients:
 * implicit_step! of u, v and every tracer inside ab2_step! (hydrostatic_free_surface_ab2_step.jl:72-85,115-128); zeros switch it off */
int ocn_hydro_set_closure(ocn_hydro* h, double nu, int ntracers, const double* kappa) {
  if (!h || ntracers != (int)h->c.size() || (ntracers > 0 && !kappa) || !(nu >= 0)) return OCN_EINVAL;
  for (int q = 0; q < ntracers; ++q)
    if (!(kappa[q] >= 0)) return OCN_EINVAL;
  if (h->lg->H[2] < 1 && (nu > 0)) return OCN_EINVAL;
  h->kap.assign(1 + (size_t)ntracers, 0.0);
  h->kap[0] = nu;
  for (int q = 0; q < ntracers; ++q) h->kap[1 + q] = kappa[q];
  return OCN_OK;
}

/* ---- third slice: calculate_tendencies! and the whole time step ---------------------------------------------------------------- */
int ocn_hydro_set_physics(ocn_hydro* h, int momentum_advection, int coriolis, double coriolis_parameter, int tracer_advection) {
  if (!h) return OCN_EINVAL;
  ocn_hgrid* g = h->lg;
  ocn_ctx* ctx = g->ctx;
  if (momentum_advection < 0 || momentum_advection > 3 || coriolis < 0 || coriolis > 3 || tracer_advection < 0 || tracer_advection > 4) {
    ocn_set_error(ctx, "ocn_hydro_set_physics: momentum_advection 0..3, coriolis 0..3, tracer_advection 0..4");
    return OCN_EINVAL;
  }
  if ((coriolis == 1 || coriolis == 2) && g->kind != HG_LATLON) {
    ocn_set_error(ctx, "ocn_hydro_set_physics: HydrostaticSphericalCoriolis needs a LatitudeLongitudeGrid");
    return OCN_EINVAL;
  }
  {
    const int need = (tracer_advection >= 3 || momentum_advection == 3) ? 3 : tracer_advection == 2 ? 2 : 1;      // halo the schemes read
    if (g->H[0] < need || g->H[1] < need || g->H[2] < need || (g->topo[0] == OCN_PERIODIC && g->N[0] < need)) {
      ocn_set_error(ctx, "ocn_hydro_set_physics: the stencils of this configuration read %d halo cell(s) in every direction", need);
      return OCN_EINVAL;
    }
  }
  OCN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  hipFree(h->frow);
  h->frow = nullptr;
  if (coriolis == 1 || coriolis == 2) {
    // f^ffa[j] = 2 Omega sin(pi phi^f[j] / 180) (hydrostatic_spherical_coriolis.jl:32-33)
    std::vector<double> f(g->N[1] + 2 * g->H[1] + 1, NAN);
    for (size_t r = 0; r < f.size() && r < g->h_phif.size(); ++r) f[r] = 2 * coriolis_parameter * sin(M_PI * g->h_phif[r] / 180.0);
    if (int rc = upload(ctx, f, &h->frow)) return rc;
  }
  h->phys.madv = momentum_advection;
  h->phys.cor = coriolis;
  h->phys.tadv = tracer_advection;
  h->phys.f0 = coriolis_parameter;
  return OCN_OK;
}

int ocn_hydro_calculate_tendencies(ocn_hydro* h) {
  if (!h) return OCN_EINVAL;
  if (h->lg->H[0] < 1 || h->lg->H[1] < 1 || h->lg->H[2] < 1) return OCN_EINVAL;
  hydro_tendencies(h);
  return api_done(h->fs->g->ctx, OCN_OK);
}

/* time_step!(model, dt; euler) (TimeSteppers/quasi_adams_bashforth_2.jl:70-104) */
int ocn_hydro_time_step(ocn_hydro* h, double dt, int euler) {
  if (!h) return OCN_EINVAL;
  ocn_ctx* ctx = h->fs->g->ctx;
  const double chi = euler ? -0.5 : h->chi;
  if (euler)
    for (ocn_hfield* f : h->gm) OCN_ASYNC(hipMemsetAsync(f->d, 0, f->n * sizeof(double), ctx->stream));
  hydro_tendencies(h);
  return ocn_hydro_step_after_tendencies(h, dt, chi, 1);
}

}  // extern "C"
