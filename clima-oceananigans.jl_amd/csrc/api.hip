// api.hip -- the C ABI of include/ocnhip.h: handles, memory, and the stream-ordered orchestration of
// time_step! (TimeSteppers/quasi_adams_bashforth_2.jl:70-104, runge_kutta_3.jl:81-152).
#include <cstdarg>

#include "internal.h"

#ifdef OCN_HOST_EMU
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
emu_barrier g_emu_barrier;
double g_emu_shfl[4096];
std::recursive_mutex g_emu_launch_mutex;
#else
thread_local int g_ocn_dry = 0;
thread_local ocn_launch_error g_ocn_launch_err = {hipSuccess, {0}};
#endif
thread_local int g_ocn_capturing = 0, g_ocn_capture_poison = 0;

static char g_last_error[512] = {0};
static int pnhs_refresh(ocn_model* m);

void ocn_set_error(ocn_ctx* ctx, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  if (ctx) memcpy(ctx->err, g_last_error, sizeof(g_last_error));
}

// every entry point that launches work returns through this: a launch or asynchronous call that failed since the last
// report (compat.h ocn_note_error) turns the result into OCN_EHIP and names the call
static int api_ret(ocn_ctx* ctx, int rc) {
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

// ---- profiling -------------------------------------------------------------------------------------------
ProfScope::ProfScope(ocn_ctx* ctx, const char* nm) : c(ctx), name(nm) {
  if (!c->profiling || (!c->prof_only.empty() && c->prof_only != nm)) return;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
  if (!c->profiling || !a) return;
  hipEventRecord(b, c->stream);
  c->prof[name].ev.emplace_back(a, b);
}
static void prof_collect(ocn_ctx* c) {
  hipStreamSynchronize(c->stream);
  for (auto& kv : c->prof) {
    for (auto& e : kv.second.ev) {
      float ms = 0;
      hipEventElapsedTime(&ms, e.first, e.second);
      kv.second.total_ms += ms;
      kv.second.count += 1;
      hipEventDestroy(e.first);
      hipEventDestroy(e.second);
    }
    kv.second.ev.clear();
  }
}

extern "C" {

int ocn_abi_version(void) { return OCN_ABI_VERSION; }

const char* ocn_last_error(ocn_ctx* ctx) { return ctx ? ctx->err : g_last_error; }

int ocn_init(int device_id, ocn_ctx** out) {
  if (!out) return OCN_EINVAL;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ocn_set_error(nullptr, "no HIP device available (hipGetDeviceCount)");
    return OCN_EHIP;
  }
  if (device_id < 0 || device_id >= n) {
    ocn_set_error(nullptr, "device %d out of range (%d devices)", device_id, n);
    return OCN_EINVAL;
  }
  ocn_ctx* c = new ocn_ctx;
  c->device = device_id;
  OCN_HIP_CHECK(nullptr, hipSetDevice(device_id));
  OCN_HIP_CHECK(nullptr, hipStreamCreate(&c->stream));
  *out = c;
  return OCN_OK;
}

void ocn_destroy(ocn_ctx* ctx) {
  if (!ctx) return;
  prof_collect(ctx);
  comm_destroy(ctx);
  hipStreamDestroy(ctx->stream);
  delete ctx;
}

int ocn_sync(ocn_ctx* ctx) {
  if (!ctx) return OCN_EINVAL;
  for (ocn_model* m : ctx->models) {   // exchanges still travelling on the communication stream belong to the step too,
    if (halo_settle(m)) return OCN_EHIP;
    if (int rc = pnhs_refresh(m)) return rc;   // and so does the deferred one of pNHS
  }
  OCN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  OCN_HIP_CHECK(ctx, hipGetLastError());
  return api_ret(ctx, OCN_OK);
}

void* ocn_stream(ocn_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ocn_profile_enable(ocn_ctx* ctx, int on) {
  if (!ctx) return OCN_EINVAL;
  if (!on) prof_collect(ctx);
  ctx->profiling = on != 0;
  return OCN_OK;
}
int ocn_profile_filter(ocn_ctx* ctx, const char* phase) {
  if (!ctx) return OCN_EINVAL;
  ctx->prof_only = phase ? phase : "";
  return OCN_OK;
}
int ocn_profile_reset(ocn_ctx* ctx) {
  if (!ctx) return OCN_EINVAL;
  prof_collect(ctx);
  ctx->prof.clear();
  return OCN_OK;
}
int ocn_profile_read(ocn_ctx* ctx, const char* phase, double* avg_ms, int64_t* count) {
  if (!ctx || !phase) return OCN_EINVAL;
  prof_collect(ctx);
  auto it = ctx->prof.find(phase);
  if (it == ctx->prof.end() || it->second.count == 0) {
    if (avg_ms) *avg_ms = 0;
    if (count) *count = 0;
    return OCN_OK;
  }
  if (avg_ms) *avg_ms = it->second.total_ms / it->second.count;
  if (count) *count = it->second.count;
  return OCN_OK;
}

// ---- grid ---------------------------------------------------------------------------------------------------
// stretched-axis spacings incl. halos: Grids/grid_generation.jl:28-75 (Bounded and Periodic variants)
static void stretched_spacings(const std::vector<double>& Fi, int N, int H, bool bounded, std::vector<double>& dzc,
                               std::vector<double>& dzf) {
  std::vector<double> dm(H), dp(H);
  for (int i = 1; i <= H; ++i) {
    if (bounded) {
      dm[i - 1] = Fi[1] - Fi[0];
      dp[i - 1] = Fi[N] - Fi[N - 1];
    } else {
      dm[i - 1] = Fi[N - H + i] - Fi[N - H + i - 1];
      dp[i - 1] = Fi[i] - Fi[i - 1];
    }
  }
  std::vector<double> dpr(dp.rbegin(), dp.rend());
  std::vector<double> F;
  for (int i = 0; i < H; ++i) {
    double s = 0;
    for (int q = i; q < H; ++q) s += dm[q];
    F.push_back(Fi[0] - s);
  }
  for (int i = 0; i <= N; ++i) F.push_back(Fi[i]);
  std::vector<double> Fp;
  for (int i = 0; i < H; ++i) {
    double s = 0;
    for (int q = i; q < H; ++q) s += dpr[q];
    Fp.push_back(Fi[N] + s);
  }
  for (int i = H - 1; i >= 0; --i) F.push_back(Fp[i]);
  int TC = N + 2 * H;
  int TF = bounded ? N + 1 + 2 * H : N + 2 * H;
  std::vector<double> C(TC);
  for (int i = 0; i < TC; ++i) C[i] = (F[i + 1] + F[i]) / 2;
  std::vector<double> dF;
  for (int i = 1; i < TC; ++i) dF.push_back(C[i] - C[i - 1]);
  F.resize(TF);
  dzc.resize(TF - 1);
  for (int i = 0; i < TF - 1; ++i) dzc[i] = F[i + 1] - F[i];
  std::vector<double> d2;
  d2.push_back(dF.front());
  for (double x : dF) d2.push_back(x);
  d2.push_back(dF.back());
  for (int i = (int)d2.size() - 1; i >= 1; --i) d2[i] = d2[i - 1];
  dzf = d2;
  // make sure kernels can index dzc[k+H] for k in [-H, N+H] and dzf[k+H+1] for k in [-H-1, N+H]
  while ((int)dzc.size() < N + 2 * H + 1) dzc.push_back(dzc.back());
  while ((int)dzf.size() < N + 2 * H + 2) dzf.push_back(dzf.back());
}

static int grid_build_dev(ocn_grid* g) {
  GridDev& d = g->dev;
  memset(&d, 0, sizeof(d));
  d.Nx = g->N[0]; d.Ny = g->N[1]; d.Nz = g->N[2];
  // Flat x / y: one cell stored with as many broadcast halo cells as the widest real halo
  int hmax = 1;
  for (int q = 0; q < 3; ++q) hmax = g->H[q] > hmax ? g->H[q] : hmax;
  for (int q = 0; q < 3; ++q) g->PH[q] = (q < 2 && g->topo[q] == OCN_FLAT) ? hmax : g->H[q];
  d.Hx = g->PH[0]; d.Hy = g->PH[1]; d.Hz = g->PH[2];
  d.xb = g->topo[0] == OCN_BOUNDED;
  d.yb = g->topo[1] == OCN_BOUNDED;
  d.sy = d.Nx + 2 * d.Hx + d.xb;                 // one pitch for Face- and Center-located fields
  d.sz = d.sy * (d.Ny + 2 * d.Hy + d.yb);
  // regular axes: L/N (grid_generation.jl:84; the reference rounds a BigFloat quotient once)
  d.dx = (double)((long double)g->L[0] / g->N[0]);
  d.dy = (double)((long double)g->L[1] / (g->dist_y ? g->Nyg : g->N[1]));
  d.rdx = 1.0 / d.dx;
  d.rdy = 1.0 / d.dy;
  d.zb = g->topo[2] == OCN_BOUNDED;
  d.zflat = g->topo[2] == OCN_FLAT;
  d.dz = d.zflat ? 1.0 : (double)((long double)g->L[2] / (g->dist ? g->Nzg : g->N[2]));
  hipFree(g->d_dzc);
  hipFree(g->d_dzf);
  hipFree(g->d_rdzc);
  hipFree(g->d_rdzf);
  g->d_dzc = g->d_dzf = g->d_rdzc = g->d_rdzf = nullptr;
  d.rdz = 1.0 / d.dz;
  if (!g->z_regular) {
    stretched_spacings(g->zF_int, g->N[2], g->H[2], g->topo[2] == OCN_BOUNDED, g->h_dzc, g->h_dzf);
    OCN_HIP_CHECK(g->ctx, hipMalloc((void**)&g->d_dzc, g->h_dzc.size() * sizeof(double)));
    OCN_HIP_CHECK(g->ctx, hipMalloc((void**)&g->d_dzf, g->h_dzf.size() * sizeof(double)));
    OCN_HIP_CHECK(g->ctx, hipMemcpy(g->d_dzc, g->h_dzc.data(), g->h_dzc.size() * sizeof(double), hipMemcpyHostToDevice));
    OCN_HIP_CHECK(g->ctx, hipMemcpy(g->d_dzf, g->h_dzf.data(), g->h_dzf.size() * sizeof(double), hipMemcpyHostToDevice));
    d.dzc = g->d_dzc;
    d.dzf = g->d_dzf;
    std::vector<double> rc(g->h_dzc.size()), rf(g->h_dzf.size());
    for (size_t i = 0; i < rc.size(); ++i) rc[i] = 1.0 / g->h_dzc[i];
    for (size_t i = 0; i < rf.size(); ++i) rf[i] = 1.0 / g->h_dzf[i];
    OCN_HIP_CHECK(g->ctx, hipMalloc((void**)&g->d_rdzc, rc.size() * sizeof(double)));
    OCN_HIP_CHECK(g->ctx, hipMalloc((void**)&g->d_rdzf, rf.size() * sizeof(double)));
    OCN_HIP_CHECK(g->ctx, hipMemcpy(g->d_rdzc, rc.data(), rc.size() * sizeof(double), hipMemcpyHostToDevice));
    OCN_HIP_CHECK(g->ctx, hipMemcpy(g->d_rdzf, rf.data(), rf.size() * sizeof(double), hipMemcpyHostToDevice));
    d.rdzc = g->d_rdzc;
    d.rdzf = g->d_rdzf;
  }
  return OCN_OK;
}

int ocn_grid_create(ocn_ctx* ctx, const ocn_grid_desc* desc, ocn_grid** out) {
  if (!ctx || !desc || !out) return OCN_EINVAL;
  for (int d = 0; d < 3; ++d) {
    if (desc->N[d] < 1 || desc->H[d] < 0) {
      ocn_set_error(ctx, "invalid size/halo in direction %d", d);
      return OCN_EINVAL;
    }
    if (desc->topology[d] < OCN_PERIODIC || desc->topology[d] > OCN_FLAT) return OCN_EINVAL;
  }
  ocn_grid* g = new ocn_grid;
  g->ctx = ctx;
  g->d = *desc;
  for (int d = 0; d < 3; ++d) {
    g->N[d] = desc->N[d];
    g->H[d] = desc->H[d];
    g->topo[d] = desc->topology[d];
    g->L[d] = desc->L[d];
    g->x0[d] = desc->x0[d];
  }
  for (int d = 0; d < 3; ++d)
    if (g->topo[d] == OCN_FLAT) {   // Grids/grid_utils.jl: Flat directions have one cell, no halo, unit spacing
      g->N[d] = 1;
      g->H[d] = 0;
      g->L[d] = 1.0;
    }
  g->z_regular = desc->z_faces == nullptr;
  if (!g->z_regular) {
    if (g->topo[2] == OCN_FLAT) {
      ocn_set_error(ctx, "z_faces given for a Flat z direction");
      delete g;
      return OCN_EINVAL;
    }
    g->zF_int.assign(desc->z_faces, desc->z_faces + g->N[2] + 1);
    for (int k = 0; k < g->N[2]; ++k)
      if (!(g->zF_int[k + 1] > g->zF_int[k])) {
        ocn_set_error(ctx, "z_faces must be strictly increasing");
        delete g;
        return OCN_EINVAL;
      }
    g->L[2] = g->zF_int[g->N[2]] - g->zF_int[0];
    g->x0[2] = g->zF_int[0];
    if (g->topo[2] == OCN_PERIODIC) {
      ocn_set_error(ctx, "a stretched Periodic z axis has no pressure solver in the reference either "
                         "(NonhydrostaticModels.jl:18-27)");
      delete g;
      return OCN_EUNSUPPORTED;
    }
  }
  g->Nzg = g->N[2];
  g->Nyg = g->N[1];
  // Decomposition (Distributed/multi_architectures.jl:20-47), chosen from the topology:
  //   triply Periodic, regular z     -> z-slabs, ranks (1, 1, R): contiguous halo planes, transpose-free Poisson
  //   (Periodic, Periodic, Bounded)  -> y-slabs, ranks (1, R, 1): every rank keeps whole columns, so walls,
  //                                     the hydrostatic integral and the tridiagonal solve stay local
  // OCNHIP_FORCE_DIST=1 exercises the slab code paths (pack / exchange with self) on one rank, where eligible.
  const bool want = ctx->nranks > 1 || (getenv("OCNHIP_FORCE_DIST") && atoi(getenv("OCNHIP_FORCE_DIST")) != 0);
  const bool xyper = g->topo[0] == OCN_PERIODIC && g->topo[1] == OCN_PERIODIC;
  const bool zslab_ok = xyper && g->topo[2] == OCN_PERIODIC && g->z_regular && g->N[2] >= 6 * ctx->nranks;
  const bool yslab_ok = xyper && g->topo[2] == OCN_BOUNDED && g->N[1] >= 6 * ctx->nranks;
  if (want && ctx->nranks > 1 && !zslab_ok && !yslab_ok) {
    ocn_set_error(ctx, "no slab decomposition for this grid: needs (Periodic, Periodic, Periodic regular) with Nz >= 6 R "
                       "or (Periodic, Periodic, Bounded) with Ny >= 6 R");
    delete g;
    return OCN_EUNSUPPORTED;
  }
  g->dist = want && zslab_ok;
  g->dist_y = want && !zslab_ok && yslab_ok;
  if (g->dist) {
    const char* dsolver = getenv("OCNHIP_DIST_SOLVER");
    const bool transposed = dsolver && strcmp(dsolver, "transpose") == 0;   // the all-to-all variant also cuts ky into R bands
    if (g->N[2] % ctx->nranks != 0 || (transposed && g->N[1] % ctx->nranks != 0)) {
      ocn_set_error(ctx, "Nz%s must be divisible by the number of ranks (%d)", transposed ? " and Ny" : "", ctx->nranks);
      delete g;
      return OCN_EINVAL;
    }
    g->N[2] = g->Nzg / ctx->nranks;
  }
  if (g->dist_y) {
    if (g->N[1] % ctx->nranks != 0) {
      ocn_set_error(ctx, "Ny must be divisible by the number of ranks (%d)", ctx->nranks);
      delete g;
      return OCN_EINVAL;
    }
    g->N[1] = g->Nyg / ctx->nranks;
  }
  for (int d = 0; d < 3; ++d)
    if (g->topo[d] != OCN_FLAT && !(g->L[d] > 0)) {
      ocn_set_error(ctx, "extent must be positive in direction %d", d);
      delete g;
      return OCN_EINVAL;
    }
  int rc = grid_build_dev(g);
  if (rc) {
    delete g;
    return rc;
  }
  *out = g;
  return OCN_OK;
}

void ocn_grid_destroy(ocn_grid* g) {
  if (!g) return;
  hipFree(g->d_dzc);
  hipFree(g->d_dzf);
  hipFree(g->d_rdzc);
  hipFree(g->d_rdzf);
  delete g;
}

}  // extern "C"

// ---- fields ---------------------------------------------------------------------------------------------------
double* Field::interior() const { return d + Hx + Hy * sy + Hz * sz; }
static bool field_dense(const Field& f) { return f.P[0] == f.T[0] && f.P[1] == f.T[1] && f.P[2] == f.T[2]; }

static int field_alloc(ocn_ctx* ctx, ocn_grid* g, Field& f, int lx, int ly, int lz) {
  f.loc[0] = lx; f.loc[1] = ly; f.loc[2] = lz;
  for (int d = 0; d < 3; ++d) {
    int loc = f.loc[d];
    if (g->topo[d] == OCN_FLAT) f.T[d] = g->N[d];
    else if (loc == OCN_FACE && g->topo[d] == OCN_BOUNDED) f.T[d] = g->N[d] + 1 + 2 * g->H[d];
    else f.T[d] = g->N[d] + 2 * g->H[d];
    f.P[d] = f.T[d];
    f.off[d] = 0;
    if (d < 2 && g->topo[d] == OCN_FLAT) {
      f.P[d] = 1 + 2 * g->PH[d];
      f.off[d] = g->PH[d];
    } else if (d < 2 && g->topo[d] == OCN_BOUNDED) {
      f.P[d] = g->N[d] + 1 + 2 * g->H[d];
    }
  }
  f.n = (size_t)f.P[0] * f.P[1] * f.P[2];
  f.sy = f.P[0];
  f.sz = (long)f.P[0] * f.P[1];
  f.Hx = g->PH[0]; f.Hy = g->PH[1]; f.Hz = g->PH[2];
  OCN_HIP_CHECK(ctx, hipMalloc((void**)&f.d, f.n * sizeof(double)));
  OCN_HIP_CHECK(ctx, hipMemsetAsync(f.d, 0, f.n * sizeof(double), ctx->stream));
  f.present = true;
  for (int s = 0; s < 6; ++s) f.bc[s] = BCdev{OCN_BC_NONE, 0.0, nullptr};
  return OCN_OK;
}
static int field_alloc(ocn_model* m, Field& f, int lx, int ly, int lz) { return field_alloc(m->ctx, m->g, f, lx, ly, lz); }

// default boundary conditions (BoundaryConditions/field_boundary_conditions.jl:13-35)
static void default_bcs(ocn_model* m, Field& f, bool auxiliary) {
  for (int d = 0; d < 3; ++d) {
    int kind;
    int topo = m->g->topo[d];
    if (topo == OCN_PERIODIC) kind = OCN_BC_PERIODIC;
    else if (topo == OCN_FLAT) kind = OCN_BC_NONE;
    else if (f.loc[d] == OCN_CENTER) kind = OCN_BC_NOFLUX;
    else kind = auxiliary ? OCN_BC_NONE : OCN_BC_IMPENETRABLE;
    f.bc[2 * d] = BCdev{kind, 0.0, nullptr};
    f.bc[2 * d + 1] = BCdev{kind, 0.0, nullptr};
  }
}

Field* model_field(ocn_model* m, int id) {
  Field* f = nullptr;
  if (id == OCN_F_U) f = &m->u;
  else if (id == OCN_F_V) f = &m->v;
  else if (id == OCN_F_W) f = &m->w;
  else if (id == OCN_F_PHY) f = &m->pHY;
  else if (id == OCN_F_PNHS) f = &m->pNHS;
  else if (id >= OCN_F_GN && id < OCN_F_GN + 3 + m->nt) f = m->gn_alias_gm ? &m->Gm[id - OCN_F_GN] : &m->Gn[id - OCN_F_GN];
  else if (id >= OCN_F_GM && id < OCN_F_GM + 3 + m->nt) f = &m->Gm[id - OCN_F_GM];
  else if (id >= OCN_F_TRACER && id < OCN_F_TRACER + m->nt) f = &m->tr[id - OCN_F_TRACER];
  else if (id == OCN_F_NU) f = &m->nu_e;
  else if (id >= OCN_F_KAPPA && id < OCN_F_KAPPA + m->nt) f = &m->kappa_e[id - OCN_F_KAPPA];
  if (f && !f->present) return nullptr;
  return f;
}

// ---- halo fills (fill_halo_regions.jl:34-102) ---------------------------------------------------------------
// Three passes in the order of permute_boundary_conditions (:55-83): sortperm of (west, south, bottom) with the
// non-strict `fill_first` comparator (:91-99), i.e. Julia's insertion sort for three elements.  lt(a, b) is
// false only when a is Periodic and b is not, so non-periodic directions come first and an all-equal triple
// comes out reversed (z, y, x).
static void fill_order(const ocn_grid* g, int order[3]) {
  auto lt = [&](int a, int b) { return !(g->topo[a] == OCN_PERIODIC && g->topo[b] != OCN_PERIODIC); };
  int v[3] = {0, 1, 2};
  for (int i = 1; i < 3; ++i) {
    int x = v[i], j = i;
    while (j > 0 && lt(x, v[j - 1])) {
      v[j] = v[j - 1];
      --j;
    }
    v[j] = x;
  }
  for (int i = 0; i < 3; ++i) order[i] = v[i];
}

static int fill_fields(ocn_model* m, Field** fs, int n) {
  if (n == 0) return OCN_OK;
  if (halo_settle(m)) return OCN_EHIP;
  ProfScope ps(m->ctx, "fill_halos");
  FieldPtrs F;
  F.n = n;
  for (int i = 0; i < n; ++i) {
    F.p[i] = fs[i]->d;
    F.Tx[i] = fs[i]->off[0] ? fs[i]->P[0] : fs[i]->T[0];
    F.Ty[i] = fs[i]->off[1] ? fs[i]->P[1] : fs[i]->T[1];
    F.Tz[i] = fs[i]->T[2];
  }
  int order[3], rc = OCN_OK;
  fill_order(m->g, order);
  for (int t = 0; t < 3; ++t) {
    const int d = order[t];
    // y directly followed by x, both Periodic and local: one pass over the whole halo frame
    if (d == 1 && t < 2 && order[t + 1] == 0 && m->g->topo[0] == OCN_PERIODIC && m->g->topo[1] == OCN_PERIODIC &&
        !m->g->dist_y && launch_fill_periodic_xy(m, F)) {
      ++t;
      continue;
    }
    if (m->g->topo[d] == OCN_BOUNDED) {
      launch_fill_bounded(m, fs, n, d);
    } else if (m->g->topo[d] == OCN_PERIODIC) {
      if (d == 2 && m->g->dist) rc = comm_halo_exchange_z(m, fs, n);
      else if (d == 1 && m->g->dist_y) rc = comm_halo_exchange_y(m, fs, n);
      else launch_fill_periodic(m, F, d);
      if (rc) return rc;
    }
  }
  // Flat x / y: refresh the broadcast copies last, over the full extent of the other directions
  for (int d = 0; d < 2; ++d)
    if (m->g->topo[d] == OCN_FLAT) launch_fill_flat(m, F, d);
  return OCN_OK;
}

static int fill_velocities_tracers(ocn_model* m, bool tracers) {
  Field* fs[OCN_NF];
  int n = 0;
  fs[n++] = &pred_u(m);   // the predictor between time-stepper update and projection, else u, v, w themselves
  fs[n++] = &pred_v(m);
  fs[n++] = &pred_w(m);
  if (tracers)
    for (int t = 0; t < m->nt; ++t) fs[n++] = &m->tr[t];
  return fill_fields(m, fs, n);
}

static int update_state(ocn_model* m) {
  // update_nonhydrostatic_model_state.jl:14-37
  if (halo_settle(m)) return OCN_EHIP;
  int rc = fill_velocities_tracers(m, true);
  if (rc) return rc;
  if (m->d.closure == OCN_CLOSURE_AMD) {
    launch_amd(m);
    Field* fs[OCN_NF];
    int n = 0;
    fs[n++] = &m->nu_e;
    for (int t = 0; t < m->nt; ++t) fs[n++] = &m->kappa_e[t];
    if ((rc = fill_fields(m, fs, n))) return rc;
  }
  launch_hydrostatic(m);
  if (m->pHY.present && m->d.buoyancy != OCN_BUOYANCY_NONE) {
    Field* fs[1] = {&m->pHY};
    if ((rc = fill_fields(m, fs, 1))) return rc;
  }
  return OCN_OK;
}

static int pressure_correction(ocn_model* m, double dt) {
  // pressure_correction.jl:10-23
  int rc = fill_velocities_tracers(m, false);
  if (rc) return rc;
  if ((rc = poisson_solve(m, dt))) return rc;
  Field* fs[1] = {&m->pNHS};
  return fill_fields(m, fs, 1);
}

// store_tendencies! (store_tendencies.jl:14-36) without the copy: G^- takes over G^n's buffers, and G^n gets
// the old G^- buffers, which the next tendency evaluation overwrites completely.
static void store_by_swap(ocn_model* m) {
  for (int f = 0; f < 3 + m->nt; ++f) std::swap(m->Gn[f], m->Gm[f]);
}

// Phase-level callers expect two distinct arrays: make G^n a real copy of G^- again when they are aliased.
static int materialize_gn(ocn_model* m) {
  if (!m->gn_alias_gm) return OCN_OK;
  for (int f = 0; f < 3 + m->nt; ++f)
    OCN_HIP_CHECK(m->ctx, hipMemcpyAsync(m->Gn[f].d, m->Gm[f].d, m->Gm[f].n * sizeof(double), hipMemcpyDeviceToDevice, m->ctx->stream));
  m->gn_alias_gm = false;
  return OCN_OK;
}

static int zero_Gm(ocn_model* m) {
  if (g_ocn_dry) return OCN_OK;
  for (int f = 0; f < 3 + m->nt; ++f)
    OCN_HIP_CHECK(m->ctx, hipMemsetAsync(m->Gm[f].d, 0, m->Gm[f].n * sizeof(double), m->ctx->stream));
  return OCN_OK;
}

// one fused (sub)step of the fast path: tendencies + update, rhs, solve, projection + halo images
// An exchange started by the previous (sub)step may still be in flight: everybody who reads z halos waits for it here.
static int halo_settle_one(ocn_model* m) {
  if (m->halo_inflight) OCN_HIP_CHECK(m->ctx, hipStreamWaitEvent(m->ctx->stream, m->ctx->ev_halo, 0));
  if (m->halo2_inflight) OCN_HIP_CHECK(m->ctx, hipStreamWaitEvent(m->ctx->stream, m->ctx->ev_halo2, 0));
  m->halo_inflight = m->halo2_inflight = false;
  return OCN_OK;
}
// The communication stream, its events and the communicator belong to the CONTEXT, the in-flight flags to a model: with two
// slab models on one context, B's exchanges on the main stream must not start while A's overlapped exchange is still
// travelling (one exchange of a communicator in flight at a time).  So settling means settling every model of the context;
// the events are recorded in stream order on the one communication stream, so waiting for the latest record covers all.
int halo_settle(ocn_model* m) {
  for (ocn_model* o : m->ctx->models)
    if (int rc = halo_settle_one(o)) return rc;
  return OCN_OK;
}
// deferred z-halo exchange of pNHS (see ocn_model::pnhs_halo_stale); collective
static int pnhs_refresh(ocn_model* m) {
  if (!m->pnhs_halo_stale) return OCN_OK;
  if (int rc = halo_settle(m)) return rc;
  m->pnhs_halo_stale = false;
  Field* fs[1] = {&m->pNHS};
  return comm_halo_exchange_z(m, fs, 1);
}
static int halo_settle_others(ocn_model* m) {
  for (ocn_model* o : m->ctx->models)
    if (o != m)
      if (int rc = halo_settle_one(o)) return rc;
  return OCN_OK;
}

static int overlap_streams(ocn_ctx* c) {
  if (c->overlap_ready) return OCN_OK;
  OCN_HIP_CHECK(c, hipStreamCreate(&c->comm_stream));
  OCN_HIP_CHECK(c, hipEventCreate(&c->ev_main));
  OCN_HIP_CHECK(c, hipEventCreate(&c->ev_halo));
  OCN_HIP_CHECK(c, hipEventCreate(&c->ev_halo2));
  c->overlap_ready = true;
  return OCN_OK;
}

static int fused_substep(ocn_model* m, double dt_full, double cn, double cm, int use_m, double dt_stage, bool swap) {
  const int Hz = m->gd.Hz, Nz = m->gd.Nz;
  if (m->g->dist)
    if (int rc0 = halo_settle_others(m)) return rc0;   // another model's planes may still be travelling on the shared stream
  if (m->halo_inflight && Nz > 2 * Hz + 2) {
    // z-slabs: the halo planes of u, v, w are still on their way (started after the last projection).  The interior levels
    // touch none of them; the first and last Hz levels follow once the planes have landed.
    launch_fused_tend_step(m, dt_full, cn, cm, use_m, Hz, Nz - Hz);
    OCN_HIP_CHECK(m->ctx, hipStreamWaitEvent(m->ctx->stream, m->ctx->ev_halo, 0));
    m->halo_inflight = false;
    launch_fused_tend_step(m, dt_full, cn, cm, use_m, 0, Hz, Nz - Hz, Nz);   // both ends in one launch
  } else {
    int rc0 = halo_settle(m);
    if (rc0) return rc0;
    launch_fused_tend_step(m, dt_full, cn, cm, use_m);
  }
  launch_tracer_steps(m, dt_full, cn, cm, use_m);                  // passive tracers: old velocities, own update
  if (m->halo2_inflight) {   // pNHS planes: one exchange of a communicator at a time -- the next one (w* below) waits for them
    OCN_HIP_CHECK(m->ctx, hipStreamWaitEvent(m->ctx->stream, m->ctx->ev_halo2, 0));
    m->halo2_inflight = false;
  }
  int rc = OCN_OK;
  // w* of the level above the slab, for the divergence at the top level: exchanged as a plane, or -- Green's-function z stage
  // with the custom passes -- left at zero here and brought in by its owner in spectral space (poisson.hip bplane): no exchange
  if (m->g->dist && !poisson_local_wstar(m) && (rc = fused_exchange_ws(m))) return rc;
  if (poisson_custom_xy(m)) {
    rc = poisson_run_from_predictor(m, dt_stage);                  // rhs fused into the x transform
  } else {
    launch_rhs_wrap(m, dt_stage, poisson_rhs_buffer(m->solver));
    rc = poisson_run(m);
  }
  if (rc) return rc;
  // p of the level below the slab, for d_z p at the first level: computed by this rank with one more level of the Green's-function
  // convolution (poisson.hip pbelow), or received from the lower neighbour
  if (m->g->dist && !poisson_local_phi_below(m) && (rc = fused_exchange_phi(m, poisson_rhs_buffer(m->solver)))) return rc;
  launch_project(m, dt_stage, poisson_rhs_buffer(m->solver));
  if (m->g->dist) {
    ocn_ctx* c = m->ctx;
    // Worth it when the transfer it hides is longer than what the split costs (0.05-0.07 ms): 19 MB per direction at config 4
    // (512 x 512 planes), 5 MB on 256 x 256 planes -- about break-even on an xGMI link, left alone.  A forced one-rank slab run
    // copies on the device: nothing to hide.
    const size_t halo_bytes = (size_t)(3 + m->nt) * Hz * m->u.sz * sizeof(double);
    const bool want = m->knob_overlap >= 0 ? m->knob_overlap != 0 : (c->nranks > 1 && halo_bytes >= ((size_t)8 << 20));
    if (want && comm_can_overlap(c) && Nz > 2 * Hz + 2 && overlap_streams(c) == OCN_OK) {
      // The halo planes travel on the communication stream while this stream goes on with the interior levels of the next
      // tendency kernel: what the next (sub)step needs first (u, v, w, tracers) in one group, pNHS (read by nobody until
      // output) behind it.  Only one exchange of the communicator is ever in flight: every later one waits for both events.
      Field* fa[3 + OCN_MAX_TRACERS] = {&m->u, &m->v, &m->w};
      int na = 3;
      for (int t = 0; t < m->nt; ++t) fa[na++] = &m->tr[t];
      OCN_HIP_CHECK(c, hipEventRecord(c->ev_main, c->stream));
      OCN_HIP_CHECK(c, hipStreamWaitEvent(c->comm_stream, c->ev_main, 0));
      if ((rc = comm_halo_exchange_z(m, fa, na, c->comm_stream))) return rc;
      OCN_HIP_CHECK(c, hipEventRecord(c->ev_halo, c->comm_stream));
      m->halo_inflight = true;
    } else {
      Field* fs[3 + OCN_MAX_TRACERS] = {&m->u, &m->v, &m->w};
      int nf = 3;
      for (int t = 0; t < m->nt; ++t) fs[nf++] = &m->tr[t];
      if ((rc = comm_halo_exchange_z(m, fs, nf))) return rc;
    }
    m->pnhs_halo_stale = true;   // the z halos of pNHS: exchanged when somebody can look at them (pnhs_refresh)
  }
  if (swap)
    for (int f = 0; f < 3 + m->nt; ++f) std::swap(m->Gn[f], m->Gm[f]);   // store_tendencies! as a pointer swap
  return OCN_OK;
}

// tendencies + time-stepper update of the general path.  With a Bounded z and an upwind scheme the momentum advection and
// the update of u, v, w come from the tiled kernel (fused.hip, ZB): the general kernels deliver the other terms.
static void tendencies_and_step(ocn_model* m, double dt, double cn, double cm, int use_m) {
  if (!m->bz_fast) {
    launch_tendencies(m);
    launch_step(m, dt, cn, cm, use_m);
    return;
  }
  const bool tr3 = m->nt > 0 && fused_tracer3_ok(m);
  launch_tendencies(m, true, tr3);             // closure, Coriolis, pressure gradient, boundary fluxes; tracers: everything but the
                                               // boundary fluxes comes from the tiled tracer kernel when it applies
  launch_fused_bz(m, dt, cn, cm, use_m);       // + advection -> G^n; stepped velocities -> us, vs, ws
  if (tr3) launch_tracer3(m, dt, cn, cm, use_m, true);   // reads the old velocities: before the swap
  else launch_step(m, dt, cn, cm, use_m, true);
  m->pred_active = true;                       // U* stays in us / vs / ws (halos filled by the pressure-correction step that
}                                              // follows); launch_pcorrect writes u = U* - dt grad p and clears the flag

static int time_step_ab2(ocn_model* m, double dt, int force_euler) {
  // quasi_adams_bashforth_2.jl:70-104
  bool euler = force_euler || (dt != m->previous_dt);
  double chi = euler ? -0.5 : m->d.chi;
  int rc0 = euler ? zero_Gm(m) : OCN_OK;
  if (rc0) return rc0;
  m->previous_dt = dt;
  if (m->iteration == 0 && (rc0 = update_state(m))) return rc0;
  if (m->fast_path) {
    m->gn_alias_gm = false;
    int rc = fused_substep(m, dt, 1.5 + chi, -(0.5 + chi), 1, dt, true);
    if (rc) return rc;
    m->gn_alias_gm = true;   // G^n == G^- after store_tendencies!
    m->time += dt;
    m->iteration += 1;
    m->stage = 1;
    return OCN_OK;           // update_state!: halos were written by the projection; no pHY', no closure
  }
  if (m->gn_alias_gm) m->gn_alias_gm = false;   // G^n gets its own (recycled) buffer again
  tendencies_and_step(m, dt, 1.5 + chi, -(0.5 + chi), 1);
  int rc = pressure_correction(m, dt);
  if (rc) return rc;
  launch_pcorrect(m, dt);
  store_by_swap(m);
  m->gn_alias_gm = true;
  m->time += dt;
  m->iteration += 1;
  m->stage = 1;
  return update_state(m);
}

static int time_step_rk3(ocn_model* m, double dt) {
  // runge_kutta_3.jl:57-62,81-152
  int rc0 = m->iteration == 0 ? update_state(m) : OCN_OK;
  if (rc0) return rc0;
  const double g1 = 8.0 / 15.0, g2 = 5.0 / 12.0, g3 = 3.0 / 4.0, z2 = -17.0 / 60.0, z3 = -5.0 / 12.0;
  const double gam[3] = {g1, g2, g3}, zet[3] = {0.0, z2, z3};
  const double sdt[3] = {g1 * dt, (g2 + z2) * dt, (g3 + z3) * dt};
  if (m->fast_path) {
    m->gn_alias_gm = false;
    for (int s = 0; s < 3; ++s) {
      int rc = fused_substep(m, dt, gam[s], zet[s], s > 0, sdt[s], s < 2);
      if (rc) return rc;
      m->time += sdt[s];
      if (s < 2) m->stage += 1;
    }
    m->iteration += 1;
    m->stage = 1;
    return OCN_OK;
  }
  m->gn_alias_gm = false;
  for (int s = 0; s < 3; ++s) {
    tendencies_and_step(m, dt, gam[s], zet[s], s > 0);
    int rc = pressure_correction(m, sdt[s]);
    if (rc) return rc;
    launch_pcorrect(m, sdt[s]);
    m->time += sdt[s];
    if (s < 2) {
      m->stage += 1;
      store_by_swap(m);
    } else {
      m->iteration += 1;
      m->stage = 1;
    }
    if ((rc = update_state(m))) return rc;
  }
  return OCN_OK;
}

// ---- whole-step hipGraphs ----------------------------------------------------------------------------------------
// A small model's step is a train of ~40-60 short launches and the host cannot issue them as fast as the GPU retires them
// (config 1, 16^3: 0.15 ms / step, all of it launch latency).  The general path therefore records the launches of one whole
// step into a hipGraph the SECOND time it meets the same (dt, stepper state, buffer rotation) and replays that graph from
// then on; the host-side part of the step (clock, G^n / G^- rotation, flags) runs again with every launch skipped
// (compat.h g_ocn_dry).  The first meeting of a key always runs launch by launch, so lazily built tables exist before a
// capture.  A capture that fails (or meets a call that must not be captured) restores the stepper state, steps normally
// and switches graphs off for the model.  Not used by the all-in-one periodic path (7 long launches), by slab models
// (the exchange is host-driven), under the phase profiler, or in the host emulation.
struct StepState {
  Field u, v, w, us, vs, ws, tr[OCN_MAX_TRACERS], trs[OCN_MAX_TRACERS], Gn[OCN_NF], Gm[OCN_NF];
  double time, previous_dt;
  int64_t iteration;
  int stage;
  bool gn_alias_gm, pred_active;
};
static void step_state_save(const ocn_model* m, StepState& s) {
  s.u = m->u; s.v = m->v; s.w = m->w; s.us = m->us; s.vs = m->vs; s.ws = m->ws;
  for (int t = 0; t < OCN_MAX_TRACERS; ++t) { s.tr[t] = m->tr[t]; s.trs[t] = m->trs[t]; }
  for (int f = 0; f < OCN_NF; ++f) { s.Gn[f] = m->Gn[f]; s.Gm[f] = m->Gm[f]; }
  s.time = m->time; s.previous_dt = m->previous_dt; s.iteration = m->iteration; s.stage = m->stage;
  s.gn_alias_gm = m->gn_alias_gm; s.pred_active = m->pred_active;
}
static void step_state_restore(ocn_model* m, const StepState& s) {
  m->u = s.u; m->v = s.v; m->w = s.w; m->us = s.us; m->vs = s.vs; m->ws = s.ws;
  for (int t = 0; t < OCN_MAX_TRACERS; ++t) { m->tr[t] = s.tr[t]; m->trs[t] = s.trs[t]; }
  for (int f = 0; f < OCN_NF; ++f) { m->Gn[f] = s.Gn[f]; m->Gm[f] = s.Gm[f]; }
  m->time = s.time; m->previous_dt = s.previous_dt; m->iteration = s.iteration; m->stage = s.stage;
  m->gn_alias_gm = s.gn_alias_gm; m->pred_active = s.pred_active;
}

static int step_plain(ocn_model* m, double dt, int force_euler) {
  if (m->d.stepper == OCN_STEPPER_AB2) return time_step_ab2(m, dt, force_euler);
  return time_step_rk3(m, dt);
}

static bool graph_eligible(const ocn_model* m) {
#ifdef OCN_HOST_EMU
  (void)m;
  return false;
#else
  // the all-in-one periodic path is seven long launches at 256^3 (nothing to gain), but below ~96^3 its step is launch bound
  // too (64^3: 0.16 of the step roofline in round 2): small boxes replay from a graph like the general path
  const bool small_box = (long)m->gd.Nx * m->gd.Ny * m->gd.Nz <= 96L * 96L * 96L;
  return m->knob_graph && !m->graph_off && (!m->fast_path || small_box) && !m->ctx->profiling && m->ctx->nranks == 1 &&
         !m->g->dist && !m->g->dist_y;
#endif
}

#ifndef OCN_HOST_EMU
static uint64_t step_key(const ocn_model* m, double dt, int force_euler, std::vector<uint64_t>& words) {
  words.clear();
  uint64_t b;
  memcpy(&b, &dt, 8);
  words.push_back(b);
  words.push_back(m->d.stepper == OCN_STEPPER_AB2 && (force_euler || dt != m->previous_dt));
  words.push_back(m->iteration == 0);
  words.push_back(m->gn_alias_gm);
  words.push_back(m->pred_active);
  auto fp = [&](const Field& f) { words.push_back((uint64_t)(uintptr_t)f.d); };
  fp(m->u); fp(m->v); fp(m->w); fp(m->us); fp(m->vs); fp(m->ws);
  for (int t = 0; t < m->nt; ++t) { fp(m->tr[t]); fp(m->trs[t]); }
  for (int f = 0; f < 3 + m->nt; ++f) { fp(m->Gn[f]); fp(m->Gm[f]); }
  uint64_t h = 1469598103934665603ull;
  for (uint64_t v : words) h = (h ^ v) * 1099511628211ull;
  return h;
}

static int step_graphed(ocn_model* m, double dt, int force_euler) {
  std::vector<uint64_t> words;
  const uint64_t key = step_key(m, dt, force_euler, words);
  int at = -1;
  for (size_t i = 0; i < m->graphs.size(); ++i)
    if (m->graphs[i].key == key && m->graphs[i].words == words) at = (int)i;   // the tuple itself, not only its hash
  if (at < 0) {
    if (m->graphs.size() >= 16) {   // an adaptive dt makes a new key every step: forget the ones that never came back
      std::vector<ocn_model::StepGraph> keep;
      for (auto& e : m->graphs)
        if (e.exec) keep.push_back(e);
      m->graphs.swap(keep);
    }
    if (m->graphs.size() < 16) m->graphs.push_back({key, 1, nullptr, words});
    return step_plain(m, dt, force_euler);
  }
  hipStream_t st = m->ctx->stream;
  if (m->graphs[at].exec) {
    OCN_HIP_CHECK(m->ctx, hipGraphLaunch((hipGraphExec_t)m->graphs[at].exec, st));
    g_ocn_dry = 1;
    int rc = step_plain(m, dt, force_euler);
    g_ocn_dry = 0;
    m->graph_replays += 1;
    return rc;
  }
  StepState saved;
  step_state_save(m, saved);
  auto give_up = [&](const char* why) {
    (void)hipGetLastError();
    step_state_restore(m, saved);
    m->graph_off = true;
    if (getenv("OCNHIP_DEBUG")) fprintf(stderr, "[ocnhip] step graphs off: %s\n", why);
    return step_plain(m, dt, force_euler);
  };
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) return give_up("hipStreamBeginCapture");
  g_ocn_capturing = 1;
  g_ocn_capture_poison = 0;
  int rc = step_plain(m, dt, force_euler);
  g_ocn_capturing = 0;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != OCN_OK || e != hipSuccess || !graph || g_ocn_capture_poison) {
    if (graph) hipGraphDestroy(graph);
    return give_up(g_ocn_capture_poison ? "a call that cannot be captured" : "capture failed");
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess || !exec) return give_up("hipGraphInstantiate");
  if (hipGraphLaunch(exec, st) != hipSuccess) {
    hipGraphExecDestroy(exec);
    return give_up("hipGraphLaunch");
  }
  m->graphs[at].exec = exec;
  m->graph_replays += 1;
  return OCN_OK;
}
#endif

extern "C" {

int ocn_model_create(ocn_grid* g, const ocn_model_desc* desc, ocn_model** out) {
  if (!g || !desc || !out) return OCN_EINVAL;
  ocn_ctx* ctx = g->ctx;
  if (desc->n_tracers < 0 || desc->n_tracers > OCN_MAX_TRACERS) return OCN_EINVAL;
  if (desc->advection < OCN_ADV_NONE || desc->advection > OCN_ADV_U3) return OCN_EINVAL;
  if (desc->closure == OCN_CLOSURE_AMD && (g->topo[0] == OCN_FLAT || g->topo[1] == OCN_FLAT || g->topo[2] == OCN_FLAT)) {
    ocn_set_error(ctx, "AnisotropicMinimumDissipation on a grid with a Flat direction is outside the path");
    return OCN_EUNSUPPORTED;
  }
  if (desc->closure == OCN_CLOSURE_AMD && desc->amd_has_Cb) {
    auto ok = [&](int i) { return i >= 0 && i < desc->n_tracers; };
    if ((desc->buoyancy == OCN_BUOYANCY_TRACER && !ok(desc->b_index)) ||
        (desc->buoyancy == OCN_BUOYANCY_LINEAR_TS && !(ok(desc->T_index) && ok(desc->S_index)))) {
      ocn_set_error(ctx, "AMD buoyancy modification needs the buoyancy model's tracers");
      return OCN_EINVAL;
    }
  }
  // halo inflation (nonhydrostatic_model.jl:140-148; Advection.jl:40)
  static const int buffer[8] = {0, 0, 1, 2, 2, 2, 1, 1};   // boundary_buffer of NONE, C2, C4, U5, WENO5 (Z, JS), U1, U3
  // The reference builds a NEW grid with the wider halo (with_halo); so does this: the caller's grid, and any
  // model already living on it, keep their halos, spacing tables and layouts.
  int need = buffer[desc->advection] + 1;
  bool changed = false;
  for (int d = 0; d < 3; ++d)
    if (g->topo[d] != OCN_FLAT && g->H[d] < need) changed = true;
  ocn_grid* own = nullptr;
  if (changed) {
    own = new ocn_grid(*g);
    own->d_dzc = own->d_dzf = own->d_rdzc = own->d_rdzf = nullptr;   // grid_build_dev allocates the copy's own tables
    for (int d = 0; d < 3; ++d)
      if (own->topo[d] != OCN_FLAT && own->H[d] < need) own->H[d] = need;
    int rc = grid_build_dev(own);
    if (rc) {
      ocn_grid_destroy(own);
      return rc;
    }
    g = own;
  }
  ocn_model* m = new ocn_model;
  m->own_grid = own;
  m->g = g;
  m->ctx = ctx;
  m->d = *desc;
  m->nt = desc->n_tracers;
  m->gd = g->dev;
  m->gd.nb = buffer[desc->advection];
  int rc = 0;
  rc |= field_alloc(m, m->u, OCN_FACE, OCN_CENTER, OCN_CENTER);
  rc |= field_alloc(m, m->v, OCN_CENTER, OCN_FACE, OCN_CENTER);
  rc |= field_alloc(m, m->w, OCN_CENTER, OCN_CENTER, OCN_FACE);
  rc |= field_alloc(m, m->pNHS, OCN_CENTER, OCN_CENTER, OCN_CENTER);
  if (g->topo[2] != OCN_FLAT) rc |= field_alloc(m, m->pHY, OCN_CENTER, OCN_CENTER, OCN_CENTER);
  for (int t = 0; t < m->nt; ++t) rc |= field_alloc(m, m->tr[t], OCN_CENTER, OCN_CENTER, OCN_CENTER);
  if (desc->closure == OCN_CLOSURE_AMD) {
    // DiffusivityFields(grid, tracers, bcs, ::AMD): nu_e and kappa_e per tracer, CenterFields with default BCs
    rc |= field_alloc(m, m->nu_e, OCN_CENTER, OCN_CENTER, OCN_CENTER);
    for (int t = 0; t < m->nt; ++t) rc |= field_alloc(m, m->kappa_e[t], OCN_CENTER, OCN_CENTER, OCN_CENTER);
  }
  for (int f = 0; f < 3 + m->nt; ++f) {
    int lx = f == 0, ly = f == 1, lz = f == 2;
    rc |= field_alloc(m, m->Gn[f], lx, ly, lz);
    rc |= field_alloc(m, m->Gm[f], lx, ly, lz);
  }
  if (rc) {
    ocn_model_destroy(m);
    return OCN_ENOMEM;
  }
  default_bcs(m, m->u, false);
  default_bcs(m, m->v, false);
  default_bcs(m, m->w, false);
  default_bcs(m, m->pNHS, true);
  if (m->pHY.present) default_bcs(m, m->pHY, true);
  for (int t = 0; t < m->nt; ++t) default_bcs(m, m->tr[t], false);
  if (m->nu_e.present) {
    default_bcs(m, m->nu_e, true);
    for (int t = 0; t < m->nt; ++t) default_bcs(m, m->kappa_e[t], true);
  }
  // user boundary conditions
  // u, v, w, tracers, then -- AnisotropicMinimumDissipation only -- the diffusivity fields nu_e and kappa_e of every tracer
  // (boundary_conditions = (; kappa_e = (; b = ...)) of the reference: nonhydrostatic_model.jl:150-160, test_boundary_conditions_integration.jl:52-66)
  const int nbc = 3 + m->nt + (desc->closure == OCN_CLOSURE_AMD ? 1 + m->nt : 0);
  for (int f = 0; f < nbc; ++f) {
    const int fd = f - (3 + m->nt);   // index among the diffusivity fields
    Field* fld = f == 0 ? &m->u : f == 1 ? &m->v : f == 2 ? &m->w : f < 3 + m->nt ? &m->tr[f - 3] : fd == 0 ? &m->nu_e : &m->kappa_e[fd - 1];
    for (int s = 0; s < 6; ++s) {
      const ocn_bc& b = f < 3 + m->nt ? desc->bcs[f][s] : fd == 0 ? desc->nu_bcs[s] : desc->kappa_bcs[fd - 1][s];
      if (b.kind == OCN_BC_DEFAULT) continue;
      int dim = s / 2;
      if (g->topo[dim] != OCN_BOUNDED) {
        if (b.kind == OCN_BC_PERIODIC || b.kind == OCN_BC_NONE) continue;
        ocn_set_error(ctx, "non-periodic boundary condition on a non-Bounded side (field %d side %d)", f, s);
        ocn_model_destroy(m);
        return OCN_EINVAL;
      }
      BCdev bd{b.kind, b.value, nullptr};
      if (b.array) {
        // arrays span the two other directions' interior sizes (boundary_condition.jl getbc for arrays)
        size_t nn = dim == 0 ? (size_t)g->N[1] * g->N[2] : dim == 1 ? (size_t)g->N[0] * g->N[2] : (size_t)g->N[0] * g->N[1];
        double* dptr = nullptr;
        if (hipMalloc((void**)&dptr, nn * sizeof(double)) != hipSuccess) {
          ocn_model_destroy(m);
          return OCN_ENOMEM;
        }
        hipMemcpy(dptr, b.array, nn * sizeof(double), hipMemcpyHostToDevice);
        m->owned.push_back(dptr);
        bd.arr = dptr;
      }
      fld->bc[s] = bd;
    }
  }
  if (g->dist && hipMalloc((void**)&m->phi_below, (size_t)g->N[0] * g->N[1] * sizeof(double)) != hipSuccess) {
    ocn_model_destroy(m);
    return OCN_ENOMEM;
  }
  if (desc->closure == OCN_CLOSURE_AMD && amd_build_table(m) != OCN_OK) {
    ocn_model_destroy(m);
    return OCN_ENOMEM;
  }
  if (hipMalloc((void**)&m->d_red, 64) != hipSuccess) {
    ocn_model_destroy(m);
    return OCN_ENOMEM;
  }
  fused_read_knobs(m);
  m->fast_path = fused_available(m) ? 1 : 0;
  m->bz_fast = (!m->fast_path && fused_bz_available(m)) ? 1 : 0;
  if (getenv("OCNHIP_DEBUG")) {
    char why[256];
    fused_describe(m, why, sizeof(why));
    fprintf(stderr, "[ocnhip] model: %s\n", why);
  }
  if (m->fast_path || m->bz_fast) {
    int r2 = field_alloc(m, m->us, OCN_FACE, OCN_CENTER, OCN_CENTER) | field_alloc(m, m->vs, OCN_CENTER, OCN_FACE, OCN_CENTER) |
             field_alloc(m, m->ws, OCN_CENTER, OCN_CENTER, OCN_FACE);
    for (int t = 0; t < m->nt; ++t) r2 |= field_alloc(m, m->trs[t], OCN_CENTER, OCN_CENTER, OCN_CENTER);
    if (r2) {
      ocn_model_destroy(m);
      return OCN_ENOMEM;
    }
    for (int s = 0; s < 6; ++s) {   // the predictor is filled with the velocities' boundary conditions
      m->us.bc[s] = m->u.bc[s];
      m->vs.bc[s] = m->v.bc[s];
      m->ws.bc[s] = m->w.bc[s];
    }
  }
  m->solver = poisson_create(m);
  if (!m->solver) {
    ocn_model_destroy(m);
    return OCN_EHIP;
  }
  update_state(m);  // nonhydrostatic_model.jl:200
  ctx->models.push_back(m);
  *out = m;
  return OCN_OK;
}

void ocn_model_destroy(ocn_model* m) {
  if (!m) return;
  if (m->ctx->overlap_ready) hipStreamSynchronize(m->ctx->comm_stream);
  hipStreamSynchronize(m->ctx->stream);
  for (auto it = m->ctx->models.begin(); it != m->ctx->models.end(); ++it)
    if (*it == m) {
      m->ctx->models.erase(it);
      break;
    }
#ifndef OCN_HOST_EMU
  for (auto& e : m->graphs)
    if (e.exec) hipGraphExecDestroy((hipGraphExec_t)e.exec);
#endif
  Field* all[] = {&m->u, &m->v, &m->w, &m->pHY, &m->pNHS, &m->nu_e, &m->us, &m->vs, &m->ws};
  for (Field* f : all) hipFree(f->d);
  for (int t = 0; t < OCN_MAX_TRACERS; ++t) {
    hipFree(m->trs[t].d);
    hipFree(m->tr[t].d);
    hipFree(m->kappa_e[t].d);
  }
  for (int f = 0; f < OCN_NF; ++f) {
    hipFree(m->Gn[f].d);
    hipFree(m->Gm[f].d);
  }
  for (double* p : m->owned) hipFree(p);
  hipFree(m->d_red);
  hipFree(m->phi_below);
  hipFree(m->amd_tab);
  hipFree(m->ypack_s);
  hipFree(m->ypack_r);
  poisson_destroy(m->solver);
  ocn_grid_destroy(m->own_grid);
  delete m;
}

int ocn_model_path(const ocn_model* m, char* buf, size_t n) {
  if (!m || !buf || n == 0) return OCN_EINVAL;
  fused_describe(m, buf, n);
  return OCN_OK;
}

int ocn_model_graph_replays(const ocn_model* m, int64_t* replays, int32_t* active) {
  if (!m) return OCN_EINVAL;
  if (replays) *replays = m->graph_replays;
  if (active) *active = graph_eligible(m) ? 1 : 0;
  return OCN_OK;
}

int ocn_model_halo(const ocn_model* m, int32_t H[3]) {
  if (!m || !H) return OCN_EINVAL;
  for (int d = 0; d < 3; ++d) H[d] = m->g->H[d];
  return OCN_OK;
}

int ocn_field_shape(const ocn_model* m, int field_id, int32_t total[3], int32_t interior[3], int32_t halo[3]) {
  Field* f = model_field(const_cast<ocn_model*>(m), field_id);
  if (!f) return OCN_EINVAL;
  for (int d = 0; d < 3; ++d) {
    if (total) total[d] = f->T[d];
    if (halo) halo[d] = m->g->H[d];
    if (interior)
      interior[d] = m->g->N[d] + ((f->loc[d] == OCN_FACE && m->g->topo[d] == OCN_BOUNDED) ? 1 : 0);
  }
  return OCN_OK;
}

void* ocn_field_device_ptr(ocn_model* m, int field_id) {
  if (m) halo_settle(m);   // work the caller enqueues on ocn_stream() after this call sees complete halos
  Field* f = model_field(m, field_id);
  return f ? f->d : nullptr;
}

// logical parent (column-major T[0] x T[1] x T[2]) <-> physical array.  A Flat x / y direction has one logical
// index, stored in every physical slot of that direction.
static void host_scatter(const Field& f, const double* logical, const int lo[3], const int ext[3], std::vector<double>& phys) {
  // writes the logical box [lo, lo+ext) (box-local column-major input) into the physical buffer
  for (int c = 0; c < ext[2]; ++c)
    for (int b = 0; b < ext[1]; ++b)
      for (int a = 0; a < ext[0]; ++a) {
        const double val = logical[a + (size_t)ext[0] * (b + (size_t)ext[1] * c)];
        const int x0 = f.off[0] ? 0 : lo[0] + a, x1 = f.off[0] ? f.P[0] : x0 + 1;
        const int y0 = f.off[1] ? 0 : lo[1] + b, y1 = f.off[1] ? f.P[1] : y0 + 1;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x) phys[x + (size_t)y * f.sy + (size_t)(lo[2] + c) * f.sz] = val;
      }
}
static void host_gather(const Field& f, const std::vector<double>& phys, const int lo[3], const int ext[3], double* logical) {
  for (int c = 0; c < ext[2]; ++c)
    for (int b = 0; b < ext[1]; ++b)
      for (int a = 0; a < ext[0]; ++a)
        logical[a + (size_t)ext[0] * (b + (size_t)ext[1] * c)] =
            phys[(f.off[0] + lo[0] + a) + (size_t)(f.off[1] + lo[1] + b) * f.sy + (size_t)(lo[2] + c) * f.sz];
}

int ocn_field_layout(const ocn_model* m, int field_id, int64_t strides[3], int64_t* origin) {
  Field* f = model_field(const_cast<ocn_model*>(m), field_id);
  if (!f) return OCN_EINVAL;
  if (strides) {
    strides[0] = 1;
    strides[1] = f->sy;
    strides[2] = f->sz;
  }
  if (origin) *origin = f->off[0] + (int64_t)f->off[1] * f->sy;
  return OCN_OK;
}

static int parent_upload(ocn_ctx* ctx, Field* f, const double* host) {
  OCN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (field_dense(*f)) {
    OCN_HIP_CHECK(ctx, hipMemcpy(f->d, host, f->n * sizeof(double), hipMemcpyHostToDevice));
    return OCN_OK;
  }
  std::vector<double> phys(f->n);
  OCN_HIP_CHECK(ctx, hipMemcpy(phys.data(), f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
  const int lo[3] = {0, 0, 0};
  host_scatter(*f, host, lo, f->T, phys);
  OCN_HIP_CHECK(ctx, hipMemcpy(f->d, phys.data(), f->n * sizeof(double), hipMemcpyHostToDevice));
  return OCN_OK;
}
static int parent_download(ocn_ctx* ctx, const Field* f, double* host) {
  OCN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (field_dense(*f)) {
    OCN_HIP_CHECK(ctx, hipMemcpy(host, f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
    return OCN_OK;
  }
  std::vector<double> phys(f->n);
  OCN_HIP_CHECK(ctx, hipMemcpy(phys.data(), f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
  const int lo[3] = {0, 0, 0};
  host_gather(*f, phys, lo, f->T, host);
  return OCN_OK;
}

int ocn_field_upload(ocn_model* m, int field_id, const double* host) {
  if (m && halo_settle(m)) return OCN_EHIP;
  if (m && field_id >= OCN_F_GN && field_id < OCN_F_GN + OCN_NF && materialize_gn(m)) return OCN_EHIP;
  Field* f = model_field(m, field_id);
  if (!f || !host) return OCN_EINVAL;
  return parent_upload(m->ctx, f, host);
}

int ocn_field_download(const ocn_model* m, int field_id, double* host) {
  if (m && halo_settle(const_cast<ocn_model*>(m))) return OCN_EHIP;
  if (m && field_id == OCN_F_PNHS)
    if (int rc = pnhs_refresh(const_cast<ocn_model*>(m))) return rc;
  Field* f = model_field(const_cast<ocn_model*>(m), field_id);
  if (!f || !host) return OCN_EINVAL;
  return parent_download(m->ctx, f, host);
}

// ---- stand-alone fields: Field{LX, LY, LZ}(grid) / zeros(FT, arch, N...) of the reference (Fields/field.jl:16-30,
// Grids/new_data.jl:16-61, Grids/zeros.jl:7): a zero-filled parent array, halos included, laid out exactly as the
// model's own fields on that grid.  The shim's arch_array / CenterField() allocate through these.
struct ocn_field {
  ocn_grid* g;
  Field f;
};

int ocn_field_create(ocn_grid* g, int locx, int locy, int locz, ocn_field** out) {
  if (!g || !out) return OCN_EINVAL;
  const int loc[3] = {locx, locy, locz};
  for (int d = 0; d < 3; ++d)
    if (loc[d] != OCN_CENTER && loc[d] != OCN_FACE) {
      ocn_set_error(g->ctx, "ocn_field_create: location %d along dimension %d is neither OCN_CENTER nor OCN_FACE", loc[d], d);
      return OCN_EINVAL;
    }
  ocn_field* h = new ocn_field;
  h->g = g;
  int rc = field_alloc(g->ctx, g, h->f, locx, locy, locz);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return OCN_OK;
}

void ocn_field_destroy(ocn_field* f) {
  if (!f) return;
  hipStreamSynchronize(f->g->ctx->stream);
  hipFree(f->f.d);
  delete f;
}

int ocn_field_parent_shape(const ocn_field* f, int32_t total[3], int32_t interior[3], int32_t halo[3]) {
  if (!f) return OCN_EINVAL;
  for (int d = 0; d < 3; ++d) {
    if (total) total[d] = f->f.T[d];
    if (halo) halo[d] = f->g->H[d];
    if (interior) interior[d] = f->g->N[d] + ((f->f.loc[d] == OCN_FACE && f->g->topo[d] == OCN_BOUNDED) ? 1 : 0);
  }
  return OCN_OK;
}

int ocn_field_parent_layout(const ocn_field* f, int64_t strides[3], int64_t* origin) {
  if (!f) return OCN_EINVAL;
  if (strides) {
    strides[0] = 1;
    strides[1] = f->f.sy;
    strides[2] = f->f.sz;
  }
  if (origin) *origin = f->f.off[0] + (int64_t)f->f.off[1] * f->f.sy;
  return OCN_OK;
}

void* ocn_field_parent_ptr(ocn_field* f) { return f ? f->f.d : nullptr; }

int ocn_field_parent_upload(ocn_field* f, const double* host) {
  if (!f || !host) return OCN_EINVAL;
  return parent_upload(f->g->ctx, &f->f, host);
}

int ocn_field_parent_download(const ocn_field* f, double* host) {
  if (!f || !host) return OCN_EINVAL;
  return parent_download(f->g->ctx, &f->f, host);
}

int ocn_field_set_interior(ocn_model* m, int field_id, const double* host) {
  if (m && halo_settle(m)) return OCN_EHIP;
  if (m && field_id >= OCN_F_GN && field_id < OCN_F_GN + OCN_NF && materialize_gn(m)) return OCN_EHIP;
  Field* f = model_field(m, field_id);
  if (!f || !host) return OCN_EINVAL;
  int32_t it[3];
  ocn_field_shape(m, field_id, nullptr, it, nullptr);
  OCN_HIP_CHECK(m->ctx, hipStreamSynchronize(m->ctx->stream));
  std::vector<double> phys(f->n);
  OCN_HIP_CHECK(m->ctx, hipMemcpy(phys.data(), f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
  const int lo[3] = {m->g->H[0], m->g->H[1], m->g->H[2]}, ext[3] = {it[0], it[1], it[2]};
  host_scatter(*f, host, lo, ext, phys);
  OCN_HIP_CHECK(m->ctx, hipMemcpy(f->d, phys.data(), f->n * sizeof(double), hipMemcpyHostToDevice));
  return OCN_OK;
}

int ocn_field_get_interior(const ocn_model* m, int field_id, double* host) {
  if (m && halo_settle(const_cast<ocn_model*>(m))) return OCN_EHIP;
  Field* f = model_field(const_cast<ocn_model*>(m), field_id);
  if (!f || !host) return OCN_EINVAL;
  int32_t it[3];
  ocn_field_shape(m, field_id, nullptr, it, nullptr);
  OCN_HIP_CHECK(m->ctx, hipStreamSynchronize(m->ctx->stream));
  std::vector<double> phys(f->n);
  OCN_HIP_CHECK(m->ctx, hipMemcpy(phys.data(), f->d, f->n * sizeof(double), hipMemcpyDeviceToHost));
  const int lo[3] = {m->g->H[0], m->g->H[1], m->g->H[2]}, ext[3] = {it[0], it[1], it[2]};
  host_gather(*f, phys, lo, ext, host);
  return OCN_OK;
}

int ocn_fill_halos(ocn_model* m, uint32_t mask) {
  if (!m) return OCN_EINVAL;
  Field* fs[OCN_NF + 2];
  int n = 0;
  const int ids[5] = {OCN_F_U, OCN_F_V, OCN_F_W, OCN_F_PHY, OCN_F_PNHS};
  for (int b = 0; b < 5; ++b)
    if (mask & (1u << b)) {
      Field* f = model_field(m, ids[b]);
      if (f) fs[n++] = f;
    }
  for (int t = 0; t < m->nt; ++t)
    if (mask & (1u << (8 + t))) fs[n++] = &m->tr[t];
  // fields of one call share one batched launch; aux fields (pressures) use their own z conditions
  if (mask & (1u << 4)) m->pnhs_halo_stale = false;   // filled right here
  return api_ret(m->ctx, fill_fields(m, fs, n));
}

int ocn_update_state(ocn_model* m) { return m ? api_ret(m->ctx, update_state(m)) : OCN_EINVAL; }

int ocn_compute_tendencies(ocn_model* m) {
  if (!m) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  m->gn_alias_gm = false;
  launch_tendencies(m);
  return api_ret(m->ctx, OCN_OK);
}

int ocn_ab2_step(ocn_model* m, double dt, double chi) {
  if (!m) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  if (materialize_gn(m)) return OCN_EHIP;
  launch_step(m, dt, 1.5 + chi, -(0.5 + chi), 1);
  return api_ret(m->ctx, OCN_OK);
}

int ocn_rk3_substep(ocn_model* m, double dt, double gamma, double zeta, int has_zeta) {
  if (!m) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  if (materialize_gn(m)) return OCN_EHIP;
  launch_step(m, dt, gamma, zeta, has_zeta);
  return api_ret(m->ctx, OCN_OK);
}

int ocn_store_tendencies(ocn_model* m) {
  if (!m) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  if (materialize_gn(m)) return OCN_EHIP;
  launch_store(m);
  return api_ret(m->ctx, OCN_OK);
}

int ocn_pressure_correction(ocn_model* m, double dt) { return m ? api_ret(m->ctx, pressure_correction(m, dt)) : OCN_EINVAL; }

int ocn_pressure_correct_velocities(ocn_model* m, double dt) {
  if (!m) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  launch_pcorrect(m, dt);
  return api_ret(m->ctx, OCN_OK);
}

int ocn_poisson_solve_host(ocn_model* m, const double* rhs, double* phi) {
  if (!m || !rhs || !phi) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  size_t n = (size_t)m->g->N[0] * m->g->N[1] * m->g->N[2];
  double *a = nullptr, *b = nullptr;
  OCN_HIP_CHECK(m->ctx, hipMalloc((void**)&a, n * sizeof(double)));
  OCN_HIP_CHECK(m->ctx, hipMalloc((void**)&b, n * sizeof(double)));
  hipMemcpy(a, rhs, n * sizeof(double), hipMemcpyHostToDevice);
  int rc = poisson_solve_rhs(m, a, b);
  hipStreamSynchronize(m->ctx->stream);
  if (!rc) hipMemcpy(phi, b, n * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(a);
  hipFree(b);
  return api_ret(m->ctx, rc);
}

int ocn_set_epilogue(ocn_model* m, int enforce_incompressibility) {
  // set_nonhydrostatic_model.jl:45-58
  if (!m) return OCN_EINVAL;
  int rc = update_state(m);
  if (rc) return rc;
  if (enforce_incompressibility) {
    if ((rc = pressure_correction(m, 1.0))) return rc;
    launch_pcorrect(m, 1.0);
    if ((rc = update_state(m))) return rc;
  }
  return api_ret(m->ctx, OCN_OK);
}

int ocn_time_step(ocn_model* m, double dt, int force_euler) {
  if (!m) return OCN_EINVAL;
  ProfScope ps(m->ctx, "time_step");
  if (!m->fast_path && halo_settle(m)) return OCN_EHIP;   // the all-in-one path overlaps its own exchanges (fused_substep)
#ifndef OCN_HOST_EMU
  if (graph_eligible(m)) return api_ret(m->ctx, step_graphed(m, dt, force_euler));
#endif
  return api_ret(m->ctx, step_plain(m, dt, force_euler));
}

int ocn_clock(const ocn_model* m, double* time, int64_t* iteration, int32_t* stage) {
  if (!m) return OCN_EINVAL;
  if (time) *time = m->time;
  if (iteration) *iteration = m->iteration;
  if (stage) *stage = m->stage;
  return OCN_OK;
}

int ocn_set_clock(ocn_model* m, double time, int64_t iteration, double previous_dt) {
  if (!m) return OCN_EINVAL;
  m->time = time;
  m->iteration = iteration;
  m->previous_dt = previous_dt;
  return OCN_OK;
}

int ocn_max_abs_divergence(ocn_model* m, double* out) {
  if (!m || !out) return OCN_EINVAL;
  if (halo_settle(m)) return OCN_EHIP;
  OCN_HIP_CHECK(m->ctx, hipMemsetAsync(m->d_red, 0, 8, m->ctx->stream));
  launch_maxdiv(m, m->d_red);
  if (int rc = api_ret(m->ctx, OCN_OK)) return rc;
  OCN_HIP_CHECK(m->ctx, hipStreamSynchronize(m->ctx->stream));
  OCN_HIP_CHECK(m->ctx, hipMemcpy(out, m->d_red, 8, hipMemcpyDeviceToHost));
  return OCN_OK;
}

int ocn_measure_copy_rate(ocn_ctx* ctx, size_t bytes, int reps, double* bytes_per_s) {
  if (!ctx || !bytes_per_s || bytes == 0 || reps < 1) return OCN_EINVAL;
  char *a = nullptr, *b = nullptr;
  OCN_HIP_CHECK(ctx, hipMalloc((void**)&a, bytes));
  if (hipMalloc((void**)&b, bytes) != hipSuccess) {
    hipFree(a);
    ocn_set_error(ctx, "ocn_measure_copy_rate: allocation of %zu bytes failed", bytes);
    return OCN_ENOMEM;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  OCN_ASYNC(hipMemsetAsync(a, 1, bytes, ctx->stream));
  OCN_ASYNC(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, ctx->stream));   // warm-up
  hipEventRecord(e0, ctx->stream);
  for (int r = 0; r < reps; ++r) OCN_ASYNC(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  hipEventRecord(e1, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(a);
  hipFree(b);
  if (e != hipSuccess || !(ms > 0) || api_ret(ctx, OCN_OK) != OCN_OK) {
    ocn_set_error(ctx, "ocn_measure_copy_rate: copy failed");
    return OCN_EHIP;
  }
  *bytes_per_s = 2.0 * (double)bytes * reps / (ms * 1e-3);
  return OCN_OK;
}

// ---- multi-GPU: ocn_comm_unique_id / ocn_comm_init live in comm.hip -----------------------------------------------
int ocn_comm_rank(const ocn_ctx* ctx, int* rank, int* nranks) {
  if (!ctx) return OCN_EINVAL;
  if (rank) *rank = ctx->rank;
  if (nranks) *nranks = ctx->nranks;
  return OCN_OK;
}

}  // extern "C"
