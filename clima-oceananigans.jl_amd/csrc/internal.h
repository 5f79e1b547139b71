// internal.h -- host-side objects behind the opaque handles of include/ocnhip.h
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/ocnhip.h"
#include "compat.h"
#include "stencils.h"

#define OCN_NF (3 + OCN_MAX_TRACERS)

struct ProfPhase {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double total_ms = 0;
  int64_t count = 0;
};

struct ocn_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  bool profiling = false;
  std::map<std::string, ProfPhase> prof;
  std::string prof_only;   // when set, only this phase records events (each record costs ~4 us of stream time)
  // multi-GPU
  int rank = 0, nranks = 1;
  void* comm = nullptr;  // ncclComm_t
  void* shm = nullptr;   // ShmWorld: host shared-memory transport (tests / rehearsals on one GPU or none), comm.hip
  // z-slab halo exchange overlapped with the interior levels of the next tendency kernel (api.hip fused_substep): the
  // exchange runs on its own stream between two events
  bool overlap_ready = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_main = nullptr, ev_halo = nullptr, ev_halo2 = nullptr;
  std::vector<struct ocn_model*> models;   // live models (ocn_sync settles their exchanges)
  int sticky_rc = 0;   // first failure of an exchange issued from a helper that cannot return it (splitexplicit.hip hfield_fill); reported by the entry point
};

void ocn_set_error(ocn_ctx* ctx, const char* fmt, ...);

struct CommOp {
  void* buf;
  size_t bytes;
  int peer;
  int tag;
};

struct ocn_grid {
  ocn_ctx* ctx;
  int Nzg = 0;          // global Nz (N[2] is this rank's slab)
  bool dist = false;    // z-slab decomposition active (also forced with one rank by OCNHIP_FORCE_DIST=1)
  int Nyg = 0;          // global Ny (N[1] is this rank's slab when dist_y)
  bool dist_y = false;  // y-slab decomposition (Bounded z: every rank keeps whole columns)
  ocn_grid_desc d;
  int N[3], H[3], topo[3];
  int PH[3];            // physical halo: H, except Flat x / y, which are stored with broadcast halos
  double L[3], x0[3];
  bool z_regular;
  std::vector<double> zF_int;            // interior faces (stretched)
  std::vector<double> h_dzc, h_dzf;      // host spacings incl. halos (stretched), layout as GridDev
  double* d_dzc = nullptr;
  double* d_dzf = nullptr;
  double *d_rdzc = nullptr, *d_rdzf = nullptr;
  GridDev dev;                           // for the current halo
};

struct BCdev {
  int kind;
  double value;
  const double* arr;  // device (Nx*Ny) or null
};

// Device layout of a field.  T is the reference's parent size (Grids/new_data.jl:16-61); P the size actually
// allocated.  They coincide unless x or y is Bounded or Flat: then every field of the model shares ONE pitch
// (as hipMallocPitch would give) -- Nx+2H+1 columns when x is Bounded, so Face- and Center-located fields have
// equal strides and one linear index addresses all of them -- and a Flat x / y direction is stored with
// broadcast halos (every difference across it is exactly 0, every interpolation the identity, as the
// reference's Flat operators).  ocn_field_layout() reports strides and origin; upload / download convert.
struct Field {
  double* d = nullptr;  // parent array on the device
  int T[3] = {0, 0, 0};
  int P[3] = {0, 0, 0};
  int off[3] = {0, 0, 0};   // physical index of the logical parent's first element
  int loc[3] = {0, 0, 0};
  size_t n = 0;             // physical element count
  BCdev bc[6];
  bool present = false;
  double* interior() const;  // pointer to the first interior cell
  long sy = 0, sz = 0;
  int Hx = 0, Hy = 0, Hz = 0;
};

struct PoissonSolver;

struct ocn_model {
  ocn_grid* g;
  ocn_ctx* ctx;
  ocn_model_desc d;
  GridDev gd;
  int nt;  // tracers
  Field u, v, w, pHY, pNHS;
  Field tr[OCN_MAX_TRACERS], Gn[OCN_NF], Gm[OCN_NF];
  Field nu_e, kappa_e[OCN_MAX_TRACERS];
  // projection scratch: predictor velocities are written here by the fused kernels
  Field us, vs, ws;
  Field trs[OCN_MAX_TRACERS];   // second tracer buffers of the fused path
  PoissonSolver* solver = nullptr;
  std::vector<double*> owned;  // extra device allocations (bc arrays)
  // clock / stepper state
  double time = 0, previous_dt = INFINITY;
  int64_t iteration = 0;
  int stage = 1;
  double* d_red = nullptr;  // reduction scratch
  int fast_path = 0;        // 1: fused periodic WENO kernels usable
  int bz_fast = 0;          // 1: Bounded z: tiled advection + update kernel on top of the general kernels' other terms
  double* amd_tab = nullptr;     // per-level factors of the AMD predictors (kernels.hip amd_build_table)
  double* phi_below = nullptr;   // (Nx,Ny): top plane of the lower neighbour's pressure (slab runs)
  double *ypack_s = nullptr, *ypack_r = nullptr;   // y-slab halo exchange staging (send / receive)
  size_t ypack_n = 0;
  bool gn_alias_gm = false; // after a fused AB2 step G^n and G^- are the same buffer (pointer swap instead of a copy)
  // Bounded-z tiled path: between the time-stepper update and the projection the predictor U* lives in us / vs / ws;
  // the projection writes u = U* - dt grad p back into u, v, w, so their device pointers never change.
  bool pred_active = false;
  ocn_grid* own_grid = nullptr;   // private copy of the caller's grid when the advection scheme needs wider halos
  int knob_fused_xt = 0, knob_no_dma = 0, knob_no_tracer3 = 0, knob_prio = 0x20FF;   // fused_read_knobs(), at creation
  // whole-step hipGraphs of the general path (api.hip step_graphed): one entry per distinct (dt, stepper state, buffer rotation)
  // whole-step hipGraph cache: `key` is a hash of `words` (dt bits, stepper flags, the device pointer of every rotating buffer);
  // a hit on the hash is only taken after the words themselves compare equal
  struct StepGraph { uint64_t key; int seen; void* exec; std::vector<uint64_t> words; };
  std::vector<StepGraph> graphs;
  int knob_overlap_cus = 16; // CUs the interior tendency launch leaves to the communication kernels (OCNHIP_OVERLAP_CUS)
  int knob_overlap = -1;     // OCNHIP_OVERLAP=0|1: z-slab halo exchange overlapped with the next tendency launch (default: with > 1 rank)
  bool halo_inflight = false, halo2_inflight = false;   // exchange of (u, v, w, tracers) started, not yet waited for (halo2: unused since round 3)
  // z-slab runs of the all-in-one path: nothing inside a time step reads the z halos of pNHS (the projection takes the one plane
  // it needs through fused_exchange_phi), so their exchange -- 2 x H planes per step -- is deferred until somebody can look:
  // ocn_sync, ocn_fill_halos(pNHS) and ocn_field_download(pNHS), all collective on slab runs (api.hip pnhs_refresh)
  bool pnhs_halo_stale = false;
  int knob_xfft_team = 0;    // OCNHIP_XFFT_TEAM=1: the fused rhs + x transform loads in team order (the older variant; tests)
  int knob_graph = 1;        // OCNHIP_NO_GRAPH=1 clears it (model creation)
  bool graph_off = false;    // a capture failed: this model steps launch by launch from then on
  int64_t graph_replays = 0;
};
extern thread_local int g_ocn_capturing, g_ocn_capture_poison;   // a library call that cannot be captured sets the poison flag
inline Field& pred_u(ocn_model* m) { return m->pred_active ? m->us : m->u; }
inline Field& pred_v(ocn_model* m) { return m->pred_active ? m->vs : m->v; }
inline Field& pred_w(ocn_model* m) { return m->pred_active ? m->ws : m->w; }

Field* model_field(ocn_model* m, int id);

// ---- kernels.hip ------------------------------------------------------------------------------------
struct FieldPtrs {   // up to velocities + both pressures + all tracers in one call (ocn_fill_halos)
  double* p[OCN_NF + 2];
  int Tx[OCN_NF + 2], Ty[OCN_NF + 2], Tz[OCN_NF + 2];   // extents the fill sweeps (the field's own parent extents)
  int n;
};
void launch_fill_periodic(ocn_model* m, const FieldPtrs& f, int dim);
void launch_fill_flat(ocn_model* m, const FieldPtrs& f, int dim);
bool launch_fill_periodic_xy(ocn_model* m, const FieldPtrs& f);   // false: not applicable, use the two passes
void launch_fill_bounded(ocn_model* m, Field** fs, int n, int dim);
void launch_tendencies(ocn_model* m, bool skip_momentum_advection = false, bool skip_tracer_advection = false);
void launch_step(ocn_model* m, double dt, double cn, double cm, int use_m, bool tracers_only = false);
void launch_store(ocn_model* m);
void launch_rhs(ocn_model* m, double dt, double* rhs, int mult_dz);
void launch_pcorrect(ocn_model* m, double dt);
void launch_hydrostatic(ocn_model* m);
void launch_copy_to_field(ocn_model* m, const double* src, Field& f);
void launch_maxdiv(ocn_model* m, double* out_dev);
void launch_amd(ocn_model* m);
int amd_build_table(ocn_model* m);

// ---- fused.hip -----------------------------------------------------------------------------------------
void fused_read_knobs(ocn_model* m);
void fused_describe(const ocn_model* m, char* buf, size_t n);
bool fused_available(const ocn_model* m);
bool fused_bz_available(const ocn_model* m);
void launch_fused_bz(ocn_model* m, double dt, double cn, double cm, int use_m);
bool launch_rest4(ocn_model* m);
bool fused_tracer3_ok(const ocn_model* m);
bool tracer_rest_shell(const ocn_model* m);
void launch_tracer3(ocn_model* m, double dt, double cn, double cm, int use_m, bool rest);
// levels [zlo, zhi) (-1: Nz), optionally followed by a second run [zlo2, zhi2) of the same length in the same launch
void launch_fused_tend_step(ocn_model* m, double dt, double cn, double cm, int use_m, int zlo = 0, int zhi = -1, int zlo2 = 0, int zhi2 = 0);
void launch_rhs_wrap(ocn_model* m, double dt, double* rhs);
void launch_project(ocn_model* m, double dt, const double* phi);
void launch_tracer_steps(ocn_model* m, double dt, double cn, double cm, int use_m);
int fused_exchange_ws(ocn_model* m);
int fused_exchange_phi(ocn_model* m, const double* phi);
int poisson_run(ocn_model* m);   // transforms + spectral solve on the solver's rhs buffer, in place

// ---- poisson.hip ------------------------------------------------------------------------------------
PoissonSolver* poisson_create(ocn_model* m);
void poisson_destroy(PoissonSolver* s);
int poisson_solve(ocn_model* m, double dt);             // rhs from velocities -> pNHS interior
int poisson_solve_rhs(ocn_model* m, const double* rhs_dev, double* phi_dev);  // generic (tests)
double* poisson_rhs_buffer(PoissonSolver* s);

// ---- zfft.hip ---------------------------------------------------------------------------------------------
void* zsolve_create(ocn_ctx* ctx, const std::vector<double>& lx_half, const std::vector<double>& ly_local);
void zsolve_destroy(void* z);
void zsolve_run(ocn_ctx* ctx, void* z, void* spec, int Nz, const double* lz, double norm, long zero_col);
void yfft_run(ocn_ctx* ctx, void* z, void* spec, int Nxh, int Ny, int Nz, int inverse);
void xfft_rhs_run(ocn_model* m, void* z, void* spec, double dt, int extra_plane = 0);
bool poisson_local_wstar(const ocn_model* m);   // z-slab runs: the w* term above the slab enters in spectral space, no w* plane exchange
bool fft_size_ok(int n);   // 128, 256, 512: sizes of the custom transform passes
bool zsolve_size_ok(int n);   // those and 320, 384: sizes of the fused z stage
bool poisson_custom_xy(const ocn_model* m);
int poisson_run_from_predictor(ocn_model* m, double dt);   // fused rhs + custom x/y passes (fast path)

// ---- zslab.hip --------------------------------------------------------------------------------------------
void* zslab_create(ocn_ctx* ctx, const std::vector<double>& lx_half, const std::vector<double>& ly, int n, int R, int rank);
void zslab_destroy(void* z);
int zslab_run(ocn_ctx* ctx, void* z, void* spec, double dz2, double scale, const void* below = nullptr, void* phi_below = nullptr);
bool poisson_local_phi_below(const ocn_model* m);
const double* poisson_phi_below(const ocn_model* m);   // that plane, (Nx, Ny), behind the solver's real buffer   // z-slab runs: the pressure plane below the slab is computed by this rank (no exchange)

// ---- comm.hip ---------------------------------------------------------------------------------------------
int comm_exchange(ocn_ctx* c, const std::vector<CommOp>& sends, const std::vector<CommOp>& recvs, hipStream_t st = nullptr);   // st: default the context's stream
int comm_halo_exchange_z(ocn_model* m, Field** fs, int n, hipStream_t st = nullptr);
bool comm_can_overlap(const ocn_ctx* c);   // the transport is stream-ordered on the device (RCCL, or self-copies of a forced one-rank slab run)
int halo_settle(ocn_model* m);             // the model's stream waits for an exchange still in flight (api.hip)
int comm_halo_exchange_y(ocn_model* m, Field** fs, int n);
int comm_alltoall(ocn_ctx* c, const void* send, void* recv, size_t block_bytes);
void comm_destroy(ocn_ctx* c);

// ---- profiling ----------------------------------------------------------------------------------------
struct ProfScope {
  ocn_ctx* c;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  ProfScope(ocn_ctx* ctx, const char* nm);
  ~ProfScope();
};
