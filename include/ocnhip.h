/*
 * ocnhip.h -- C ABI of libocnhip.so: an MI355X (gfx950) implementation of Oceananigans'
 * NonhydrostaticModel `time_step!` hot path on a RectilinearGrid.
 *
 * The reference (Oceananigans.jl v0.76.8) is pure Julia and has no FFI / plugin registry: its only
 * extension point is multiple dispatch on the architecture type carried by the grid
 * (src/Architectures.jl:53-142, src/Grids/grid_utils.jl:39).  A `ROCmGPU <: AbstractArchitecture`
 * shim overloads the phase-level functions listed beside each entry point below
 * and forwards them here with `ccall`.  Paths are relative to /root/reference/src.
 *
 * Conventions
 *   - every function returns 0 (OCN_OK) or a negative OCN_E* code; nothing throws across the boundary;
 *     the message of the last failure is available from ocn_last_error().
 *   - the library owns all device memory behind handles; host buffers belong to the caller and only
 *     need to stay alive for the duration of the (synchronous) upload / download call.
 *   - field memory is the reference's *parent* array: column-major (x fastest), halos included,
 *     size total_size(loc, grid) (Grids/new_data.jl:16-61, Grids/grid_utils.jl:105-128), so a Julia
 *     `OffsetArray` can alias ocn_field_device_ptr() with the usual offsets.  Exception: when x or y is
 *     Bounded or Flat the rows / planes are pitched (like hipMallocPitch: one pitch of N+2H+1 for Face- and
 *     Center-located fields; a Flat x / y direction is stored with broadcast copies).  ocn_field_layout()
 *     returns the element strides and the origin of the logical parent inside the allocation, so the
 *     alias becomes a strided view; ocn_field_upload / download always speak the dense parent layout.
 *   - pointer stability: ocn_field_device_ptr() of u, v, w, pHY', pNHS, nu_e and kappa_e never changes during the life
 *     of a model (every time-stepping path writes the corrected velocities back into the same arrays).  The
 *     tendency sets G^n / G^- and, on the tiled kernels' paths, the tracers are double buffered and ROTATE:
 *     re-query their pointers after every ocn_time_step (tests/test_model_contracts.py asserts both halves).
 *   - a model that needs wider halos than its grid has (WENO5 / U5: 3) works on a private copy of the grid with the
 *     wider halo, as the reference's with_halo does (nonhydrostatic_model.jl:140-148); the caller's grid and other
 *     models on it are left alone.
 *   - one context = one device + one in-order HIP stream.  Compute calls are asynchronous with
 *     respect to the host and ordered on that stream; only ocn_sync, uploads and downloads block.
 *   - handles are not thread-safe: one host thread per context.
 */
#ifndef OCNHIP_H
#define OCNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCN_ABI_VERSION 5

/* error codes */
enum {
  OCN_OK = 0,
  OCN_EINVAL = -1,       /* bad argument / unsupported configuration             */
  OCN_ENOMEM = -2,       /* device or host allocation failed                     */
  OCN_EHIP = -3,         /* a HIP / hipFFT / RCCL call failed                    */
  OCN_EUNSUPPORTED = -4, /* valid in the reference, outside this library's scope */
  OCN_ESTATE = -5        /* call made in the wrong state                         */
};

/* topology of one direction (Grids/Grids.jl:60-93) */
enum { OCN_PERIODIC = 0, OCN_BOUNDED = 1, OCN_FLAT = 2 };
/* location of a field along one direction (Grids/Grids.jl: Center, Face) */
enum { OCN_CENTER = 0, OCN_FACE = 1 };
/* advection schemes (Advection/: centered_second_order.jl, centered_fourth_order.jl,
 * upwind_biased_fifth_order.jl, weno_fifth_order.jl:162-180 with zweno = true / false) */
enum { OCN_ADV_NONE = 0, OCN_ADV_C2 = 1, OCN_ADV_C4 = 2, OCN_ADV_U5 = 3, OCN_ADV_WENO5_Z = 4, OCN_ADV_WENO5_JS = 5,
       OCN_ADV_U1 = 6, OCN_ADV_U3 = 7 };   /* upwind_biased_first_order.jl, upwind_biased_third_order.jl (boundary buffer 1) */
/* time steppers (TimeSteppers/quasi_adams_bashforth_2.jl, runge_kutta_3.jl) */
enum { OCN_STEPPER_AB2 = 0, OCN_STEPPER_RK3 = 1 };
/* closures (TurbulenceClosures/turbulence_closure_implementations/{scalar_diffusivity,anisotropic_minimum_dissipation}.jl) */
enum { OCN_CLOSURE_NONE = 0, OCN_CLOSURE_SCALAR = 1, OCN_CLOSURE_AMD = 2 };
/* buoyancy models (BuoyancyModels/buoyancy_tracer.jl:12, linear_equation_of_state.jl:69-77) */
enum { OCN_BUOYANCY_NONE = 0, OCN_BUOYANCY_TRACER = 1, OCN_BUOYANCY_LINEAR_TS = 2 };
/* boundary-condition kinds (BoundaryConditions/boundary_condition.jl:8-11,77-96) */
enum { OCN_BC_DEFAULT = 0, OCN_BC_PERIODIC = 1, OCN_BC_NOFLUX = 2, OCN_BC_FLUX = 3, OCN_BC_VALUE = 4,
       OCN_BC_GRADIENT = 5, OCN_BC_IMPENETRABLE = 6, OCN_BC_NONE = 7 };
/* sides */
enum { OCN_WEST = 0, OCN_EAST = 1, OCN_SOUTH = 2, OCN_NORTH = 3, OCN_BOTTOM = 4, OCN_TOP = 5 };

#define OCN_MAX_TRACERS 8

/* field identifiers inside a model (Fields/field_tuples.jl:133-246) */
enum {
  OCN_F_U = 0, OCN_F_V = 1, OCN_F_W = 2,
  OCN_F_PHY = 3,  /* pHY'  (absent for Flat z)  */
  OCN_F_PNHS = 4, /* pNHS                        */
  OCN_F_GN = 16,  /* G^n : OCN_F_GN + {0,1,2, 3+tracer}   (TimeSteppers: timestepper.G^n) */
  OCN_F_GM = 32,  /* G^- : OCN_F_GM + {0,1,2, 3+tracer}                                    */
  OCN_F_TRACER = 48, /* tracers: OCN_F_TRACER + index                                      */
  OCN_F_NU = 64,     /* AMD eddy viscosity nu_e;  OCN_F_KAPPA + index: kappa_e             */
  OCN_F_KAPPA = 72
};

typedef struct ocn_ctx ocn_ctx;
typedef struct ocn_grid ocn_grid;
typedef struct ocn_model ocn_model;
typedef struct ocn_field ocn_field;   /* a stand-alone field on a grid (ocn_field_create) */

/* RectilinearGrid(size=, halo=, topology=, x=, y=, z=)  -- Grids/rectilinear_grid.jl:249-279.
 * x and y must be regular (extent given); z is regular when z_faces == NULL, otherwise z_faces
 * holds the Nz+1 interior face positions (a "vertically stretched" grid even if the spacing is
 * uniform, which selects the Fourier-tridiagonal solver exactly as NonhydrostaticModels.jl:18-27). */
typedef struct ocn_grid_desc {
  int32_t N[3];        /* Nx, Ny, Nz (1 for Flat directions)              */
  int32_t H[3];        /* halo (0 for Flat directions)                    */
  int32_t topology[3]; /* OCN_PERIODIC / OCN_BOUNDED / OCN_FLAT           */
  double x0[3];        /* left end of the domain per direction            */
  double L[3];         /* extent per direction (ignored for stretched z)  */
  const double* z_faces; /* NULL or Nz+1 doubles (host)                   */
  /* domain decomposition (Distributed/multi_architectures.jl:20-47): N is the GLOBAL size; after ocn_comm_init the
   * library cuts triply periodic grids into z-slabs and (Periodic, Periodic, Bounded) grids into y-slabs and
   * every field of the model has this rank's local shape (ocn_field_shape) */
  int32_t rank, nranks;
} ocn_grid_desc;

/* one side of one field: kind + constant value or host array over the two other (interior) dims */
typedef struct ocn_bc {
  int32_t kind;         /* OCN_BC_*  (OCN_BC_DEFAULT -> field_boundary_conditions.jl:13-35) */
  double value;         /* used when array == NULL                                            */
  const double* array;  /* optional host array (N_a x N_b, column-major), copied at creation */
} ocn_bc;

/* NonhydrostaticModel(; grid, advection, buoyancy, coriolis, closure, boundary_conditions, tracers,
 * timestepper) -- Models/NonhydrostaticModels/nonhydrostatic_model.jl:102-203.  Forcings, Stokes drift,
 * background fields, immersed boundaries and particles are user Julia closures / out of scope: not
 * representable across a C ABI, the shim rejects them. */
typedef struct ocn_model_desc {
  int32_t advection;   /* OCN_ADV_*                                   */
  int32_t stepper;     /* OCN_STEPPER_*                               */
  double chi;          /* AB2 parameter (quasi_adams_bashforth_2.jl:45: 0.1) */
  int32_t n_tracers;   /* <= OCN_MAX_TRACERS                          */
  int32_t closure;     /* OCN_CLOSURE_*                               */
  double nu;           /* ScalarDiffusivity nu                        */
  double kappa[OCN_MAX_TRACERS]; /* ScalarDiffusivity kappa per tracer */
  double amd_Cnu;      /* AMD Poincare constants                      */
  double amd_Ckappa[OCN_MAX_TRACERS];
  double amd_Cb;       /* AMD buoyancy-modification multiplier (anisotropic_minimum_dissipation.jl:142-154,299-312) */
  int32_t amd_has_Cb;  /* 0: Cb = nothing (the default; the term is skipped), 1: amd_Cb is used */
  int32_t coriolis_fplane; /* 0 / 1 */
  double f;            /* FPlane f (Coriolis/f_plane.jl:42-44)         */
  int32_t buoyancy;    /* OCN_BUOYANCY_*                              */
  int32_t b_index, T_index, S_index; /* tracer indices used by the buoyancy model (-1: absent) */
  double g, alpha, beta; /* SeawaterBuoyancy(LinearEquationOfState)   */
  ocn_bc bcs[3 + OCN_MAX_TRACERS][6]; /* [field: u,v,w,tracers...][side] */
  /* boundary conditions of the AMD diffusivity fields (boundary_conditions = (; nu_e = ..., kappa_e = (; T = ...)) in the
   * reference: nonhydrostatic_model.jl:150-160); OCN_BC_DEFAULT (0) everywhere = the auxiliary-field defaults */
  ocn_bc nu_bcs[6];
  ocn_bc kappa_bcs[OCN_MAX_TRACERS][6];
} ocn_model_desc;

/* ---- context (Architectures.jl:53-142: device, array_type, arch_array, device_event) ---------- */
int ocn_abi_version(void);
int ocn_init(int device_id, ocn_ctx** out);
void ocn_destroy(ocn_ctx* ctx);
int ocn_sync(ocn_ctx* ctx);                       /* wait(device(arch), event) everywhere in the reference.  Slab runs: the
                                                   * halo planes of the last step may still be travelling on the library's
                                                   * communication stream when ocn_time_step returns; ocn_sync, every
                                                   * field accessor and every phase-level call wait for them first */
const char* ocn_last_error(ocn_ctx* ctx);         /* ctx may be NULL: last global error */
void* ocn_stream(ocn_ctx* ctx);                   /* the hipStream_t, for callers that enqueue their own work */

/* ---- grid (Grids/rectilinear_grid.jl:249-279, Grids/zeros.jl:7) --------------------------------- */
int ocn_grid_create(ocn_ctx* ctx, const ocn_grid_desc* desc, ocn_grid** out);
void ocn_grid_destroy(ocn_grid* g);

/* ---- model -------------------------------------------------------------------------------------- */
int ocn_model_create(ocn_grid* g, const ocn_model_desc* desc, ocn_model** out);
void ocn_model_destroy(ocn_model* m);
/* which kernels serve this model, and if not the fastest ones, why (e.g. "general kernels: parent arrays of 2 GiB or
 * more exceed the tiled kernels' 32-bit byte offsets").  Writes at most n bytes incl. the terminating 0. */
int ocn_model_path(const ocn_model* m, char* buf, size_t n);
/* Whole-step hipGraphs: models on the general kernels and all-in-one periodic boxes of up to 96^3 cells (not slab-decomposed) replay the
 * launches of a step from a hipGraph once a (dt, stepper state) pair repeats -- the launch train of a small model is
 * latency-bound (time_step! of the reference issues the same ~50 launches from Julia: TimeSteppers/quasi_adams_bashforth_2.jl:70-104).
 * *replays = steps served by a graph so far; *active = 1 while the model may use graphs.  OCNHIP_NO_GRAPH=1 disables. */
int ocn_model_graph_replays(const ocn_model* m, int64_t* replays, int32_t* active);
/* halo actually used (the model inflates it like nonhydrostatic_model.jl:140-148) */
int ocn_model_halo(const ocn_model* m, int32_t H[3]);

/* ---- stand-alone fields: Field{LX,LY,LZ}(grid) / zeros(FT, arch, N...) (Fields/field.jl:16-30, Grids/new_data.jl:16-61,
 * Grids/zeros.jl:7).  A zero-filled parent array, halos included, with the layout a model's own field of that location has
 * on this grid (ocn_field_parent_layout reports strides / origin; dense column-major on (Periodic, Periodic, *) grids).
 * loc*: OCN_CENTER / OCN_FACE.  The grid must outlive the field. */
int ocn_field_create(ocn_grid* g, int locx, int locy, int locz, ocn_field** out);
void ocn_field_destroy(ocn_field* f);
int ocn_field_parent_shape(const ocn_field* f, int32_t total[3], int32_t interior[3], int32_t halo[3]);
int ocn_field_parent_layout(const ocn_field* f, int64_t strides[3], int64_t* origin);
void* ocn_field_parent_ptr(ocn_field* f);                                  /* Julia unsafe_wrap / torch from_blob */
int ocn_field_parent_upload(ocn_field* f, const double* host_parent);      /* logical parent array, column-major  */
int ocn_field_parent_download(const ocn_field* f, double* host_parent);

/* ---- fields: parent arrays incl. halos (Fields/field.jl:16-30; OutputWriters/fetch_output.jl:26) - */
int ocn_field_shape(const ocn_model* m, int field_id, int32_t total[3], int32_t interior[3], int32_t halo[3]);
void* ocn_field_device_ptr(ocn_model* m, int field_id);           /* Julia unsafe_wrap / torch from_blob */
/* element strides (x, y, z) of the device array and the element offset of the logical parent's first entry;
 * strides == (1, T0, T0*T1) and origin == 0 on (Periodic, Periodic, *) grids */
int ocn_field_layout(const ocn_model* m, int field_id, int64_t strides[3], int64_t* origin);
int ocn_field_upload(ocn_model* m, int field_id, const double* host_parent);   /* arch_array(GPU(), a) */
int ocn_field_download(const ocn_model* m, int field_id, double* host_parent); /* arch_array(CPU(), a) */
int ocn_field_set_interior(ocn_model* m, int field_id, const double* host_interior); /* Fields/set!.jl:21-39 */
int ocn_field_get_interior(const ocn_model* m, int field_id, double* host_interior);

/* ---- phase-level entry points (each replaces one overloaded Julia method) ------------------------- */
/* fill_halo_regions!(c, bcs, loc, grid)                      BoundaryConditions/fill_halo_regions.jl:34-46
 * mask: bit f set -> fill field f in {u,v,w,pHY,pNHS} ; bit (8+i) -> tracer i                           */
int ocn_fill_halos(ocn_model* m, uint32_t field_mask);
/* update_state!(model)          Models/NonhydrostaticModels/update_nonhydrostatic_model_state.jl:14-37 */
int ocn_update_state(ocn_model* m);
/* calculate_tendencies!(model)  .../calculate_nonhydrostatic_tendencies.jl:12-36 (interior + boundary) */
int ocn_compute_tendencies(ocn_model* m);
/* ab2_step!(model, dt, chi)     TimeSteppers/quasi_adams_bashforth_2.jl:116-150.  The caller of the phase-level entry
 * points owns the clock (ocn_set_clock): previous_dt decides between Euler and AB2 inside ocn_time_step only. */
int ocn_ab2_step(ocn_model* m, double dt, double chi);
/* rk3_substep!(model, dt, gamma, zeta)  TimeSteppers/runge_kutta_3.jl:161-218 ; has_zeta = 0 for stage 1 */
int ocn_rk3_substep(ocn_model* m, double dt, double gamma, double zeta, int has_zeta);
/* store_tendencies!(model)      TimeSteppers/store_tendencies.jl:14-36 */
int ocn_store_tendencies(ocn_model* m);
/* calculate_pressure_correction!(model, dt)  .../pressure_correction.jl:10-23 (fill, rhs, solve, fill) */
int ocn_pressure_correction(ocn_model* m, double dt);
/* solve!(phi, solver, rhs): Solvers/fft_based_poisson_solver.jl:93-120 or
 * Solvers/fourier_tridiagonal_poisson_solver.jl:74-101 (rhs NOT pre-multiplied by dz).
 * rhs / phi: host arrays (Nx,Ny,Nz), used by the Poisson property tests.                                */
int ocn_poisson_solve_host(ocn_model* m, const double* rhs, double* phi);
/* pressure_correct_velocities!(model, dt)    .../pressure_correction.jl:43-56 */
int ocn_pressure_correct_velocities(ocn_model* m, double dt);
/* set!(model; ...) epilogue: update_state!, unit-dt projection, update_state!
 *                                .../set_nonhydrostatic_model.jl:45-58 */
int ocn_set_epilogue(ocn_model* m, int enforce_incompressibility);
/* time_step!(model, dt; euler)  TimeSteppers/quasi_adams_bashforth_2.jl:70-104 / runge_kutta_3.jl:81-152.
 * Whole sequence enqueued on the stream without host synchronisation.                                   */
int ocn_time_step(ocn_model* m, double dt, int force_euler);
/* model.clock (TimeSteppers/clock.jl:14-18) */
int ocn_clock(const ocn_model* m, double* time, int64_t* iteration, int32_t* stage);
int ocn_set_clock(ocn_model* m, double time, int64_t iteration, double previous_dt);
/* max |div U| over the interior (test helper `divergence!`, test/utils_for_runtests.jl:60-67) */
int ocn_max_abs_divergence(ocn_model* m, double* out);

/* ---- multi-GPU: one process per GPU, RCCL (Distributed/multi_architectures.jl:20-137, halo_communication.jl:68-183,
 * distributed_fft_based_poisson_solver.jl:95-196).  Call ocn_comm_init BEFORE ocn_grid_create: grids created afterwards
 * are cut into z-slabs (triply Periodic) or y-slabs ((Periodic, Periodic, Bounded)); every compute call is then
 * collective and must be issued by all ranks in the same order. ---------------------------------------- */
int ocn_comm_unique_id(void* out128);                                      /* ncclGetUniqueId (128 bytes) */
int ocn_comm_init(ocn_ctx* ctx, int rank, int nranks, const void* unique_id128);
/* Can a communicator of `nranks` form?  A throw-away non-blocking ncclCommInitRankConfig polled against `timeout_s` and
 * aborted: returns on EVERY rank (OCN_OK, RCCL's error, or a time-out) where the blocking ocn_comm_init would leave the
 * healthy ranks waiting for ever for one that failed.  Collective; use a unique id of its own.  (The reference's
 * MPI.Init has no counterpart: an MPI job that loses a rank at start-up is killed by its launcher.)               */
int ocn_comm_probe(ocn_ctx* ctx, int rank, int nranks, const void* unique_id128, double timeout_s);
int ocn_comm_rank(const ocn_ctx* ctx, int* rank, int* nranks);

/* ---- HydrostaticFreeSurfaceModel, first slice: SplitExplicitFreeSurface on a RectilinearGrid or LatitudeLongitudeGrid
 * (BASELINE config 5; Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface.jl, split_explicit_free_surface_kernels.jl,
 * Grids/latitude_longitude_grid.jl).  The 3-D momentum tendencies of that model are not part of this library yet: the caller
 * hands their arrays in (ocn_hfield) exactly as ab2_step_free_surface! receives them from the time stepper. ------------------ */
enum { OCN_NOTHING = 2 };   /* third field location: reduced along that direction (Field{LX, LY, Nothing}, Fields/field.jl:441-449) */
enum { OCN_HGRID_RECTILINEAR = 0, OCN_HGRID_LATLON = 1 };
typedef struct ocn_hgrid ocn_hgrid;     /* the grid as the free surface sees it: sizes, halos, topology, per-row metrics, level thicknesses */
typedef struct ocn_hfield ocn_hfield;   /* a field on it: (Face|Center, Face|Center, Center|Face|Nothing), dense parent array incl. halos     */
typedef struct ocn_sefs ocn_sefs;       /* SplitExplicitFreeSurface(grid; gravitational_acceleration, settings)                            */

/* RectilinearGrid(size, x, y, z, halo, topology) with regular x, y (metres), or
 * LatitudeLongitudeGrid(size, longitude, latitude, z, halo, radius, precompute_metrics = true) with regular longitude / latitude
 * (degrees; latitude_longitude_grid.jl:174-213: pass Periodic for x when the longitude spans 360 degrees, else Bounded; y Bounded).
 * z is Bounded, regular (z_faces == NULL, x0[2] .. x0[2] + L[2]) or given by its Nz + 1 faces. */
typedef struct ocn_hgrid_desc {
  int32_t kind;          /* OCN_HGRID_*                                  */
  int32_t N[3], H[3], topology[3];
  double x0[3], L[3];
  const double* z_faces; /* NULL or Nz + 1 doubles (host)                */
  double radius;         /* lat-lon: sphere radius (<= 0: R_Earth, latitude_longitude_grid.jl:3) */
  int32_t partition;     /* 0: the whole grid on this context.  1: latitude bands (y-slabs) over the context's ranks (ocn_comm_init
                          * first): N, x0, L describe the GLOBAL grid, the handle is rank r's band of N[1] / nranks rows (the role
                          * of Partition(1, R) in the reference's DistributedArch, Distributed/multi_architectures.jl); its fields keep the
                          * Bounded shape, fill_halos exchanges rows with the neighbouring bands.  ocn_hgrid_band reports the rows. */
  int32_t band_overlap;  /* with partition = 1: W > 0 makes the handle the band EXTENDED by W rows towards each neighbouring band (none
                          * towards a wall), as a stand-alone Bounded grid -- what a banded SplitExplicitFreeSurface lives on: it sub-cycles
                          * W substeps on the extended rows (the artificial walls spoil one row per substep from the outside in) and
                          * refreshes the overlap rows from the neighbours.  0: the plain band. */
} ocn_hgrid_desc;
int ocn_hgrid_create(ocn_ctx* ctx, const ocn_hgrid_desc* desc, ocn_hgrid** out);
/* rows of the global grid this handle holds: first row (0-based offset j0) and count; global count; whole grid: 0, Ny, Ny */
int ocn_hgrid_band(const ocn_hgrid* g, int32_t* j0, int32_t* ny_local, int32_t* ny_global);
void ocn_hgrid_destroy(ocn_hgrid* g);
/* metrics and nodes as the grid object holds them (grid.Δxᶠᶜᵃ ... latitude_longitude_grid.jl:418-445; grid.φᵃᶠᵃ ...): which =
 * 0 Δx^fc, 1 Δx^cf, 2 Δy^fc, 3 Δy^cf, 4 Az^cc (per row, first entry = row 1 - Hy), 5 Δz^c (levels 1..Nz), 6 / 7 x nodes Face /
 * Center, 8 / 9 y nodes Face / Center (incl. halos).  Copies min(n, entries) doubles, returns the number of entries. */
int ocn_hgrid_metric(const ocn_hgrid* g, int which, double* host, int n);
/* Field{LX, LY, LZ}(grid), LZ = OCN_CENTER or OCN_NOTHING; zero-filled dense parent array (Grids/new_data.jl:16-61) */
int ocn_hfield_create(ocn_hgrid* g, int locx, int locy, int locz, ocn_hfield** out);
void ocn_hfield_destroy(ocn_hfield* f);
int ocn_hfield_shape(const ocn_hfield* f, int32_t total[3], int32_t interior[3], int32_t halo[3]);
void* ocn_hfield_ptr(ocn_hfield* f);                       /* device pointer of the parent array (Julia: unsafe_wrap)     */
int ocn_hfield_upload(ocn_hfield* f, const double* host_parent);
int ocn_hfield_download(const ocn_hfield* f, double* host_parent);
/* fill_halo_regions!(field) with the default conditions: z (no-flux for Center, faces 1 and Nz + 1 zeroed for a ZFaceField:
 * fill_halo_regions_open.jl:34-39), then x and y (Bounded directions before Periodic ones, fill_halo_regions.jl:76-99) */
int ocn_hfield_fill_halos(ocn_hfield* f);
/* SplitExplicitFreeSurface(grid; gravitational_acceleration, settings = SplitExplicitSettings(substeps))
 * (split_explicit_free_surface.jl:46-60; H^fc, H^cf, H^cc = sum of dz :103-110; uniform weights :137-154) */
int ocn_sefs_create(ocn_hgrid* g, double gravitational_acceleration, int substeps, ocn_sefs** out);
void ocn_sefs_destroy(ocn_sefs* s);
/* the free surface's own fields (owned by it): 0 η, 1 U, 2 V, 3 η̅, 4 U̅, 5 V̅, 6 Gᵁ, 7 Gⱽ, 8 Hᶠᶜ, 9 Hᶜᶠ, 10 Hᶜᶜ */
ocn_hfield* ocn_sefs_field(ocn_sefs* s, int which);
/* SplitExplicitSettings(substeps, velocity_weights, free_surface_weights)  (:130-135) */
int ocn_sefs_set_weights(ocn_sefs* s, int n, const double* velocity_weights, const double* free_surface_weights);
/* split_explicit_free_surface_substep!(η, state, auxiliary, settings, arch, grid, g, Δτ, substep_index)  (kernels.jl:31-58) */
int ocn_sefs_substep(ocn_sefs* s, double dtau, int substep_index);
/* `for substep in first:first+count-1 substep!(...) end` (kernels.jl:154-156).  fused = 1: two launches per substep instead
 * of five; fused = 2: one (eta, U, V double buffered inside the object); fused = 3: four substeps per launch on tiles with a ring
 * of four ghost cells (grids of 64 x 16 cells and more; smaller ones run as 2), the last substep of the train by the one-launch
 * kernel; whichever, the whole train is replayed from a hipGraph and leaves the same bits in every parent array, halos included.
 * 0: the reference's five launches. */
int ocn_sefs_substeps(ocn_sefs* s, double dtau, int first_index, int count, int fused);
int ocn_sefs_graph_replays(const ocn_sefs* s, int64_t* replays);
/* barotropic_mode!(U, V, grid, u, v) (:76-81): into_forcing == 0 -> state.U, state.V; != 0 -> auxiliary.Gᵁ, Gⱽ */
int ocn_sefs_barotropic_mode(ocn_sefs* s, const ocn_hfield* u, const ocn_hfield* v, int into_forcing);
int ocn_sefs_set_average_to_zero(ocn_sefs* s);             /* (:83-87) */
/* barotropic_split_explicit_corrector!(u, v, free_surface, grid) (:97-113) */
int ocn_sefs_corrector(ocn_sefs* s, ocn_hfield* u, ocn_hfield* v);
/* split_explicit_free_surface_step!(free_surface, model, Δt, χ, velocities_update) (:124-171) given timestepper.Gⁿ.u / .v and
 * G⁻.u / .v: averages reset, barotropic mode of the AB2-combined tendencies, `substeps` substeps of Δτ = 2Δt / substeps,
 * η ← η̅, halos of η filled */
int ocn_sefs_step(ocn_sefs* s, const ocn_hfield* Gn_u, const ocn_hfield* Gn_v, const ocn_hfield* Gm_u, const ocn_hfield* Gm_v,
                  double dt, double chi);

/* ---- HydrostaticFreeSurfaceModel: the AB2 time step around the tendency evaluation (config 5, second slice) --------------
 * time_step!(model::HydrostaticFreeSurfaceModel, dt) (TimeSteppers/quasi_adams_bashforth_2.jl:70-104) is
 *   calculate_tendencies!  ->  ab2_step!  ->  pressure_correct_velocities!  ->  store_tendencies!  ->  update_state!
 * The entry points below are everything after calculate_tendencies!; the caller (until the tendency kernels are in the
 * library too) fills G^n.  No closure / implicit vertical solve, no immersed boundary, flat bottom. */
typedef struct ocn_hydro ocn_hydro;
/* ab2_step_field! (quasi_adams_bashforth_2.jl:158-166): f += dt ((1.5 + chi) G^n - (0.5 + chi) G^-) over the grid's cells */
int ocn_hfield_ab2_step(ocn_hfield* f, const ocn_hfield* Gn, const ocn_hfield* Gm, double dt, double chi);
/* store_field_tendencies! (TimeSteppers/store_tendencies.jl:8-11): G^-[i, j, k] = G^n[i, j, k] over the grid's cells */
int ocn_hfield_store_tendency(ocn_hfield* Gm, const ocn_hfield* Gn);
/* compute_w_from_continuity! (compute_w_from_continuity.jl:31-36): w[1] = 0, w[k] = w[k-1] - dz^c[k-1] div_xy(u, v)[k-1];
 * u, v with filled x / y halos */
int ocn_hydro_compute_w(const ocn_hfield* u, const ocn_hfield* v, ocn_hfield* w);
/* update_hydrostatic_pressure! (Models/NonhydrostaticModels/update_hydrostatic_pressure.jl:10-18) with the buoyancy
 * kind 0: none, 1: BuoyancyTracer (T is b), 2: SeawaterBuoyancy(gravitational_acceleration, LinearEquationOfState(alpha, beta))
 * (BuoyancyModels/linear_equation_of_state.jl:69-71); T, S with filled z halos */
int ocn_hydro_pressure(ocn_hfield* pHY, int buoyancy_kind, double gravitational_acceleration, double thermal_expansion,
                       double haline_contraction, const ocn_hfield* T, const ocn_hfield* S);
typedef struct ocn_hydro_desc {
  ocn_sefs* free_surface;
  ocn_hfield *u, *v, *w, *pHY;                /* (F,C,C), (C,F,C), (C,C,F), (C,C,C)                                   */
  int32_t ntracers;
  ocn_hfield** tracers;                       /* ntracers (C,C,C) fields                                              */
  ocn_hfield **Gn, **Gm;                      /* 2 + ntracers each: u, v, then the tracers in order                   */
  int32_t buoyancy_kind, T_index, S_index;    /* as ocn_hydro_pressure; indices into `tracers`                        */
  double gravitational_acceleration, thermal_expansion, haline_contraction;
} ocn_hydro_desc;
/* binds the model's fields (borrowed: they must outlive the handle) */
int ocn_hydro_create(const ocn_hydro_desc* desc, ocn_hydro** out);
void ocn_hydro_destroy(ocn_hydro* h);
/* update_state!(model) (update_hydrostatic_free_surface_model_state.jl:21-48): halo fills of u, v, eta and the tracers, w from
 * continuity, the hydrostatic pressure, halo fills of w and pHY' */
int ocn_hydro_update_state(ocn_hydro* h);
/* ab2_step!(model, dt, chi) (hydrostatic_free_surface_ab2_step.jl:15-48): barotropic mode of u, v; AB2 steps of u, v, tracers;
 * ocn_sefs_step */
int ocn_hydro_ab2_step(ocn_hydro* h, double dt, double chi);
/* ab2_step!, barotropic correction of u and v (barotropic_pressure_correction.jl:41-47), G^- <- G^n, update_state!.
 * fused = 0 issues the reference's kernels one by one; fused = 1 merges the passes over the 3-D fields (about 28 instead of
 * 40 field sweeps) and leaves the same bits in every field, halos included. */
int ocn_hydro_step_after_tendencies(ocn_hydro* h, double dt, double chi, int fused);

/* ---- third slice: calculate_tendencies! (no closure, forcing or immersed boundary) and the whole time_step! ---------------
 * momentum_advection: 0 nothing, 1 VectorInvariant(scheme = EnstrophyConservingScheme()) -- the default --, 2 EnergyConservingScheme
 *   (Advection/vector_invariant_advection.jl:25-80), 3 WENO5(vector_invariant = VorticityStencil()): the vertical-vorticity term as
 *   transporting velocity times the upwind-biased WENO5 interpolation of zeta (vector_invariant_advection.jl:54-66), halo 3;
 * coriolis: 0 nothing, 1 / 2 HydrostaticSphericalCoriolis(rotation_rate = coriolis_parameter) with the Enstrophy- / Energy-
 *   ConservingScheme (Coriolis/hydrostatic_spherical_coriolis.jl:29-66; LatitudeLongitudeGrid only), 3 FPlane(f = coriolis_parameter);
 * tracer_advection: 0 nothing, 1 CenteredSecondOrder() -- the default; bit-exact against the oracle --, 2 CenteredFourthOrder(),
 *   3 UpwindBiasedFifthOrder(), 4 WENO5() (Z weights, uniform coefficients): flux form with this grid's areas
 *   (tracer_advection_operators.jl:33-37, upwind_biased_advective_fluxes.jl:103-128), second-order inside the boundary buffer of a
 *   Bounded direction (topologically_conditional_interpolation.jl:19-83); halos of 2 / 3 / 3 cells.
 * Defaults of a new handle: 1, 0, 1 -- the reference model's. */
int ocn_hydro_set_physics(ocn_hydro* h, int momentum_advection, int coriolis, double coriolis_parameter, int tracer_advection);
/* closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa = (kappa per tracer)) with constant
 * coefficients: implicit_step! of u, v and every tracer inside ab2_step! (hydrostatic_free_surface_ab2_step.jl:72-85,115-128;
 * TurbulenceClosures/vertically_implicit_diffusion_solver.jl:46-100, Solvers/batched_tridiagonal_solver.jl:89-121); no flux through top
 * and bottom; ntracers must be the handle's; all zeros (the default) switch it off.  Other closures stay on the reference's path. */
int ocn_hydro_set_closure(ocn_hydro* h, double nu, int32_t ntracers, const double* kappa);
/* calculate_tendencies!(model) (calculate_hydrostatic_free_surface_tendencies.jl:15-160): G^n of u, v and every tracer over the
 * grid's cells, from the state update_state! left (filled halos, w, pHY') */
int ocn_hydro_calculate_tendencies(ocn_hydro* h);
/* time_step!(model, dt; euler) (TimeSteppers/quasi_adams_bashforth_2.jl:70-104): chi = -1/2 and G^- = 0 when euler, else 0.1 */
int ocn_hydro_time_step(ocn_hydro* h, double dt, int euler);

/* ---- measurement helpers (bench.py) -------------------------------------------------------------- */
/* average device time [ms] of the `n` most recent launches of the named phase, measured with HIP
 * events on the context stream when profiling is enabled */
int ocn_profile_enable(ocn_ctx* ctx, int on);
/* restrict event recording to one phase (NULL / "": all phases).  Every recorded event costs a few microseconds of
 * stream time, so a benchmark that times whole steps records only the phase it reports. */
int ocn_profile_filter(ocn_ctx* ctx, const char* phase);
int ocn_profile_read(ocn_ctx* ctx, const char* phase, double* avg_ms, int64_t* count);
int ocn_profile_reset(ocn_ctx* ctx);
/* measured device-to-device copy rate of this GPU: `reps` stream-ordered copies of `bytes` bytes, timed with HIP events;
 * *bytes_per_s counts bytes read + bytes written (the second roofline denominator of BASELINE.md section 2) */
int ocn_measure_copy_rate(ocn_ctx* ctx, size_t bytes, int reps, double* bytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* OCNHIP_H */
