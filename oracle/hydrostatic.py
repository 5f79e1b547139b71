"""The AB2 time step of the HydrostaticFreeSurfaceModel around its tendency evaluation (oracle; test infrastructure only):
everything `time_step!` (TimeSteppers/quasi_adams_bashforth_2.jl:70-104) does for that model EXCEPT `calculate_tendencies!`,
which the caller supplies as G^n.  Second slice of BASELINE config 5, on the grids of ``split_explicit.py``.

Restates (paths relative to /root/reference/src):
  * ``Models/HydrostaticFreeSurfaceModels/hydrostatic_free_surface_ab2_step.jl:15-48`` -- ``ab2_step!``: ``local_ab2_step!``
    (``barotropic_mode!(U, V, grid, u, v)`` of the velocities BEFORE they are stepped, ``ab2_step_velocities!``, ``ab2_step_tracers!``;
    no implicit vertical solve: closures are out of this slice), then ``ab2_step_free_surface!`` = ``split_explicit_free_surface_step!``;
  * ``TimeSteppers/quasi_adams_bashforth_2.jl:158-166`` -- ``ab2_step_field!`` over the grid's (Nx, Ny, Nz);
  * ``barotropic_pressure_correction.jl:41-47`` -- ``pressure_correct_velocities!`` = ``barotropic_split_explicit_corrector!``;
  * ``TimeSteppers/store_tendencies.jl:8-36`` -- ``G^- <- G^n`` for u, v and the tracers, over the grid's cells (the tendency of
    eta, which the split-explicit free surface never reads, is not carried);
  * ``update_hydrostatic_free_surface_model_state.jl:21-48`` -- ``update_state!``: halo fills of the prognostic fields (u, v, eta,
    tracers), ``compute_w_from_continuity!`` (``compute_w_from_continuity.jl:31-36``), ``update_hydrostatic_pressure!``
    (``Models/NonhydrostaticModels/update_hydrostatic_pressure.jl:10-18``), fills of w and pHY';
  * halo fills in z: Center -> no-flux (first halo cell), Face -> the boundary faces are set to zero
    (``fill_halo_regions_open.jl:34-39``; ``w = ZFaceField(grid)`` carries the default impenetrable conditions,
    ``hydrostatic_free_surface_field_tuples.jl:7``, so the top face computed from continuity is zeroed by the fill -- as written).

The reference holds no known answers for these pieces (its hydrostatic tests are "a time step runs"): parity unpinned beyond the
properties asserted in tests/test_hydrostatic_step.py (discrete continuity, a resting ocean stays at rest, the AB2 formula).
"""
import numpy as np

from .grid import Bounded, Center, Face, Periodic, total_length
from . import split_explicit as SE


Field3 = SE.Field3
fill_halo_regions = SE.fill_halo_regions


def compute_w_from_continuity(u, v, w):
    """w[1] = 0; w[k] = w[k-1] - dz^c[k-1] div_xy^ccc(k-1) for k = 2..Nz+1, over i = 1..Nx, j = 1..Ny (compute_w_from_continuity.jl:31-36)"""
    g = w.grid
    Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
    I, J = slice(Hx, Hx + Nx), slice(Hy, Hy + Ny)
    Ip, Jp = slice(Hx + 1, Hx + Nx + 1), slice(Hy + 1, Hy + Ny + 1)
    row = lambda a: a[Hy:Hy + Ny].reshape(1, -1)        # noqa: E731
    rowp = lambda a: a[Hy + 1:Hy + Ny + 1].reshape(1, -1)   # noqa: E731
    dz = g.dz_centers()
    w.data[I, J, Hz] = 0.0
    for k in range(1, Nz + 1):                           # reference k = 2..Nz+1 -> parent level Hz + k
        U, V = u.data[:, :, Hz + k - 1], v.data[:, :, Hz + k - 1]
        div = 1 / row(g.Az_cc) * ((row(g.dy_fc) * U[Ip, J] - row(g.dy_fc) * U[I, J]) + (rowp(g.dx_cf) * V[I, Jp] - row(g.dx_cf) * V[I, J]))
        w.data[I, J, Hz + k] = w.data[I, J, Hz + k - 1] - dz[k - 1] * div


def ab2_step_field(f, Gn, Gm, dt, chi):
    """u[i, j, k] += dt ((1.5 + chi) G^n - (0.5 + chi) G^-) over i = 1..Nx, j = 1..Ny, k = 1..Nz (:158-166)"""
    g = f.grid
    idx = (slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny), slice(g.Hz, g.Hz + g.Nz))
    f.data[idx] += dt * ((1.5 + chi) * Gn.data[idx] - (0.5 + chi) * Gm.data[idx])


def buoyancy_perturbation(buoyancy, tracers):
    """parent array of b incl. halos.  None: 0; ("b", name): BuoyancyTracer; ("TS", g, alpha, beta, Tname, Sname): SeawaterBuoyancy
    with a LinearEquationOfState, g (alpha T - beta S) (BuoyancyModels/linear_equation_of_state.jl:69-71)"""
    if buoyancy is None:
        return None
    if buoyancy[0] == "b":
        return tracers[buoyancy[1]].data
    _, g, al, be, Tn, Sn = buoyancy
    return g * (al * tracers[Tn].data - be * tracers[Sn].data)


def update_hydrostatic_pressure(pHY, buoyancy, tracers):
    """pHY'[Nz] = -I_z(b)[Nz+1] dz^f[Nz+1]; pHY'[k] = pHY'[k+1] - I_z(b)[k+1] dz^f[k+1], k = Nz-1..1, I_z(b)[k] = (b[k] + b[k-1]) / 2
    with b read from the tracers incl. their z halos (update_hydrostatic_pressure.jl:10-18)"""
    g = pHY.grid
    Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
    I, J = slice(Hx, Hx + Nx), slice(Hy, Hy + Ny)
    az = g.ax[2]
    dzf = lambda k: az.df if az.regular else float(az.d_face(k))   # noqa: E731
    bd = buoyancy_perturbation(buoyancy, tracers)
    if bd is None:
        bd = np.zeros_like(pHY.data)
    bf = lambda k: (bd[I, J, Hz + k - 1] + bd[I, J, Hz + k - 2]) / 2   # noqa: E731   face k: levels k and k - 1
    pHY.data[I, J, Hz + Nz - 1] = -bf(Nz + 1) * dzf(Nz + 1)
    for k in range(Nz - 1, 0, -1):
        pHY.data[I, J, Hz + k - 1] = pHY.data[I, J, Hz + k] - bf(k + 1) * dzf(k + 1)


class HydrostaticState:
    """the fields of a HydrostaticFreeSurfaceModel{SplitExplicitFreeSurface} this slice touches"""

    def __init__(self, grid, tracers=("T", "S"), buoyancy=None, substeps=20, gravitational_acceleration=SE.G_EARTH, F3=Field3,
                 free_surface=None, momentum_advection="VectorInvariantEnstrophyConserving", coriolis=None,
                 tracer_advection="CenteredSecondOrder", closure=None):
        self.grid = grid
        self.closure = closure        # None | (nu, kappa | {tracer: kappa}): VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa)
        self.momentum_advection, self.coriolis, self.tracer_advection = momentum_advection, coriolis, tracer_advection
        self.u, self.v, self.w = F3(grid, Face, Center, Center), F3(grid, Center, Face, Center), F3(grid, Center, Center, Face)
        self.tracers = {n: F3(grid, Center, Center, Center) for n in tracers}
        names = ["u", "v"] + list(tracers)
        loc = {"u": (Face, Center), "v": (Center, Face)}
        self.Gn = {n: F3(grid, *loc.get(n, (Center, Center)), Center) for n in names}
        self.Gm = {n: F3(grid, *loc.get(n, (Center, Center)), Center) for n in names}
        self.pHY = F3(grid, Center, Center, Center)
        self.buoyancy = buoyancy                                         # None | ("b", name) | ("TS", g, alpha, beta, Tname, Sname)
        self.free_surface = free_surface or SE.SplitExplicitFreeSurface(grid, gravitational_acceleration, substeps)
        self.chi = 0.1


def update_state(st):
    """update_hydrostatic_free_surface_model_state.jl:21-48 (no immersed boundary, no closure)"""
    for f in [st.u, st.v, st.free_surface.eta] + list(st.tracers.values()):
        fill_halo_regions(f)
    compute_w_from_continuity(st.u, st.v, st.w)
    update_hydrostatic_pressure(st.pHY, st.buoyancy, st.tracers)
    fill_halo_regions(st.w)
    fill_halo_regions(st.pHY)


def implicit_step(f, kappa, dt):
    """implicit_step!(field, solver, closure::VerticalScalarDiffusivity{VerticallyImplicitTimeDiscretization}, ...) with a constant
    diffusivity: (1 - dt d_z kappa d_z) c^{n+1} = c* per column, coefficients of vertically_implicit_diffusion_solver.jl:46-100
    (kappa / dz^c / dz^f; no flux through top and bottom), solved in place by the modified Thomas algorithm of
    Solvers/batched_tridiagonal_solver.jl:89-121 (its |beta| <= 10 eps early exit cannot trigger: beta >= 1 here)"""
    if not kappa:
        return
    g = f.grid
    Nz, Hz = g.Nz, g.Hz
    az = g.ax[2]
    dzc = lambda k: az.dc if az.regular else float(az.d_center(k))      # noqa: E731  1-based level
    dzf = lambda k: az.df if az.regular else float(az.d_face(k))        # noqa: E731  1-based face
    upper = lambda k: 0.0 if k > Nz - 1 else -dt * (kappa / dzc(k) / dzf(k + 1))          # noqa: E731  ivd_upper_diagonal(k)
    lower = lambda k: 0.0 if k < 1 else -dt * (kappa / dzc(k + 1) / dzf(k + 1))           # noqa: E731  ivd_lower_diagonal(k), k' = k + 1
    diag = lambda k: (1.0 - dt * 0.0 - upper(k)) - lower(k - 1)                           # noqa: E731  ivd_diagonal(k)
    I, J = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny)
    P = f.data
    lev = lambda k: (I, J, Hz + k - 1)                                                    # noqa: E731
    beta = diag(1)
    P[lev(1)] = P[lev(1)] / beta
    t = [0.0] * (Nz + 2)
    for k in range(2, Nz + 1):
        t[k] = upper(k - 1) / beta
        beta = diag(k) - lower(k - 1) * t[k]
        assert abs(beta) > 10 * np.finfo(float).eps
        P[lev(k)] = (P[lev(k)] - lower(k - 1) * P[lev(k - 1)]) / beta
    for k in range(Nz - 1, 0, -1):
        P[lev(k)] = P[lev(k)] - t[k + 1] * P[lev(k + 1)]


def ab2_step(st, dt, chi):
    """ab2_step!(model::HydrostaticFreeSurfaceModel, dt, chi) (hydrostatic_free_surface_ab2_step.jl:15-48, :60-130): barotropic mode,
    explicit steps of the velocities then their implicit vertical-diffusion solves, the same for the tracers, the free surface"""
    fs = st.free_surface
    nu, kap = getattr(st, "closure", None) or (0.0, {})
    fs.barotropic_mode(fs.U, fs.V, st.u, st.v)             # local_ab2_step!: the barotropic mode of the velocities before the step
    for n in ("u", "v"):
        ab2_step_field(getattr(st, n), st.Gn[n], st.Gm[n], dt, chi)
    for n in ("u", "v"):
        implicit_step(getattr(st, n), nu, dt)
    for n, c in st.tracers.items():
        ab2_step_field(c, st.Gn[n], st.Gm[n], dt, chi)
    for n, c in st.tracers.items():
        implicit_step(c, kap.get(n, 0.0) if isinstance(kap, dict) else kap, dt)
    fs.step(st.Gn["u"], st.Gn["v"], st.Gm["u"], st.Gm["v"], dt, chi)


def time_step_after_tendencies(st, dt, chi, fused=False):
    """time_step! from `ab2_step!` on (quasi_adams_bashforth_2.jl:94-100): step, barotropic correction, store tendencies, update_state!"""
    ab2_step(st, dt, chi)
    st.free_surface.corrector(st.u, st.v)                  # pressure_correct_velocities!(::SplitExplicitFreeSurfaceHFSM)
    g = st.grid
    idx = (slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny), slice(g.Hz, g.Hz + g.Nz))
    for n in st.Gn:                                         # store_tendencies!: the grid's cells (store_tendencies.jl:8-11,24-28)
        st.Gm[n].data[idx] = st.Gn[n].data[idx]
    update_state(st)


# ---- third slice: calculate_tendencies! ---------------------------------------------------------------------------------------
# Restates, for a HydrostaticFreeSurfaceModel with a SplitExplicitFreeSurface, no closure, no forcing, no immersed boundary:
#   * ``hydrostatic_free_surface_tendency_kernel_functions.jl:24-125`` -- G_u, G_v, G_c as sums of the terms below, in that order;
#   * ``Advection/vector_invariant_advection.jl:25-80`` -- ``VectorInvariant`` momentum advection (the only one a curvilinear grid
#     accepts), EnstrophyConservingScheme (default) or EnergyConservingScheme: vertical vorticity, vertical advection, Bernoulli head;
#   * ``Operators/vorticity_operators.jl:2-5`` -- circulation and vertical vorticity at (Face, Face, Center);
#   * ``Coriolis/hydrostatic_spherical_coriolis.jl:29-66`` -- ``HydrostaticSphericalCoriolis`` (both schemes), and
#     ``Coriolis/f_plane.jl:42-43`` -- ``FPlane``;
#   * ``Advection/tracer_advection_operators.jl:33-37`` with ``centered_advective_fluxes.jl:31-33`` -- flux-form tracer advection,
#     ``CenteredSecondOrder`` (the model's default);
#   * ``split_explicit_free_surface.jl:178-179`` -- the explicit barotropic pressure gradient is zero for this free surface;
#   * ``calculate_hydrostatic_free_surface_tendencies.jl:55-160`` -- every kernel runs over i = 1..Nx, j = 1..Ny, k = 1..Nz.
# Operators: difference / interpolation / derivative / metric-product operators of ``Operators/*.jl`` with their operand order.
OMEGA_EARTH = 7.292115e-5


class _Stencil:
    """views of parent arrays over the compute region, shifted by (di, dj, dk); metric rows as (1, Ny, 1), levels as (1, 1, Nz)"""

    def __init__(self, g):
        self.g = g
        az = g.ax[2]
        ks = np.arange(1 - g.Hz, g.Nz + g.Hz + 1)
        self.dzc = np.full(ks.size, az.dc) if az.regular else np.array([float(az.d_center(k)) if 1 - g.Hz <= k <= g.Nz + g.Hz else np.nan for k in ks])
        kf = np.arange(1 - g.Hz, g.Nz + g.Hz + 2)
        self.dzf = np.full(kf.size, az.df) if az.regular else np.array([float(az.d_face(k)) if 0 <= k + az.H < az._df.size else np.nan for k in kf])

    def S(self, a, di=0, dj=0, dk=0):
        g = self.g
        return a[g.Hx + di:g.Hx + g.Nx + di, g.Hy + dj:g.Hy + g.Ny + dj, g.Hz + dk:g.Hz + g.Nz + dk]

    def R(self, m, dj=0):
        g = self.g
        return m[g.Hy + dj:g.Hy + g.Ny + dj].reshape(1, -1, 1)

    def Zc(self, dk=0):
        g = self.g
        return self.dzc[g.Hz + dk:g.Hz + g.Nz + dk].reshape(1, 1, -1)

    def Zf(self, dk=0):
        g = self.g
        return self.dzf[g.Hz + dk:g.Hz + g.Nz + dk].reshape(1, 1, -1)


def coriolis_parameter_rows(grid, coriolis):
    """f at the rows of (Face, Face) points: 2 Omega sin(phi^f[j]) (hydrostatic_spherical_coriolis.jl:32-33), or the FPlane's f"""
    if coriolis[0] == "FPlane":
        return np.full(grid.Ny + 2 * grid.Hy + 1, float(coriolis[1]))
    if grid.phi_f is None:
        raise ValueError("HydrostaticSphericalCoriolis needs a LatitudeLongitudeGrid")
    return 2 * float(coriolis[1]) * np.sin(np.pi * grid.phi_f / 180)


def momentum_tendencies(st, momentum_advection="VectorInvariantEnstrophyConserving", coriolis=None):
    """G^n.u, G^n.v over the grid's cells.  momentum_advection: None | "VectorInvariantEnstrophyConserving" | "VectorInvariantEnergyConserving"
    | "WENOVectorInvariantVorticityStencil";
    coriolis: None | ("HydrostaticSphericalCoriolis", rotation_rate, "EnergyConserving" | "EnstrophyConserving") | ("FPlane", f)"""
    g = st.grid
    o = _Stencil(g)
    S, R = o.S, o.R
    u, v, w, p = st.u.data, st.v.data, st.w.data, st.pHY.data
    dxfc, dxcf, dyfc, dycf, azcc, azff = g.dx_fc, g.dx_cf, g.dy_fc, g.dy_cf, g.Az_cc, g.Az_ff
    azfc, azcf = azcc, azff                                   # regular x: Az^fc = Az^cc and Az^cf = Az^ff (latitude_longitude_grid.jl:442-445)

    def zeta(di=0, dj=0):                                     # zeta_3^ffc at (i + di, j + dj)
        circ = ((R(dycf, dj) * S(v, di, dj) - R(dycf, dj) * S(v, di - 1, dj))
                - (R(dxfc, dj) * S(u, di, dj) - R(dxfc, dj - 1) * S(u, di, dj - 1)))
        return circ / R(azff, dj)

    def Kh(di=0, dj=0):                                       # Kh^ccc at (i + di, j + dj)
        return (0.5 * (S(u, di, dj) ** 2 + S(u, di + 1, dj) ** 2) + 0.5 * (S(v, di, dj) ** 2 + S(v, di, dj + 1) ** 2)) / 2

    def Iy_dxv(di=0):                                         # I_y^c(dx_q^cfc v) at (i + di, j)
        return 0.5 * (R(dxcf, 0) * S(v, di, 0) + R(dxcf, 1) * S(v, di, 1))

    def Ix_dyu(dj=0):                                         # I_x^c(dy_q^fcc u) at (i, j + dj)
        return 0.5 * (R(dyfc, dj) * S(u, 0, dj) + R(dyfc, dj) * S(u, 1, dj))

    def Ix_dxv(dj=0):                                         # I_x^f(dx_q^cfc v) at (i, j + dj)
        return 0.5 * (R(dxcf, dj) * S(v, -1, dj) + R(dxcf, dj) * S(v, 0, dj))

    def Iy_dyu(di=0):                                         # I_y^f(dy_q^fcc u) at (i + di, j)
        return 0.5 * (R(dyfc, -1) * S(u, di, -1) + R(dyfc, 0) * S(u, di, 0))

    zero = np.zeros((g.Nx, g.Ny, g.Nz))
    if momentum_advection is None:
        Au, Av = zero, zero
    else:
        if momentum_advection == "VectorInvariantEnstrophyConserving":
            vvU = -(0.5 * (zeta(0, 0) + zeta(0, 1))) * (0.5 * (Iy_dxv(-1) + Iy_dxv(0))) / R(dxfc)
            vvV = +(0.5 * (zeta(0, 0) + zeta(1, 0))) * (0.5 * (Ix_dyu(-1) + Ix_dyu(0))) / R(dycf)
        elif momentum_advection == "VectorInvariantEnergyConserving":
            vvU = -(0.5 * (zeta(0, 0) * Ix_dxv(0) + zeta(0, 1) * Ix_dxv(1))) / R(dxfc)
            vvV = +(0.5 * (zeta(0, 0) * Iy_dyu(0) + zeta(1, 0) * Iy_dyu(1))) / R(dycf)
        elif momentum_advection == "WENOVectorInvariantVorticityStencil":
            # WENO5(vector_invariant = VorticityStencil()) (vector_invariant_advection.jl:54-66): the transporting velocity times the
            # upwind-biased WENO5 interpolation of zeta_3^ffc to the velocity point, smoothness measured on the vorticity itself
            # (weno_fifth_order.jl:380-403 through pass_stencil :475-476), second order inside the boundary buffer
            # (topologically_conditional_interpolation.jl:49-62); vertical advection and Bernoulli head as for VectorInvariant
            from . import advection as A
            adv = A.Advection(_SphereOps(g), A.WENO5())
            zf = lambda o: zeta(o[0], o[1])                                                    # noqa: E731   (o[2] = 0 throughout)
            up = lambda q, L, Rr: ((q + np.abs(q)) * L + (q - np.abs(q)) * Rr) / 2               # noqa: E731   upwind_biased_product
            vhat = (0.5 * (Iy_dxv(-1) + Iy_dxv(0))) / R(dxfc)
            uhat = (0.5 * (Ix_dyu(-1) + Ix_dyu(0))) / R(dycf)
            with np.errstate(all="ignore"):            # rows beyond the walls hold no metric: their stencils are the ones the buffer test discards
                vvU = -up(vhat, adv.leftC(1, zf)((0, 0, 0)), adv.rightC(1, zf)((0, 0, 0)))
                vvV = +up(uhat, adv.leftC(0, zf)((0, 0, 0)), adv.rightC(0, zf)((0, 0, 0)))
        else:
            raise ValueError(momentum_advection)

        def z2w(dk):                                          # zeta_2 w^fcf at level k + dk
            return (0.5 * (R(azcc) * S(w, -1, 0, dk) + R(azcc) * S(w, 0, 0, dk))) * ((S(u, 0, 0, dk) - S(u, 0, 0, dk - 1)) / o.Zf(dk))

        def z1w(dk):                                          # zeta_1 w^cff at level k + dk
            return (0.5 * (R(azcc, -1) * S(w, 0, -1, dk) + R(azcc, 0) * S(w, 0, 0, dk))) * ((S(v, 0, 0, dk) - S(v, 0, 0, dk - 1)) / o.Zf(dk))
        vaU = 0.5 * (z2w(0) + z2w(1)) / R(azfc)
        vaV = 0.5 * (z1w(0) + z1w(1)) / R(azcf)
        bhU = (Kh(0, 0) - Kh(-1, 0)) / R(dxfc)
        bhV = (Kh(0, 0) - Kh(0, -1)) / R(dycf)
        Au = (vvU + vaU) + bhU
        Av = (vvV + vaV) + bhV
    if coriolis is None:
        Cu, Cv = zero, zero
    else:
        f = coriolis_parameter_rows(g, coriolis)
        if coriolis[0] == "FPlane":
            f0 = float(coriolis[1])
            Cu = -f0 * (0.5 * (0.5 * (S(v, -1, 0) + S(v, 0, 0)) + 0.5 * (S(v, -1, 1) + S(v, 0, 1))))
            Cv = f0 * (0.5 * (0.5 * (S(u, 0, -1) + S(u, 1, -1)) + 0.5 * (S(u, 0, 0) + S(u, 1, 0))))
        elif coriolis[2] == "EnstrophyConserving":
            Cu = -(0.5 * (R(f, 0) + R(f, 1))) * (0.5 * (Iy_dxv(-1) + Iy_dxv(0))) / R(dxfc)
            Cv = +(0.5 * (R(f, 0) + R(f, 0))) * (0.5 * (Ix_dyu(-1) + Ix_dyu(0))) / R(dycf)
        elif coriolis[2] == "EnergyConserving":
            Cu = -(0.5 * (R(f, 0) * Ix_dxv(0) + R(f, 1) * Ix_dxv(1))) / R(dxfc)
            Cv = +(0.5 * (R(f, 0) * Iy_dyu(0) + R(f, 0) * Iy_dyu(1))) / R(dycf)
        else:
            raise ValueError(coriolis)
    px = (S(p, 0, 0) - S(p, -1, 0)) / R(dxfc)
    py = (S(p, 0, 0) - S(p, 0, -1)) / R(dycf)
    S(st.Gn["u"].data)[...] = ((-Au - 0) - Cu) - px
    S(st.Gn["v"].data)[...] = ((-Av - 0) - Cv) - py


class _SphereOps:
    """what oracle/advection.py's Advection needs of a grid, for the grids of split_explicit.py: offset functions over the compute
    region, areas and volumes with the per-row metrics (Ax^fcc = dy^fc dz, Ay^cfc = dx^cf[j] dz, Az^ccf = Az^cc[j], V^ccc = Az^cc[j] dz;
    spacings_and_areas_and_volumes.jl:203-232)"""

    def __init__(self, grid):
        from .operators import Ops
        self.st = _Stencil(grid)

        class G:                                    # the attributes Advection._cond / Ops.index read
            topo = grid.topo
            N = (grid.Nx, grid.Ny, grid.Nz)
        self.g = G
        self.flat = (False, False, False)
        self._ops = Ops
        self.grid = grid

    def dC(self, d, f):
        return self._ops.dC(self, d, f)

    def dF(self, d, f):
        return self._ops.dF(self, d, f)

    def iC(self, d, f):
        return self._ops.iC(self, d, f)

    def iF(self, d, f):
        return self._ops.iF(self, d, f)

    def index(self, d, o):
        return self._ops.index(self, d, o)

    def Ax(self, lz, o):
        return self.st.R(self.grid.dy_fc, o[1]) * self.st.Zc(o[2])

    def Ay(self, lz, o):
        return self.st.R(self.grid.dx_cf, o[1]) * self.st.Zc(o[2])

    def Az(self):
        return self.st.R(self.grid.Az_cc)

    def V(self, lz, o):
        return self.st.R(self.grid.Az_cc, o[1]) * self.st.Zc(o[2])

    def field(self, f):
        return lambda o: self.st.S(f.data, *o)


TRACER_SCHEMES = ("CenteredSecondOrder", "CenteredFourthOrder", "UpwindBiasedFifthOrder", "WENO5")


def tracer_tendency(st, name, tracer_advection="CenteredSecondOrder"):
    """G^n.c = -div_Uc over the grid's cells.  tracer_advection: None | "CenteredSecondOrder" (the model's default; written out below)
    | "CenteredFourthOrder" | "UpwindBiasedFifthOrder" | "WENO5" (Z weights, uniform coefficients) -- the flux-form operators of
    oracle/advection.py (tracer_advection_operators.jl:31-35, upwind_biased_advective_fluxes.jl:103-128,
    topologically_conditional_interpolation.jl:19-83) with this grid's areas and volumes"""
    g = st.grid
    o = _Stencil(g)
    S, R = o.S, o.R
    if tracer_advection is None:
        S(st.Gn[name].data)[...] = 0.0
        return
    if tracer_advection != "CenteredSecondOrder":
        from . import advection as A
        scheme = {"CenteredFourthOrder": A.CenteredFourthOrder, "UpwindBiasedFifthOrder": A.UpwindBiasedFifthOrder, "WENO5": A.WENO5}[tracer_advection]()
        ops = _SphereOps(g)
        adv = A.Advection(ops, scheme)
        div = adv.div_Uc(ops.field(st.u), ops.field(st.v), ops.field(st.w), ops.field(st.tracers[name]))((0, 0, 0))
        S(st.Gn[name].data)[...] = -div
        return
    u, v, w, c = st.u.data, st.v.data, st.w.data, st.tracers[name].data
    Fx = lambda di: ((R(g.dy_fc) * o.Zc()) * S(u, di)) * (0.5 * (S(c, di - 1) + S(c, di)))                    # noqa: E731
    Fy = lambda dj: ((R(g.dx_cf, dj) * o.Zc()) * S(v, 0, dj)) * (0.5 * (S(c, 0, dj - 1) + S(c, 0, dj)))       # noqa: E731
    Fz = lambda dk: (R(g.Az_cc) * S(w, 0, 0, dk)) * (0.5 * (S(c, 0, 0, dk - 1) + S(c, 0, 0, dk)))             # noqa: E731
    div = 1 / (R(g.Az_cc) * o.Zc()) * (((Fx(1) - Fx(0)) + (Fy(1) - Fy(0))) + (Fz(1) - Fz(0)))
    S(st.Gn[name].data)[...] = -div


def calculate_tendencies(st):
    """calculate_tendencies!(model) with st.momentum_advection, st.coriolis, st.tracer_advection"""
    momentum_tendencies(st, getattr(st, "momentum_advection", "VectorInvariantEnstrophyConserving"), getattr(st, "coriolis", None))
    for n in st.tracers:
        tracer_tendency(st, n, getattr(st, "tracer_advection", "CenteredSecondOrder"))


def time_step(st, dt, euler=False):
    """time_step!(model, dt; euler) (TimeSteppers/quasi_adams_bashforth_2.jl:70-104); the caller sets euler on the first step (the
    reference infers it from dt != previous dt as well)"""
    chi = -0.5 if euler else st.chi
    if euler:
        for f in st.Gm.values():
            f.data[...] = 0.0
    calculate_tendencies(st)
    time_step_after_tendencies(st, dt, chi)
