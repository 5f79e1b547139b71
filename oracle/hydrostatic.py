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
                 free_surface=None):
        self.grid = grid
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


def ab2_step(st, dt, chi):
    """ab2_step!(model::HydrostaticFreeSurfaceModel, dt, chi) (hydrostatic_free_surface_ab2_step.jl:15-48)"""
    fs = st.free_surface
    fs.barotropic_mode(fs.U, fs.V, st.u, st.v)             # local_ab2_step!: the barotropic mode of the velocities before the step
    for n in ("u", "v"):
        ab2_step_field(getattr(st, n), st.Gn[n], st.Gm[n], dt, chi)
    for n, c in st.tracers.items():
        ab2_step_field(c, st.Gn[n], st.Gm[n], dt, chi)
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
