"""NonhydrostaticModel + time steppers (oracle; test infrastructure only).

Restates the phase sequence of
  * ``TimeSteppers/quasi_adams_bashforth_2.jl:70-166`` (AB2; Euler when dt changes, chi = 0.1),
  * ``TimeSteppers/runge_kutta_3.jl:57-62,81-218`` (3-stage RK3, a projection per stage),
  * ``TimeSteppers/store_tendencies.jl:8-36``, ``TimeSteppers/clock.jl:48-60``,
  * ``Models/NonhydrostaticModels/nonhydrostatic_model.jl:102-203`` (constructor, halo inflation),
  * ``.../calculate_nonhydrostatic_tendencies.jl:12-200`` + ``nonhydrostatic_tendency_kernel_functions.jl:44-232``,
  * ``.../solve_for_pressure.jl:15-89``, ``pressure_correction.jl:10-56``,
  * ``.../update_nonhydrostatic_model_state.jl:14-37``, ``update_hydrostatic_pressure.jl:10-40``,
  * ``.../set_nonhydrostatic_model.jl:32-59``,
  * ``BoundaryConditions/apply_flux_bcs.jl:111-160``,
  * ``Coriolis/f_plane.jl:42-44``, ``BuoyancyModels/linear_equation_of_state.jl:69-77``,
    ``buoyancy_tracer.jl:12``, ``g_dot_b.jl:5-7``.
"""
import numpy as np

from .grid import RectilinearGrid, Periodic, Bounded, Flat, Center, Face  # noqa: F401
from .fields import Field, fill_halo_regions, FluxBC, ValueBC, GradientBC, SIDES  # noqa: F401
from .operators import Ops, sh
from .advection import (Advection, WENO5, CenteredSecondOrder, CenteredFourthOrder,  # noqa: F401
                        UpwindBiasedFifthOrder, UpwindBiasedFirstOrder, UpwindBiasedThirdOrder)
from .closures import ScalarDiffusivity, AnisotropicMinimumDissipation, Closure
from .poisson import FFTBasedPoissonSolver, FourierTridiagonalPoissonSolver

Z3 = (0, 0, 0)


class FPlane:
    def __init__(self, f):
        self.f = float(f)


class BuoyancyTracer:
    """buoyancy_tracer.jl:12: b = C.b"""
    tracers = ("b",)

    def perturbation(self, C):
        return C["b"]


class SeawaterBuoyancy:
    """seawater_buoyancy.jl + linear_equation_of_state.jl:69-71 (LinearEquationOfState only)."""
    tracers = ("T", "S")

    def __init__(self, gravitational_acceleration=9.80665, thermal_expansion=1.67e-4, haline_contraction=7.80e-4):
        self.g, self.alpha, self.beta = gravitational_acceleration, thermal_expansion, haline_contraction

    def perturbation(self, C):
        T, S = C["T"], C["S"]
        return lambda o: self.g * (self.alpha * T(o) - self.beta * S(o))


class NonhydrostaticModel:
    def __init__(self, grid, advection=None, buoyancy=None, coriolis=None, closure=None,
                 boundary_conditions=None, tracers=(), timestepper="QuasiAdamsBashforth2", chi=0.1):
        advection = CenteredSecondOrder() if advection is None else advection
        if isinstance(tracers, str):
            tracers = (tracers,)
        tracers = tuple(tracers)
        # halo inflation (nonhydrostatic_model.jl:140-148; Advection.jl:40; automatic_halo_sizing.jl:22-36)
        need = max(advection.buffer + 1, 1, (closure.required_halo if closure is not None else 1))
        req = tuple(0 if grid.topo[d] == Flat else max(grid.H[d], need) for d in range(3))
        if any(grid.H[d] < req[d] for d in range(3)):
            grid = grid.with_halo(req)
        self.grid = grid
        self.advection_scheme = advection
        self.ops = Ops(grid)
        self.adv = Advection(self.ops, advection)
        self.buoyancy, self.coriolis = buoyancy, coriolis
        self.closure = closure
        self.tracer_names = tracers
        bcs = boundary_conditions or {}
        self.u = Field(grid, (Face, Center, Center), bcs.get("u"))
        self.v = Field(grid, (Center, Face, Center), bcs.get("v"))
        self.w = Field(grid, (Center, Center, Face), bcs.get("w"))
        self.tracers = {n: Field(grid, (Center, Center, Center), bcs.get(n)) for n in tracers}
        self.pHY = None if grid.topo[2] == Flat else Field(grid, (Center, Center, Center))
        self.pNHS = Field(grid, (Center, Center, Center))
        names = ("u", "v", "w") + tracers
        locs = {"u": self.u.loc, "v": self.v.loc, "w": self.w.loc}
        self.Gn = {n: Field(grid, locs.get(n, (Center,) * 3)) for n in names}
        self.Gm = {n: Field(grid, locs.get(n, (Center,) * 3)) for n in names}
        self.closure_impl = Closure(self, closure, bcs)
        # PressureSolver (NonhydrostaticModels.jl:18-27)
        if grid.z_regular or grid.topo[2] == Flat:
            self.solver = FFTBasedPoissonSolver(grid)
        else:
            self.solver = FourierTridiagonalPoissonSolver(grid)
        self.timestepper = timestepper
        self.chi = float(chi)
        self.previous_dt = np.inf
        self.time, self.iteration, self.stage = 0.0, 0, 1
        update_state(self)

    def prognostic(self):
        d = {"u": self.u, "v": self.v, "w": self.w}
        d.update(self.tracers)
        return d


# ---------------------------------------------------------------------------------------------------
def update_state(m):
    """update_nonhydrostatic_model_state.jl:14-37."""
    fill_halo_regions([m.u, m.v, m.w] + list(m.tracers.values()))
    m.closure_impl.calculate_diffusivities()
    fill_halo_regions(m.closure_impl.diffusivity_fields())
    update_hydrostatic_pressure(m)
    if m.pHY is not None:
        fill_halo_regions(m.pHY)


def update_hydrostatic_pressure(m):
    """update_hydrostatic_pressure.jl:10-18 (skipped for Flat z :21)."""
    if m.pHY is None:
        return
    g = m.grid
    Nz = g.Nz
    p = m.pHY
    if m.buoyancy is None:
        b = lambda o: 0.0   # noqa: E731
    else:
        b = m.buoyancy.perturbation(m.tracers)
    H = g.Hz
    az = g.ax[2]
    # I_z^f(b) at faces k+1, k = 1..Nz   (b at k and k+1)
    if m.buoyancy is None:
        bf = np.zeros((g.Nx, g.Ny, Nz))
    else:
        bf = 0.5 * (b(Z3) + b((0, 0, 1)))

    def dzf(k):
        return az.df if az.regular else az.d_face(k)
    it = p.data[g.Hx:g.Hx + g.Nx, g.Hy:g.Hy + g.Ny, :]
    it[:, :, H + Nz - 1] = -bf[:, :, Nz - 1] * dzf(Nz + 1)
    for k in range(Nz - 1, 0, -1):
        it[:, :, H + k - 1] = it[:, :, H + k] - bf[:, :, k - 1] * dzf(k + 1)


def calculate_tendencies(m):
    """calculate_nonhydrostatic_tendencies.jl:12-36."""
    o_, adv, g = m.ops, m.adv, m.grid
    u, v, w = m.u, m.v, m.w
    cl = m.closure_impl
    # hydrostatic pressure gradient (nonhydrostatic_tendency_kernel_functions.jl:10-15)
    if m.pHY is not None:
        px, py = o_.ddF(0, m.pHY), o_.ddF(1, m.pHY)
    else:
        px = py = lambda o: 0.0   # noqa: E731
    if m.coriolis is not None:
        f = m.coriolis.f
        fx = lambda o: -f * o_.iC(1, o_.iF(0, v))(o)     # noqa: E731  -f * I_xy^{fc}(v)   f_plane.jl:42
        fy = lambda o: f * o_.iF(1, o_.iC(0, u))(o)      # noqa: E731  +f * I_xy^{cf}(u)   f_plane.jl:43
    else:
        fx = fy = lambda o: 0.0   # noqa: E731
    adv_on = m.advection_scheme is not None
    Gu = (- (adv.div_Uu(u, v, w, u)(Z3) if adv_on else 0.0) - fx(Z3) - px(Z3) - cl.div_tau(0)(Z3))
    Gv = (- (adv.div_Uv(u, v, w, v)(Z3) if adv_on else 0.0) - fy(Z3) - py(Z3) - cl.div_tau(1)(Z3))
    Gw = (- (adv.div_Uw(u, v, w, w)(Z3) if adv_on else 0.0) - cl.div_tau(2)(Z3))
    m.Gn["u"]()[...] = Gu
    m.Gn["v"]()[...] = Gv
    m.Gn["w"]()[...] = Gw
    for n, c in m.tracers.items():
        Gc = - (adv.div_Uc(u, v, w, c)(Z3) if adv_on else 0.0) - cl.div_q(n)(Z3)
        m.Gn[n]()[...] = Gc
    # boundary contributions (apply_flux_bcs.jl:111-160)
    for n, fld in m.prognostic().items():
        _apply_flux_bcs(m, m.Gn[n], fld)


def _apply_flux_bcs(m, G, fld):
    g = m.grid
    for d in range(3):
        if g.topo[d] != Bounded:
            continue
        for side_i, side in enumerate(SIDES[d]):
            bc = fld.bcs[side]
            if bc is None or bc.kind != "flux" or bc.condition is None:
                continue
            N, H = g.N[d], g.H[d]
            idx = [slice(g.H[e], g.H[e] + g.N[e]) if g.topo[e] != Flat else slice(None) for e in range(3)]
            iI = 1 if side_i == 0 else N
            iB = 1 if side_i == 0 else N + 1
            idx[d] = iI - 1 + H
            cond = bc.condition
            if not np.isscalar(cond):
                cond = np.asarray(cond, dtype=np.float64).reshape([g.N[e] for e in range(3) if e != d])
            if side_i == 0:
                G.data[tuple(idx)] += cond * _area_over_volume(m, fld, d, iB, iI)
            else:
                G.data[tuple(idx)] -= cond * _area_over_volume(m, fld, d, iB, iI)


def _area_over_volume(m, fld, d, iB, iI):
    """A(iB at flip(loc_d)) / V(iI at loc).  For x/y-regular grids only z spacings can differ."""
    g = m.grid
    az = g.ax[2]
    lz = fld.loc[2]

    def dz(loc, k):
        if g.topo[2] == Flat or az.regular:
            return az.dc
        return az.d_center(k) if loc == Center else az.d_face(k)
    if d == 2:
        A = g.dx * g.dy
        V = g.dx * g.dy * dz(lz, iI)
        return A / V
    # x or y boundary: area = other horizontal spacing * dz(lz, k) (per k) -> ratio = 1/d(dir)
    k = np.arange(1, g.Nz + 1)
    dzk = dz(lz, k)
    if d == 0:
        A = g.dy * dzk
        V = g.dx * g.dy * dzk
    else:
        A = g.dx * dzk
        V = g.dx * g.dy * dzk
    r = A / V
    return r if np.isscalar(r) else r.reshape(1, -1)


def ab2_step(m, dt, chi):
    """quasi_adams_bashforth_2.jl:116-166."""
    for n, fld in m.prognostic().items():
        U = fld()
        U += dt * ((1.5 + chi) * m.Gn[n]() - (0.5 + chi) * m.Gm[n]())


def rk3_substep(m, dt, gamma, zeta):
    """runge_kutta_3.jl:204-218."""
    for n, fld in m.prognostic().items():
        U = fld()
        if zeta is None:
            U += dt * gamma * m.Gn[n]()
        else:
            U += dt * (gamma * m.Gn[n]() + zeta * m.Gm[n]())


def store_tendencies(m):
    for n in m.Gn:
        m.Gm[n]()[...] = m.Gn[n]()


def solve_for_pressure(m, dt):
    """solve_for_pressure.jl:15-89."""
    o_, g = m.ops, m.grid
    div = o_.div_ccc(m.u, m.v, m.w)(Z3)
    if isinstance(m.solver, FFTBasedPoissonSolver):
        rhs = div / dt
        phi = m.solver.solve(rhs)
    else:
        rhs = o_.dz(Center, Z3) * div / dt
        phi = m.solver.solve(rhs)
    m.pNHS()[...] = phi


def calculate_pressure_correction(m, dt):
    """pressure_correction.jl:10-23."""
    fill_halo_regions([m.u, m.v, m.w])
    solve_for_pressure(m, dt)
    fill_halo_regions(m.pNHS)


def pressure_correct_velocities(m, dt):
    """pressure_correction.jl:34-40."""
    o_ = m.ops
    p = m.pNHS
    gx, gy, gz = o_.ddF(0, p)(Z3), o_.ddF(1, p)(Z3), o_.ddF(2, p)(Z3)
    m.u()[...] -= gx * dt
    m.v()[...] -= gy * dt
    m.w()[...] -= gz * dt


def tick(m, dt, stage=False):
    m.time += dt
    if stage:
        m.stage += 1
    else:
        m.iteration += 1
        m.stage = 1


def time_step(m, dt, euler=False):
    if m.timestepper in ("QuasiAdamsBashforth2", "AB2"):
        return _time_step_ab2(m, dt, euler)
    return _time_step_rk3(m, dt)


def _time_step_ab2(m, dt, euler=False):
    """quasi_adams_bashforth_2.jl:70-104."""
    euler = euler or (dt != m.previous_dt)
    chi = -0.5 if euler else m.chi
    if euler:
        for f in m.Gm.values():
            f.data[...] = 0
    m.previous_dt = dt
    if m.iteration == 0:
        update_state(m)
    calculate_tendencies(m)
    ab2_step(m, dt, chi)
    calculate_pressure_correction(m, dt)
    pressure_correct_velocities(m, dt)
    store_tendencies(m)
    tick(m, dt)
    update_state(m)


def _time_step_rk3(m, dt):
    """runge_kutta_3.jl:81-152."""
    if m.iteration == 0:
        update_state(m)
    g1, g2, g3 = 8 / 15, 5 / 12, 3 / 4
    z2, z3 = -17 / 60, -5 / 12
    dt1, dt2, dt3 = g1 * dt, (g2 + z2) * dt, (g3 + z3) * dt
    for (gam, zet, sdt, last) in ((g1, None, dt1, False), (g2, z2, dt2, False), (g3, z3, dt3, True)):
        calculate_tendencies(m)
        rk3_substep(m, dt, gam, zet)
        calculate_pressure_correction(m, sdt)
        pressure_correct_velocities(m, sdt)
        tick(m, sdt, stage=not last)
        if not last:
            store_tendencies(m)
        update_state(m)


def set_model(m, enforce_incompressibility=True, **kw):
    """set_nonhydrostatic_model.jl:32-59."""
    pf = m.prognostic()
    for n, val in kw.items():
        if n not in pf:
            raise ValueError(f"name {n} not found in model.velocities or model.tracers.")
        pf[n].set(val)
    update_state(m)
    if enforce_incompressibility:
        calculate_pressure_correction(m, 1.0)
        pressure_correct_velocities(m, 1.0)
        update_state(m)
