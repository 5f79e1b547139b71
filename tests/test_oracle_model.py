"""Pins the oracle's model path with the reference's own property / known-answer tests:
test/test_halo_regions.jl:1-41, test/test_time_stepping.jl:81-146,154-188,346-374,
test/test_dynamics.jl:170-258, test/test_nonhydrostatic_models.jl:55-60,
validation/convergence_tests/one_dimensional_advection_schemes.jl:41-120."""
import numpy as np
import pytest

import oracle as O
from oracle.fields import Field, fill_halo_regions
from oracle.model import update_state

P, B, F = O.Periodic, O.Bounded, O.Flat
Z3 = (0, 0, 0)


def max_div(m):
    return np.abs(m.ops.div_ccc(m.u, m.v, m.w)(Z3)).max()


# ---- halos (test_halo_regions.jl) ----------------------------------------------------------------
@pytest.mark.parametrize("N", [(1, 1, 1), (1, 8, 8), (8, 1, 8), (8, 8, 1), (8, 8, 8), (5, 7, 9)])
def test_halo_regions_initialized_and_filled(N):
    rng = np.random.default_rng(0)
    g = O.RectilinearGrid(size=N, extent=(100, 200, 300), halo=(1, 1, 1), topology=(P, P, B))
    f = Field(g, (O.Center,) * 3)
    f.set(rng.random(N))
    d = f.data
    assert (d[0] == 0).all() and (d[-1] == 0).all() and (d[:, 0] == 0).all() and (d[:, -1] == 0).all()
    assert (d[:, :, 0] == 0).all() and (d[:, :, -1] == 0).all()
    fill_halo_regions(f)
    Nx, Ny, Nz = N
    it = slice(1, -1)
    assert (d[0, it, it] == d[Nx, it, it]).all() and (d[Nx + 1, it, it] == d[1, it, it]).all()
    assert (d[it, 0, it] == d[it, Ny, it]).all() and (d[it, Ny + 1, it] == d[it, 1, it]).all()
    assert (d[it, it, 0] == d[it, it, 1]).all() and (d[it, it, Nz + 1] == d[it, it, Nz]).all()


def test_value_gradient_open_fills():
    g = O.RectilinearGrid(size=(4, 4, 4), extent=(1, 1, 2), topology=(P, P, B))
    c = Field(g, (O.Center,) * 3, {"top": O.ValueBC(3.0), "bottom": O.GradientBC(0.5)})
    c.set(np.arange(64, dtype=float).reshape(4, 4, 4))
    fill_halo_regions(c)
    H, dz = 3, 0.5
    it = c.data[3:-3, 3:-3]
    assert np.allclose(it[:, :, H + 4], it[:, :, H + 3] + 2 * (3.0 - it[:, :, H + 3]))     # value: mirror about the wall
    assert np.allclose(it[:, :, H - 1], it[:, :, H] - 0.5 * dz)                            # gradient
    w = Field(g, (O.Center, O.Center, O.Face))
    w.data[...] = 1.0
    fill_halo_regions(w)
    assert (w.data[3:-3, 3:-3, H] == 0).all() and (w.data[3:-3, 3:-3, H + 4] == 0).all()   # impenetrable faces


# ---- model construction -------------------------------------------------------------------------
def test_halo_inflation_for_weno5():
    """test_nonhydrostatic_models.jl:55-60."""
    g = O.RectilinearGrid(size=(4, 4, 4), extent=(1, 1, 1), halo=(1, 1, 1), topology=(P, P, P))
    for scheme in (O.WENO5(), O.UpwindBiasedFifthOrder()):
        m = O.NonhydrostaticModel(g, advection=scheme)
        assert m.grid.H == (3, 3, 3)
    m = O.NonhydrostaticModel(g, advection=O.CenteredFourthOrder())
    assert m.grid.H == (2, 2, 2)


# ---- AB2 (test_time_stepping.jl:81-105) ---------------------------------------------------------
def test_first_ab2_step_is_euler():
    g = O.RectilinearGrid(size=(13, 17, 19), extent=(1, 2, 3))
    rng = np.random.default_rng(1)
    u0 = rng.random((13, 17, 19)) * 0.1

    def run(euler):
        m = O.NonhydrostaticModel(g, buoyancy=O.SeawaterBuoyancy(), tracers=("T", "S"))
        O.set_model(m, u=u0, T=lambda x, y, z: 1 + 0 * z)
        m.Gm["u"].data[...] = np.nan if euler else 0.0     # Euler must zero G^- first (qab2.jl:82-84)
        O.time_step(m, 1.0, euler=euler)
        return m
    a, b = run(True), run(False)   # previous_dt = Inf  =>  both are Euler steps
    assert np.array_equal(a.u.data, b.u.data) and np.isfinite(a.u.data).all()
    assert np.allclose(a.tracers["S"].interior(), 0)
    # explicit Euler check: u1 = P(u0 + dt * G(u0))
    m = O.NonhydrostaticModel(g, buoyancy=O.SeawaterBuoyancy(), tracers=("T", "S"))
    O.set_model(m, u=u0, T=lambda x, y, z: 1 + 0 * z)
    O.time_step(m, 0.25)
    G0 = m.Gm["u"].interior().copy()
    O.time_step(m, 0.25)
    # second step with the same dt is a genuine AB2 step: G^- holds the first tendency
    assert np.array_equal(m.Gm["u"].interior(), m.Gn["u"].interior())
    assert not np.array_equal(G0, m.Gn["u"].interior())


# ---- incompressibility (test_time_stepping.jl:112-146,346-374) ---------------------------------
def _grids32():
    N = 32
    yield O.RectilinearGrid(size=(N, N, N), x=(0, 1), y=(0, 1), z=(-1, 1))
    S = 1.3
    yield O.RectilinearGrid(size=(N, N, N), x=(0, 1), y=(0, 1),
                            z=lambda k: np.tanh(S * (2 * (k - 1) / N - 1)) / np.tanh(S))
    yield O.RectilinearGrid(size=(N, N, N), x=(0, 1), y=(0, 1), z=np.linspace(0, 1, N + 1))


@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
@pytest.mark.parametrize("Nt", [1, 10])
def test_incompressible_in_time(stepper, Nt):
    for g in _grids32():
        m = O.NonhydrostaticModel(g, timestepper=stepper, buoyancy=O.SeawaterBuoyancy(), tracers=("T", "S"))
        m.tracers["T"].interior()[7:24, 7:24, 7:24] += 0.01
        update_state(m)
        for _ in range(Nt):
            O.time_step(m, 0.05)
        assert np.abs(m.w.interior()).max() > 0
        assert max_div(m) < 5e-8


def test_incompressible_weno5_periodic_100_steps():
    g = O.RectilinearGrid(size=(16, 16, 16), extent=(1, 1, 1), topology=(P, P, P))
    m = O.NonhydrostaticModel(g, advection=O.WENO5())
    rng = np.random.default_rng(2)
    O.set_model(m, u=rng.random(g.N) - 0.5, v=rng.random(g.N) - 0.5, w=rng.random(g.N) - 0.5)
    for _ in range(100):
        O.time_step(m, 2e-3)
    assert max_div(m) < 5e-8 and np.isfinite(m.u.data).all()


# The same invariants on grids with walls in x / y and on 2-D slices (the topologies of test_poisson_solvers.jl:8-9
# and test_flat_dimensions): projection leaves max|div U| at round-off, impenetrable walls stay closed, no-flux
# walls conserve tracers.  These pin the oracle's Bounded / Flat x, y code paths used by the parity cases.
@pytest.mark.parametrize("topo", [(B, B, B), (P, B, B), (B, P, P), (B, "Flat", B), (B, B, "Flat"), ("Flat", P, B)])
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_incompressible_with_walls_and_slices(topo, stepper):
    N = 10
    size = tuple(N for t in topo if t != "Flat")
    g = O.RectilinearGrid(size=size, extent=tuple(1.0 for _ in size), topology=topo)
    m = O.NonhydrostaticModel(g, timestepper=stepper, advection=O.WENO5(), tracers=("c",),
                              closure=O.ScalarDiffusivity(nu=1e-3, kappa=1e-3))
    rng = np.random.default_rng(5)
    O.set_model(m, u=0.2 * (rng.random(m.u.interior().shape) - 0.5), v=0.2 * (rng.random(m.v.interior().shape) - 0.5),
                c=rng.random(m.tracers["c"].interior().shape))
    c0 = m.tracers["c"].interior().sum()
    for _ in range(3):
        O.time_step(m, 2e-3)
    assert max_div(m) < 5e-8
    if topo[0] == B:
        assert (m.u.interior()[0] == 0).all() and (m.u.interior()[-1] == 0).all()
    if topo[1] == B:
        assert (m.v.interior()[:, 0] == 0).all() and (m.v.interior()[:, -1] == 0).all()
    assert abs(m.tracers["c"].interior().sum() - c0) < 1e-10 * abs(c0)


# ---- tracer conservation (test_time_stepping.jl:154-188; isotropic diffusivity variant) -------
def test_tracer_conserved():
    Nx, Ny, Nz = 16, 32, 16
    g = O.RectilinearGrid(size=(Nx, Ny, Nz), extent=(160e3, 320e3, 1024))
    m = O.NonhydrostaticModel(g, closure=O.ScalarDiffusivity(nu=1.28, kappa=1.28),
                              buoyancy=O.SeawaterBuoyancy(), tracers=("T", "S"))
    rng = np.random.default_rng(4)
    O.set_model(m, T=lambda x, y, z: 10 + 1e-4 * y + 5e-3 * z + 1e-4 * rng.random((Nx, Ny, Nz)))
    T0 = m.tracers["T"].interior().mean()
    for _ in range(10):
        O.time_step(m, 600.0)
    assert abs(m.tracers["T"].interior().mean() - T0) <= Nx * Ny * Nz * np.finfo(float).eps


# ---- Taylor-Green (test_dynamics.jl:210-258) ----------------------------------------------------
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_taylor_green_vortex(stepper):
    N, Nt, nu = 64, 10, 1.0
    g = O.RectilinearGrid(size=(N, N, 2), extent=(1, 1, 1))
    dt = (1 / (10 * np.pi)) * (1 / N) ** 2 / nu
    m = O.NonhydrostaticModel(g, timestepper=stepper, closure=O.ScalarDiffusivity(nu=nu))
    O.set_model(m, u=lambda x, y, z: -np.sin(2 * np.pi * y), v=lambda x, y, z: np.sin(2 * np.pi * x))
    for _ in range(Nt):
        O.time_step(m, dt)
    t = m.time
    decay = np.exp(-4 * np.pi ** 2 * nu * t)
    xF, yC = g.xnodes(O.Face).reshape(-1, 1, 1), g.ynodes(O.Center).reshape(1, -1, 1)
    xC, yF = g.xnodes(O.Center).reshape(-1, 1, 1), g.ynodes(O.Face).reshape(1, -1, 1)
    ua = -np.sin(2 * np.pi * yC) * decay + 0 * xF
    va = np.sin(2 * np.pi * xC) * decay + 0 * yF
    assert np.abs((m.u.interior() - ua) / ua).max() < 5e-6
    assert np.abs((m.v.interior() - va) / va).max() < 5e-6


# ---- Gaussian tracer advection (test_dynamics.jl:170-202) ---------------------------------------
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_passive_tracer_advection(stepper):
    N, Nt, kap = 128, 100, 1e-12
    L, U, V = 1.0, 0.5, 0.8
    dl, x0, y0 = L / 15, L / 2, L / 2
    dt = 0.05 * L / N / np.sqrt(U ** 2 + V ** 2)

    def T(x, y, t):
        return np.exp(-((x - U * t - x0) ** 2 + (y - V * t - y0) ** 2) / (2 * dl ** 2))
    g = O.RectilinearGrid(size=(N, N, 2), extent=(L, L, L))
    m = O.NonhydrostaticModel(g, closure=O.ScalarDiffusivity(nu=kap, kappa=kap), timestepper=stepper,
                              buoyancy=O.SeawaterBuoyancy(), tracers=("T", "S"))
    O.set_model(m, u=U, v=V, T=lambda x, y, z: T(x, y, 0) + 0 * z)
    for _ in range(Nt):
        O.time_step(m, dt)
    x, y = g.xnodes(O.Center).reshape(-1, 1, 1), g.ynodes(O.Center).reshape(1, -1, 1)
    Ta = T(x, y, m.time) + np.zeros((1, 1, 2))
    rel = np.mean((m.tracers["T"].interior() - Ta) ** 2) / np.mean(Ta ** 2)     # relative_error (utils_for_runtests)
    assert rel < 1e-4


# WENO5 / U5 / C4 / C2 / U3 convergence orders (one_dimensional_advection_schemes.jl): tests/test_reference_convergence.py,
# on the oracle AND through the library.


# ---- AMD closure: vanishes for laminar flows (pure strain, uniform shear), positive for a turbulent-like field --
def test_amd_predictor_properties():
    g = O.RectilinearGrid(size=(8, 8, 8), extent=(1, 1, 1), topology=(P, P, B))
    m = O.NonhydrostaticModel(g, closure=O.AnisotropicMinimumDissipation(), tracers=("b",), buoyancy=O.BuoyancyTracer())
    # uniform shear in z of u (periodic in x, y): du/dz = 1  -> r = 0 -> nu_e = 0
    m.u.set(lambda x, y, z: z + 0 * x)
    update_state(m)
    assert np.abs(m.closure_impl.nu_e.interior()[:, :, 1:-1]).max() < 1e-12
    rng = np.random.default_rng(0)
    O.set_model(m, u=rng.random((8, 8, 8)) - 0.5, v=rng.random((8, 8, 8)) - 0.5, b=rng.random((8, 8, 8)))
    nu = m.closure_impl.nu_e.interior()
    assert (nu >= 0).all() and nu.max() > 0 and np.isfinite(nu).all()
    kap = m.closure_impl.kappa_e["b"].interior()
    assert (kap >= 0).all() and kap.max() > 0
