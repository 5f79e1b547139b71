"""The reference's own property / known-answer tests driven through the C ABI (host emulation of the kernel
sources on CPU, HIP on the GPU box): test/test_time_stepping.jl:112-146 (incompressibility), test/test_dynamics.jl:
170-258 (Taylor-Green decay, Gaussian tracer advection), test/test_halo_regions.jl:22-41,
test/test_poisson_solvers*.jl (lap(phi) == R), test_nonhydrostatic_models.jl:55-60 (halo inflation)."""
import numpy as np
import pytest

P, B = "Periodic", "Bounded"


def _run(ocn, backend, marker_gpu):
    if marker_gpu and backend != "gpu":
        pytest.skip("HIP run only")
    if not marker_gpu and backend != "hostemu":
        pytest.skip("host-emulation run only")


def _incompressible(ocn, N, stepper, Nt):
    S = 1.3
    grids = [dict(x=(0, 1), y=(0, 1), z=(-1.0, 1.0)),
             dict(x=(0, 1), y=(0, 1), z=lambda k: np.tanh(S * (2 * (k - 1) / N - 1)) / np.tanh(S)),
             dict(x=(0, 1), y=(0, 1), z=np.linspace(0, 1, N + 1))]
    for kw in grids:
        g = ocn.RectilinearGrid(size=(N, N, N), **kw)
        m = ocn.NonhydrostaticModel(g, timestepper=stepper, buoyancy=ocn.SeawaterBuoyancy(), tracers=("T", "S"))
        T = np.zeros((N, N, N))
        a, b = N // 4, 3 * N // 4
        T[a:b, a:b, a:b] += 0.01
        ocn.set_model(m, T=T, enforce_incompressibility=False)
        for _ in range(Nt):
            ocn.time_step(m, 0.05)
        assert np.abs(m.w.interior()).max() > 0
        assert m.max_abs_divergence() < 5e-8


def _taylor_green(ocn, N, stepper):
    nu, Nt = 1.0, 10
    g = ocn.RectilinearGrid(size=(N, N, 2), extent=(1, 1, 1))
    dt = (1 / (10 * np.pi)) * (1 / N) ** 2 / nu
    m = ocn.NonhydrostaticModel(g, timestepper=stepper, closure=ocn.ScalarDiffusivity(nu=nu))
    ocn.set_model(m, u=lambda x, y, z: -np.sin(2 * np.pi * y), v=lambda x, y, z: np.sin(2 * np.pi * x))
    for _ in range(Nt):
        ocn.time_step(m, dt)
    decay = np.exp(-4 * np.pi ** 2 * nu * m.time)
    X, Y, _ = m.nodes("u")
    ua = -np.sin(2 * np.pi * Y) * decay + 0 * X
    X, Y, _ = m.nodes("v")
    va = np.sin(2 * np.pi * X) * decay + 0 * Y
    assert np.abs((m.u.interior() - ua) / ua).max() < 5e-6
    assert np.abs((m.v.interior() - va) / va).max() < 5e-6


def _tracer_advection(ocn, N, stepper):
    Nt, kap = 100, 1e-12
    L, U, V = 1.0, 0.5, 0.8
    dl, x0, y0 = L / 15, L / 2, L / 2
    dt = 0.05 * L / N / np.sqrt(U ** 2 + V ** 2)

    def T(x, y, t):
        return np.exp(-((x - U * t - x0) ** 2 + (y - V * t - y0) ** 2) / (2 * dl ** 2))
    g = ocn.RectilinearGrid(size=(N, N, 2), extent=(L, L, L))
    m = ocn.NonhydrostaticModel(g, closure=ocn.ScalarDiffusivity(nu=kap, kappa=kap), timestepper=stepper,
                                buoyancy=ocn.SeawaterBuoyancy(), tracers=("T", "S"))
    ocn.set_model(m, u=U, v=V, T=lambda x, y, z: T(x, y, 0) + 0 * z)
    for _ in range(Nt):
        ocn.time_step(m, dt)
    X, Y, _ = m.nodes("T")
    Ta = T(X, Y, m.time) + np.zeros((1, 1, 2))
    assert np.mean((m.tracers["T"].interior() - Ta) ** 2) / np.mean(Ta ** 2) < 1e-4


def _halos(ocn):
    N = (5, 7, 9)
    rng = np.random.default_rng(0)
    g = ocn.RectilinearGrid(size=N, extent=(100, 200, 300), halo=(1, 1, 1), topology=(P, P, B))
    m = ocn.NonhydrostaticModel(g, tracers=("c",))
    assert m.halo == (1, 1, 1)
    f = m.tracers["c"]
    f.set(rng.random(N))
    d = f.parent()
    assert (d[0] == 0).all() and (d[-1] == 0).all() and (d[:, :, 0] == 0).all()     # halos untouched by set!
    ocn.update_state(m)
    d = f.parent()
    it = slice(1, -1)
    Nx, Ny, Nz = N
    assert (d[0, it, it] == d[Nx, it, it]).all() and (d[Nx + 1, it, it] == d[1, it, it]).all()
    assert (d[it, 0, it] == d[it, Ny, it]).all() and (d[it, Ny + 1, it] == d[it, 1, it]).all()
    assert (d[it, it, 0] == d[it, it, 1]).all() and (d[it, it, Nz + 1] == d[it, it, Nz]).all()
    m2 = ocn.NonhydrostaticModel(ocn.RectilinearGrid(size=N, extent=(1, 1, 1), halo=(1, 1, 1), topology=(P, P, B)),
                                 advection=ocn.WENO5())
    assert m2.halo == (3, 3, 3)                                                  # halo inflation


def _halos_degenerate(ocn):
    """test_halo_regions.jl:22-41 sizes, including directions with fewer cells than halo points (N = 1 with H = 1 and
    with the WENO halo of 3: the periodic fill then feeds on halo cells it has just written), all six topologies
    of z x {Periodic, Bounded} with walls in x / y: the library's parent arrays equal the oracle's, element by element."""
    import oracle as O
    rng = np.random.default_rng(0)
    for N in [(1, 1, 1), (1, 8, 8), (8, 1, 8), (8, 8, 1), (2, 3, 2), (5, 7, 9)]:
        for topo in [(P, P, B), (P, P, P), (B, P, B), (P, B, P), (B, B, B)]:
            for adv in (None, "WENO5"):
                kw = dict(size=N, extent=(100, 200, 300), halo=(1, 1, 1), topology=topo)
                m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), tracers=("c",), advection=ocn.WENO5() if adv else None)
                om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), tracers=("c",), advection=O.WENO5() if adv else None)
                vals = {n: rng.random(f.interior().shape) for n, f in om.prognostic().items()}
                for n, v in vals.items():
                    m.prognostic()[n].set(v)
                    om.prognostic()[n].set(v)
                ocn.update_state(m)
                from oracle.model import update_state as o_update_state
                o_update_state(om)
                for n in vals:
                    assert np.array_equal(m.prognostic()[n].parent(), om.prognostic()[n].data), (N, topo, adv, n)


def _poisson(ocn):
    import oracle as O
    from oracle.fields import Field, fill_halo_regions
    from oracle.operators import Ops
    rng = np.random.default_rng(3)
    faces = np.array([1, 2, 4, 7, 11, 16, 22, 29, 37.0])
    cases = [((16, 11, 7), dict(extent=(1, 1, 1), topology=(P, P, P))),
             ((11, 16, 7), dict(extent=(1, 1, 1), topology=(P, P, B))),
             ((8, 7, 8), dict(x=(0, 1), y=(0, 1), z=faces, topology=(P, P, B)))]
    for N, kw in cases:
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(size=N, **kw))
        R = rng.random(N)
        R -= R.mean()
        if "z" in kw:    # volume-weighted compatibility on the stretched grid
            dz = np.diff(faces).reshape(1, 1, -1)
            R -= (R * dz).sum() / (dz.sum() * N[0] * N[1])
        phi = m.poisson_solve(R)
        og = O.RectilinearGrid(size=N, **kw)
        f = Field(og, (O.Center,) * 3)
        f.set(phi)
        fill_halo_regions(f)
        assert np.allclose(Ops(og).laplacian_ccc(f)((0, 0, 0)), R, rtol=1.5e-8, atol=1e-9)


def _poisson_all_topologies(ocn, sizes):
    """test_poisson_solvers.jl:8-9,45-85: every (x, y, z) in {Periodic, Bounded}^3 (+ Flat slices, + stretched z
    under Bounded x / y): lap(phi) == R, and phi equals the oracle's solution (same zero-mean gauge)."""
    import itertools
    import oracle as O
    from oracle.fields import Field, fill_halo_regions
    from oracle.operators import Ops
    from oracle.poisson import FFTBasedPoissonSolver, FourierTridiagonalPoissonSolver
    F = "Flat"
    rng = np.random.default_rng(11)
    topos = list(itertools.product((P, B), repeat=3)) + [(B, F, B), (F, P, B), (P, F, P), (B, B, F), (F, B, F)]
    worst = 0.0
    for topo in topos:
        for N3 in sizes:
            N = tuple(n for n, t in zip(N3, topo) if t != F)
            kw = dict(extent=tuple(1.0 + 0.5 * a for a, t in enumerate(topo) if t != F), topology=topo)
            m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(size=N, **kw))
            og = O.RectilinearGrid(size=N, **kw)
            full = tuple(1 if t == F else n for n, t in zip(N3, topo))
            R = rng.random(full)
            R -= R.mean()
            phi = m.poisson_solve(R)
            ref = FFTBasedPoissonSolver(og).solve(R)
            worst = max(worst, np.abs(phi - ref).max() / np.abs(ref).max())
            f = Field(og, (O.Center,) * 3)
            f.set(phi)
            fill_halo_regions(f)
            assert np.allclose(Ops(og).laplacian_ccc(f)((0, 0, 0)), R, rtol=1.5e-8, atol=1e-9), (topo, N)
    assert worst < 1e-11, worst
    # Fourier-tridiagonal with walls in x / y (fourier_tridiagonal_poisson_solver.jl with Bounded x, y)
    faces = np.array([1, 2, 4, 7, 11, 16, 22, 29, 37.0])
    for topo in [(B, B, B), (P, B, B), (B, P, B)]:
        N = (7, 6, 8)
        kw = dict(x=(0, 1), y=(0, 2), z=faces, topology=topo)
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(size=N, **kw))
        og = O.RectilinearGrid(size=N, **kw)
        R = rng.random(N)
        dz = np.diff(faces).reshape(1, 1, -1)
        R -= (R * dz).sum() / (dz.sum() * N[0] * N[1])
        phi = m.poisson_solve(R)
        ref = FourierTridiagonalPoissonSolver(og).solve_source(R)
        assert np.abs(phi - ref).max() / np.abs(ref).max() < 1e-11
        f = Field(og, (O.Center,) * 3)
        f.set(phi)
        fill_halo_regions(f)
        assert np.allclose(Ops(og).laplacian_ccc(f)((0, 0, 0)), R, rtol=1.5e-8, atol=1e-9), topo


def _incompressible_walls(ocn, N, stepper, Nt):
    """test_time_stepping.jl:112-146 transplanted to grids with walls in x / y and to 2-D slices."""
    F = "Flat"
    for topo in [(B, B, B), (P, B, B), (B, P, P), (B, F, B), (B, B, F)]:
        size = tuple(N for t in topo if t != F)
        g = ocn.RectilinearGrid(size=size, extent=tuple(1.0 for _ in size), topology=topo)
        m = ocn.NonhydrostaticModel(g, timestepper=stepper, advection=ocn.WENO5(), tracers=("c",),
                                    closure=ocn.ScalarDiffusivity(nu=1e-3, kappa=1e-3))
        rng = np.random.default_rng(5)
        ocn.set_model(m, u=0.2 * (rng.random(m.u.size) - 0.5), v=0.2 * (rng.random(m.v.size) - 0.5),
                      c=rng.random(m.tracers["c"].size))
        c0 = m.tracers["c"].interior().sum()
        for _ in range(Nt):
            ocn.time_step(m, 0.02 / N)
        assert m.max_abs_divergence() < 5e-8
        u = m.u.interior()
        if topo[0] == B:
            assert (u[0] == 0).all() and (u[-1] == 0).all()              # impenetrable walls
        assert abs(m.tracers["c"].interior().sum() - c0) < 1e-10 * abs(c0)   # no-flux walls conserve the tracer


# ---- CPU (host emulation) at reduced sizes ----------------------------------------------------------------------
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_incompressible_hostemu(ocn, backend, stepper):
    _run(ocn, backend, False)
    _incompressible(ocn, 16, stepper, 3)


def test_taylor_green_hostemu(ocn, backend):
    _run(ocn, backend, False)
    _taylor_green(ocn, 64, "RungeKutta3")


def test_halos_and_poisson_hostemu(ocn, backend):
    _run(ocn, backend, False)
    _halos(ocn)
    _halos_degenerate(ocn)
    _poisson(ocn)


def test_poisson_all_topologies_hostemu(ocn, backend):
    _run(ocn, backend, False)
    _poisson_all_topologies(ocn, [(7, 11, 16), (16, 7, 11)])


def test_incompressible_walls_hostemu(ocn, backend):
    _run(ocn, backend, False)
    _incompressible_walls(ocn, 10, "RungeKutta3", 3)


# ---- GPU at the reference's sizes -----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
@pytest.mark.parametrize("Nt", [1, 10, 100])
def test_incompressible_gpu(ocn, backend, stepper, Nt):
    _run(ocn, backend, True)
    _incompressible(ocn, 32, stepper, Nt)


@pytest.mark.gpu
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_taylor_green_gpu(ocn, backend, stepper):
    _run(ocn, backend, True)
    _taylor_green(ocn, 64, stepper)


@pytest.mark.gpu
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_tracer_advection_gpu(ocn, backend, stepper):
    _run(ocn, backend, True)
    _tracer_advection(ocn, 128, stepper)


@pytest.mark.gpu
def test_poisson_all_topologies_gpu(ocn, backend):
    _run(ocn, backend, True)
    _poisson_all_topologies(ocn, [(7, 11, 16), (16, 7, 11), (32, 24, 20)])


@pytest.mark.gpu
@pytest.mark.parametrize("stepper", ["QuasiAdamsBashforth2", "RungeKutta3"])
def test_incompressible_walls_gpu(ocn, backend, stepper):
    _run(ocn, backend, True)
    _incompressible_walls(ocn, 24, stepper, 10)


@pytest.mark.gpu
def test_halos_and_poisson_gpu(ocn, backend):
    _run(ocn, backend, True)
    _halos(ocn)
    _halos_degenerate(ocn)
    _poisson(ocn)


# ---- phase-level entry points (what a Julia shim overloading one phase at a time would call) ---------------------------
def _phase_level_equals_time_step(ocn, cfg_name):
    """time_step! assembled by hand from the phase-level C entry points (quasi_adams_bashforth_2.jl:70-104 order:
    tendencies, ab2_step, pressure correction, velocity correction, store, update_state) must reproduce ocn_time_step --
    also right after fused steps, when G^n / G^- are pointer-rotated inside the library."""
    import ctypes as C
    import parity_cases as pc
    cfg = pc.CASES[cfg_name]
    a, b = pc.build(ocn, cfg), pc.build(ocn, cfg)
    lib = a.lib
    dt, chi = cfg["dt"], 0.1
    for step in range(3):
        ocn.time_step(a, dt)
        if step == 1:
            ocn.time_step(b, dt)             # one library-driven step in between (leaves rotated / aliased buffers)
            continue
        euler = step == 0
        if euler:                            # first step: chi = -1/2 and G^- = 0 (:74-84)
            for n, f in b.Gm.items():
                f.set_parent(np.zeros(f.total))
        assert lib.ocn_compute_tendencies(b.h) == 0
        assert lib.ocn_ab2_step(b.h, C.c_double(dt), C.c_double(-0.5 if euler else chi)) == 0
        assert lib.ocn_pressure_correction(b.h, C.c_double(dt)) == 0
        assert lib.ocn_pressure_correct_velocities(b.h, C.c_double(dt)) == 0
        assert lib.ocn_store_tendencies(b.h) == 0
        assert lib.ocn_update_state(b.h) == 0
        # the caller of the phase-level entry points owns the clock (TimeSteppers/clock.jl; previous_dt decides Euler vs AB2)
        assert lib.ocn_set_clock(b.h, C.c_double((step + 1) * dt), C.c_int64(step + 1), C.c_double(dt)) == 0
    for n in a.prognostic():
        x, y = a.prognostic()[n].parent(), b.prognostic()[n].parent()
        assert np.abs(x - y).max() <= 1e-12 * np.abs(x).max(), n
    assert np.abs(a.pNHS.parent() - b.pNHS.parent()).max() <= 1e-11 * np.abs(a.pNHS.parent()).max()


@pytest.mark.parametrize("cfg_name", ["ppp_weno_ab2", "ppb_c2_btracer", "pbb_u5_channel"])
def test_phase_level_equals_time_step_hostemu(ocn, backend, cfg_name):
    _run(ocn, backend, False)
    _phase_level_equals_time_step(ocn, cfg_name)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["ppp_weno_ab2", "ppb_c2_btracer", "pbb_u5_channel"])
def test_phase_level_equals_time_step_gpu(ocn, backend, cfg_name):
    _run(ocn, backend, True)
    _phase_level_equals_time_step(ocn, cfg_name)
