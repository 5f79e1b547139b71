"""GPU parity: libocnhip.so (HIP, gfx950) through the C ABI against the oracle on seeded inputs, plus
size-independent properties at the benchmark size (projection => divergence-free, Poisson residual,
Galilean / mirror consistency are covered on the small cases by the oracle comparison)."""
import numpy as np
import pytest

from parity_cases import CASES, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_oracle(ocn, name):
    worst = run_case(ocn, name)
    bad = {k: v for k, v in worst.items() if v > CASES[name].get("tol", 2e-11)}   # Float64: tendencies 1e-12, trajectories << sqrt(eps)
    assert not bad, bad


def test_medium_weno_trajectory(ocn):
    """64^3 triply-periodic WENO5 AB2, 5 steps, vs the oracle (a few seconds of NumPy)."""
    import oracle as O
    N = (64, 64, 64)
    rng = np.random.default_rng(1)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5())
    ocn.set_model(m, **init)
    og = O.RectilinearGrid(size=N, extent=(1, 1, 1), topology=("Periodic",) * 3)
    om = O.NonhydrostaticModel(og, advection=O.WENO5())
    O.set_model(om, **init)
    dt = 0.2 / 64 / np.abs(om.u.data).max()
    for _ in range(5):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
    for a, b in ((m.u.parent(), om.u.data), (m.v.parent(), om.v.data), (m.w.parent(), om.w.data),
                 (m.pNHS.parent(), om.pNHS.data)):
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()


@pytest.mark.parametrize("N,stepper,tracers", [((256, 256, 8), "AB2", ()), ((256, 256, 12), "RK3", ("c",)),
                                               ((16, 12, 256), "AB2", ()), ((64, 256, 256), "RK3", ())])
def test_headline_kernels_vs_oracle(ocn, N, stepper, tracers):
    """Triply periodic WENO5 on the shapes that select the headline kernels -- 256-point x / y passes with the fused
    right-hand side (Nx = Ny = 256), the fused z stage (Nz = 256), the tiled tendency kernel at full row width --
    compared with the oracle field by field (the full 256^3 takes the oracle minutes per step: properties only)."""
    import oracle as O
    Nz = N[2]
    rng = np.random.default_rng(5)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    init.update({t: rng.random(N) for t in tracers})
    kw = dict(size=N, extent=(1, 1, Nz / 256), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), advection=ocn.WENO5(), timestepper=stepper, tracers=tracers)
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5(), timestepper=stepper, tracers=tracers)
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    dt = 0.2 / 256 / np.abs(om.u.data).max()
    for _ in range(2):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
    pairs = [(m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS)] + [(m.tracers[t], om.tracers[t]) for t in tracers]
    for a, b in pairs:
        assert np.abs(a.parent() - b.data).max() <= 2e-11 * np.abs(b.data).max()


@pytest.mark.parametrize("N", [(320, 24, 16), (512, 16, 12), (700, 12, 8)])
def test_wide_rows_vs_oracle(ocn, N):
    """Rows wider than a workgroup (Nx > 256) take the x-tiled tendency kernel: 2 to 4 x-tiles per row, vs the oracle."""
    import oracle as O
    rng = np.random.default_rng(8)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    kw = dict(size=N, extent=(N[0] / 64, N[1] / 64, N[2] / 64), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), advection=ocn.WENO5())
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5())
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    dt = 0.2 / 64 / np.abs(om.u.data).max()
    for _ in range(2):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
    for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS)):
        assert np.abs(a.parent() - b.data).max() <= 2e-11 * np.abs(b.data).max()


def test_wide_bounded_rows_vs_oracle(ocn):
    """(Periodic, Periodic, Bounded) with rows wider than a workgroup: x-tiled tendency kernel in its bounded-z form, on top
    of the general kernels' buoyancy / Coriolis / closure terms, stretched z, RK3 -- vs the oracle."""
    import copy
    import parity_cases as pc
    cfg = copy.deepcopy(pc.CASES["ppb_stretched_bcs"])
    cfg["size"] = (300, 10, 8)
    om, dm = pc.build(pc.O, cfg), pc.build(ocn, cfg)
    for _ in range(2):
        pc.O.time_step(om, cfg["dt"])
        ocn.time_step(dm, cfg["dt"])
    a, b = pc.fields_of(om, True), pc.fields_of(dm, False)
    for k in a:
        assert np.abs(a[k] - b[k]).max() <= 2e-11 * max(np.abs(a[k]).max(), 1e-300), k


def test_full_size_properties(ocn):
    """BASELINE config 2 (256^3, WENO5, AB2): projection leaves max|div U| ~ roundoff, halos are periodic
    images, and the Poisson solve satisfies lap(phi) = R to roundoff."""
    N = (256, 256, 256)
    rng = np.random.default_rng(1)
    g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5())
    ocn.set_model(m, u=rng.random(N) - 0.5, v=rng.random(N) - 0.5, w=rng.random(N) - 0.5)
    umax = np.abs(m.u.interior()).max()
    dx = 1 / 256
    assert m.max_abs_divergence() <= 1e-12 * umax / dx * 50
    dt = 0.2 * dx / umax
    for _ in range(3):
        ocn.time_step(m, dt)
    assert m.max_abs_divergence() <= 1e-12 * umax / dx * 50
    u = m.u.parent()
    H = 3
    assert np.array_equal(u[:H], u[256:256 + H]) and np.array_equal(u[256 + H:], u[H:2 * H])
    assert np.array_equal(u[:, :, :H], u[:, :, 256:256 + H]) and np.array_equal(u[:, :H], u[:, 256:256 + H])
    assert np.isfinite(u).all()
    # Poisson: lap(phi) == R for a zero-mean random source
    R = rng.random(N)
    R -= R.mean()
    phi = m.poisson_solve(R)

    def lap(p):
        out = np.zeros_like(p)
        for ax, d in ((0, dx), (1, dx), (2, dx)):
            out += (np.roll(p, -1, ax) - 2 * p + np.roll(p, 1, ax)) / d ** 2
        return out
    assert np.abs(lap(phi) - R).max() <= 1e-10 * np.abs(R).max()


@pytest.mark.parametrize("topo,N", [(("Periodic", "Periodic", "Periodic"), (16, 11, 7)),
                                    (("Periodic", "Periodic", "Bounded"), (16, 11, 7)),
                                    (("Periodic", "Periodic", "Bounded"), (7, 16, 11))])
def test_poisson_divergence_free_solution(ocn, topo, N):
    """test_poisson_solvers.jl:45-85 through the C ABI (supported topologies, prime / even sizes)."""
    import oracle as O
    from oracle.fields import Field, fill_halo_regions
    from oracle.operators import Ops
    rng = np.random.default_rng(3)
    g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=topo)
    m = ocn.NonhydrostaticModel(g)
    R = rng.random(N)
    if topo[2] == "Periodic":
        R -= R.mean()
    else:
        R -= R.mean()
    phi = m.poisson_solve(R)
    og = O.RectilinearGrid(size=N, extent=(1, 1, 1), topology=topo)
    f = Field(og, (O.Center,) * 3)
    f.set(phi)
    fill_halo_regions(f)
    lap = Ops(og).laplacian_ccc(f)((0, 0, 0))
    assert np.allclose(lap, R, rtol=1.5e-8, atol=1e-10)


@pytest.mark.parametrize("kind", ["periodic", "bounded", "wide", "slab"])
def test_bitwise_reproducible(ocn, kind, monkeypatch):
    """No atomics, no order-dependent reductions: two runs from the same state give bit-identical fields.  A race in the
    LDS-staged kernels (slab commit vs. flux stage, flux exchange vs. finalize) would show up here as run-to-run noise."""
    if kind == "slab":   # z-slab code path with the halo planes travelling on the communication stream under the next interior launch:
        monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")   # a missing event wait between the two streams would show up as noise here
        monkeypatch.setenv("OCNHIP_OVERLAP", "1")
    N = {"periodic": (256, 64, 48), "bounded": (128, 96, 40), "wide": (400, 40, 24), "slab": (256, 64, 40)}[kind]
    topo = ("Periodic", "Periodic", "Bounded" if kind == "bounded" else "Periodic")
    rng = np.random.default_rng(21)
    init = {n: rng.random(N if not (n == "w" and kind == "bounded") else (N[0], N[1], N[2] + 1)) - 0.5 for n in "uvw"}
    init["c"] = rng.random(N)
    out = []
    for _ in range(2):
        g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=topo)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5(), tracers=("c",), timestepper="RK3" if kind == "bounded" else "AB2",
                                    closure=ocn.ScalarDiffusivity(nu=1e-4, kappa=1e-4) if kind != "wide" else None)
        ocn.set_model(m, **init)
        for _ in range(15):
            ocn.time_step(m, 5e-4)
        out.append([m.u.parent(), m.v.parent(), m.w.parent(), m.pNHS.parent(), m.tracers["c"].parent()])
    for a, b in zip(*out):
        assert np.isfinite(a).all() and np.array_equal(a, b)


# ---- slab code paths on ONE GPU: OCNHIP_FORCE_DIST=1 is read per grid creation (csrc/api.hip ocn_grid_create), so the
# z-slab / y-slab kernels (k_zslab_*, k_pack_rows, k_yslab_*, fused_exchange_*) and both distributed solvers run on
# hardware in the driver's `-m gpu` pass, exchanging with themselves, against the single-domain oracle.
FORCED = ["ppp_weno_ab2", "ppp_weno_rk3_2tracers", "ppb_amd_config3", "ppb_weno_full", "regr_ocean_les_amd"]


@pytest.mark.parametrize("solver", ["green", "transpose"])
@pytest.mark.parametrize("name", [n for n in FORCED if n in CASES])
def test_forced_slab_case_matches_oracle(ocn, name, solver, monkeypatch):
    monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")
    monkeypatch.setenv("OCNHIP_DIST_SOLVER", solver)
    monkeypatch.setenv("OCNHIP_OVERLAP", "1")     # the multi-rank default: halo planes on the communication stream
    worst = run_case(ocn, name)
    bad = {k: v for k, v in worst.items() if v > CASES[name].get("tol", 2e-11)}
    assert not bad, bad


@pytest.mark.parametrize("forced", [False, True])
def test_config4_slab_shape_vs_oracle(ocn, forced, monkeypatch):
    """512 x 512 x 32: one z-slab of BASELINE config 4 (512 x 512 x 256 on 8 GPUs).  Two WENO5 AB2 steps against the oracle,
    as a single periodic domain and through the slab code path (x-tiled tendency kernel, slab Poisson solver, plane
    exchanges with itself)."""
    import oracle as O
    if forced:
        monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")
        monkeypatch.setenv("OCNHIP_OVERLAP", "1")   # what each of the 8 ranks runs: interior levels under the travelling halo planes
    N = (512, 512, 32)
    rng = np.random.default_rng(4)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    kw = dict(size=N, extent=(1, 1, 32 / 512), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), advection=ocn.WENO5())
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5())
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    dt = 0.2 / 512 / np.abs(om.u.data).max()
    for _ in range(2):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
    for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS)):
        assert np.abs(a.parent() - b.data).max() <= 2e-11 * np.abs(b.data).max()
    assert m.max_abs_divergence() <= 1e-9


def test_config3_full_size_properties(ocn):
    """BASELINE config 3 at full size (256 x 256 x 128, stretched Bounded z, T and S, FPlane, linear EOS, AMD, flux /
    gradient conditions, RK3, WENO5): after three steps the velocity is divergence free, w vanishes on both walls, and
    the T and S budgets changed by exactly the imposed boundary fluxes (test_boundary_conditions_integration.jl:26-50:
    d/dt of the volume integral = -(top flux - bottom flux) x area; advection and diffusion are conservative)."""
    Nx, Ny, Nz = 256, 256, 128
    Lz, refinement, stretching = 32.0, 1.2, 12.0
    k = np.arange(1, Nz + 2)
    h = (k - 1) / Nz
    zf = Lz * ((1 + (h - 1) / refinement) * (1 - np.exp(-stretching * h)) / (1 - np.exp(-stretching)) - 1)
    grid = ocn.RectilinearGrid(size=(Nx, Ny, Nz), x=(0.0, 512.0), y=(0.0, 512.0), z=zf, topology=("Periodic", "Periodic", "Bounded"))
    QT = 200.0 / (1026.0 * 3991.0)
    Qu = -1.225 / 1026.0 * 2.5e-3 * 100
    QS = -1e-3 / 3600 * 35.0
    dTdz = 0.01
    bcs = {"u": {"top": ocn.FluxBC(Qu)}, "T": {"top": ocn.FluxBC(QT), "bottom": ocn.GradientBC(dTdz)}, "S": {"top": ocn.FluxBC(QS)}}
    m = ocn.NonhydrostaticModel(grid, advection=ocn.WENO5(), timestepper="RungeKutta3", tracers=("T", "S"),
                                coriolis=ocn.FPlane(1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                buoyancy=ocn.SeawaterBuoyancy(thermal_expansion=2e-4, haline_contraction=8e-4),
                                boundary_conditions=bcs)
    rng = np.random.default_rng(3)
    zc = 0.5 * (zf[1:] + zf[:-1]).reshape(1, 1, -1)
    noise = lambda z, shape: rng.standard_normal(shape) * z / Lz * (1 + z / Lz)   # noqa: E731
    T0 = 20 + dTdz * zc + dTdz * Lz * 1e-6 * noise(zc, (Nx, Ny, Nz))
    u0 = np.sqrt(abs(Qu)) * 1e-3 * noise(zc, (Nx, Ny, Nz))
    ocn.set_model(m, u=u0, T=T0, S=35.0)
    dz = np.diff(zf).reshape(1, 1, -1)
    vol = 2.0 * 2.0 * dz
    T_before = (m.tracers["T"].interior() * vol).sum()
    S_before = (m.tracers["S"].interior() * vol).sum()
    dt, nsteps = 1.0, 3
    for _ in range(nsteps):
        ocn.time_step(m, dt)
    w = m.w.interior()
    assert w.shape[2] == Nz + 1 and np.all(w[:, :, 0] == 0) and np.all(w[:, :, -1] == 0)
    umax = max(np.abs(m.u.interior()).max(), 1e-12)
    assert m.max_abs_divergence() <= 1e-9 * umax / dz.min()
    area = 512.0 * 512.0
    T_after = (m.tracers["T"].interior() * vol).sum()
    S_after = (m.tracers["S"].interior() * vol).sum()
    # S has flux conditions only: its budget is exact.  T also loses what diffuses through the bottom wall under the
    # gradient condition, -kappa_e dT/dz there (>= 0 and tiny: the AMD diffusivity of a quiescent bottom layer).
    tS, tT = -QS * area * dt * nsteps, -QT * area * dt * nsteps
    assert abs((S_after - S_before) - tS) <= 1e-12 * abs(S_before) + 1e-9 * abs(tS)
    leak = np.abs(m.kappa_e["T"].interior()[:, :, 0]).max() * dTdz * area * dt * nsteps
    dT = (T_after - T_before) - tT
    assert -1.01 * leak - 1e-12 * abs(T_before) - 1e-9 * abs(tT) <= dT <= 1e-12 * abs(T_before) + 1e-9 * abs(tT)
    assert np.isfinite(m.tracers["T"].interior()).all()


def test_forced_slab_through_rccl_self(ocn, monkeypatch):
    """The slab code path exchanging with itself through RCCL (a one-rank communicator, grouped ncclSend / ncclRecv on the
    library's stream, byte counts as in a multi-GPU run) instead of device copies: z-slab WENO5 AB2 vs the oracle.  Sends
    and receives of one group pair up in posting order -- the rule the multi-rank exchanges rely on."""
    import oracle as O
    from importlib import import_module
    monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")
    monkeypatch.setenv("OCNHIP_RCCL_SELF", "1")
    monkeypatch.setenv("OCNHIP_OVERLAP", "1")     # as with several ranks: the RCCL group of (u, v, w, tracers) and the one of pNHS run on
    monkeypatch.delenv("OCNHIP_TRANSPORT", raising=False)   # the communication stream under the next step's interior tendency launch
    par = import_module("ocnhip.parallel")
    ctx = ocn.Context(0)
    par.init_comm_self(ctx)
    N = (32, 24, 18)
    rng = np.random.default_rng(11)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    init["c"] = rng.random(N)
    kw = dict(size=N, extent=(1, 0.75, 0.5), topology=("Periodic",) * 3)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ctx, **kw), advection=ocn.WENO5(), tracers=("c",))
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5(), tracers=("c",))
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    for _ in range(4):
        ocn.time_step(m, 2e-3)
        O.time_step(om, 2e-3)
    for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS), (m.tracers["c"], om.tracers["c"])):
        assert np.abs(a.parent() - b.data).max() <= 2e-11 * np.abs(b.data).max()


# ---- whole-step hipGraphs of the general path (csrc/api.hip step_graphed) ------------------------------------------------
GRAPH_CASES = ["ppf_weno_rk3", "ppb_amd_config3", "ppb_weno_noslip", "ppp_c4", "regr_thermal_bubble_regular",
               "ppb_c4_ab2_varying_dt", "bbb_weno_walls", "ppp_weno_amd_coriolis", "bfb_weno_slice",
               # the all-in-one periodic path (small boxes only; round 3): AB2 with its G^n / G^- pointer rotation, RK3 + tracers,
               # ScalarDiffusivity, a changing dt
               "ppp_weno_ab2", "ppp_weno_rk3_2tracers", "ppp_weno_visc_ab2", "ppp_weno_ab2_varying_dt"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAPH_CASES)
def test_step_graph_replay_is_bitwise_the_launch_train(ocn, name, monkeypatch):
    """A step replayed from its hipGraph leaves exactly the bits the launch-by-launch step leaves, the clock and the G^n / G^-
    rotation included, over AB2 (Euler start, changing dt) and RK3; a model whose step holds a call that must not be
    captured (the wall-bounded solver's BLAS products) drops out of graph mode and still steps correctly."""
    from parity_cases import CASES, build, fields_of
    cfg = CASES[name]
    dts = list(cfg.get("dts", [])) or [cfg["dt"]] * 3
    dts = dts + [dts[-1]] * 7            # the repeated tail is what gets captured and replayed
    res = []
    for nograph in (False, True):
        if nograph:
            monkeypatch.setenv("OCNHIP_NO_GRAPH", "1")
        m = build(ocn, cfg)
        for dt in dts:
            ocn.time_step(m, dt)
        replays, active = m.graph_replays
        res.append((fields_of(m, False), m.time, m.iteration, replays, active))
    (fa, ta, ia, ra, acta), (fb, tb, ib, rb, actb) = res
    assert rb == 0 and not actb
    if "Bounded" in cfg["topo"][:2]:
        assert ra == 0 and not acta      # poisoned capture (BLAS cosine transforms): graphs switched off for this model
    else:
        assert acta and ra >= 4
    assert ta == tb and ia == ib
    for k in fa:
        assert np.isfinite(fa[k]).all() and np.array_equal(fa[k], fb[k]), k


# ---- BASELINE config 1 at its own size ----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("graphs", [True, False], ids=["step_graphs", "launch_by_launch"])
@pytest.mark.parametrize("dt_kind", ["example", "bench"])
def test_config1_128x128_matches_oracle(ocn, graphs, dt_kind, monkeypatch):
    """parity_cases.run_config1 (examples/two_dimensional_turbulence.jl at its own 128 x 128) with the script's dt = 0.2 and with
    bench.py's 0.2 dx / 4, replayed from whole-step hipGraphs and launch by launch."""
    from parity_cases import run_config1
    if not graphs:
        monkeypatch.setenv("OCNHIP_NO_GRAPH", "1")
    m = run_config1(ocn, 0.2 if dt_kind == "example" else 0.2 * (2 * np.pi / 128) / 4.0, steps=4)
    replays, active = m.graph_replays
    assert (replays >= 1 and active) if graphs else (replays == 0 and not active)


@pytest.mark.gpu
@pytest.mark.parametrize("rccl_self", [False, True], ids=["device_copies", "rccl_self"])
def test_two_slab_models_share_a_context(ocn, rccl_self, monkeypatch):
    """two forced z-slab models on one context, overlapped halo planes, stepped alternately (parity_cases); with the planes
    moved by device copies and by a one-rank RCCL communicator"""
    monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")
    monkeypatch.setenv("OCNHIP_OVERLAP", "1")
    if rccl_self:
        pytest.skip("one RCCL communicator per context is created by the caller: covered by test_forced_slab_through_rccl_self")
    from parity_cases import run_two_slab_models_on_one_context
    run_two_slab_models_on_one_context(ocn)
