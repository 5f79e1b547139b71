"""z-slab decomposition logic on the CPU: R host-emulation ranks in one process (threads) against the
single-domain oracle.  Mirrors test/test_distributed_models.jl:361-517 and test_distributed_poisson_solvers.jl:
101-116 (halo == neighbour, R ~ lap(phi), decomposition-independent trajectories), for the z-slab layout the
MI355X build uses.  The RCCL back end itself can only run on GPUs (driver's 2/4/8-GPU bench)."""
import threading

import numpy as np
import pytest

import oracle as O

P = "Periodic"


def run_ranks(ocn, R, fn):
    out, err = [None] * R, []

    def work(r):
        try:
            ctx = ocn.Context(0)
            par = __import__("ocnhip.parallel", fromlist=["x"])
            par.init_comm_local(ctx, r, R)
            out[r] = fn(ctx, r)
        except Exception as e:   # noqa: BLE001
            err.append((r, repr(e)))
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not err, err
    return out


# nzl: levels per rank.  More than 2 H + 2 = 8 of them and the all-in-one path splits its tendency launch into interior
# levels (started while the previous step's halo planes are still travelling) and boundary levels (csrc/api.hip fused_substep).
@pytest.mark.parametrize("R,stepper,adv,nzl", [(2, "AB2", "WENO5", 8), (2, "RK3", "WENO5", 8), (2, "AB2", "C2", 8), (4, "AB2", "WENO5", 8),
                                               (4, "AB2", "C2", 8), (2, "AB2", "WENO5", 10), (2, "RK3", "WENO5", 11), (3, "AB2", "WENO5", 9)])
def test_slab_trajectory_matches_single_domain_oracle(ocn, backend, R, stepper, adv, nzl, monkeypatch):
    _slab_trajectory(ocn, backend, R, stepper, adv, nzl, monkeypatch)


# The target machine: R = 8 (BASELINE config 4 is 512 x 512 x 256 in eight z-slabs).  Slabs of exactly 9 and of 8 levels sit on
# either side of the overlap split's limit (more than 2 H + 2 = 8 levels: interior levels start while the halo planes travel),
# AB2 and RK3 + tracer, with the Green's-function z stage and with the all-to-all transposed one.
@pytest.mark.parametrize("solver", ["green", "transpose"])
@pytest.mark.parametrize("stepper,nzl", [("AB2", 9), ("RK3", 9), ("AB2", 8), ("RK3", 8)])
def test_eight_rank_slabs_match_single_domain_oracle(ocn, backend, stepper, nzl, solver, monkeypatch):
    monkeypatch.setenv("OCNHIP_DIST_SOLVER", solver)
    _slab_trajectory(ocn, backend, 8, stepper, "WENO5", nzl, monkeypatch, overlap=True)


def test_four_rank_slabs_with_local_planes(ocn, backend, monkeypatch):
    """R = 4 slabs of a 128 x 128 x 32 box: 128-point rows put the run on the custom transform passes, where the w* term above a
    slab and the pressure plane below it are computed by the rank itself (two exchanges per step; csrc/zslab.hip `bel`,
    k_zslab_below) -- the carries then run over three other ranks and the slab's own periodic image"""
    _slab_trajectory(ocn, backend, 4, "AB2", "WENO5", 8, monkeypatch, overlap=True, Nxy=(128, 128), nsteps=1)


def _slab_trajectory(ocn, backend, R, stepper, adv, nzl, monkeypatch, overlap=None, Nxy=(8, 8), nsteps=None):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    if nzl != 8 or overlap:
        monkeypatch.setenv("OCNHIP_OVERLAP", "1")   # by default only with 8 MB or more of halo planes per direction
    N = (Nxy[0], Nxy[1], nzl * R)
    rng = np.random.default_rng(5)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    tracers = ("c",) if stepper == "RK3" else ()
    for t in tracers:
        init[t] = rng.random(N)
    ext = (1, 1, float(R)) if Nxy == (8, 8) else (1.0, N[1] / N[0], N[2] / N[0])
    og = O.RectilinearGrid(size=N, extent=ext, topology=(P,) * 3)
    om = O.NonhydrostaticModel(og, advection=O.WENO5() if adv == "WENO5" else O.CenteredSecondOrder(), timestepper=stepper,
                               tracers=tracers)
    O.set_model(om, **init)
    dt = 2e-3 if Nxy == (8, 8) else 0.1 / N[0] / np.abs(om.u.data).max()
    nsteps = nsteps or (2 if nzl == 8 else 3)
    for _ in range(nsteps):
        O.time_step(om, dt)

    def rank_fn(ctx, r):
        g = ocn.RectilinearGrid(ctx, size=N, extent=ext, topology=(P,) * 3)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5() if adv == "WENO5" else ocn.CenteredSecondOrder(),
                                    timestepper=stepper, tracers=tracers)
        nz = N[2] // R
        ocn.set_model(m, **{n: a[:, :, r * nz:(r + 1) * nz] for n, a in init.items()})
        for _ in range(nsteps):
            ocn.time_step(m, dt)
        out = {n: f.parent() for n, f in (("u", m.u), ("v", m.v), ("w", m.w), ("p", m.pNHS))}
        out.update({t: m.tracers[t].parent() for t in tracers})
        return out, m.max_abs_divergence()
    res = run_ranks(ocn, R, rank_fn)
    H, nz = 3, N[2] // R
    for r, (flds, div) in enumerate(res):
        assert div < 1e-11
        refs = [("u", om.u.data), ("v", om.v.data), ("w", om.w.data), ("p", om.pNHS.data)]
        refs += [(t, om.tracers[t].data) for t in tracers]
        for n, ref in refs:
            # slab parent (incl. z halos) == the matching window of the periodic global parent array
            idx = (np.arange(-H, nz + H) + r * nz) % N[2] + H
            want = ref[:, :, idx]
            err = np.abs(flds[n] - want).max() / np.abs(ref).max()
            assert err < 1e-11, (r, n, err)


@pytest.mark.parametrize("solver", ["green", "transpose"])
def test_slab_poisson_residual(ocn, backend, solver, monkeypatch):
    """test_distributed_poisson_solvers.jl:101-116: R == lap(phi) for a random source on 4 ranks, with the
    transpose-free Green's-function z stage (default) and with the all-to-all transposed z-FFT."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    monkeypatch.setenv("OCNHIP_DIST_SOLVER", solver)
    R, N = 4, (12, 8, 24)
    rng = np.random.default_rng(9)
    src = rng.random(N)
    src -= src.mean()

    def rank_fn(ctx, r):
        g = ocn.RectilinearGrid(ctx, size=N, extent=(1, 2, 3), topology=(P,) * 3)
        m = ocn.NonhydrostaticModel(g)
        nz = N[2] // R
        return m.poisson_solve(np.ascontiguousarray(src[:, :, r * nz:(r + 1) * nz]))
    phi = np.concatenate(run_ranks(ocn, R, rank_fn), axis=2)
    lap = np.zeros(N)
    for ax, d in ((0, 1 / N[0]), (1, 2 / N[1]), (2, 3 / N[2])):
        lap += (np.roll(phi, -1, ax) - 2 * phi + np.roll(phi, 1, ax)) / d ** 2
    assert np.abs(lap - src).max() < 1e-10 * np.abs(src).max()


# ---- y-slabs for a Bounded z (the reference has no distributed Fourier-tridiagonal solver) --
def _yslab_case(kind):
    from parity_cases import CASES
    import copy
    cfg = copy.deepcopy(CASES["ppb_amd_config3" if kind == "amd" else "ppb_weno_full"])
    return cfg


@pytest.mark.parametrize("R,kind", [(2, "amd"), (2, "scalar"), (3, "scalar"), (8, "amd")])
def test_yslab_trajectory_matches_single_domain_oracle(ocn, backend, R, kind):
    """(Periodic, Periodic, Bounded) with T/S, buoyancy, Coriolis, closure (AMD or scalar), flux / gradient boundary
    conditions, WENO5, RK3 on R y-slabs against the single-domain oracle: every field, parent arrays with halos."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    import parity_cases as pc
    cfg = _yslab_case(kind)
    Nx, Nz = cfg["size"][0], cfg["size"][2]
    Ny = 6 * R
    cfg["size"] = (Nx, Ny, Nz)
    om = pc.build(O, cfg)
    names = ["u", "v", "w"] + list(cfg["tracers"])
    init = {n: (getattr(om, n) if n in "uvw" else om.tracers[n]).interior().copy() for n in names}
    # rebuild the oracle from the captured (pre-projection) state so both sides start from identical arrays
    rng = np.random.default_rng(77)
    init = {n: rng.random(a.shape) - (0.5 if n in "uvw" else 0.0) for n, a in init.items()}
    init["w"][:, :, 0] = 0
    init["w"][:, :, -1] = 0
    O.set_model(om, **init)
    for _ in range(cfg["steps"]):
        O.time_step(om, cfg["dt"])
    nyl = Ny // R

    def rank_fn(ctx, r):
        cfg_r = dict(cfg)
        m = _build_on(ocn, ctx, cfg_r)
        ocn.set_model(m, **{n: np.ascontiguousarray(a[:, r * nyl:(r + 1) * nyl]) for n, a in init.items()})
        for _ in range(cfg["steps"]):
            ocn.time_step(m, cfg["dt"])
        out = {n: (getattr(m, n) if n in "uvw" else m.tracers[n]).parent() for n in names}
        out["p"] = m.pNHS.parent()
        out["pHY"] = m.pHY.parent()
        return out, m.max_abs_divergence()
    res = run_ranks(ocn, R, rank_fn)
    H = 3
    refs = {n: (getattr(om, n) if n in "uvw" else om.tracers[n]).data for n in names}
    refs["p"], refs["pHY"] = om.pNHS.data, om.pHY.data
    for r, (flds, div) in enumerate(res):
        assert div < 1e-10
        for n, ref in refs.items():
            idx = (np.arange(-H, nyl + H) + r * nyl) % Ny + H          # slab rows incl. y halos in the global parent
            want = ref[:, idx]
            err = np.abs(flds[n] - want).max() / max(np.abs(ref).max(), 1e-300)
            assert err < 2e-11, (r, n, err)


def _build_on(mod, ctx, cfg):
    """parity_cases.build for a given context (grid bound to this rank's communicator), without the initial set!"""
    kw = {}
    if "extent" in cfg:
        kw["extent"] = cfg["extent"]
    else:
        kw["x"], kw["y"] = cfg["xy"]
        kw["z"] = np.array(cfg["zfaces"], dtype=float)
    g = mod.RectilinearGrid(ctx, size=cfg["size"], topology=cfg["topo"], **kw)
    mk = {}
    if cfg.get("closure") == "amd":
        mk["closure"] = mod.AnisotropicMinimumDissipation()
    elif cfg.get("closure"):
        mk["closure"] = mod.ScalarDiffusivity(nu=cfg["closure"][0], kappa=cfg["closure"][1])
    if cfg.get("coriolis"):
        mk["coriolis"] = mod.FPlane(cfg["coriolis"])
    if cfg.get("buoyancy") == "TS":
        a_, b_ = cfg.get("eos", (2e-1, 8e-1))
        mk["buoyancy"] = mod.SeawaterBuoyancy(thermal_expansion=a_, haline_contraction=b_)
    if cfg.get("bcs"):
        ctor = {"flux": mod.FluxBC, "value": mod.ValueBC, "gradient": mod.GradientBC}
        mk["boundary_conditions"] = {f: {s: ctor[k](v) for s, (k, v) in sides.items()} for f, sides in cfg["bcs"].items()}
    from parity_cases import _adv
    return mod.NonhydrostaticModel(g, advection=_adv(mod, cfg["adv"]), timestepper=cfg["stepper"],
                                   tracers=cfg.get("tracers", ()), **mk)


def test_yslab_poisson_matches_oracle(ocn, backend):
    """Fourier-tridiagonal solve on 3 y-slabs (Nxh = 6 is not divisible by 3: padded kx bands) == the oracle's solve."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    from oracle.poisson import FourierTridiagonalPoissonSolver
    R, N = 3, (10, 18, 8)
    faces = np.array([1, 2, 4, 7, 11, 16, 22, 29, 37.0])
    kw = dict(x=(0, 1), y=(0, 2), z=faces, topology=(P, P, "Bounded"))
    rng = np.random.default_rng(13)
    src = rng.random(N)
    dz = np.diff(faces).reshape(1, 1, -1)
    src -= (src * dz).sum() / (dz.sum() * N[0] * N[1])
    ref = FourierTridiagonalPoissonSolver(O.RectilinearGrid(size=N, **kw)).solve_source(src)
    nyl = N[1] // R

    def rank_fn(ctx, r):
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ctx, size=N, **kw))
        return m.poisson_solve(np.ascontiguousarray(src[:, r * nyl:(r + 1) * nyl]))
    phi = np.concatenate(run_ranks(ocn, R, rank_fn), axis=1)
    assert np.abs(phi - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("R", [2, 3])
def test_slab_viscous_tracer_matches_single_domain_oracle(ocn, backend, R):
    """z-slabs with ScalarDiffusivity on the fused path: the viscous face fluxes reach one level into the neighbour's slab
    (div U below the first level, w one level up in the west / south column)."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    N = (8, 9, 6 * R)
    rng = np.random.default_rng(15)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    init["c"] = rng.random(N)
    kw = dict(size=N, extent=(1, 1.1, 0.7 * R), topology=(P,) * 3)
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5(), tracers=("c",), closure=O.ScalarDiffusivity(nu=3e-2, kappa=2e-2))
    O.set_model(om, enforce_incompressibility=False, **init)      # non-solenoidal start: the grad-div term is alive
    dt = 2e-3
    for _ in range(3):
        O.time_step(om, dt)
    nz = N[2] // R

    def rank_fn(ctx, r):
        g = ocn.RectilinearGrid(ctx, **kw)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5(), tracers=("c",), closure=ocn.ScalarDiffusivity(nu=3e-2, kappa=2e-2))
        ocn.set_model(m, enforce_incompressibility=False, **{n: a[:, :, r * nz:(r + 1) * nz] for n, a in init.items()})
        for _ in range(3):
            ocn.time_step(m, dt)
        return {n: f.interior() for n, f in (("u", m.u), ("v", m.v), ("w", m.w), ("p", m.pNHS), ("c", m.tracers["c"]))}
    res = run_ranks(ocn, R, rank_fn)
    refs = {"u": om.u.interior(), "v": om.v.interior(), "w": om.w.interior(), "p": om.pNHS.interior(), "c": om.tracers["c"].interior()}
    for r, flds in enumerate(res):
        for n, ref in refs.items():
            err = np.abs(flds[n] - ref[:, :, r * nz:(r + 1) * nz]).max() / np.abs(ref).max()
            assert err < 1e-11, (r, n, err)
