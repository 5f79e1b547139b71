"""Seeded random configurations of the hydrostatic time step -- grid kind, sizes, halos, stretched / regular z, wall-bounded or periodic
x, buoyancy, Coriolis, advection schemes, number of ranks (latitude bands, replicated or banded free surface) -- through the library
(host emulation; `-m gpu`: libocnhip.so on one rank) against the single-domain oracle: owned rows bit for bit with the default tracer
scheme (to 2e-11 with the higher-order ones, whose kernels share the Nonhydrostatic reconstructions)."""
import numpy as np
import pytest

from oracle import hydrostatic as OH
from oracle import split_explicit as OS
from test_distributed_hostemu import run_ranks

P, B = "Periodic", "Bounded"
OMEGA = 7.292115e-5


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    latlon = rng.random() < 0.7
    R = int(rng.choice([1, 1, 2, 3, 4]))
    H = int(rng.choice([1, 2, 3]))
    scheme = str(rng.choice(["CenteredSecondOrder", "CenteredSecondOrder", "CenteredFourthOrder", "UpwindBiasedFifthOrder", "WENO5"]))
    madv = [None, "VectorInvariantEnstrophyConserving", "VectorInvariantEnergyConserving", "WENOVectorInvariantVorticityStencil"][int(rng.integers(4))]
    need = {"CenteredSecondOrder": 1, "CenteredFourthOrder": 2}.get(scheme, 3)
    if madv == "WENOVectorInvariantVorticityStencil":
        need = 3
    H = max(H, need)
    nl = int(rng.integers(H + 1, H + 6))                   # rows per band: more than H
    Ny = nl * R
    Nx = int(rng.integers(max(6, H + 1), 20))
    Nz = int(rng.integers(2, 7))
    z = (-float(rng.integers(100, 4000)), 0.0)
    if rng.random() < 0.5:
        zf = np.sort(rng.random(Nz - 1)) if Nz > 1 else np.array([])
        z = list(z[0] * (1 - np.concatenate([[0.0], 0.1 + 0.8 * zf, [1.0]])))
    if latlon:
        full = rng.random() < 0.5
        lon = (-180, 180) if full else (float(rng.integers(-60, 0)), float(rng.integers(10, 90)))
        lat0 = float(rng.integers(-70, 0))
        kw = dict(size=(Nx, Ny, Nz), longitude=lon, latitude=(lat0, lat0 + float(rng.integers(30, 70))), z=z, halo=(H, H, H))
        ctor = "LatitudeLongitudeGrid"
        coriolis = [None, ("HydrostaticSphericalCoriolis", OMEGA, "EnstrophyConserving"), ("HydrostaticSphericalCoriolis", OMEGA, "EnergyConserving")][int(rng.integers(3))]
        ybounded = True
    else:
        topo = (str(rng.choice([P, B])), str(rng.choice([P, B])), B)
        kw = dict(size=(Nx, Ny, Nz), x=(0.0, 1e5), y=(0.0, 2e5), z=z, halo=(H, H, H), topology=topo)
        ctor = "HRectilinearGrid"
        coriolis = [None, ("FPlane", 1e-4)][int(rng.integers(2))]
        ybounded = topo[1] == B
    buoyancy = [None, ("b", "T"), ("TS", 9.8, 2e-4, 8e-4, "T", "S")][int(rng.integers(3))]
    overlap = 0
    if R > 1 and ybounded and rng.random() < 0.5:
        overlap = int(rng.integers(1, nl + 1))
    substeps = int(rng.integers(3, 12))
    closure = None if rng.random() < 0.5 else (float(rng.choice([0.0, 1e-2, 1.0])), {"T": float(rng.choice([0.0, 1e-3, 0.5])), "S": float(rng.choice([0.0, 2e-3]))})
    return dict(ctor=ctor, kw=kw, R=R, coriolis=coriolis, buoyancy=buoyancy, madv=madv, scheme=scheme, overlap=overlap, substeps=substeps,
                ybounded=ybounded, seed=seed, closure=closure)


def fields_of(cfg):
    g = getattr(OS, cfg["ctor"])(**cfg["kw"])
    st = OH.HydrostaticState(g, tracers=("T", "S"), buoyancy=cfg["buoyancy"], substeps=cfg["substeps"])
    rng = np.random.default_rng(cfg["seed"])
    init = {"u": 0.05 * rng.standard_normal(st.u.interior().shape), "v": 0.05 * rng.standard_normal(st.v.interior().shape),
            "eta": 0.02 * rng.standard_normal(st.free_surface.eta.interior().shape),
            "T": 10 + rng.standard_normal(st.tracers["T"].interior().shape), "S": 35 + 0.1 * rng.standard_normal(st.tracers["S"].interior().shape)}
    if g.topo[0] == B:
        init["u"][0], init["u"][-1] = 0, 0
    if g.topo[1] == B:
        init["v"][:, 0], init["v"][:, -1] = 0, 0
    return init


def run_oracle(cfg, steps, dt):
    g = getattr(OS, cfg["ctor"])(**cfg["kw"])
    st = OH.HydrostaticState(g, tracers=("T", "S"), buoyancy=cfg["buoyancy"], substeps=cfg["substeps"], momentum_advection=cfg["madv"],
                             coriolis=cfg["coriolis"], tracer_advection=cfg["scheme"], closure=cfg["closure"])
    init = fields_of(cfg)
    st.u.set(init["u"]); st.v.set(init["v"]); st.free_surface.eta.set(init["eta"]); st.tracers["T"].set(init["T"]); st.tracers["S"].set(init["S"])
    OH.update_state(st)
    for q in range(steps):
        OH.time_step(st, dt, euler=(q == 0))
    return st


def run_rank(ocn, ctx, r, cfg, steps, dt):
    H = ocn.hydrostatic
    R = cfg["R"]
    grid = getattr(H, cfg["ctor"])(arch=ctx, partition="y" if R > 1 else None, **cfg["kw"])
    st = H.HydrostaticState(grid, tracers=("T", "S"), buoyancy=cfg["buoyancy"], substeps=cfg["substeps"], momentum_advection=cfg["madv"],
                            coriolis=cfg["coriolis"], tracer_advection=cfg["scheme"], barotropic_overlap=cfg["overlap"], closure=cfg["closure"])
    init = fields_of(cfg)
    j0, nl = grid.j0, grid.Ny
    fg = st.free_surface.grid
    st.u.set(init["u"][:, j0:j0 + nl])
    vl = np.zeros(st.v.interior().shape)
    src = init["v"][:, j0:j0 + vl.shape[1]]
    vl[:, :src.shape[1]] = src
    st.v.set(vl)
    st.free_surface.eta.set(init["eta"][:, fg.j0:fg.j0 + fg.Ny])
    st.tracers["T"].set(init["T"][:, j0:j0 + nl]); st.tracers["S"].set(init["S"][:, j0:j0 + nl])
    H.update_state(st)
    for q in range(steps):
        H.time_step(st, dt, euler=(q == 0))
    last = r == R - 1
    nv = nl + (1 if (cfg["ybounded"] and last) else 0)
    return {"j0": j0, "u": st.u.interior(), "v": st.v.interior()[:, :nv], "w": st.w.interior(), "T": st.tracers["T"].interior(),
            "S": st.tracers["S"].interior(), "pHY": st.pHY.interior(), "eta": st.free_surface.eta.interior()[:, j0 - fg.j0:j0 - fg.j0 + nl]}


def check(cfg, outs, so):
    exact = cfg["scheme"] == "CenteredSecondOrder" and cfg["madv"] != "WENOVectorInvariantVorticityStencil"
    want = {"u": so.u.interior(), "v": so.v.interior(), "w": so.w.interior(), "T": so.tracers["T"].interior(), "S": so.tracers["S"].interior(),
            "pHY": so.pHY.interior(), "eta": so.free_surface.eta.interior()}
    for o in outs:
        for k, wv in want.items():
            got = o[k].reshape(o[k].shape[0], o[k].shape[1], -1)
            ref = wv.reshape(wv.shape[0], wv.shape[1], -1)[:, o["j0"]:o["j0"] + got.shape[1]]
            if exact:
                ok = np.array_equal(got, ref) or np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)   # libm vs NumPy sines
            else:
                ok = np.abs(got - ref).max() <= 2e-11 * max(np.abs(wv).max(), 1e-300)
            assert ok, (cfg, k, o["j0"], float(np.abs(got - ref).max()), float(np.abs(ref).max()))
    assert np.isfinite(want["u"]).all()


@pytest.mark.parametrize("seed", range(24))
def test_random_hydrostatic_configurations_hostemu(ocn, backend, seed):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    cfg = draw(seed)
    steps, dt = 2, 100.0
    so = run_oracle(cfg, steps, dt)
    outs = run_ranks(ocn, cfg["R"], lambda ctx, r: run_rank(ocn, ctx, r, cfg, steps, dt))
    check(cfg, outs, so)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 112))
def test_random_hydrostatic_configurations_gpu(ocn, seed):
    cfg = draw(seed)
    cfg["R"], cfg["overlap"] = 1, 0
    cfg["kw"]["size"] = (cfg["kw"]["size"][0], cfg["kw"]["size"][1], cfg["kw"]["size"][2])
    steps, dt = 2, 100.0
    so = run_oracle(cfg, steps, dt)
    check(cfg, [run_rank(ocn, ocn.hydrostatic.default_context(), 0, cfg, steps, dt)], so)
