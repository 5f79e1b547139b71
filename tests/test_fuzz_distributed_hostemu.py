"""Seeded random slab runs on host-emulated ranks (threads of one process, in-process mailbox) against the single-domain
oracle: z-slabs of a triply periodic box (2 to 4 ranks, 6 to 13 levels per rank, scheme, stepper, tracers, viscosity, a
changing dt, both distributed solvers, halo planes overlapped with the interior tendency launch or not) and y-slabs of the
(Periodic, Periodic, Bounded) ocean-LES miniature (2 or 3 ranks, random stretched z, AMD or scalar closure, scheme, stepper).
Longer sweeps of the same generators (40 z-slab and 46 y-slab configurations) ran clean while the round was built."""
import copy

import numpy as np
import pytest

import oracle as O
import parity_cases as pc
from test_distributed_hostemu import _build_on, run_ranks

P = "Periodic"


@pytest.mark.parametrize("seed", range(5))
def test_random_zslab_run(ocn, backend, seed, monkeypatch):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    rng = np.random.default_rng(300 + seed)
    R, nzl = int(rng.choice([2, 3, 4])), int(rng.integers(6, 14))
    N = (int(rng.integers(6, 13)), R * int(rng.integers(2, 5)), nzl * R)
    stepper, adv = str(rng.choice(["AB2", "RK3"])), str(rng.choice(["WENO5", "U5", "WENO5JS"]))
    tracers = tuple("abc"[:int(rng.integers(0, 3))])
    nu = float(rng.choice([0.0, 1e-2]))
    monkeypatch.setenv("OCNHIP_OVERLAP", str(rng.choice(["0", "1"])))
    monkeypatch.setenv("OCNHIP_DIST_SOLVER", str(rng.choice(["green", "transpose"])))
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    init.update({t: rng.random(N) for t in tracers})
    ext = (1.0, 1.2, 0.1 * N[2])
    dts = [2e-3, 2e-3, 1e-3] if rng.random() < 0.3 else [2e-3] * 3

    def mk(mod):
        return dict(advection={"WENO5": mod.WENO5(), "U5": mod.UpwindBiasedFifthOrder(), "WENO5JS": mod.WENO5(zweno=False)}[adv],
                    timestepper=stepper, tracers=tracers, closure=(mod.ScalarDiffusivity(nu=nu, kappa=nu) if nu else None))
    om = O.NonhydrostaticModel(O.RectilinearGrid(size=N, extent=ext, topology=(P,) * 3), **mk(O))
    O.set_model(om, **init)
    for dt in dts:
        O.time_step(om, dt)

    def rank_fn(ctx, r):
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ctx, size=N, extent=ext, topology=(P,) * 3), **mk(ocn))
        ocn.set_model(m, **{n: a[:, :, r * nzl:(r + 1) * nzl] for n, a in init.items()})
        for dt in dts:
            ocn.time_step(m, dt)
        out = {n: f.parent() for n, f in (("u", m.u), ("v", m.v), ("w", m.w), ("p", m.pNHS))}
        out.update({t: m.tracers[t].parent() for t in tracers})
        return out
    H = 3
    refs = [("u", om.u.data), ("v", om.v.data), ("w", om.w.data), ("p", om.pNHS.data)] + [(t, om.tracers[t].data) for t in tracers]
    for r, flds in enumerate(run_ranks(ocn, R, rank_fn)):
        for n, ref in refs:
            idx = (np.arange(-H, nzl + H) + r * nzl) % N[2] + H
            assert np.abs(flds[n] - ref[:, :, idx]).max() <= 1e-10 * np.abs(ref).max(), (R, N, stepper, adv, tracers, nu, n)


@pytest.mark.parametrize("seed", range(4))
def test_random_yslab_run(ocn, backend, seed):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    rng = np.random.default_rng(700 + seed)
    R, kind = int(rng.choice([2, 3])), str(rng.choice(["amd", "scalar"]))
    cfg = copy.deepcopy(pc.CASES["ppb_amd_config3" if kind == "amd" else "ppb_weno_full"])
    Nx, Nz, nyl = int(rng.integers(6, 12)), int(rng.integers(6, 10)), int(rng.integers(6, 9))
    Ny = nyl * R
    cfg.update(size=(Nx, Ny, Nz), stepper=str(rng.choice(["AB2", "RK3"])), adv=str(rng.choice(["WENO5", "U5", "C4", "C2"])), steps=2)
    if "zfaces" in cfg:
        zf = [-1.0] + [float(v) for v in np.sort(rng.random(Nz - 1)) - 1.0] + [0.0]
        cfg["zfaces"] = zf if min(np.diff(zf)) >= 2e-2 else list(np.linspace(-1, 0, Nz + 1))
    om = pc.build(O, cfg)
    names = ["u", "v", "w"] + list(cfg["tracers"])
    fld = lambda m, n: getattr(m, n) if n in "uvw" else m.tracers[n]      # noqa: E731
    init = {n: rng.random(fld(om, n).interior().shape) - (0.5 if n in "uvw" else 0.0) for n in names}
    init["w"][:, :, 0] = 0
    init["w"][:, :, -1] = 0
    O.set_model(om, **init)
    for _ in range(cfg["steps"]):
        O.time_step(om, cfg["dt"])

    def rank_fn(ctx, r):
        m = _build_on(ocn, ctx, dict(cfg))
        ocn.set_model(m, **{n: np.ascontiguousarray(a[:, r * nyl:(r + 1) * nyl]) for n, a in init.items()})
        for _ in range(cfg["steps"]):
            ocn.time_step(m, cfg["dt"])
        out = {n: fld(m, n).parent() for n in names}
        out["p"], out["pHY"] = m.pNHS.parent(), m.pHY.parent()
        return out
    H = om.grid.Hy
    refs = {n: fld(om, n).data for n in names}
    refs["p"], refs["pHY"] = om.pNHS.data, om.pHY.data
    for r, flds in enumerate(run_ranks(ocn, R, rank_fn)):
        for n, ref in refs.items():
            idx = (np.arange(-H, nyl + H) + r * nyl) % Ny + H
            assert np.abs(flds[n] - ref[:, idx]).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (R, kind, cfg["size"], cfg["adv"], n)
