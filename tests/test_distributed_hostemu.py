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


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("stepper,adv", [("AB2", "WENO5"), ("RK3", "WENO5"), ("AB2", "C2")])
def test_slab_trajectory_matches_single_domain_oracle(ocn, backend, R, stepper, adv):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    N = (8, 8, 8 * R)
    rng = np.random.default_rng(5)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    tracers = ("c",) if stepper == "RK3" else ()
    for t in tracers:
        init[t] = rng.random(N)
    og = O.RectilinearGrid(size=N, extent=(1, 1, float(R)), topology=(P,) * 3)
    om = O.NonhydrostaticModel(og, advection=O.WENO5() if adv == "WENO5" else O.CenteredSecondOrder(), timestepper=stepper,
                               tracers=tracers)
    O.set_model(om, **init)
    dt = 2e-3
    for _ in range(2):
        O.time_step(om, dt)

    def rank_fn(ctx, r):
        g = ocn.RectilinearGrid(ctx, size=N, extent=(1, 1, float(R)), topology=(P,) * 3)
        m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5() if adv == "WENO5" else ocn.CenteredSecondOrder(),
                                    timestepper=stepper, tracers=tracers)
        nz = N[2] // R
        ocn.set_model(m, **{n: a[:, :, r * nz:(r + 1) * nz] for n, a in init.items()})
        for _ in range(2):
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
