"""HydrostaticFreeSurfaceModel on latitude bands (y-slabs) over R ranks -- BASELINE config 5 is "slab-distributed" -- against the
single-domain oracle: R host-emulation ranks as threads of this process (in-process transport of csrc/comm.hip), and, under
`-m gpu`, R processes on the one GPU of the box (tests/test_distributed_procs.py::hydro_bands).

Layout (csrc/splitexplicit.hip): a rank holds Ny / R rows of every 3-D field (Bounded shape: Face-y fields carry one more row, the
upper neighbour's first or the wall); `fill_halo_regions!` exchanges whole rows with the two neighbouring bands; the free surface is
REPLICATED -- it lives on the whole grid on every rank, which sub-cycles the complete barotropic problem after one all-gather of
the bands' rows of U, V, G^U, G^V.  Since every cell sees the operands of the single-domain run, the owned rows agree with it bit
for bit (mirrors the decomposition-independence checks of test/test_distributed_models.jl:361-453)."""
import numpy as np
import pytest

from oracle import hydrostatic as OH
from oracle import split_explicit as OS
from test_distributed_hostemu import run_ranks

P, B = "Periodic", "Bounded"
TS = ("TS", 9.80665, 1.67e-4, 7.8e-4, "T", "S")
OMEGA = 7.292115e-5
CASES = {
    "sphere": ("LatitudeLongitudeGrid", dict(size=(24, 16, 5), longitude=(-180, 180), latitude=(-60, 60), z=[-3000, -1500, -600, -200, -50, 0],
                                             halo=(3, 3, 3)), ("HydrostaticSphericalCoriolis", OMEGA, "EnstrophyConserving")),
    "sphere8": ("LatitudeLongitudeGrid", dict(size=(16, 64, 3), longitude=(-180, 180), latitude=(-80, 80), z=(-2000, 0), halo=(3, 3, 3)),
                ("HydrostaticSphericalCoriolis", OMEGA, "EnstrophyConserving")),
    "sector": ("LatitudeLongitudeGrid", dict(size=(12, 16, 4), longitude=(0, 40), latitude=(10, 70), z=(-800, 0), halo=(2, 2, 2)),
               ("HydrostaticSphericalCoriolis", OMEGA, "EnergyConserving")),
    "periodic_box": ("HRectilinearGrid", dict(size=(12, 16, 4), x=(0, 1e5), y=(0, 2e5), z=(-500, 0), halo=(2, 2, 2), topology=(P, P, B)),
                     ("FPlane", 1e-4)),
}


def initial(gridname, seed=2):
    ctor, kw, _ = CASES[gridname]
    g = getattr(OS, ctor)(**kw)
    st = OH.HydrostaticState(g, tracers=("T", "S"), buoyancy=TS, substeps=10)
    rng = np.random.default_rng(seed)
    init = {"u": 0.05 * rng.standard_normal(st.u.interior().shape), "v": 0.05 * rng.standard_normal(st.v.interior().shape),
            "eta": 0.02 * rng.standard_normal(st.free_surface.eta.interior().shape)}
    if g.topo[0] == B:
        init["u"][0], init["u"][-1] = 0, 0
    if g.topo[1] == B:
        init["v"][:, 0], init["v"][:, -1] = 0, 0
    z = g.nodes("Center", 2).reshape(1, 1, -1) if hasattr(g, "nodes") else None
    init["T"] = 20 + 5e-3 * z + 0.3 * rng.standard_normal(st.tracers["T"].interior().shape)
    init["S"] = 35 + 0.1 * rng.standard_normal(st.tracers["S"].interior().shape)
    return init


def run_single_domain_oracle(gridname, steps, dt):
    ctor, kw, coriolis = CASES[gridname]
    g = getattr(OS, ctor)(**kw)
    st = OH.HydrostaticState(g, tracers=("T", "S"), buoyancy=TS, substeps=10, coriolis=coriolis)
    init = initial(gridname)
    st.u.set(init["u"]); st.v.set(init["v"]); st.free_surface.eta.set(init["eta"])
    st.tracers["T"].set(init["T"]); st.tracers["S"].set(init["S"])
    OH.update_state(st)
    for q in range(steps):
        OH.time_step(st, dt, euler=(q == 0))
    return st


def rows(a, j0, n):
    return a[:, j0:j0 + n]


def band_run(ocn, ctx, r, R, gridname, steps, dt, overlap=0):
    """rank r of R: build the band, set it from the global initial arrays, step, return the interiors of the owned rows"""
    init = initial(gridname)
    ctor, kw, coriolis = CASES[gridname]
    H = ocn.hydrostatic
    grid = getattr(H, ctor)(arch=ctx, partition="y", **kw)
    assert grid.Ny == kw["size"][1] // R and grid.j0 == r * grid.Ny and grid.global_Ny == kw["size"][1]
    st = H.HydrostaticState(grid, tracers=("T", "S"), buoyancy=TS, substeps=10, coriolis=coriolis, barotropic_overlap=overlap)
    j0, nl = grid.j0, grid.Ny
    last = r == R - 1
    fg = st.free_surface.grid                          # the whole grid (replicated free surface) or the extended band (banded)
    if overlap:
        assert fg.Ny == nl + (0 if r == 0 else overlap) + (0 if last else overlap) and fg.j0 == j0 - (0 if r == 0 else overlap)
    else:
        assert fg.Ny == kw["size"][1] and fg.j0 == 0
    facey = kw.get("topology", (None, B))[1] == B
    st.u.set(rows(init["u"], j0, nl))
    vloc = np.zeros(st.v.interior().shape)
    src = rows(init["v"], j0, nl + 1 if facey else nl)
    vloc[:, :src.shape[1]] = src                       # the band's extra row: the upper neighbour's first (the exchange fills it anyway)
    st.v.set(vloc)
    st.free_surface.eta.set(rows(init["eta"], fg.j0, fg.Ny) if overlap else init["eta"])
    st.tracers["T"].set(rows(init["T"], j0, nl))
    st.tracers["S"].set(rows(init["S"], j0, nl))
    H.update_state(st)
    for q in range(steps):
        H.time_step(st, dt, euler=(q == 0))
    return {"u": st.u.interior(), "v": st.v.interior()[:, :nl + (1 if (last and facey) else 0)], "w": st.w.interior(), "pHY": st.pHY.interior(),
            "T": st.tracers["T"].interior(), "S": st.tracers["S"].interior(),
            "eta": st.free_surface.eta.interior()[:, j0 - fg.j0:j0 - fg.j0 + nl],
            "Gm_u": st.Gm["u"].interior()[:, :nl], "j0": j0, "nl": nl}


def band_check(o, so, exact=False):
    """the band's owned rows against the single-domain oracle: bit for bit (exact), or to 1e-11 where the host's libm and NumPy may
    round the sines of the latitudes differently (a decomposition mistake is an O(1) error either way)"""
    want = {"u": so.u.interior(), "v": so.v.interior(), "w": so.w.interior(), "pHY": so.pHY.interior(), "T": so.tracers["T"].interior(),
            "S": so.tracers["S"].interior(), "Gm_u": so.Gm["u"].interior()}
    eta = so.free_surface.eta.interior()[:, o["j0"]:o["j0"] + o["nl"]]
    for k, wv in want.items():
        got = o[k]
        ref = wv[:, o["j0"]:o["j0"] + got.shape[1]]
        ok = np.array_equal(got, ref) or (not exact and np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max())
        assert ok, f"{k} of the band at row {o['j0']}: max abs diff {np.abs(got - ref).max()}"
    assert np.array_equal(o["eta"].reshape(eta.shape), eta) or (not exact and np.abs(o["eta"].reshape(eta.shape) - eta).max() <= 1e-11 * np.abs(eta).max())
    assert np.abs(want["w"]).max() > 0 and np.isfinite(want["u"]).all()


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("gridname", ["sphere", "sector", "periodic_box"])
def test_bands_match_single_domain_oracle_hostemu(ocn, backend, gridname, R):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    steps, dt = 3, 150.0
    so = run_single_domain_oracle(gridname, steps, dt)
    for o in run_ranks(ocn, R, lambda ctx, r: band_run(ocn, ctx, r, R, gridname, steps, dt)):
        band_check(o, so, exact=True)


@pytest.mark.parametrize("R,overlap", [(2, 4), (2, 3), (4, 3), (4, 4)])
@pytest.mark.parametrize("gridname", ["sphere", "sector"])
def test_banded_free_surface_matches_single_domain_oracle_hostemu(ocn, backend, gridname, R, overlap):
    """the free surface on the band extended by `overlap` rows, refreshed every `overlap` substeps (10 substeps: blocks of 4, 4, 2 or
    3, 3, 3, 1) -- the artificial walls of the extended band never reach the band's own rows"""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    steps, dt = 3, 150.0
    so = run_single_domain_oracle(gridname, steps, dt)
    for o in run_ranks(ocn, R, lambda ctx, r: band_run(ocn, ctx, r, R, gridname, steps, dt, overlap=overlap)):
        band_check(o, so, exact=True)


@pytest.mark.parametrize("overlap", [0, 5, 8])
def test_eight_bands_match_single_domain_oracle_hostemu(ocn, backend, overlap):
    """the target machine's rank count: eight latitude bands of eight rows, replicated (overlap 0) and banded free surface (overlap
    rows 5: blocks of 5 + 5; 8: the whole neighbouring band)"""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    steps, dt = 2, 150.0
    so = run_single_domain_oracle("sphere8", steps, dt)
    for o in run_ranks(ocn, 8, lambda ctx, r: band_run(ocn, ctx, r, 8, "sphere8", steps, dt, overlap=overlap)):
        band_check(o, so, exact=True)


def test_band_grid_arguments_hostemu(ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    H = ocn.hydrostatic

    def rank_fn(ctx, r):
        with pytest.raises(ocn.OcnError):       # 10 rows do not split into 4 bands
            H.LatitudeLongitudeGrid(size=(8, 10, 2), longitude=(0, 40), latitude=(0, 40), z=(-10, 0), halo=(1, 1, 1), arch=ctx, partition="y")
        g = H.LatitudeLongitudeGrid(size=(8, 16, 2), longitude=(0, 40), latitude=(0, 40), z=(-10, 0), halo=(1, 1, 1), arch=ctx, partition="y")
        with pytest.raises(ocn.OcnError):       # the free surface is replicated: it wants the whole grid
            H.SplitExplicitFreeSurface(g, substeps=4)
        # the band's latitudes are the global ones of its rows
        og = OS.LatitudeLongitudeGrid(size=(8, 16, 2), longitude=(0, 40), latitude=(0, 40), z=(-10, 0), halo=(1, 1, 1))
        assert np.array_equal(g.nodes("Center", 1), og.nodes("Center", 1)[g.j0:g.j0 + g.Ny])
        return True
    assert all(run_ranks(ocn, 4, rank_fn))
