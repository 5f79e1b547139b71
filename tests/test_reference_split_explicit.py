"""SplitExplicitFreeSurface (BASELINE config 5, first slice): the reference's own analytic known answers, asserted with the
reference's tolerances on the oracle, on the host emulation of the library (`-m "not gpu"`) and on libocnhip.so (`-m gpu`):

  * test/test_split_explicit_free_surface_solver.jl:20-52   one substep of  d_t U = -d_x eta  from eta = sin x   (< 1e-3)
  *                                          :55-100  a full period of the linear wave returns (U < 1e-3, eta < 1e-6)
  *                                          :103-156 uniform fields stay what they are, averages included (eps(100))
  *                                          :160-251 forced two-dimensional wave: eta, U, V and their time averages (1e-2), mean(eta) kept
  * test/test_split_explicit_vertical_integrals.jl:43-148 set_average_to_zero!, barotropic_mode! (inexact and exact integrals),
    barotropic_split_explicit_corrector!
  * test/test_grids.jl:389-485 LatitudeLongitudeGrid: topology, spacings and node positions of the two basic grids

and, beyond the reference's tests: the library against the oracle bit for bit... to round-off (whole parent arrays, halos
included) on a LatitudeLongitudeGrid, and the two-launch hipGraph train of ocn_sefs_substeps against the five-launch substeps.
"""
import numpy as np
import pytest

from oracle import split_explicit as OS

P, B, C, F = "Periodic", "Bounded", "Center", "Face"
EPS = np.finfo(float).eps


class OracleBackend:
    name = "oracle"
    HRectilinearGrid, LatitudeLongitudeGrid = OS.HRectilinearGrid, OS.LatitudeLongitudeGrid
    SplitExplicitFreeSurface = OS.SplitExplicitFreeSurface

    @staticmethod
    def field3(grid, lx, ly):
        return OS.Field3(grid, lx, ly)

    @staticmethod
    def reduced(grid, lx, ly):
        return OS.ReducedField(grid, lx, ly)


class LibBackend:
    def __init__(self, ocn):
        H = ocn.hydrostatic
        self.name = "library"
        self.HRectilinearGrid, self.LatitudeLongitudeGrid = H.HRectilinearGrid, H.LatitudeLongitudeGrid
        self.SplitExplicitFreeSurface = H.SplitExplicitFreeSurface
        self.field3 = lambda grid, lx, ly: H.HField(grid, (lx, ly, C))
        self.reduced = lambda grid, lx, ly: H.HField(grid, (lx, ly, H.Nothing))


def interior2(f):
    a = f.interior()
    return a.reshape(a.shape[0], a.shape[1]) if a.ndim == 3 and a.shape[2] == 1 else a


def _backend(kind, ocn, backend):
    if kind == "oracle":
        return OracleBackend
    if kind == "hostemu" and backend != "hostemu":
        pytest.skip("host-emulation run only")
    if kind == "gpu" and backend != "gpu":
        pytest.skip("HIP run only")
    return LibBackend(ocn)


KINDS = ["oracle", "hostemu", pytest.param("gpu", marks=pytest.mark.gpu)]
Nx, Ny, Nz = 128, 64, 16
Lx = Ly = Lz = 2 * np.pi


def solver_setup(be):
    grid = be.HRectilinearGrid(size=(Nx, Ny, Nz), x=(0, Lx), y=(0, Ly), z=(-Lz, 0), halo=(1, 1, 1), topology=(P, P, B))
    sefs = be.SplitExplicitFreeSurface(grid)
    g = OS.G_EARTH
    sefs.Hfc.set(1 / g)                                  # `Hᶠᶜ .= 1 / g`
    sefs.Hcf.set(1 / g)
    for f in (sefs.etabar, sefs.Ubar, sefs.Vbar, sefs.GU, sefs.GV, sefs.U, sefs.V):
        f.set(0.0)
    return grid, sefs


@pytest.mark.parametrize("kind", KINDS)
def test_one_substep(kind, ocn, backend):
    """:20-52"""
    be = _backend(kind, ocn, backend)
    grid, sefs = solver_setup(be)
    sefs.eta.set(lambda x, y: np.sin(x))
    sefs.substep(1.0, 1)
    xf = grid.nodes(F, 0).reshape(-1, 1)
    assert np.abs(-np.cos(xf) - interior2(sefs.U)).max() < 1e-3


@pytest.mark.parametrize("kind", KINDS)
def test_full_period_returns(kind, ocn, backend):
    """:55-100"""
    be = _backend(kind, ocn, backend)
    grid, sefs = solver_setup(be)
    T = 2 * np.pi
    dtau = 2 * np.pi / max(Nx, Ny) * 5e-2
    Nt = int(np.floor(T / dtau))
    sefs.eta.set(lambda x, y: np.sin(x))
    eta0 = interior2(sefs.eta).copy()
    if be.name == "library":                             # the same substep (index 1: weights play no role here) Nt times
        sefs.set_weights(np.zeros(Nt + 1), np.zeros(Nt + 1))
        sefs.substeps_train(dtau, 1, Nt, fused=False)
    else:
        for _ in range(Nt):
            sefs.substep(dtau, 1)
    sefs.substep(T - Nt * dtau, 1)
    assert np.abs(interior2(sefs.U)).max() < 1e-3
    assert np.abs(interior2(sefs.eta) - eta0).max() < 1e-6


@pytest.mark.parametrize("kind", KINDS)
def test_averaging_do_nothing(kind, ocn, backend):
    """:103-156"""
    be = _backend(kind, ocn, backend)
    grid, sefs = solver_setup(be)
    dtau = 2 * np.pi / max(Nx, Ny) * 5e-2
    sefs.eta.set(1.0)
    sefs.U.set(2.0)
    sefs.V.set(3.0)
    for _ in range(sefs.substeps):                       # index 1 every time, as in the reference: weight 1 / substeps each
        sefs.substep(dtau, 1)
    tol = np.spacing(100.0)                              # eps(100.0)
    for f, want in ((sefs.U, 2.0), (sefs.eta, 1.0), (sefs.V, 3.0), (sefs.Ubar, 2.0), (sefs.etabar, 1.0), (sefs.Vbar, 3.0)):
        assert np.abs(interior2(f) - want).max() < tol


@pytest.mark.parametrize("kind", KINDS)
def test_forced_two_dimensional_wave(kind, ocn, backend):
    """:160-251"""
    be = _backend(kind, ocn, backend)
    grid, sefs = solver_setup(be)
    kx, ky = 2, 3
    om = np.sqrt(kx ** 2 + ky ** 2)
    T = 2 * np.pi / om / 3 * 2
    dtau = 2 * np.pi / max(Nx, Ny) * 1e-2
    Nt = int(np.floor(T / dtau))
    dtau_end = T - Nt * dtau
    gu_c, gv_c = 1.0, 2.0
    eta0 = lambda x, y: np.sin(kx * x) * np.sin(ky * y) + 1     # noqa: E731
    sefs.eta.set(eta0)
    mean_before = interior2(sefs.eta).mean()
    sefs.GU.set(gu_c)
    sefs.GV.set(gv_c)
    w = np.ones(Nt + 1) / Nt
    w[-1] = dtau_end / T
    sefs.set_weights(w, w)
    if be.name == "library":
        sefs.substeps_train(dtau, 1, Nt, fused=True)     # the hot path: two launches per substep, one hipGraph on the GPU
    else:
        for i in range(1, Nt + 1):
            sefs.substep(dtau, i)
    sefs.substep(dtau_end, Nt + 1)
    assert abs(interior2(sefs.eta).mean() - mean_before) < np.spacing(10.0)
    xc, yc = grid.nodes(C, 0).reshape(-1, 1), grid.nodes(C, 1).reshape(1, -1)
    xf, yf = grid.nodes(F, 0).reshape(-1, 1), grid.nodes(F, 1).reshape(1, -1)
    e0 = eta0(xc, yc)
    U0 = kx * np.cos(kx * xf) * np.sin(ky * yc)
    V0 = ky * np.sin(kx * xc) * np.cos(ky * yf)
    eta_exact = np.cos(om * T) * (e0 - 1) + 1
    U_exact = -(np.sin(om * T) / om) * U0 + gu_c * T
    V_exact = -(np.sin(om * T) / om) * V0 + gv_c * T
    etabar_exact = (np.sin(om * T) / om) / T * (e0 - 1) + 1
    Ubar_exact = (np.cos(om * T) / om ** 2 - 1 / om ** 2) / T * U0 + gu_c * T / 2
    Vbar_exact = (np.cos(om * T) / om ** 2 - 1 / om ** 2) / T * V0 + gv_c * T / 2
    tol = 1e-2
    for f, ex in ((sefs.U, U_exact), (sefs.V, V_exact), (sefs.eta, eta_exact)):
        assert np.abs(interior2(f) - ex).max() / np.abs(ex).max() < tol
    for f, ex in ((sefs.Ubar, Ubar_exact), (sefs.Vbar, Vbar_exact), (sefs.etabar, etabar_exact)):
        assert np.abs(interior2(f) - ex).max() < tol


# ---- test_split_explicit_vertical_integrals.jl -------------------------------------------------------------------------------
def integrals_setup(be):
    grid = be.HRectilinearGrid(size=(128, 64, 32), x=(0, Lx), y=(0, Ly), z=(-Lz, 0), halo=(3, 3, 3), topology=(P, P, B))
    sefs = be.SplitExplicitFreeSurface(grid)
    return grid, sefs, be.field3(grid, F, C), be.field3(grid, C, F)


@pytest.mark.parametrize("kind", KINDS)
def test_average_to_zero(kind, ocn, backend):
    """:43-56"""
    be = _backend(kind, ocn, backend)
    grid, sefs, u, v = integrals_setup(be)
    for f in (sefs.etabar, sefs.Ubar, sefs.Vbar):
        f.set(1.0)
    sefs.set_average_to_zero()
    for f in (sefs.etabar, sefs.Ubar, sefs.Vbar):
        f.fill_halo_regions()
        assert np.all(f.parent() == 0.0)


@pytest.mark.parametrize("kind", KINDS)
def test_vertical_integrals(kind, ocn, backend):
    """:58-112"""
    be = _backend(kind, ocn, backend)
    grid, sefs, u, v = integrals_setup(be)
    U, V = sefs.U, sefs.V
    xf = grid.nodes(F, 0).reshape(-1, 1)
    xc, yf = grid.nodes(C, 0).reshape(-1, 1), grid.nodes(F, 1).reshape(1, -1)
    # "Inexact integration" (:58-79): midpoint rule of cos(pi z / 2 Lz), tolerance 1e-3 (one-sided, as written)
    u.set(lambda x, y, z: np.cos((np.pi / 2) * z / Lz))
    sefs.barotropic_mode(U, V, u, v)
    assert np.all(interior2(U) - 2 * Lz / np.pi < 1e-3)
    v.set(lambda x, y, z: np.sin(x * y) * np.cos((np.pi / 2) * z / Lz))
    sefs.barotropic_mode(U, V, u, v)
    assert np.all(interior2(V) - np.sin(xc * yf) * (2 * Lz / np.pi) < 1e-3)
    # "Vertical Integral" (:81-112)
    u.set(0.0)
    U.set(1.0)
    sefs.barotropic_mode(U, V, u, v)
    assert np.all(U.parent() == 0.0)
    u.set(1.0)
    U.set(1.0)
    sefs.barotropic_mode(U, V, u, v)
    assert np.allclose(interior2(U), Lz, rtol=np.sqrt(EPS), atol=0)
    u.set(lambda x, y, z: np.sin(x))
    sefs.barotropic_mode(U, V, u, v)
    assert np.allclose(interior2(U), np.sin(xf) * Lz + 0 * interior2(U), rtol=np.sqrt(EPS), atol=1e-13)
    v.set(lambda x, y, z: np.sin(x) * z * np.cos(y))
    sefs.barotropic_mode(U, V, u, v)
    assert np.allclose(interior2(V), -np.sin(xc) * Lz ** 2 / 2.0 * np.cos(yf), rtol=np.sqrt(EPS), atol=1e-13)


@pytest.mark.parametrize("kind", KINDS)
def test_barotropic_correction(kind, ocn, backend):
    """:114-148"""
    be = _backend(kind, ocn, backend)
    grid, sefs, u, v = integrals_setup(be)
    u.set(lambda x, y, z: z + Lz / 2 + np.sin(x))
    sefs.Ubar.set(lambda x, y: np.cos(x) * Lz)
    v.set(lambda x, y, z: (z + Lz / 2) * np.sin(y) + np.sin(x))
    sefs.Vbar.set(lambda x, y: (np.cos(x) + x) * Lz)
    sefs.Hfc.set(Lz)
    sefs.Hcf.set(Lz)
    sefs.corrector(u, v)
    xf, xc, yf = grid.nodes(F, 0).reshape(-1, 1, 1), grid.nodes(C, 0).reshape(-1, 1, 1), grid.nodes(F, 1).reshape(1, -1, 1)
    zc = (-Lz + (np.arange(32) + 0.5) * (Lz / 32)).reshape(1, 1, -1)
    assert np.all(u.interior() - (zc + Lz / 2 + np.cos(xf)) < 1e-14)
    assert np.all(v.interior() - ((zc + Lz / 2) * np.sin(yf) + np.cos(xc) + xc) < 1e-14)
    assert np.abs(u.interior() - (zc + Lz / 2 + np.cos(xf))).max() < 1e-13           # two-sided (the reference's is one-sided)


# ---- LatitudeLongitudeGrid (test_grids.jl:389-485) -------------------------------------------------------------------------------
def _llg_arrays(be, grid):
    if be is OracleBackend:
        return grid.ax[0].F, grid.ax[0].C, grid.ax[1].F, grid.ax[1].C, grid.topo
    return grid.metric(6), grid.metric(7), grid.metric(8), grid.metric(9), grid.topology


@pytest.mark.parametrize("kind", KINDS)
def test_basic_lat_lon_grids(kind, ocn, backend):
    be = _backend(kind, ocn, backend)
    # bounded domain (:389-433)
    N = 18
    grid = be.LatitudeLongitudeGrid(size=(N, N, 1), longitude=(-90, 90), latitude=(-45, 45), z=(0, 1), halo=(1, 1, 1))
    lamF, lamC, phiF, phiC, topo = _llg_arrays(be, grid)
    assert tuple(topo) == (B, B, B)
    assert len(lamF) == N + 2 + 1 and len(lamC) == N + 2 and len(phiF) == N + 2 + 1 and len(phiC) == N + 2
    assert lamF[1] == -90 and lamF[N + 1] == 90 and phiF[1] == -45 and phiF[N + 1] == 45
    assert lamF[0] == -90 - 10 and lamF[N + 2] == 90 + 10 and phiF[0] == -45 - 5 and phiF[N + 2] == 45 + 5
    assert np.allclose(np.diff(lamF), 10, rtol=0, atol=1e-12) and np.allclose(np.diff(phiC), 5, rtol=0, atol=1e-12)
    # periodic domain (:435-480)
    Nl, Np = 36, 32
    grid = be.LatitudeLongitudeGrid(size=(Nl, Np, 1), longitude=(-180, 180), latitude=(-80, 80), z=(0, 1), halo=(1, 1, 1))
    lamF, lamC, phiF, phiC, topo = _llg_arrays(be, grid)
    assert tuple(topo) == (P, B, B)
    assert len(lamF) == Nl + 2 and len(lamC) == Nl + 2 and len(phiF) == Np + 2 + 1 and len(phiC) == Np + 2
    assert lamF[1] == -180 and lamF[Nl] == 180 - 10 and phiF[1] == -80 and phiF[Np + 1] == 80
    assert lamF[0] == -180 - 10 and lamF[Nl + 1] == 180 and phiF[0] == -80 - 5 and phiF[Np + 2] == 80 + 5
    # metrics (latitude_longitude_grid.jl:436-445): the sphere's area between the two latitude circles, summed over the cells
    if be is OracleBackend:
        Az, dxfc = grid.Az_cc, grid.dx_fc
    else:
        Az, dxfc = grid.metric(4), grid.metric(0)
    R = OS.R_EARTH
    assert np.isclose(Nl * Az[1:Np + 1].sum(), 2 * np.pi * R ** 2 * 2 * np.sin(np.deg2rad(80)), rtol=1e-13)
    assert np.allclose(dxfc[1:Np + 1], R * np.cos(np.deg2rad(phiC[1:Np + 1])) * np.deg2rad(10), rtol=1e-14)


# ---- the library against the oracle on a sphere, and the fused train against the plain substeps ----------------------------------
def _llg_case(be, seed=4):
    grid = be.LatitudeLongitudeGrid(size=(48, 30, 6), longitude=(-180, 180), latitude=(-60, 75), z=[-4000, -2500, -1500, -800, -300, -100, 0],
                                    halo=(3, 3, 3))
    sefs = be.SplitExplicitFreeSurface(grid, substeps=12)
    rng = np.random.default_rng(seed)
    sefs.eta.set(rng.standard_normal((48, 30)))
    Gn = [be.field3(grid, F, C), be.field3(grid, C, F)]
    Gm = [be.field3(grid, F, C), be.field3(grid, C, F)]
    for f in Gn + Gm:
        f.set(1e-5 * rng.standard_normal(f.interior().shape))
    u, v = be.field3(grid, F, C), be.field3(grid, C, F)
    u.set(rng.standard_normal(u.interior().shape))
    v.set(rng.standard_normal(v.interior().shape))
    return grid, sefs, Gn, Gm, u, v


def _assert_same(a, b, tol):
    for name in ("eta", "U", "V", "etabar", "Ubar", "Vbar", "GU", "GV", "Hfc", "Hcf", "Hcc"):
        x, y = getattr(a, name).parent(), getattr(b, name).parent()
        assert x.shape == y.shape, name
        assert np.abs(x - y).max() <= tol * max(np.abs(x).max(), 1e-300), name


@pytest.mark.parametrize("kind", ["hostemu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_step_on_the_sphere_matches_oracle(kind, ocn, backend):
    """split_explicit_free_surface_step! followed by the corrector on a (Periodic, Bounded, Bounded) LatitudeLongitudeGrid with
    stretched z: every parent array of the free surface (halos included) and the corrected u, v against the oracle"""
    be = _backend(kind, ocn, backend)
    res = []
    for b in (OracleBackend, be):
        grid, sefs, Gn, Gm, u, v = _llg_case(b)
        dt = 200.0
        sefs.step(Gn[0], Gn[1], Gm[0], Gm[1], dt, 0.1)
        sefs.corrector(u, v)
        res.append((sefs, u, v))
    _assert_same(res[0][0], res[1][0], 1e-12)
    for q in (1, 2):
        x, y = res[0][q].parent(), res[1][q].parent()
        assert np.abs(x - y).max() <= 1e-12 * np.abs(x).max()


@pytest.mark.parametrize("size", [(40, 24), (72, 20), (66, 34), pytest.param((200, 90), marks=pytest.mark.gpu)],
                         ids=["40x24", "72x20", "66x34", "200x90"])
@pytest.mark.parametrize("topo_y", [P, B])
@pytest.mark.parametrize("kind", ["hostemu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_fused_train_is_bitwise_the_plain_substeps(kind, topo_y, size, ocn, backend):
    """ocn_sefs_substeps(fused = 1, 2, 3) -- two launches per substep, one launch per substep, four or eight substeps per launch (tiles with
    ghost rings; four from 64 x 16 cells up, eight from 64 x 32 up: 72 x 20, 66 x 34 and 200 x 90 have ragged last tiles), replayed from hipGraphs on the GPU -- leave exactly the bits
    of the reference's five-launch substeps in every parent array, halos included (periodic and wall-bounded y)"""
    be = _backend(kind, ocn, backend)
    out = []
    for fused in (0, 1, 2, 3):
        if topo_y == P:
            grid = be.HRectilinearGrid(size=size + (4,), x=(0, 3.0), y=(0, 2.0), z=(-100, 0), halo=(3, 3, 3), topology=(P, P, B))
        else:
            grid = be.LatitudeLongitudeGrid(size=size + (4,), longitude=(-180, 180), latitude=(-70, 70), z=(-100, 0), halo=(3, 3, 3))
        sefs = be.SplitExplicitFreeSurface(grid, substeps=10)
        rng = np.random.default_rng(8)
        sefs.eta.set(rng.standard_normal(size))
        sefs.GU.set(1e-3 * rng.standard_normal(sefs.GU.interior().shape))
        sefs.GV.set(1e-3 * rng.standard_normal(sefs.GV.interior().shape))
        sefs.GU.fill_halo_regions()
        sefs.GV.fill_halo_regions()
        w = rng.random(10)
        sefs.set_weights(w / w.sum(), w[::-1] / w.sum())
        dtau = 0.5 if topo_y == P else 20.0
        for rep in range(3):                              # the second and third call replay the recorded train
            sefs.substeps_train(dtau, 1, 10, fused=fused)
        sefs.substeps_train(dtau, 1, 7, fused=fused)      # an odd train: the one-launch form ends in its second set and copies home
        sefs.substeps_train(dtau, 3, 2, fused=fused)      # 1 + 1 substeps
        sefs.substeps_train(dtau, 2, 6, fused=fused)      # 4 + 1 + 1: a short multi-substep launch in the middle
        out.append(sefs)
        if fused and kind == "gpu":
            assert sefs.graph_replays >= 3
    for other in out[1:]:                                   # two launches per substep, one launch per substep, four substeps per launch
        for name in ("eta", "U", "V", "etabar", "Ubar", "Vbar"):
            x, y = getattr(out[0], name).parent(), getattr(other, name).parent()
            assert np.isfinite(x).all() and np.array_equal(x, y), name
