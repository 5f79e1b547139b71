"""HydrostaticFreeSurfaceModel, AB2 time step around the tendency evaluation (BASELINE config 5, second slice): `ab2_step!`, the
barotropic correction, `store_tendencies!`, `update_state!` (`compute_w_from_continuity!`, `update_hydrostatic_pressure!`,
halo fills incl. the z direction and a ZFaceField).

parity unpinned: the reference's tests for these pieces are "a time step runs / the field has the right type"
(test/test_hydrostatic_free_surface_models.jl:11-47,121-209) -- no known answers.  What is asserted instead:
  * properties the reference's equations imply, on the oracle, the host emulation and libocnhip.so: discrete continuity of
    (u, v, w) after `update_state!`, hydrostatic balance dz(pHY') = b at the faces, the AB2 formula, a resting ocean with
    zero tendencies stays at rest, the corrected velocities carry the free surface's time-averaged barotropic transport;
  * the library against the oracle on whole parent arrays, halos included -- bit for bit (these kernels are compiled without
    contraction into FMAs and sum in the oracle's order);
  * the merged passes of `ocn_hydro_step_after_tendencies(fused = 1)` against the kernel-by-kernel sequence, bit for bit.
"""
import numpy as np
import pytest

from oracle import hydrostatic as OH
from oracle import split_explicit as OS

P, B, C, F = "Periodic", "Bounded", "Center", "Face"
KINDS = ["oracle", "hostemu", pytest.param("gpu", marks=pytest.mark.gpu)]
TS = ("TS", 9.80665, 1.67e-4, 7.8e-4, "T", "S")


class OracleBackend:
    name = "oracle"
    HRectilinearGrid, LatitudeLongitudeGrid = OS.HRectilinearGrid, OS.LatitudeLongitudeGrid
    H = OH


class LibBackend:
    def __init__(self, ocn):
        self.name = "library"
        self.H = ocn.hydrostatic
        self.HRectilinearGrid, self.LatitudeLongitudeGrid = self.H.HRectilinearGrid, self.H.LatitudeLongitudeGrid


def _backend(kind, ocn, backend):
    if kind == "oracle":
        return OracleBackend
    if kind == "hostemu" and backend != "hostemu":
        pytest.skip("host-emulation run only")
    if kind == "gpu" and backend != "gpu":
        pytest.skip("HIP run only")
    return LibBackend(ocn)


def parent(f):
    return f.parent()


GRIDS = {
    # name: (constructor, kwargs)
    "sphere": ("LatitudeLongitudeGrid", dict(size=(48, 24, 8), longitude=(-180, 180), latitude=(-75, 75),
                                             z=[-4000, -2500, -1500, -900, -500, -250, -100, -30, 0], halo=(3, 3, 3))),
    "sector": ("LatitudeLongitudeGrid", dict(size=(20, 18, 5), longitude=(0, 60), latitude=(15, 75), z=(-1000, 0), halo=(2, 2, 2))),
    "sector3": ("LatitudeLongitudeGrid", dict(size=(20, 18, 6), longitude=(0, 60), latitude=(15, 75), z=(-1000, 0), halo=(3, 3, 3))),
    "box": ("HRectilinearGrid", dict(size=(16, 12, 6), x=(0, 1e5), y=(0, 8e4), z=(-600, 0), halo=(1, 1, 1), topology=(P, P, B))),
    "channel": ("HRectilinearGrid", dict(size=(24, 10, 4), x=(0, 2e5), y=(-5e4, 5e4), z=[-500, -300, -120, -40, 0], halo=(3, 3, 3),
                                         topology=(P, B, B))),
}


def make_state(be, gridname, buoyancy=TS, tracers=("T", "S"), substeps=12, seed=3, amplitude=0.1):
    ctor, kw = GRIDS[gridname]
    grid = getattr(be, ctor)(**kw)
    st = be.H.HydrostaticState(grid, tracers=tracers, buoyancy=buoyancy, substeps=substeps)
    rng = np.random.default_rng(seed)

    def rnd(f, a):
        x = a * rng.standard_normal(f.interior().shape)
        return x

    u, v = rnd(st.u, amplitude), rnd(st.v, amplitude)
    topo = kw.get("topology", None) or ((P if kw.get("longitude", (0, 0))[1] - kw.get("longitude", (0, 0))[0] == 360 else B), B, B)
    if topo[0] == B:
        u[0], u[-1] = 0, 0
    if topo[1] == B:
        v[:, 0], v[:, -1] = 0, 0
    st.u.set(u)
    st.v.set(v)
    for n, c in st.tracers.items():
        if n == "T":
            c.set(lambda x, y, z: 20 + 8e-3 * z + 0.5 * np.cos(np.pi * y / 90) + 0 * x)
        elif n == "S":
            c.set(lambda x, y, z: 35 - 1e-3 * z + 0 * x + 0 * y)
        else:
            c.set(rnd(c, 1.0))
    for n in st.Gn:
        a = 1e-5 if n in ("u", "v") else 1e-6
        gn, gm = rnd(st.Gn[n], a), rnd(st.Gm[n], a)
        st.Gn[n].set(gn)
        st.Gm[n].set(gm)
    st.free_surface.eta.set(0.05 * rng.standard_normal(st.free_surface.eta.interior().shape))
    return grid, st, topo


def all_fields(st):
    fs = st.free_surface
    out = {"u": st.u, "v": st.v, "w": st.w, "pHY": st.pHY, "eta": fs.eta, "U": fs.U, "V": fs.V, "Ubar": fs.Ubar, "Vbar": fs.Vbar,
           "etabar": fs.etabar, "GU": fs.GU, "GV": fs.GV}
    out.update({"c_" + n: c for n, c in st.tracers.items()})
    out.update({"Gm_" + n: c for n, c in st.Gm.items()})
    return {k: parent(f).reshape(parent(f).shape[0], parent(f).shape[1], -1) for k, f in out.items()}


def metrics(gridname):
    """the oracle grid's metrics, for the property checks of any backend (the library's own are compared in test_reference_split_explicit)"""
    ctor, kw = GRIDS[gridname]
    return getattr(OS, ctor)(**kw)


# ---- properties -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gridname", ["sphere", "sector", "channel"])
@pytest.mark.parametrize("kind", KINDS)
def test_update_state_continuity_and_hydrostatic_balance(kind, gridname, ocn, backend):
    be = _backend(kind, ocn, backend)
    grid, st, topo = make_state(be, gridname)
    be.H.update_state(st)
    g = metrics(gridname)
    Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
    u, v, w, p = parent(st.u), parent(st.v), parent(st.w), parent(st.pHY)
    T, S = parent(st.tracers["T"]), parent(st.tracers["S"])
    I, J = slice(Hx, Hx + Nx), slice(Hy, Hy + Ny)
    Ip, Jp = slice(Hx + 1, Hx + Nx + 1), slice(Hy + 1, Hy + Ny + 1)
    row = lambda a: a[Hy:Hy + Ny].reshape(1, -1)            # noqa: E731
    rowp = lambda a: a[Hy + 1:Hy + Ny + 1].reshape(1, -1)   # noqa: E731
    dz = g.dz_centers()
    # w[1] = 0 and, below the top face (which the impenetrable fill zeroes, as the reference does), div_xy + dz(w) = 0
    assert np.all(w[I, J, Hz] == 0) and np.all(w[I, J, Hz + Nz] == 0)
    worst = 0.0
    for k in range(1, Nz):
        div = 1 / row(g.Az_cc) * ((row(g.dy_fc) * u[Ip, J, Hz + k - 1] - row(g.dy_fc) * u[I, J, Hz + k - 1])
                                  + (rowp(g.dx_cf) * v[I, Jp, Hz + k - 1] - row(g.dx_cf) * v[I, J, Hz + k - 1]))
        res = div + (w[I, J, Hz + k] - w[I, J, Hz + k - 1]) / dz[k - 1]
        worst = max(worst, np.abs(res).max() / np.abs(div).max())
    assert worst < 1e-13
    # hydrostatic balance: (pHY'[k] - pHY'[k-1]) / dz^f[k] = I_z(b)[k] at the interior faces; the no-flux halos of T, S, pHY'
    b = TS[1] * (TS[2] * T - TS[3] * S)
    az = g.ax[2]
    for k in range(2, Nz + 1):
        dzf = az.df if az.regular else float(az.d_face(k))
        lhs = (p[I, J, Hz + k - 1] - p[I, J, Hz + k - 2]) / dzf
        rhs = 0.5 * (b[I, J, Hz + k - 1] + b[I, J, Hz + k - 2])
        assert np.abs(lhs - rhs).max() <= 1e-12 * np.abs(rhs).max()
    for a in (T, S, p):
        assert np.array_equal(a[I, J, Hz - 1], a[I, J, Hz]) and np.array_equal(a[I, J, Hz + Nz], a[I, J, Hz + Nz - 1])


@pytest.mark.parametrize("kind", KINDS)
def test_resting_ocean_stays_at_rest(kind, ocn, backend):
    """no flow, flat surface, zero tendencies: u, v, w, eta stay exactly zero and the tracers are untouched"""
    be = _backend(kind, ocn, backend)
    grid, st, _ = make_state(be, "sphere", amplitude=0.0)
    for n in st.Gn:
        st.Gn[n].set(0.0)
        st.Gm[n].set(0.0)
    st.free_surface.eta.set(0.0)
    T0 = st.tracers["T"].interior().copy()
    for fused in (False, True):
        be.H.time_step_after_tendencies(st, 600.0, 0.1, fused=fused)
    for f in (st.u, st.v, st.w, st.free_surface.eta, st.free_surface.U, st.free_surface.V):
        assert np.all(parent(f) == 0)
    assert np.array_equal(st.tracers["T"].interior(), T0)
    assert np.abs(parent(st.pHY)).max() > 0


@pytest.mark.parametrize("kind", KINDS)
def test_ab2_formula_and_barotropic_transport(kind, ocn, backend):
    """tracers: c += dt ((1.5 + chi) G^n - (0.5 + chi) G^-), G^- <- G^n; velocities: the depth integral of the corrected u equals
    the free surface's averaged transport U-bar (that is what the corrector is for)"""
    be = _backend(kind, ocn, backend)
    grid, st, _ = make_state(be, "channel", tracers=("T", "S", "e"))
    dt, chi = 300.0, 0.1
    c0 = {n: c.interior().copy() for n, c in st.tracers.items()}
    gn = {n: f.interior().copy() for n, f in st.Gn.items()}
    gm = {n: f.interior().copy() for n, f in st.Gm.items()}
    be.H.time_step_after_tendencies(st, dt, chi)
    g = metrics("channel")
    for n in st.tracers:
        want = c0[n][:g.Nx, :g.Ny] + dt * ((1.5 + chi) * gn[n] - (0.5 + chi) * gm[n])
        assert np.array_equal(st.tracers[n].interior(), want)
    for n in st.Gn:
        assert np.array_equal(st.Gm[n].interior()[:g.Nx, :g.Ny], gn[n][:g.Nx, :g.Ny])
    dz = g.dz_centers().reshape(1, 1, -1)
    U = (st.u.interior() * dz).sum(axis=2)[:g.Nx, :g.Ny]
    Ubar = st.free_surface.Ubar.interior().reshape(st.free_surface.Ubar.interior().shape[0], -1)[:g.Nx, :g.Ny]
    assert np.abs(U - Ubar).max() <= 1e-12 * np.abs(Ubar).max()


# ---- the library against the oracle, whole parent arrays ----------------------------------------------------------------------
def metrics_identical(st, gridname):
    """True when the library's per-row metrics equal the oracle's bit for bit: every comparison with the oracle is then exact (the
    kernels of this model are compiled without contraction and sum in the oracle's order); where libm and NumPy round a sine or a
    cosine of a latitude differently the comparisons fall back to 1e-12"""
    ctor, kw = GRIDS[gridname]
    og = getattr(OS, ctor)(**kw)
    g = st.grid
    pairs = [(g.metric(0), og.dx_fc), (g.metric(1), og.dx_cf), (g.metric(2), og.dy_fc), (g.metric(3), og.dy_cf), (g.metric(4), og.Az_cc),
             (g.metric(11), og.Az_ff)]
    same = True
    for a, b in pairs:
        ok = np.isfinite(b) & np.isfinite(a[:b.size])
        assert np.allclose(a[:b.size][ok], b[ok], rtol=1e-14, atol=0)
        same &= bool(np.array_equal(a[:b.size][ok], b[ok]))
    return same


def close(got, want, exact, what):
    if exact:
        assert np.array_equal(got, want), f"{what}: max rel {np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)}"
    else:
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300), what


def _compare_with_oracle(be, gridname, buoyancy, tracers, fused, steps=2):
    _, st, _ = make_state(be, gridname, buoyancy=buoyancy, tracers=tracers)
    _, so, _ = make_state(OracleBackend, gridname, buoyancy=buoyancy, tracers=tracers)
    be.H.update_state(st)
    OH.update_state(so)
    rng = np.random.default_rng(11)
    for s in range(steps):
        dt, chi = 400.0, (-0.5 if s == 0 else 0.1)          # the first step of a run is forward Euler (chi = -1/2)
        be.H.time_step_after_tendencies(st, dt, chi, fused=fused)
        OH.time_step_after_tendencies(so, dt, chi)
        for n in so.Gn:                                      # "new tendencies" for the next step
            a = (1e-5 if n in ("u", "v") else 1e-6) * rng.standard_normal(so.Gn[n].interior().shape)
            st.Gn[n].set(a)
            so.Gn[n].set(a)
    got, want = all_fields(st), all_fields(so)
    exact = metrics_identical(st, gridname)
    for k in want:
        close(got[k], want[k], exact, f"{k} on {gridname}")


CASES = [("sphere", TS, ("T", "S")), ("sector", ("b", "b"), ("b",)), ("box", None, ()), ("channel", TS, ("S", "e", "T"))]


@pytest.mark.parametrize("fused", [False, True], ids=["sequence", "fused"])
@pytest.mark.parametrize("gridname,buoyancy,tracers", CASES, ids=[c[0] for c in CASES])
def test_step_matches_oracle_hostemu(gridname, buoyancy, tracers, fused, ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    _compare_with_oracle(LibBackend(ocn), gridname, buoyancy, tracers, fused)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["sequence", "fused"])
@pytest.mark.parametrize("gridname,buoyancy,tracers", CASES, ids=[c[0] for c in CASES])
def test_step_matches_oracle_gpu(gridname, buoyancy, tracers, fused, ocn):
    _compare_with_oracle(LibBackend(ocn), gridname, buoyancy, tracers, fused)


@pytest.mark.gpu
def test_fused_step_bitwise_at_size(ocn):
    """256 x 128 x 32 on the sphere, three steps: the merged passes leave the bits of the kernel-by-kernel sequence everywhere"""
    GRIDS["big"] = ("LatitudeLongitudeGrid", dict(size=(256, 128, 32), longitude=(-180, 180), latitude=(-80, 80), z=(-4000, 0), halo=(3, 3, 3)))
    be = LibBackend(ocn)
    states = [make_state(be, "big", substeps=30)[1] for _ in range(2)]
    for st in states:
        be.H.update_state(st)
    for s in range(3):
        for fused, st in zip((False, True), states):
            be.H.time_step_after_tendencies(st, 200.0, 0.1, fused=fused)
    a, b = all_fields(states[0]), all_fields(states[1])
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["u"]).all() and np.abs(a["w"]).max() > 0


# ---- argument checking ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["hostemu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_entry_points_refuse_mismatched_fields(kind, ocn, backend):
    be = _backend(kind, ocn, backend)
    H = be.H
    grid, st, _ = make_state(be, "box", buoyancy=None, tracers=())
    with pytest.raises(ocn.OcnError):
        H.compute_w_from_continuity(st.v, st.u, st.w)                  # locations swapped
    with pytest.raises(ocn.OcnError):
        H.ab2_step_field(st.u, st.Gn["v"], st.Gm["u"], 1.0, 0.1)       # tendency of another location
    with pytest.raises(ocn.OcnError):
        H.update_hydrostatic_pressure(st.w, None, {})                 # pHY' lives at (Center, Center, Center)
    with pytest.raises(KeyError):
        H.HydrostaticState(grid, tracers=("T",), buoyancy=TS)          # the buoyancy needs a tracer that is not there
    with pytest.raises(ocn.OcnError):                                  # pressures on another grid than the free surface's
        other = be.HRectilinearGrid(size=(8, 8, 4), x=(0, 1), y=(0, 1), z=(-1, 0), halo=(1, 1, 1), topology=(P, P, B))
        H.HydrostaticState(grid, tracers=(), free_surface=H.SplitExplicitFreeSurface(other, substeps=4))


# ---- VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization()) ---------------------------------------------------------------
@pytest.mark.parametrize("kind", KINDS)
def test_implicit_vertical_diffusion_decays_a_cosine_mode_exactly(kind, ocn, backend):
    """ANALYTIC: cos(pi z / L) is an eigenvector of the discrete no-flux operator d_z kappa d_z on a regular z grid, eigenvalue
    -kappa (2 - 2 cos(pi dz / L)) / dz^2: one implicit step multiplies it by 1 / (1 + dt kappa lambda) in every column, and the column
    integral of any profile is conserved (no flux through top and bottom); zero tendencies, no flow"""
    be = _backend(kind, ocn, backend)
    grid = be.LatitudeLongitudeGrid(size=(12, 8, 40), longitude=(-180, 180), latitude=(-60, 60), z=(-1000, 0), halo=(3, 3, 3))
    st = be.H.HydrostaticState(grid, tracers=("T", "S"), buoyancy=None, substeps=4, closure=(1e-2, {"T": 1e-1, "S": 0.0}))
    rng = np.random.default_rng(3)
    st.tracers["T"].set(lambda x, y, z: np.cos(np.pi * z / 1000) + 0 * x + 0 * y)
    st.tracers["S"].set(rng.standard_normal(st.tracers["S"].interior().shape))
    st.u.set(lambda x, y, z: 0.1 * np.cos(np.pi * z / 1000) * np.cos(np.pi * y / 180) + 0 * x)
    T0, S0, u0 = st.tracers["T"].interior().copy(), st.tracers["S"].interior().copy(), st.u.interior().copy()
    be.H.update_state(st)
    dt, dz = 1000.0, 25.0
    lam = (2 - 2 * np.cos(np.pi * dz / 1000)) / dz ** 2
    be.H.time_step_after_tendencies(st, dt, -0.5)                  # G^n = G^- = 0: only the implicit solves (and the free surface) act
    T1, S1 = st.tracers["T"].interior(), st.tracers["S"].interior()
    assert np.abs(T1 / T0 - 1 / (1 + dt * 1e-1 * lam)).max() < 1e-13
    assert np.array_equal(S1, S0)                                   # kappa_S = 0: untouched
    assert np.abs(T1.sum(axis=2) - T0.sum(axis=2)).max() < 1e-12
    # u: the same mode with nu, then the barotropic correction adds a depth-independent velocity: compare the baroclinic parts
    u1 = st.u.interior()
    bc = lambda a: a - a.mean(axis=2, keepdims=True)                # noqa: E731
    assert np.abs(bc(u1) - bc(u0) / (1 + dt * 1e-2 * lam)).max() < 1e-13 * np.abs(u0).max()


CLOSURE_CASES = [("sphere", TS, ("T", "S"), (5e-3, {"T": 2e-3, "S": 1e-3})), ("channel", TS, ("S", "e", "T"), (1e-2, 3e-3)),
                 ("sector", ("b", "b"), ("b",), (0.0, {"b": 4e-3}))]


def _compare_closure(be, gridname, buoyancy, tracers, closure, fused):
    states = []
    for b in (be, OracleBackend):
        _, st, _ = make_state(b, gridname, buoyancy=buoyancy, tracers=tracers)
        if b is OracleBackend:
            st.closure = closure
        else:
            st.set_closure(closure)
        b.H.update_state(st)
        for q in range(2):
            b.H.time_step_after_tendencies(st, 400.0, -0.5 if q == 0 else 0.1, fused=fused)
        states.append(st)
    st, so = states
    got, want = all_fields(st), all_fields(so)
    exact = metrics_identical(st, gridname)
    for k in want:
        close(got[k], want[k], exact, f"{k} on {gridname} with implicit vertical diffusion")


@pytest.mark.parametrize("fused", [False, True], ids=["sequence", "fused"])
@pytest.mark.parametrize("gridname,buoyancy,tracers,closure", CLOSURE_CASES, ids=[c[0] for c in CLOSURE_CASES])
def test_implicit_vertical_diffusion_matches_oracle_hostemu(gridname, buoyancy, tracers, closure, fused, ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    _compare_closure(LibBackend(ocn), gridname, buoyancy, tracers, closure, fused)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["sequence", "fused"])
@pytest.mark.parametrize("gridname,buoyancy,tracers,closure", CLOSURE_CASES, ids=[c[0] for c in CLOSURE_CASES])
def test_implicit_vertical_diffusion_matches_oracle_gpu(gridname, buoyancy, tracers, closure, fused, ocn):
    _compare_closure(LibBackend(ocn), gridname, buoyancy, tracers, closure, fused)
