"""Contracts of the C ABI that a drop-in shim relies on (include/ocnhip.h), on both back ends
(`-m "not gpu"`: host emulation of the same sources; `-m gpu`: libocnhip.so):
  * device pointers of u, v, w, pressures and tracers' *fields* are stable where the header says so,
  * two models on one grid do not disturb each other when one of them needs wider halos
    (the reference builds a new grid: nonhydrostatic_model.jl:140-148),
  * checkpoint / restart through upload / download + ocn_set_clock continues bit-identically
    (OutputWriters/checkpointer.jl:158-180,201-265: u, v, w, tracers, G^n, G^- and the clock are what is stored).
"""
import numpy as np
import pytest

P, B = "Periodic", "Bounded"


def both(fn):
    """run on the host emulation in CPU runs and on the GPU in `-m gpu` runs"""
    cpu = fn
    gpu = pytest.mark.gpu(fn)
    return cpu, gpu


def _pointer_stability(ocn, topo, stepper):
    N = (12, 10, 8)
    rng = np.random.default_rng(2)
    g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=topo)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5(), timestepper=stepper, tracers=("c",))
    wshape = (N[0], N[1], N[2] + 1) if topo[2] == B else N
    w0 = rng.random(wshape) - 0.5
    if topo[2] == B:
        w0[:, :, 0] = 0
        w0[:, :, -1] = 0
    ocn.set_model(m, u=rng.random(N) - 0.5, v=rng.random(N) - 0.5, w=w0, c=rng.random(N))
    before = {n: f.device_ptr for n, f in (("u", m.u), ("v", m.v), ("w", m.w), ("p", m.pNHS))}
    for _ in range(3):
        ocn.time_step(m, 1e-3)
        after = {n: f.device_ptr for n, f in (("u", m.u), ("v", m.v), ("w", m.w), ("p", m.pNHS))}
        assert after == before      # include/ocnhip.h: velocities and pressures never rotate; G^n, G^- and tracers may


@pytest.mark.parametrize("topo", [(P, P, P), (P, P, B)])
@pytest.mark.parametrize("stepper", ["AB2", "RK3"])
def test_velocity_pointers_are_stable(ocn, backend, topo, stepper):
    if backend != "hostemu":
        pytest.skip("host-emulation run; the GPU twin is test_velocity_pointers_are_stable_gpu")
    _pointer_stability(ocn, topo, stepper)


@pytest.mark.gpu
@pytest.mark.parametrize("topo", [(P, P, P), (P, P, B)])
@pytest.mark.parametrize("stepper", ["AB2", "RK3"])
def test_velocity_pointers_are_stable_gpu(ocn, topo, stepper):
    _pointer_stability(ocn, topo, stepper)


def _two_models(ocn):
    import oracle as O
    N = (10, 9, 8)
    rng = np.random.default_rng(6)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=(P, P, P), halo=(1, 1, 1))
    m1 = ocn.NonhydrostaticModel(g, advection=ocn.CenteredSecondOrder())
    ocn.set_model(m1, **init)
    m2 = ocn.NonhydrostaticModel(g, advection=ocn.WENO5())          # needs halo 3: must not touch m1's layout
    ocn.set_model(m2, **init)
    assert m1.halo == (1, 1, 1) and m2.halo == (3, 3, 3)
    assert m1.u.parent().shape == (12, 11, 10) and m2.u.parent().shape == (16, 15, 14)
    oms = []
    for adv, H in ((O.CenteredSecondOrder(), 1), (O.WENO5(), 3)):
        og = O.RectilinearGrid(size=N, extent=(1, 1, 1), topology=(P, P, P), halo=(H, H, H))
        om = O.NonhydrostaticModel(og, advection=adv)
        O.set_model(om, **init)
        oms.append(om)
    for _ in range(2):
        for m, om in ((m1, oms[0]), (m2, oms[1])):
            ocn.time_step(m, 1e-3)
            O.time_step(om, 1e-3)
    for m, om in ((m1, oms[0]), (m2, oms[1])):
        for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS)):
            assert np.abs(a.parent() - b.data).max() <= 2e-11 * np.abs(b.data).max()
        assert np.array_equal(m.u.interior(), m.u.parent()[m.halo[0]:-m.halo[0], m.halo[1]:-m.halo[1], m.halo[2]:-m.halo[2]])


def test_two_models_on_one_grid(ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _two_models(ocn)


@pytest.mark.gpu
def test_two_models_on_one_grid_gpu(ocn):
    _two_models(ocn)


def _checkpoint_roundtrip(ocn, topo, stepper, N):
    """run 2 steps, checkpoint, run 3 more; restore the checkpoint into a NEW model and run the same 3: bit-identical."""
    import ctypes as C
    rng = np.random.default_rng(12)
    wshape = (N[0], N[1], N[2] + 1) if topo[2] == B else N
    w0 = rng.random(wshape) - 0.5
    if topo[2] == B:
        w0[:, :, 0] = 0
        w0[:, :, -1] = 0
    init = dict(u=rng.random(N) - 0.5, v=rng.random(N) - 0.5, w=w0, c=rng.random(N))

    def make():
        g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), topology=topo)
        return ocn.NonhydrostaticModel(g, advection=ocn.WENO5(), timestepper=stepper, tracers=("c",),
                                       closure=ocn.ScalarDiffusivity(nu=1e-3, kappa=1e-3))
    dt = 2e-3
    a = make()
    ocn.set_model(a, **init)
    for _ in range(2):
        ocn.time_step(a, dt)
    # checkpointer.jl:158-180: prognostic fields, both tendency sets, clock
    names = ("u", "v", "w", "c")
    ck = {"fields": {n: f.parent() for n, f in a.prognostic().items()},
          "Gn": {n: a.Gn[n].parent() for n in names}, "Gm": {n: a.Gm[n].parent() for n in names},
          "p": a.pNHS.parent(), "clock": a.clock}
    for _ in range(3):
        ocn.time_step(a, dt)
    want = {n: f.parent() for n, f in a.prognostic().items()}
    want["p"] = a.pNHS.parent()
    b = make()
    for n, f in b.prognostic().items():
        f.set_parent(ck["fields"][n])
    for n in names:
        b.Gn[n].set_parent(ck["Gn"][n])
        b.Gm[n].set_parent(ck["Gm"][n])
    b.pNHS.set_parent(ck["p"])
    t, it, _ = ck["clock"]
    from ocnhip._lib import check
    check(b.lib.ocn_set_clock(b.h, C.c_double(t), C.c_int64(it), C.c_double(dt)), b.ctx.h)
    ocn.update_state(b)          # set!(model, checkpoint) ends with update_state! (checkpointer.jl:262)
    for _ in range(3):
        ocn.time_step(b, dt)
    got = {n: f.parent() for n, f in b.prognostic().items()}
    got["p"] = b.pNHS.parent()
    assert b.clock[1] == a.clock[1] and b.clock[0] == a.clock[0]
    for n in want:
        assert np.array_equal(want[n], got[n]), (n, np.abs(want[n] - got[n]).max())


CK = [((P, P, P), "AB2", (12, 10, 8)), ((P, P, P), "RK3", (12, 10, 8)), ((P, P, B), "AB2", (12, 10, 8)), ((P, P, B), "RK3", (10, 8, 8))]


@pytest.mark.parametrize("topo,stepper,N", CK)
def test_checkpoint_restart_is_bit_identical(ocn, backend, topo, stepper, N):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _checkpoint_roundtrip(ocn, topo, stepper, N)


@pytest.mark.gpu
@pytest.mark.parametrize("topo,stepper,N", CK + [((P, P, P), "AB2", (256, 16, 12))])
def test_checkpoint_restart_is_bit_identical_gpu(ocn, topo, stepper, N):
    _checkpoint_roundtrip(ocn, topo, stepper, N)


def test_model_reports_its_kernel_path(ocn, backend):
    """falling off the tiled kernels is never silent: ocn_model_path names the path and the reason"""
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    g = ocn.RectilinearGrid(size=(12, 10, 8), extent=(1, 1, 1), topology=(P, P, P))
    assert "all-in-one" in ocn.NonhydrostaticModel(g, advection=ocn.WENO5()).kernel_path
    assert "advection scheme" in ocn.NonhydrostaticModel(g, advection=ocn.CenteredSecondOrder()).kernel_path
    gb = ocn.RectilinearGrid(size=(12, 10, 8), extent=(1, 1, 1), topology=(P, P, B))
    assert "tiled advection" in ocn.NonhydrostaticModel(gb, advection=ocn.WENO5(), coriolis=ocn.FPlane(1e-4)).kernel_path
    gs = ocn.RectilinearGrid(size=(4, 10, 8), extent=(1, 1, 1), topology=(P, P, P))
    assert "fewer than 6 cells" in ocn.NonhydrostaticModel(gs, advection=ocn.WENO5()).kernel_path


# ---- stand-alone fields (ocn_field_create: Field{LX,LY,LZ}(grid), Fields/field.jl:16-30, Grids/new_data.jl:16-61) ---------
def _standalone_fields(ocn):
    rng = np.random.default_rng(5)
    for topo, N, H in [((P, P, P), (7, 6, 5), (3, 3, 3)), ((P, P, B), (8, 5, 6), (2, 1, 3)), ((B, B, B), (5, 6, 4), (1, 2, 1)),
                       ((P, "Flat", B), (6, 7), (2, 2))]:
        g = ocn.RectilinearGrid(size=N, extent=tuple(1.0 for _ in N), topology=topo, halo=H)
        N3 = [g.Nx, g.Ny, g.Nz]
        H3, it = [], 0
        for t in topo:
            H3.append(0 if t == "Flat" else H[it])
            it += 0 if t == "Flat" else 1
        for loc in [("Center",) * 3, ("Face", "Center", "Center"), ("Center", "Face", "Center"), ("Center", "Center", "Face"),
                    ("Face", "Face", "Face")]:
            f = ocn.Field(loc, g)
            # total_size of new_data: N + 2H, one more for a Face location along a Bounded direction, N alone when Flat
            want = tuple(N3[d] if topo[d] == "Flat" else N3[d] + 2 * H3[d] + (1 if loc[d] == "Face" and topo[d] == B else 0)
                         for d in range(3))
            assert f.total == want and f.halo == tuple(H3)
            assert not f.parent().any()                       # zeros(FT, arch, N...)
            a = rng.random(f.total)
            f.set_parent(a)
            assert np.array_equal(f.parent(), a)              # parent array incl. halos round-trips bit for bit
            assert f.device_ptr
            # the layout equals the layout of a model's own field of the same location on this grid
            m = ocn.NonhydrostaticModel(g, advection=ocn.CenteredSecondOrder(), tracers=("c",))
            if tuple(m.halo) == tuple(H3):
                twin = {("Face", "Center", "Center"): m.u, ("Center", "Face", "Center"): m.v, ("Center", "Center", "Face"): m.w,
                        ("Center",) * 3: m.tracers["c"]}.get(loc)
                if twin is not None:
                    assert twin.total == f.total and twin.layout == f.layout
    import ctypes
    h = ctypes.c_void_p()
    assert ocn._lib.load().ocn_field_create(g.h, 0, 0, 7, ctypes.byref(h)) != 0     # a location that is neither Center nor Face


def test_standalone_fields(ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _standalone_fields(ocn)


@pytest.mark.gpu
def test_standalone_fields_gpu(ocn):
    _standalone_fields(ocn)
