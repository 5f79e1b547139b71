"""Analytic known answers the reference keeps in test/test_dynamics.jl, run on the oracle, on the host emulation of the
library (`-m "not gpu"`) and on libocnhip.so (`-m gpu`) -- the same set-ups, the same tolerances:

  * "Simple diffusion" (:12-29, :401-408): a constant field stays constant under ScalarDiffusivity(nu=1, kappa=1),
    size (1, 1, 16), ten steps of dt = 1, fields u, v, c, both time steppers (isapprox to pi);
  * "Budgets in isotropic diffusion" (:31-53, :410-458): the mean of a random field is conserved over ten steps on 4^3 grids
    of the four topologies, for c and for every velocity component whose direction is Periodic;
  * "Diffusion of a cosine" (:62-80, :494-600; the ScalarDiffusivity / explicit rows on the three one-direction grids):
    cos(2 xi) decays as exp(-kappa m^2 t), atol = rtol = 1e-6 after five steps;
  * "Internal wave" (test_internal_wave_dynamics.jl:1-77, test_dynamics.jl:625-682): an inertia-gravity wave packet on
    128 x 1 x 128 (Periodic y), 128 x 128 (Flat y), and both again with the z faces given explicitly (Fourier-tridiagonal
    solver): relative error of u against the linear solution below 1e-4 after ten steps.
VerticallyImplicit / Horizontal / Vertical / biharmonic closures, immersed grids, background fields, tilted gravity and
rotation about an arbitrary axis (:260-396) are outside the path: not ported.
"""
import numpy as np
import pytest

import oracle as O

P, B, F = "Periodic", "Bounded", "Flat"
STEPPERS = ["QuasiAdamsBashforth2", "RungeKutta3"]


def _lib(ocn, backend, gpu):
    if gpu and backend != "gpu":
        pytest.skip("HIP run only")
    if not gpu and backend != "hostemu":
        pytest.skip("host-emulation run only")
    return ocn


def _field(m, name):
    return {"u": m.u, "v": m.v, "w": m.w}.get(name) or m.tracers[name]


# ---- simple diffusion ----------------------------------------------------------------------------------------------------
def simple_diffusion(mod, name, stepper):
    g = mod.RectilinearGrid(size=(1, 1, 16), extent=(1, 1, 1), halo=(1, 1, 1))
    m = mod.NonhydrostaticModel(g, timestepper=stepper, advection=mod.CenteredSecondOrder(),
                                closure=mod.ScalarDiffusivity(nu=1, kappa=1), tracers=("c",))
    mod.set_model(m, enforce_incompressibility=False, **{name: np.pi})
    for _ in range(10):
        mod.time_step(m, 1.0)
    a = _field(m, name).interior()
    assert np.allclose(a, np.pi, rtol=np.sqrt(np.finfo(float).eps), atol=0)      # isapprox(value, x)


@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("name", ["u", "v", "c"])
def test_simple_diffusion_oracle(name, stepper):
    simple_diffusion(O, name, stepper)


@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("name", ["u", "v", "c"])
def test_simple_diffusion_library(ocn, backend, name, stepper):
    simple_diffusion(_lib(ocn, backend, False), name, stepper)


@pytest.mark.gpu
@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("name", ["u", "v", "c"])
def test_simple_diffusion_library_gpu(ocn, backend, name, stepper):
    simple_diffusion(_lib(ocn, backend, True), name, stepper)


# ---- budgets --------------------------------------------------------------------------------------------------------------
TOPOS = [(P, P, P), (P, P, B), (P, B, B), (B, B, B)]


def diffusion_budget(mod, topo, stepper):
    names = ["c"] + [n for n, t in zip("uvw", topo) if t == P]
    rng = np.random.default_rng(3)
    for name in names:
        g = mod.RectilinearGrid(size=(4, 4, 4), extent=(1, 1, 1), topology=topo, halo=(1, 1, 1))
        m = mod.NonhydrostaticModel(g, timestepper=stepper, advection=mod.CenteredSecondOrder(),
                                    closure=mod.ScalarDiffusivity(nu=1, kappa=1), tracers=("c",))
        f = _field(m, name)
        mod.set_model(m, enforce_incompressibility=False, **{name: rng.random(f.interior().shape)})
        mean0 = f.interior().mean()
        dt = 1e-4 * (1 / 4) ** 2 / 1.0
        for _ in range(10):
            mod.time_step(m, dt)
        mean1 = _field(m, name).interior().mean()
        assert np.isclose(mean0, mean1, rtol=np.sqrt(np.finfo(float).eps), atol=0), (name, mean0, mean1)


@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("topo", TOPOS, ids=["".join(t[0] for t in T) for T in TOPOS])
def test_diffusion_budget_oracle(topo, stepper):
    diffusion_budget(O, topo, stepper)


@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("topo", TOPOS, ids=["".join(t[0] for t in T) for T in TOPOS])
def test_diffusion_budget_library(ocn, backend, topo, stepper):
    diffusion_budget(_lib(ocn, backend, False), topo, stepper)


@pytest.mark.gpu
@pytest.mark.parametrize("stepper", STEPPERS)
@pytest.mark.parametrize("topo", TOPOS, ids=["".join(t[0] for t in T) for T in TOPOS])
def test_diffusion_budget_library_gpu(ocn, backend, topo, stepper):
    diffusion_budget(_lib(ocn, backend, True), topo, stepper)


# ---- diffusion of a cosine ------------------------------------------------------------------------------------------------
def diffusion_cosine(mod, axis):
    N, L, kap, mw = 128, np.pi / 2, 1.0, 2
    size = [1, 1, 1]
    size[axis] = N
    topo = [P, P, P]
    topo[axis] = B
    ext = {"x": (0, 1), "y": (0, 1), "z": (0, 1)}
    ext["xyz"[axis]] = (0, L)
    names = [n for a, n in enumerate("uvw") if a != axis] + ["c"]      # the fields that are Center-located along `axis`
    for name in names:
        g = mod.RectilinearGrid(size=tuple(size), topology=tuple(topo), halo=(1, 1, 1), **ext)
        m = mod.NonhydrostaticModel(g, advection=mod.CenteredSecondOrder(), closure=mod.ScalarDiffusivity(nu=1, kappa=1),
                                    tracers=("c",))
        xi = (np.arange(N) + 0.5) * (L / N)
        shape = [1, 1, 1]
        shape[axis] = N
        xi = xi.reshape(shape)
        f = _field(m, name)
        mod.set_model(m, enforce_incompressibility=False, **{name: np.cos(mw * xi) + np.zeros(f.interior().shape)})
        Lz = 1.0 if axis != 2 else L
        dt = 1e-6 * Lz ** 2 / kap
        for _ in range(5):
            mod.time_step(m, dt)
        want = np.exp(-kap * mw ** 2 * m.time) * np.cos(mw * xi) + np.zeros(f.interior().shape)
        assert np.allclose(_field(m, name).interior(), want, atol=1e-6, rtol=1e-6), name


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_diffusion_cosine_oracle(axis):
    diffusion_cosine(O, axis)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_diffusion_cosine_library(ocn, backend, axis):
    diffusion_cosine(_lib(ocn, backend, False), axis)


@pytest.mark.gpu
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_diffusion_cosine_library_gpu(ocn, backend, axis):
    diffusion_cosine(_lib(ocn, backend, True), axis)


# ---- internal wave --------------------------------------------------------------------------------------------------------
def internal_wave(mod, flat_y, explicit_faces, N=128):
    Lx = 2 * np.pi
    nu = 1e-9
    z0, dl, a0, mz, kx, f, Nb = -Lx / 3, Lx / 20, 1e-3, 16, 1, 0.2, 1.0
    sig = np.sqrt((Nb ** 2 * kx ** 2 + f ** 2 * mz ** 2) / (kx ** 2 + mz ** 2))
    dt = 0.01 / sig
    cg = mz * sig / (kx ** 2 + mz ** 2) * (f ** 2 / sig ** 2 - 1)
    U = a0 * kx * sig / (sig ** 2 - f ** 2)
    V = a0 * kx * f / (sig ** 2 - f ** 2)
    W = a0 * mz * sig / (sig ** 2 - Nb ** 2)
    Bb = a0 * mz * Nb ** 2 / (sig ** 2 - Nb ** 2)
    env = lambda z, t: np.exp(-(z - cg * t - z0) ** 2 / (2 * dl) ** 2)                 # noqa: E731
    u = lambda x, y, z, t=0.0: env(z, t) * U * np.cos(kx * x + mz * z - sig * t) + 0 * y   # noqa: E731
    v = lambda x, y, z, t=0.0: env(z, t) * V * np.sin(kx * x + mz * z - sig * t) + 0 * y   # noqa: E731
    w = lambda x, y, z, t=0.0: env(z, t) * W * np.cos(kx * x + mz * z - sig * t) + 0 * y   # noqa: E731
    b = lambda x, y, z, t=0.0: env(z, t) * Bb * np.sin(kx * x + mz * z - sig * t) + Nb ** 2 * z + 0 * (x + y)   # noqa: E731
    zspec = np.linspace(-Lx, 0, N + 1) if explicit_faces else (-Lx, 0)
    if flat_y:
        g = mod.RectilinearGrid(size=(N, N), topology=(P, F, B), x=(0, Lx), z=zspec, halo=(1, 1))
    else:
        g = mod.RectilinearGrid(size=(N, 1, N), topology=(P, P, B), x=(0, Lx), y=(0, Lx), z=zspec, halo=(1, 1, 1))
    m = mod.NonhydrostaticModel(g, advection=mod.CenteredSecondOrder(), closure=mod.ScalarDiffusivity(nu=nu, kappa=nu),
                                buoyancy=mod.BuoyancyTracer(), tracers=("b",), coriolis=mod.FPlane(f))
    mod.set_model(m, u=u, v=v, w=w, b=b)
    for _ in range(10):
        mod.time_step(m, dt)
    if mod is O:
        gr = m.grid
        X, Y, Z = gr.xnodes(m.u.loc[0]).reshape(-1, 1, 1), gr.ynodes(m.u.loc[1]).reshape(1, -1, 1), gr.znodes(m.u.loc[2]).reshape(1, 1, -1)
    else:
        X, Y, Z = m.nodes("u")
    ua = u(X, Y, Z, m.time)
    un = m.u.interior()
    rel = np.mean((un - ua) ** 2) / np.mean(ua ** 2)
    assert rel < 1e-4, rel


WAVE = [(False, False), (True, False), (False, True), (True, True)]
WAVE_IDS = ["periodic-y", "flat-y", "periodic-y-zfaces", "flat-y-zfaces"]


@pytest.mark.parametrize("flat_y,faces", WAVE, ids=WAVE_IDS)
def test_internal_wave_oracle(flat_y, faces):
    internal_wave(O, flat_y, faces)


@pytest.mark.parametrize("flat_y,faces", WAVE[:2], ids=WAVE_IDS[:2])
def test_internal_wave_library(ocn, backend, flat_y, faces):
    internal_wave(_lib(ocn, backend, False), flat_y, faces, N=128)


@pytest.mark.gpu
@pytest.mark.parametrize("flat_y,faces", WAVE, ids=WAVE_IDS)
def test_internal_wave_library_gpu(ocn, backend, flat_y, faces):
    internal_wave(_lib(ocn, backend, True), flat_y, faces)
