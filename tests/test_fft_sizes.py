"""The custom transform passes of the triply periodic Poisson solve at their three sizes (128, 256, 512 points; csrc/zfft.hip:
four-step 16 x 8, 16 x 16, and a radix-2 decimation step in front of the 256-point network) -- x pass fused with the
right-hand side, y passes, fused z stage -- through whole time steps against the oracle (numpy.fft), which is how the
reference's own solver tests judge a transform (test_poisson_solvers.jl:45-93: residuals and convergence, no stored spectra)."""
import numpy as np
import pytest

import oracle as O

P = "Periodic"


def _steps_match_oracle(ocn, N, steps=2, tol=2e-11, path="all-in-one"):
    rng = np.random.default_rng(5)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    kw = dict(size=N, extent=tuple(x / N[0] for x in N), topology=(P,) * 3)
    m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), advection=ocn.WENO5())
    om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5())
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    assert path in m.kernel_path, m.kernel_path
    dt = 0.2 / N[0] / np.abs(om.u.data).max()
    for _ in range(steps):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
    for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS)):
        assert np.abs(a.parent() - b.data).max() <= tol * np.abs(b.data).max()
    assert m.max_abs_divergence() <= 1e-9


@pytest.mark.parametrize("N", [(16, 16, 128), (128, 128, 6), (8, 8, 512)], ids=lambda n: "x".join(map(str, n)))
def test_transform_sizes_hostemu(ocn, backend, N):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _steps_match_oracle(ocn, N, steps=1)


# 320 and 384 levels: the mixed-radix fused z stage (16 x 20 with five-point butterflies, 16 x 24 with three-point ones) behind the
# library's per-plane x / y transforms (round 3)
@pytest.mark.parametrize("N", [(8, 8, 320), (8, 6, 384)], ids=lambda n: "x".join(map(str, n)))
def test_mixed_radix_z_stage_hostemu(ocn, backend, N):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _steps_match_oracle(ocn, N, steps=1, path="")


@pytest.mark.gpu
@pytest.mark.parametrize("N", [(64, 48, 320), (40, 64, 384), (320, 320, 320)], ids=lambda n: "x".join(map(str, n)))
def test_mixed_radix_z_stage_gpu(ocn, N):
    _steps_match_oracle(ocn, N, steps=1 if N[0] == 320 else 2, path="")


GPU_SHAPES = [(128, 128, 128), (128, 256, 12), (256, 128, 512), (512, 512, 8), (512, 128, 16), (128, 512, 12), (16, 12, 512),
              (24, 16, 128), (192, 128, 16)]


@pytest.mark.gpu
@pytest.mark.parametrize("N", GPU_SHAPES, ids=lambda n: "x".join(map(str, n)))
def test_transform_sizes_gpu(ocn, N):
    _steps_match_oracle(ocn, N)


# ---- the same passes around the batched Thomas sweeps of the Fourier-tridiagonal solver (Bounded z; round 3) ------------------
def _bounded_steps_match_oracle(ocn, N, stretched, steps=2, tol=2e-11, stepper="RungeKutta3"):
    """(Periodic, Periodic, Bounded) with 128- / 256- / 512-point x and y: the right-hand side (times dz) is formed inside the x
    pass from the predictor, custom y passes, tridiagonal sweeps, custom inverse y, library inverse x -- whole RK3 steps with a
    buoyant tracer against the oracle (the BASELINE config 3 solver at its own row width)."""
    rng = np.random.default_rng(6)
    Nz = N[2]
    kw = dict(size=N, topology=(P, P, "Bounded"))
    if stretched:
        zf = -np.cumsum(np.concatenate([[0.0], np.linspace(1.0, 2.0, Nz)]))[::-1] / Nz
        kw.update(x=(0, 4.0), y=(0, 4.0), z=zf)
    else:
        kw.update(extent=(4.0, 4.0, 1.0))
    mk = lambda mod: mod.NonhydrostaticModel(mod.RectilinearGrid(**kw), advection=mod.WENO5(), timestepper=stepper,   # noqa: E731
                                             tracers=("b",), buoyancy=mod.BuoyancyTracer(), closure=mod.ScalarDiffusivity(nu=1e-3, kappa=1e-3))
    m, om = mk(ocn), mk(O)
    init = {n: 0.1 * (rng.random(getattr(om, n).interior().shape) - 0.5) for n in "uvw"}
    init["w"][:, :, 0] = 0
    init["w"][:, :, -1] = 0
    init["b"] = rng.random(N)
    ocn.set_model(m, **init)
    O.set_model(om, **init)
    for _ in range(steps):
        ocn.time_step(m, 2e-3)
        O.time_step(om, 2e-3)
    for a, b in ((m.u, om.u), (m.v, om.v), (m.w, om.w), (m.pNHS, om.pNHS), (m.tracers["b"], om.tracers["b"])):
        assert np.abs(a.parent() - b.data).max() <= tol * np.abs(b.data).max()
    assert m.max_abs_divergence() <= 1e-9


def test_bounded_z_custom_passes_hostemu(ocn, backend):
    """one AB2 step on a stretched grid (the emulation runs one OS thread per lane: `-m gpu` covers RK3, regular z and the sizes)"""
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _bounded_steps_match_oracle(ocn, (128, 128, 6), True, steps=1, stepper="QuasiAdamsBashforth2")


@pytest.mark.gpu
@pytest.mark.parametrize("N,stretched", [((128, 128, 16), True), ((256, 256, 12), True), ((256, 128, 10), False), ((512, 256, 8), True)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_bounded_z_custom_passes_gpu(ocn, N, stretched):
    _bounded_steps_match_oracle(ocn, N, stretched)
