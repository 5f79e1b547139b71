"""The C++/OpenMP restatement of the reference's CPU() path (oracle/cpu/ocn_cpu.cpp -- bench.py's `cpu_baseline`) against
the NumPy oracle: two independent restatements of the same reference files must agree to round-off on BASELINE config 2
in miniature (triply periodic, WENO5 Z weights, AB2 with its Euler first step, FFT Poisson)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "cpu", "libocn_cpu.so")


def load():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(LIB)
    PD = C.POINTER(C.c_double)
    L.ocncpu_run.restype = C.c_int
    L.ocncpu_run.argtypes = [C.c_int] * 3 + [C.c_double] * 3 + [PD] * 4 + [C.c_double, C.c_int, C.c_int, PD]
    return L


@pytest.mark.parametrize("N,threads", [((16, 8, 32), 1), ((16, 16, 16), 4)])
def test_cpp_restatement_matches_numpy_oracle(N, threads):
    L = load()
    rng = np.random.default_rng(1)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    ext = (1.0, 0.7, 1.3)
    og = O.RectilinearGrid(size=N, extent=ext, topology=("Periodic",) * 3)
    om = O.NonhydrostaticModel(og, advection=O.WENO5())
    O.set_model(om, **init)
    dt = 0.1 / max(N) / np.abs(om.u.data).max()
    nsteps = 3
    for _ in range(nsteps):
        O.time_step(om, dt)
    arrs = {n: np.asfortranarray(a.copy()) for n, a in init.items()}
    p = np.zeros(N, order="F")
    secs = C.c_double()
    PD = C.POINTER(C.c_double)
    rc = L.ocncpu_run(N[0], N[1], N[2], ext[0], ext[1], ext[2], arrs["u"].ctypes.data_as(PD), arrs["v"].ctypes.data_as(PD),
                      arrs["w"].ctypes.data_as(PD), p.ctypes.data_as(PD), dt, nsteps, threads, C.byref(secs))
    assert rc == 0
    for n, ref in (("u", om.u.interior()), ("v", om.v.interior()), ("w", om.w.interior())):
        assert np.abs(arrs[n] - ref).max() <= 1e-11 * np.abs(ref).max(), n
    assert np.abs(p - om.pNHS.interior()).max() <= 1e-10 * np.abs(om.pNHS.interior()).max()


def test_rejects_sizes_it_cannot_transform():
    L = load()
    z = np.zeros((12, 8, 8), order="F")
    PD = C.POINTER(C.c_double)
    assert L.ocncpu_run(12, 8, 8, 1.0, 1.0, 1.0, z.ctypes.data_as(PD), z.ctypes.data_as(PD), z.ctypes.data_as(PD), None, 1e-3, 1, 1, None) != 0
