"""Literal values of the reference's grid tests (test/test_grids.jl:10-143, 336-388) on the oracle's grid -- the object every
spacing, node and halo index of the restated algorithm comes from -- and, where the C ABI exposes them (field shapes, halos),
on the library.  `==` where the reference uses `==`, isapprox where it uses isapprox."""
import numpy as np
import pytest

import oracle as O
from oracle.grid import Center, Face

P, B, F = "Periodic", "Bounded", "Flat"


def at(axis_nodes, H, idx):
    """element with the reference's 1-based index `idx` of an offset array whose first entry has index 1 - H"""
    return axis_nodes[idx - 1 + H]


def test_correct_size_extent_and_halo():                      # :10-60
    g = O.RectilinearGrid(size=(4, 6, 8), extent=(2 * np.pi, 4 * np.pi, 9 * np.pi), halo=(1, 2, 3))
    assert (g.Nx, g.Ny, g.Nz) == (4, 6, 8)
    assert (g.Lx, g.Ly, g.Lz) == (2 * np.pi, 4 * np.pi, 9 * np.pi)
    assert (g.Hx, g.Hy, g.Hz) == (1, 2, 3)
    g = O.RectilinearGrid(size=(4, 6, 8), x=(1, 2), y=(np.pi, 3 * np.pi), z=(0, 4))
    assert g.Lx == 1.0 and g.Ly == 2 * np.pi and g.Lz == 4.0


def test_halo_faces_first_cells_end_faces():                 # :62-110
    N, H, L = 4, 1, 2.0
    d = L / N
    g = O.RectilinearGrid(topology=(P, B, B), size=(N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H))
    xF, yF, zF = g.ax[0].F, g.ax[1].F, g.ax[2].F
    assert at(xF, H, 0) == -H * d and at(yF, H, 0) == -H * d and at(zF, H, 0) == -H * d
    assert at(xF, H, N + 1) == L                 # Periodic: no face beyond N + H
    assert at(yF, H, N + 2) == L + H * d and at(zF, H, N + 2) == L + H * d
    L4 = 4.0
    g = O.RectilinearGrid(size=(N, N, N), x=(0, L4), y=(0, L4), z=(0, L4), halo=(H, H, H))
    for a in g.ax:
        assert at(a.C, H, 1) == (L4 / N) / 2


def test_ranges_have_correct_length_and_no_roundoff():       # :112-141
    Nx, Ny, Nz, Hx, Hy, Hz = 8, 9, 10, 1, 2, 1
    g = O.RectilinearGrid(size=(Nx, Ny, Nz), extent=(1, 1, 1), halo=(Hx, Hy, Hz), topology=(B, B, B))
    assert (len(g.ax[0].C), len(g.ax[1].C), len(g.ax[2].C)) == (Nx + 2 * Hx, Ny + 2 * Hy, Nz + 2 * Hz)
    assert (len(g.ax[0].F), len(g.ax[1].F), len(g.ax[2].F)) == (Nx + 1 + 2 * Hx, Ny + 1 + 2 * Hy, Nz + 1 + 2 * Hz)
    g = O.RectilinearGrid(size=(1, 1, 64), extent=(1, 1, np.pi / 2), halo=(1, 1, 1))       # issue 480 of the reference
    assert len(g.ax[2].C) == 64 + 2 and len(g.ax[2].F) == 64 + 2 + 1


@pytest.mark.parametrize("Nz", [16, 17])
def test_constant_quadratic_and_tanh_spacings(Nz):           # :336-386
    H = 1
    g = O.RectilinearGrid(size=(1, 1, Nz), x=(0, 1), y=(0, 1), z=np.arange(0, Nz + 1, dtype=float), halo=(1, 1, H))
    k = np.arange(1, Nz + 1)
    assert (g.ax[2].d_center(k) == 1).all() and (g.ax[2].d_face(k) == 1).all()
    for zf in (lambda kk: (kk - 1.0) ** 2, lambda kk: np.tanh(3 * (2 * (kk - 1) / Nz - 1)) / np.tanh(3)):
        g = O.RectilinearGrid(size=(1, 1, Nz), x=(0, 1), y=(0, 1), z=zf, halo=(1, 1, H))
        a = g.ax[2]
        zc = lambda kk: (zf(kk) + zf(kk + 1.0)) / 2                                     # noqa: E731
        assert np.allclose([at(a.F, H, i) for i in range(1, Nz + 2)], [zf(float(i)) for i in range(1, Nz + 2)], rtol=1.5e-8, atol=0)
        assert np.allclose([at(a.C, H, i) for i in range(1, Nz + 1)], [zc(float(i)) for i in range(1, Nz + 1)], rtol=1.5e-8, atol=0)
        assert np.allclose(a.d_center(k), [zf(i + 1.0) - zf(float(i)) for i in k], rtol=1.5e-8, atol=0)
        k2 = np.arange(2, Nz + 1)       # the spacing at face 1 involves a halo point
        assert np.allclose(a.d_face(k2), [zc(float(i)) - zc(i - 1.0) for i in k2], rtol=1.5e-8, atol=0)


def _library_shapes(ocn):
    """total_size of the fields the library allocates == the reference's new_data sizes for the same grids"""
    for topo, N, H in [((P, B, B), (4, 4, 4), (1, 1, 1)), ((B, B, B), (8, 9, 10), (1, 2, 1)), ((P, P, P), (4, 6, 8), (1, 2, 3))]:
        g = ocn.RectilinearGrid(size=N, extent=(1, 1, 1), halo=H, topology=topo)
        og = O.RectilinearGrid(size=N, extent=(1, 1, 1), halo=H, topology=topo)
        for loc in [(Center, Center, Center), (Face, Center, Center), (Center, Face, Center), (Center, Center, Face)]:
            f = ocn.Field(tuple("Face" if l is Face else "Center" for l in loc), g)
            assert f.total == tuple(og.total_size(loc)) and f.halo == tuple(H)


def test_library_field_sizes(ocn, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run")
    _library_shapes(ocn)


@pytest.mark.gpu
def test_library_field_sizes_gpu(ocn):
    _library_shapes(ocn)
