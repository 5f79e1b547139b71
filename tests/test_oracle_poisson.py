"""Pins the oracle's Poisson solvers with the reference's own property tests
(test/test_poisson_solvers.jl:45-93, test/dependencies_for_poisson_solvers.jl:13-148,
test/test_poisson_solvers_vertically_stretched_grid.jl:12-43, test/test_batched_tridiagonal_solver.jl,
src/Solvers/index_permutations.jl:10-32)."""
import itertools

import numpy as np
import pytest

import oracle as O
from oracle.fields import Field, default_bcs, fill_halo_regions
from oracle.operators import Ops
from oracle import poisson

P, B, F = O.Periodic, O.Bounded, O.Flat
TOPOS = list(itertools.product((P, B), repeat=3))
Z3 = (0, 0, 0)


def random_divergent_source(grid, rng):
    """dependencies_for_poisson_solvers.jl:13-42: *Center* fields carrying u/v/w boundary conditions."""
    C3 = (O.Center,) * 3
    locs = ((O.Face, O.Center, O.Center), (O.Center, O.Face, O.Center), (O.Center, O.Center, O.Face))
    U = []
    for loc in locs:
        f = Field(grid, C3, default_bcs(grid, loc))
        f.set(rng.random(grid.N))
        fill_halo_regions(f)
        U.append(f)
    ops = Ops(grid)
    return ops.div_ccc(*U)(Z3), U


def laplacian_of(grid, phi):
    f = Field(grid, (O.Center,) * 3)
    f.set(phi)
    fill_halo_regions(f)
    return Ops(grid).laplacian_ccc(f)(Z3)


def grids_for(topo, N):
    yield O.RectilinearGrid(size=(N, N, N), extent=(1, 1, 1), topology=topo)
    yield O.RectilinearGrid(size=(1, N, N), extent=(1, 1, 1), topology=topo)
    yield O.RectilinearGrid(size=(N, 1, N), extent=(1, 1, 1), topology=topo)
    yield O.RectilinearGrid(size=(N, N, 1), extent=(1, 1, 1), topology=topo)


@pytest.mark.parametrize("topo", TOPOS)
@pytest.mark.parametrize("N", [7, 16])
def test_divergence_free_solution_all_topologies(topo, N):
    rng = np.random.default_rng(7 * N)
    for grid in grids_for(topo, N):
        R, _ = random_divergent_source(grid, rng)
        phi = poisson.FFTBasedPoissonSolver(grid).solve(R / 1.0)
        assert np.allclose(laplacian_of(grid, phi), R, rtol=1.5e-8, atol=1e-10)


@pytest.mark.parametrize("topo2", [(P, P, F), (P, B, F), (B, B, F), (F, P, P), (P, F, B)])
def test_divergence_free_solution_flat(topo2):
    rng = np.random.default_rng(3)
    grid = O.RectilinearGrid(size=(16, 11), extent=(1, 1), topology=topo2)
    R, _ = random_divergent_source(grid, rng)
    phi = poisson.FFTBasedPoissonSolver(grid).solve(R)
    assert np.allclose(laplacian_of(grid, phi), R, rtol=1.5e-8, atol=1e-10)


@pytest.mark.parametrize("topo", TOPOS)
def test_rectangular_prime_and_even(topo):
    rng = np.random.default_rng(11)
    for Nx, Ny, Nz in itertools.product((11, 16), repeat=3):
        grid = O.RectilinearGrid(size=(Nx, Ny, Nz), extent=(1, 1, 1), topology=topo)
        R, _ = random_divergent_source(grid, rng)
        phi = poisson.FFTBasedPoissonSolver(grid).solve(R)
        assert np.allclose(laplacian_of(grid, phi), R, rtol=1.5e-8, atol=1e-10)


def _analytic_error(N, topo, mode):
    grid = O.RectilinearGrid(size=(N, N, N), x=(0, 2 * np.pi), y=(0, 2 * np.pi), z=(0, 2 * np.pi), topology=topo)
    x = grid.xnodes(O.Center).reshape(-1, 1, 1)
    y = grid.ynodes(O.Center).reshape(1, -1, 1)
    z = grid.znodes(O.Center).reshape(1, 1, -1)

    def psi(t, s):
        return np.cos(mode * s / 2) if t == B else np.cos(mode * s)

    def k2(t):
        return (mode / 2) ** 2 if t == B else mode ** 2
    Psi = psi(topo[0], x) * psi(topo[1], y) * psi(topo[2], z)
    f = -(k2(topo[0]) + k2(topo[1]) + k2(topo[2])) * Psi
    phi = poisson.FFTBasedPoissonSolver(grid).solve(f)
    return np.mean(np.abs(phi - Psi))


@pytest.mark.parametrize("topo", TOPOS)
@pytest.mark.parametrize("N1,N2,mode", [(64, 128, 1), (67, 131, 2)])
def test_second_order_convergence(topo, N1, N2, mode):
    e1, e2 = _analytic_error(N1, topo, mode), _analytic_error(N2, topo, mode)
    rate = np.log(e1 / e2) / np.log(N2 / N1)
    assert abs(rate - 2) <= 5e-3 * 2


@pytest.mark.parametrize("faces", [[1, 2, 4, 7, 11, 16, 22, 29, 37], [1, 2, 4, 7, 11, 16, 22, 29, 37, 51]])
@pytest.mark.parametrize("Nxy", [(8, 8), (7, 8), (8, 7)])
def test_vertically_stretched_poisson(faces, Nxy):
    """test_poisson_solvers_vertically_stretched_grid.jl:12-43."""
    rng = np.random.default_rng(5)
    Nz = len(faces) - 1
    for topo in ((P, P, B), (P, B, B), (B, P, B), (B, B, B)):
        grid = O.RectilinearGrid(size=(Nxy[0], Nxy[1], Nz), x=(0, 1), y=(0, 1), z=np.array(faces, float), topology=topo)
        R, _ = random_divergent_source(grid, rng)
        phi = poisson.FourierTridiagonalPoissonSolver(grid).solve_source(R)
        assert np.allclose(laplacian_of(grid, phi), R, rtol=1.5e-8, atol=1e-9)
        assert abs(phi.mean()) < 1e-12


@pytest.mark.parametrize("N", [8, 11, 18])
def test_thomas_against_dense(N):
    """test_batched_tridiagonal_solver.jl:6-152."""
    rng = np.random.default_rng(N)
    Nx, Ny = 3, 4
    a, c = rng.random(N - 1), rng.random(N - 1)
    b = 3 + rng.random((Nx, Ny, N))
    f = rng.random((Nx, Ny, N)) + 1j * rng.random((Nx, Ny, N))
    phi = poisson.thomas_batched(a, b, c, f, N)
    for i in range(Nx):
        for j in range(Ny):
            M = np.diag(b[i, j]) + np.diag(a, -1) + np.diag(c, 1)
            assert np.allclose(phi[i, j], np.linalg.solve(M, f[i, j]), rtol=1e-12)


def test_dct_index_permutations():
    """index_permutations.jl:10-14,28-32 doc examples."""
    assert [poisson.permute_index(i, 8) for i in range(1, 9)] == [1, 8, 2, 7, 3, 6, 4, 5]
    assert [poisson.permute_index(i, 9) for i in range(1, 10)] == [1, 9, 2, 8, 3, 7, 4, 6, 5]
    assert [poisson.unpermute_index(i, 8) for i in range(1, 9)] == [1, 3, 5, 7, 8, 6, 4, 2]
    assert [poisson.unpermute_index(i, 9) for i in range(1, 10)] == [1, 3, 5, 7, 9, 8, 6, 4, 2]
