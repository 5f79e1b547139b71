"""Pressure Poisson solvers (oracle; test infrastructure only).

Restates ``Solvers/poisson_eigenvalues.jl:8-31``, ``Solvers/fft_based_poisson_solver.jl:50-125``,
``Solvers/plan_transforms.jl:21-39`` + ``discrete_transforms.jl:25-33`` (FFTW conventions:
unnormalised forward FFT / REDFT10, ``ifft`` normalised by 1/N, REDFT01 normalised by 1/2N),
``Solvers/fourier_tridiagonal_poisson_solver.jl:16-101``, ``Solvers/batched_tridiagonal_solver.jl:91-122``
and ``Solvers/index_permutations.jl:10-32``.

Third-party arithmetic: the reference calls FFTW (FFTW.jl 1.5.0 / FFTW_jll 3.3.10, not in
/root/reference); here ``scipy.fft`` (pocketfft) provides the same transforms:
``REDFT10 == 2 * dct-II (norm=None)`` i.e. ``scipy.fft.dct(type=2)``; ``REDFT01 == scipy.fft.dct(type=3)``.
"""
import numpy as np
import scipy.fft as sfft

from .grid import Periodic, Bounded, Flat


def poisson_eigenvalues(N, L, topo):
    """poisson_eigenvalues.jl:8-31 (1-D vector; ``inds - 1`` = 0..N-1)."""
    i = np.arange(N, dtype=np.float64)
    if topo == Periodic:
        return (2 * np.sin(i * np.pi / N) / (L / N)) ** 2
    if topo == Bounded:
        return (2 * np.sin(i * np.pi / (2 * N)) / (L / N)) ** 2
    return np.zeros(N)


def _forward(a, topo, dims=(0, 1, 2)):
    """forward transforms: Bounded first, then Periodic (plan_transforms.jl:134-137)."""
    for d in dims:
        if topo[d] == Bounded:
            # FFTW REDFT10 on a complex array acts on real and imaginary parts separately
            a = sfft.dct(a.real, type=2, axis=d) + 1j * sfft.dct(a.imag, type=2, axis=d)
    per = [d for d in dims if topo[d] == Periodic]
    if per:
        a = sfft.fftn(a, axes=per)
    return a


def _backward(a, topo, N, dims=(0, 1, 2)):
    per = [d for d in dims if topo[d] == Periodic]
    if per:
        a = sfft.ifftn(a, axes=per)
    for d in dims:
        if topo[d] == Bounded:
            a = (sfft.dct(a.real, type=3, axis=d) + 1j * sfft.dct(a.imag, type=3, axis=d)) / (2 * N[d])
    return a


class FFTBasedPoissonSolver:
    """fft_based_poisson_solver.jl:50-72."""

    def __init__(self, grid):
        assert grid.z_regular or grid.topo[2] == Flat
        self.grid = grid
        g = grid
        self.lx = poisson_eigenvalues(g.Nx, g.Lx, g.topo[0]).reshape(-1, 1, 1)
        self.ly = poisson_eigenvalues(g.Ny, g.Ly, g.topo[1]).reshape(1, -1, 1)
        self.lz = poisson_eigenvalues(g.Nz, g.Lz, g.topo[2]).reshape(1, 1, -1)

    def solve(self, rhs, m=0):
        """solve!: returns the real (Nx,Ny,Nz) solution of (lap + m) phi = rhs  (:93-120)."""
        g = self.grid
        b = _forward(np.asarray(rhs, dtype=np.complex128), g.topo)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = -b / (self.lx + self.ly + self.lz - m)
        if m == 0:
            phi[0, 0, 0] = 0
        phi = _backward(phi, g.topo, g.N)
        return np.ascontiguousarray(phi.real)


def thomas_batched(a, b, c, f, Nz):
    """batched_tridiagonal_solver.jl:91-122.  a, c: 1-D (Nz-1); b: (Nx,Ny,Nz); f: complex (Nx,Ny,Nz).
    Includes the reference's early ``break`` when |beta| <= 10 eps (per column)."""
    phi = np.zeros_like(f)
    t = np.zeros(b.shape, dtype=np.float64)
    beta = b[:, :, 0].copy()
    phi[:, :, 0] = f[:, :, 0] / beta
    alive = np.ones(beta.shape, dtype=bool)
    for k in range(1, Nz):
        with np.errstate(divide="ignore", invalid="ignore"):
            tk = c[k - 1] / beta
        t[:, :, k] = np.where(alive, tk, t[:, :, k])
        beta_new = b[:, :, k] - a[k - 1] * t[:, :, k]
        beta = np.where(alive, beta_new, beta)
        dd = np.abs(beta) > 10 * np.finfo(np.float64).eps
        alive = alive & dd
        with np.errstate(divide="ignore", invalid="ignore"):
            val = (f[:, :, k] - a[k - 1] * phi[:, :, k - 1]) / beta
        phi[:, :, k] = np.where(alive, val, phi[:, :, k])
    for k in range(Nz - 2, -1, -1):
        phi[:, :, k] -= t[:, :, k + 1] * phi[:, :, k + 1]
    return phi


class FourierTridiagonalPoissonSolver:
    """fourier_tridiagonal_poisson_solver.jl:30-101 (x, y Periodic or Bounded-regular; z Bounded)."""

    def __init__(self, grid):
        g = grid
        assert g.topo[2] == Bounded
        self.grid = g
        lx = poisson_eigenvalues(g.Nx, g.Lx, g.topo[0]).reshape(-1, 1)
        ly = poisson_eigenvalues(g.Ny, g.Ly, g.topo[1]).reshape(1, -1)
        Nz = g.Nz
        az = g.ax[2]
        dzf = np.array([az.d_face(k) if not az.regular else az.df for k in range(1, Nz + 2)])   # k = 1..Nz+1
        dzc = np.array([az.d_center(k) if not az.regular else az.dc for k in range(1, Nz + 1)])
        self.dzc = dzc
        self.lower = 1.0 / dzf[1:Nz]            # [1/dzf(k) for k in 2:Nz]
        D = np.zeros((g.Nx, g.Ny, Nz))
        lam = lx + ly
        D[:, :, 0] = -1 / dzf[1] - dzc[0] * lam
        for k in range(2, Nz):                   # k = 2..Nz-1 (1-based)
            D[:, :, k - 1] = -(1 / dzf[k] + 1 / dzf[k - 1]) - dzc[k - 1] * lam
        D[:, :, Nz - 1] = -1 / dzf[Nz - 1] - dzc[Nz - 1] * lam
        self.D = D

    def solve(self, rhs_times_dz):
        """``solve!(x, solver)``: the source term already multiplied by dz_c (solve_for_pressure.jl:30-33)."""
        g = self.grid
        b = _forward(np.asarray(rhs_times_dz, dtype=np.complex128), g.topo, dims=(0, 1))
        phi = thomas_batched(self.lower, self.D, self.lower, b, g.Nz)
        phi = _backward(phi, g.topo, g.N, dims=(0, 1))
        phi = phi.real
        phi = phi - np.mean(phi)
        return np.ascontiguousarray(phi)

    def solve_source(self, rhs):
        """``solve!(x, solver, b)`` -> set_source_term! multiplies by dz_c (:109-123)."""
        return self.solve(np.asarray(rhs) * self.dzc.reshape(1, 1, -1))


def permute_index(i, N):
    """index_permutations.jl:18-20 (1-based): [1..8] -> [1,8,2,7,3,6,4,5]."""
    if i % 2 == 1:
        return i // 2 + 1
    return N - (i - 1) // 2


def unpermute_index(i, N):
    """index_permutations.jl:36 (1-based): [1..8] -> [1,3,5,7,8,6,4,2]."""
    if i <= (N + 1) // 2:
        return 2 * i - 1
    return 2 * (N - i + 1)
