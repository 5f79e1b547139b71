"""RectilinearGrid restatement (oracle; test infrastructure only).

Follows ``Grids/rectilinear_grid.jl:249-279`` (constructor), ``Grids/grid_generation.jl``
(``generate_coordinate``: regular axes ``:77-107``, stretched axes ``:28-75``, Flat ``:110-112``),
``Grids/grid_utils.jl:105-128`` (total lengths) and ``Grids/new_data.jl:16-61`` (parent array layout).

Index conventions: the reference indexes fields 1-based with halos at ``1-H..0`` and ``N+1..N+H``.
Here a *parent* array is 0-based: reference index ``i`` lives at parent index ``i - 1 + H``.
Parent arrays are Fortran-ordered (x fastest), exactly the memory layout of the reference's
``OffsetArray`` parents, so they can be handed to the C ABI unchanged.
"""
from fractions import Fraction

import numpy as np

Periodic, Bounded, Flat = "Periodic", "Bounded", "Flat"
Center, Face = "Center", "Face"


def total_length(loc, topo, N, H):
    """``Grids/grid_utils.jl:105-110``."""
    if topo == Flat:
        return N
    if loc == Face and topo == Bounded:
        return N + 1 + 2 * H
    return N + 2 * H


class Axis:
    """One coordinate axis: spacings at centres/faces as functions of the (1-based) index."""

    def __init__(self, topo, N, H, coord):
        self.topo, self.N, self.H = topo, int(N), int(H)
        if topo == Flat:
            # grid_generation.jl:110-112 -> L = 1, spacings = 1
            self.N, self.H = 1 if N is None else int(N), 0
            self.L, self.regular = 1.0, True
            self.dc = self.df = 1.0
            self.F = np.ones(self.N)
            self.C = np.ones(self.N)
            self.offset = 0
            return
        coord = tuple(coord) if not callable(coord) else coord
        if not callable(coord) and len(coord) == 2 and not isinstance(coord, np.ndarray):
            # regular axis, grid_generation.jl:77-107.  BigFloat arithmetic then one rounding to FT.
            c1, c2 = Fraction(float(coord[0])), Fraction(float(coord[1]))
            assert c1 < c2
            L = c2 - c1
            d = L / self.N
            self.L = float(L)
            self.regular = True
            self.dc = self.df = float(d)
            TF = total_length(Face, topo, self.N, self.H)
            TC = total_length(Center, topo, self.N, self.H)
            Fm = c1 - self.H * d
            # range(FT(F-), FT(F+), length=TF): evaluate nodes (not used in the time step itself)
            ext = L + (2 * self.H if topo == Bounded else 2 * self.H - 1) * d
            Fp = Fm + ext
            Cm = Fm + d / 2
            Cp = Cm + L + d * (2 * self.H - 1)
            self.F = np.linspace(float(Fm), float(Fp), TF)
            self.C = np.linspace(float(Cm), float(Cp), TC)
            self.offset = self.H
            return
        # stretched axis, grid_generation.jl:28-75
        N, H = self.N, self.H
        if callable(coord):
            interiorF = np.array([float(coord(i)) for i in range(1, N + 2)], dtype=np.float64)
        else:
            interiorF = np.array(coord, dtype=np.float64)
            assert interiorF.size == N + 1, "stretched axis needs N+1 faces"
        self.L = float(interiorF[N] - interiorF[0])
        if topo == Bounded:
            dm = np.full(H, interiorF[1] - interiorF[0])
            dp = np.full(H, interiorF[-1] - interiorF[-2])
        else:
            # lower_exterior: Fi[end-H+i] - Fi[end-H+i-1], i=1:H ; upper: Fi[i+1]-Fi[i]
            dm = np.array([interiorF[N - H + i] - interiorF[N - H + i - 1] for i in range(1, H + 1)])
            dp = np.array([interiorF[i] - interiorF[i - 1] for i in range(1, H + 1)])
        dp = dp[::-1]  # reverse(upper_exterior...)
        c1, cN1 = interiorF[0], interiorF[N]
        Fm = np.array([c1 - np.sum(dm[i:H]) for i in range(H)])
        Fp = np.array([cN1 + np.sum(dp[i:H]) for i in range(H)])[::-1]
        F = np.concatenate([Fm, interiorF, Fp])
        TC = total_length(Center, topo, N, H)
        C = np.array([(F[i + 1] + F[i]) / 2 for i in range(TC)])
        dF = np.array([C[i] - C[i - 1] for i in range(1, TC)])
        TF = total_length(Face, topo, N, H)
        F = F[:TF]
        dC = np.array([F[i + 1] - F[i] for i in range(TF - 1)])
        dF = np.concatenate([[dF[0]], dF, [dF[-1]]])
        for i in range(len(dF) - 1, 0, -1):
            dF[i] = dF[i - 1]
        self.regular = False
        # reference: dC = OffsetArray(dC, -H)  -> index k at parent k-1+H
        #            dF = OffsetArray(dF, -H-1)-> index k at parent k+H
        self._dc, self._df = dC, dF
        self.F, self.C = F, C
        self.offset = H

    # spacing at cell centres / faces for reference index arrays (1-based, may reach into halos)
    def d_center(self, idx):
        if self.regular:
            return self.dc
        return self._dc[np.asarray(idx) - 1 + self.H]

    def d_face(self, idx):
        if self.regular:
            return self.df
        return self._df[np.asarray(idx) + self.H]


class RectilinearGrid:
    """``RectilinearGrid(size=, extent=/x,y,z=, halo=, topology=)``; x and y must be regular or Flat."""

    def __init__(self, size, extent=None, x=None, y=None, z=None, halo=None,
                 topology=(Periodic, Periodic, Bounded)):
        topo = tuple(topology)
        nonflat = [t != Flat for t in topo]
        size = tuple(np.atleast_1d(size).tolist())
        assert len(size) == sum(nonflat), "size must have one entry per non-Flat dimension"
        if halo is None:
            halo = tuple(3 for _ in size)  # input_validation.jl:55-58
        halo = tuple(np.atleast_1d(halo).tolist())
        if extent is not None:
            extent = tuple(np.atleast_1d(extent).tolist())
        N, H, coords = [], [], []
        it = 0
        given = [x, y, z]
        for d in range(3):
            if not nonflat[d]:
                N.append(1); H.append(0); coords.append(None)
                continue
            N.append(size[it]); H.append(halo[it])
            if extent is not None:
                # validate_rectilinear_domain: x,y = (0, L); z = (-L, 0)
                coords.append((0.0, extent[it]) if d < 2 else (-extent[it], 0.0))
            else:
                coords.append(given[d])
            it += 1
        self.topo = topo
        self.ax = [Axis(topo[d], N[d], H[d], coords[d]) for d in range(3)]
        self.Nx, self.Ny, self.Nz = (a.N for a in self.ax)
        self.Hx, self.Hy, self.Hz = (a.H for a in self.ax)
        self.Lx, self.Ly, self.Lz = (a.L for a in self.ax)
        assert self.ax[0].regular and self.ax[1].regular, "x and y must be regular"
        self.dx, self.dy = self.ax[0].dc, self.ax[1].dc
        self.z_regular = self.ax[2].regular

    @property
    def N(self):
        return (self.Nx, self.Ny, self.Nz)

    @property
    def H(self):
        return (self.Hx, self.Hy, self.Hz)

    def total_size(self, loc):
        return tuple(total_length(loc[d], self.topo[d], self.N[d], self.H[d]) for d in range(3))

    def with_halo(self, halo):
        """``rectilinear_grid.jl:382-403``: same grid, new halo."""
        g = object.__new__(RectilinearGrid)
        g.__dict__.update(self.__dict__)
        newax = []
        for d in range(3):
            a = self.ax[d]
            if a.topo == Flat:
                newax.append(a)
            elif a.regular:
                f0 = a.F[a.H]
                newax.append(Axis(a.topo, a.N, halo[d], (f0, f0 + a.L)))
                newax[-1].dc = newax[-1].df = a.dc
                newax[-1].L = a.L
            else:
                newax.append(Axis(a.topo, a.N, halo[d], a.F[a.H:a.H + a.N + 1]))
        g.ax = newax
        g.Hx, g.Hy, g.Hz = (a.H for a in newax)
        return g

    # ---- metric helpers (arrays broadcastable against (Nx,Ny,Nz) work arrays) --------------
    def dzc(self, dk=0):
        """Delta z at centres k+dk, k=1..Nz  (``Operators/spacings_and_areas_and_volumes.jl:63-67``)."""
        a = self.ax[2]
        if a.regular:
            return a.dc
        return a.d_center(np.arange(1, self.Nz + 1) + dk).reshape(1, 1, -1)

    def dzf(self, dk=0):
        a = self.ax[2]
        if a.regular:
            return a.df
        return a.d_face(np.arange(1, self.Nz + 1) + dk).reshape(1, 1, -1)

    def znodes(self, loc):
        a = self.ax[2]
        if loc == Center:
            return a.C[a.H:a.H + a.N].copy()
        n = a.N + 1 if a.topo == Bounded else a.N
        return a.F[a.H:a.H + n].copy()

    def xnodes(self, loc):
        a = self.ax[0]
        if loc == Center:
            return a.C[a.H:a.H + a.N].copy()
        n = a.N + 1 if a.topo == Bounded else a.N
        return a.F[a.H:a.H + n].copy()

    def ynodes(self, loc):
        a = self.ax[1]
        if loc == Center:
            return a.C[a.H:a.H + a.N].copy()
        n = a.N + 1 if a.topo == Bounded else a.N
        return a.F[a.H:a.H + n].copy()
