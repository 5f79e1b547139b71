"""Staggered-grid operators in "offset-function" form (oracle; test infrastructure only).

The reference writes every operator as ``op(i, j, k, grid, f, args...)`` where ``f`` is an array or
another index function (``Operators/difference_operators.jl:7-27``,
``interpolation_operators.jl:20-40``, ``derivative_operators.jl:6-29``).  Here an *offset function*
is a callable ``F(o)`` with ``o = (di, dj, dk)`` that returns the (Nx, Ny, Nz) array of values at
``(i+di, j+dj, k+dk)`` for all kernel indices ``i=1..Nx, j=1..Ny, k=1..Nz`` -- the launch range of
every ``:xyz`` kernel (``Utils/kernel_launching.jl:52-66``).  A :class:`oracle.fields.Field` is
itself an offset function.  Operator composition is then literally the reference's.

Only x/y-regular grids are supported (z may be stretched), so horizontal spacings are numbers and
only the z location of a metric matters (``spacings_and_areas_and_volumes.jl:56-113``).
"""
import numpy as np

from .grid import Flat, Center, Face

E = ((1, 0, 0), (0, 1, 0), (0, 0, 1))


def sh(o, d, n=1):
    """o + n * e_d"""
    o = list(o)
    o[d] += n
    return tuple(o)


class Ops:
    def __init__(self, grid):
        self.g = grid
        self.flat = tuple(t == Flat for t in grid.topo)

    # ---- differences (difference_operators.jl:7-49) -------------------------------------------
    def dC(self, d, f):
        """delta^c along d: result at Center from a Face-located f:  f[i+1] - f[i]."""
        if self.flat[d]:
            return lambda o: 0.0
        return lambda o: f(sh(o, d, 1)) - f(o)

    def dF(self, d, f):
        """delta^f along d: result at Face from a Center-located f:  f[i] - f[i-1]."""
        if self.flat[d]:
            return lambda o: 0.0
        return lambda o: f(o) - f(sh(o, d, -1))

    # ---- interpolation (interpolation_operators.jl:20-40, 94-114) ----------------------------
    def iC(self, d, f):
        if self.flat[d]:
            return f
        return lambda o: 0.5 * (f(o) + f(sh(o, d, 1)))

    def iF(self, d, f):
        if self.flat[d]:
            return f
        return lambda o: 0.5 * (f(sh(o, d, -1)) + f(o))

    # ---- spacings / areas / volumes -------------------------------------------------------------
    def delta(self, d, lz_or_loc, o):
        """spacing along d at location ``loc`` (Center/Face along d) for index + o[d]."""
        g = self.g
        if d == 0:
            return g.dx
        if d == 1:
            return g.dy
        if g.topo[2] == Flat or g.z_regular:
            return g.ax[2].dc
        return g.dzc(o[2]) if lz_or_loc == Center else g.dzf(o[2])

    def dz(self, lz, o):
        return self.delta(2, lz, o)

    def Ax(self, lz, o):
        return self.g.dy * self.dz(lz, o)

    def Ay(self, lz, o):
        return self.g.dx * self.dz(lz, o)

    def Az(self):
        return self.g.dx * self.g.dy

    def V(self, lz, o):
        return self.Az() * self.dz(lz, o)

    # ---- derivatives (derivative_operators.jl:6-29): difference / spacing at the result location --
    def ddC(self, d, f, lz=Center):
        """partial^c along d (result Center along d).  ``lz``: z-location of the result (for d != 2)."""
        dl = self.dC(d, f)
        if d == 2:
            return lambda o: dl(o) / self.dz(Center, o)
        return lambda o: dl(o) / self.delta(d, Center, o)

    def ddF(self, d, f, lz=Center):
        dl = self.dF(d, f)
        if d == 2:
            return lambda o: dl(o) / self.dz(Face, o)
        return lambda o: dl(o) / self.delta(d, Face, o)

    # ---- index arrays for near-boundary conditionals ---------------------------------------------
    def index(self, d, o):
        n = self.g.N[d]
        shape = [1, 1, 1]
        shape[d] = n
        return (np.arange(1, n + 1) + o[d]).reshape(shape)

    # ---- composite operators ------------------------------------------------------------------------
    def div_ccc(self, u, v, w):
        """``Operators/divergence_operators.jl:16-19``."""
        Axu = lambda o: self.Ax(Center, o) * u(o)     # noqa: E731  Ax_q^{fcc}
        Ayv = lambda o: self.Ay(Center, o) * v(o)     # noqa: E731  Ay_q^{cfc}
        Azw = lambda o: self.Az() * w(o)              # noqa: E731  Az_q^{ccf}
        return lambda o: 1 / self.V(Center, o) * (self.dC(0, Axu)(o) + self.dC(1, Ayv)(o) + self.dC(2, Azw)(o))

    def laplacian_ccc(self, c):
        """``Operators/laplacian_operators.jl:36-40``: 1/V [dx(Ax dx c) + dy(Ay dy c) + dz(Az dz c)]."""
        fx = lambda o: self.Ax(Center, o) * self.ddF(0, c)(o)   # noqa: E731  at fcc
        fy = lambda o: self.Ay(Center, o) * self.ddF(1, c)(o)   # noqa: E731  at cfc
        fz = lambda o: self.Az() * self.ddF(2, c)(o)            # noqa: E731  at ccf
        return lambda o: 1 / self.V(Center, o) * (self.dC(0, fx)(o) + self.dC(1, fy)(o) + self.dC(2, fz)(o))


Z3 = (0, 0, 0)
__all__ = ["Ops", "sh", "Z3", "Center", "Face"]
