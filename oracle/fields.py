"""Fields, boundary conditions and halo fills (oracle; test infrastructure only).

Restates ``Fields/field.jl:16-30`` (a field = parent array with halos + location + BCs),
``BoundaryConditions/field_boundary_conditions.jl:13-35`` (defaults),
``BoundaryConditions/fill_halo_regions.jl:34-102`` (three passes, non-periodic first),
``fill_halo_regions_periodic.jl:37-65``, ``fill_halo_regions_flux.jl:16-35``,
``fill_halo_regions_value_gradient.jl:7-99``, ``fill_halo_regions_open.jl:34-39``.
"""
import numpy as np

from .grid import Periodic, Bounded, Flat, Center, Face


class BC:
    """kind in {'periodic','flux','value','gradient','open',None}; condition = number, 2-D array or None."""

    def __init__(self, kind, condition=None):
        self.kind, self.condition = kind, condition

    def get(self):
        return 0.0 if self.condition is None else self.condition

    def __repr__(self):
        return f"BC({self.kind}, {self.condition})"


def FluxBC(v):
    return BC("flux", v)


def ValueBC(v):
    return BC("value", v)


def GradientBC(v):
    return BC("gradient", v)


PeriodicBC = lambda: BC("periodic")          # noqa: E731
NoFluxBC = lambda: BC("flux", None)          # noqa: E731
ImpenetrableBC = lambda: BC("open", None)    # noqa: E731

SIDES = (("west", "east"), ("south", "north"), ("bottom", "top"))


def default_bcs(grid, loc, auxiliary=False):
    """``field_boundary_conditions.jl:13-35`` (prognostic) / ``:30-35`` (auxiliary)."""
    bcs = {}
    for d in range(3):
        topo, l = grid.topo[d], loc[d]
        if topo == Periodic:
            bc = (PeriodicBC(), PeriodicBC())
        elif topo == Flat:
            bc = (None, None)
        elif l == Center:
            bc = (NoFluxBC(), NoFluxBC())
        else:  # Bounded + Face
            bc = (None, None) if auxiliary else (ImpenetrableBC(), ImpenetrableBC())
        bcs[SIDES[d][0]], bcs[SIDES[d][1]] = bc
    return bcs


class Field:
    def __init__(self, grid, loc, bcs=None, auxiliary=False):
        self.grid, self.loc = grid, tuple(loc)
        self.data = np.zeros(grid.total_size(self.loc), dtype=np.float64, order="F")
        self.bcs = default_bcs(grid, self.loc, auxiliary)
        if bcs:
            self.bcs.update(bcs)

    # number of interior nodes per dim (``grid_utils.jl:66-70``)
    def size(self):
        g = self.grid
        return tuple(g.N[d] + 1 if (self.loc[d] == Face and g.topo[d] == Bounded) else g.N[d]
                     for d in range(3))

    def interior(self):
        g = self.grid
        s = self.size()
        return self.data[g.Hx:g.Hx + s[0], g.Hy:g.Hy + s[1], g.Hz:g.Hz + s[2]]

    def set(self, value):
        """``Fields/set!.jl``: sets the interior; halos untouched."""
        it = self.interior()
        if callable(value):
            g = self.grid
            X = g.xnodes(self.loc[0]).reshape(-1, 1, 1)
            Y = g.ynodes(self.loc[1]).reshape(1, -1, 1)
            Z = g.znodes(self.loc[2]).reshape(1, 1, -1)
            it[...] = value(X, Y, Z) + 0 * (X + Y + Z)
        else:
            it[...] = value

    def __call__(self, o=(0, 0, 0)):
        """Work-array view: values at (i+o0, j+o1, k+o2) for i=1..Nx, j=1..Ny, k=1..Nz."""
        g = self.grid
        sl = []
        for d in range(3):
            if g.topo[d] == Flat:
                sl.append(slice(0, g.N[d]))
            else:
                a = g.H[d] + o[d]
                assert a >= 0 and a + g.N[d] <= self.data.shape[d], "stencil reaches outside the halo"
                sl.append(slice(a, a + g.N[d]))
        return self.data[tuple(sl)]

    def copy(self):
        f = Field(self.grid, self.loc, self.bcs)
        f.data[...] = self.data
        return f


def _axis_slice(arr, d, sl):
    idx = [slice(None)] * 3
    idx[d] = sl
    return arr[tuple(idx)]


def _interior_other_dims(field, d):
    """index tuple restricting the two other dims to 1..N (the ``:xy``-style launch range)."""
    g = field.grid
    idx = []
    for e in range(3):
        if e == d:
            idx.append(None)
        elif g.topo[e] == Flat:
            idx.append(slice(None))
        else:
            idx.append(slice(g.H[e], g.H[e] + g.N[e]))  # work_layout uses the *centre* size
    return idx


def _bc_array(cond, field, d):
    """broadcast a BC condition (number or 2-D array over the two other dims) to a plane."""
    if np.isscalar(cond):
        return cond
    c = np.asarray(cond, dtype=np.float64)
    shp = [field.grid.N[e] for e in range(3) if e != d]
    c = c.reshape(shp)
    return np.expand_dims(c, d)


def _fill_one_direction(field, d):
    g = field.grid
    topo = g.topo[d]
    if topo == Flat:
        return
    N, H = g.N[d], g.H[d]
    p = field.data
    left, right = field.bcs[SIDES[d][0]], field.bcs[SIDES[d][1]]
    if left is not None and left.kind == "periodic":
        # fill_halo_regions_periodic.jl:37-65 -- over the *full parent extent* of the other dims
        for i in range(H):
            _axis_slice(p, d, i)[...] = _axis_slice(p, d, N + i)
            _axis_slice(p, d, N + H + i)[...] = _axis_slice(p, d, H + i)
        return
    idx = _interior_other_dims(field, d)
    loc = field.loc[d]

    def at(i):  # reference index i along d, interior range in the others
        ii = list(idx)
        ii[d] = i - 1 + H
        return tuple(ii)

    def spacing(iB):
        # Delta between first interior and first halo point, at flip(loc)
        ax = g.ax[d]
        return ax.d_face(iB) if loc == Center else ax.d_center(iB)

    for side, bc in (("left", left), ("right", right)):
        if bc is None:
            continue
        if bc.kind == "flux":
            # fill_halo_regions_flux.jl:16-35: only the first halo cell
            if side == "left":
                p[at(0)] = p[at(1)]
            else:
                p[at(N + 1)] = p[at(N)]
        elif bc.kind in ("value", "gradient"):
            # fill_halo_regions_value_gradient.jl:7-99
            if side == "left":
                iI, iB, iH, sgn = 1, 1, 0, -1.0
            else:
                iI, iB, iH, sgn = N, N + 1, N + 1, 1.0
            D = spacing(iB)
            cI = p[at(iI)]
            val = _bc_array(bc.get(), field, d)
            if np.ndim(val) == 3:
                val = np.squeeze(val, axis=d)
            if bc.kind == "gradient":
                grad = val
            elif side == "left":
                grad = (cI - val) / (D / 2)
            else:
                grad = (val - cI) / (D / 2)
            p[at(iH)] = cI + grad * (sgn * D)
        elif bc.kind == "open":
            # fill_halo_regions_open.jl:34-39: the boundary *face* itself
            val = _bc_array(bc.get(), field, d)
            if np.ndim(val) == 3:
                val = np.squeeze(val, axis=d)
            if side == "left":
                p[at(1)] = val
            else:
                p[at(N + 1)] = val
        else:
            raise ValueError(bc.kind)


def fill_halo_regions(fields):
    """``fill_halo_regions.jl:34-102``: non-periodic directions first, periodic last."""
    if isinstance(fields, Field):
        fields = [fields]
    for f in fields:
        if f is None:
            continue
        kinds = [f.bcs[SIDES[d][0]] for d in range(3)]
        # Julia's insertion sort with the (non-strict) ``fill_first`` comparator: see DESIGN.md.
        order = [0, 1, 2]
        isper = lambda b: b is not None and b.kind == "periodic"  # noqa: E731

        def lt(a, b):
            if isper(kinds[a]) and not isper(kinds[b]):
                return False
            return True
        v = [0, 1, 2]
        for i in range(1, 3):
            x = v[i]
            j = i
            while j > 0 and lt(x, v[j - 1]):
                v[j] = v[j - 1]
                j -= 1
            v[j] = x
        order = v
        for d in order:
            _fill_one_direction(f, d)
