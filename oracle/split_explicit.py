"""SplitExplicitFreeSurface of the HydrostaticFreeSurfaceModel, and the horizontal grids it runs on (oracle; test
infrastructure only -- nothing in the product imports this file).

Restates, in plain NumPy on parent arrays (reference index i lives at parent index i - 1 + H):
  * ``Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface_kernels.jl:14-29`` (the two substep kernels),
    ``:31-58`` (``split_explicit_free_surface_substep!``: fill eta, kernel 1, fill U and V, kernel 2),
    ``:63-81`` (``barotropic_mode!``: vertical sum of u dz, v dz, then fill), ``:83-87`` (``set_average_to_zero!``),
    ``:89-113`` (the corrector), ``:115`` (``calc_ab2_tendencies``), ``:124-171`` (``split_explicit_free_surface_step!``);
  * ``split_explicit_free_surface.jl:78-117`` (state / auxiliary fields, H = sum of dz), ``:137-154`` (settings: uniform
    weights 1 / substeps);
  * ``Operators/derivative_operators.jl`` (d/dx at fcc = delta_x / dx^fcc), ``divergence_operators.jl:35-37``
    (div_xy^ccc = [delta_x(dy^fcc U) + delta_y(dx^cfc V)] / Az^ccc);
  * ``Grids/latitude_longitude_grid.jl:174-213`` (constructor: Periodic longitude when it spans 360 degrees, latitude and z
    Bounded), ``:418-443`` (metrics, regular longitude: dx^fc[j] = R cos(phi^c_j) dlam, dx^cf[j] = R cos(phi^f_j) dlam,
    dy = R dphi, Az^cc[j] = R^2 dlam (sin phi^f_{j+1} - sin phi^f_j) with hack_cosd / hack_sind = cos / sin(pi phi / 180));
  * halo fills of reduced fields ``Field{LX, LY, Nothing}``: ``BoundaryConditions/fill_halo_regions*.jl`` in x and y
    (Bounded directions first, then Periodic; Center: no-flux = first halo cell copies the edge cell; Face in a Bounded
    direction: impenetrable = the two boundary faces are set to zero, ``fill_halo_regions_open.jl:34-39``).

Only regular longitude / latitude axes (a stretched latitude axis runs into ``precompute_Δy_kernel!`` assigning Δy^cf to
Δy^fc as written, ``latitude_longitude_grid.jl:520-530`` -- not restated).
"""
import numpy as np

from .grid import Axis, Bounded, Center, Face, Flat, Periodic, total_length

R_EARTH = 6371.0e3            # latitude_longitude_grid.jl:3
G_EARTH = 9.80665             # Oceananigans.jl g_Earth


class _HGrid:
    """what the free surface needs from a grid: sizes, topology, nodes, per-row horizontal metrics, level thicknesses"""

    def total(self, lx, ly):
        return (total_length(lx, self.topo[0], self.Nx, self.Hx), total_length(ly, self.topo[1], self.Ny, self.Hy))

    def nodes(self, loc, d):
        a = self.ax[d]
        if loc == Center:
            return a.C[a.H:a.H + a.N].copy()
        n = a.N + 1 if a.topo == Bounded else a.N
        return a.F[a.H:a.H + n].copy()

    def dz_centers(self):
        """dz^aac[k], k = 1..Nz  (dz^fcc = dz^cfc = dz^ccc on these grids)"""
        a = self.ax[2]
        return np.full(self.Nz, a.dc) if a.regular else np.asarray(a.d_center(np.arange(1, self.Nz + 1)), dtype=float)


class HRectilinearGrid(_HGrid):
    """RectilinearGrid (regular x, y) seen by the free surface"""
    kind = "rectilinear"

    def __init__(self, size, x, y, z, halo=(3, 3, 3), topology=(Periodic, Periodic, Bounded)):
        self.topo = tuple(topology)
        assert Flat not in self.topo and self.topo[2] == Bounded
        self.ax = [Axis(self.topo[d], size[d], halo[d], c) for d, c in enumerate((x, y, z))]
        self.Nx, self.Ny, self.Nz = size
        self.Hx, self.Hy, self.Hz = halo
        ny = self.Ny + 2 * self.Hy + 1
        dx, dy = self.ax[0].dc, self.ax[1].dc
        # per-row metrics, entry [j - 1 + Hy] for reference row j (constant on this grid)
        self.dx_fc = np.full(ny, dx); self.dx_cf = np.full(ny, dx)
        self.dy_fc = np.full(ny, dy); self.dy_cf = np.full(ny, dy)
        self.Az_cc = np.full(ny, dx * dy)
        self.Az_ff = np.full(ny, dx * dy)                                       # Az = dx dy at every location
        self.phi_f = None


class LatitudeLongitudeGrid(_HGrid):
    """LatitudeLongitudeGrid(size, longitude, latitude, z, halo, radius) with precomputed metrics; regular longitude / latitude"""
    kind = "latlon"

    def __init__(self, size, longitude, latitude, z, halo=(3, 3, 3), radius=R_EARTH, topology=None):
        l1, l2 = longitude
        p1, p2 = latitude
        assert l1 <= l2 and l2 - l1 <= 360 and -90 <= p1 <= p2 <= 90          # validate_lat_lon_grid_args :258-262
        if topology is None:
            topology = (Periodic if (l2 - l1) == 360 else Bounded, Bounded, Bounded)      # :269-271
        self.topo = tuple(topology)
        self.ax = [Axis(self.topo[d], size[d], halo[d], c) for d, c in enumerate((longitude, latitude, z))]
        self.Nx, self.Ny, self.Nz = size
        self.Hx, self.Hy, self.Hz = halo
        self.radius = float(radius)
        a = self.ax[1]
        dlam, dphi = self.ax[0].dc, a.dc
        hack_cosd = lambda p: np.cos(np.pi * p / 180)      # noqa: E731   :418-419
        hack_sind = lambda p: np.sin(np.pi * p / 180)      # noqa: E731
        ny = self.Ny + 2 * self.Hy + 1
        phic = np.full(ny, np.nan); phic[:a.C.size] = a.C
        phif = np.full(ny + 1, np.nan); phif[:a.F.size] = a.F
        R = self.radius
        self.dx_fc = R * hack_cosd(phic) * np.deg2rad(dlam)                    # :436, rows where phi^c exists
        self.dx_cf = R * hack_cosd(phif[:ny]) * np.deg2rad(dlam)               # :437
        self.dy_fc = np.full(ny, R * np.deg2rad(dphi))                         # :441 (YRegLatLonGrid: one number)
        self.dy_cf = np.full(ny, R * np.deg2rad(dphi))                         # :440
        self.Az_cc = R ** 2 * np.deg2rad(dlam) * (hack_sind(phif[1:ny + 1]) - hack_sind(phif[:ny]))   # :445
        self.Az_ff = np.full(ny, np.nan)                                         # :444; regular longitude: Az^fc = Az^cc, Az^cf = Az^ff
        self.Az_ff[1:] = R ** 2 * np.deg2rad(dlam) * (hack_sind(phic[1:]) - hack_sind(phic[:-1]))
        self.phi_f = phif[:ny]                                                   # latitude of the rows of faces (Coriolis)


class ReducedField:
    """Field{LX, LY, Nothing}(grid): a 2-D parent array with x / y halos (Fields/field.jl:441-449)"""

    def __init__(self, grid, lx, ly):
        self.grid, self.loc = grid, (lx, ly)
        self.data = np.zeros(grid.total(lx, ly), order="F")

    def size(self):
        g = self.grid
        return tuple(n + 1 if (l == Face and t == Bounded) else n for n, l, t in zip((g.Nx, g.Ny), self.loc, g.topo))

    def interior(self):
        g, s = self.grid, self.size()
        return self.data[g.Hx:g.Hx + s[0], g.Hy:g.Hy + s[1]]

    def interior3(self):
        return self.interior().reshape(self.size() + (1,))

    def set(self, value):
        it = self.interior()
        if callable(value):
            g = self.grid
            X, Y = g.nodes(self.loc[0], 0).reshape(-1, 1), g.nodes(self.loc[1], 1).reshape(1, -1)
            it[...] = value(X, Y) + 0 * (X + Y)
        else:
            it[...] = value

    # the accessors of the library's fields (clima-oceananigans.jl_amd/hydrostatic.py HField), so tests run on either
    def parent(self):
        return self.data.reshape(self.data.shape + (1,)).copy()

    def set_parent(self, a):
        self.data[...] = np.asarray(a, dtype=float).reshape(self.data.shape) if np.ndim(a) else a

    def fill(self, value):
        self.data[...] = value

    def fill_halo_regions(self):
        fill_halo_regions(self)


class Field3:
    """Field{LX, LY, LZ}(grid) on one of the grids above, LZ = Center (u, v, tracers, tendencies, pHY') or Face (w); z is Bounded"""

    def __init__(self, grid, lx, ly, lz=Center):
        self.grid, self.loc = grid, (lx, ly, lz)
        self.data = np.zeros(grid.total(lx, ly) + (grid.Nz + 2 * grid.Hz + (1 if lz == Face else 0),), order="F")

    def size(self):
        g = self.grid
        s = tuple(n + 1 if (l == Face and t == Bounded) else n for n, l, t in zip((g.Nx, g.Ny), self.loc, g.topo))
        return s + (g.Nz + 1 if self.loc[2] == Face else g.Nz,)

    def interior(self):
        g, s = self.grid, self.size()
        return self.data[g.Hx:g.Hx + s[0], g.Hy:g.Hy + s[1], g.Hz:g.Hz + s[2]]

    def set(self, value):
        it = self.interior()
        if callable(value):
            g = self.grid
            X, Y = g.nodes(self.loc[0], 0).reshape(-1, 1, 1), g.nodes(self.loc[1], 1).reshape(1, -1, 1)
            Z = g.nodes(Center, 2).reshape(1, 1, -1)
            it[...] = value(X, Y, Z) + 0 * (X + Y + Z)
        else:
            it[...] = value

    def parent(self):
        return self.data.copy()

    def set_parent(self, a):
        self.data[...] = a

    def fill(self, value):
        self.data[...] = value

    def fill_halo_regions(self):
        fill_halo_regions(self)


def fill_halo_regions(f):
    """fills of a reduced field (x, y) or a 3-D field (z, x, y) with the default conditions: Bounded directions first, over
    the interior cells of the other directions; Periodic last, over the whole parent"""
    g = f.grid
    if len(f.loc) == 3 and (g.Hz > 0 or f.loc[2] == Face):
        # z (Bounded) over i = 1..Nx, j = 1..Ny: Center -> no-flux, Face -> the two boundary faces are zeroed (the default
        # impenetrable condition of a ZFaceField, fill_halo_regions_open.jl:34-39)
        q, Hz, Nz = f.data, g.Hz, g.Nz
        I, J = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny)
        if f.loc[2] == Face:
            q[I, J, Hz] = 0.0
            q[I, J, Hz + Nz] = 0.0
        else:
            q[I, J, Hz - 1] = q[I, J, Hz]
            q[I, J, Hz + Nz] = q[I, J, Hz + Nz - 1]
    order = sorted((0, 1), key=lambda d: g.topo[d] == Periodic)       # stable: x before y within a class (fill_halo_regions.jl:76-99)
    p = f.data
    for d in order:
        N, H, loc = (g.Nx, g.Ny)[d], (g.Hx, g.Hy)[d], f.loc[d]
        other = 1 - d
        No, Ho = (g.Nx, g.Ny)[other], (g.Hx, g.Hy)[other]

        def at(i, full):
            """reference index i along d; the other direction over its interior cells (:xy-style launch) or whole extent"""
            idx = [slice(None)] * p.ndim
            idx[d] = i - 1 + H
            if not full:
                idx[other] = slice(Ho, Ho + No)
                if len(f.loc) == 3:                       # the `:yz` / `:xz` launch: levels 1..Nz of the grid only
                    idx[2] = slice(g.Hz, g.Hz + g.Nz)
            return tuple(idx)
        if g.topo[d] == Periodic:
            for i in range(1, H + 1):                     # fill_halo_regions_periodic.jl:37-65, sequential, whole parent extent
                p[at(i - H, True)] = p[at(N + i - H, True)]
                p[at(N + i, True)] = p[at(i, True)]
        elif loc == Center:                               # no-flux: fill_halo_regions_flux.jl:16-35
            p[at(0, False)] = p[at(1, False)]
            p[at(N + 1, False)] = p[at(N, False)]
        else:                                             # Face in a Bounded direction: impenetrable, fill_halo_regions_open.jl:34-39
            p[at(1, False)] = 0.0
            p[at(N + 1, False)] = 0.0


class SplitExplicitFreeSurface:
    """SplitExplicitFreeSurface(grid; gravitational_acceleration, settings = SplitExplicitSettings(substeps))"""

    def __init__(self, grid, gravitational_acceleration=G_EARTH, substeps=200):
        g = self.grid = grid
        self.g = float(gravitational_acceleration)
        self.eta = ReducedField(g, Center, Center)
        self.U, self.Ubar = ReducedField(g, Face, Center), ReducedField(g, Face, Center)
        self.V, self.Vbar = ReducedField(g, Center, Face), ReducedField(g, Center, Face)
        self.etabar = ReducedField(g, Center, Center)
        self.GU, self.GV = ReducedField(g, Face, Center), ReducedField(g, Center, Face)
        self.Hfc, self.Hcf, self.Hcc = ReducedField(g, Face, Center), ReducedField(g, Center, Face), ReducedField(g, Center, Center)
        H = 0.0
        for dz in g.dz_centers():                          # sum!(H, dz): interior only, level by level (:103-110)
            H = H + dz
        for f in (self.Hfc, self.Hcf, self.Hcc):
            f.interior()[...] = H
        self.set_weights(np.ones(substeps) / substeps, np.ones(substeps) / substeps)

    def set_weights(self, velocity_weights, free_surface_weights):
        self.velocity_weights = np.asarray(velocity_weights, dtype=float)
        self.free_surface_weights = np.asarray(free_surface_weights, dtype=float)
        assert self.velocity_weights.size == self.free_surface_weights.size
        self.substeps = self.velocity_weights.size

    # ---- split_explicit_free_surface_substep! (:31-58) --------------------------------------------------------------------
    def substep(self, dtau, substep_index):
        g = self.grid
        Hx, Hy, Nx, Ny = g.Hx, g.Hy, g.Nx, g.Ny
        I, J = slice(Hx, Hx + Nx), slice(Hy, Hy + Ny)
        Im, Jm = slice(Hx - 1, Hx + Nx - 1), slice(Hy - 1, Hy + Ny - 1)
        Ip, Jp = slice(Hx + 1, Hx + Nx + 1), slice(Hy + 1, Hy + Ny + 1)
        row = lambda a: a[Hy:Hy + Ny].reshape(1, -1)       # noqa: E731  metrics of rows j = 1..Ny
        rowp = lambda a: a[Hy + 1:Hy + Ny + 1].reshape(1, -1)   # noqa: E731  rows j + 1
        eta, U, V = self.eta.data, self.U.data, self.V.data
        fill_halo_regions(self.eta)
        # kernel 1 (:14-19): i = 1..Nx, j = 1..Ny
        ddx = (eta[I, J] - eta[Im, J]) / row(g.dx_fc)
        ddy = (eta[I, J] - eta[I, Jm]) / row(g.dy_cf)
        U[I, J] += dtau * (-self.g * self.Hfc.data[I, J] * ddx + self.GU.data[I, J])
        V[I, J] += dtau * (-self.g * self.Hcf.data[I, J] * ddy + self.GV.data[I, J])
        fill_halo_regions(self.U)
        fill_halo_regions(self.V)
        # kernel 2 (:21-29)
        div = 1 / row(g.Az_cc) * ((row(g.dy_fc) * U[Ip, J] - row(g.dy_fc) * U[I, J]) + (rowp(g.dx_cf) * V[I, Jp] - row(g.dx_cf) * V[I, J]))
        eta[I, J] -= dtau * div
        vw, fw = self.velocity_weights[substep_index - 1], self.free_surface_weights[substep_index - 1]
        self.Ubar.data[I, J] += vw * U[I, J]
        self.Vbar.data[I, J] += vw * V[I, J]
        self.etabar.data[I, J] += fw * eta[I, J]

    # ---- barotropic_mode! (:63-81) --------------------------------------------------------------------------------------------
    def barotropic_mode(self, U, V, u, v):
        """sum!(U, u * dz); sum!(V, v * dz) over the interior of the reduced field, level 1 first; then fill_halo_regions!((U, V))"""
        dz = self.grid.dz_centers()
        for F, f in ((U, u), (V, v)):
            it, src = F.interior(), f.interior()
            acc = dz[0] * src[:, :, 0]
            for k in range(1, self.grid.Nz):
                acc = acc + dz[k] * src[:, :, k]
            it[...] = acc
        fill_halo_regions(U)
        fill_halo_regions(V)

    def set_average_to_zero(self):
        for f in (self.etabar, self.Ubar, self.Vbar):
            f.data[...] = 0.0                              # fill!: the whole parent array

    # ---- barotropic_split_explicit_corrector! (:89-113) -------------------------------------------------------------------
    def corrector(self, u, v):
        self.barotropic_mode(self.U, self.V, u, v)
        g = self.grid
        I, J, K = slice(g.Hx, g.Hx + g.Nx), slice(g.Hy, g.Hy + g.Ny), slice(g.Hz, g.Hz + g.Nz)
        du = (-self.U.data[I, J] + self.Ubar.data[I, J]) / self.Hfc.data[I, J]
        dv = (-self.V.data[I, J] + self.Vbar.data[I, J]) / self.Hcf.data[I, J]
        u.data[I, J, K] = u.data[I, J, K] + du[:, :, None]
        v.data[I, J, K] = v.data[I, J, K] + dv[:, :, None]

    # ---- split_explicit_free_surface_step! (:124-171) ---------------------------------------------------------------------
    def substeps_train(self, dtau, first, count, fused=True):
        for s in range(first, first + count):
            self.substep(dtau, s)

    def step(self, Gnu, Gnv, Gmu, Gmv, dt, chi):
        dtau = 2 * dt / self.substeps
        Gu, Gv = Field3(self.grid, Face, Center), Field3(self.grid, Center, Face)
        Gu.data[...] = (1.5 + chi) * Gnu.data - (0.5 + chi) * Gmu.data          # calc_ab2_tendencies (:115)
        Gv.data[...] = (1.5 + chi) * Gnv.data - (0.5 + chi) * Gmv.data
        self.set_average_to_zero()
        self.barotropic_mode(self.GU, self.GV, Gu, Gv)
        for s in range(1, self.substeps + 1):
            self.substep(dtau, s)
        self.eta.data[...] = self.etabar.data              # set!(eta, etabar) copies the PARENT array (Fields/set!.jl:41-44) ...
        fill_halo_regions(self.eta)                        # ... and the halos are filled afterwards (:166)
