"""Host-side mirror of the HydrostaticFreeSurfaceModel pieces the library carries so far (BASELINE config 5, first two slices):
``LatitudeLongitudeGrid`` / ``RectilinearGrid`` as the free surface sees them, ``Field{LX, LY, LZ}`` on them, and
``SplitExplicitFreeSurface`` with the reference's verbs -- ``split_explicit_free_surface_substep!``, ``barotropic_mode!``,
``set_average_to_zero!``, ``barotropic_split_explicit_corrector!``, ``split_explicit_free_surface_step!``
(Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface.jl, split_explicit_free_surface_kernels.jl).
Second slice: the AB2 time step around the tendency evaluation -- ``ab2_step!``, the barotropic correction, ``store_tendencies!``
and ``update_state!`` (``compute_w_from_continuity!``, ``update_hydrostatic_pressure!``, halo fills) on a ``HydrostaticState``.
Everything numerical happens in libocnhip.so (csrc/splitexplicit.hip); this file only marshals.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import check
from .api import Bounded, Center, Face, Periodic, default_context

Nothing = "Nothing"
_TOPO = {Periodic: L.PERIODIC, Bounded: L.BOUNDED}
_LOC = {Center: L.CENTER, Face: L.FACE, Nothing: L.NOTHING, None: L.NOTHING}
R_Earth = 6371.0e3
OMEGA_EARTH = 7.292115e-5
g_Earth = 9.80665


class _HGrid:
    def _create(self, ctx, kind, size, halo, topology, lo, ext, z, radius, partition=None, band_overlap=0):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        d = L.HGridDesc()
        d.kind = kind
        self.topology = tuple(topology)
        self.Nx, self.Ny, self.Nz = (int(n) for n in size)
        self.Hx, self.Hy, self.Hz = (int(h) for h in halo)
        for q in range(3):
            d.N[q], d.H[q], d.topology[q] = int(size[q]), int(halo[q]), _TOPO[topology[q]]
            d.x0[q], d.L[q] = float(lo[q]), float(ext[q])
        self._zf = None
        if z is not None and len(z) != 2:
            self._zf = np.ascontiguousarray(z, dtype=np.float64)
            if self._zf.size != self.Nz + 1:
                raise ValueError("z must be (z1, z2) or hold Nz + 1 faces")
            d.z_faces = self._zf.ctypes.data_as(C.POINTER(C.c_double))
        d.radius = float(radius)
        if partition not in (None, "y"):
            raise ValueError("partition: None or 'y' (latitude bands over the context's ranks)")
        d.partition = 1 if partition == "y" else 0
        d.band_overlap = int(band_overlap)
        self.band_overlap = int(band_overlap)
        self.h = C.c_void_p()
        check(self.lib.ocn_hgrid_create(self.ctx.h, C.byref(d), C.byref(self.h)), self.ctx.h)
        j0, nl, ng = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.ocn_hgrid_band(self.h, C.byref(j0), C.byref(nl), C.byref(ng)), self.ctx.h)
        self.j0, self.Ny, self.global_Ny = j0.value, nl.value, ng.value      # this handle's rows of the global grid
        self.partition = partition if self.Ny != self.global_Ny else None

    def metric(self, which):
        n = self.lib.ocn_hgrid_metric(self.h, int(which), (C.c_double * 1)(), 0)
        out = np.zeros(n)
        self.lib.ocn_hgrid_metric(self.h, int(which), out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out

    # the grid's own arrays, reference names; rows / nodes include halos (first entry = index 1 - H)
    Δxᶠᶜᵃ = property(lambda s: s.metric(0)); Δxᶜᶠᵃ = property(lambda s: s.metric(1))
    Δyᶠᶜᵃ = property(lambda s: s.metric(2)); Δyᶜᶠᵃ = property(lambda s: s.metric(3))
    Azᶜᶜᵃ = property(lambda s: s.metric(4)); Δzᵃᵃᶜ = property(lambda s: s.metric(5))

    def nodes(self, loc, d):
        """interior nodes along x (d = 0) or y (d = 1)"""
        a = self.metric((6 if loc == Face else 7) + 2 * d)
        N, H = (self.Nx, self.Ny)[d], (self.Hx, self.Hy)[d]
        n = N + 1 if (loc == Face and self.topology[d] == Bounded) else N
        return a[H:H + n]

    def znodes(self, loc=Center):
        if self._zf is not None:
            return self._zf.copy() if loc == Face else 0.5 * (self._zf[1:] + self._zf[:-1])
        dz = self.Δzᵃᵃᶜ
        if loc == Face:
            return self._z0 + np.concatenate([[0.0], np.cumsum(dz)])
        return self._z0 + np.cumsum(dz) - dz / 2

    Δzᵃᵃᶠ = property(lambda s: s.metric(10))

    def whole(self):
        """the unpartitioned grid of a latitude band (what the replicated free surface lives on); the grid itself otherwise"""
        if self.partition is None:
            return self
        cls, kw = self._ctor
        return cls(**kw)

    def extended(self, overlap):
        """the band extended by `overlap` rows towards each neighbouring band, as a stand-alone Bounded grid: what a banded free
        surface sub-cycles on (ocn_hgrid_desc.band_overlap)"""
        cls, kw = self._ctor
        return cls(partition="y", band_overlap=int(overlap), **kw)

    def __del__(self):
        try:
            if self.h:
                self.lib.ocn_hgrid_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


class HRectilinearGrid(_HGrid):
    """RectilinearGrid(size, x, y, z, halo, topology) with regular x and y, for the hydrostatic pieces"""

    def __init__(self, size, x, y, z, halo=(3, 3, 3), topology=(Periodic, Periodic, Bounded), arch=None, partition=None, band_overlap=0):
        zr = z if len(z) == 2 else (z[0], z[-1])
        self._z0 = float(zr[0])
        self._ctor = (HRectilinearGrid, dict(size=size, x=x, y=y, z=z, halo=halo, topology=topology, arch=arch))
        self._create(arch, L.HGRID_RECTILINEAR, size, halo, topology, (x[0], y[0], zr[0]), (x[1] - x[0], y[1] - y[0], zr[1] - zr[0]), z, 0.0,
                     partition, band_overlap)


class LatitudeLongitudeGrid(_HGrid):
    """LatitudeLongitudeGrid(size, longitude, latitude, z, halo, radius) -- Grids/latitude_longitude_grid.jl:174-213; regular
    longitude and latitude, metrics precomputed.  Topology as the reference chooses it: Periodic longitude iff it spans 360."""

    def __init__(self, size, longitude, latitude, z, halo=(3, 3, 3), radius=R_Earth, topology=None, arch=None, partition=None,
                 band_overlap=0):
        l1, l2 = longitude
        p1, p2 = latitude
        if not (l1 <= l2 and l2 - l1 <= 360 and -90 <= p1 <= p2 <= 90):
            raise ValueError("longitude must span at most 360 degrees and latitude lie within [-90, 90]")
        if topology is None:
            topology = (Periodic if (l2 - l1) == 360 else Bounded, Bounded, Bounded)
        zr = z if len(z) == 2 else (z[0], z[-1])
        self._z0 = float(zr[0])
        self.radius = float(radius)
        self._ctor = (LatitudeLongitudeGrid, dict(size=size, longitude=longitude, latitude=latitude, z=z, halo=halo, radius=radius,
                                                  topology=topology, arch=arch))
        self._create(arch, L.HGRID_LATLON, size, halo, topology, (l1, p1, zr[0]), (l2 - l1, p2 - p1, zr[1] - zr[0]), z, radius, partition,
                     band_overlap)


class HField:
    """Field{LX, LY, LZ}(grid), LZ = Center, Face or Nothing: a dense parent array on the device"""

    def __init__(self, grid, loc, handle=None):
        self.grid, self.loc, self.lib = grid, tuple(loc), grid.lib
        self._owned = handle is None
        if handle is None:
            self.h = C.c_void_p()
            check(self.lib.ocn_hfield_create(grid.h, _LOC[loc[0]], _LOC[loc[1]], _LOC[loc[2]], C.byref(self.h)), grid.ctx.h)
        else:
            self.h = C.c_void_p(handle)
        T, S, H = (C.c_int32 * 3)(), (C.c_int32 * 3)(), (C.c_int32 * 3)()
        check(self.lib.ocn_hfield_shape(self.h, C.byref(T), C.byref(S), C.byref(H)), grid.ctx.h)
        self.total, self.size, self.halo = tuple(T), tuple(S), tuple(H)

    def parent(self):
        a = np.zeros(self.total, order="F")
        check(self.lib.ocn_hfield_download(self.h, a.ctypes.data_as(C.POINTER(C.c_double))), self.grid.ctx.h)
        return a

    def set_parent(self, a):
        a = np.asarray(a, dtype=np.float64)
        a = np.full(self.total, float(a)) if a.ndim == 0 else a.reshape(self.total)
        a = np.asfortranarray(a)
        check(self.lib.ocn_hfield_upload(self.h, a.ctypes.data_as(C.POINTER(C.c_double))), self.grid.ctx.h)

    def _interior_slices(self):
        return tuple(slice(h, h + s) for h, s in zip(self.halo, self.size))

    def interior(self):
        return self.parent()[self._interior_slices()]

    def set(self, value):
        """set!(field, number | array | function of the nodes): the interior; halos untouched (Fields/set!.jl)"""
        p = self.parent()
        it = p[self._interior_slices()]
        if callable(value):
            g = self.grid
            X = g.nodes(self.loc[0], 0).reshape(-1, 1, 1)
            Y = g.nodes(self.loc[1], 1).reshape(1, -1, 1)
            if self.loc[2] in (Nothing, None):
                it[...] = value(X, Y) + 0 * (X + Y)
            else:
                Z = g.znodes(self.loc[2]).reshape(1, 1, -1)
                it[...] = value(X, Y, Z) + 0 * (X + Y + Z)
        else:
            it[...] = np.asarray(value, dtype=np.float64).reshape(it.shape) if np.ndim(value) else value
        self.set_parent(p)

    def fill(self, value):
        """`field .= value` on the WHOLE parent array"""
        self.set_parent(np.full(self.total, float(value)))

    def fill_halo_regions(self):
        check(self.lib.ocn_hfield_fill_halos(self.h), self.grid.ctx.h)

    def device_ptr(self):
        return self.lib.ocn_hfield_ptr(self.h)

    def __del__(self):
        try:
            if self._owned and self.h:
                self.lib.ocn_hfield_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


_SEFS_FIELDS = (("η", (Center, Center)), ("U", (Face, Center)), ("V", (Center, Face)), ("η̅", (Center, Center)),
                ("U̅", (Face, Center)), ("V̅", (Center, Face)), ("Gᵁ", (Face, Center)), ("Gⱽ", (Center, Face)),
                ("Hᶠᶜ", (Face, Center)), ("Hᶜᶠ", (Center, Face)), ("Hᶜᶜ", (Center, Center)))
_ASCII = {"η": "eta", "U": "U", "V": "V", "η̅": "etabar", "U̅": "Ubar", "V̅": "Vbar", "Gᵁ": "GU", "Gⱽ": "GV",
          "Hᶠᶜ": "Hfc", "Hᶜᶠ": "Hcf", "Hᶜᶜ": "Hcc"}


class SplitExplicitFreeSurface:
    """SplitExplicitFreeSurface(grid; gravitational_acceleration = g_Earth, settings = SplitExplicitSettings(substeps))"""

    def __init__(self, grid, gravitational_acceleration=g_Earth, substeps=200):
        self.grid, self.lib = grid, grid.lib
        self.gravitational_acceleration = float(gravitational_acceleration)
        self.h = C.c_void_p()
        check(self.lib.ocn_sefs_create(grid.h, self.gravitational_acceleration, int(substeps), C.byref(self.h)), grid.ctx.h)
        self.substeps = int(substeps)
        self.fields = {}
        for q, (name, loc) in enumerate(_SEFS_FIELDS):
            f = HField(grid, loc + (Nothing,), handle=self.lib.ocn_sefs_field(self.h, q))
            self.fields[name] = f
            setattr(self, _ASCII[name], f)

    def set_weights(self, velocity_weights, free_surface_weights):
        vw = np.ascontiguousarray(velocity_weights, dtype=np.float64)
        fw = np.ascontiguousarray(free_surface_weights, dtype=np.float64)
        PD = C.POINTER(C.c_double)
        check(self.lib.ocn_sefs_set_weights(self.h, vw.size, vw.ctypes.data_as(PD), fw.ctypes.data_as(PD)), self.grid.ctx.h)
        self.substeps = vw.size

    def substep(self, dtau, substep_index):
        check(self.lib.ocn_sefs_substep(self.h, float(dtau), int(substep_index)), self.grid.ctx.h)

    def substeps_train(self, dtau, first, count, fused=True):
        """fused: False / 0 the reference's five launches per substep, True / 1 two launches, 2 one launch, 3 four substeps per
        launch (hipGraph trains)"""
        check(self.lib.ocn_sefs_substeps(self.h, float(dtau), int(first), int(count), int(fused)), self.grid.ctx.h)

    @property
    def graph_replays(self):
        n = C.c_int64()
        check(self.lib.ocn_sefs_graph_replays(self.h, C.byref(n)), self.grid.ctx.h)
        return n.value

    def barotropic_mode(self, U, V, u, v):
        """barotropic_mode!(U, V, grid, u, v); (U, V) must be this free surface's (state.U, state.V) or (auxiliary.Gᵁ, Gⱽ)"""
        if U is self.U and V is self.V:
            into = 0
        elif U is self.GU and V is self.GV:
            into = 1
        else:
            raise ValueError("barotropic_mode: targets must be (state.U, state.V) or (auxiliary.Gᵁ, auxiliary.Gⱽ)")
        check(self.lib.ocn_sefs_barotropic_mode(self.h, u.h, v.h, into), self.grid.ctx.h)

    def set_average_to_zero(self):
        check(self.lib.ocn_sefs_set_average_to_zero(self.h), self.grid.ctx.h)

    def corrector(self, u, v):
        check(self.lib.ocn_sefs_corrector(self.h, u.h, v.h), self.grid.ctx.h)

    def step(self, Gnu, Gnv, Gmu, Gmv, dt, chi):
        check(self.lib.ocn_sefs_step(self.h, Gnu.h, Gnv.h, Gmu.h, Gmv.h, float(dt), float(chi)), self.grid.ctx.h)

    def __del__(self):
        try:
            if self.h:
                self.lib.ocn_sefs_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


# ---- second slice: the AB2 step of the hydrostatic model around its tendencies ----------------------------------------------------
def Field3(grid, lx, ly, lz=Center):
    return HField(grid, (lx, ly, lz))


def fill_halo_regions(f):
    f.fill_halo_regions()


def ab2_step_field(f, Gn, Gm, dt, chi):
    """ab2_step_field! (TimeSteppers/quasi_adams_bashforth_2.jl:158-166)"""
    check(f.lib.ocn_hfield_ab2_step(f.h, Gn.h, Gm.h, float(dt), float(chi)), f.grid.ctx.h)


def compute_w_from_continuity(u, v, w):
    """compute_w_from_continuity! (compute_w_from_continuity.jl:31-36)"""
    check(w.lib.ocn_hydro_compute_w(u.h, v.h, w.h), w.grid.ctx.h)


def _buoyancy_args(buoyancy, tracers):
    """None | ("b", name) | ("TS", g, alpha, beta, Tname, Sname) -> (kind, g, alpha, beta, T field, S field)"""
    if buoyancy is None:
        return 0, 0.0, 0.0, 0.0, None, None
    if buoyancy[0] == "b":
        return 1, 0.0, 0.0, 0.0, tracers[buoyancy[1]], None
    if buoyancy[0] == "TS":
        _, g, al, be, Tn, Sn = buoyancy
        return 2, float(g), float(al), float(be), tracers[Tn], tracers[Sn]
    raise ValueError(f"unsupported buoyancy {buoyancy!r}: None, ('b', name) or ('TS', g, alpha, beta, Tname, Sname)")


def update_hydrostatic_pressure(pHY, buoyancy, tracers):
    """update_hydrostatic_pressure! (Models/NonhydrostaticModels/update_hydrostatic_pressure.jl:10-18)"""
    kind, g, al, be, T, S = _buoyancy_args(buoyancy, tracers)
    check(pHY.lib.ocn_hydro_pressure(pHY.h, kind, g, al, be, T.h if T else None, S.h if S else None), pHY.grid.ctx.h)


class HydrostaticState:
    """the fields of a HydrostaticFreeSurfaceModel{SplitExplicitFreeSurface} the step after the tendencies touches: u, v, w, the
    tracers, G^n and G^- of the prognostic fields, pHY' and the free surface (hydrostatic_free_surface_model.jl:92-211)"""

    def __init__(self, grid, tracers=("T", "S"), buoyancy=None, substeps=20, gravitational_acceleration=g_Earth, free_surface=None,
                 momentum_advection="VectorInvariantEnstrophyConserving", coriolis=None, tracer_advection="CenteredSecondOrder",
                 barotropic_overlap=0, closure=None):
        self.grid, self.lib = grid, grid.lib
        self.chi = 0.1
        self.u, self.v, self.w = Field3(grid, Face, Center), Field3(grid, Center, Face), Field3(grid, Center, Center, Face)
        self.tracers = {n: Field3(grid, Center, Center) for n in tracers}
        names = ["u", "v"] + list(tracers)
        loc = {"u": (Face, Center), "v": (Center, Face)}
        self.Gn = {n: Field3(grid, *loc.get(n, (Center, Center))) for n in names}
        self.Gm = {n: Field3(grid, *loc.get(n, (Center, Center))) for n in names}
        self.pHY = Field3(grid, Center, Center)
        self.buoyancy = buoyancy
        # on latitude bands the free surface is replicated (it lives on the whole grid, every rank sub-cycles all of it) or, with
        # barotropic_overlap = W > 0, banded: it lives on the band extended by W rows and refreshes them every W substeps
        fsgrid = grid.whole() if not (grid.partition and barotropic_overlap) else grid.extended(barotropic_overlap)
        self.free_surface = free_surface or SplitExplicitFreeSurface(fsgrid, gravitational_acceleration, substeps)
        d = L.HydroDesc()
        d.free_surface = self.free_surface.h
        d.u, d.v, d.w, d.pHY = self.u.h, self.v.h, self.w.h, self.pHY.h
        tl = list(self.tracers.values())
        d.ntracers = len(tl)
        self._arr = [(C.c_void_p * max(1, len(tl)))(*[t.h for t in tl]), (C.c_void_p * len(names))(*[self.Gn[n].h for n in names]),
                     (C.c_void_p * len(names))(*[self.Gm[n].h for n in names])]
        d.tracers, d.Gn, d.Gm = self._arr
        kind, g, al, be, T, S = _buoyancy_args(buoyancy, self.tracers)
        d.buoyancy_kind = kind
        d.T_index = tl.index(T) if T is not None else -1
        d.S_index = tl.index(S) if S is not None else -1
        d.gravitational_acceleration, d.thermal_expansion, d.haline_contraction = g, al, be
        self.h = C.c_void_p()
        check(self.lib.ocn_hydro_create(C.byref(d), C.byref(self.h)), grid.ctx.h)
        self.set_physics(momentum_advection, coriolis, tracer_advection)
        self.set_closure(closure)

    def set_closure(self, closure):
        """None | (nu, kappa | {tracer: kappa}): VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu, kappa), constants"""
        self.closure = closure
        nu, kap = closure or (0.0, 0.0)
        names = list(self.tracers)
        k = np.array([float(kap.get(n, 0.0)) if isinstance(kap, dict) else float(kap) for n in names] or [0.0])
        check(self.lib.ocn_hydro_set_closure(self.h, float(nu), len(names), k.ctypes.data_as(C.POINTER(C.c_double))), self.grid.ctx.h)

    def set_physics(self, momentum_advection, coriolis, tracer_advection):
        """momentum_advection: None | "VectorInvariantEnstrophyConserving" | "VectorInvariantEnergyConserving" |
        "WENOVectorInvariantVorticityStencil" (WENO5(vector_invariant = VorticityStencil()));
        coriolis: None | ("HydrostaticSphericalCoriolis", rotation_rate, "EnstrophyConserving" | "EnergyConserving") | ("FPlane", f);
        tracer_advection: None | "CenteredSecondOrder" | "CenteredFourthOrder" | "UpwindBiasedFifthOrder" | "WENO5" """
        ma = {None: 0, "VectorInvariantEnstrophyConserving": 1, "VectorInvariantEnergyConserving": 2,
              "WENOVectorInvariantVorticityStencil": 3}[momentum_advection]
        ta = {None: 0, "CenteredSecondOrder": 1, "CenteredFourthOrder": 2, "UpwindBiasedFifthOrder": 3, "WENO5": 4}[tracer_advection]
        if coriolis is None:
            ck, cp = 0, 0.0
        elif coriolis[0] == "FPlane":
            ck, cp = 3, float(coriolis[1])
        elif coriolis[0] == "HydrostaticSphericalCoriolis":
            ck, cp = {"EnstrophyConserving": 1, "EnergyConserving": 2}[coriolis[2]], float(coriolis[1])
        else:
            raise ValueError(f"unsupported coriolis {coriolis!r}")
        check(self.lib.ocn_hydro_set_physics(self.h, ma, ck, cp, ta), self.grid.ctx.h)
        self.momentum_advection, self.coriolis, self.tracer_advection = momentum_advection, coriolis, tracer_advection

    def __del__(self):
        try:
            if self.h:
                self.lib.ocn_hydro_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


def update_state(st):
    """update_state!(model) (update_hydrostatic_free_surface_model_state.jl:21-48)"""
    check(st.lib.ocn_hydro_update_state(st.h), st.grid.ctx.h)


def ab2_step(st, dt, chi):
    """ab2_step!(model, dt, chi) (hydrostatic_free_surface_ab2_step.jl:15-48)"""
    check(st.lib.ocn_hydro_ab2_step(st.h, float(dt), float(chi)), st.grid.ctx.h)


def time_step_after_tendencies(st, dt, chi, fused=True):
    """time_step!(model, dt) from `ab2_step!` on (quasi_adams_bashforth_2.jl:94-100); fused=False issues the reference's kernels
    one by one, fused=True the merged passes (same bits)"""
    check(st.lib.ocn_hydro_step_after_tendencies(st.h, float(dt), float(chi), int(bool(fused))), st.grid.ctx.h)


def calculate_tendencies(st):
    """calculate_tendencies!(model) (calculate_hydrostatic_free_surface_tendencies.jl:15-160)"""
    check(st.lib.ocn_hydro_calculate_tendencies(st.h), st.grid.ctx.h)


def time_step(st, dt, euler=False):
    """time_step!(model, dt; euler) (TimeSteppers/quasi_adams_bashforth_2.jl:70-104)"""
    check(st.lib.ocn_hydro_time_step(st.h, float(dt), int(bool(euler))), st.grid.ctx.h)
