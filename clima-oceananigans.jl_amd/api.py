"""Host-side mirror of the reference's user API for the time_step! path.

Each class / function names the reference constructor or verb it mirrors (paths relative to
/root/reference/src).  Nothing here computes: every call forwards to libocnhip.so.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import OcnError, check

Periodic, Bounded, Flat = "Periodic", "Bounded", "Flat"
Center, Face = "Center", "Face"
_TOPO = {Periodic: L.PERIODIC, Bounded: L.BOUNDED, Flat: L.FLAT}


# ---- advection schemes (Advection/) -----------------------------------------------------------------
class NoAdvection:
    """``advection = nothing`` (nonhydrostatic_model.jl:106): no advective terms at all."""
    code = L.ADV_NONE


class CenteredSecondOrder:
    code = L.ADV_C2


class CenteredFourthOrder:
    code = L.ADV_C4


class UpwindBiasedFifthOrder:
    code = L.ADV_U5


class UpwindBiasedFirstOrder:
    code = L.ADV_U1


class UpwindBiasedThirdOrder:
    code = L.ADV_U3


class WENO5:
    """``WENO5(; zweno=true)`` (weno_fifth_order.jl:162-180); uniform coefficients."""

    def __init__(self, zweno=True):
        self.zweno = zweno
        self.code = L.ADV_WENO5_Z if zweno else L.ADV_WENO5_JS


# ---- closures / physics ---------------------------------------------------------------------------------
class ScalarDiffusivity:
    """``ScalarDiffusivity(ν=, κ=)`` explicit, three-dimensional (scalar_diffusivity.jl)."""

    def __init__(self, nu=0.0, kappa=0.0):
        self.nu, self.kappa = nu, kappa


class AnisotropicMinimumDissipation:
    """``AnisotropicMinimumDissipation(; C=1/12, Cν, Cκ, Cb=nothing)`` (anisotropic_minimum_dissipation.jl:110-119)."""

    def __init__(self, C=1 / 12, Cnu=None, Ckappa=None, Cb=None):
        self.Cnu = C if Cnu is None else Cnu
        self.Ckappa = C if Ckappa is None else Ckappa
        self.Cb = Cb


class FPlane:
    def __init__(self, f):
        self.f = float(f)


class BuoyancyTracer:
    tracers = ("b",)


class SeawaterBuoyancy:
    """``SeawaterBuoyancy(equation_of_state=LinearEquationOfState(α, β))``."""
    tracers = ("T", "S")

    def __init__(self, gravitational_acceleration=9.80665, thermal_expansion=1.67e-4, haline_contraction=7.80e-4):
        self.g, self.alpha, self.beta = gravitational_acceleration, thermal_expansion, haline_contraction


class _BC:
    def __init__(self, kind, condition):
        self.kind, self.condition = kind, condition


def FluxBC(v):
    return _BC(L.BC_FLUX, v)


def ValueBC(v):
    return _BC(L.BC_VALUE, v)


def GradientBC(v):
    return _BC(L.BC_GRADIENT, v)


_SIDES = {"west": L.WEST, "east": L.EAST, "south": L.SOUTH, "north": L.NORTH, "bottom": L.BOTTOM, "top": L.TOP}


class Context:
    """One device + one in-order HIP stream (``Architectures.jl``: the `ROCmGPU()` singleton)."""

    def __init__(self, device=0):
        self.lib = L.load()
        self.h = C.c_void_p()
        check(self.lib.ocn_init(int(device), C.byref(self.h)))

    def sync(self):
        check(self.lib.ocn_sync(self.h), self.h)

    def close(self):
        if self.h:
            self.lib.ocn_destroy(self.h)
            self.h = C.c_void_p()

    def profile(self, on=True):
        check(self.lib.ocn_profile_enable(self.h, int(on)), self.h)

    def profile_filter(self, phase=None):
        check(self.lib.ocn_profile_filter(self.h, phase.encode() if phase else None), self.h)

    def profile_reset(self):
        check(self.lib.ocn_profile_reset(self.h), self.h)

    def copy_rate(self, nbytes=1 << 30, reps=20):
        """measured device-to-device copy rate [B/s, read + write]"""
        out = C.c_double()
        check(self.lib.ocn_measure_copy_rate(self.h, int(nbytes), int(reps), C.byref(out)), self.h)
        return out.value

    def profile_read(self, phase):
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.ocn_profile_read(self.h, phase.encode(), C.byref(ms), C.byref(n)), self.h)
        return ms.value, n.value


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class RectilinearGrid:
    """``RectilinearGrid(arch; size, extent | x,y,z, halo, topology)`` (Grids/rectilinear_grid.jl:249-279)."""

    def __init__(self, arch=None, size=None, extent=None, x=None, y=None, z=None, halo=None,
                 topology=(Periodic, Periodic, Bounded)):
        self.ctx = arch if isinstance(arch, Context) else default_context()
        topo = tuple(topology)
        nonflat = [t != Flat for t in topo]
        size = tuple(np.atleast_1d(size).tolist())
        if len(size) != sum(nonflat):
            raise ValueError("size must have one entry per non-Flat dimension")
        halo = tuple(3 for _ in size) if halo is None else tuple(np.atleast_1d(halo).tolist())
        if extent is not None:
            extent = tuple(np.atleast_1d(extent).tolist())
        d = L.GridDesc()
        it = 0
        given = [x, y, z]
        self._zfaces = None
        for a in range(3):
            d.topology[a] = _TOPO[topo[a]]
            if not nonflat[a]:
                d.N[a], d.H[a], d.x0[a], d.L[a] = 1, 0, 0.0, 1.0
                continue
            d.N[a], d.H[a] = int(size[it]), int(halo[it])
            if extent is not None:
                c = (0.0, extent[it]) if a < 2 else (-extent[it], 0.0)
            else:
                c = given[a]
                if c is None:
                    raise ValueError(f"Must supply extent or coordinate keyword for direction {a}")
            regular = isinstance(c, tuple) and len(c) == 2
            if callable(c):
                c = np.array([c(k) for k in range(1, size[it] + 2)], dtype=np.float64)
            c = np.asarray(c, dtype=np.float64)
            if regular:
                d.x0[a], d.L[a] = float(c[0]), float(c[1] - c[0])
            else:
                if a != 2:
                    raise OcnError("OCN_EUNSUPPORTED: stretched x / y axes are outside the path")
                if c.size != size[it] + 1:
                    raise ValueError("stretched axis needs N+1 faces")
                self._zfaces = np.ascontiguousarray(c)
                d.z_faces = self._zfaces.ctypes.data_as(C.POINTER(C.c_double))
                d.x0[a], d.L[a] = float(c[0]), float(c[-1] - c[0])
            it += 1
        d.rank, d.nranks = 0, 1
        self.desc = d
        self.topo = topo
        self.h = C.c_void_p()
        check(self.ctx.lib.ocn_grid_create(self.ctx.h, C.byref(d), C.byref(self.h)), self.ctx.h)
        self.Nx, self.Ny, self.Nz = (1 if topo[a] == Flat else d.N[a] for a in range(3))
        self.Lx, self.Ly, self.Lz = d.L[0], d.L[1], d.L[2]

    @property
    def N(self):
        return (self.Nx, self.Ny, self.Nz)


class FieldView:
    """A model field: parent array with halos on the device (Fields/field.jl:16-30)."""

    def __init__(self, model, fid, name):
        self.m, self.id, self.name = model, fid, name
        t, i, h = (C.c_int32 * 3)(), (C.c_int32 * 3)(), (C.c_int32 * 3)()
        check(model.lib.ocn_field_shape(model.h, fid, C.byref(t), C.byref(i), C.byref(h)), model.ctx.h)
        self.total, self.size, self.halo = tuple(t), tuple(i), tuple(h)

    def parent(self):
        a = np.zeros(self.total, dtype=np.float64, order="F")
        check(self.m.lib.ocn_field_download(self.m.h, self.id, a.ctypes.data_as(C.POINTER(C.c_double))), self.m.ctx.h)
        return a

    def set_parent(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        assert a.shape == self.total
        check(self.m.lib.ocn_field_upload(self.m.h, self.id, a.ctypes.data_as(C.POINTER(C.c_double))), self.m.ctx.h)

    def interior(self):
        a = np.zeros(self.size, dtype=np.float64, order="F")
        check(self.m.lib.ocn_field_get_interior(self.m.h, self.id, a.ctypes.data_as(C.POINTER(C.c_double))),
              self.m.ctx.h)
        return a

    def set(self, value):
        """``set!(field, value)`` (Fields/set!.jl): interior only, array / number / function of (x, y, z)."""
        if callable(value):
            X, Y, Z = self.m.nodes(self.name)
            value = value(X, Y, Z) + 0 * (X + Y + Z)
        a = np.empty(self.size, dtype=np.float64, order="F")
        a[...] = value
        check(self.m.lib.ocn_field_set_interior(self.m.h, self.id, a.ctypes.data_as(C.POINTER(C.c_double))),
              self.m.ctx.h)

    @property
    def device_ptr(self):
        return self.m.lib.ocn_field_device_ptr(self.m.h, self.id)

    @property
    def layout(self):
        """(element strides, element offset of the parent's first entry) of the device array."""
        st, org = (C.c_int64 * 3)(), C.c_int64()
        check(self.m.lib.ocn_field_layout(self.m.h, self.id, C.byref(st), C.byref(org)), self.m.ctx.h)
        return tuple(st), org.value


class Field:
    """``Field{LX, LY, LZ}(grid)`` / ``CenterField(grid)``: a stand-alone, zero-filled parent array on the device, halos
    included (Fields/field.jl:16-30, Grids/new_data.jl:16-61).  ``loc``: three of ``Center`` / ``Face``."""

    def __init__(self, loc, grid):
        self.grid, self.loc = grid, tuple(loc)
        self.lib, self.ctx = grid.ctx.lib, grid.ctx
        self.h = C.c_void_p()
        code = [L.FACE if l in (Face, "Face") else L.CENTER for l in self.loc]
        check(self.lib.ocn_field_create(grid.h, code[0], code[1], code[2], C.byref(self.h)), self.ctx.h)
        t, i, h = (C.c_int32 * 3)(), (C.c_int32 * 3)(), (C.c_int32 * 3)()
        check(self.lib.ocn_field_parent_shape(self.h, C.byref(t), C.byref(i), C.byref(h)), self.ctx.h)
        self.total, self.size, self.halo = tuple(t), tuple(i), tuple(h)

    def parent(self):
        a = np.zeros(self.total, dtype=np.float64, order="F")
        check(self.lib.ocn_field_parent_download(self.h, a.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.h)
        return a

    def set_parent(self, a):
        a = np.asfortranarray(a, dtype=np.float64)
        assert a.shape == self.total
        check(self.lib.ocn_field_parent_upload(self.h, a.ctypes.data_as(C.POINTER(C.c_double))), self.ctx.h)

    @property
    def device_ptr(self):
        return self.lib.ocn_field_parent_ptr(self.h)

    @property
    def layout(self):
        st, org = (C.c_int64 * 3)(), C.c_int64()
        check(self.lib.ocn_field_parent_layout(self.h, C.byref(st), C.byref(org)), self.ctx.h)
        return tuple(st), org.value

    def __del__(self):
        try:
            if self.h:
                self.lib.ocn_field_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


def CenterField(grid):
    return Field((Center, Center, Center), grid)


class NonhydrostaticModel:
    """``NonhydrostaticModel(; grid, advection, buoyancy, coriolis, closure, boundary_conditions, tracers,
    timestepper)`` (Models/NonhydrostaticModels/nonhydrostatic_model.jl:102-203)."""

    def __init__(self, grid, advection=None, buoyancy=None, coriolis=None, closure=None,
                 boundary_conditions=None, tracers=(), timestepper="QuasiAdamsBashforth2", chi=0.1):
        self.grid, self.ctx, self.lib = grid, grid.ctx, grid.ctx.lib
        if isinstance(tracers, str):
            tracers = (tracers,)
        self.tracer_names = tuple(tracers)
        if len(self.tracer_names) > L.MAX_TRACERS:
            raise ValueError("too many tracers")
        d = L.ModelDesc()
        d.advection = L.ADV_C2 if advection is None else advection.code
        if timestepper in ("QuasiAdamsBashforth2", "AB2"):
            d.stepper = L.STEPPER_AB2
        elif timestepper in ("RungeKutta3", "RK3"):
            d.stepper = L.STEPPER_RK3
        else:
            raise ValueError(f"unknown timestepper {timestepper}")
        d.chi = float(chi)
        d.n_tracers = len(self.tracer_names)
        if closure is None:
            d.closure = L.CLOSURE_NONE
        elif isinstance(closure, ScalarDiffusivity):
            d.closure, d.nu = L.CLOSURE_SCALAR, float(closure.nu)
            for i, n in enumerate(self.tracer_names):
                d.kappa[i] = float(closure.kappa[n] if isinstance(closure.kappa, dict) else closure.kappa)
        elif isinstance(closure, AnisotropicMinimumDissipation):
            d.closure, d.amd_Cnu = L.CLOSURE_AMD, float(closure.Cnu)
            if closure.Cb is not None:
                d.amd_Cb, d.amd_has_Cb = float(closure.Cb), 1
            for i, n in enumerate(self.tracer_names):
                d.amd_Ckappa[i] = float(closure.Ckappa[n] if isinstance(closure.Ckappa, dict) else closure.Ckappa)
        else:
            raise OcnError("OCN_EUNSUPPORTED: closure outside the path")
        if coriolis is not None:
            d.coriolis_fplane, d.f = 1, float(coriolis.f)
        d.b_index = d.T_index = d.S_index = -1
        if buoyancy is None:
            d.buoyancy = L.BUOYANCY_NONE
        else:
            # validate_buoyancy (nonhydrostatic_model.jl:135): required tracers must exist
            for n in buoyancy.tracers:
                if n not in self.tracer_names:
                    raise ValueError(f"buoyancy model requires tracer {n}")
            if isinstance(buoyancy, BuoyancyTracer):
                d.buoyancy, d.b_index = L.BUOYANCY_TRACER, self.tracer_names.index("b")
            else:
                d.buoyancy = L.BUOYANCY_LINEAR_TS
                d.T_index, d.S_index = self.tracer_names.index("T"), self.tracer_names.index("S")
                d.g, d.alpha, d.beta = buoyancy.g, buoyancy.alpha, buoyancy.beta
        self._keep = []
        names = ("u", "v", "w") + self.tracer_names
        def fill(slot, side, bc):
            b = slot[_SIDES[side]]
            b.kind = bc.kind
            if np.isscalar(bc.condition):
                b.value = float(bc.condition)
            else:
                arr = np.asfortranarray(bc.condition, dtype=np.float64)
                # arrays span the interior of the two directions tangential to the boundary
                tang = tuple(n for a, n in enumerate(grid.N) if a != _SIDES[side] // 2)
                if arr.shape != tang:
                    raise ValueError(f"array boundary condition on side {side} must have shape {tang}")
                self._keep.append(arr)
                b.array = arr.ctypes.data_as(C.POINTER(C.c_double))
        for fname, sides in (boundary_conditions or {}).items():
            if fname in ("nu_e", "kappa_e"):
                # boundary_conditions = (; κₑ = (; b = FieldBoundaryConditions(...))) of the reference: the AMD diffusivity fields
                if not isinstance(closure, AnisotropicMinimumDissipation):
                    raise ValueError(f"boundary conditions for {fname} need a closure with diffusivity fields")
                if fname == "nu_e":
                    for side, bc in sides.items():
                        fill(d.nu_bcs, side, bc)
                else:
                    for tname, tsides in sides.items():
                        for side, bc in tsides.items():
                            fill(d.kappa_bcs[self.tracer_names.index(tname)], side, bc)
                continue
            if fname not in names:
                raise ValueError(f"boundary conditions for unknown field {fname}")
            for side, bc in sides.items():
                fill(d.bcs[names.index(fname)], side, bc)
        self.desc = d
        self.h = C.c_void_p()
        check(self.lib.ocn_model_create(grid.h, C.byref(d), C.byref(self.h)), self.ctx.h)
        H = (C.c_int32 * 3)()
        check(self.lib.ocn_model_halo(self.h, C.byref(H)), self.ctx.h)
        self.halo = tuple(H)
        self.u = FieldView(self, L.F_U, "u")
        self.v = FieldView(self, L.F_V, "v")
        self.w = FieldView(self, L.F_W, "w")
        self.pNHS = FieldView(self, L.F_PNHS, "p")
        self.pHY = FieldView(self, L.F_PHY, "p") if grid.topo[2] != Flat else None
        self.tracers = {n: FieldView(self, L.F_TRACER + i, n) for i, n in enumerate(self.tracer_names)}
        self.nu_e = self.kappa_e = None
        if isinstance(closure, AnisotropicMinimumDissipation):
            self.nu_e = FieldView(self, L.F_NU, "nu")
            self.kappa_e = {n: FieldView(self, L.F_KAPPA + i, n) for i, n in enumerate(self.tracer_names)}
        self.Gn = {n: FieldView(self, L.F_GN + i, n) for i, n in enumerate(names)}
        self.Gm = {n: FieldView(self, L.F_GM + i, n) for i, n in enumerate(names)}

    def __del__(self):
        try:
            if self.h:
                self.lib.ocn_model_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    @property
    def kernel_path(self):
        """which kernels serve this model and, when it is not the fastest set, why"""
        buf = C.create_string_buffer(256)
        check(self.lib.ocn_model_path(self.h, buf, 256), self.ctx.h)
        return buf.value.decode()

    @property
    def graph_replays(self):
        """(steps replayed from a hipGraph so far, whether this model may use step graphs)"""
        n, act = C.c_int64(0), C.c_int32(0)
        check(self.lib.ocn_model_graph_replays(self.h, C.byref(n), C.byref(act)), self.ctx.h)
        return n.value, bool(act.value)

    def prognostic(self):
        d = {"u": self.u, "v": self.v, "w": self.w}
        d.update(self.tracers)
        return d

    # node coordinates of a prognostic field (for set! with functions); regular x,y; z from the grid
    def nodes(self, name):
        g = self.grid
        d = g.desc
        dx, dy = d.L[0] / g.Nx, d.L[1] / g.Ny
        xC = d.x0[0] + dx * (np.arange(g.Nx) + 0.5)
        yC = d.x0[1] + dy * (np.arange(g.Ny) + 0.5)
        xF = d.x0[0] + dx * np.arange(g.Nx + 1 if g.topo[0] == Bounded else g.Nx)
        yF = d.x0[1] + dy * np.arange(g.Ny + 1 if g.topo[1] == Bounded else g.Ny)
        if g.topo[0] == Flat:
            xF = xC = np.ones(1)
        if g.topo[1] == Flat:
            yF = yC = np.ones(1)
        if g.topo[2] == Flat:
            zF = zC = np.ones(1)
        elif g._zfaces is not None:
            zF = g._zfaces if g.topo[2] == Bounded else g._zfaces[:-1]
            zC = 0.5 * (g._zfaces[1:] + g._zfaces[:-1])
        else:
            dz = d.L[2] / g.Nz
            n = g.Nz + 1 if g.topo[2] == Bounded else g.Nz
            zF = d.x0[2] + dz * np.arange(n)
            zC = d.x0[2] + dz * (np.arange(g.Nz) + 0.5)
        X = (xF if name == "u" else xC).reshape(-1, 1, 1)
        Y = (yF if name == "v" else yC).reshape(1, -1, 1)
        Z = (zF if name == "w" else zC).reshape(1, 1, -1)
        return X, Y, Z

    @property
    def clock(self):
        t, it, st = C.c_double(), C.c_int64(), C.c_int32()
        check(self.lib.ocn_clock(self.h, C.byref(t), C.byref(it), C.byref(st)), self.ctx.h)
        return t.value, it.value, st.value

    @property
    def time(self):
        return self.clock[0]

    @property
    def iteration(self):
        return self.clock[1]

    def max_abs_divergence(self):
        out = C.c_double()
        check(self.lib.ocn_max_abs_divergence(self.h, C.byref(out)), self.ctx.h)
        return out.value

    def poisson_solve(self, rhs):
        """``solve!(ϕ, model.pressure_solver, rhs)`` with a host source term."""
        rhs = np.asfortranarray(rhs, dtype=np.float64)
        phi = np.zeros_like(rhs, order="F")
        PD = C.POINTER(C.c_double)
        check(self.lib.ocn_poisson_solve_host(self.h, rhs.ctypes.data_as(PD), phi.ctypes.data_as(PD)), self.ctx.h)
        return phi


def time_step(model, dt, euler=False):
    """``time_step!(model, Δt; euler=false)``."""
    check(model.lib.ocn_time_step(model.h, float(dt), int(bool(euler))), model.ctx.h)


def update_state(model):
    """``update_state!(model)``."""
    check(model.lib.ocn_update_state(model.h), model.ctx.h)


def set_model(model, enforce_incompressibility=True, **kw):
    """``set!(model; enforce_incompressibility=true, kwargs...)`` (set_nonhydrostatic_model.jl:32-59)."""
    pf = model.prognostic()
    for n, val in kw.items():
        if n not in pf:
            raise ValueError(f"name {n} not found in model.velocities or model.tracers.")
        pf[n].set(val)
    check(model.lib.ocn_set_epilogue(model.h, int(bool(enforce_incompressibility))), model.ctx.h)
