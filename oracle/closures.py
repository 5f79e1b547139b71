"""Turbulence closures on the hot path (oracle; test infrastructure only).

Restates ``TurbulenceClosures/closure_kernel_operators.jl:22-48`` (flux divergences),
``abstract_scalar_diffusivity_closure.jl:172-207`` (isotropic viscous / diffusive fluxes),
``velocity_tracer_gradients.jl`` (strain rates) and
``turbulence_closure_implementations/scalar_diffusivity.jl`` (constant nu, kappa).
"""
import numpy as np

from .grid import Center, Face
from .fields import Field


class ScalarDiffusivity:
    """Explicit, ThreeDimensionalFormulation ``ScalarDiffusivity(nu=, kappa=)``; kappa: number or {tracer: number}."""
    required_halo = 1   # scalar_diffusivity.jl:104

    def __init__(self, nu=0.0, kappa=0.0):
        self.nu, self.kappa = nu, kappa

    def kappa_of(self, name):
        if isinstance(self.kappa, dict):
            return self.kappa[name]
        return self.kappa


class AnisotropicMinimumDissipation:
    """anisotropic_minimum_dissipation.jl:51-59,110-119 (C = 1/12; Cb = nothing switches the buoyancy modification off)."""
    required_halo = 1

    def __init__(self, C=1 / 12, Cnu=None, Ckappa=None, Cb=None):
        self.Cnu = C if Cnu is None else Cnu
        self.Ckappa = C if Ckappa is None else Ckappa
        self.Cb = Cb


class Closure:
    def __init__(self, model, closure, boundary_conditions=None):
        self.m, self.c = model, closure
        self.nu_e, self.kappa_e = None, {}
        if isinstance(closure, AnisotropicMinimumDissipation):
            # DiffusivityFields(grid, tracer_names, bcs, closure) (anisotropic_minimum_dissipation.jl:346-370): user boundary
            # conditions for nu_e / kappa_e arrive as boundary_conditions = (; nu_e = ..., kappa_e = (; T = ...))
            g = model.grid
            bcs = boundary_conditions or {}
            kb = bcs.get("kappa_e") or {}
            self.nu_e = Field(g, (Center,) * 3, bcs.get("nu_e"))
            self.kappa_e = {n: Field(g, (Center,) * 3, kb.get(n)) for n in model.tracer_names}

    def diffusivity_fields(self):
        if self.nu_e is None:
            return []
        return [self.nu_e] + list(self.kappa_e.values())

    # ---- viscosity / diffusivity at the flux locations (closure_kernel_operators.jl:72-90) --------
    def _nu(self, where):
        c, o_ = self.c, self.m.ops
        if isinstance(c, ScalarDiffusivity):
            return lambda o: c.nu
        nu = self.nu_e
        if where == "ccc":
            return nu
        if where == "ffc":
            return o_.iF(1, o_.iF(0, nu))
        if where == "fcf":
            return o_.iF(2, o_.iF(0, nu))
        if where == "cff":
            return o_.iF(2, o_.iF(1, nu))
        raise ValueError(where)

    def _kappa(self, name, d):
        c, o_ = self.c, self.m.ops
        if isinstance(c, ScalarDiffusivity):
            k = c.kappa_of(name)
            return lambda o: k
        return o_.iF(d, self.kappa_e[name])

    # ---- strain rates (velocity_tracer_gradients.jl) -------------------------------------------------
    def _strain(self):
        m, o_ = self.m, self.m.ops
        u, v, w = m.u, m.v, m.w
        S11 = o_.ddC(0, u)
        S22 = o_.ddC(1, v)
        S33 = o_.ddC(2, w)
        S12 = lambda o: 0.5 * (o_.ddF(1, u)(o) + o_.ddF(0, v)(o))      # noqa: E731  at ffc
        S13 = lambda o: 0.5 * (o_.ddF(2, u)(o) + o_.ddF(0, w)(o))      # noqa: E731  at fcf
        S23 = lambda o: 0.5 * (o_.ddF(2, v)(o) + o_.ddF(1, w)(o))      # noqa: E731  at cff
        return S11, S22, S33, S12, S13, S23

    def div_tau(self, comp):
        """d_j tau_{comp j}  (closure_kernel_operators.jl:22-41); zero for ``closure = nothing``."""
        if self.c is None:
            return lambda o: 0.0
        o_ = self.m.ops
        S11, S22, S33, S12, S13, S23 = self._strain()
        Axc = lambda o: o_.Ax(Center, o)   # noqa: E731
        Ayc = lambda o: o_.Ay(Center, o)   # noqa: E731
        Axf = lambda o: o_.Ax(Face, o)     # noqa: E731
        Ayf = lambda o: o_.Ay(Face, o)     # noqa: E731
        Az = lambda o: o_.Az()             # noqa: E731

        def flux(area, nu, S):
            return lambda o: area(o) * (-2 * (nu(o) * S(o)))
        if comp == 0:
            fx = flux(Axc, self._nu("ccc"), S11)     # Ax_q^{ccc} viscous_flux_ux
            fy = flux(Ayc, self._nu("ffc"), S12)     # Ay_q^{ffc} viscous_flux_uy
            fz = flux(Az, self._nu("fcf"), S13)      # Az_q^{fcf} viscous_flux_uz
            return lambda o: 1 / o_.V(Center, o) * (o_.dF(0, fx)(o) + o_.dC(1, fy)(o) + o_.dC(2, fz)(o))
        if comp == 1:
            fx = flux(Axc, self._nu("ffc"), S12)
            fy = flux(Ayc, self._nu("ccc"), S22)
            fz = flux(Az, self._nu("cff"), S23)
            return lambda o: 1 / o_.V(Center, o) * (o_.dC(0, fx)(o) + o_.dF(1, fy)(o) + o_.dC(2, fz)(o))
        fx = flux(Axf, self._nu("fcf"), S13)
        fy = flux(Ayf, self._nu("cff"), S23)
        fz = flux(Az, self._nu("ccc"), S33)
        return lambda o: 1 / o_.V(Face, o) * (o_.dC(0, fx)(o) + o_.dC(1, fy)(o) + o_.dF(2, fz)(o))

    def div_q(self, name):
        """div q_c (closure_kernel_operators.jl:43-48; fluxes abstract_scalar_diffusivity_closure.jl:205-207)."""
        if self.c is None:
            return lambda o: 0.0
        o_ = self.m.ops
        c = self.m.tracers[name]
        areas = (lambda o: o_.Ax(Center, o), lambda o: o_.Ay(Center, o), lambda o: o_.Az())
        F = []
        for d in range(3):
            kap, grad = self._kappa(name, d), o_.ddF(d, c)
            F.append(lambda o, d=d, kap=kap, grad=grad: areas[d](o) * (-(kap(o)) * grad(o)))
        return lambda o: 1 / o_.V(Center, o) * (o_.dC(0, F[0])(o) + o_.dC(1, F[1])(o) + o_.dC(2, F[2])(o))

    def calculate_diffusivities(self):
        if isinstance(self.c, AnisotropicMinimumDissipation):
            from .amd import calculate_amd_diffusivities
            calculate_amd_diffusivities(self)
