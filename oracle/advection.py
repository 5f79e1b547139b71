"""Advection schemes and flux-form advection operators (oracle; test infrastructure only).

Restates, in offset-function form (see ``operators.py``):
  * ``Advection/weno_fifth_order.jl:12-19,240-317,368-403,476-524`` (WENO5, uniform coefficients;
    Z-weights by default ``:167``; the right-biased beta_0/beta_2 formulas are reproduced *as
    written* ``:315,317`` -- they are not the mirror image of the left-biased ones),
  * ``Advection/upwind_biased_fifth_order.jl:24-46`` (linear 5-point upwind),
  * ``Advection/centered_fourth_order.jl:17-33`` and ``centered_second_order.jl:16-32``,
  * ``Advection/upwind_biased_advective_fluxes.jl:10-128``, ``centered_advective_fluxes.jl:15-33``,
  * ``Advection/topologically_conditional_interpolation.jl:19-83`` (2nd-order fallback inside the
    boundary buffer of Bounded directions),
  * ``Advection/momentum_advection_operators.jl:52-86``, ``tracer_advection_operators.jl:31-35``.
"""
import numpy as np

from .grid import Bounded, Center, Face
from .operators import sh

C3_0, C3_1, C3_2 = 3 / 10, 3 / 5, 1 / 10      # weno_fifth_order.jl:12-14
EPS = 1e-6                                    # :19


class CenteredSecondOrder:
    buffer = 0
    kind = "C2"


class CenteredFourthOrder:
    buffer = 1
    kind = "centered"


class UpwindBiasedFifthOrder:
    buffer = 2
    kind = "upwind"


class UpwindBiasedFirstOrder:
    """upwind_biased_first_order.jl:6-35: boundary_buffer 1, two-point symmetric interpolation"""
    buffer = 1
    kind = "upwind"
    sym2 = True


class UpwindBiasedThirdOrder:
    """upwind_biased_third_order.jl:6-35"""
    buffer = 1
    kind = "upwind"
    sym2 = True


class WENO5:
    """``WENO5()`` with no grid: uniform coefficients everywhere (weno_fifth_order.jl:186-191)."""
    buffer = 2
    kind = "upwind"

    def __init__(self, zweno=True):
        self.zweno = zweno


class Advection:
    """All advective operators for one (grid, scheme)."""

    def __init__(self, ops, scheme):
        self.o, self.s = ops, scheme
        self.g = ops.g

    # ------------------------------------------------------------------ symmetric (4th order) ----
    def _i3F(self, d, f):
        """I3^f (centered_fourth_order.jl:17,20,23): f - delta^f(delta^c f)/6 for Face-located f."""
        o_ = self.o
        return lambda o: f(o) - o_.dF(d, o_.dC(d, f))(o) / 6

    def _i3C(self, d, f):
        o_ = self.o
        return lambda o: f(o) - o_.dC(d, o_.dF(d, f))(o) / 6

    def sym_C(self, d, f):
        """symmetric_interpolate_x^c: Face-located f -> Center (centered_fourth_order.jl:26,29,32)."""
        if self.s.kind == "C2" or getattr(self.s, "sym2", False):
            return self.o.iC(d, f)
        return self.o.iC(d, self._i3F(d, f))

    def sym_F(self, d, f):
        if self.s.kind == "C2" or getattr(self.s, "sym2", False):
            return self.o.iF(d, f)
        return self.o.iF(d, self._i3C(d, f))

    # ------------------------------------------------------------------ biased (face form) --------
    def _left_face(self, d, f):
        s = self.s
        if isinstance(s, UpwindBiasedFirstOrder):
            return lambda o: f(sh(o, d, -1))                                                   # c[i-1]
        if isinstance(s, UpwindBiasedThirdOrder):
            return lambda o: (2 * f(o) + 5 * f(sh(o, d, -1)) - f(sh(o, d, -2))) / 6             # upwind_biased_third_order.jl:21
        if isinstance(s, UpwindBiasedFifthOrder):
            return lambda o: (-3 * f(sh(o, d, 1)) + 27 * f(o) + 47 * f(sh(o, d, -1))
                              - 13 * f(sh(o, d, -2)) + 2 * f(sh(o, d, -3))) / 60

        def weno(o):
            a3, a2, a1, a0, b1 = (f(sh(o, d, -3)), f(sh(o, d, -2)), f(sh(o, d, -1)), f(o), f(sh(o, d, 1)))
            # stencils psi2=(i-3,i-2,i-1), psi1=(i-2,i-1,i), psi0=(i-1,i,i+1)   (:266-268)
            b0_ = 13 / 12 * (a1 - 2 * a0 + b1) ** 2 + 1 / 4 * (3 * a1 - 4 * a0 + b1) ** 2     # left beta0 :311
            b1_ = 13 / 12 * (a2 - 2 * a1 + a0) ** 2 + 1 / 4 * (a2 - a0) ** 2                   # :312
            b2_ = 13 / 12 * (a3 - 2 * a2 + a1) ** 2 + 1 / 4 * (a3 - 4 * a2 + 3 * a1) ** 2     # :313
            w0, w1, w2 = _weights(s, b0_, b1_, b2_, (C3_0, C3_1, C3_2))
            p0 = 1 / 3 * a1 + 5 / 6 * a0 - 1 / 6 * b1        # coeff_left_p0 :518
            p1 = -1 / 6 * a2 + 5 / 6 * a1 + 1 / 3 * a0       # :519
            p2 = 1 / 3 * a3 - 7 / 6 * a2 + 11 / 6 * a1       # :520
            return w0 * p0 + w1 * p1 + w2 * p2
        return weno

    def _right_face(self, d, f):
        s = self.s
        if isinstance(s, UpwindBiasedFirstOrder):
            return lambda o: f(o)                                                              # c[i]
        if isinstance(s, UpwindBiasedThirdOrder):
            return lambda o: (-f(sh(o, d, 1)) + 5 * f(o) + 2 * f(sh(o, d, -1))) / 6             # :29
        if isinstance(s, UpwindBiasedFifthOrder):
            return lambda o: (2 * f(sh(o, d, 2)) - 13 * f(sh(o, d, 1)) + 47 * f(o)
                              + 27 * f(sh(o, d, -1)) - 3 * f(sh(o, d, -2))) / 60

        def weno(o):
            a2, a1, a0, b1, b2 = (f(sh(o, d, -2)), f(sh(o, d, -1)), f(o), f(sh(o, d, 1)), f(sh(o, d, 2)))
            # stencils psi2=(i-2,i-1,i), psi1=(i-1,i,i+1), psi0=(i,i+1,i+2)     (:270-272)
            b0_ = 13 / 12 * (a0 - 2 * b1 + b2) ** 2 + 1 / 4 * (a0 - 4 * b1 + 3 * b2) ** 2     # right beta0 :315 (as written)
            b1_ = 13 / 12 * (a1 - 2 * a0 + b1) ** 2 + 1 / 4 * (a1 - b1) ** 2                   # :316
            b2_ = 13 / 12 * (a2 - 2 * a1 + a0) ** 2 + 1 / 4 * (3 * a2 - 4 * a1 + a0) ** 2     # :317 (as written)
            w0, w1, w2 = _weights(s, b0_, b1_, b2_, (C3_2, C3_1, C3_0))   # reversed optimal weights :368
            p0 = 11 / 6 * a0 - 7 / 6 * b1 + 1 / 3 * b2       # reverse(coeff_left_p2) :522
            p1 = 1 / 3 * a1 + 5 / 6 * a0 - 1 / 6 * b1        # reverse(coeff_left_p1)
            p2 = -1 / 6 * a2 + 5 / 6 * a1 + 1 / 3 * a0       # reverse(coeff_left_p0)
            return w0 * p0 + w1 * p1 + w2 * p2
        return weno

    # ------------------------------------------------------------------ conditional wrappers -----
    def _cond(self, d, kind, high, low):
        """topologically_conditional_interpolation.jl:19-21,46-79."""
        g, nb = self.g, self.s.buffer
        if g.topo[d] != Bounded:
            return high
        N = g.N[d]

        def f(o):
            i = self.o.index(d, o)
            if kind == "sym":
                outside = (i > nb) & (i < N + 1 - nb)
            elif kind == "left":
                outside = (i > nb) & (i < N + 1 - (nb - 1))
            else:
                outside = (i > nb - 1) & (i < N + 1 - nb)
            if np.all(outside):
                return high(o)
            return np.where(outside, high(o), low(o))
        return f

    def symC(self, d, f):   # _symmetric_interpolate_*^c
        return self._cond(d, "sym", self.sym_C(d, f), self.o.iC(d, f))

    def symF(self, d, f):
        return self._cond(d, "sym", self.sym_F(d, f), self.o.iF(d, f))

    def leftF(self, d, f):
        return self._cond(d, "left", self._left_face(d, f), self.o.iF(d, f))

    def rightF(self, d, f):
        return self._cond(d, "right", self._right_face(d, f), self.o.iF(d, f))

    def leftC(self, d, f):  # left_biased_interpolate_x^c(i) = face form at i+1 (weno :248-250, U5 :35-37)
        hf = self._left_face(d, f)
        return self._cond(d, "left", lambda o: hf(sh(o, d, 1)), self.o.iC(d, f))

    def rightC(self, d, f):
        hf = self._right_face(d, f)
        return self._cond(d, "right", lambda o: hf(sh(o, d, 1)), self.o.iC(d, f))

    # ------------------------------------------------------------------ fluxes ----------------------
    def _flux(self, area, adv_interp, d_q, q, q_at_center):
        """one advective momentum flux: area * (advecting velocity) * (reconstructed q)."""
        s = self.s
        if s.kind == "upwind":
            L = (self.leftC if q_at_center else self.leftF)(d_q, q)
            R = (self.rightC if q_at_center else self.rightF)(d_q, q)

            def f(o):
                ut = adv_interp(o)
                return area(o) * (((ut + np.abs(ut)) * L(o) + (ut - np.abs(ut)) * R(o)) / 2)  # :10
            return f
        S = (self.symC if q_at_center else self.symF)(d_q, q)
        return lambda o: area(o) * adv_interp(o) * S(o)

    def div_Uu(self, U, V, W, u):
        """div_vu (momentum_advection_operators.jl:52-56) at fcc."""
        o_ = self.o
        if self.s.kind == "C2":
            return self._div_c2(0, U, V, W, u)
        Fx = self._flux(lambda o: o_.Ax(Center, o), self.symC(0, U), 0, u, True)     # Uu at ccc
        Fy = self._flux(lambda o: o_.Ay(Center, o), self.symF(0, V), 1, u, False)    # Vu at ffc
        Fz = self._flux(lambda o: o_.Az(), self.symF(0, W), 2, u, False)             # Wu at fcf
        return lambda o: 1 / o_.V(Center, o) * (o_.dF(0, Fx)(o) + o_.dC(1, Fy)(o) + o_.dC(2, Fz)(o))

    def div_Uv(self, U, V, W, v):
        o_ = self.o
        if self.s.kind == "C2":
            return self._div_c2(1, U, V, W, v)
        Fx = self._flux(lambda o: o_.Ax(Center, o), self.symF(1, U), 0, v, False)    # Uv at ffc
        Fy = self._flux(lambda o: o_.Ay(Center, o), self.symC(1, V), 1, v, True)     # Vv at ccc
        Fz = self._flux(lambda o: o_.Az(), self.symF(1, W), 2, v, False)             # Wv at cff
        return lambda o: 1 / o_.V(Center, o) * (o_.dC(0, Fx)(o) + o_.dF(1, Fy)(o) + o_.dC(2, Fz)(o))

    def div_Uw(self, U, V, W, w):
        o_ = self.o
        if self.s.kind == "C2":
            return self._div_c2(2, U, V, W, w)
        Fx = self._flux(lambda o: o_.Ax(Face, o), self.symF(2, U), 0, w, False)      # Uw at fcf
        Fy = self._flux(lambda o: o_.Ay(Face, o), self.symF(2, V), 1, w, False)      # Vw at cff
        Fz = self._flux(lambda o: o_.Az(), self.symC(2, W), 2, w, True)              # Ww at ccc
        return lambda o: 1 / o_.V(Face, o) * (o_.dC(0, Fx)(o) + o_.dC(1, Fy)(o) + o_.dF(2, Fz)(o))

    def _div_c2(self, comp, U, V, W, q):
        """centered_second_order.jl:16-26: interpolated *area-weighted* velocities."""
        o_ = self.o
        AxU = lambda o: o_.Ax(Center, o) * U(o)   # noqa: E731
        AyV = lambda o: o_.Ay(Center, o) * V(o)   # noqa: E731
        AzW = lambda o: o_.Az() * W(o)            # noqa: E731
        if comp == 0:
            Fx = lambda o: o_.iC(0, AxU)(o) * o_.iC(0, q)(o)   # noqa: E731
            Fy = lambda o: o_.iF(0, AyV)(o) * o_.iF(1, q)(o)   # noqa: E731
            Fz = lambda o: o_.iF(0, AzW)(o) * o_.iF(2, q)(o)   # noqa: E731
            return lambda o: 1 / o_.V(Center, o) * (o_.dF(0, Fx)(o) + o_.dC(1, Fy)(o) + o_.dC(2, Fz)(o))
        if comp == 1:
            Fx = lambda o: o_.iF(1, AxU)(o) * o_.iF(0, q)(o)   # noqa: E731
            Fy = lambda o: o_.iC(1, AyV)(o) * o_.iC(1, q)(o)   # noqa: E731
            Fz = lambda o: o_.iF(1, AzW)(o) * o_.iF(2, q)(o)   # noqa: E731
            return lambda o: 1 / o_.V(Center, o) * (o_.dC(0, Fx)(o) + o_.dF(1, Fy)(o) + o_.dC(2, Fz)(o))
        Fx = lambda o: o_.iF(2, AxU)(o) * o_.iF(0, q)(o)       # noqa: E731
        Fy = lambda o: o_.iF(2, AyV)(o) * o_.iF(1, q)(o)       # noqa: E731
        Fz = lambda o: o_.iC(2, AzW)(o) * o_.iC(2, q)(o)       # noqa: E731
        return lambda o: 1 / o_.V(Face, o) * (o_.dC(0, Fx)(o) + o_.dC(1, Fy)(o) + o_.dF(2, Fz)(o))

    def div_Uc(self, U, V, W, c):
        """tracer_advection_operators.jl:31-35; fluxes upwind_biased_advective_fluxes.jl:103-128,
        centered_advective_fluxes.jl:31-33, centered_second_order.jl:30-32."""
        o_, s = self.o, self.s
        vel = (U, V, W)
        areas = (lambda o: o_.Ax(Center, o), lambda o: o_.Ay(Center, o), lambda o: o_.Az())
        F = []
        for d in range(3):
            if s.kind == "upwind":
                L, R = self.leftF(d, c), self.rightF(d, c)

                def f(o, d=d, L=L, R=R):
                    ut = vel[d](o)
                    return areas[d](o) * (((ut + np.abs(ut)) * L(o) + (ut - np.abs(ut)) * R(o)) / 2)
            elif s.kind == "C2":
                def f(o, d=d):
                    return areas[d](o) * vel[d](o) * o_.iF(d, c)(o)
            else:
                S = self.symF(d, c)

                def f(o, d=d, S=S):
                    return areas[d](o) * vel[d](o) * S(o)
            F.append(f)
        return lambda o: 1 / o_.V(Center, o) * (o_.dC(0, F[0])(o) + o_.dC(1, F[1])(o) + o_.dC(2, F[2])(o))


def _weights(s, b0, b1, b2, C):
    """weno_fifth_order.jl:380-403."""
    if s.zweno:
        tau = np.abs(b2 - b0)
        a0 = C[0] * (1 + (tau / (b0 + EPS)) ** 2)
        a1 = C[1] * (1 + (tau / (b1 + EPS)) ** 2)
        a2 = C[2] * (1 + (tau / (b2 + EPS)) ** 2)
    else:
        a0 = C[0] / (b0 + EPS) ** 2
        a1 = C[1] / (b1 + EPS) ** 2
        a2 = C[2] / (b2 + EPS) ** 2
    sa = a0 + a1 + a2
    return a0 / sa, a1 / sa, a2 / sa
