"""AnisotropicMinimumDissipation eddy viscosity / diffusivities (oracle; test infrastructure only).

Restates ``TurbulenceClosures/turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:138-178``
(predictors), ``:213-226`` (filter widths: twice the *centre* spacing evaluated at the calling index, at
every location), ``:229-340`` ("the 30 terms") and ``velocity_tracer_gradients.jl:126-250`` (normalised
gradients; note ``norm_dx_u = dx_u`` etc. are NOT normalised, and ``cy_uy`` uses ``I_xz`` on ``norm_dy_w``
``:326`` -- reproduced as written).  The buoyancy modification ``Cb`` follows ``:142-154,297-312``.
"""
import numpy as np

from .grid import Center, Face

Z3 = (0, 0, 0)


def calculate_amd_diffusivities(cl):
    m, o_, c = cl.m, cl.m.ops, cl.c
    g = m.grid
    u, v, w = m.u, m.v, m.w
    sq = lambda f: (lambda o: f(o) ** 2)                    # noqa: E731
    mul = lambda f, h: (lambda o: f(o) * h(o))              # noqa: E731

    # filter widths (always the ccc spacings at the calling index)
    Dx = lambda o: 2 * g.dx                                 # noqa: E731
    Dy = lambda o: 2 * g.dy                                 # noqa: E731
    Dz = lambda o: 2 * o_.dz(Center, o)                     # noqa: E731

    # plain gradients (velocity_tracer_gradients.jl:1-20)
    dxu, dyv, dzw = o_.ddC(0, u), o_.ddC(1, v), o_.ddC(2, w)
    dxv, dyu = o_.ddF(0, v), o_.ddF(1, u)                   # ffc
    dxw, dzu = o_.ddF(0, w), o_.ddF(2, u)                   # fcf
    dyw, dzv = o_.ddF(1, w), o_.ddF(2, v)                   # cff
    # normalised (:126-143)
    n_dxu, n_dyv, n_dzw = dxu, dyv, dzw
    n_dxv = lambda o: Dx(o) / Dy(o) * dxv(o)                # noqa: E731
    n_dyu = lambda o: Dy(o) / Dx(o) * dyu(o)                # noqa: E731
    n_dxw = lambda o: Dx(o) / Dz(o) * dxw(o)                # noqa: E731
    n_dzu = lambda o: Dz(o) / Dx(o) * dzu(o)                # noqa: E731
    n_dyw = lambda o: Dy(o) / Dz(o) * dyw(o)                # noqa: E731
    n_dzv = lambda o: Dz(o) / Dy(o) * dzv(o)                # noqa: E731
    S11, S22, S33 = n_dxu, n_dyv, n_dzw
    S12 = lambda o: 0.5 * (n_dyu(o) + n_dxv(o))             # noqa: E731
    S13 = lambda o: 0.5 * (n_dzu(o) + n_dxw(o))             # noqa: E731
    S23 = lambda o: 0.5 * (n_dzv(o) + n_dyw(o))             # noqa: E731
    # double interpolations to ccc
    Ixy = lambda f: o_.iC(1, o_.iC(0, f))                   # noqa: E731  ffc -> ccc
    Ixz = lambda f: o_.iC(2, o_.iC(0, f))                   # noqa: E731  fcf -> ccc
    Iyz = lambda f: o_.iC(2, o_.iC(1, f))                   # noqa: E731  cff -> ccc

    def r_term(o):   # norm_u_ia u_ja Sigma_ij  (:229-276)
        a = (S11(o) * n_dxu(o) ** 2
             + S22(o) * Ixy(sq(n_dxv))(o)
             + S33(o) * Ixz(sq(n_dxw))(o)
             + 2 * n_dxu(o) * Ixy(mul(n_dxv, S12))(o)
             + 2 * n_dxu(o) * Ixz(mul(n_dxw, S13))(o)
             + 2 * Ixy(n_dxv)(o) * Ixz(n_dxw)(o) * Iyz(S23)(o))
        b = (S11(o) * Ixy(sq(n_dyu))(o)
             + S22(o) * n_dyv(o) ** 2
             + S33(o) * Iyz(sq(n_dyw))(o)
             + 2 * n_dyv(o) * Ixy(mul(n_dyu, S12))(o)
             + 2 * Ixy(n_dyu)(o) * Iyz(n_dyw)(o) * Ixz(S13)(o)
             + 2 * n_dyv(o) * Iyz(mul(n_dyw, S23))(o))
        cc = (S11(o) * Ixz(sq(n_dzu))(o)
              + S22(o) * Iyz(sq(n_dzv))(o)
              + S33(o) * n_dzw(o) ** 2
              + 2 * Ixz(n_dzu)(o) * Iyz(n_dzv)(o) * Ixy(S12)(o)
              + 2 * n_dzw(o) * Ixz(mul(n_dzu, S13))(o)
              + 2 * n_dzw(o) * Iyz(mul(n_dzv, S23))(o))
        return a + b + cc

    def q_term(o):   # norm_tr_grad_u (:282-304): UN-normalised gradients squared
        return (dxu(o) ** 2 + dyv(o) ** 2 + dzw(o) ** 2
                + Ixy(sq(n_dxv))(o) + Ixy(sq(n_dyu))(o)
                + Ixz(sq(n_dxw))(o) + Ixz(sq(n_dzu))(o)
                + Iyz(sq(n_dyw))(o) + Iyz(sq(n_dzv))(o))

    d2 = 3 / (1 / Dx(Z3) ** 2 + 1 / Dy(Z3) ** 2 + 1 / Dz(Z3) ** 2)
    q = q_term(Z3)
    r = r_term(Z3)
    Cb_zeta = 0.0
    if getattr(c, "Cb", None) is not None and m.buoyancy is not None:
        # Cb_norm_w_i_b_i (:299-312) / Delta_z: the buoyancy perturbation's gradients interpolated to ccc, against the
        # w gradients normalised as above
        bfun = m.buoyancy.perturbation(m.tracers)   # offset -> array, like every other operand here
        o = Z3
        wx_bx = Ixz(n_dxw)(o) * Dx(o) * o_.iC(0, o_.ddF(0, bfun))(o)
        wy_by = Iyz(n_dyw)(o) * Dy(o) * o_.iC(1, o_.ddF(1, bfun))(o)
        wz_bz = n_dzw(o) * Dz(o) * o_.iC(2, o_.ddF(2, bfun))(o)
        Cb_zeta = c.Cb * (wx_bx + wy_by + wz_bz) / Dz(o)
    with np.errstate(divide="ignore", invalid="ignore"):
        nu = np.where(q == 0, 0.0, -c.Cnu * d2 * (r - Cb_zeta) / q)
    cl.nu_e()[...] = np.maximum(0.0, nu)

    for n, cf in m.tracers.items():
        Ck = c.Ckappa[n] if isinstance(c.Ckappa, dict) else c.Ckappa
        n_dxc = lambda o, cf=cf: Dx(o) * o_.ddF(0, cf)(o)   # noqa: E731  fcc
        n_dyc = lambda o, cf=cf: Dy(o) * o_.ddF(1, cf)(o)   # noqa: E731  cfc
        n_dzc = lambda o, cf=cf: Dz(o) * o_.ddF(2, cf)(o)   # noqa: E731  ccf
        Ix, Iy, Iz = (lambda f: o_.iC(0, f)), (lambda f: o_.iC(1, f)), (lambda f: o_.iC(2, f))
        o = Z3
        sigma = Ix(sq(n_dxc))(o) + Iy(sq(n_dyc))(o) + Iz(sq(n_dzc))(o)        # norm_theta_i^2 (:342-344)
        cx = (n_dxu(o) * Ix(sq(n_dxc))(o)
              + Ixy(n_dxv)(o) * Ix(n_dxc)(o) * Iy(n_dyc)(o)
              + Ixz(n_dxw)(o) * Ix(n_dxc)(o) * Iz(n_dzc)(o))
        cy = (Ixy(n_dyu)(o) * Iy(n_dyc)(o) * Ix(n_dxc)(o)
              + n_dyv(o) * Iy(sq(n_dyc))(o)
              + Ixz(n_dyw)(o) * Iy(n_dyc)(o) * Iz(n_dzc)(o))                  # as written: I_xz on a cff quantity (:326)
        cz = (Ixz(n_dzu)(o) * Iz(n_dzc)(o) * Ix(n_dxc)(o)
              + Iyz(n_dzv)(o) * Iz(n_dzc)(o) * Iy(n_dyc)(o)
              + n_dzw(o) * Iz(sq(n_dzc))(o))
        theta = cx + cy + cz
        with np.errstate(divide="ignore", invalid="ignore"):
            kap = np.where(sigma == 0, 0.0, -Ck * d2 * theta / sigma)
        cl.kappa_e[n]()[...] = np.maximum(0.0, kap)
