"""Shared parity cases: the same seeded configuration is run through the oracle and through the C ABI
(HIP on the GPU box, host emulation of the same kernels elsewhere) and compared field by field.

Tolerances (Float64): one tendency evaluation 1e-12 relative to max|G|;
trajectories rtol sqrt(eps) with atol 1e-12 max|field| -- in practice we assert much tighter bounds.
"""
import numpy as np

import oracle as O

P, B, F = "Periodic", "Bounded", "Flat"


def _adv(mod, name):
    return {None: None, "WENO5": mod.WENO5(), "WENO5JS": mod.WENO5(zweno=False), "U5": mod.UpwindBiasedFifthOrder(),
            "U1": mod.UpwindBiasedFirstOrder(), "U3": mod.UpwindBiasedThirdOrder(),
            "C4": mod.CenteredFourthOrder(), "C2": mod.CenteredSecondOrder()}[name]


CASES = {
    # headline configuration, small
    "ppp_weno_ab2": dict(size=(16, 12, 10), topo=(P, P, P), extent=(1, 2, 1.5), adv="WENO5", stepper="AB2", steps=3, dt=2e-3),
    "ppp_weno_rk3": dict(size=(12, 16, 8), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2, dt=4e-3),
    "ppp_wenojs_tracer": dict(size=(8, 8, 8), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5JS", stepper="AB2", steps=2,
                              dt=2e-3, tracers=("c",)),
    "ppp_weno_rk3_2tracers": dict(size=(10, 8, 12), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2,
                                  dt=3e-3, tracers=("a", "b2")),
    "ppp_u5_visc": dict(size=(10, 9, 8), topo=(P, P, P), extent=(1, 1, 1), adv="U5", stepper="AB2", steps=2, dt=2e-3,
                        tracers=("c",), closure=(1e-2, 2e-2)),
    # UpwindBiasedFirstOrder / ThirdOrder (boundary buffer 1, halo 2): general kernels
    "ppp_u3_tracer": dict(size=(9, 8, 10), topo=(P, P, P), extent=(1, 1, 1), adv="U3", stepper="AB2", steps=2, dt=2e-3,
                          tracers=("c",), halo=(2, 2, 2)),
    "ppb_u1_btracer": dict(size=(8, 9, 7), topo=(P, P, B), extent=(1, 1, 1), adv="U1", stepper="RK3", steps=2, dt=3e-3,
                           tracers=("b",), buoyancy="b", halo=(2, 2, 2)),
    "bbb_u3_walls": dict(size=(8, 7, 9), topo=(B, B, B), extent=(1, 1, 1), adv="U3", stepper="AB2", steps=2, dt=3e-3,
                         tracers=("c",), closure=(1e-3, 1e-3), halo=(2, 2, 2)),
    "ppp_c4": dict(size=(8, 10, 12), topo=(P, P, P), extent=(1, 1, 1), adv="C4", stepper="RK3", steps=2, dt=2e-3),
    "ppp_c2_default": dict(size=(8, 8, 8), topo=(P, P, P), extent=(1, 1, 1), adv="C2", stepper="AB2", steps=2, dt=2e-3,
                           halo=(1, 1, 1)),
    # DNS-style: triply periodic with ScalarDiffusivity stays on the fused path (viscous / diffusive face fluxes)
    "ppp_weno_visc_ab2": dict(size=(12, 10, 9), topo=(P, P, P), extent=(1, 1.2, 0.9), adv="WENO5", stepper="AB2", steps=3,
                              dt=2e-3, tracers=("c",), closure=(2e-2, 3e-2)),
    "ppp_weno_visc_rk3_noproj": dict(size=(10, 12, 8), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2,
                                     dt=2e-3, tracers=("a", "b2"), closure=(5e-2, 1e-2), project_init=False),
    # triply periodic with terms the all-in-one fused path does not carry (AMD, Coriolis): the tiled kernel then adds
    # advection on top of the general kernels' other terms, as on bounded-z grids
    "ppp_weno_amd_coriolis": dict(size=(10, 8, 9), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2,
                                  dt=2e-3, tracers=("c",), closure="amd", coriolis=2e-1),
    "ppp_u5_coriolis_ab2": dict(size=(8, 10, 8), topo=(P, P, P), extent=(1, 1, 1), adv="U5", stepper="AB2", steps=3, dt=2e-3,
                                coriolis=1e-1),
    # adaptive time stepping (TimeStepWizard): every change of dt is an Euler step with G^- zeroed
    # (quasi_adams_bashforth_2.jl:74-84), on the fused path (pointer-rotated G buffers) and on the general one
    "ppp_weno_ab2_varying_dt": dict(size=(12, 10, 8), topo=(P, P, P), extent=(1, 1, 1), adv="WENO5", stepper="AB2", steps=5,
                                    dt=2e-3, dts=[2e-3, 2e-3, 1e-3, 1e-3, 3e-3], tracers=("c",)),
    "ppb_c4_ab2_varying_dt": dict(size=(8, 8, 8), topo=(P, P, B), extent=(1, 1, 1), adv="C4", stepper="AB2", steps=5, dt=2e-3,
                                  dts=[2e-3, 1e-3, 1e-3, 1e-3, 2e-3], tracers=("b",), buoyancy="b", closure=(1e-3, 1e-3)),
    # bounded z, regular (reference: FFT + cosine transform; here: Fourier-tridiagonal, same discrete system)
    "ppb_weno_full": dict(size=(12, 10, 9), topo=(P, P, B), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2, dt=2e-3,
                          tracers=("T", "S"), closure=(1e-3, 2e-3), coriolis=1e-1, buoyancy="TS"),
    "ppb_c2_btracer": dict(size=(8, 8, 8), topo=(P, P, B), extent=(1, 1, 1), adv="C2", stepper="AB2", steps=3, dt=5e-3,
                           tracers=("b",), buoyancy="b"),
    # bounded, stretched z + flux / gradient boundary conditions (ocean_wind_mixing-style)
    "ppb_stretched_bcs": dict(size=(8, 8, 8), topo=(P, P, B), xy=((0, 1), (0, 1)),
                              zfaces=[-1.0, -0.8, -0.62, -0.46, -0.32, -0.2, -0.1, -0.04, 0.0],
                              adv="WENO5", stepper="RK3", steps=2, dt=2e-3, tracers=("T", "S"), closure=(1e-3, 1e-3),
                              coriolis=1e-2, buoyancy="TS",
                              bcs={"u": {"top": ("flux", -1e-2)}, "T": {"top": ("flux", 2e-3), "bottom": ("gradient", 0.01)},
                                   "S": {"top": ("flux", -1e-3)}}),
    "ppb_stretched_c2": dict(size=(6, 7, 8), topo=(P, P, B), xy=((0, 1), (0, 1)),
                             zfaces=[0, 1, 2, 4, 7, 11, 16, 22, 29], adv="C2", stepper="AB2", steps=2, dt=1e-2,
                             tracers=("b",), buoyancy="b", closure=(1e-2, 1e-2)),
    # no-slip bottom, wind-stress top, regular z: the bounded-z tiled kernel next to value conditions
    "ppb_weno_noslip": dict(size=(10, 9, 8), topo=(P, P, B), extent=(1, 1, 0.5), adv="WENO5", stepper="AB2", steps=3, dt=2e-3,
                            tracers=("b",), buoyancy="b", closure=(1e-2, 1e-2), coriolis=1e-1,
                            bcs={"u": {"bottom": ("value", 0.0), "top": ("flux", -1e-2)}, "v": {"bottom": ("value", 0.0)},
                                 "b": {"top": ("value", 1.0), "bottom": ("gradient", 0.3)}}),
    # BASELINE config 3 in miniature: stretched bounded z, T/S, FPlane, linear EOS, AMD, flux/gradient BCs, WENO5, RK3
    "ppb_amd_config3": dict(size=(8, 8, 8), topo=(P, P, B), xy=((0, 2), (0, 2)),
                            zfaces=[-1.0, -0.8, -0.62, -0.46, -0.32, -0.2, -0.1, -0.04, 0.0],
                            adv="WENO5", stepper="RK3", steps=2, dt=2e-3, tracers=("T", "S"), closure="amd",
                            coriolis=1e-2, buoyancy="TS",
                            bcs={"u": {"top": ("flux", -1e-2)}, "T": {"top": ("flux", 2e-3), "bottom": ("gradient", 0.01)},
                                 "S": {"top": ("flux", -1e-3)}}),
    # AMD with the buoyancy modification term (Cb = 1 of Abkar et al.; anisotropic_minimum_dissipation.jl:142-154,299-312)
    "ppb_amd_cb_seawater": dict(size=(8, 7, 8), topo=(P, P, B), xy=((0, 2), (0, 2)),
                                zfaces=[-1.0, -0.8, -0.62, -0.46, -0.32, -0.2, -0.1, -0.04, 0.0],
                                adv="WENO5", stepper="RK3", steps=2, dt=2e-3, tracers=("T", "S"), closure="amd", amd_Cb=1.0,
                                coriolis=1e-2, buoyancy="TS",
                                bcs={"T": {"top": ("flux", 2e-3), "bottom": ("gradient", 0.01)}}),
    "ppb_amd_cb_btracer": dict(size=(8, 6, 7), topo=(P, P, B), extent=(1, 1, 1), adv="C2", stepper="AB2", steps=2,
                               dt=2e-3, tracers=("b",), closure="amd", amd_Cb=0.7, buoyancy="b"),
    "ppb_amd_regular_c2": dict(size=(8, 6, 7), topo=(P, P, B), extent=(1, 1, 1), adv="C2", stepper="AB2", steps=2,
                               dt=2e-3, tracers=("b",), closure="amd", buoyancy="b"),
    # Bounded / Flat x and y: walls, cosine-transform topologies, 2-D slices
    "bbb_weno_walls": dict(size=(8, 7, 6), topo=(B, B, B), extent=(1, 1.5, 0.8), adv="WENO5", stepper="RK3", steps=2, dt=2e-3,
                           tracers=("c",), closure=(1e-2, 2e-2),
                           bcs={"u": {"south": ("value", 0.0), "north": ("value", 0.1), "bottom": ("value", 0.0), "top": ("value", 0.0)},
                                "v": {"west": ("value", 0.0), "east": ("value", 0.0), "bottom": ("value", 0.0), "top": ("gradient", 0.2)},
                                "w": {"west": ("value", 0.0), "east": ("value", 0.0), "south": ("flux", 1e-3), "north": ("value", 0.0)},
                                "c": {"west": ("value", 1.0), "east": ("gradient", -0.5), "south": ("flux", "rand:7:1e-2"),
                                      "top": ("flux", "rand:8:1e-2"), "bottom": ("value", "rand:9:1.0")}}),
    "pbb_u5_channel": dict(size=(8, 8, 8), topo=(P, B, B), extent=(2, 1, 1), adv="U5", stepper="AB2", steps=3, dt=2e-3,
                           tracers=("b",), buoyancy="b", coriolis=5e-2, closure=(1e-3, 1e-3),
                           bcs={"u": {"south": ("value", 0.0), "north": ("value", 0.0)}}),
    "bpp_c4": dict(size=(9, 8, 6), topo=(B, P, P), extent=(1, 1, 1), adv="C4", stepper="RK3", steps=2, dt=2e-3),
    "bpb_c2_stretched": dict(size=(6, 5, 8), topo=(B, P, B), xy=((0, 1), (-1, 1)), zfaces=[0, 1, 2, 4, 7, 11, 16, 22, 29],
                             adv="C2", stepper="AB2", steps=2, dt=1e-2, tracers=("b",), buoyancy="b", closure=(1e-2, 1e-2)),
    "pbb_amd": dict(size=(8, 6, 7), topo=(P, B, B), extent=(1, 1, 1), adv="WENO5", stepper="RK3", steps=2, dt=2e-3,
                    tracers=("b",), closure="amd", buoyancy="b"),
    "bfb_weno_slice": dict(size=(12, 10), topo=(B, F, B), extent=(2, 1), adv="WENO5", stepper="RK3", steps=2, dt=2e-3,
                           tracers=("b",), buoyancy="b", closure=(1e-3, 1e-3)),
    "fpb_u5_slice": dict(size=(10, 9), topo=(F, P, B), extent=(1, 1), adv="U5", stepper="AB2", steps=2, dt=2e-3,
                         tracers=("c",)),
    "pfp_weno_slice": dict(size=(12, 8), topo=(P, F, P), extent=(1, 1), adv="WENO5", stepper="AB2", steps=2, dt=2e-3),
    "bbf_weno_box": dict(size=(10, 12), topo=(B, B, F), extent=(1, 1), adv="WENO5", stepper="RK3", steps=2, dt=5e-3,
                         closure=(1e-4, 0.0)),
    # The reference's regression-test configurations (test/regression_tests/*.jl).  Their stored fields are remote
    # DataDeps (test/data_dependencies.jl) and cannot be fetched here; the same set-ups pin HIP against the oracle.
    "regr_thermal_bubble_regular": dict(size=(16, 16, 16), topo=(P, P, B), extent=(100, 100, 100), halo=(1, 1, 1), adv="C2",
                                        stepper="AB2", steps=10, dt=6.0, tracers=("T", "S"), closure=(4e-2, 4e-2),
                                        coriolis=1e-4, buoyancy="TS", eos=(1.67e-4, 7.8e-4), init="thermal_bubble",
                                        # pHY' carries the O(25) hydrostatic load of T = 9.85, S = 35 while the dynamics
                                        # come from a 0.01 K anomaly: tendencies are differences of nearly equal numbers
                                        tol=5e-10),
    "regr_thermal_bubble_unstretched": dict(size=(16, 16, 16), topo=(P, P, B), xy=((0, 100), (0, 100)),
                                            zfaces=list(np.linspace(-100, 0, 17)), halo=(1, 1, 1), adv="C2", stepper="AB2",
                                            steps=10, dt=6.0, tracers=("T", "S"), closure=(4e-2, 4e-2), coriolis=1e-4,
                                            buoyancy="TS", eos=(1.67e-4, 7.8e-4), init="thermal_bubble", tol=5e-10),
    "regr_ocean_les_amd": dict(size=(16, 16, 16), topo=(P, P, B), extent=(16, 16, 16), halo=(1, 1, 1), adv="C2", stepper="AB2",
                               steps=10, dt=2.0, tracers=("T", "S"), closure="amd", coriolis=1e-4, buoyancy="TS",
                               eos=(2e-4, 8e-4), init="ocean_les",
                               # S starts uniform: kappa_e(S) = -C d2 theta/sigma is then a ratio of round-off sized
                               # gradients wherever the surface flux has not reached, i.e. ill-conditioned in any
                               # implementation (the reference compares this run with isapprox as well)
                               tol=2e-9,
                               bcs={"u": {"top": ("flux", -2e-5)}, "T": {"top": ("flux", 5e-5), "bottom": ("gradient", 0.005)},
                                    "S": {"top": ("flux", 5e-8)}}),
    # columns of 64 / 96 levels: the segmented hydrostatic-pressure kernel (8 segments per column; kernels.hip k_hydrostatic_seg)
    "ppb_weno_tall64": dict(size=(8, 6, 64), topo=(P, P, B), xy=((0, 1), (0, 1)), zfaces=list(-np.linspace(1.0, 0.0, 65) ** 1.5),
                            adv="WENO5", stepper="RK3", steps=1, dt=1e-3, tracers=("T", "S"), closure=(1e-3, 1e-3), coriolis=1e-2,
                            buoyancy="TS", bcs={"T": {"top": ("flux", 2e-3), "bottom": ("gradient", 0.01)}}),
    "ppb_c2_tall96_btracer": dict(size=(6, 8, 96), topo=(P, P, B), extent=(1, 1, 2), adv="C2", stepper="AB2", steps=2, dt=2e-3,
                                  tracers=("b",), buoyancy="b", halo=(1, 1, 1)),
    # two-dimensional turbulence (BASELINE config 1): Flat z
    "ppf_weno_rk3": dict(size=(16, 16), topo=(P, P, F), extent=(2 * np.pi, 2 * np.pi), adv="WENO5", stepper="RK3",
                         steps=2, dt=0.05, closure=(1e-5, 0.0)),
}


def _nodes(m, name, mod):
    if mod is O:
        f = m.w if name == "w" else m.tracers[name]
        g = m.grid
        return g.xnodes(f.loc[0]).reshape(-1, 1, 1), g.ynodes(f.loc[1]).reshape(1, -1, 1), g.znodes(f.loc[2]).reshape(1, 1, -1)
    return m.nodes(name)


def _full_size(cfg):
    it = iter(cfg["size"])
    return [1 if t == F else next(it) for t in cfg["topo"]]


def build(mod, cfg, rng_seed=1234):
    """build + initialise a model with module `mod` (oracle or the ocnhip package)."""
    kw = {}
    if "extent" in cfg:
        kw["extent"] = cfg["extent"]
    else:
        kw["x"], kw["y"] = cfg["xy"]
        kw["z"] = np.array(cfg["zfaces"], dtype=float)
    if "halo" in cfg:
        kw["halo"] = cfg["halo"]
    g = mod.RectilinearGrid(size=cfg["size"], topology=cfg["topo"], **kw)
    mk = {}
    if cfg.get("closure") == "amd":
        mk["closure"] = mod.AnisotropicMinimumDissipation(**({"Cb": cfg["amd_Cb"]} if "amd_Cb" in cfg else {}))
    elif cfg.get("closure"):
        mk["closure"] = mod.ScalarDiffusivity(nu=cfg["closure"][0], kappa=cfg["closure"][1])
    if cfg.get("coriolis"):
        mk["coriolis"] = mod.FPlane(cfg["coriolis"])
    if cfg.get("buoyancy") == "TS":
        a_, b_ = cfg.get("eos", (2e-1, 8e-1))
        mk["buoyancy"] = mod.SeawaterBuoyancy(thermal_expansion=a_, haline_contraction=b_)
    elif cfg.get("buoyancy") == "b":
        mk["buoyancy"] = mod.BuoyancyTracer()
    if cfg.get("bcs"):
        ctor = {"flux": mod.FluxBC, "value": mod.ValueBC, "gradient": mod.GradientBC}
        dim_of = {"west": 0, "east": 0, "south": 1, "north": 1, "bottom": 2, "top": 2}
        gN = [1 if t == F else n for t, n in zip(cfg["topo"], _full_size(cfg))]

        def cond(side, v):
            if isinstance(v, str):     # "rand:<seed>:<amplitude>": array over the two tangential directions
                _, seed, amp = v.split(":")
                shp = tuple(n for a, n in enumerate(gN) if a != dim_of[side])
                return float(amp) * (np.random.default_rng(int(seed)).random(shp) - 0.5)
            return v
        mk["boundary_conditions"] = {f: {s: ctor[k](cond(s, v)) for s, (k, v) in sides.items()}
                                     for f, sides in cfg["bcs"].items()}
    m = mod.NonhydrostaticModel(g, advection=_adv(mod, cfg["adv"]), timestepper=cfg["stepper"],
                                tracers=cfg.get("tracers", ()), **mk)
    rng = np.random.default_rng(rng_seed)
    init = {}
    if cfg.get("init") == "thermal_bubble":      # thermal_bubble_regression_test.jl:17-27
        Nx, Ny, Nz = cfg["size"]
        T = np.full((Nx, Ny, Nz), 9.85)
        T[round(Nx / 4) - 1:round(3 * Nx / 4), round(Ny / 4) - 1:round(3 * Ny / 4), round(Nz / 4) - 1:round(3 * Nz / 4)] += 0.01
        mod.set_model(m, T=T, S=35.0)
        return m
    if cfg.get("init") == "ocean_les":           # ocean_large_eddy_simulation_regression_test.jl:52-56
        Lz, dTdz, Qu = 16.0, 0.005, -2e-5
        xi = lambda z: rng.standard_normal(z.shape) * z / Lz * (1 + z / Lz)    # noqa: E731
        _, _, zc = _nodes(m, "T", mod)
        _, _, zw = _nodes(m, "w", mod)
        shp = m.u.interior().shape
        zc3, zw3 = zc + np.zeros(shp), zw + np.zeros(m.w.interior().shape)
        mod.set_model(m, u=np.sqrt(abs(Qu)) * 1e-3 * xi(zc3), w=np.sqrt(abs(Qu)) * 1e-3 * xi(zw3),
                      T=20 + dTdz * zc3 + dTdz * Lz * 1e-2 * xi(zc3), S=35.0)
        return m
    for n in ("u", "v", "w"):
        if n == "w" and cfg["topo"][2] == F:
            continue
        a = rng.random(getattr(m, n).interior().shape) - 0.5
        d = "uvw".index(n)
        if cfg["topo"][d] == B:        # impenetrable walls
            idx = [slice(None)] * 3
            for side in (0, -1):
                idx[d] = side
                a[tuple(idx)] = 0
        init[n] = a
    for t in cfg.get("tracers", ()):
        init[t] = rng.random(m.tracers[t].interior().shape)
    mod.set_model(m, enforce_incompressibility=cfg.get("project_init", True), **init)
    return m


def fields_of(m, oracle):
    out = {}
    names = ["u", "v", "w"] + list(m.tracers.keys())
    for n in names:
        f = getattr(m, n) if n in "uvw" else m.tracers[n]
        out[n] = f.data.copy() if oracle else f.parent()
        g = m.Gn[n]
        out["Gn_" + n] = g.data.copy() if oracle else g.parent()
    out["pNHS"] = m.pNHS.data.copy() if oracle else m.pNHS.parent()
    if m.pHY is not None:
        out["pHY"] = m.pHY.data.copy() if oracle else m.pHY.parent()
    if oracle and m.closure_impl.nu_e is not None:
        out["nu_e"] = m.closure_impl.nu_e.data.copy()
        for n, f in m.closure_impl.kappa_e.items():
            out["kappa_e_" + n] = f.data.copy()
    elif not oracle and getattr(m, "nu_e", None) is not None:
        out["nu_e"] = m.nu_e.parent()
        for n, f in m.kappa_e.items():
            out["kappa_e_" + n] = f.parent()
    return out


def run_case(ocn, name, check_each_step=True):
    """returns the worst relative error over all fields and steps"""
    cfg = CASES[name]
    om = build(O, cfg)
    dm = build(ocn, cfg)
    worst = {}

    def compare(tag):
        a, b = fields_of(om, True), fields_of(dm, False)
        for k in a:
            assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
            scale = max(np.abs(a[k]).max(), 1e-300)
            err = np.abs(a[k] - b[k]).max() / scale
            if np.abs(a[k]).max() < 1e-13:   # identically-zero fields: absolute comparison
                err = np.abs(a[k] - b[k]).max()
            worst[k] = max(worst.get(k, 0.0), err)
    compare("init")
    for s in range(cfg["steps"]):
        dt = cfg["dts"][s] if "dts" in cfg else cfg["dt"]   # a changing time step makes AB2 fall back to Euler (:74)
        O.time_step(om, dt)
        ocn.time_step(dm, dt)
        if check_each_step or s == cfg["steps"] - 1:
            compare(f"step{s}")
    assert abs(om.time - dm.time) < 1e-14 and om.iteration == dm.iteration
    return worst


def run_config1(ocn, dt, steps, N=(128, 128)):
    """BASELINE config 1 at its own size: examples/two_dimensional_turbulence.jl:26-31,42-50 -- 128 x 128 (Periodic, Periodic,
    Flat), 2 pi box, RungeKutta3, ScalarDiffusivity(nu = 1e-5), WENO5 in place of the script's UpwindBiasedFifthOrder, u, v ~
    U[0, 1) minus their means (PCG64(20240601)), then set!'s projection -- through the library and through the oracle,
    every parent array compared after every step.  Returns the library's model."""
    rng = np.random.Generator(np.random.PCG64(20240601))
    u0, v0 = rng.random(N + (1,)), rng.random(N + (1,))
    u0 -= u0.mean()
    v0 -= v0.mean()
    kw = dict(size=N, extent=(2 * np.pi, 2 * np.pi), topology=(P, P, F))
    mk = lambda mod: mod.NonhydrostaticModel(mod.RectilinearGrid(**kw), advection=mod.WENO5(), timestepper="RungeKutta3",   # noqa: E731
                                             closure=mod.ScalarDiffusivity(nu=1e-5))
    m, om = mk(ocn), mk(O)
    ocn.set_model(m, u=u0, v=v0)
    O.set_model(om, u=u0, v=v0)
    for step in range(steps):
        ocn.time_step(m, dt)
        O.time_step(om, dt)
        a, b = fields_of(om, True), fields_of(m, False)
        for k in a:
            assert a[k].shape == b[k].shape
            assert np.abs(a[k] - b[k]).max() <= 2e-11 * max(np.abs(a[k]).max(), 1e-300), (step, k)
    assert abs(om.time - m.time) < 1e-14 and om.iteration == m.iteration == steps
    assert m.max_abs_divergence() <= 1e-12 * np.abs(u0).max() / (2 * np.pi / N[0]) * 50
    return m


def run_two_slab_models_on_one_context(ocn, steps=3):
    """Two forced z-slab models (OCNHIP_FORCE_DIST=1, OCNHIP_OVERLAP=1 set by the caller) sharing ONE context -- its communication
    stream, events and communicator -- stepped alternately, each against its own oracle: the overlapped exchange of one must
    have settled before the other's exchanges start (csrc/api.hip halo_settle)."""
    ctx = ocn.Context(0)
    rng = np.random.default_rng(21)
    pairs = []
    for N, stepper, tr in (((16, 12, 18), "AB2", ()), ((12, 16, 20), "RK3", ("c",))):
        kw = dict(size=N, extent=(1, 1, 1), topology=(P, P, P))
        init = {n: rng.random(N) - 0.5 for n in "uvw"}
        init.update({t: rng.random(N) for t in tr})
        m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(ctx, **kw), advection=ocn.WENO5(), timestepper=stepper, tracers=tr)
        om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5(), timestepper=stepper, tracers=tr)
        ocn.set_model(m, **init)
        O.set_model(om, **init)
        assert "z-slabs" in m.kernel_path, m.kernel_path
        pairs.append((m, om))
    for _ in range(steps):
        for m, om in pairs:
            ocn.time_step(m, 2e-3)
            O.time_step(om, 2e-3)
    for m, om in pairs:
        a, b = fields_of(om, True), fields_of(m, False)
        for k in a:
            assert np.abs(a[k] - b[k]).max() <= 2e-11 * max(np.abs(a[k]).max(), 1e-300), k
