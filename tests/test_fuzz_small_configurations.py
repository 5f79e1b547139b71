"""Seeded random small configurations -- topology (Periodic / Bounded / Flat per direction), 1 to 7 cells per direction (so
extents smaller than the halo occur), scheme, stepper, closure, Coriolis, buoyancy, tracers, halo sizes, random Flux / Value /
Gradient conditions on Bounded sides -- through the host emulation of the library and through the oracle, two steps each,
every parent array compared.  The degenerate shapes are the point: a one-level (Flat, Flat, Bounded) column once came out NaN."""
import numpy as np
import pytest

import parity_cases as pc

P, B, F = "Periodic", "Bounded", "Flat"


def random_case(rng):
    while True:
        topo = tuple(str(rng.choice([P, B, F], p=[0.5, 0.35, 0.15])) for _ in range(3))
        if sum(t == F for t in topo) < 3:
            break
    size = tuple(int(rng.integers(1, 8)) for t in topo if t != F)
    adv = str(rng.choice(["WENO5", "WENO5JS", "U5", "U3", "U1", "C4", "C2"]))
    cfg = dict(size=size, topo=topo, extent=tuple(1.0 + 0.3 * i for i in range(len(size))), adv=adv,
               stepper=str(rng.choice(["AB2", "RK3"])), steps=2, dt=1e-3)
    ntr = int(rng.integers(0, 3))
    closure = str(rng.choice(["none", "scalar", "amd"], p=[0.4, 0.4, 0.2]))
    if closure == "amd" and F in topo:
        closure = "scalar"
    buoy = None
    if ntr == 1 and rng.random() < 0.5:
        cfg["tracers"], buoy = ("b",), "b"
    elif ntr == 2 and rng.random() < 0.5:
        cfg["tracers"], buoy = ("T", "S"), "TS"
    elif ntr:
        cfg["tracers"] = tuple("abc"[:ntr])
    if buoy and topo[2] != F:
        cfg["buoyancy"] = buoy
    if closure == "scalar":
        cfg["closure"] = (1e-2, 2e-2)
    if closure == "amd":
        cfg["closure"] = "amd"
    if rng.random() < 0.4:
        cfg["coriolis"] = 0.1
    if adv == "C2":
        cfg["halo"] = tuple(int(rng.integers(1, 4)) for _ in size)
    elif adv in ("C4", "U3", "U1"):
        cfg["halo"] = tuple(int(rng.integers(2, 4)) for _ in size)
    if rng.random() < 0.6:
        bcs, sides = {}, {0: ("west", "east"), 1: ("south", "north"), 2: ("bottom", "top")}
        for fn in ["u", "v", "w"] + list(cfg.get("tracers", ())):
            for d in range(3):
                if topo[d] != B or (fn in "uvw" and "uvw".index(fn) == d):      # wall-normal velocity stays impenetrable
                    continue
                for sd in sides[d]:
                    if rng.random() < 0.35:
                        bcs.setdefault(fn, {})[sd] = (str(rng.choice(["flux", "value", "gradient"])), float(rng.normal() * 1e-2))
        if bcs:
            cfg["bcs"] = bcs
    return cfg


def _run(ocn, seed):
    cfg = random_case(np.random.default_rng(1000 + seed))
    pc.CASES["_fuzz"] = cfg
    try:
        om, dm = pc.build(pc.O, cfg), pc.build(ocn, cfg)
        for s in range(cfg["steps"]):
            pc.O.time_step(om, cfg["dt"])
            ocn.time_step(dm, cfg["dt"])
            a, b = pc.fields_of(om, True), pc.fields_of(dm, False)
            scale = max(max(np.abs(v).max() for v in a.values()), 1e-300)      # one scale for all fields: fields that vanish
            for k in a:                                                           # analytically hold round-off only
                assert a[k].shape == b[k].shape, (cfg, k)
                assert np.isfinite(b[k]).all(), (cfg, k)
                err = np.abs(a[k] - b[k]).max()
                assert err <= 2e-10 * max(np.abs(a[k]).max(), 1e-2 * scale) + 1e-14, (cfg, k, err)   # rtol + the atol 1e-12 max|field| of SURVEY 8c
    finally:
        pc.CASES.pop("_fuzz", None)


@pytest.mark.parametrize("seed", range(150))
def test_random_small_configuration(ocn, backend, seed):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    _run(ocn, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(80))
def test_random_small_configuration_gpu(ocn, seed):
    _run(ocn, seed)


# ---- medium shapes: the tiled kernels' variants (row widths around the workgroup sizes, odd extents, x-tiles, walls, slabs) -----
def random_medium_case(rng):
    zt = str(rng.choice([P, B], p=[0.5, 0.5]))
    xt, yt = (str(rng.choice([P, B], p=[0.8, 0.2])) for _ in range(2)) if zt == B else (P, P)
    nx = int(rng.choice([16, 31, 32, 57, 64, 65, 100, 127, 128, 129, 192, 255, 256, 257, 300, 384]))
    size = (nx, int(rng.integers(6, 28)), int(rng.integers(6, 26)))
    cfg = dict(size=size, topo=(xt, yt, zt), extent=(1.0, 0.8, 0.6), adv=str(rng.choice(["WENO5", "WENO5JS", "U5"])),
               stepper=str(rng.choice(["AB2", "RK3"])), steps=2, dt=2e-3 / nx * 16)
    ntr = int(rng.integers(0, 3))
    closure = str(rng.choice(["none", "scalar", "amd"], p=[0.4, 0.35, 0.25]))
    if ntr == 2 and zt == B and rng.random() < 0.6:
        cfg["tracers"], cfg["buoyancy"] = ("T", "S"), "TS"
    elif ntr:
        cfg["tracers"] = tuple("ab"[:ntr])
    if closure == "scalar":
        cfg["closure"] = (1e-3, 2e-3)
    elif closure == "amd":
        cfg["closure"] = "amd"
    if rng.random() < 0.4:
        cfg["coriolis"] = 0.1
    if zt == B and rng.random() < 0.5:      # stretched z
        f = np.sort(rng.random(size[2] - 1)) * 0.6 - 0.6
        cfg["zfaces"] = [-0.6] + [float(v) for v in f] + [0.0]
        cfg["xy"] = ((0, 1.0), (0, 0.8))
        del cfg["extent"]
        if min(np.diff(cfg["zfaces"])) < 1e-3:
            cfg["zfaces"] = list(np.linspace(-0.6, 0, size[2] + 1))
    if zt == B and rng.random() < 0.5:
        cfg["bcs"] = {"u": {"top": ("flux", -1e-3)}}
        if "tracers" in cfg:
            cfg["bcs"][cfg["tracers"][0]] = {"top": ("flux", 2e-3), "bottom": ("gradient", 0.01)}
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_random_medium_configuration_gpu(ocn, seed, monkeypatch):
    rng = np.random.default_rng(5000 + seed)
    cfg = random_medium_case(rng)
    if cfg["topo"] == (P, P, P) and rng.random() < 0.4 and cfg["size"][2] >= 8:
        monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")      # the slab code path, with the halo planes on the communication stream
        monkeypatch.setenv("OCNHIP_OVERLAP", "1")
    pc.CASES["_fuzz"] = cfg
    try:
        worst = pc.run_case(ocn, "_fuzz")
        bad = {k: v for k, v in worst.items() if not v <= 5e-10}
        assert not bad, (cfg, bad)
    finally:
        pc.CASES.pop("_fuzz", None)
