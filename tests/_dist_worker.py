"""One rank of a multi-process run of the LIBRARY (not a restatement): `python tests/_dist_worker.py CASE BACKEND`.

Started by tests/test_distributed_procs.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one process per
rank.  torch.distributed (gloo) carries the rendezvous only (broadcast of the communicator id, barriers); every byte of
field data moves through csrc/comm.hip -- over the host shared-memory transport here (BACKEND hostemu: the host
emulation of the kernels, CPU box; BACKEND gpu: libocnhip.so, all ranks on the one GPU of a test box), over RCCL on a
multi-GPU node.  Each rank steps its slab and compares every parent array, halos included, with the matching window of
the single-domain oracle -- decomposition-independent results, as test/test_distributed_models.jl:361-517 and
test_distributed_poisson_solvers.jl:68-117 demand of the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

P = "Periodic"


def zslab_case(ocn, O, ctx, rank, R, stepper, tracers, N, steps=2, tol=1e-11):
    rng = np.random.default_rng(5)
    init = {n: rng.random(N) - 0.5 for n in "uvw"}
    for t in tracers:
        init[t] = rng.random(N)
    ext = (1.0, N[1] / N[0], N[2] / N[0])
    og = O.RectilinearGrid(size=N, extent=ext, topology=(P,) * 3)
    om = O.NonhydrostaticModel(og, advection=O.WENO5(), timestepper=stepper, tracers=tracers)
    O.set_model(om, **init)
    dt = 0.1 / N[0] / np.abs(om.u.data).max()
    g = ocn.RectilinearGrid(ctx, size=N, extent=ext, topology=(P,) * 3)
    m = ocn.NonhydrostaticModel(g, advection=ocn.WENO5(), timestepper=stepper, tracers=tracers)
    nz = N[2] // R
    ocn.set_model(m, **{n: np.ascontiguousarray(a[:, :, rank * nz:(rank + 1) * nz]) for n, a in init.items()})
    H = 3
    idx = (np.arange(-H, nz + H) + rank * nz) % N[2] + H

    def check(tag):
        refs = [("u", om.u.data, m.u), ("v", om.v.data, m.v), ("w", om.w.data, m.w), ("p", om.pNHS.data, m.pNHS)]
        refs += [(t, om.tracers[t].data, m.tracers[t]) for t in tracers]
        for n, ref, f in refs:
            err = np.abs(f.parent() - ref[:, :, idx]).max() / max(np.abs(ref).max(), 1e-300)
            assert err < tol, (tag, rank, n, err)
    check("set!")
    for s in range(steps):
        O.time_step(om, dt)
        ocn.time_step(m, dt)
        check(f"step {s + 1}")
    assert m.max_abs_divergence() < 1e-9


def yslab_case(ocn, O, ctx, rank, R, kind):
    import copy
    import parity_cases as pc
    from test_distributed_hostemu import _build_on
    cfg = copy.deepcopy(pc.CASES["ppb_amd_config3" if kind == "amd" else "ppb_weno_full"])
    Nx, Nz = cfg["size"][0], cfg["size"][2]
    Ny = 6 * R
    cfg["size"] = (Nx, Ny, Nz)
    om = pc.build(O, cfg)
    names = ["u", "v", "w"] + list(cfg["tracers"])
    rng = np.random.default_rng(77)
    init = {n: rng.random((getattr(om, n) if n in "uvw" else om.tracers[n]).interior().shape) - (0.5 if n in "uvw" else 0.0)
            for n in names}
    init["w"][:, :, 0] = 0
    init["w"][:, :, -1] = 0
    O.set_model(om, **init)
    for _ in range(cfg["steps"]):
        O.time_step(om, cfg["dt"])
    nyl = Ny // R
    m = _build_on(ocn, ctx, dict(cfg))
    ocn.set_model(m, **{n: np.ascontiguousarray(a[:, rank * nyl:(rank + 1) * nyl]) for n, a in init.items()})
    for _ in range(cfg["steps"]):
        ocn.time_step(m, cfg["dt"])
    H = 3
    idx = (np.arange(-H, nyl + H) + rank * nyl) % Ny + H
    refs = {n: ((getattr(om, n) if n in "uvw" else om.tracers[n]).data, getattr(m, n) if n in "uvw" else m.tracers[n]) for n in names}
    refs["p"] = (om.pNHS.data, m.pNHS)
    refs["pHY"] = (om.pHY.data, m.pHY)
    for n, (ref, f) in refs.items():
        err = np.abs(f.parent() - ref[:, idx]).max() / max(np.abs(ref).max(), 1e-300)
        assert err < 2e-11, (rank, n, err)
    assert m.max_abs_divergence() < 1e-9


def poisson_case(ocn, O, ctx, rank, R):
    """test_distributed_poisson_solvers.jl:101-116: lap(phi) == R for a random source, solved on R slabs"""
    import torch
    import torch.distributed as dist
    N = (12, 8, 12 * R)
    rng = np.random.default_rng(9)
    src = rng.random(N)
    src -= src.mean()
    g = ocn.RectilinearGrid(ctx, size=N, extent=(1, 2, 3), topology=(P,) * 3)
    m = ocn.NonhydrostaticModel(g)
    nz = N[2] // R
    mine = m.poisson_solve(np.ascontiguousarray(src[:, :, rank * nz:(rank + 1) * nz]))
    parts = [torch.zeros(mine.shape, dtype=torch.float64) for _ in range(R)]
    dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(mine)))
    phi = np.concatenate([p.numpy() for p in parts], axis=2)
    lap = np.zeros(N)
    for ax, d in ((0, 1 / N[0]), (1, 2 / N[1]), (2, 3 / N[2])):
        lap += (np.roll(phi, -1, ax) - 2 * phi + np.roll(phi, 1, ax)) / d ** 2
    assert np.abs(lap - src).max() < 1e-10 * np.abs(src).max()
    from oracle.poisson import FFTBasedPoissonSolver
    ref = FFTBasedPoissonSolver(O.RectilinearGrid(size=N, extent=(1, 2, 3), topology=(P,) * 3)).solve(src)
    assert np.abs(phi - ref).max() < 1e-11 * np.abs(ref).max()


def main():
    case, backend = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "hostemu":
        os.environ["OCNHIP_LIB"] = os.path.join(ROOT, "tests", "hostemu", "libocnhip_hostemu.so")
    else:
        os.environ.pop("OCNHIP_LIB", None)
        os.environ["OCNHIP_TRANSPORT"] = "shm"      # R ranks on the one GPU of a test box
    import __graft_entry__ as ge
    ocn = ge.load_package()
    ocn._lib.load()
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    par = import_module("ocnhip.parallel")
    ctx = ocn.Context(0)
    par.init_comm(ctx, dist, rank, world)
    import oracle as O
    big = backend == "gpu"
    if case == "zslab_ab2":
        zslab_case(ocn, O, ctx, rank, world, "AB2", (), (16, 12, 16 * world) if big else (8, 8, 8 * world))
    elif case == "zslab_rk3_tracer":
        zslab_case(ocn, O, ctx, rank, world, "RK3", ("c",), (16, 12, 16 * world) if big else (8, 8, 8 * world))
    elif case == "zslab_wide":      # rows wider than a workgroup: the x-tiled kernel on slabs (config-4 shape in miniature)
        zslab_case(ocn, O, ctx, rank, world, "AB2", (), (272, 8, 8 * world), steps=1)
    elif case == "zslab_custom":    # 128- / 256-point rows: the custom transform passes on slabs; the w* term above each slab enters
        # in spectral space through its owner's Green's-function sums and the pressure plane below each slab is one more level of
        # its own convolution -- two exchanges per step (OCNHIP_WSTAR_EXCHANGE=1 / OCNHIP_PHI_EXCHANGE=1: the planes are shipped)
        if big:
            zslab_case(ocn, O, ctx, rank, world, "RK3", ("c",), (256, 128, 16 * world), steps=2)
        else:
            zslab_case(ocn, O, ctx, rank, world, "AB2", (), (128, 128, 8 * world), steps=1)
    elif case == "yslab_amd":
        yslab_case(ocn, O, ctx, rank, world, "amd")
    elif case == "yslab_scalar":
        yslab_case(ocn, O, ctx, rank, world, "scalar")
    elif case == "poisson":
        poisson_case(ocn, O, ctx, rank, world)
    elif case.startswith("hydro_bands"):   # HydrostaticFreeSurfaceModel on latitude bands, replicated free surface (config 5's layout)
        import test_hydrostatic_bands as hb
        gridname = case.split(":")[1]                                      # "hydro_bands:<grid>[:<overlap rows of a banded free surface>]"
        overlap = int(case.split(":")[2]) if case.count(":") > 1 else 0
        hb.band_check(hb.band_run(ocn, ctx, rank, world, gridname, 3, 150.0, overlap=overlap), hb.run_single_domain_oracle(gridname, 3, 150.0))
    else:
        raise SystemExit(f"unknown case {case}")
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {case} ok", flush=True)


if __name__ == "__main__":
    main()
