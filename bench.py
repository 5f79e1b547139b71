#!/usr/bin/env python3
"""bench.py -- cell-updates/s of NonhydrostaticModel time_step! (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1: BASELINE config 2 -- 256^3 triply-periodic RectilinearGrid, WENO5 (Z weights), AB2, Float64,
no tracers; u, v, w ~ U[-0.5, 0.5) (PCG64 seed 1), projected by set!; fixed dt = 0.2 dx / max|u|.
N > 1: one process per GPU, z-slab decomposition inside libocnhip.so (RCCL halo exchange, slab Poisson solver):
    N = 2, 4 : the same 256^3 box cut into N slabs            (strong scaling of the metric's workload)
    N = 8    : BASELINE config 4 -- 512 x 512 x 256 in z-slabs of 32 (dx = dy = dz = 1/512)
    --scaling weak : 256 x 256 x 256 N instead.
Started without WORLD_SIZE (plain `python bench.py --gpus N`), this process only spawns the N ranks -- before
anything touches a GPU -- and relays rank 0's line; under torch.distributed.run it is one of the ranks.

One JSON line is printed by rank 0.  `value` is whole-job cell-updates/s with all inputs resident in
HBM when the timed region starts.  `roofline` is for the dominant kernel (measured live with HIP
events on the library's stream); `cpu_baseline` is the CPU restatement timed on this host (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r03_traffic.json"
HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
# algorithmic bytes per cell: whole AB2 step and per phase
B_ALG_STEP = 336.0          # config 2 / 4: AB2, no tracers; +49 per passive tracer; RK3 = 3 stages
B_ALG_TRACER = 49.0
B_ALG_STAGE_C3 = 508.0      # config 3 per RK3 stage (SURVEY.md 8d estimate, frozen here): 1524 per step
B_ALG_STAGE_C1 = 228.0      # config 1 per RK3 stage: 2-D, u and v only (tendencies 32, updates 64, halos 4, rhs 24,
                            # r2c transforms fwd + inv over 2 axes 64, projection 40): 684 per step


def alg_bytes_per_cell_update(args):
    if args.config == 3:
        return 3 * B_ALG_STAGE_C3
    if args.config == 1:
        return 3 * B_ALG_STAGE_C1
    per = B_ALG_STEP + B_ALG_TRACER * args.tracers
    return per * (3 if args.stepper.upper() in ("RK3", "RUNGEKUTTA3") else 1)

B_ALG_PHASE = {"tendencies": 48.0, "step": 96.0, "rhs": 32.0, "fft_forward": 48.0, "spectral_solve": 16.0,
               "fft_backward": 48.0, "pcorrect": 56.0, "fused_tendency_step": 96.0, "poisson_fused": 80.0}


def cpu_baseline():
    """CPU restatement of the reference's CPU() path on a bounded sample of the same workload (config 2: triply periodic
    WENO5 AB2 + FFT Poisson), timed on this host: the C++/OpenMP restatement (oracle/cpu/ocn_cpu.cpp) with one thread
    at 128^3 and with every available core at 256^3, and the NumPy oracle at 64^3.  About 20 s in all."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:    # a container's CPU share (cgroup v2 quota) is what this process can really use: more threads only thrash
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = min(cores, int(os.environ.get("OCNHIP_CPU_THREADS", "64")))
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"unit": "cell-updates/s", "kind": "port", "cpu_model": model, "host_cores": cores}
    lib = os.path.join(ROOT, "oracle", "cpu", "libocn_cpu.so")
    # run in a child process: whatever happens to the baseline (a host without the library's ISA level, an out-of-memory
    # kill), the bench line of the GPU path is still printed
    child = r"""
import ctypes as C, json, sys, numpy as np
L = C.CDLL(sys.argv[1]); PD = C.POINTER(C.c_double)
L.ocncpu_run.restype = C.c_int
L.ocncpu_run.argtypes = [C.c_int] * 3 + [C.c_double] * 3 + [PD] * 4 + [C.c_double, C.c_int, C.c_int, PD]
def run(n, threads, steps):
    rng = np.random.default_rng(1)
    a = [np.asfortranarray(rng.random((n, n, n)) - 0.5) for _ in range(3)]
    secs = C.c_double()
    rc = L.ocncpu_run(n, n, n, 1.0, 1.0, 1.0, a[0].ctypes.data_as(PD), a[1].ctypes.data_as(PD), a[2].ctypes.data_as(PD),
                      None, 0.2 / n / 0.5, steps, threads, C.byref(secs))
    assert rc == 0, rc
    return n ** 3 * steps / secs.value
cores = int(sys.argv[2])
one = run(128, 1, 2)
print(json.dumps({"one": one, "all": run(256, cores, 4) if cores > 1 else one}))
"""
    try:
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, "-c", child, lib, str(cores)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError(f"exit code {r.returncode}: {r.stderr[-200:]}")
        res = json.loads(r.stdout.strip().splitlines()[-1])
        one, allc = res["one"], res["all"]
        out.update({"value": allc, "cores": cores,
                    "sample": f"4 AB2 WENO5 steps at 256^3 (the whole workload), C++/OpenMP restatement, {cores} threads",
                    "single_core": {"value": one, "cores": 1, "sample": "2 steps at 128^3, same code, 1 thread"}})
    except Exception as e:   # noqa: BLE001 -- the baseline is a reported extra; the bench line must still be printed
        out.update({"value": None, "cores": 0, "sample": f"C++ restatement unavailable: {e!r}"})
    import oracle as O
    n = 64
    N = (n, n, n)
    rng = np.random.default_rng(1)
    g = O.RectilinearGrid(size=N, extent=(1, 1, 1), topology=(O.Periodic,) * 3)
    m = O.NonhydrostaticModel(g, advection=O.WENO5())
    O.set_model(m, u=rng.random(N) - 0.5, v=rng.random(N) - 0.5, w=rng.random(N) - 0.5)
    dt = 0.2 / n / np.abs(m.u.data).max()
    O.time_step(m, dt)   # warm-up (Euler step)
    t0 = time.perf_counter()
    steps = 0
    while steps < 3 or (time.perf_counter() - t0 < 4.0 and steps < 12):
        O.time_step(m, dt)
        steps += 1
    el = time.perf_counter() - t0
    out["numpy_oracle"] = {"value": n ** 3 * steps / el, "cores": 1, "sample": f"{steps} steps at {n}^3, NumPy oracle"}
    out["reference_published"] = {"value": 8.58e5, "note": "single-core Julia CPU() at 256^3, other host (BASELINE.md); context only"}
    return out


def build_config3(ocn, ctx, args):
    """BASELINE config 3: 256x256x128 (Periodic, Periodic, Bounded), the stretched z of
    examples/ocean_wind_mixing_and_convection.jl:38-60 with Nz = 128, T and S, FPlane, linear EOS, AMD, wind stress /
    heat flux / (constant) evaporation / bottom gradient BCs, RK3, WENO5 in place of the script's U5."""
    Nx, Ny, Nz = tuple(args.size) if args.size else (256, 256, 128)
    world = int(os.environ.get("WORLD_SIZE", "1"))      # weak scaling along y: the library cuts (P,P,Bounded) grids into y-slabs
    Lz, refinement, stretching = 32.0, 1.2, 12.0
    k = np.arange(1, Nz + 2)
    h = (k - 1) / Nz
    zf = Lz * ((1 + (h - 1) / refinement) * (1 - np.exp(-stretching * h)) / (1 - np.exp(-stretching)) - 1)
    grid = ocn.RectilinearGrid(ctx, size=(Nx, Ny * world, Nz), x=(0.0, 2.0 * Nx), y=(0.0, 2.0 * Ny * world), z=zf,
                               topology=("Periodic", "Periodic", "Bounded"))
    QT = 200.0 / (1026.0 * 3991.0)
    Qu = -1.225 / 1026.0 * 2.5e-3 * 10 * 10
    dTdz = 0.01
    bcs = {"u": {"top": ocn.FluxBC(Qu)}, "T": {"top": ocn.FluxBC(QT), "bottom": ocn.GradientBC(dTdz)},
           "S": {"top": ocn.FluxBC(-1e-3 / 3600 * 35.0)}}
    model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO5(), timestepper="RungeKutta3", tracers=("T", "S"),
                                    coriolis=ocn.FPlane(1e-4), closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(thermal_expansion=2e-4, haline_contraction=8e-4),
                                    boundary_conditions=bcs)
    rng = np.random.default_rng(3 + int(os.environ.get("RANK", "0")))
    zc = 0.5 * (zf[1:] + zf[:-1]).reshape(1, 1, -1)
    zw = zf.reshape(1, 1, -1)
    noise = lambda z, shape: rng.standard_normal(shape) * z / Lz * (1 + z / Lz)   # noqa: E731
    T0 = 20 + dTdz * zc + dTdz * Lz * 1e-6 * noise(zc, (Nx, Ny, Nz))
    u0 = np.sqrt(abs(Qu)) * 1e-3 * noise(zc, (Nx, Ny, Nz))
    w0 = np.sqrt(abs(Qu)) * 1e-3 * noise(zw, (Nx, Ny, Nz + 1))
    w0[:, :, 0] = 0
    w0[:, :, -1] = 0
    ocn.set_model(model, u=u0, w=w0, T=T0, S=35.0)
    return model, 1.0, (Nx, Ny * world, Nz), (Nx, Ny, Nz)


def build_config1(ocn, ctx, args):
    """BASELINE config 1: examples/two_dimensional_turbulence.jl -- 128 x 128 (Periodic, Periodic, Flat), 2 pi box,
    RK3, ScalarDiffusivity(nu = 1e-5), WENO5 in place of the script's UpwindBiasedFifthOrder."""
    Nx, Ny = (args.size[0], args.size[1]) if args.size else (128, 128)
    grid = ocn.RectilinearGrid(ctx, size=(Nx, Ny), extent=(2 * np.pi, 2 * np.pi), topology=("Periodic", "Periodic", "Flat"))
    model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO5(), timestepper="RungeKutta3", closure=ocn.ScalarDiffusivity(nu=1e-5))
    rng = np.random.default_rng(1)
    u, v = rng.standard_normal((Nx, Ny, 1)), rng.standard_normal((Nx, Ny, 1))
    ocn.set_model(model, u=u - u.mean(), v=v - v.mean())
    return model, 0.2 * (2 * np.pi / Nx) / 4.0, (Nx, Ny, 1), (Nx, Ny, 1)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    touched a GPU and never will), relay rank 0's JSON line, exit non-zero if any rank fails."""
    import socket
    import subprocess
    n = args.gpus
    if not args.rehearse_hostemu:
        import torch
        have = torch.cuda.device_count()          # does not initialise the GPU
        ndev = int(os.environ.get("OCNHIP_BENCH_NDEV", "0"))
        if have < n and not ndev:
            raise SystemExit(f"bench.py: --gpus {n} but this node shows {have} GPU(s); nothing was measured "
                             "(rehearse ranks on fewer GPUs with OCNHIP_TRANSPORT=shm OCNHIP_BENCH_NDEV=<gpus>)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("OCNHIP_BENCH_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for pr in list(alive):
            code = pr.poll()
            if code is not None:
                alive.remove(pr)
                rc = rc or code
        if alive and (rc or time.time() > deadline):   # a rank failed (or hung): stop exactly the children started here
            for pr in alive:
                pr.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    for pr in procs:
        pr.wait()
    raise SystemExit(rc)



# ---- BASELINE config 5: HydrostaticFreeSurfaceModel on a LatitudeLongitudeGrid, SplitExplicitFreeSurface --------------------------
B_ALG_C5_3D = 41 * 8.0      # field sweeps of one time step per 3-D cell: tendencies 13 (u, v, w, pHY' read + G_u, G_v written; u, v, w + T, S read +
                            # G_T, G_S written), the merged passes after them 28
B_ALG_C5_2D = 128.0         # per 2-D cell and barotropic substep (eta, U, V, the three averages read and written, G^U, G^V, H^fc, H^cf read)


def run_config5(args, ocn, ctx, dist, rank, world, transport, real_stdout):
    """1024 x 512 x 128, T and S with a linear equation of state, HydrostaticSphericalCoriolis, VectorInvariant momentum advection,
    CenteredSecondOrder tracer advection, 200 barotropic substeps; one GPU, or latitude bands with a banded free surface (N ranks)"""
    import numpy as np
    H = ocn.hydrostatic
    Nx, Ny, Nz = tuple(args.size) if args.size else ((32, 8 * world, 4) if args.rehearse_hostemu else (1024, 512, 128))
    substeps = 12 if args.rehearse_hostemu else 200
    kw = dict(size=(Nx, Ny, Nz), longitude=(-180, 180), latitude=(-75, 75), z=(-4000, 0), halo=(3, 3, 3), arch=ctx)
    overlap = min(20, Ny // world) if world > 1 else 0
    grid = H.LatitudeLongitudeGrid(partition="y" if world > 1 else None, **kw)
    st = H.HydrostaticState(grid, tracers=("T", "S"), buoyancy=("TS", 9.80665, 1.67e-4, 7.8e-4, "T", "S"), substeps=substeps,
                            coriolis=("HydrostaticSphericalCoriolis", 7.292115e-5, "EnstrophyConserving"), barotropic_overlap=overlap,
                            tracer_advection=args.tracer_advection, momentum_advection=args.momentum_advection,
                            closure=(1e-2, 1e-4) if args.implicit_diffusion else None)
    # solid-body rotation in balance with the free surface (Williamson et al. 1992, case 2): stays bounded however many steps are timed
    R, Om, U0, g = 6371.0e3, 7.292115e-5, 10.0, 9.80665
    st.u.set(lambda x, y, z: U0 * np.cos(np.pi * y / 180) + 0 * x + 0 * z)
    st.free_surface.eta.set(lambda x, y: -(R * Om * U0 + U0 ** 2 / 2) * np.sin(np.pi * y / 180) ** 2 / g + 0 * x)
    st.tracers["T"].set(lambda x, y, z: 20 * np.cos(np.pi * y / 180) + 5e-3 * z + 0 * x)
    st.tracers["S"].set(35.0)
    H.update_state(st)
    dt = 60.0

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
    H.time_step(st, dt, euler=True)
    for _ in range(max(args.warmup - 1, 0)):
        H.time_step(st, dt)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        H.time_step(st, dt)
    ctx.sync()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])
    ms = el / args.steps * 1e3
    cells = Nx * Ny * Nz
    value = cells / (ms * 1e-3)
    vmax = float(np.abs(st.v.interior()).max())
    finite = bool(np.isfinite(st.u.interior()).all())
    if rank != 0:
        return
    b_alg = B_ALG_C5_3D + B_ALG_C5_2D * substeps / Nz
    out = {"metric": "cell-updates/sec per time_step!, 1024x512x128 HydrostaticFreeSurfaceModel (BASELINE config 5)", "value": value,
           "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": ("rehearsal on the host emulation of the kernels: plumbing check, NOT a measurement" if args.rehearse_hostemu else
                    "synthetic; REHEARSAL: ranks share GPUs and exchange through host shared memory -- not a scaling measurement"
                    if world > 1 and transport and transport.startswith("shm") else "synthetic"),
           "config": {"workload": f"{Nx}x{Ny}x{Nz} LatitudeLongitudeGrid, HydrostaticFreeSurfaceModel, SplitExplicitFreeSurface ({substeps} substeps), "
                                  f"{args.momentum_advection} momentum advection, {args.tracer_advection} tracers T + S, linear equation of state, "
                                  "HydrostaticSphericalCoriolis, AB2 (BASELINE config 5"
                                  + (", implicit vertical diffusion)" if args.implicit_diffusion else " without closures)"),
                      "decomposition": f"latitude bands x{world}" + (f", banded free surface with {overlap} overlap rows" if world > 1 else ""),
                      "dt": dt, "init": "solid-body rotation in balance with the free surface", "transport": transport},
           # the whole step against the HBM roofline (no single kernel dominates: sub-cycle 29 %, k_hy_Guv 16 %, k_hy_tracers 13 %);
           # duration = the timed region itself (host clock between stream synchronisations)
           "roofline": {"bound": "hbm", "kernel": "time_step! (all launches)", "achieved": value / world * b_alg / 1e9, "peak": (HBM_PEAK / 1e9),
                        "unit": "GB/s", "frac": value / world * b_alg / 1e9 / (HBM_PEAK / 1e9), "traffic": None,
                        "alg_bytes_per_cell_update": b_alg},
           "max_abs_v": vmax, "finite": finite, "cpu_baseline": None}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_config5()
    line = json.dumps(out)
    if real_stdout is not None:
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())
    else:
        print(line)


def cpu_baseline_config5():
    """the NumPy oracle (oracle/hydrostatic.py: the checker, timed here as the CPU baseline) on a bounded sample of config 5:
    128 x 64 x 16 cells, 20 substeps, two time steps on one core"""
    import numpy as np
    from oracle import hydrostatic as OH
    from oracle import split_explicit as OS
    N = (128, 64, 16)
    g = OS.LatitudeLongitudeGrid(size=N, longitude=(-180, 180), latitude=(-75, 75), z=(-4000, 0), halo=(3, 3, 3))
    st = OH.HydrostaticState(g, tracers=("T", "S"), buoyancy=("TS", 9.80665, 1.67e-4, 7.8e-4, "T", "S"), substeps=20,
                             coriolis=("HydrostaticSphericalCoriolis", 7.292115e-5, "EnstrophyConserving"))
    st.u.set(lambda x, y, z: 10 * np.cos(np.pi * y / 180) + 0 * x + 0 * z)
    st.tracers["T"].set(lambda x, y, z: 20 + 5e-3 * z + 0 * x + 0 * y)
    st.tracers["S"].set(35.0)
    OH.update_state(st)
    OH.time_step(st, 60.0, euler=True)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 10.0:
        OH.time_step(st, 60.0)
        n += 1
    el = time.perf_counter() - t0
    return {"value": N[0] * N[1] * N[2] * n / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{N[0]}x{N[1]}x{N[2]} cells, 20 substeps, {n} NumPy-oracle time steps in {el:.1f} s (single thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="general-path models: replay steps from hipGraphs in the timed region (auto: when a step is under 0.5 ms)")
    ap.add_argument("--size", type=int, nargs=3, default=None, help="override the GLOBAL size (debug)")
    ap.add_argument("--stepper", default="AB2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tracers", type=int, default=0, help="passive tracers (config 2b: 1)")
    ap.add_argument("--nu", type=float, default=0.0, help="config 2 with ScalarDiffusivity(nu = kappa = NU) (DNS-style; 0: inviscid headline)")
    ap.add_argument("--topology", default="PPP", help="config 2 with other x/y/z topologies, e.g. PBB (debug / widening rows)")
    ap.add_argument("--tracer-advection", default="CenteredSecondOrder",
                    help="config 5: CenteredSecondOrder (the model's default), CenteredFourthOrder, UpwindBiasedFifthOrder or WENO5")
    ap.add_argument("--implicit-diffusion", action="store_true",
                    help="config 5: closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(); nu = 1e-2, kappa = 1e-4)")
    ap.add_argument("--momentum-advection", default="VectorInvariantEnstrophyConserving",
                    help="config 5: VectorInvariantEnstrophyConserving (default), VectorInvariantEnergyConserving or WENOVectorInvariantVorticityStencil")
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config: 2 (headline, default), 1 (2-D turbulence), 3 (ocean LES) or 5 (hydrostatic model on the sphere)")
    ap.add_argument("--lib", default=None, help="another build of libocnhip.so (kernel experiments; never the host emulation)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="N > 1: strong = 256^3 on 2 / 4 GPUs and config 4 (512x512x256) on 8; weak = 256x256x256N")
    ap.add_argument("--init", default="random", choices=("random", "smooth"),
                    help="config 2 initial velocities: U[-0.5,0.5) per cell (SURVEY 8d, the metric) or a smooth sign-coherent field")
    ap.add_argument("--rehearse-hostemu", action="store_true",
                    help="plumbing rehearsal on the host emulation of the kernels (CPU only, tiny grid): NOT a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)           # never returns

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    os.environ.pop("OCNHIP_LIB", None)      # the product library only
    if args.graph == "off":                 # the knob is read when the model is created (csrc/fused.hip fused_read_knobs)
        os.environ["OCNHIP_NO_GRAPH"] = "1"
    if args.rehearse_hostemu:
        os.environ["OCNHIP_LIB"] = os.path.join(ROOT, "tests", "hostemu", "libocnhip_hostemu.so")
        args.no_cpu_baseline = True
    elif args.lib:
        if "hostemu" in args.lib:
            raise SystemExit("--lib is for GPU builds of the library only")
        os.environ["OCNHIP_LIB"] = os.path.abspath(args.lib)
    import __graft_entry__ as ge
    ocn = ge.load_package()                 # loads libocnhip.so before anything touches torch
    ocn._lib.load()

    dist = None
    real_stdout = None
    transport = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        # rendezvous / barriers / max-reduce only; the data path is RCCL inside the library.  gloo prints a connection
        # banner on stdout at its first collective: file descriptor 1 is parked on stderr for the whole run and the one
        # JSON line goes to the saved descriptor.  No collective before the library has initialised the GPU (below): a
        # torch-side GPU initialisation first would leave the library without a device.
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("gloo")

    ndev = int(os.environ.get("OCNHIP_BENCH_NDEV", "0"))     # rehearsal on fewer GPUs than ranks (shm transport only)
    ctx = ocn.Context(0 if args.rehearse_hostemu else (local_rank % ndev if ndev else local_rank))
    if world > 1:
        from importlib import import_module
        par = import_module("ocnhip.parallel")
        transport = par.init_comm(ctx, dist, rank, world, allow_fallback=True)
    # global grid: the library cuts triply periodic grids into z-slabs (Nz / world levels per rank) and
    # (Periodic, Periodic, Bounded) grids into y-slabs
    if args.size:
        Nglobal = tuple(args.size)
    elif args.rehearse_hostemu:
        Nglobal = (16, 16, 8 * world)
    elif world > 1 and args.scaling == "weak":
        Nglobal = (256, 256, 256 * world)
    elif world == 8:
        Nglobal = (512, 512, 256)           # BASELINE config 4
    else:
        Nglobal = (256, 256, 256)
    extent = tuple(float(x) / Nglobal[0] for x in Nglobal)    # cubic cells, x extent 1
    if args.config == 5:
        run_config5(args, ocn, ctx, dist, rank, world, transport, real_stdout)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.config == 3:
        model, dt3, Nglobal, n = build_config3(ocn, ctx, args)
    elif args.config == 1:
        model, dt3, Nglobal, n = build_config1(ocn, ctx, args)
    else:
        topo = tuple({"P": "Periodic", "B": "Bounded"}[c] for c in args.topology.upper())
        grid = ocn.RectilinearGrid(ctx, size=Nglobal, extent=extent, topology=topo)
        tnames = tuple(f"c{i}" for i in range(args.tracers))
        closure = ocn.ScalarDiffusivity(nu=args.nu, kappa=args.nu) if args.nu else None
        model = ocn.NonhydrostaticModel(grid, advection=ocn.WENO5(), timestepper=args.stepper, tracers=tnames, closure=closure)
        n = model.u.size                   # this rank's slab
        rng = np.random.default_rng(1 + rank)
        if args.init == "smooth":
            # sign-coherent velocities (what resolved flows look like): a few Fourier modes, amplitude 0.5
            X, Y, Z = model.nodes("u")
            off = rank * n[2] * extent[2] / Nglobal[2]
            two_pi = 2 * np.pi
            init = dict(u=0.5 * np.sin(two_pi * X) * np.cos(two_pi * Y) + 0 * Z,
                        v=-0.5 * np.cos(two_pi * X) * np.sin(two_pi * Y) + 0 * Z,
                        w=0.25 * np.sin(two_pi * (Z + off) / extent[2]) + 0 * X + 0 * Y)
        else:
            init = dict(u=rng.random(model.u.size) - 0.5, v=rng.random(model.v.size) - 0.5, w=rng.random(model.w.size) - 0.5)
        init.update({t: rng.random(model.tracers[t].size) for t in tnames})
        ocn.set_model(model, **init)
    umax = np.abs(model.u.interior()).max()
    if dist is not None:
        import torch
        t = torch.tensor([umax], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        umax = float(t[0])
    dt = 0.2 * (extent[0] / Nglobal[0]) / umax
    if args.config != 2:
        dt = dt3

    # Warm-up.  Its last steps run with an event pair around every phase: they give the per-phase table and name the
    # dominant phase.  In the timed region only that phase records events -- each record costs ~4 us of stream time,
    # and a dozen of them per step is 3 % of a 1.4 ms step.
    PH_ALL = list(B_ALG_PHASE) + ["fused_tracer_step", "fill_halos", "store", "copy_pressure", "time_step", "halo_exchange",
                                  "transpose", "amd_diffusivities", "hydrostatic"]
    # The card needs ~40 ms of load to leave its idle clocks (a 5-step run of config 2 measures 1.52 ms/step, a 200-step
    # run 1.32): untimed spin-up steps run until 80 ms of stepping have passed (none if the warm-up already covers that).
    spinup = 0
    if args.rehearse_hostemu:
        pass
    elif world > 1:
        for _ in range(40):            # every step is collective: all ranks must run the same number of them
            ocn.time_step(model, dt)
        spinup = 40
    else:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.08:
            ocn.time_step(model, dt)
            ctx.sync()
            spinup += 1
    nprobe = min(2, args.warmup)
    for _ in range(args.warmup - nprobe):
        ocn.time_step(model, dt)
    ctx.sync()
    ctx.profile_filter(None)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(nprobe):
        ocn.time_step(model, dt)
    ctx.sync()
    phases = {}
    for ph in PH_ALL:
        avg, cnt = ctx.profile_read(ph)
        if cnt:
            phases[ph] = {"avg_ms": avg, "launches": cnt}
    cand = {p: v for p, v in phases.items() if p in B_ALG_PHASE}
    dom = max(cand, key=lambda p: cand[p]["avg_ms"] * cand[p]["launches"]) if cand else None
    ctx.profile(False)
    ctx.profile_reset()
    ctx.profile_filter(dom)
    # Launch-bound models (config 1: ~50 launches of a few us each) replay whole steps from hipGraphs (csrc/api.hip
    # step_graphed; captured during the spin-up above).  Phase events cannot sit inside a replayed graph, so for those the
    # timed region records none and the dominant kernel's duration is the warm-up probe's.
    replays0, graph_ok = (0, False) if args.rehearse_hostemu else model.graph_replays
    step_probe = phases.get("time_step", {}).get("avg_ms", 1e9)
    use_graph = graph_ok and args.graph != "off" and (args.graph == "on" or step_probe < 0.5)
    ctx.profile(not use_graph and not os.environ.get("OCNHIP_BENCH_NOPROF"))   # NOPROF: debug, cost of the events themselves
    if dist is not None:
        dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ocn.time_step(model, dt)
    ctx.sync()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    ctx.profile(False)
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])

    div = model.max_abs_divergence()
    cells = Nglobal[0] * Nglobal[1] * Nglobal[2]
    ms = el / args.steps * 1e3
    value = cells * args.steps / el

    cells_local = n[0] * n[1] * n[2]
    timed = {}
    if dom:
        avg, cnt = ctx.profile_read(dom)
        if cnt:
            timed[dom] = {"avg_ms": avg, "launches": cnt}
    ctx.profile_filter(None)
    cand = timed if timed else {p: v for p, v in phases.items() if p == dom}
    roofline = None
    if dom:
        ach = B_ALG_PHASE[dom] * cells_local / (cand[dom]["avg_ms"] * 1e-3)
        traffic, traffic_source = None, None
        try:   # HBM bytes per launch are NOT measured in this run: they come from the committed rocprofv3 --pmc passes
            tname = TRAFFIC_FILE
            tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
            if world == 1 and args.config == 2 and not args.size and args.tracers == 0 and dom in tj["kernels"]:
                traffic = tj["kernels"][dom]["traffic_bytes_per_launch"]
                traffic_source = f"profiles/{tname} (separate FETCH_SIZE / WRITE_SIZE passes of the same command)"
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": ach / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                    "alg_bytes_per_launch": B_ALG_PHASE[dom] * cells_local, "avg_launch_ms": cand[dom]["avg_ms"],
                    "launches_timed": cand[dom]["launches"], "measured_in": "timed region" if timed else "warm-up"}
    b_alg = alg_bytes_per_cell_update(args)
    step_frac = value / world * b_alg / HBM_PEAK
    copy_rate = None
    if rank == 0 and not args.rehearse_hostemu:
        try:
            copy_rate = ctx.copy_rate(1 << 30, 20)      # measured D2D copy ceiling of this GPU (read + write bytes / s)
        except Exception:   # noqa: BLE001
            copy_rate = None

    if rank == 0:
        out = {
            "metric": "cell-updates/sec per time_step!, 256^3 Nonhydrostatic WENO5", "value": value,
            "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps": spinup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": ("rehearsal on the host emulation of the kernels: plumbing check, NOT a measurement" if args.rehearse_hostemu else
                     "synthetic; REHEARSAL: ranks share GPUs and exchange through host shared memory -- not a scaling measurement"
                     if world > 1 and transport and transport.startswith("shm") else "synthetic"),
            "config": {"workload": (f"{Nglobal[0]}x{Nglobal[1]}x{Nglobal[2]} "
                                    + ("triply-periodic" if args.topology.upper() == "PPP" else f"topology {args.topology.upper()}")
                                    + " RectilinearGrid, "
                                    f"NonhydrostaticModel WENO5(zweno) + FFT Poisson, {args.stepper}, halo 3, {args.tracers} tracers"
                                    + (f", ScalarDiffusivity nu={args.nu}" if args.nu else ""))
                       if args.config == 2 else
                       (f"{Nglobal[0]}x{Nglobal[1]} (Periodic,Periodic,Flat) two_dimensional_turbulence, WENO5, RK3, "
                        "ScalarDiffusivity (BASELINE config 1)") if args.config == 1 else
                       (f"{Nglobal[0]}x{Nglobal[1]}x{Nglobal[2]} (Periodic,Periodic,Bounded) stretched z, WENO5, RK3, T+S, "
                        "FPlane, linear EOS, AMD, flux/gradient BCs, Fourier-tridiagonal Poisson (BASELINE config 3)"),
                       "decomposition": f"{'y' if args.config == 3 else 'z'}-slabs x{world}", "dt": dt, "init": args.init,
                       "local_size": list(n), "transport": transport,
                       "kernel_path": None if args.rehearse_hostemu else model.kernel_path},
            "roofline": roofline,
            "step_roofline": {"alg_bytes_per_cell_update": b_alg, "frac_of_hbm_peak": step_frac,
                              "measured_copy_rate_GBps": copy_rate / 1e9 if copy_rate else None,
                              "frac_of_measured_copy_rate": value / world * b_alg / copy_rate if copy_rate else None},
            "phases_ms_warmup": {k: round(v["avg_ms"], 4) for k, v in phases.items()},
            "max_abs_divergence": div,
            "step_graphs": ({"used": bool(use_graph), "steps_replayed_in_timed_region": model.graph_replays[0] - replays0}
                            if not args.rehearse_hostemu else None),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
