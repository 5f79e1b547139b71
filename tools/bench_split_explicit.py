"""Times split_explicit_free_surface_step! (Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface_kernels.jl:124-171)
at BASELINE config 5's horizontal size -- 1024 x 512 LatitudeLongitudeGrid, 128 levels, 200 substeps -- through libocnhip.so:
the reference's launch train (five launches per substep) against the fused two-launch train replayed from a hipGraph.
One JSON line; algorithmic bytes per substep and cell: kernel 1 reads eta, U, V, G^U, G^V, H^fc, H^cf and writes U, V (72 B),
kernel 2 reads eta, U, V and the three averages and writes eta and the averages (80 B)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("OCNHIP_LIB", None)
import __graft_entry__ as ge   # noqa: E402

ocn = ge.load_package()
H = ocn.hydrostatic
Nx, Ny, Nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 512, 128)
substeps = 200
grid = H.LatitudeLongitudeGrid(size=(Nx, Ny, Nz), longitude=(-180, 180), latitude=(-75, 75), z=(-4000, 0), halo=(3, 3, 3))
sefs = H.SplitExplicitFreeSurface(grid, substeps=substeps)
rng = np.random.default_rng(0)
sefs.eta.set(0.1 * rng.standard_normal((Nx, Ny)))
Gn = [H.HField(grid, ("Face", "Center", "Center")), H.HField(grid, ("Center", "Face", "Center"))]
Gm = [H.HField(grid, ("Face", "Center", "Center")), H.HField(grid, ("Center", "Face", "Center"))]
for f in Gn + Gm:
    f.set(1e-6 * rng.standard_normal(f.size))
ctx = grid.ctx
dt = 60.0
dtau = 2 * dt / substeps
out = {"workload": f"{Nx}x{Ny}x{Nz} LatitudeLongitudeGrid, SplitExplicitFreeSurface, {substeps} substeps (BASELINE config 5, free-surface part)"}
for name, fused in (("reference_launch_train", 0), ("fused_graph_train", 1), ("one_launch_graph_train", 2), ("four_substeps_per_launch_graph_train", 3)):
    for _ in range(3):
        sefs.substeps_train(dtau, 1, substeps, fused=fused)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        sefs.substeps_train(dtau, 1, substeps, fused=fused)
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    out[name] = {"ms_per_200_substeps": ms, "us_per_substep": ms * 1e3 / substeps,
                 "GB_per_s_at_152B": 152.0 * Nx * Ny * substeps / (ms * 1e-3) / 1e9,
                 "GB_per_s_at_128B": 128.0 * Nx * Ny * substeps / (ms * 1e-3) / 1e9,
                 "GB_per_s_at_50B": 50.0 * Nx * Ny * substeps / (ms * 1e-3) / 1e9}
for _ in range(2):
    sefs.step(Gn[0], Gn[1], Gm[0], Gm[1], dt, 0.1)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    sefs.step(Gn[0], Gn[1], Gm[0], Gm[1], dt, 0.1)
ctx.sync()
out["split_explicit_free_surface_step_ms"] = (time.perf_counter() - t0) / 5 * 1e3
out["graph_replays"] = sefs.graph_replays
out["finite"] = bool(np.isfinite(sefs.eta.parent()).all())
# one band of an 8-rank run with a BANDED free surface: Ny / 8 own rows + 2 W overlap rows, 200 substeps in blocks of W (the refresh of
# the overlap rows between blocks is the only thing missing here: one rank) -- the kernel side of DESIGN.md section 7's projection
out["band_subcycle"] = {}
for W in (10, 20, 25, 40):
    rows = Ny // 8 + 2 * W
    lat = 150.0 * rows / Ny
    bg = H.LatitudeLongitudeGrid(size=(Nx, rows, Nz), longitude=(-180, 180), latitude=(-lat / 2, lat / 2), z=(-4000, 0), halo=(3, 3, 3))
    bs = H.SplitExplicitFreeSurface(bg, substeps=substeps)
    bs.eta.set(0.1 * rng.standard_normal((Nx, rows)))

    def cycle():
        first = 1
        while first <= substeps:
            n = min(W, substeps - first + 1)
            bs.substeps_train(dtau, first, n, fused=3)
            first += n
    for _ in range(3):
        cycle()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        cycle()
    ctx.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    out["band_subcycle"][f"W={W}"] = {"rows": rows, "blocks": -(-substeps // W), "ms_per_200_substeps": ms, "graph_replays": bs.graph_replays}
print(json.dumps(out))
