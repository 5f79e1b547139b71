"""Times the hydrostatic AB2 step around its tendencies (ocn_hydro_step_after_tendencies: ab2_step!, barotropic correction,
store_tendencies!, update_state!) at BASELINE config 5's size on one MI355X -- 1024 x 512 x 128 LatitudeLongitudeGrid, T and S
with a linear equation of state, 200 barotropic substeps -- kernel by kernel as the reference issues it and with the merged passes.
One JSON line.  Field sweeps (one 3-D field read or written once; 8 B x Nx Ny Nz each): the sequence makes 40, the merged passes 28
(DESIGN.md section 7); `GB_per_s` prices the time outside the substep train at those counts."""
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
substeps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
grid = H.LatitudeLongitudeGrid(size=(Nx, Ny, Nz), longitude=(-180, 180), latitude=(-75, 75), z=(-4000, 0), halo=(3, 3, 3))
st = H.HydrostaticState(grid, tracers=("T", "S"), buoyancy=("TS", 9.80665, 1.67e-4, 7.8e-4, "T", "S"), substeps=substeps)
rng = np.random.default_rng(0)
for f, a in ((st.u, 0.1), (st.v, 0.1)):
    x = a * rng.standard_normal(f.size)
    if f is st.v:
        x[:, 0], x[:, -1] = 0, 0
    f.set(x)
st.tracers["T"].set(lambda x, y, z: 20 + 5e-3 * z + 0 * x + 0 * y)
st.tracers["S"].set(35.0)
for n in st.Gn:
    st.Gn[n].set((1e-6 if n in "uv" else 1e-8) * rng.standard_normal(st.Gn[n].size))
H.update_state(st)
ctx = grid.ctx
dt = 60.0
cells = Nx * Ny * Nz
out = {"workload": f"{Nx}x{Ny}x{Nz} LatitudeLongitudeGrid, HydrostaticFreeSurfaceModel step after the tendencies, T + S linear EOS, "
                   f"{substeps} substeps (BASELINE config 5, one GPU)"}
# the substep train alone, to separate it from the 3-D work
fs = st.free_surface
for _ in range(3):
    fs.substeps_train(2 * dt / substeps, 1, substeps, fused=3)
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    fs.substeps_train(2 * dt / substeps, 1, substeps, fused=3)
ctx.sync()
train = (time.perf_counter() - t0) / 10 * 1e3
out["substep_train_ms"] = train
for name, fused, sweeps in (("kernel_by_kernel", False, 40), ("merged_passes", True, 28)):
    for _ in range(3):
        H.time_step_after_tendencies(st, dt, 0.1, fused=fused)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        H.time_step_after_tendencies(st, dt, 0.1, fused=fused)
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    out[name] = {"ms_per_step": ms, "ms_outside_substeps": ms - train, "field_sweeps": sweeps,
                 "GB_per_s": sweeps * 8.0 * cells / ((ms - train) * 1e-3) / 1e9}
# the whole time_step!: tendencies (vector-invariant advection, spherical Coriolis, pressure gradient, centered tracer advection) + the above
st.set_physics("VectorInvariantEnstrophyConserving", ("HydrostaticSphericalCoriolis", 7.292115e-5, "EnstrophyConserving"), "CenteredSecondOrder")
st.u.set(lambda x, y, z: 10 * np.cos(np.pi * y / 180) + 0 * x + 0 * z)      # solid-body rotation: stays bounded however many steps are timed
st.v.set(0.0)
st.free_surface.eta.set(lambda x, y: -(6371.0e3 * 7.292115e-5 * 10 + 50) * np.sin(np.pi * y / 180) ** 2 / 9.80665 + 0 * x)
H.update_state(st)
H.time_step(st, dt, euler=True)
for _ in range(2):
    H.time_step(st, dt)
ctx.sync()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    H.time_step(st, dt)
ctx.sync()
ms = (time.perf_counter() - t0) / reps * 1e3
out["time_step"] = {"ms_per_step": ms, "cell_updates_per_s": cells / (ms * 1e-3), "ms_tendencies": ms - out["merged_passes"]["ms_per_step"]}
out["max_abs_v_after_steps"] = float(np.abs(st.v.interior()).max())
out["finite"] = bool(np.isfinite(st.u.parent()).all() and np.isfinite(st.pHY.parent()).all())
print(json.dumps(out))
