"""The library stepped by several PROCESSES (one per rank) against the single-domain oracle.

CPU runs (`-m "not gpu"`): host emulation of the kernels + the host shared-memory transport of csrc/comm.hip, rendezvous
over gloo with world_size 2 (and 3 for the y-slabs) -- the control flow of bench.py's multi-rank leg end to end,
including `python bench.py --gpus 2` spawning its own ranks.
GPU runs (`-m gpu`): the same workers on libocnhip.so, both ranks on the one GPU of the test box, transport shm.  Only
the RCCL send/recv itself is then left to the multi-GPU node.
Reference tests mirrored: test/test_distributed_models.jl:361-453,500-517, test_distributed_poisson_solvers.jl:68-117.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(case, backend, world, timeout=900, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("OCNHIP_LIB", None)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, case, backend], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:          # exactly the children started here
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {case} failed:\n{o[-3000:]}"


CPU_CASES = [("zslab_ab2", 2, {}), ("zslab_ab2", 2, {"OCNHIP_DIST_SOLVER": "transpose"}), ("zslab_rk3_tracer", 2, {}),
             ("poisson", 2, {}), ("poisson", 2, {"OCNHIP_DIST_SOLVER": "transpose"}), ("yslab_amd", 2, {}),
             ("yslab_scalar", 3, {}),
             ("zslab_custom", 2, {"OCNHIP_OVERLAP": "1"}),     # 128 x 128 x 16 on two processes: the local w* term (two exchanges per step)
             ("hydro_bands:sphere", 2, {}), ("hydro_bands:periodic_box", 4, {}),   # the hydrostatic model on latitude bands, replicated free surface
             ("hydro_bands:sphere:3", 4, {})]                                      # ... and the free surface banded with 3 overlap rows


@pytest.mark.parametrize("case,world,env", CPU_CASES, ids=[f"{c}-{w}-{'-'.join(e.values()) or 'green'}" for c, w, e in CPU_CASES])
def test_library_across_processes_hostemu(case, world, env, backend):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    run_world(case, "hostemu", world, extra_env=env)


def test_bench_spawns_its_ranks_hostemu(backend):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself and reports n_gpus 2 (plumbing rehearsal on the
    host emulation: the number it prints is not a measurement and says so)."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OCNHIP_LIB"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-hostemu", "--steps", "2",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["decomposition"] == "z-slabs x2"
    assert "NOT a measurement" in rec["data"]
    assert rec["max_abs_divergence"] < 1e-10


def test_bench_config5_on_two_ranks_hostemu(backend):
    """`python bench.py --config 5 --gpus 2`: the hydrostatic model on two latitude bands with a banded free surface, ranks spawned by
    the bench itself (plumbing rehearsal on the host emulation)"""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OCNHIP_LIB"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", "2", "--rehearse-hostemu", "--steps", "2",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and "latitude bands x2, banded free surface" in rec["config"]["decomposition"]
    assert rec["finite"] and rec["max_abs_v"] < 0.1 and "NOT a measurement" in rec["data"]   # the temperature front drives a weak meridional flow


def test_bench_refuses_missing_gpus():
    """--gpus N on a node with fewer GPUs exits non-zero instead of silently measuring one GPU"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OCNHIP_TRANSPORT", "OCNHIP_BENCH_NDEV"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a node with 64 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0
    assert "n_gpus" not in out.stdout


@pytest.mark.gpu
def test_bench_falls_back_to_shm_when_rccl_cannot_start():
    """parallel.init_comm(allow_fallback=True): if the RCCL communicator cannot be created, every rank switches to the host
    shared-memory transport together and the bench line says so.  The failure is a real one: two ranks bound to the one GPU
    of the box, which RCCL refuses (`invalid usage`: duplicate device) -- seen by the non-blocking probe on both ranks."""
    env = dict(os.environ, OCNHIP_BENCH_NDEV="1", OCNHIP_COMM_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OCNHIP_LIB", "OCNHIP_TRANSPORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "64", "64",
                          "64", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["transport"].startswith("shm (fallback:")
    assert "REHEARSAL" in rec["data"] and rec["max_abs_divergence"] < 1e-10


@pytest.mark.gpu
def test_bench_config5_two_ranks_one_gpu():
    """`python bench.py --config 5 --gpus 2` with both ranks on the one GPU of the box (RCCL refuses the duplicate device, the ranks
    agree on the shared-memory transport): latitude bands, banded free surface, libocnhip.so"""
    env = dict(os.environ, OCNHIP_BENCH_NDEV="1", OCNHIP_COMM_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OCNHIP_LIB", "OCNHIP_TRANSPORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", "2", "--steps", "3", "--warmup", "1", "--size",
                          "128", "64", "16", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["transport"].startswith("shm") and "banded free surface" in rec["config"]["decomposition"]
    assert rec["finite"] and "REHEARSAL" in rec["data"]


GPU_CASES = [("zslab_ab2", {"OCNHIP_OVERLAP": "1"}), ("zslab_rk3_tracer", {"OCNHIP_DIST_SOLVER": "transpose", "OCNHIP_OVERLAP": "1"}),
             ("zslab_wide", {}), ("zslab_custom", {"OCNHIP_OVERLAP": "1"}), ("zslab_custom", {"OCNHIP_WSTAR_EXCHANGE": "1"}),
             ("zslab_custom", {"OCNHIP_PHI_EXCHANGE": "1"}),
             ("yslab_amd", {}), ("poisson", {}), ("hydro_bands:sphere", {}), ("hydro_bands:sector", {}),
             ("hydro_bands:sphere:4", {})]


@pytest.mark.gpu
@pytest.mark.parametrize("case,env", GPU_CASES, ids=[c + "".join("-" + k[7:].lower() for k in e) for c, e in GPU_CASES])
def test_library_two_ranks_one_gpu(case, env):
    """two processes, one MI355X, host shared-memory transport: slab kernels, pack / unpack, exchange order and the
    distributed solvers of the product library against the single-domain oracle"""
    run_world(case, "gpu", 2, extra_env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("case,env", [("zslab_custom", {"OCNHIP_OVERLAP": "1"}), ("zslab_rk3_tracer", {"OCNHIP_OVERLAP": "1"}),
                                      ("hydro_bands:sphere", {}), ("hydro_bands:sphere:3", {})],
                         ids=["zslab_custom", "zslab_rk3_tracer", "hydro_bands", "hydro_bands_banded"])
def test_library_four_ranks_one_gpu(case, env):
    """four processes on the one MI355X (the box allows six): the carries of the slab Poisson solve run over three other ranks and
    the slab's own periodic image, the halo ring has four members -- closer to config 4's eight than the two-rank runs"""
    run_world(case, "gpu", 4, extra_env=env)
