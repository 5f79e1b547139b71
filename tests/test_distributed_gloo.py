"""(Superseded as the multi-process check by tests/test_distributed_procs.py, which steps the LIBRARY on 2 - 3 processes; kept as an
independent NumPy cross-check of the slab algorithms.)
world_size-2 `gloo` tests (CPU): (a) the rendezvous the multi-GPU bench uses -- RCCL-id broadcast through
torch.distributed into ocn_comm_init; (b) the z-slab algorithm of csrc/comm.hip + poisson.hip restated with
NumPy + gloo collectives (halo exchange of contiguous planes, ky-block all-to-all, z-FFT, and back) against
the single-domain oracle.  The device kernels of the same algorithm are covered by test_distributed_hostemu.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["OCNHIP_LIB"] = os.path.join(ROOT, "tests", "hostemu", "libocnhip_hostemu.so")
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        import __graft_entry__ as ge
        ocn = ge.load_package()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # (a) rendezvous path of bench.py
        from importlib import import_module
        par = import_module("ocnhip.parallel")
        ctx = ocn.Context(0)
        par.init_comm(ctx, dist, rank, world)
        import ctypes as C
        r, n = C.c_int(), C.c_int()
        ctx.lib.ocn_comm_rank(ctx.h, C.byref(r), C.byref(n))
        assert (r.value, n.value) == (rank, world)

        def all_to_all(recv, send):
            """grouped send/recv to every peer -- exactly how csrc/comm.hip builds its all-to-all (gloo has none)"""
            reqs = []
            for q in range(world):
                if q == rank:
                    recv[q].copy_(send[q])
                else:
                    reqs.append(dist.isend(send[q], q, tag=50))
                    reqs.append(dist.irecv(recv[q], q, tag=50))
            [r.wait() for r in reqs]

        # (b) slab algorithm with gloo collectives
        import oracle as O
        from oracle import poisson
        N = (8, 6, 12)
        R = world
        nzl, nyl = N[2] // R, N[1] // R
        rng = np.random.default_rng(3)
        src = rng.random(N)
        src -= src.mean()
        og = O.RectilinearGrid(size=N, extent=(1, 2, 3), topology=("Periodic",) * 3)
        phi_ref = poisson.FFTBasedPoissonSolver(og).solve(src)
        mine = src[:, :, rank * nzl:(rank + 1) * nzl]
        spec = np.fft.rfft2(mine, axes=(1, 0)) if False else np.fft.fft(np.fft.rfft(mine, axis=0), axis=1)   # (Nxh, Ny, nzl)
        nxh = spec.shape[0]
        send = [torch.from_numpy(np.ascontiguousarray(spec[:, q * nyl:(q + 1) * nyl, :])) for q in range(R)]
        recv = [torch.empty_like(send[0]) for _ in range(R)]
        all_to_all(recv, send)
        col = np.concatenate([t.numpy() for t in recv], axis=2)          # (Nxh, nyl, Nz): ky-slab, z complete
        colh = np.fft.fft(col, axis=2)
        lx = poisson.poisson_eigenvalues(N[0], 1.0, "Periodic")[:nxh].reshape(-1, 1, 1)
        ly = poisson.poisson_eigenvalues(N[1], 2.0, "Periodic")[rank * nyl:(rank + 1) * nyl].reshape(1, -1, 1)
        lz = poisson.poisson_eigenvalues(N[2], 3.0, "Periodic").reshape(1, 1, -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            colh = -colh / (lx + ly + lz)
        if rank == 0:
            colh[0, 0, 0] = 0
        col = np.fft.ifft(colh, axis=2)
        send = [torch.from_numpy(np.ascontiguousarray(col[:, :, q * nzl:(q + 1) * nzl])) for q in range(R)]
        recv = [torch.empty_like(send[0]) for _ in range(R)]
        all_to_all(recv, send)
        spec2 = np.concatenate([t.numpy() for t in recv], axis=1)        # (Nxh, Ny, nzl)
        phi = np.fft.irfft(np.fft.ifft(spec2, axis=1), n=N[0], axis=0)
        err = np.abs(phi - phi_ref[:, :, rank * nzl:(rank + 1) * nzl]).max() / np.abs(phi_ref).max()
        assert err < 1e-12, err

        # halo exchange of contiguous planes with the ring neighbours (comm_halo_exchange_z)
        H = 3
        glob = rng.random((N[0] + 2 * H, N[1] + 2 * H, N[2]))            # x/y halos travel with the planes
        loc = np.zeros((glob.shape[0], glob.shape[1], nzl + 2 * H))
        loc[:, :, H:H + nzl] = glob[:, :, rank * nzl:(rank + 1) * nzl]
        up, dn = (rank + 1) % R, (rank - 1) % R
        top = torch.from_numpy(np.ascontiguousarray(loc[:, :, nzl:nzl + H]))
        bot = torch.from_numpy(np.ascontiguousarray(loc[:, :, H:2 * H]))
        rb, rt = torch.empty_like(top), torch.empty_like(bot)
        ops = [dist.P2POp(dist.isend, top, up), dist.P2POp(dist.irecv, rb, dn),
               dist.P2POp(dist.isend, bot, dn), dist.P2POp(dist.irecv, rt, up)]
        if R == 2:
            # same peer both ways: order the messages like the RCCL group does (first send <-> first recv)
            for o in (dist.isend(top, up, tag=0), dist.irecv(rb, dn, tag=0), dist.isend(bot, dn, tag=1), dist.irecv(rt, up, tag=1)):
                o.wait()
        else:
            [w.wait() for w in dist.batch_isend_irecv(ops)]
        loc[:, :, :H], loc[:, :, nzl + H:] = rb.numpy(), rt.numpy()
        idx = (np.arange(-H, nzl + H) + rank * nzl) % N[2]
        assert np.array_equal(loc, glob[:, :, idx])
        # (c) y-slab algorithm for a Bounded z (csrc/poisson.hip run_yslab): local x transform, all-to-all to kx bands with
        #     every y, local y transform + Thomas sweeps down z, and back; padded band when Nxh is not divisible by R
        Nb = (10, 8, 7)
        faces = np.array([0.0, 1, 2, 4, 7, 11, 16, 22])
        kwb = dict(x=(0, 1), y=(0, 2), z=faces, topology=("Periodic", "Periodic", "Bounded"))
        ogb = O.RectilinearGrid(size=Nb, **kwb)
        srcb = rng.random(Nb)
        dzc = np.diff(faces).reshape(1, 1, -1)
        srcb -= (srcb * dzc).sum() / (dzc.sum() * Nb[0] * Nb[1])
        fts = poisson.FourierTridiagonalPoissonSolver(ogb)
        ref_b = fts.solve_source(srcb)
        nyl_b = Nb[1] // R
        mine_b = (srcb * dzc)[:, rank * nyl_b:(rank + 1) * nyl_b, :]
        sx = np.fft.rfft(mine_b, axis=0)                                   # (Nxh, nyl, Nz)
        nxh_b = sx.shape[0]
        w = -(-nxh_b // R)                                                # kx columns per rank, padded
        pad = np.zeros((w * R, nyl_b, Nb[2]), dtype=complex)
        pad[:nxh_b] = sx
        send = [torch.from_numpy(np.ascontiguousarray(pad[q * w:(q + 1) * w])) for q in range(R)]
        recv = [torch.empty_like(send[0]) for _ in range(R)]
        all_to_all(recv, send)
        band = np.concatenate([t.numpy() for t in recv], axis=1)          # (w, Ny, Nz): my kx band, all y
        bh = np.fft.fft(band, axis=1)
        kx = np.minimum(np.arange(rank * w, (rank + 1) * w), nxh_b - 1)   # padding columns: any non-singular eigenvalue
        Dband = fts.D[kx]                                                  # diagonal of my kx rows (lx + ly inside)
        sol = poisson.thomas_batched(fts.lower, Dband, fts.lower, bh, Nb[2])
        band2 = np.fft.ifft(sol, axis=1)
        send = [torch.from_numpy(np.ascontiguousarray(band2[:, q * nyl_b:(q + 1) * nyl_b])) for q in range(R)]
        recv = [torch.empty_like(send[0]) for _ in range(R)]
        all_to_all(recv, send)
        spec_b = np.concatenate([t.numpy() for t in recv], axis=0)[:nxh_b]  # (Nxh, nyl, Nz)
        phi_b = np.fft.irfft(spec_b, n=Nb[0], axis=0)
        tot = torch.tensor([phi_b.sum()], dtype=torch.float64)
        dist.all_reduce(tot)
        phi_b -= float(tot[0]) / np.prod(Nb)                                # phi .-= mean(phi) over the global domain
        err = np.abs(phi_b - ref_b[:, rank * nyl_b:(rank + 1) * nyl_b]).max() / np.abs(ref_b).max()
        assert err < 1e-11, err
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + traceback.format_exc()))


def test_world_size_2_gloo():
    import torch.multiprocessing as mp
    if not os.path.exists(os.path.join(ROOT, "tests", "hostemu", "libocnhip_hostemu.so")):
        import __graft_entry__ as ge
        ge.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), res
