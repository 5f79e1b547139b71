"""Multi-GPU plumbing: one process per GPU (torch.distributed.run), z-slab decomposition.

torch.distributed (gloo) is used only for the rendezvous -- broadcasting the RCCL unique id, barriers,
max-reductions of timings; every byte of field data moves through RCCL inside libocnhip.so
(csrc/comm.hip).  Mirrors Distributed/multi_architectures.jl:20-47 (`MultiArch(ranks=(1,1,R))`).
"""
import ctypes as C

from ._lib import check


def init_comm(ctx, dist, rank, world):
    """ocn_comm_init with a unique id created on rank 0 and broadcast over the (CPU) process group."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        raw = (C.c_char * 128)()
        check(ctx.lib.ocn_comm_unique_id(raw))
        buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=0)
    raw = (C.c_char * 128).from_buffer_copy(bytes(buf.numpy().tobytes()))
    check(ctx.lib.ocn_comm_init(ctx.h, int(rank), int(world), raw), ctx.h)


def init_comm_self(ctx):
    """one-rank communicator (tests on a one-GPU box, with OCNHIP_RCCL_SELF=1): forced slab runs then exchange with
    themselves through RCCL's grouped send / recv instead of plain device copies."""
    raw = (C.c_char * 128)()
    check(ctx.lib.ocn_comm_unique_id(raw))
    check(ctx.lib.ocn_comm_init(ctx.h, 0, 1, raw), ctx.h)


def init_comm_local(ctx, rank, world):
    """host-emulation ranks living in one process (tests): the id is ignored."""
    raw = (C.c_char * 128)()
    check(ctx.lib.ocn_comm_init(ctx.h, int(rank), int(world), raw), ctx.h)


def slab_of(global_array, rank, world):
    """z-slab of a global (Nx, Ny, Nz) array owned by `rank`"""
    nz = global_array.shape[2] // world
    return global_array[:, :, rank * nz:(rank + 1) * nz]
