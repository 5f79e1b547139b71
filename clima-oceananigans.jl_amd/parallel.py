"""Multi-GPU plumbing: one process per GPU (torch.distributed.run), z-slab decomposition.

torch.distributed (gloo) is used only for the rendezvous -- broadcasting the RCCL unique id, barriers,
max-reductions of timings; every byte of field data moves through RCCL inside libocnhip.so
(csrc/comm.hip).  Mirrors Distributed/multi_architectures.jl:20-47 (`MultiArch(ranks=(1,1,R))`).
"""
import ctypes as C

from ._lib import check


def _broadcast_id(ctx, dist, rank):
    import torch
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        raw = (C.c_char * 128)()
        check(ctx.lib.ocn_comm_unique_id(raw))
        buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=0)
    return (C.c_char * 128).from_buffer_copy(bytes(buf.numpy().tobytes()))


def init_comm(ctx, dist, rank, world, allow_fallback=False, probe_timeout_s=None):
    """ocn_comm_init with a unique id created on rank 0 and broadcast over the (CPU) process group.

    Returns the transport in use: "rccl", "shm" (asked for with OCNHIP_TRANSPORT=shm, or the host-emulation build) or, with
    ``allow_fallback``, "shm (fallback: <RCCL error>)" when the RCCL communicator could not be created on some rank -- every
    rank then switches to the host shared-memory transport together, so that a node whose RCCL set-up is broken still
    produces (slow, clearly labelled) numbers instead of none.

    ncclCommInitRank is collective: a failure on one rank alone would leave the others in its bootstrap for ever.  So the
    ranks first PROBE (ocn_comm_probe: a throw-away non-blocking communicator polled against a deadline, which returns on
    every rank), agree on the outcome with an all_reduce over ``dist``, and only then create the real communicator from a
    second unique id -- or fall back together."""
    import os
    import torch
    raw = _broadcast_id(ctx, dist, rank)
    is_shm = bytes(raw.raw[:4]) == b"SHM:"
    if is_shm:
        check(ctx.lib.ocn_comm_init(ctx.h, int(rank), int(world), raw), ctx.h)
        return "shm"
    if probe_timeout_s is None:
        probe_timeout_s = float(os.environ.get("OCNHIP_COMM_TIMEOUT_S", "180"))
    rc = ctx.lib.ocn_comm_probe(ctx.h, int(rank), int(world), raw, float(probe_timeout_s))
    why = ctx.lib.ocn_last_error(ctx.h).decode(errors="replace") if rc else ""
    worst = torch.tensor([rc], dtype=torch.int64)
    dist.all_reduce(worst, op=dist.ReduceOp.MIN)          # error codes are negative
    if int(worst[0]) == 0:
        raw = _broadcast_id(ctx, dist, rank)              # the probe consumed the first id
        rc = ctx.lib.ocn_comm_init(ctx.h, int(rank), int(world), raw)
        why = ctx.lib.ocn_last_error(ctx.h).decode(errors="replace") if rc else ""
        worst = torch.tensor([rc], dtype=torch.int64)
        dist.all_reduce(worst, op=dist.ReduceOp.MIN)
        if int(worst[0]) == 0:
            return "rccl"
    if not allow_fallback:
        raise RuntimeError(f"RCCL communicator of {world} ranks could not be created: {why or 'another rank failed'}")
    os.environ["OCNHIP_TRANSPORT"] = "shm"
    raw = _broadcast_id(ctx, dist, rank)
    check(ctx.lib.ocn_comm_init(ctx.h, int(rank), int(world), raw), ctx.h)
    return f"shm (fallback: {why or 'another rank failed'})"


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
