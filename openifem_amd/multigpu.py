"""Multi-GPU bootstrap for bench.py: one process per GPU, block partition of the channel, RCCL unique id
broadcast over the torch.distributed (gloo) group that bench.py already holds for its barrier."""
import ctypes as C
import time

import numpy as np

from . import capi, host

PROC_GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def _unique_id(rank, dist):
    """a fresh RCCL unique id from rank 0, broadcast over the (gloo) group bench.py already holds"""
    import torch
    L = capi.load()
    uid = np.zeros(128, np.uint8)
    if rank == 0:
        rc = L.ifem_comm_unique_id(uid.ctypes.data_as(C.c_void_p))
        if rc < 0:
            raise RuntimeError("ifem_comm_unique_id failed")
    t = torch.from_numpy(uid)
    dist.broadcast(t, src=0)
    return uid


def make_channel_solver(n, rank, world, local_rank, dist, multigrid=True, kind="InsIM", min_cells=0):
    """the bench's channel on `world` ranks (n^3 cells per rank) through the C++ host mirror.  With multigrid (the mirror's
    default on box meshes) InsIM<3>::initialize_system builds the chain of coarser levels itself (csrc/host/multigrid.cpp,
    insim.cpp::attach_multigrid_levels): every level is a context of its own; all of them name the same unique id and
    therefore share one RCCL communicator (comm.hip keeps one per id and process)"""
    if world not in PROC_GRIDS:
        raise SystemExit(f"--gpus must be one of {sorted(PROC_GRIDS)}")
    P = PROC_GRIDS[world]
    extent = (2.0, 0.2, 0.2)
    uid = _unique_id(rank, dist) if world > 1 else None

    reps = tuple(n * p for p in P)
    t0 = time.time()
    solver = getattr(host, kind)(host.channel_prm(3), reps, (0, 0, 0), extent, device=local_rank, verbose=False)
    if world > 1:
        solver.set_partition(P, rank, nccl_unique_id=uid)
    solver.set_multigrid(bool(multigrid), min_cells)
    solver.setup(0)
    return solver, reps, time.time() - t0
