"""Multi-GPU bootstrap for bench.py: one process per GPU, block partition of the channel, RCCL unique id
broadcast over the torch.distributed (gloo) group that bench.py already holds for its barrier."""
import ctypes as C
import time

import numpy as np

from . import capi, host

PROC_GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def make_channel_solver(n, rank, world, local_rank, dist):
    import torch
    if world not in PROC_GRIDS:
        raise SystemExit(f"--gpus must be one of {sorted(PROC_GRIDS)}")
    P = PROC_GRIDS[world]
    reps = tuple(n * p for p in P)
    L = capi.load()
    uid = np.zeros(128, np.uint8)
    if rank == 0:
        rc = L.ifem_comm_unique_id(uid.ctypes.data_as(C.c_void_p))
        if rc < 0:
            raise RuntimeError("ifem_comm_unique_id failed")
    t = torch.from_numpy(uid)
    dist.broadcast(t, src=0)
    solver = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), (2.0, 0.2, 0.2), device=local_rank, verbose=False)
    solver.set_partition(P, rank, nccl_unique_id=uid)
    t0 = time.time()
    solver.setup(0)
    return solver, reps, time.time() - t0
