"""Multi-GPU bootstrap for bench.py: one process per GPU, block partition of the channel, RCCL unique id
broadcast over the torch.distributed (gloo) group that bench.py already holds for its barrier."""
import ctypes as C
import time

import numpy as np

from . import capi, host

PROC_GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def coarse_level_chain(cells_per_rank, P, extent, min_cells=4, max_levels=8):
    """Per-rank cell counts of the coarser multigrid levels of a box mesh: the directions with the smallest cells are
    halved until the cells are within 1.5x of isotropic (semi-coarsening for stretched cells), then all directions, as
    long as every halved direction keeps min_cells cells per rank.  Returns [(nx, ny, nz), ...], finest coarse level first."""
    n = list(cells_per_rank)
    dim = len(n)
    out = []
    for _ in range(max_levels):
        h = [extent[d] / (n[d] * P[d]) for d in range(dim)]
        hmin = min(h)
        go = [d for d in range(dim) if h[d] <= 1.5 * hmin and n[d] % 2 == 0 and n[d] // 2 >= min_cells]
        if not go or any(h[d] <= 1.5 * hmin and d not in go for d in range(dim)):
            break
        for d in go:
            n[d] //= 2
        out.append(tuple(n))
    return out


def attach_levels(make_solver, fine, cells_per_rank, P, extent, min_cells=4):
    """make_solver(reps_global, level) -> a set-up host solver of the same problem on the coarser box mesh; attaches the
    chain below `fine` and returns the list of coarse solvers (kept alive by the caller)."""
    chain, prev, levels = coarse_level_chain(cells_per_rank, P, extent, min_cells), fine, []
    for lev, n in enumerate(chain, start=1):
        reps = tuple(n[d] * P[d] for d in range(len(n)))
        s = make_solver(reps, lev)
        prev.attach_coarse(s)
        levels.append(s)
        prev = s
    return levels


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


def make_channel_solver(n, rank, world, local_rank, dist, multigrid=True, kind="InsIM"):
    """the bench's channel on `world` ranks (n^3 cells per rank) and, with multigrid, its chain of coarser levels: every
    level is a context of its own; all of them name the same unique id and therefore share one RCCL communicator
    (comm.hip keeps one per id and process)"""
    if world not in PROC_GRIDS:
        raise SystemExit(f"--gpus must be one of {sorted(PROC_GRIDS)}")
    P = PROC_GRIDS[world]
    extent = (2.0, 0.2, 0.2)
    uid = _unique_id(rank, dist) if world > 1 else None

    def make(reps, level):
        s = getattr(host, kind)(host.channel_prm(3), reps, (0, 0, 0), extent, device=local_rank, verbose=False)
        if world > 1:
            s.set_partition(P, rank, nccl_unique_id=uid)
        s.setup(0)
        return s

    reps = tuple(n * p for p in P)
    t0 = time.time()
    solver = make(reps, 0)
    solver._levels = attach_levels(make, solver, (n, n, n), P, extent) if multigrid else []
    return solver, reps, time.time() - t0
