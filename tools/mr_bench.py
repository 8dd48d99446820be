"""Cost of the partitioned code path, measured on ONE GPU: a Newton step of the 2n x n x n channel in one context, and
the same problem cut into two virtual ranks (in-process transport; both ranks share the GPU, so the total work is the
same), both in the bench configuration (multigrid levels attached, IFEM_AINV_MG).  The difference is the overhead of the
multi-rank path: ghost cell layer, halo packing, split launches, the distributed S_m -- plus what only the validation
transport pays (host barriers and synchronous copies where RCCL runs stream-ordered).

    python tools/mr_bench.py [n] [halo_overlap 0|1] [Px,Py,Pz] [mg_min_cells] [mg_replica_cells]   (default partition 2,1,1; n^3 cells per
    virtual rank; mg_replica_cells: coarse meshes up to this size are replicated per rank, 0 = all levels partitioned, default = the mirror's)"""
import ctypes as C
import sys
import threading
import time

sys.path.insert(0, ".")
from openifem_amd import host, capi, multigpu  # noqa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
PART = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 1, 1)
WORLD = PART[0] * PART[1] * PART[2]
steps = 3
EXTENT = (2.0, 0.2, 0.2)
L = capi.load()


MIN_CELLS = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # coarse-level policy: smallest number of cells per rank a halved direction keeps
REPLICA = int(sys.argv[5]) if len(sys.argv) > 5 else None


def hierarchy(cells, P, rank, worlds):
    """the channel through the C++ host mirror; its initialize_system attaches the multigrid levels"""
    reps = tuple(cells[d] * (P[d] if P else 1) for d in range(3))
    s = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), EXTENT, verbose=False)
    if P is not None:
        s.set_partition(P, rank, local_world=worlds[0])
    s.set_multigrid(True, MIN_CELLS, worlds[1:] if worlds else None)
    if REPLICA is not None:
        s.set_mg_replica_cells(REPLICA)
    s.setup(0)
    return s


def configure(s):
    s.opts.ainv_kind = 4
    s.opts.inner_rel = 1e-2
    s.opts.inner_restart = 16  # as bench.py
    tun = capi.Tuning()
    L.ifem_default_tuning(C.byref(tun))
    tun.halo_overlap = overlap
    for c in s.all_ctxs():
        assert L.ifem_set_tuning(c, C.byref(tun)) == 0
    s.channel_state()


def timed(s, sync=None):
    s.assemble(False); s.solve(False)
    if sync: sync()
    t0 = time.time()
    its = None
    s.comm_stats(reset=True)
    for _ in range(steps):
        s.assemble(False)
        its = s.solve(False)
    s.synchronize()
    return (time.time() - t0) / steps, its, s.comm_stats(), capi.comm_stats_levels(L, s.ctx), [r for r, _ in s.mg_levels()]


s = hierarchy(tuple(n * p for p in PART), None, 0, None)
configure(s)
t1, st = timed(s)[:2]
print(f"one context, {n*PART[0]}x{n*PART[1]}x{n*PART[2]}: {t1*1e3:.1f} ms/step, fgmres {st.fgmres_iters} cg_mp {st.cg_mp_iters} cg_sm {st.cg_sm_iters} inner {st.inner_iters}", flush=True)
depth = L.ifem_mg_depth(s.ctx)
s.close()

worlds = [C.c_void_p(L.ifem_local_world_create(WORLD)) for _ in range(depth + 1)]
bar = threading.Barrier(WORLD)
res = [None] * WORLD


def work(rank):
    s = hierarchy((n, n, n), PART, rank, worlds)
    configure(s)
    L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL)
    res[rank] = timed(s, bar.wait)
    s.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(WORLD)]
for t in th: t.start()
for t in th: t.join()
t2, st, cs, lv, reps = res[0]
print(f"{WORLD} virtual ranks of {n}^3 on one GPU (halo_overlap {overlap}): {t2*1e3:.1f} ms/step (one context: {t1*1e3:.1f}, {100*(t2/t1-1):+.1f} %), "
      f"fgmres {st.fgmres_iters} cg_mp {st.cg_mp_iters} cg_sm {st.cg_sm_iters} inner {st.inner_iters}; per step: "
      f"{cs['halo_exchanges'] / steps:.0f} halo exchanges, {cs['allreduce_dev'] / steps:.0f} stream-ordered + {cs['allreduce_host'] / steps:.0f} host-waited all-reduces, "
      f"{cs['allreduce_vec'] / steps:.0f} vector all-reduces over {cs['levels']} levels")
names = [f"{n*PART[0]}x{n*PART[1]}x{n*PART[2]}"] + ["x".join(str(v) for v in r) for r in reps]
for nm, k in zip(names, lv):
    print(f"    level {nm:>12} ({'partitioned' if k['nranks'] > 1 else 'replicated'}): {k['halo_exchanges'] / steps:6.0f} exchanges, "
          f"{(k['allreduce_dev'] + k['allreduce_host']) / steps:5.0f} scalar + {k['allreduce_vec'] / steps:3.0f} vector all-reduces per step")
