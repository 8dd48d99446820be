"""The cylinder benchmark (tests/fluid_cylinder_mpi) refined 5 times (0.87 M DoF) in one context and cut into strips on 2 / 4 virtual
ranks: solver counts of one time step.  On several ranks the refinement history hangs below the partitioned mesh as replicated
single-rank levels (host/insim.cpp::attach_nested_levels), S_m is applied as two SpMVs on the finest level.
    python tools/cyl_ranks.py [refinements] [mg_replica_cells: 0 = the round-3 path (no levels on several ranks)]"""
import ctypes as C
import os
import sys
import threading
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
sys.path.insert(0, os.path.join(os.path.dirname(here), "tests"))
import re  # noqa
from openifem_amd import capi, host  # noqa

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
REPL = int(sys.argv[2]) if len(sys.argv) > 2 else None
L = capi.load()


def run(partition=None):
    prm = open(os.path.join(os.path.dirname(here), "tests", "golden", "prm", "fluid_cylinder_mpi.prm")).read()
    prm = re.sub(r"set Global refinements\s*=\s*\d+", f"set Global refinements = {R}", prm)
    flow = host.InsIM(prm, mesh="cylinder")
    flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0)
    if partition is not None:
        flow.set_partition((partition[0], 1, 1), partition[1], local_world=partition[2])
    if REPL is not None:
        flow.set_mg_replica_cells(REPL)
    if REPL == 0 and partition is not None:
        flow.opts.inner_rel = 1e-3
        flow.opts.inner_maxit = 4000
    flow.setup(R)
    nl = len(flow.mg_levels())
    flow.synchronize(); t0 = time.time()
    flow.run_one_step(True)
    flow.synchronize(); dt = time.time() - t0
    st = flow.last_stats()
    v, p = flow.get_current_solution()
    if partition is not None:
        t = flow.partition_tables()
        v, p = v[:2 * t["n_unodes_owned"]], p[:t["n_pnodes_owned"]]
    cs = flow.comm_stats()
    out = dict(levels=nl, vmax=v.max(), pmax=p.max(), fgmres=st.fgmres_iters, applies=st.precond_applies,
               inner=st.inner_iters / max(st.precond_applies, 1), cg_sm=st.cg_sm_iters / max(st.precond_applies, 1),
               cg_mp=st.cg_mp_iters / max(st.precond_applies, 1), n=len(v) + len(p), s=dt, ex=cs["halo_exchanges"], vec=cs["allreduce_vec"])
    flow.close()
    return out


def show(tag, o, n=None):
    print(f"{tag}: {o['levels']} levels below, {n or o['n']} DoF, last solve of the step: FGMRES {o['fgmres']}, per application: inner {o['inner']:.1f}, "
          f"CG(S_m) {o['cg_sm']:.1f}, CG(M_p) {o['cg_mp']:.1f}; max|v| {o['vmax']:.6f} max p {o['pmax']:.4f}; step {o['s']*1e3:.0f} ms"
          + (f", {o['ex']} halo exchanges + {o['vec']} vector all-reduces since set-up" if o['ex'] else ""), flush=True)


one = run()
show("one context", one)
for world in (2, 4):
    w = C.c_void_p(L.ifem_local_world_create(world))
    out = [None] * world
    th = [threading.Thread(target=lambda r=r: out.__setitem__(r, run((world, r, w)))) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join()
    o = dict(out[0]); o["vmax"] = max(x["vmax"] for x in out); o["pmax"] = max(x["pmax"] for x in out)
    show(f"{world} virtual ranks", o, sum(x["n"] for x in out))
    L.ifem_local_world_destroy(w)
