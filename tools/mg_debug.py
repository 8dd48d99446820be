"""debug aid: CG(S_m) counts with / without multigrid on one context and on virtual ranks"""
import ctypes as C
import sys
import threading
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from openifem_amd import capi, host

EXTENT = (2.0, 0.2, 0.2)


def run(n, P, mg, nu=2, ratio=4.0, ainv=3, nu_u=3, ratio_u=8.0, restart=16):
    L = capi.load()
    world = int(np.prod(P))
    depth = len(host.coarse_level_chain(n, P, EXTENT))
    worlds = [C.c_void_p(L.ifem_local_world_create(world)) for _ in range(depth + 1)] if world > 1 else None
    out = [None] * world

    def work(rank):
        reps = tuple(n[d] * P[d] for d in range(3))
        s = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), EXTENT)
        if worlds is not None:
            s.set_partition(P, rank, local_world=worlds[0])
            s.set_multigrid(True, 0, worlds[1:])
        s.setup(0)  # the C++ host mirror attaches the multigrid levels
        s.channel_state()
        if world > 1:
            L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL)
        s.opts.ainv_kind = ainv
        s.opts.mg_smooth_u = nu_u
        s.opts.mg_cheb_ratio_u = ratio_u
        s.opts.inner_restart = restart
        s.opts.sm_mg = mg
        s.opts.mg_smooth = nu
        s.opts.mg_cheb_ratio = ratio
        s.assemble(False)
        st = s.solve(False)
        import time
        t0 = time.time()
        st = s.solve(False)
        out[rank] = (st.fgmres_iters, st.precond_applies, st.cg_sm_iters, st.sm_mg_levels, st.cg_mp_iters, st.inner_iters,
                     round(st.t_ainv_ms, 1), round((time.time() - t0) * 1e3, 1))
        s.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    return out[0]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "uu":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
        print("n", n, "kind 3:", run((n, n, n), (1, 1, 1), 1), flush=True)
        for nu_u in (2, 3, 4):
            for ratio_u in (4.0, 8.0, 16.0):
                print("n", n, "kind 4 nu_u", nu_u, "ratio_u", ratio_u, run((n, n, n), (1, 1, 1), 1, ainv=4, nu_u=nu_u, ratio_u=ratio_u), flush=True)
        sys.exit(0)
    for n, P in (((16, 8, 8), (1, 1, 1)), ((8, 8, 8), (2, 1, 1)), ((16, 4, 8), (1, 2, 1)), ((16, 16, 16), (1, 1, 1)), ((8, 16, 16), (2, 1, 1)), ((8, 8, 8), (2, 2, 2))):
        for mg in (0, 1):
            print(n, P, "mg", mg, "fgmres, applies, cg_sm, levels, cg_mp, inner =", run(n, P, mg), flush=True)
