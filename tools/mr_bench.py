"""Cost of the partitioned code path, measured on ONE GPU: a Newton step of the 2n x n x n channel in one context, and
the same problem cut into two virtual ranks (in-process transport; both ranks share the GPU, so the total work is the
same).  The difference is the overhead of the multi-rank path: ghost cell layer, halo packing, host-synchronised dots,
the distributed S_m."""
import ctypes as C
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from openifem_amd import host, capi  # noqa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = 3


def configure(s):
    s.opts.ainv_kind = 3
    s.opts.inner_rel = 1e-2
    s.opts.inner_restart = 16  # as bench.py
    s.channel_state()


def timed(s, sync=None):
    s.assemble(False); s.solve(False)
    if sync: sync()
    t0 = time.time()
    its = None
    for _ in range(steps):
        s.assemble(False)
        its = s.solve(False)
    s.synchronize()
    return (time.time() - t0) / steps, its


s = host.InsIM(host.channel_prm(3), (2 * n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
s.setup(0)
configure(s)
t1, st = timed(s)
print(f"one context, {2*n}x{n}x{n}: {t1*1e3:.1f} ms/step, fgmres {st.fgmres_iters} cg_mp {st.cg_mp_iters} cg_sm {st.cg_sm_iters} inner {st.inner_iters}")
s.close()

L = capi.load()
w = C.c_void_p(L.ifem_local_world_create(2))
bar = threading.Barrier(2)
res = [None, None]


def work(rank):
    s = host.InsIM(host.channel_prm(3), (2 * n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
    s.set_partition((2, 1, 1), rank, local_world=w)
    s.setup(0)
    configure(s)
    L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL)
    res[rank] = timed(s, bar.wait)
    s.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
for t in th: t.start()
for t in th: t.join()
t2, st = res[0]
print(f"two virtual ranks of {n}^3 on one GPU: {t2*1e3:.1f} ms/step (one context: {t1*1e3:.1f}), fgmres {st.fgmres_iters} cg_mp {st.cg_mp_iters} cg_sm {st.cg_sm_iters} inner {st.inner_iters}")
