"""time_step leg of bench.py (a whole InsIM::run_one_step(true) Newton loop from the perturbed bench state) for several inner
tolerances of the A_uu^-1 replacement -- does a tighter inner solve in the later (pressure-dominated) Newton iterations buy outer
iterations (profiles/r04_newton_counts.txt: the reference's exact LU needs [3, 7, 1] where the 1e-2 inner solve needs [4, 8, 2])?
    python tools/newton_sweep.py [n] [inner_rel[:inner_rel_pressure] ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from openifem_amd import capi, host, multigpu  # noqa: E402
from cases import CHANNEL_KW  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
combos = sys.argv[2:] or ["1e-2", "3e-3", "1e-3", "1e-4"]
solver, reps, t_setup = multigpu.make_channel_solver(n, 0, 1, 0, None, multigrid=True, min_cells=0)
solver.opts.inner_restart = 16
solver.opts.ainv_kind = 4
solver.opts.inner_rel_first = 5e-5
L, ctx = solver.L, solver.ctx
P = capi.make_params(**CHANNEL_KW)
for c in combos:
    f = c.split(":")
    solver.opts.inner_rel = float(f[0])
    if len(f) > 1 and hasattr(solver.opts, "inner_rel_pressure"):
        solver.opts.inner_rel_pressure = float(f[1])
    for rep in range(2):  # the first repetition warms the caches of the setting
        solver.channel_state()
        assert L.ifem_vec_copy(ctx, capi.VEC_PRESENT, capi.VEC_EVAL) == 0
        log = np.zeros((16, 4))
        solver.synchronize()
        t0 = time.time()
        its = L.ifem_ins_newton_step(ctx, C.byref(P), C.byref(solver.opts), 1, 1e-6, 8, log.ctypes.data_as(C.c_void_p))
        solver.synchronize()
        dt = time.time() - t0
    print(f"n {n} inner_rel {c}: time_step {dt * 1e3:.0f} ms, Newton its {its}, FGMRES its {[int(v) for v in log[:max(its, 0), 2]]}, "
          f"rel residuals {[float('%.2e' % v) for v in log[:max(its, 0), 1]]}", flush=True)
