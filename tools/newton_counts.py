"""FGMRES iterations per Newton iteration of a whole InsIM::run_one_step(true) from the bench state (present := the perturbed
Poiseuille state) on the CPU oracle, with the reference's exact A_uu^-1 (scipy splu standing in for MUMPS, mpi_insim.cpp:124-127)
and with the oracle's iterative A_uu^-1 at the GPU's inner tolerance -- what VERDICT r3 item 3 asks for: does the REFERENCE
ALGORITHM need fewer outer iterations in the later (pressure-dominated) Newton iterations than the GPU's 1 / 6 / 3?
    python tools/newton_counts.py 8 12 16 [24]        (CPU only; 24^3 needs ~20 GB and several minutes for the LU)"""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import orc  # noqa: E402
from boxmesh import BoxMesh  # noqa: E402
from cases import channel3d_state  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [8, 12, 16]:
    m = BoxMesh([n] * 3, (0, 0, 0), (2.0, 0.2, 0.2), kv=2)
    dofs, vals, present, ev, kw = channel3d_state(m)
    for label, ainv, inner_rel in (("exact LU (reference: MUMPS)", "lu", None), ("inner GMRES to 1e-2 (block Jacobi)", None, 1e-2),
                                   ("inner GMRES to 1e-6", None, 1e-6)):
        S = orc.System(m)
        S.set_constraints(0, dofs, None)
        S.set_constraints(1, dofs, vals)
        if inner_rel is not None:
            S.opts.inner_rel = inner_rel
            S.opts.inner_restart = 30
            S.opts.inner_maxit = 4000
        S.opts.n_threads = os.cpu_count()
        x = ev.copy()
        t0 = time.time()
        rc, log = S.run_one_step(orc.make_params(**kw), True, x, ainv=orc.SpluAinv() if ainv else None)
        print(f"n {n:3d} {m.n_dofs:8d} DoF  {label:36s}: Newton its {rc}, FGMRES its {[int(v) for v in log[:, 2]]}, "
              f"rel. residuals {[float('%.2e' % v) for v in log[:, 1]]}  ({time.time() - t0:.0f} s)", flush=True)
