"""Do the Chebyshev bounds of the A_uu V-cycle hold on the refined cylinder mesh?  One time step of tests/fluid_cylinder_mpi at the given
refinement with the default cold estimate (12 power steps) and with longer ones; prints the per-level estimates (verbose solver output on
stderr), the inner iterations per preconditioner application and the time per Newton iteration.
    python tools/cyl_eig_probe.py [refinements] [eig_steps,eig_steps,...]"""
import ctypes as C
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
sys.path.insert(0, os.path.dirname(here))
import cylbench  # noqa
from openifem_amd import capi  # noqa

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
STEPS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 40, 120]
W = sys.argv[3] if len(sys.argv) > 3 else "cylinder2d"
for es in STEPS:
    flow = cylbench.make_flow(W, R)
    flow.setup(R)
    tun = capi.Tuning()
    flow.L.ifem_default_tuning(C.byref(tun))
    tun.eig_steps = es
    for c in flow.all_ctxs():
        assert flow.L.ifem_set_tuning(c, C.byref(tun)) == 0
    flow.opts.verbose = 1
    flow.synchronize()
    t0 = time.time()
    flow.run_one_step(True)
    flow.synchronize()
    dt = time.time() - t0
    nit, fg = flow.last_newton()
    st = flow.last_stats()
    v, p = flow.get_current_solution()
    print(f"refinements {R}, eig_steps {es or 'default'}: {dt * 1e3:.0f} ms, {nit} Newton its, {fg} FGMRES its, last solve: inner {st.inner_iters / max(st.precond_applies, 1):.1f} per application, "
          f"CG(S_m) {st.cg_sm_iters / max(st.precond_applies, 1):.1f}; vmax {v.max():.6f} pmax {p.max():.4f}", flush=True)
    flow.close()
