"""inner iterations of the SCnsIM preconditioner per application on the refined cylinder mesh / a 3D box, over the
preconditioner of the inner GMRES on T_pp: python tools/tpp_sweep.py [cyl level | box n]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from openifem_amd import capi

kind = sys.argv[1] if len(sys.argv) > 1 else "cyl"
arg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if kind == "cyl":
    from cylmesh import CylinderMesh
    m = CylinderMesh(arg, kv=1)
    inflow = lambda p, c: 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0  # noqa: E731
    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
else:
    from boxmesh import BoxMesh
    m = BoxMesh((arg,) * 3, (0, 0, 0), (1.0, 1.0, 1.0), kv=1)
    dofs, vals = m.dirichlet({0: (7, [0.5, 0, 0]), 2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (7, [0, 0, 0]), 5: (7, [0, 0, 0])})
P = capi.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2)
print(kind, arg, "pressure rows", m.n_pnodes, flush=True)
combos = ((-1, 0, 0), (0, 950, 0), (1, 0, 0), (0, 950, 2), (0, 950, 4), (0, 950, 8), (0, 950, 16), (0, 0, 4), (0, 0, 8), (0, 1000, 8), (1, 0, 4), (1, 0, 8))
for order, milu, sweeps in combos:
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    t = capi.Tuning()
    ctx.L.ifem_default_tuning(C.byref(t))
    t.tpp_ilu_order, t.tpp_milu_permille, t.tpp_tri_sweeps = order, milu, sweeps
    assert ctx.L.ifem_set_tuning(ctx.h, C.byref(t)) == 0
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.scns_assemble(P, True)
    t0 = time.time()
    try:
        st = ctx.scns_solve(True)
        print(f"order {order} milu {milu} tri_sweeps {sweeps}: outer {st.fgmres_iters}, inner per application {st.inner_iters / max(st.precond_applies, 1):.1f}, "
              f"solve {time.time() - t0:.2f} s", flush=True)
    except Exception as e:
        print(f"order {order} milu {milu} tri_sweeps {sweeps}: FAILED {e}", flush=True)
    ctx.close()
