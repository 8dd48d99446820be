"""SCnsIM block preconditioner variants on the cylinder mesh (first Newton iteration of tests/fluid_cylinder_mpi_scnsim):
python tools/scns_pc_sweep.py [refinements ...]   -- prints outer / inner iterations and ms per solve for every variant."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openifem_amd import capi
from cylmesh import CylinderMesh

VARIANTS = [("default", {}), ("reorth", dict(scns_inner_reorth=1)), ("exact", dict(pvv_sweeps=-1, b2pp_sweeps=-1)),
            ("s2/4", dict(pvv_sweeps=2, b2pp_sweeps=4)), ("s2/3", dict(pvv_sweeps=2, b2pp_sweeps=3)), ("s4/6", dict(pvv_sweeps=4, b2pp_sweeps=6)),
            ("s5/8", dict(pvv_sweeps=5, b2pp_sweeps=8)), ("nograph", dict(scns_graph=0)), ("right", dict(scns_inner_left=0)), ("left+re", dict(scns_inner_reorth=1)), ("legacy", dict(scns_pc=1))]
if os.environ.get("SCNS_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["SCNS_VARIANTS"].split(",")]


def inflow(p, c):
    return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0


for ref in [int(a) for a in sys.argv[1:]] or [3]:
    m = CylinderMesh(ref, kv=1)
    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    for name, kw in VARIANTS:
        ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
        t = capi.Tuning()
        ctx.L.ifem_default_tuning(C.byref(t))
        for k, v in kw.items():
            setattr(t, k, v)
        assert ctx.L.ifem_set_tuning(ctx.h, C.byref(t)) == 0
        ctx.set_constraints(0, dofs, None)
        ctx.set_constraints(1, dofs, vals)
        P = capi.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2)
        best = None
        for rep in range(3):
            ctx.scns_assemble(P, True)
            t0 = time.time()
            st = ctx.scns_solve(True)
            dt = (time.time() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        print(f"ref {ref} n_dofs {m.n_dofs} {name:8s}: outer {st.fgmres_iters:3d} inner {st.inner_iters:5d} ({st.inner_iters / max(st.fgmres_iters, 1):6.1f} per application)  {best:8.1f} ms per solve", flush=True)
        ctx.close()
