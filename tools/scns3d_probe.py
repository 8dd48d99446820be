"""SCnsIM on a 3D Q1/Q1 box (n^3 cells, lid-type Dirichlet data): the block preconditioner in the reference's structure (scns_pc = 2) against
rounds 2-5's (scns_pc = 1).  python tools/scns3d_probe.py [n ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from boxmesh import BoxMesh
from openifem_amd import capi

for n in [int(a) for a in sys.argv[1:]] or [32]:
    m = BoxMesh((n, n, n), (0, 0, 0), (1.0, 1.0, 1.0), kv=1)
    dofs, vals = m.dirichlet({0: (7, [0.5, 0, 0]), 2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (7, [0, 0, 0]), 5: (7, [0, 0, 0])})
    for pc in (2, 1):
        ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
        t = capi.Tuning()
        ctx.L.ifem_default_tuning(C.byref(t))
        t.scns_pc = pc
        assert ctx.L.ifem_set_tuning(ctx.h, C.byref(t)) == 0
        ctx.set_constraints(0, dofs, None)
        ctx.set_constraints(1, dofs, vals)
        P = capi.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2)
        best = None
        for rep in range(2):
            ctx.scns_assemble(P, True)
            t0 = time.time()
            st = ctx.scns_solve(True)
            dt = (time.time() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        print(f"n {n} ({m.n_dofs} DoF) scns_pc {pc}: outer {st.fgmres_iters} inner {st.inner_iters} ({st.inner_iters / max(st.precond_applies, 1):.1f} per application) {best:.1f} ms per solve", flush=True)
        ctx.close()
