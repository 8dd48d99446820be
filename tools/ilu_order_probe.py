import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from openifem_amd import host, capi
prm = open(os.path.join(ROOT, 'tests', 'golden', 'prm', 'fluid_body_force_mpi.prm')).read()
def body_force(pt, component):
    return 1.0e3 / 1.3e-3 if (3.5 - 5e-4 < pt[0] < 4.5 + 5e-4 and component == 0) else 0.0
def sigma_pml(pt, component):
    s = 0.0
    for b in (0.0, 8.0):
        if abs(pt[0] - b) < 3.0:
            s = 340000 * ((3.0 - abs(pt[0] - b)) / 3.0) ** 4
    return s
for order, sweeps in ((0, 0), (1, 0), (0, 2), (0, 3)):
    flow = host.SCnsIM(prm, (160, 30), (0, 0), (8, 2))
    flow.set_body_force(body_force); flow.set_sigma_pml_field(sigma_pml)
    flow.setup(0)
    t = capi.Tuning(); flow.L.ifem_default_tuning(C.byref(t)); t.tpp_ilu_order = order; t.tpp_tri_sweeps = sweeps
    assert flow.L.ifem_set_tuning(flow.ctx, C.byref(t)) == 0
    flow.run_one_step(True)
    t0 = time.time(); inner = 0; app = 0
    for k in range(60):
        flow.run_one_step(False)
        st = flow.last_stats(); inner += st.inner_iters; app += st.precond_applies
    dt = (time.time() - t0) / 60
    _, p = flow.get_current_solution()
    print(f"order {order} tri_sweeps {sweeps}: {dt*1e3:.1f} ms per step, inner its per application {inner/max(app,1):.1f}, applications per step {app/60:.1f}, dp {p.max()-p.min():.3f}", flush=True)
    flow.close()
