"""A/B of the (removed, profiles/r06_fill_overlap.txt) ifem_tuning::auu_double_buffer at n^3; without that field it times the plain step: host-timed assemble / solve per step with a device fence after each.
python tools/fill_overlap_probe.py [n]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from openifem_amd import capi, multigpu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for db in (1, 0):
    solver, reps, _ = multigpu.make_channel_solver(n, 0, 1, 0, None, multigrid=True)
    tun = capi.Tuning()
    solver.L.ifem_default_tuning(C.byref(tun))
    if hasattr(tun, "auu_double_buffer"):
        tun.auu_double_buffer = db
    elif db:
        continue
    for c_ in solver.all_ctxs():
        assert solver.L.ifem_set_tuning(c_, C.byref(tun)) == 0
    solver.channel_state()
    for it in range(9):
        solver.synchronize()
        t0 = time.time()
        solver.assemble(False)
        t1 = time.time()
        solver.synchronize()
        t2 = time.time()
        solver.solve(False)
        solver.synchronize()
        t3 = time.time()
        print(f"double_buffer {db} step {it}: assemble call {1e3 * (t1 - t0):7.1f} ms (+ fence {1e3 * (t2 - t1):5.1f}), kernel {solver.timing().assemble_kernel_ms:6.1f}, solve {1e3 * (t3 - t2):7.1f} ms", flush=True)
    # unfenced pairs
    solver.synchronize()
    t0 = time.time()
    for it in range(5):
        solver.assemble(False)
        solver.solve(False)
    solver.synchronize()
    print(f"double_buffer {db}: {1e3 * (time.time() - t0) / 5:7.1f} ms per unfenced step", flush=True)
    solver.close()
