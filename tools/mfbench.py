"""Time the matrix-free A_uu (variant 3: fp64 cell arithmetic; 4: the inner solve's single-precision kernel) on the meshes of
the multigrid chain: python tools/mfbench.py [variant]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host, capi
import ctypes as C
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for reps in ((128, 128, 128), (128, 64, 64), (128, 32, 32), (128, 16, 16), (64, 8, 8)):
    s = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), (2.0, 0.2, 0.2))
    s.set_multigrid(False); s.setup(0); s.channel_state(); s.assemble(False)
    L, ctx = s.L, s.ctx
    for rep in range(2):
        s.synchronize(); t0 = time.time()
        for _ in range(20):
            assert L.ifem_uu_vmult(ctx, capi.VEC_UPDATE, capi.VEC_RHS, variant) == 0
        s.synchronize(); dt = (time.time() - t0) / 20
    nc = reps[0] * reps[1] * reps[2]
    print(f"variant {variant} {reps}: {dt*1e3:.3f} ms per apply (cell kernel + gather), {dt*1e9/nc:.2f} ns per cell", flush=True)
    s.close()
