import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from openifem_amd import host, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
s.setup(0); s.channel_state(); s.assemble(False)
L = s.L; ctx = s.ctx
for mode in sys.argv[2].split(","):  # "mf" (matrix-free), "f32", "f64" (stored matrix)
    var = 3
    if mode == "f32": var = 1
    if mode == "f64": var = 0
    for rep in range(2):
        t0 = time.time()
        for _ in range(10):
            rc = L.ifem_uu_vmult(ctx, capi.VEC_UPDATE, capi.VEC_RHS, var)
            assert rc == 0
        dt = (time.time() - t0) / 10
    print(f"mode {mode}: {dt*1e3:.3f} ms per apply", flush=True)
