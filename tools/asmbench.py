"""Time ifem_ins_assemble at n^3 (kernel time from HIP events): python tools/asmbench.py [n] [asm_skip]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
s.setup(0); s.channel_state()
if len(sys.argv) > 2:
    import ctypes as C
    from openifem_amd import capi
    t = capi.Tuning(); s.L.ifem_default_tuning(C.byref(t)); t.asm_skip = int(sys.argv[2])
    assert s.L.ifem_set_tuning(s.ctx, C.byref(t)) == 0
for _ in range(3):
    s.assemble(False)
    print("asm_skip", sys.argv[2] if len(sys.argv) > 2 else 0, "assemble kernel ms", round(s.timing().assemble_kernel_ms, 2), flush=True)
