"""Time ifem_ins_assemble at n^3 (kernel time from HIP events): python tools/asmbench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
s.setup(0); s.channel_state()
for _ in range(3):
    s.assemble(False)
    print("assemble kernel ms", round(s.timing().assemble_kernel_ms, 2), flush=True)
