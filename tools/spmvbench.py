"""Time y_u = A_uu x_u of the stored fp64 matrix at n^3: python tools/spmvbench.py [n] [uu_row_order]  (sweeps ifem_tuning::spmv_lanes / spmv_pipe;
uu_row_order 0: blocks of a row sorted by column, 1 (default): the assembly kernel's order)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
ORDER = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s.setup(0); s.channel_state()
L, ctx = s.L, s.ctx
t = capi.Tuning(); L.ifem_default_tuning(C.byref(t)); t.uu_row_order = ORDER
assert L.ifem_set_tuning(ctx, C.byref(t)) == 0  # the blocks are ordered when the first assembly allocates the values
s.assemble(False)
nnz = L.ifem_nnz(ctx, 0)
_, n_u, _ = s.sizes()
gb = (nnz * 76 + n_u / 3 * 8 + 2 * n_u * 8) / 1e9
for pipe, lanes in ((0, 32), (1, 8), (1, 16), (1, 32), (1, 64)):
    t = capi.Tuning(); L.ifem_default_tuning(C.byref(t)); t.spmv_pipe = pipe; t.spmv_lanes = lanes; t.uu_row_order = ORDER
    assert L.ifem_set_tuning(ctx, C.byref(t)) == 0
    for rep in range(2):
        s.synchronize(); t0 = time.time()
        for _ in range(10):
            assert L.ifem_uu_vmult(ctx, capi.VEC_UPDATE, capi.VEC_RHS, 0) == 0
        s.synchronize(); dt = (time.time() - t0) / 10
    print(f"uu_row_order {ORDER} spmv_pipe {pipe} spmv_lanes {lanes}: {dt*1e3:.3f} ms per apply = {gb/dt/1e3:.2f} TB/s", flush=True)
