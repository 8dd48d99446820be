"""Can the A_uu assembly (bound by the memory-side atomic rate, SIMDs mostly idle) and the preconditioner's kernels (matrix-free
A_uu: VALU-bound; gathers: HBM reads) share the device?  Two contexts on one GPU, each with its own stream, driven from two host
threads: the n1^3 context assembles in a loop, the n2^3 context solves in a loop; rates alone and together.

    python tools/overlap_probe.py [n1] [n2]"""
import sys
import threading
import time

sys.path.insert(0, ".")
from openifem_amd import host, capi  # noqa

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def make(n):
    s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
    s.set_multigrid(True)
    s.setup(0)
    s.opts.ainv_kind, s.opts.inner_restart, s.opts.inner_rel = 4, 16, 1e-2
    s.channel_state()
    s.assemble(False); s.solve(False)
    return s


A, B = make(n1), make(n2)


def loop(fn, stop, out):
    k, t0 = 0, time.time()
    while not stop.is_set():
        fn(); k += 1
    out.append((k, time.time() - t0))


def rate(fn, seconds):
    stop, out = threading.Event(), []
    th = threading.Thread(target=loop, args=(fn, stop, out))
    th.start(); time.sleep(seconds); stop.set(); th.join()
    k, dt = out[0]
    return dt / k * 1e3


fa = lambda: A.assemble(False)
fb = lambda: B.solve(False)
ta, tb = rate(fa, 3.0), rate(fb, 3.0)
print(f"alone: assemble {n1}^3 {ta:.1f} ms, solve {n2}^3 {tb:.1f} ms")
stop, oa, ob = threading.Event(), [], []
t1 = threading.Thread(target=loop, args=(fa, stop, oa)); t2 = threading.Thread(target=loop, args=(fb, stop, ob))
t1.start(); t2.start(); time.sleep(5.0); stop.set(); t1.join(); t2.join()
ca, cb = oa[0][1] / oa[0][0] * 1e3, ob[0][1] / ob[0][0] * 1e3
print(f"together: assemble {ca:.1f} ms ({ca / ta:.2f} x), solve {cb:.1f} ms ({cb / tb:.2f} x); device time per (assemble + solve) pair if serial {ta + tb:.1f}, "
      f"concurrent throughput: {1 / (1 / ca + 0):.1f} ms per assemble while {ca / cb:.2f} solves ride along (= {ca / cb * tb:.1f} ms of solve work hidden)")
