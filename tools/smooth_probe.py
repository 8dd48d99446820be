"""A_uu V-cycle on the refined cylinder (grad-div term 100 x the viscous one): smoothing steps and Chebyshev interval of the V-cycle against the
inner iterations and the time of the first time step.   python tools/smooth_probe.py [refinements]"""
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
sys.path.insert(0, os.path.dirname(here))
import cylbench  # noqa

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for nu, ratio in ((2, 4.0), (3, 8.0), (4, 16.0), (6, 30.0), (8, 60.0)):
    flow = cylbench.make_flow("cylinder2d", R)
    flow.setup(R)
    flow.opts.mg_smooth_u = nu
    flow.opts.mg_cheb_ratio_u = ratio
    flow.synchronize()
    t0 = time.time()
    flow.run_one_step(True)
    flow.synchronize()
    dt = time.time() - t0
    nit, fg = flow.last_newton()
    st = flow.last_stats()
    v, p = flow.get_current_solution()
    print(f"refinements {R}, smoothing steps {nu}, interval ratio {ratio:g}: {dt * 1e3:.0f} ms, {nit} Newton / {fg} FGMRES, last solve {st.inner_iters / max(st.precond_applies, 1):.1f} inner per application; "
          f"vmax {v.max():.6f} pmax {p.max():.4f}", flush=True)
    flow.close()
