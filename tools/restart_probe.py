import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cylbench
R = int(sys.argv[1])
for m in (0, 40, 100):
    flow = cylbench.make_flow("cylinder2d", R)
    flow.setup(R)
    if m:
        flow.opts.inner_restart = m
    print("inner_restart", flow.opts.inner_restart, "inner_rel", flow.opts.inner_rel, "maxit", flow.opts.inner_maxit, flush=True)
    flow.synchronize(); t0 = time.time()
    flow.run_one_step(True)
    flow.synchronize(); dt = time.time() - t0
    nit, fg = flow.last_newton(); st = flow.last_stats()
    print(f"refinements {R} restart {m or 'default'}: {dt*1e3:.0f} ms, {nit} Newton, {fg} FGMRES, inner {st.inner_iters / max(st.precond_applies,1):.1f} per application", flush=True)
    flow.close()
