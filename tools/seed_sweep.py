#!/usr/bin/env python
"""Robustness of the bench's solver configuration over the timed state (VERDICT r2, next #1c): for every seed x amplitude
of the perturbation of the evaluation point, one Newton step (assemble + solve) with and without the tight first inner
solve (ifem_solver_opts::inner_rel_first); prints outer / inner iteration counts, the recurrence residual FGMRES stopped
on and the TRUE residual ||b - A x|| / ||b|| recomputed with the assembled operator (ifem_true_residual).

    python tools/seed_sweep.py --cells 128 --seeds 5 --out gpurun_out/seed_sweep_128.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--amps", default="1e-4,1e-3,1e-2")
    ap.add_argument("--first", default="0,5e-5", help="values of inner_rel_first to compare")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from openifem_amd import multigpu
    solver, reps, _ = multigpu.make_channel_solver(args.cells, 0, 1, 0, None)
    solver.opts.inner_rel = 1e-2
    solver.opts.inner_restart = 16
    rows = []
    for amp in [float(v) for v in args.amps.split(",")]:
        for seed in range(1234, 1234 + args.seeds):
            for first in [float(v) for v in args.first.split(",")]:
                solver.opts.inner_rel_first = first
                solver.channel_state(seed=seed, rel=amp)
                solver.assemble(False)
                solver.synchronize()
                t0 = time.time()
                st = solver.solve(False)
                solver.synchronize()
                ms = (time.time() - t0) * 1e3
                r, b = solver.true_residual()
                row = {"cells": args.cells, "amp": amp, "seed": seed, "inner_rel_first": first, "fgmres_iters": st.fgmres_iters,
                       "inner_iters": st.inner_iters, "cg_sm_iters": st.cg_sm_iters, "cg_mp_iters": st.cg_mp_iters,
                       "fgmres_rel_residual": st.fgmres_res / b, "true_rel_residual": r / b, "solve_ms": ms,
                       "inner_first_tight": int(st.inner_first_tight)}  # 0 with inner_rel_first > 0: the option has backed off after a miss
                rows.append(row)
                print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")
    # summary: per (amp, first) the iteration counts over the seeds and the worst true residual
    print("# amp  inner_rel_first  fgmres_iters(per seed)  max true_rel_residual  median solve ms")
    for amp in sorted({r["amp"] for r in rows}):
        for first in sorted({r["inner_rel_first"] for r in rows}):
            sel = [r for r in rows if r["amp"] == amp and r["inner_rel_first"] == first]
            ms = sorted(r["solve_ms"] for r in sel)
            print(f"# {amp:g}  {first:g}  {[r['fgmres_iters'] for r in sel]}  {max(r['true_rel_residual'] for r in sel):.3e}  {ms[len(ms) // 2]:.1f}  tight used {[r['inner_first_tight'] for r in sel]}")


if __name__ == "__main__":
    main()
