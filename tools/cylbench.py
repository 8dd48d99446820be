"""bench.py --workload cylinder2d | cylinder2d_scnsim | cylinder3d: the reference's cylinder test drivers on the C++ host mirror
(BASELINE configs 2 and 4; SURVEY 8d: "Configs 2/4 (2D cylinder): ... report DoF/s too"), and the extruded 3D cylinder of
Utils::GridCreator<3>::flow_around_cylinder (utilities.cpp:526-570) refined to bench scale -- the matrix-core cell kernel and its
fused scatter on an UNSTRUCTURED hexahedral mesh.

A "step" here is one Newton iteration (assemble + solve) inside InsIM::run_one_step / SUPGFluidSolver::run_one_step, as in the
headline metric; the time steps are the reference drivers' (reference .prm files, tests/golden/prm)."""
import ctypes as C
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = {
    "cylinder2d": dict(prm="fluid_cylinder_mpi.prm", vmax=0.374235, pmax=46.5226, driver="tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:93-95",
                       comment="'about 33 s' for the whole program (set-up + 1 time step + output), ctest default 2 ranks, hardware unspecified "
                               "(tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:6)", seconds=33.0),
    "cylinder2d_scnsim": dict(prm="fluid_cylinder_mpi_scnsim.prm", vmax=4.5, pmax=1.03544, driver="tests/fluid_cylinder_mpi_scnsim/fluid_cylinder_mpi_scnsim.cpp:85-87",
                              comment="'about 33 s' for the whole program, hardware unspecified (tests/fluid_cylinder_mpi_scnsim/fluid_cylinder_mpi_scnsim.cpp:6)",
                              seconds=33.0),
    "cylinder3d": dict(prm="fluid_cylinder_mpi.prm", vmax=None, pmax=None, driver="tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:98-104 (dim == 3 branch)",
                       comment="no constant and no timing exists for the 3D branch", seconds=None),
}


def prm_text(workload, refinements):
    prm = open(os.path.join(ROOT, "tests", "golden", "prm", REFERENCE[workload]["prm"])).read()
    prm = re.sub(r"set Global refinements\s*=\s*\d+", f"set Global refinements = {refinements}", prm)
    if workload == "cylinder3d":  # the dim == 3 branch of the driver: six no-slip boundaries, parabolic inflow in y and z
        prm = re.sub(r"set Dimension = 2", "set Dimension = 3", prm)
        prm = re.sub(r"set Gravity = 0.0, 0.0", "set Gravity = 0.0, 0.0, 0.0", prm)
        prm = re.sub(r"set Initial velocity = 0.0, 0.0", "set Initial velocity = 0.0, 0.0, 0.0", prm)
        prm = re.sub(r"set Number of Dirichlet BCs = 4", "set Number of Dirichlet BCs = 6", prm)
        prm = re.sub(r"set Dirichlet boundary id = 0, 2, 3, 4", "set Dirichlet boundary id = 0, 2, 3, 4, 5, 6", prm)
        prm = re.sub(r"set Dirichlet boundary components = 3, 3, 3, 3", "set Dirichlet boundary components = 7, 7, 7, 7, 7, 7", prm)
        prm = re.sub(r"set Dirichlet boundary values = 0.2, 0, 0, 0, 0, 0, 0, 0", "set Dirichlet boundary values = " + ", ".join(["0"] * 18), prm)
    return prm


def make_flow(workload, refinements, device=0):
    from openifem_amd import host
    prm = prm_text(workload, refinements)
    if workload == "cylinder2d_scnsim":
        dt = 1e-2
        flow = host.SCnsIM(prm, mesh="cylinder", device=device)
        # the pulse of the driver: active while time < 2 dt (fluid_cylinder_mpi_scnsim.cpp:44-62)
        flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10 and t < 2 * dt) else 0.0)
    elif workload == "cylinder3d":
        from cylmesh import inflow_bc_3d
        flow = host.InsIM(prm, mesh="cylinder", device=device)
        flow.add_hard_coded_boundary_condition(0, lambda p, c, t: inflow_bc_3d(p, c))
    else:
        flow = host.InsIM(prm, mesh="cylinder", device=device)
        flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0)
    return flow


def segments_per_cell(flow, n_sample=200, seed=1):
    """64-byte segments the fused A_uu scatter of k_ins_assemble3 touches per cell, replayed from the DEVICE's stored pattern
    (ifem_export_uu_pattern) with the lane mapping of assemble3.hip (tools/scatter_sim.py, slots "full": a matrix row's 27 blocks leave
    as the image of their memory, shifted to the row's alignment, 64 lanes per instruction).  Returns (segments per cell, layout
    floor of this row order, blocks per velocity row of the sample)."""
    from openifem_amd import capi
    L, ctx = flow.L, flow.ctx
    cu, _, _, _ = flow.cell_tables()
    rng = np.random.default_rng(seed)
    cells = rng.choice(len(cu), size=min(len(cu), n_sample), replace=False)
    nodes = np.unique(cu[cells])
    pat = {}
    for a in nodes:  # one small download per row: the sample touches a few thousand rows
        rp, col = capi.export_uu_pattern(L, ctx, int(a), 1)
        pat[int(a)] = (int(rp[0]), {int(b): k for k, b in enumerate(col)})
    tot = floor = 0
    e9 = 8 * np.arange(9)
    for c in cells:
        nd = cu[c]
        for a in nd:
            base, pos = pat[int(a)]
            o = np.sort(np.array([base + pos[int(b)] for b in nd], np.int64) * 72)
            floor += len(np.unique((o[:, None] + e9[None, :]) // 64))
            s0 = (o[0] // 8) & 7
            img = np.full(256 + 8, -1, np.int64)
            for k in range(27):
                img[s0 + 9 * k:s0 + 9 * k + 9] = o[k] + e9
            for rr in range(4):
                seg = img[64 * rr:64 * rr + 64]
                v = seg >= 0
                if v.any():
                    tot += len(np.unique(seg[v] // 64))
    row_len = np.array([len(pat[int(a)][1]) for a in nodes])
    return tot / len(cells), floor / len(cells), float(row_len.mean()), int(row_len.max())


def run(workload, refinements, steps=2, device=0, kernels=True):
    """set-up + the first time step (apply_nonzero_constraints = true, as run() does) + `steps` further time steps; returns the JSON
    object of the bench line"""
    from openifem_amd import capi
    ref = REFERENCE[workload]
    t0 = time.time()
    flow = make_flow(workload, refinements, device)
    flow.setup(refinements)
    flow.synchronize()
    t_setup = time.time() - t0
    n_cells, n_u, n_p = flow.sizes()
    n_dofs = n_u + n_p
    levels = len(flow.mg_levels())

    def one_step(first):
        flow.synchronize()
        t = time.time()
        flow.run_one_step(first)
        flow.synchronize()
        dt = time.time() - t
        nit, fg = flow.last_newton()
        return dict(ms=dt * 1e3, newton_iterations=nit, fgmres_iterations=fg, ms_per_newton_iteration=dt * 1e3 / max(nit, 1),
                    dofs_per_s=n_dofs * nit / dt)

    first = one_step(True)
    v, p = flow.get_current_solution()
    vmax, pmax = float(v.max()), float(p.max())
    later = [one_step(False) for _ in range(steps)]
    tot_ms = sum(s["ms"] for s in later)
    tot_it = sum(s["newton_iterations"] for s in later)
    st = flow.last_stats()
    out = {
        "metric": "DoF/s per Newton step (assemble+solve), " + {"cylinder2d": "2D cylinder Q2/Q1 mpi_insim", "cylinder2d_scnsim": "2D cylinder Q1/Q1 mpi_scnsim",
                                                                 "cylinder3d": "3D extruded cylinder Q2/Q1 mpi_insim"}[workload],
        "value": n_dofs * tot_it / (tot_ms * 1e-3) if tot_ms > 0 else first["dofs_per_s"], "unit": "DoF/s", "n_gpus": 1,
        "steps": tot_it, "warmup": first["newton_iterations"], "ms_per_step": tot_ms / max(tot_it, 1) if tot_it else first["ms_per_newton_iteration"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{workload}: {ref['driver']} on the host mirror, reference .prm with Global refinements = {refinements}",
                   "n_cells": n_cells, "n_dofs": n_dofs, "multigrid_levels_below": levels, "ainv_kind": int(flow.opts.ainv_kind), "setup_s": t_setup,
                   "first_time_step": first, "later_time_steps": later, "vmax": vmax, "pmax": pmax,
                   "last_solve": {"fgmres_iters": st.fgmres_iters, "precond_applies": st.precond_applies, "inner_iters": st.inner_iters,
                                  "cg_mp_iters": st.cg_mp_iters, "cg_sm_iters": st.cg_sm_iters},
                   "reference_timing_comment": ref["comment"]},
    }
    if ref["vmax"] is not None:
        rel = max(abs(vmax - ref["vmax"]) / ref["vmax"], abs(pmax - ref["pmax"]) / ref["pmax"])
        # the constants hold after the reference's single time step at the reference's refinement level
        out["config"]["reference_constants"] = {"vmax": ref["vmax"], "pmax": ref["pmax"], "tolerance": 1e-3,
                                                "applies": refinements == 3, "max_rel_deviation_after_first_step": rel,
                                                "met": bool(rel < 1e-3) if refinements == 3 else None}
    if ref["seconds"] and refinements == 3:
        out["config"]["whole_program_s_here"] = t_setup + first["ms"] * 1e-3
    if kernels:
        # one more time step under the per-kernel-family event log
        flow.kprof_begin()
        ks = one_step(False)
        prof = flow.kprof_end()
        rows = []
        for fam, e in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            ms = e["ms"] / max(ks["newton_iterations"], 1)
            rows.append({"family": fam, "launches_per_step": e["scopes"] / max(ks["newton_iterations"], 1), "ms_per_step": ms,
                         "algorithmic_bytes": e["bytes"] / max(ks["newton_iterations"], 1), "gb_s": e["bytes"] / max(e["ms"], 1e-9) / 1e6,
                         "hbm_frac": e["bytes"] / max(e["ms"], 1e-9) / 1e6 / 8000.0})
        out["roofline"] = {"kernels": rows, "profiled_time_step": ks,
                           "kernel_ms_sum_per_newton_iteration": sum(r["ms_per_step"] for r in rows)}
        asm = prof.get("assemble_cells")
        if asm and workload == "cylinder3d":
            ns_cell = asm["ms"] * 1e6 / (asm["scopes"] * n_cells)
            seg, floor, row_mean, row_max = segments_per_cell(flow)
            out["roofline"]["assemble3_unstructured"] = {
                "ns_per_cell": ns_cell, "ns_per_cell_morton_box_128": 86.45e6 / 128 ** 3, "segments_per_cell": seg, "segments_floor_of_the_row_order": floor,
                "segments_per_cell_morton_box": 871.0, "segment_rate_g_s": seg / ns_cell, "segment_peak_g_s": 24.0,
                "blocks_per_row_mean": row_mean, "blocks_per_row_max": row_max,
                "tflop_s": 27 * 729 * 53 / ns_cell / 1e3, "fp64_mfma_frac": 27 * 729 * 53 / ns_cell / 1e3 / 78.6,
                "note": "k_ins_assemble3<2> on the extruded cylinder (unstructured hexahedra, node order of the host mirror); segments from "
                        "the device's stored pattern (ifem_export_uu_pattern) replayed with the kernel's lane mapping on 200 random cells"}
    flow.close()
    return out


if __name__ == "__main__":
    import json
    w = sys.argv[1] if len(sys.argv) > 1 else "cylinder2d"
    r = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print(json.dumps(run(w, r)), flush=True)
