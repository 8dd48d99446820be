"""Times the device-side FSI inputs (csrc/fsi.hip: FSI::update_indicator + FSI::find_fluid_bc, mpi_fsi.cpp:291-663 -- the
"Update indicator" and "Find fluid BC" timer sections of the reference) on the bench's 3D channel, with the CPU oracle's
restatement of the same two functions beside it on the same mesh and solid.

    python tools/fsibench.py [--cells 128] [--cpu-cells 64] [--solid 24,12,12]

Also used by bench.py (leg "fsi_inputs").  The counts (artificial cells, nodes inside the solid) do not depend on the cell
order, so the device run (Morton-ordered host mirror) and the oracle run (lexicographic BoxMesh) must agree on them: the
size-independent check of this path at bench scale."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_solid3d(reps=(24, 12, 12), lo=(0.62, 0.052, 0.047), hi=(1.03, 0.151, 0.149), angle=0.21):
    """a rotated Q1 block inside the channel [0,2] x [0,0.2]^2 with smooth nodal fields (deal.II vertex order)"""
    reps = tuple(reps)
    nv1 = [r + 1 for r in reps]
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in nv1], indexing="ij"), axis=-1).reshape(-1, 3)  # x slowest here
    vid = (idx[:, 0] + nv1[0] * (idx[:, 1] + nv1[1] * idx[:, 2]))
    pts = np.zeros((len(idx), 3))
    pts[vid] = np.asarray(lo) + idx * (np.asarray(hi) - np.asarray(lo)) / np.array(reps)
    c = pts.mean(axis=0)
    r = pts - c
    ca, sa = np.cos(angle), np.sin(angle)
    # a rotation about z by `angle` followed by one about x by angle / 3: no face stays axis-aligned
    x, y = ca * r[:, 0] - sa * r[:, 1], sa * r[:, 0] + ca * r[:, 1]
    cb, sb = np.cos(angle / 3), np.sin(angle / 3)
    y, z = cb * y - sb * r[:, 2], sb * y + cb * r[:, 2]
    pts = c + np.stack([x, y, z], axis=1)
    ci = np.stack(np.meshgrid(*[np.arange(n) for n in reps], indexing="ij"), axis=-1).reshape(-1, 3)
    cells = np.zeros((len(ci), 8), np.int32)
    for v in range(8):
        d = np.array([(v >> 0) & 1, (v >> 1) & 1, (v >> 2) & 1])
        j = ci + d
        cells[:, v] = j[:, 0] + nv1[0] * (j[:, 1] + nv1[1] * j[:, 2])
    vel = np.stack([0.02 + 0.1 * (pts[:, 1] - c[1]), -0.1 * (pts[:, 0] - c[0]), 0.01 * np.sin(5 * pts[:, 0])], axis=1)
    acc = np.stack([0.3 * np.cos(3 * pts[:, 0]), 0.2 * pts[:, 2], -0.1 * pts[:, 1]], axis=1)
    stress = np.stack([np.sin((k + 1) * pts[:, 0]) + 0.1 * k * pts[:, 1] for k in range(6)], axis=0)
    return {"vertices": np.ascontiguousarray(pts), "cells": cells, "velocity": np.ascontiguousarray(vel),
            "acceleration": np.ascontiguousarray(acc), "stress": np.ascontiguousarray(stress)}


def inside_solid3d(points, lo=(0.62, 0.052, 0.047), hi=(1.03, 0.151, 0.149), angle=0.21):
    """analytic inside test of make_solid3d's block: (strictly inside, distance to the nearest face in the block's frame)"""
    lo, hi = np.asarray(lo), np.asarray(hi)
    c = 0.5 * (lo + hi)
    r = np.asarray(points) - c
    cb, sb = np.cos(angle / 3), np.sin(angle / 3)
    y, z = cb * r[:, 1] + sb * r[:, 2], -sb * r[:, 1] + cb * r[:, 2]  # undo the rotation about x ...
    ca, sa = np.cos(angle), np.sin(angle)
    x, y = ca * r[:, 0] + sa * y, -sa * r[:, 0] + ca * y               # ... then the one about z
    q = np.stack([x, y, z], axis=1)
    gap = 0.5 * (hi - lo) - np.abs(q)
    return (gap > 0).all(axis=1), np.abs(gap).min(axis=1)


class _Solid:  # what orc.FsiSolid wants
    def __init__(self, d):
        self.dim, self.vertices, self.cells, self.bfaces = 3, d["vertices"], d["cells"], None
        self.velocity, self.acceleration, self.stress = d["velocity"], d["acceleration"], d["stress"]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_leg(L, ctx, capi, n_cells, solid, dt=1e-3, reps=3):
    """set_solid, update_indicator, find_fluid_bc (both modes) on an existing context; wall-clock around the synchronous
    C-ABI calls (each ends with a stream synchronise).  The Dirichlet-mode call comes last: it edits the constraint sets."""
    def chk(rc):
        if rc < 0:
            raise RuntimeError(L.ifem_last_error().decode())
    # fluid_solver.make_constraints() of every FSI step (:1191): the boundary lines the context holds, set again
    n_local = L.ifem_n_local_dofs(ctx)
    flags, vals = np.zeros(n_local, np.uint8), np.zeros(n_local)
    chk(L.ifem_get_constraints(ctx, 1, _ptr(flags), _ptr(vals)))
    bd = np.nonzero(flags)[0].astype(np.int32)
    bv = np.ascontiguousarray(vals[bd])
    t_mc = []
    for _ in range(reps):
        t0 = time.time()
        chk(L.ifem_set_constraints(ctx, 1, len(bd), _ptr(bd), _ptr(bv)))
        chk(L.ifem_set_constraints(ctx, 0, len(bd), _ptr(bd), None))
        t_mc.append(time.time() - t0)
    # find_solid_bc / update_solid_displacement (:727-760, :268-271): the fluid solution at the solid's vertices
    pts = np.ascontiguousarray(solid["vertices"])
    nv_s = len(pts)
    vals, stv, cl = np.zeros((nv_s, 4)), np.zeros((nv_s, 3, 3)), np.zeros(nv_s, np.int32)
    t_pt = []
    for _ in range(reps + 1):  # the first call builds the bins over the fluid cells
        t0 = time.time()
        chk(L.ifem_fsi_fluid_at_points(ctx, nv_s, _ptr(pts), _ptr(vals), _ptr(stv), _ptr(cl)))
        t_pt.append(time.time() - t0)
    s = capi.FsiSolid(len(solid["vertices"]), len(solid["cells"]), 0, _ptr(solid["vertices"]), _ptr(solid["cells"]), None,
                      _ptr(solid["velocity"]), _ptr(solid["acceleration"]), _ptr(solid["stress"]))
    t0 = time.time()
    chk(L.ifem_fsi_set_solid(ctx, C.byref(s)))
    t_set = time.time() - t0
    cnt = C.c_int64()
    t_ind = []
    for _ in range(reps + 1):
        t0 = time.time()
        chk(L.ifem_fsi_update_indicator(ctx, None, C.byref(cnt)))
        t_ind.append(time.time() - t0)
    st = capi.FsiStats()
    t_acc, t_dir = [], []
    for _ in range(reps + 1):
        t0 = time.time()
        chk(L.ifem_fsi_find_fluid_bc(ctx, dt, 0, None, C.byref(st)))
        t_acc.append(time.time() - t0)
    acc_stats = {"n_candidates": st.n_candidates, "n_inside": st.n_inside}
    t0 = time.time()
    chk(L.ifem_fsi_find_fluid_bc(ctx, dt, 1, None, C.byref(st)))
    t_dir.append(time.time() - t0)
    return {"solid_cells": len(solid["cells"]), "fluid_cells": int(n_cells), "boundary_lines": int(len(bd)),
            "set_constraints_x2_ms": float(np.median(t_mc)) * 1e3, "set_solid_ms": t_set * 1e3,
            "fluid_at_points": {"points": int(nv_s), "found": int((cl >= 0).sum()), "first_call_ms": t_pt[0] * 1e3,
                                "ms": float(np.median(t_pt[1:])) * 1e3},
            "update_indicator_ms": float(np.median(t_ind[1:])) * 1e3, "n_artificial_cells": cnt.value,
            "find_fluid_bc_ms": float(np.median(t_acc[1:])) * 1e3, **acc_stats,
            "find_fluid_bc_dirichlet_ms": t_dir[0] * 1e3, "dirichlet_candidates": st.n_candidates, "dirichlet_inside": st.n_inside,
            "dirichlet_lines": st.n_lines,
            "note": "wall clock of the synchronous C-ABI calls; fsi_stress + fsi_acceleration mode (use_dirichlet_bc = 0) and the "
                    "Dirichlet-line mode (merge into both constraint sets + their identity, compared on the device); "
                    "set_constraints_x2 = make_constraints of one FSI step (both sets re-made from their line lists)"}


def cpu_leg(n, solid, dt=1e-3):
    """the oracle's restatement on the lexicographic n^3 channel with the same solid (serial C, one host core)"""
    import orc
    from boxmesh import BoxMesh
    m = BoxMesh([n] * 3, (0, 0, 0), (2.0, 0.2, 0.2), kv=2)
    S = orc.FsiSolid(_Solid(solid))
    present = np.zeros(m.n_dofs)
    present[:m.n_u] = 0.01 * np.sin(np.arange(m.n_u))
    t0 = time.time()
    ind = orc.fsi_update_indicator(m, S)
    t1 = time.time()
    fs = np.zeros((6, m.n_unodes))
    acc, flag, val, nf = orc.fsi_find_fluid_bc(m, S, ind, dt, False, present, None, fs)
    t2 = time.time()
    return {"n": n, "fluid_cells": m.n_cells, "update_indicator_ms": (t1 - t0) * 1e3, "find_fluid_bc_ms": (t2 - t1) * 1e3,
            "n_artificial_cells": int(ind.sum()), "n_inside": int((fs[0] != 0).sum()), "n_not_found": int(nf), "cores": 1, "kind": "port"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=128)
    ap.add_argument("--cpu-cells", type=int, default=64, help="oracle sample (0 = skip); --cells itself checks the counts")
    ap.add_argument("--solid", default="24,12,12")
    args = ap.parse_args()
    from openifem_amd import capi, multigpu
    solver, reps, t_setup = multigpu.make_channel_solver(args.cells, 0, 1, 0, None, multigrid=False)
    solver.channel_state()
    n_cells, n_u, n_p = solver.sizes()
    solid = make_solid3d(tuple(int(v) for v in args.solid.split(",")))
    out = {"device": device_leg(solver.L, solver.ctx, capi, n_cells, solid)}
    if args.cpu_cells:
        out["cpu"] = cpu_leg(args.cpu_cells, solid)
        if args.cpu_cells == args.cells:
            assert out["cpu"]["n_artificial_cells"] == out["device"]["n_artificial_cells"], "indicator counts differ"
            assert out["cpu"]["n_inside"] == out["device"]["n_inside"], "inside-node counts differ"
            out["counts_agree"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
