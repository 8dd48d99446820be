#!/usr/bin/env python
"""bench.py -- DoF/s for one Newton step (assemble + FGMRES solve) of the 3D Q2/Q1 channel (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one InsIM Newton iteration -- assemble(false) + solve(false), mpi_insim.cpp:436-437 -- at the timed state
of SURVEY 8(d): present = analytic plane Poiseuille, evaluation point = present + seeded 1e-3 perturbation, all
vectors and mesh tables resident in HBM before the timed region.  Weak scaling: every GPU owns an n^3 block of the
channel (n = 128 by default: config "3D channel flow 128^3 Q2/Q1" at N = 1, 256^3 at N = 8).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (MI355X_MICROARCH.md; 68-77 TF measured, profiles/r01_microbench.txt)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X FP32 vector FMA peak (MI355X_MICROARCH.md)


def pmc_traffic(kernel, n, variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[kernel][str(n)][variant]["traffic_bytes"]
    except Exception:
        return None


def kernel_models(L, ctx, n_cells, n_u, n_p, dim=3, nu=27, npn=8, cached_blocks=True):
    """Algorithmic bytes / flops per launch of the two heaviest kernels (DESIGN.md section 4 states the same figures).

    assembly (k_ins_assemble3): every stored matrix / vector value written once + per cell the mesh tables and the three
      nodal vectors it gathers (~46 kB per 3D Q2/Q1 cell); flops of the component-block form:
      per (node pair, point) 24 FMA + 5 mul, per (u-node, p-node, point) 1 + dim FMA (the latter and the B / B^T / M_p /
      diag(M_u) bytes only when those blocks are integrated: they are cached while the constraint set is unchanged).  The
      MFMA kernel executes 1.44x of that because 27 pads to 32 and K = 27 to 28; the padding is not counted as useful work.
    matrix-free A_uu (k_apply_uu_mf2<float>, the inner solve's operator): x, evaluation point (fp64 in HBM), constraint
      flags once per entry, the per-cell results in fp32 (two-stage scatter) + per cell vertex coordinates and node ids;
      flops of the sum-factorised passes + the quadrature-point stage, executed as fp32 vector FMAs.
    """
    nnz_uu, nnz_b, nnz_mp = L.ifem_nnz(ctx, 0), L.ifem_nnz(ctx, 1), L.ifem_nnz(ctx, 2)
    nd = nu * dim + npn
    # cached_blocks: the timed launches keep B, B^T, M_p and diag(M_u) of the warm-up assembly (same constraint set) and
    # write A_uu and the right-hand side only
    geo_vals = 0 if cached_blocks else 2 * nnz_b * dim + nnz_mp + n_u
    asm_bytes = 8.0 * (nnz_uu * dim * dim + geo_vals + n_u + n_p) + n_cells * (npn * dim * 8 + (nu + npn) * 4 + 3 * nd * 8)
    asm_flops = n_cells * nu * (nu * nu * (2 * 24 + 5) + (0 if cached_blocks else nu * npn * 2 * (1 + dim)))
    mf_bytes = n_u * (8 + 8 + 1) + n_cells * (npn * dim * 8 + nu * 4 + nu * dim * 4)
    n1 = round(nu ** (1.0 / dim))
    passes = (2 * dim) * dim * nu * n1 * 2 * 2 + dim * dim * nu * n1 * 2 * 2  # eval+grad of 2*dim fields, transposed grad+eval of dim fields
    mf_flops = n_cells * (passes + nu * 190)
    return {"asm": (asm_bytes, asm_flops), "mf": (mf_bytes, mf_flops)}


# what each kernel family of ifem_kprof is priced against: HBM for everything, plus the compute ceiling of the two kernels that
# have one (FP64 matrix cores for the cell integrals, FP32 vector FMA for the single-precision matrix-free cells)
FAMILY_KERNELS = {
    "assemble_cells": "k_ins_assemble3<2> (cell integrals + fused scatter)",
    "zero_fill": "__amd_rocclr_fillBufferAligned (system_matrix = 0, system_rhs = 0)",
    "spmv_uu": "k_spmv_uu_pipe (stored fp64 A_uu of the outer operator)",
    "spmv_b_bt": "k_spmv_planar<1,3,32,double> / <3,1,8,double> / k_spmv_planar_add (B, B^T)",
    "mf_cell": "k_apply_uu_mf2<3,2,4,true,float,*> (matrix-free A_uu, all levels)",
    "mf_gather": "k_mf_gather<3,float,*> (node gather + fused smoother update, all levels)",
    "spmv_sm": "k_spmv_planar<1,1,32,float> (S_m, all levels)",
    "spmv_mp": "k_spmv_planar<1,1,8,float> (M_p)",
    "mdot": "k_mdot<K> + k_reduce_final",
    "maxpy": "k_maxpy<K>",
    "vector_ops": "k_axpy / k_axpby / k_scale / copies / conversions / Chebyshev updates",
    "mg_transfer": "k_mg_csr_nodes / k_mg_csr / k_mg_inject",
    "smoother_setup": "k_uu_diag + k_block_invert + k_bjac_setup",
    "cg_recurrence": "k_cgd_init / k_cgd_update / k_cgd_p / k_cgd_scalars",
    "schur_setup": "k_schur_numeric, masked geometry blocks",
    "other": "constraints, hanging nodes",
    "tpp_ilu": "k_tpp_numeric, k_ilu_factor, k_ilu_solve / k_ilu_solve_batch, SpMV of T_pp (SCnsIM)",
}
COMPUTE_PEAK = {"assemble_cells": ("FP64 MFMA", FP64_PEAK_TFLOPS), "mf_cell": ("FP32 vector FMA", FP32_VECTOR_PEAK_TFLOPS)}


def pmc_family_traffic(n, world):
    """HBM bytes per step and family from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, "families"), single rank"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            fam = json.load(f)["families"][str(n)]
        return fam
    except Exception:
        return {}


def kernel_table(prof, prof_steps, n, world):
    """roofline.kernels[]: per kernel family of one profiled step (ifem_kprof: HIP event pairs on the context stream around
    every launch wrapper, read once after the step) launches, ms, algorithmic bytes / flops as the wrappers state them
    (DESIGN.md section 4), achieved GB/s and TFLOP/s against the peaks, and the committed PMC traffic with its ratio."""
    pmc = pmc_family_traffic(n, world)
    rows = []
    for fam, e in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        ms = e["ms"] / prof_steps
        if ms <= 0:
            continue
        nb, nf = e["bytes"] / prof_steps, e["flops"] / prof_steps
        row = {"family": fam, "kernel": FAMILY_KERNELS.get(fam, fam), "launches_per_step": e["scopes"] / prof_steps, "ms_per_step": ms,
               "algorithmic_bytes": nb, "algorithmic_flops": nf,
               "gb_s": nb / (ms * 1e-3) / 1e9, "hbm_frac": nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if fam in COMPUTE_PEAK and nf > 0:
            row.update({"tflop_s": nf / (ms * 1e-3) / 1e12, "compute_peak": COMPUTE_PEAK[fam][0],
                        "compute_frac": nf / (ms * 1e-3) / 1e12 / COMPUTE_PEAK[fam][1]})
        t = pmc.get(fam)
        if t:
            row.update({"traffic": t["traffic_bytes"], "traffic_over_algorithmic": t["traffic_bytes"] / nb if nb else None,
                        "traffic_source": t.get("source", "profiles/pmc_traffic.json") + ("" if world == 1 else " (single-rank pass)")})
        rows.append(row)
    return rows


def _r(v, sig=6):
    """floats to `sig` significant digits (the compact line is for a parser with a small capture, not for archiving)"""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and (d[k] is not None or k == "traffic")}


COMPACT_LIMIT = 4096  # bytes of the ONE stdout line (the driver's capture keeps ~8 kB of stdout: VERDICT r5 item 1)


def compact_line(out, detail_path=None):
    """The one stdout line of a run: the contract's keys, the few config entries that say what was timed, the dominant kernel's
    roofline and the CPU baseline -- everything else of `out` (kernel families, cold / sustained legs in full, cylinder and FSI
    legs, CPU scaling table) goes to bench_detail.json and to stderr.  Guaranteed < COMPACT_LIMIT bytes: optional entries are
    dropped from the end of the priority list until it fits."""
    cfg = out.get("config", {})
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    wl = str(cfg.get("workload", ""))
    line["config"] = dict({"workload": wl if len(wl) <= 200 else wl[:197] + "..."},
                          **_pick(cfg, ("n_dofs", "cells_per_gpu", "parallelism", "assemble_ms", "solve_ms", "assemble_kernel_ms", "fgmres_iters",
                                        "inner_iters", "cg_mp_iters", "cg_sm_iters", "t_cg_mp_ms", "t_cg_sm_ms", "t_ainv_ms", "true_rel_residual", "fgmres_rel_tol", "solver_opts",
                                        "rccl_nranks", "comm_transport", "halo_neighbors", "halo_exchanges_per_step", "allreduce_stream_per_step",
                                        "allreduce_host_per_step", "hbm_used_gb")))
    for k in ("value_cold", "value_sustained", "value_matrix_free"):
        if k in out:
            line[k] = out[k]
    ts = out.get("time_step", {})
    if "newton_iterations" in ts:
        line["time_step"] = _pick(ts, ("ms", "newton_iterations", "fgmres_iters"))
    roof = out.get("roofline")
    if roof:
        line["roofline"] = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "launches_timed",
                                        "algorithmic_bytes", "algorithmic_flops", "hbm_frac", "atomic_segment_frac",
                                        "kernel_ms_per_step"))
        if len(str(line["roofline"].get("kernel", ""))) > 120:
            line["roofline"]["kernel"] = line["roofline"]["kernel"][:117] + "..."
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "threads", "cpu_model", "cgroup_cpu_quota", "physical_cores",
                                          "limited_by", "sample", "reused_from"))
        if len(str(line["cpu_baseline"].get("sample", ""))) > 400:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:397] + "..."
    legs = {}
    for name, leg in (out.get("cylinder_workloads") or {}).items():
        if isinstance(leg, dict) and "value" in leg:
            legs[name] = {"value": leg["value"], "ms_per_step": leg["ms_per_step"]}
    if legs:
        line["side_legs"] = legs
    if detail_path:
        line["detail"] = detail_path
    line = _r(line)
    # optional entries, least important first, leave until the line fits
    for drop in (("side_legs",), ("roofline", "kernel_ms_per_step"), ("time_step",), ("config", "solver_opts"), ("cpu_baseline", "sample"),
                 ("config", "hbm_used_gb"), ("config", "comm_transport"), ("roofline", "kernel")):
        if len(json.dumps(line, separators=(",", ":"))) < COMPACT_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) < COMPACT_LIMIT, len(txt)
    return txt


def emit(out):
    """full record -> bench_detail.json (repo root, and gpurun_out/ when it exists) + stderr; compact record -> the ONE stdout line"""
    detail = "bench_detail.json"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, detail), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
    print("[bench detail] " + json.dumps(out), file=sys.stderr, flush=True)
    print(compact_line(out, detail), flush=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    """physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo), not hardware threads"""
    pairs, phys = set(), None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return len(pairs) or (os.cpu_count() or 1)


def cpu_allowance():
    """what the container may actually use of the host: cgroup CPU quota (cpu.max, in CPUs) and the affinity mask"""
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count()
    return quota, aff


def _cpu_step(n_cpu, threads):
    """One assemble + solve of the n_cpu^3 channel on the CPU oracle with one subdomain per thread (owner computes row, no
    atomics: oracle.c::orc_ins_assemble_subdomains, the shared-memory restatement of the reference's one-rank-per-core
    assembly, mpi_insim.cpp:206-209); returns (n_dofs, assemble s, solve s, FGMRES its)."""
    import orc
    from boxmesh import BoxMesh, block_partition
    from cases import channel3d_state
    m = BoxMesh([n_cpu] * 3, (0, 0, 0), (2.0, 0.2, 0.2), kv=2)
    dofs, vals, present, ev, kw = channel3d_state(m)
    try:
        S = orc.System(m, native=True)
    except Exception:
        S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    S.opts.inner_rel = 1e-2
    S.opts.inner_restart = 16
    S.opts.inner_maxit = 400
    S.opts.n_threads = threads
    P = orc.make_params(**kw)
    parts = threads
    while True:  # the largest block count <= threads the lattice can be cut into
        part, used = block_partition(m.reps, parts)
        if used == parts or parts == 1:
            break
        parts -= 1
    import resource
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.time()
    S.assemble_subdomains(P, False, ev, present, part, used, threads)
    t1 = time.time()
    rc, upd, it, res = S.solve(P, False)
    t2 = time.time()
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    # CPU seconds (user + system, all threads of the process) over wall seconds = the number of cores the step kept busy
    busy = ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) / max(t2 - t0, 1e-9)
    return m.n_dofs, t1 - t0, t2 - t1, it, busy


def np_log2(x):
    import math
    return math.log2(x)


def cpu_baseline(sizes, sweep_n=24, budget_s=100.0):
    """The CPU oracle (a port of the reference algorithm, oracle/oracle.c: dense per-cell Ke, CSR scatter, FGMRES with the
    block Schur preconditioner, same inner-solver settings as the GPU run) on a bounded sample of the same workload:
    one Newton step of the n^3 channel for every n in `sizes` (BASELINE.md section 3 plans n = 32 and 64) on ALL physical
    cores of the host, one subdomain per core as the reference runs one MPI rank per core.  A strong-scaling table
    16 -> all cores on a sweep_n^3 mesh goes into the line (threads, seconds, parallel efficiency against the smallest
    count)."""
    ncores = physical_cores()
    quota, affinity = cpu_allowance()
    # what the job may use: the affinity mask and the cgroup quota both cap it (the pool's boxes: quota 16 of 2 x 64 cores)
    allowed = min(x for x in (ncores, affinity, quota) if x)
    cand = sorted({t for t in (2, 4, 8, 16, 32, 64, 128, ncores) if t <= ncores and t <= 4 * allowed} | {max(1, int(allowed))})
    sweep = []
    for t in cand:
        nd, ta, ts, _, busy = _cpu_step(sweep_n, t)
        sweep.append({"threads": t, "assemble_s": ta, "solve_s": ts, "cores_busy": busy})
    base = sweep[0]
    for r in sweep:
        r["assemble_efficiency"] = base["assemble_s"] * base["threads"] / (r["assemble_s"] * r["threads"])
        r["step_efficiency"] = (base["assemble_s"] + base["solve_s"]) * base["threads"] / ((r["assemble_s"] + r["solve_s"]) * r["threads"])
    # the samples below run on the thread count that is FASTEST on this host (the table above is in the line: on the pool's
    # boxes the container does not get the whole socket pair to itself and more threads than ~32 run slower, whatever the
    # parallelisation -- the cell integrals share nothing)
    fastest = min(sweep, key=lambda r: r["assemble_s"] + r["solve_s"])["threads"]
    runs, skipped = [], []
    for n in sorted(sizes):
        # keep the default run bounded: a sample is skipped when the previous (smaller) one predicts more than `budget_s`
        # for it (cost grows ~ 5-6x per doubling of n at a fixed thread count: 13.6 s -> 63 s measured in round 2)
        if runs and budget_s > 0:
            prev = runs[-1]
            predicted = (prev["assemble_s"] + prev["solve_s"]) * 6.0 ** (np_log2(n / prev["n"]))
            if predicted > budget_s:
                skipped.append({"n": n, "predicted_s": predicted, "budget_s": budget_s})
                continue
        nd, ta, ts, it, busy = _cpu_step(n, fastest)
        runs.append({"n": n, "n_dofs": nd, "assemble_s": ta, "solve_s": ts, "fgmres_iters": it, "dofs_per_s": nd / (ta + ts),
                     "assemble_dofs_per_s": nd / ta, "solve_dofs_per_s": nd / ts, "cores_busy": busy})
    big = runs[-1]
    # `cores` = the cores the timed sample actually kept busy (CPU seconds / wall seconds, rounded up), never more than the threads
    # it ran or than the job is allowed; `threads` = the OpenMP threads it ran
    cores_used = int(min(fastest, max(1.0, -(-big["cores_busy"] // 1))))
    return {"value": big["dofs_per_s"], "unit": "DoF/s", "cores": cores_used, "threads": fastest, "cores_busy_measured": big["cores_busy"],
            "kind": "port", "cpu_model": _cpu_model(), "nproc": os.cpu_count(),
            "limited_by": ("cgroup CPU quota of the job (%.0f CPUs of %d physical cores), not the code: the cell loop shares nothing "
                           "between subdomains" % (quota, ncores)) if quota and quota < ncores else "host cores",
            "host_threads_available": os.cpu_count(), "physical_cores": ncores, "cgroup_cpu_quota": quota, "affinity_cpus": affinity,
            "omp_binding": os.environ.get("OMP_PLACES", "none"),
            "parallelisation": "one subdomain per core, owner computes row, no atomics (orc_ins_assemble_subdomains); OpenMP loops in the solve",
            "scaling_table": {"mesh": f"{sweep_n}^3", "rows": sweep}, "runs": runs, "skipped": skipped,
            "not_sampled": "128^3: the oracle's CSR with all couplings (as the reference's BlockSparsityPattern, mpi_fluid_solver.cpp:311-322) "
                           "needs ~0.25 TB there and ~10 min per step; the 64^3 step is the bounded sample",
            "sample": f"1 Newton step (assemble {big['assemble_s']:.2f}s + solve {big['solve_s']:.2f}s, FGMRES its "
                      f"{big['fgmres_iters']}) of the {big['n']}^3 Q2/Q1 channel ({big['n_dofs']} DoF), oracle/oracle.c with "
                      f"OpenMP on {fastest} threads keeping {big['cores_busy']:.1f} cores busy (CPU s / wall s; fastest of the scaling table; "
                      f"{ncores} physical cores, affinity mask {affinity}, cgroup quota {quota}) of {_cpu_model()}"}


def bench_insimex(args, host):
    """Side measurement: steady-state InsIMEX time step = rhs-only assembly + FGMRES to
    min(1e-9, 1e-8 ||rhs||) on the same channel; the matrix is assembled in the two warm-up steps as InsIMEX::run does."""
    n = args.n
    from openifem_amd import multigpu
    solver, _, _ = multigpu.make_channel_solver(n, 0, 1, int(os.environ.get("LOCAL_RANK", "0")), None, multigrid=bool(args.mg), kind="InsIMEX")
    n_cells, n_u, n_p = solver.sizes()
    # product defaults: what InsIMEX::initialize_system leaves in solver_opts (host/insim.cpp: one V-cycle as A~^-1, the outer
    # velocity block matrix-free); the flags are experiment overrides
    if args.ainv is not None:
        solver.opts.ainv_kind = args.ainv
    if args.inner_rel is not None:
        solver.opts.inner_rel = args.inner_rel
    if args.inner_maxit is not None:
        solver.opts.inner_maxit = args.inner_maxit
    if args.inner_restart:
        solver.opts.inner_restart = args.inner_restart
    args.ainv, args.inner_rel = solver.opts.ainv_kind, solver.opts.inner_rel
    if args.mg_smooth_u is not None:
        solver.opts.mg_smooth_u = args.mg_smooth_u
    solver.opts.verbose = args.verbose
    solver.channel_state()
    # start from the perturbed state (the unperturbed Poiseuille flow is a fixed point: its rhs is rounding noise)
    from openifem_amd import capi
    assert solver.L.ifem_vec_copy(solver.ctx, capi.VEC_PRESENT, capi.VEC_EVAL) == 0
    solver.run_one_step(True, True)
    solver.run_one_step(False, True)
    if args.outer_mf is not None:
        solver.opts.outer_matrix_free = args.outer_mf
    if args.sm_rel is not None:
        solver.opts.sm_rel = args.sm_rel
    if args.mp_rel is not None:
        solver.opts.mp_rel = args.mp_rel
    solver.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        solver.run_one_step(False, False)
    solver.synchronize()
    dt = (time.time() - t0) / args.steps
    st = solver.last_stats()
    # one more step under the per-kernel-family event log: where the time step goes
    solver.kprof_begin()
    solver.run_one_step(False, False)
    prof = solver.kprof_end()
    fam = {k: v["ms"] for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0.05}
    emit({"metric": "DoF/s per InsIMEX time step (rhs assembly + solve), 3D Q2/Q1", "value": (n_u + n_p) / dt,
          "unit": "DoF/s", "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": dt * 1e3,
          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
          "config": {"workload": f"3D channel flow {n}^3 Q2/Q1, mpi_insimex steady-state time step", "n_dofs": n_u + n_p,
                     "ainv_kind": args.ainv, "inner_rel": args.inner_rel, "inner_maxit": solver.opts.inner_maxit, "mg_smooth_u": solver.opts.mg_smooth_u,
                     "outer_matrix_free": solver.opts.outer_matrix_free, "cg_mp_rel": solver.opts.mp_rel, "cg_sm_rel": solver.opts.sm_rel,
                     "multigrid_levels": 1 + len(solver.mg_levels()),
                     "fgmres_iters": st.fgmres_iters, "cg_mp_iters": st.cg_mp_iters, "cg_sm_iters": st.cg_sm_iters,
                     "inner_iters": st.inner_iters, "t_cg_mp_ms": st.t_cg_mp_ms, "t_cg_sm_ms": st.t_cg_sm_ms, "t_ainv_ms": st.t_ainv_ms,
                     "t_solve_ms": st.t_total_ms},
          "roofline": {"kernel_ms_per_step": fam, "kernel_ms_sum": sum(v["ms"] for v in prof.values())}})


def extras(solver, capi, n_dofs, warm_ms):
    """N = 1 side measurements next to the warm `value`:
    new_constraint_set_step -- one Newton iteration right after the set of constrained dofs changed (ifem_tuning::geo_cache = 2
      treats every assembly that way): masked copies of the mesh-only B / B^T, S_m re-formed, nothing re-integrated.
    cold_step -- one Newton iteration with nothing kept from the previous one: B, B^T, M_p, diag(M_u) re-integrated and
      S_m = B diag(M_u)^-1 B^T re-formed, which is what the reference does in EVERY iteration (assemble() zeroes all blocks,
      mpi_insim.cpp:163-165; solve() rebuilds the preconditioner, :369 and :44-49).  Here it is the cost of the first
      iteration after the set of constrained dofs changed (every FSI step), ifem_tuning::geo_cache = 0.
    time_step -- a whole InsIM::run_one_step(apply_nonzero_constraints = true) Newton loop (mpi_insim.cpp:416-473,
      tolerance 1e-6, first iteration with nonzero_constraints, the others with zero_constraints) from the bench state.
      Both AffineConstraints objects of make_constraints list the same dofs, so the cached blocks survive the switch."""
    import numpy as np
    from cases import CHANNEL_KW
    out = {}
    L, ctx = solver.L, solver.ctx
    tun = capi.Tuning()
    L.ifem_default_tuning(C.byref(tun))

    def set_geo_cache(v):  # on every multigrid level: the coarser ones re-form their S_m as well
        tun.geo_cache = v
        for c_ in solver.all_ctxs():
            assert L.ifem_set_tuning(c_, C.byref(tun)) == 0

    def one_step(label, note):
        solver.assemble(False)
        solver.solve(False)  # state: previous iteration done
        solver.synchronize()
        t0 = time.time()
        solver.assemble(False)
        t1 = time.time()
        st = solver.solve(False)
        solver.synchronize()
        t2 = time.time()
        out[label] = {"ms_per_step": (t2 - t0) * 1e3, "value": n_dofs / (t2 - t0), "unit": "DoF/s",
                      "assemble_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "fgmres_iters": st.fgmres_iters,
                      "vs_warm": (t2 - t0) * 1e3 / warm_ms, "note": note}

    set_geo_cache(0)
    one_step("cold_step", "geometry blocks re-integrated and S_m re-formed inside the step, as the reference does every "
                          "Newton iteration; `value` above keeps them (same constrained-dof set)")
    set_geo_cache(2)
    one_step("new_constraint_set_step", "what a change of the constrained-dof set costs here (every FSI step): B / B^T as masked "
                                        "copies of the unconstrained blocks (integrated once per mesh), S_m re-formed on every level")
    set_geo_cache(1)
    # matrix_free_step (never `value`): the same Newton iteration with ifem_tuning::stored_uu = 0 -- no A_uu values are
    # written or read: the cell kernel integrates the right-hand side, the outer operator applies A_uu matrix-free in fp64
    try:
        tun.stored_uu = 0
        for c_ in solver.all_ctxs():
            assert L.ifem_set_tuning(c_, C.byref(tun)) == 0
        one_step("matrix_free_step", "ifem_tuning::stored_uu = 0: A_uu never stored (rhs-only cell kernel, fp64 matrix-free outer operator, node blocks of "
                                     "the smoother from the cell integrals); same stopping rule, residual below recomputed with the matrix-free operator "
                                     "(equal to the block CSR to 1e-13: tests/test_gpu_multigrid.py)")
        res, bn = solver.true_residual()
        out["matrix_free_step"]["true_rel_residual"] = res / bn if bn > 0 else None
    except Exception as e:  # a side measurement
        out["matrix_free_step"] = {"error": repr(e)}
    tun.stored_uu = 1
    for c_ in solver.all_ctxs():
        assert L.ifem_set_tuning(c_, C.byref(tun)) == 0
    solver.assemble(False)  # the block CSR again for the legs below
    solver.channel_state()
    # present := perturbed state, so that the Newton loop has something to converge from
    assert L.ifem_vec_copy(ctx, capi.VEC_PRESENT, capi.VEC_EVAL) == 0
    log = np.zeros((16, 4))
    P = capi.make_params(**CHANNEL_KW)  # the parameters of host.channel_prm
    solver.synchronize()
    t0 = time.time()
    its = L.ifem_ins_newton_step(ctx, C.byref(P), C.byref(solver.opts), 1, 1e-6, 8, log.ctypes.data_as(C.c_void_p))
    solver.synchronize()
    dt = time.time() - t0
    if its > 0:
        out["time_step"] = {"ms": dt * 1e3, "newton_iterations": int(its), "ms_per_newton_iteration": dt * 1e3 / its,
                            "dofs_per_s_per_iteration": n_dofs * its / dt, "rel_residuals": [float(v) for v in log[:its, 1]],
                            "fgmres_iters": [int(v) for v in log[:its, 2]],
                            "note": "InsIM::run_one_step(true): Newton loop to 1e-6 from the bench state"}
        # the same loop once more under the per-kernel-family event log (from the same state): where the sustained figure goes
        try:
            solver.channel_state()
            assert L.ifem_vec_copy(ctx, capi.VEC_PRESENT, capi.VEC_EVAL) == 0
            solver.kprof_begin()
            its2 = L.ifem_ins_newton_step(ctx, C.byref(P), C.byref(solver.opts), 1, 1e-6, 8, None)
            prof = solver.kprof_end()
            if its2 == its:
                out["time_step"]["kernel_ms_per_loop"] = {k: v["ms"] for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0.05}
                out["time_step"]["kernel_ms_sum"] = sum(v["ms"] for v in prof.values())
        except Exception as e:  # a side measurement
            out["time_step"]["kernel_ms_error"] = repr(e)
    else:
        out["time_step"] = {"error": L.ifem_last_error().decode()}
    return out


def _hbm_used_gb():
    """device memory in use (hipMemGetInfo: total - free), GB; None when the runtime is not reachable"""
    try:
        hip = C.CDLL("libamdhip64.so")
        free, total = C.c_size_t(0), C.c_size_t(0)
        if hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0:
            return (total.value - free.value) / 1e9
    except OSError:
        pass
    return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in their environment, what torch.distributed.run --nnodes=1 --nproc-per-node N would give them),
    pass their output through, and return the first non-zero exit code (the others are ended: a rank that died would leave
    its peers waiting in a collective)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # exactly the processes started above
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", dest="n", type=int, default=128, help="cells per direction per GPU")
    ap.add_argument("--cpu-cells", dest="cpu_n", default="32,64", help="cells per direction of the CPU baseline sample(s), comma separated (0 = skip; BASELINE.md section 3 plans 32 and 64: 14 s and 63 s on the 64 threads the sweep picks)")
    ap.add_argument("--extras", type=int, default=1, help="N = 1 only: also measure cold_step (geometry blocks and S_m rebuilt, as the reference does every iteration) and time_step (a whole run_one_step Newton loop)")
    ap.add_argument("--inner-rel", type=float, default=None, help="relative residual target of the inner A_uu solve inside the preconditioner (the reference applies an exact LU there, mpi_insim.cpp:124-127)")
    ap.add_argument("--inner-rel-first", type=float, default=None, help="experiment (default: what InsIM::initialize_system sets, 5e-5 with multigrid levels): the same for the FIRST preconditioner application of a solve (0: as --inner-rel).  Measured at 128^3 (profiles/r02_inner_sweep.txt): with 1e-2 the outer FGMRES needs two iterations (relative residual 9.1e-4 after the first), with 5e-5 in the first application one (7.8e-5 <= 1e-4) for the same four inner iterations in total: 275 -> 239 ms per step; later applications (later Newton iterations need 3-5 outer iterations whatever the inner accuracy) keep the cheap setting")
    ap.add_argument("--ainv", type=int, default=None, help="IFEM_AINV_* kind of the A_uu^-1 replacement (4 = matrix-free operator + geometric multigrid V-cycle, 3 = matrix-free inner operator + block Jacobi, 1 = fp32 inner matrix, 0 = fp64 matrix)")
    ap.add_argument("--sm-rel", type=float, default=None, help="experiment: relative tolerance of CG(S_m) inside the preconditioner (reference and default: 1e-3)")
    ap.add_argument("--mp-rel", type=float, default=None, help="experiment: relative tolerance of CG(M_p) inside the preconditioner (reference and default: 1e-6)")
    ap.add_argument("--inner-restart", type=int, default=None, help="restart length of the inner GMRES of the A_uu^-1 replacement (measured at 128^3: 8/10/12/15/20/30/45 -> 557/533/537/518/522/531/543 ms per step; 16 = one multi-dot pass of the single-precision basis; the library default is 30)")
    ap.add_argument("--tuned", type=int, default=1, help="also time the relaxed-preconditioner variant (reported as tuned_preconditioner, N = 1 only)")
    ap.add_argument("--mg", type=int, default=1, help="1 (default): attach the chain of coarser (semi-coarsened) box meshes so that CG(S_m) inside the preconditioner is multigrid-preconditioned; 0: plain CG as in the reference")
    ap.add_argument("--inner-maxit", type=int, default=None, help="cap of the inner A_uu iterations (--ainv 4: 0 = exactly one V-cycle)")
    ap.add_argument("--mg-smooth-u", type=int, default=None, help="smoothing steps of the A_uu V-cycle (--ainv 4)")
    ap.add_argument("--mg-post-u", type=int, default=None, help="smoothing steps after the coarse correction of the A_uu V-cycle (default: as before it)")
    ap.add_argument("--mg-ratio-u", type=float, default=None, help="Chebyshev interval ratio of the A_uu V-cycle (--ainv 4)")
    ap.add_argument("--tune", action="append", default=[], help="experiment: ifem_tuning field=value (e.g. --tune spmv_pipe=0), applied to every multigrid level")
    ap.add_argument("--fsi", type=int, default=32, help="N = 1 only: also time the device-side FSI inputs (FSI::update_indicator + find_fluid_bc, csrc/fsi.hip) on the bench mesh with a 24x12x12-cell solid, and the oracle's restatement on the CPU at this many cells per direction (0 = skip the leg)")
    ap.add_argument("--seed", type=int, default=1234, help="seed of the perturbation of the timed state")
    ap.add_argument("--rel", type=float, default=1e-3, help="relative amplitude of the perturbation of the timed state")
    ap.add_argument("--mg-min-cells", type=int, default=0, help="multigrid levels: a direction is halved only while it keeps this many cells per rank (0: the host mirror's default, 4)")
    ap.add_argument("--fgmres-rel", type=float, default=None, help="test aid: relative tolerance of the outer FGMRES (reference and default: 1e-4)")
    ap.add_argument("--dump-update", default=None, help="test aid: every rank writes its owned entries of the last Newton update and their global lattice ids to <path>.rank<r>.npz")
    ap.add_argument("--verbosity", dest="verbose", type=int, default=0)
    ap.add_argument("--outer-mf", type=int, default=None, help="experiment: apply the u-u block of the OUTER operator matrix-free too")
    ap.add_argument("--workload", default="channel3d", choices=["channel3d", "cylinder2d", "cylinder2d_scnsim", "cylinder3d"],
                    help="channel3d (default): the BASELINE metric's 3D channel; cylinder2d / cylinder2d_scnsim: BASELINE configs 2 and 4, the "
                         "reference's cylinder drivers on the host mirror with the reference .prm; cylinder3d: the extruded cylinder of "
                         "GridCreator<3>::flow_around_cylinder at --refinements (3: 0.43 M cells, 11 M DoF) -- tools/cylbench.py")
    ap.add_argument("--refinements", type=int, default=3, help="global refinements of the cylinder workloads (the reference's .prm: 3)")
    ap.add_argument("--cylinder-legs", type=int, default=1, help="N = 1 only: append the cylinder workloads (configs 2 and 4, and the 3D "
                                                                 "cylinder at 3 refinements) to the channel line as side measurements")
    ap.add_argument("--solver", default="insim", choices=["insim", "insimex"],
                    help="insim (default, the BASELINE metric): one Newton iteration of MPI::InsIM; insimex: one steady-state "
                         "time step of MPI::InsIMEX (rhs-only assembly + solve), reported as a side measurement")
    args = ap.parse_args()

    # the CPU baseline leg pins one OpenMP thread per physical core (read by libgomp when the oracle library loads) -- only when
    # the job owns the machine: with an affinity mask or a cgroup quota smaller than the core count, places outside the allowance
    # would stack the threads on a few CPUs
    _quota, _aff = cpu_allowance()
    if min(x for x in (physical_cores(), _aff, _quota) if x) >= physical_cores():
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly (`python bench.py --gpus N`, the way the driver starts --gpus 1): become the launcher of N ranks
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "IFEM_BENCH_DEVICE" in os.environ:  # debugging aid: several ranks on one GPU (RCCL normally refuses this)
        local_rank = int(os.environ["IFEM_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it plainly (it launches its own ranks) or "
                         f"with torch.distributed.run --nproc-per-node {args.gpus}")
    dist = None
    if world > 1:
        import torch.distributed as dist  # rendezvous + barrier + max-over-ranks only (gloo on the host)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        print(f"[bench rank {rank}] gloo rendezvous of {world} ranks complete", file=sys.stderr, flush=True)

    from openifem_amd import host, capi
    # one rank per GPU: every rank needs its own device (the reference: one MPI rank per core, tests/CMakeLists.txt:52,71)
    n_dev = capi.load().ifem_device_count()
    need = 1 if "IFEM_BENCH_DEVICE" in os.environ else max(world, 1)
    if n_dev < need:
        msg = (f"[bench rank {rank}] IFEM_E_NODEVICE: {n_dev} HIP device(s) visible, --gpus {args.gpus} needs {need} "
               f"(one rank per GPU, no CPU fallback)")
        print(msg, file=sys.stderr, flush=True)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(capi.E_NODEVICE_EXIT)
    if args.workload != "channel3d":
        if world != 1:
            raise SystemExit("the cylinder workloads run on one GPU (the host mirror cuts unstructured meshes into strips: tools/cyl_ranks.py)")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cylbench
        emit(cylbench.run(args.workload, args.refinements, steps=max(args.steps, 1), device=local_rank))
        return
    if args.solver == "insimex":
        return bench_insimex(args, host)
    n = args.n
    from openifem_amd import multigpu
    solver, reps, t_setup = multigpu.make_channel_solver(n, rank, world, local_rank, dist, multigrid=bool(args.mg), min_cells=args.mg_min_cells)
    n_cells, n_u, n_p = solver.sizes()
    n_dofs_global = solver.global_dofs() if world > 1 else n_u + n_p
    # The timed configuration is the PRODUCT default: whatever InsIM<3>::initialize_system leaves in solver_opts (host/insim.cpp:
    # V-cycle inner solver, restart 16, inner_rel 1e-2, inner_rel_first 5e-5 when multigrid levels attach) -- the flags below are
    # experiment overrides and the line names every one that was used (config.solver_opts_overridden)
    overridden = [k for k in ("inner_rel", "inner_rel_first", "inner_restart", "ainv", "sm_rel", "mp_rel", "fgmres_rel", "inner_maxit",
                              "mg_smooth_u", "mg_post_u", "mg_ratio_u") if getattr(args, k) is not None] + (["outer_mf"] if args.outer_mf is not None else []) + (["tune"] if args.tune else []) + ([] if args.mg else ["mg=0"])
    if args.inner_rel is not None:
        solver.opts.inner_rel = args.inner_rel
    if args.inner_rel_first is not None:
        solver.opts.inner_rel_first = args.inner_rel_first
    if args.inner_restart:
        solver.opts.inner_restart = args.inner_restart
    if args.sm_rel is not None:
        solver.opts.sm_rel = args.sm_rel
    if args.mp_rel is not None:
        solver.opts.mp_rel = args.mp_rel
    if args.ainv is not None:
        solver.opts.ainv_kind = args.ainv
    if args.fgmres_rel is not None:
        solver.opts.fgmres_rel = args.fgmres_rel
    if args.inner_maxit is not None:
        solver.opts.inner_maxit = args.inner_maxit
    if args.mg_smooth_u is not None:
        solver.opts.mg_smooth_u = args.mg_smooth_u
    if args.mg_post_u is not None:
        solver.opts.mg_smooth_u_post = args.mg_post_u
    if args.mg_ratio_u is not None:
        solver.opts.mg_cheb_ratio_u = args.mg_ratio_u
    if args.outer_mf is not None:
        solver.opts.outer_matrix_free = args.outer_mf
    solver.opts.verbose = args.verbose if rank == 0 else 0
    if args.tune:
        tun = capi.Tuning()
        solver.L.ifem_default_tuning(C.byref(tun))
        for kv in args.tune:
            k, v = kv.split("=")
            setattr(tun, k, int(v))
        for c_ in solver.all_ctxs():
            assert solver.L.ifem_set_tuning(c_, C.byref(tun)) == 0
    solver.channel_state(seed=args.seed, rel=args.rel)

    def step():
        solver.assemble(False)
        return solver.solve(False)

    for _ in range(args.warmup):
        step()

    def fence():
        # barrier + device idle.  The device work runs on the library's own HIP runtime and stream, so the library's
        # hipDeviceSynchronize is the device fence; torch holds no device state in this process (torch.cuda is never
        # initialised: it would bring up a second HIP runtime next to the library's).
        if dist:
            dist.barrier()
        solver.synchronize()

    # ---- the timed region: K steps, no per-kernel profiling (its event synchronisations would drain the queue)
    fence()
    t0 = time.time()
    t_asm = t_solve = asm_kernel_ms = 0.0
    last = None
    for _ in range(args.steps):
        ta = time.time()
        solver.assemble(False)
        tb = time.time()
        last = solver.solve(False)
        tc = time.time()
        t_asm += tb - ta
        t_solve += tc - tb
        asm_kernel_ms += solver.timing().assemble_kernel_ms  # HIP events around the cell kernel on the context stream
    fence()
    elapsed = time.time() - t0
    comm = solver.comm_stats(reset=True)  # exchanges / all-reduces of the K timed steps (all multigrid levels)
    # the TRUE residual of the last timed solve, recomputed with the assembled operator (collective; outside the timed region)
    true_res, rhs_norm = solver.true_residual()
    solver.comm_stats(reset=True)
    if args.dump_update:
        import numpy as np
        t = solver.partition_tables()
        nuo, npo = t["n_unodes_owned"], t["n_pnodes_owned"]
        upd = np.zeros(3 * nuo + npo)
        assert solver.L.ifem_vec_get(solver.ctx, capi.VEC_UPDATE, upd.ctypes.data_as(C.c_void_p)) == 0
        np.savez(f"{args.dump_update}.rank{rank}.npz", update=upd, l2g_u=t["l2g_u"][:nuo], l2g_p=t["l2g_p"][:npo],
                 n_unodes_global=t["n_unodes_global"], n_pnodes_global=t["n_pnodes_global"])
    if dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    asm_kernel_ms /= max(args.steps, 1)
    ms_per_step = elapsed / args.steps * 1e3
    hbm_used_gb = _hbm_used_gb()  # of this rank's device, after the timed steps (the steady state of the step)
    # ---- per-kernel pass (outside the timed region): more steps with an event pair around every launch wrapper on the context
    # stream (ifem_kprof: nothing waits until the end of the pass, so the queue runs ahead of the host as in the timed steps)
    prof_steps = 2
    solver.kprof_begin()
    tp0 = time.time()
    for _ in range(prof_steps):
        step()
    solver.synchronize()
    prof_wall_ms = (time.time() - tp0) / prof_steps * 1e3
    prof = solver.kprof_end()
    # device memory once the run has settled: the unconstrained copies of B / B^T / S_m (kept for a change of the constrained-dof set) are
    # given back at the second cached assembly of a never-changed set on every level (assemble.hip)
    hbm_steady_gb = _hbm_used_gb()
    if rank == 0:
        kernels = kernel_table(prof, prof_steps, n, world)
        kernel_ms_sum = sum(k["ms_per_step"] for k in kernels)
        cached = args.warmup >= 1  # the timed steps keep B, B^T, M_p, diag(M_u), S_m of the warm-up (same constrained-dof set)
        models = kernel_models(solver.L, solver.ctx, n_cells, n_u, n_p, cached_blocks=cached)
        by_fam = {k["family"]: k for k in kernels}
        # dominant kernel = the family with the largest time per step; the assembly's launch time is the live HIP-event
        # measurement of the TIMED steps (ifem_timing::assemble_kernel_ms), the others come from the profiled steps
        dom = kernels[0]["family"] if kernels else "assemble_cells"
        if dom == "assemble_cells" or asm_kernel_ms >= kernels[0]["ms_per_step"]:
            ms = asm_kernel_ms
            nb, nf = models["asm"]
            gbs, tfs = nb / (ms * 1e-3) / 1e9, nf / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": tfs, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfs / FP64_PEAK_TFLOPS,
                    "kernel": "k_ins_assemble3 (one wavefront per cell: sum-factorised front, FP64 MFMA contraction, row-image atomic scatter)",
                    "traffic": pmc_traffic("k_ins_assemble3", n, "f64_cached" if cached else "f64"),
                    "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the 128^3 launch on one GPU"
                                      + ("" if world == 1 else " (single-rank pass; the per-rank launch of this run is the same kernel on the same block size)"),
                    "launch_ms": ms, "launches_timed": args.steps, "algorithmic_bytes": nb, "algorithmic_flops": nf,
                    "hbm_frac": gbs / HBM_PEAK_GBS, "compute_frac": tfs / FP64_PEAK_TFLOPS, "compute_peak": "FP64 MFMA"}
            # what actually bounds this kernel: the memory-side rate of its f64 atomic requests.  One atomic per
            # (cell, dof pair) of the velocity-velocity block; the unit retires ~24 G requests of up to 64 bytes per second
            # whatever the lane count (tools/atomics_types.hip: 23.9 G/s f64, 188 G atomics/s only when 8 lanes share a
            # segment); the kernel's staged scatter touches 871 segments per cell (tools/scatter_sim.py, CPU replay)
            n_atom = float(n_cells) * (27 * 3) ** 2
            n_seg = float(n_cells) * 871.0
            roof.update({"atomic_adds": n_atom, "atomic_rate_gatom_s": n_atom / (ms * 1e-3) / 1e9, "atomic_peak_gatom_s": 188.0,
                         "atomic_frac": n_atom / (ms * 1e-3) / 1e9 / 188.0,
                         "atomic_segments": n_seg, "atomic_segment_rate_g_s": n_seg / (ms * 1e-3) / 1e9, "atomic_segment_peak_g_s": 24.0,
                         "atomic_segment_frac": n_seg / (ms * 1e-3) / 1e9 / 24.0})
        else:
            k = kernels[0]
            if "compute_frac" in k and k["compute_frac"] >= k["hbm_frac"]:
                roof = {"bound": "mfma", "achieved": k["tflop_s"], "peak": COMPUTE_PEAK[dom][1], "unit": "TFLOP/s", "frac": k["compute_frac"]}
            else:
                roof = {"bound": "hbm", "achieved": k["gb_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["hbm_frac"]}
            roof.update({"kernel": k["kernel"], "traffic": k.get("traffic"), "launch_ms": k["ms_per_step"] / max(k["launches_per_step"], 1),
                         "launches_timed": k["launches_per_step"] * prof_steps, "algorithmic_bytes": k["algorithmic_bytes"],
                         "algorithmic_flops": k["algorithmic_flops"]})
        mfk = by_fam.get("mf_cell", {})
        mf_calls, mf_ms = mfk.get("launches_per_step", 0), mfk.get("ms_per_step", 0.0)
        roof.update({"kernels": kernels, "kernel_ms_sum": kernel_ms_sum, "profiled_step_ms": prof_wall_ms,
                     "unattributed_ms": ms_per_step - kernel_ms_sum,
                     "kernels_note": "one entry per kernel family of the step, from HIP event pairs around every launch wrapper over "
                                     f"{prof_steps} profiled steps after the timed region (include/ifem_hip.h: ifem_kprof_begin / _end); "
                                     "unattributed_ms = ms_per_step - kernel_ms_sum: host launch gaps and the waits for reduction results",
                     "kernel_ms_per_step": {k["family"]: k["ms_per_step"] for k in kernels}})
        out = {
            "metric": "DoF/s per Newton step (assemble+solve), 3D INS Q2/Q1",
            "value": n_dofs_global / (elapsed / args.steps),
            "unit": "DoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D channel flow {'x'.join(str(r) for r in reps)} Q2/Q1, mpi_insim Newton step "
                                   f"(plane Poiseuille + seeded 1e-3 perturbation), {n}^3 cells per GPU",
                       "n_dofs": n_dofs_global, "cells_per_gpu": n_cells, "parallelism": f"dd{world}",
                       "assemble_ms": t_asm / args.steps * 1e3, "solve_ms": t_solve / args.steps * 1e3,
                       "assemble_kernel_ms": asm_kernel_ms, "setup_s": t_setup, "hbm_used_gb": hbm_used_gb, "hbm_used_gb_steady": hbm_steady_gb,
                       "true_rel_residual": true_res / rhs_norm if rhs_norm > 0 else None, "fgmres_rel_residual": last.fgmres_res / rhs_norm if rhs_norm > 0 else None,
                       "fgmres_rel_tol": solver.opts.fgmres_rel, "seed": args.seed, "perturbation": args.rel,
                       "rccl_nranks": comm["rccl_nranks"], "comm_transport": {0: "none (single rank)", 1: "rccl", 2: "local world"}[comm["transport"]],
                       "halo_neighbors": comm["n_neighbors"], "halo_stream": comm["halo_stream"],
                       "halo_exchanges_per_step": comm["halo_exchanges"] / args.steps, "allreduce_stream_per_step": comm["allreduce_dev"] / args.steps,
                       "allreduce_host_per_step": comm["allreduce_host"] / args.steps, "allreduce_vector_per_step": comm["allreduce_vec"] / args.steps,
                       "fgmres_iters": last.fgmres_iters, "cg_mp_iters": last.cg_mp_iters,
                       "cg_sm_iters": last.cg_sm_iters, "inner_iters": last.inner_iters, "inner_rel": solver.opts.inner_rel, "inner_rel_first": solver.opts.inner_rel_first, "ainv_kind": solver.opts.ainv_kind, "outer_matrix_free": solver.opts.outer_matrix_free,
                       "solver_opts": {"ainv_kind": solver.opts.ainv_kind, "inner_restart": solver.opts.inner_restart, "inner_rel": solver.opts.inner_rel,
                                       "inner_rel_first": solver.opts.inner_rel_first, "source": "InsIM::initialize_system defaults" if not overridden else "overridden: " + ",".join(overridden)},
                       "solver_opts_overridden": overridden,
                       "t_cg_mp_ms": last.t_cg_mp_ms, "t_cg_sm_ms": last.t_cg_sm_ms, "t_ainv_ms": last.t_ainv_ms,
                       "mf_apply_ms": mf_ms / max(mf_calls, 1), "mf_applies": mf_calls,
                       "sm_multigrid_levels": int(last.sm_mg_levels),
                       "coarse_levels": [list(r) for r, _ in solver.mg_levels()],
                       "hierarchy": "csrc/host/insim.cpp::attach_multigrid_levels (C++ host mirror)"},
            "roofline": roof,
        }
        out["config"].update({"cg_mp_rel": solver.opts.mp_rel, "cg_sm_rel": solver.opts.sm_rel})
        # Side measurement, never `value`: the same Newton step with the accuracy knobs of the PRECONDITIONER relaxed
        # (pressure CG solves to 1e-2 / 1e-1 instead of the reference's 1e-6 / 1e-3) and the u-u block of the outer operator
        # applied matrix-free.  The outer FGMRES still stops at the reference's 1e-4 ||rhs|| on the same operator.
        if world == 1 and args.tuned and args.sm_rel is None and args.mp_rel is None and not solver.opts.outer_matrix_free:
            keep = (solver.opts.mp_rel, solver.opts.sm_rel, solver.opts.outer_matrix_free)
            solver.opts.mp_rel, solver.opts.sm_rel, solver.opts.outer_matrix_free = 1e-2, 1e-1, 1
            step()
            fence()
            t0 = time.time()
            for _ in range(args.steps):
                st = step()
            solver.synchronize()
            dt_t = (time.time() - t0) / args.steps
            out["tuned_preconditioner"] = {"ms_per_step": dt_t * 1e3, "value": n_dofs_global / dt_t, "unit": "DoF/s",
                                           "fgmres_iters": st.fgmres_iters, "cg_mp_rel": 1e-2, "cg_sm_rel": 1e-1, "outer_matrix_free": 1,
                                           "note": "side measurement; `value` above uses the reference's tolerances"}
            solver.opts.mp_rel, solver.opts.sm_rel, solver.opts.outer_matrix_free = keep  # the legs below use the reference's again
        if world == 1 and args.extras:
            out.update(extras(solver, capi, n_dofs_global, ms_per_step))
            # the three figures side by side: `value` keeps B, B^T, M_p, diag(M_u), S_m of the previous iteration
            # (same constrained-dof set) and is the FIRST Newton iteration of a step; value_cold re-integrates / re-forms them
            # inside the step as the reference does in every iteration; value_sustained = DoF x Newton iterations / time of a
            # whole run_one_step loop from the same state (later iterations need more FGMRES iterations)
            if "cold_step" in out:
                out["value_cold"] = out["cold_step"]["value"]
            if "dofs_per_s_per_iteration" in out.get("time_step", {}):
                out["value_sustained"] = out["time_step"]["dofs_per_s_per_iteration"]
            if "value" in out.get("matrix_free_step", {}):
                out["value_matrix_free"] = out["matrix_free_step"]["value"]
        if world == 1 and args.fsi:
            # side measurement, last device leg: its Dirichlet-mode call edits the constraint sets of the context
            try:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                import fsibench
                solid = fsibench.make_solid3d()
                leg = fsibench.device_leg(solver.L, solver.ctx, capi, n_cells, solid)
                leg["cpu_baseline"] = fsibench.cpu_leg(args.fsi, solid)
                out["fsi_inputs"] = leg
            except Exception as e:  # a side leg must not take the headline line with it
                out["fsi_inputs"] = {"error": repr(e)}
        if world == 1 and args.cylinder_legs:
            # side measurements (configs 2 and 4 report DoF/s too; VERDICT r4 item 5: the cell kernel on an unstructured 3D
            # mesh of scale).  The channel's contexts are released first: the 3D cylinder needs ~40 GB of its own.
            try:
                solver.close()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import cylbench
                legs = {}
                for w, r in (("cylinder2d", 3), ("cylinder2d_scnsim", 3), ("cylinder3d", 3)):
                    o = cylbench.run(w, r, steps=2, device=local_rank)
                    legs[w] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "config": o["config"]}
                    if "roofline" in o and "assemble3_unstructured" in o["roofline"]:
                        legs[w]["assemble3_unstructured"] = o["roofline"]["assemble3_unstructured"]
                        legs[w]["kernel_ms_per_step"] = {k["family"]: k["ms_per_step"] for k in o["roofline"]["kernels"]}
                out["cylinder_workloads"] = legs
            except Exception as e:  # a side leg must not take the headline line with it
                out["cylinder_workloads"] = {"error": repr(e)}
        cpu_sizes = [int(v) for v in str(args.cpu_n).split(",") if int(v) > 0]
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ifem_cpu_baseline_n1.json")
        if cpu_sizes and world == 1:  # the CPU baseline is a rank-0, N = 1 measurement
            out["cpu_baseline"] = cpu_baseline(cpu_sizes)
            try:  # the N > 1 runs that follow on this host quote it instead of repeating it
                with open(cache, "w") as f:
                    json.dump(dict(out["cpu_baseline"], measured_unix_time=time.time()), f)
            except OSError:
                pass
        elif world > 1:
            # measured on rank 0 at N = 1 only: an N > 1 line names the N = 1 sample it re-uses -- the one the --gpus 1 run left on
            # this host, else the committed one of the round's single-GPU run (profiles/cpu_baseline_n1.json)
            for path, origin in ((cache, "the --gpus 1 run of this bench on this host"),
                                 (os.path.join(ROOT, "profiles", "cpu_baseline_n1.json"), "committed sample of the round's 1-GPU run (profiles/cpu_baseline_n1.json)")):
                try:
                    with open(path) as f:
                        cb = json.load(f)
                    cb["reused_from"] = f"N = 1 sample, not re-measured at N = {world}: {origin}"
                    out["cpu_baseline"] = cb
                    break
                except (OSError, ValueError):
                    continue
        emit(out)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
