"""Multigrid-preconditioned CG(S_m) inside the block Schur preconditioner (ifem_mg_attach, solver.hip::pcg_mg_sm).

The reference solves S_m with unpreconditioned CG to 1e-3 ||v|| (mpi_insim.cpp:86-112).  The multigrid variant keeps that
stopping rule on the true residual, so everything downstream (FGMRES to 1e-4 ||rhs|| on the assembled operator) must be
unaffected, while the CG count drops and stops growing with the mesh.  Checked here: the S_m^-1 of the two variants
agree to the CG tolerance, the solve still meets the reference's stopping rule against the ORACLE's matrix, iteration
counts, the same on virtual ranks (transfers through halo refresh / reverse scatter-add), and InsIMEX, whose tight
tolerance makes plain CG(S_m) impractical on fine meshes."""
import ctypes as C
import threading

import numpy as np
import pytest

import orc
from boxmesh import BoxMesh
from cases import channel3d_state

pytestmark = pytest.mark.gpu

EXTENT = (2.0, 0.2, 0.2)


def _hierarchy(n, P=(1, 1, 1), rank=0, worlds=None, kind="InsIM", replica_cells=None):
    """the channel through the C++ host mirror: FluidSolver::attach_multigrid_levels (csrc/host/insim.cpp) builds the level
    chain, the transfers and the ifem_mg_attach calls inside initialize_system; on virtual ranks every level gets the
    world handed in for it"""
    from openifem_amd import host
    reps = tuple(n[d] * P[d] for d in range(3))
    s = getattr(host, kind)(host.channel_prm(3), reps, (0, 0, 0), EXTENT)
    if worlds is not None:
        s.set_partition(P, rank, local_world=worlds[0])
        s.set_multigrid(True, 0, worlds[1:])
    if replica_cells is not None:
        s.set_mg_replica_cells(replica_cells)
    s.setup(0)
    return s


def _get(s, vec, n):
    x = np.zeros(n)
    assert s.L.ifem_vec_get(s.ctx, vec, x.ctypes.data_as(C.c_void_p)) == 0
    return x


def test_host_mirror_attaches_the_level_chain_and_selects_the_multigrid_inner_solver():
    # InsIM<3>::initialize_system on a box: the chain of multigrid.hpp hangs below the context and the inner solver of
    # A~^-1 defaults to the V-cycle-preconditioned one (what a C++ caller of InsIM<3>::run() gets)
    from openifem_amd import capi, host
    s = _hierarchy((16, 16, 16))
    want = host.coarse_level_chain((16, 16, 16), (1, 1, 1), EXTENT)
    assert [r for r, _ in s.mg_levels()] == want and len(want) >= 2
    assert s.L.ifem_mg_depth(s.ctx) == len(want)
    assert s.opts.ainv_kind == capi.AINV_MG and s.opts.inner_restart == 16
    s.close()
    s = getattr(host, "InsIM")(host.channel_prm(3), (16, 16, 16), (0, 0, 0), EXTENT)
    s.set_multigrid(False)
    s.setup(0)
    assert s.mg_levels() == [] and s.opts.ainv_kind == 0
    s.close()


@pytest.mark.parametrize("n", [(16, 16, 16), (24, 16, 16)])
def test_multigrid_cg_sm_keeps_the_reference_stopping_rule(n):
    from openifem_amd import capi
    s = _hierarchy(n)
    depth = s.L.ifem_mg_depth(s.ctx)
    assert depth >= 2
    s.channel_state()
    s.opts.ainv_kind = 3
    s.opts.inner_restart = 16
    s.assemble(False)
    _, n_u, n_p = s.sizes()
    nt = n_u + n_p
    out = {}
    for mg in (0, 1):
        s.opts.sm_mg = mg
        st = s.solve(False)
        out[mg] = (_get(s, capi.VEC_UPDATE, nt), st.fgmres_iters, st.cg_sm_iters, st.sm_mg_levels, st.fgmres_res)
    assert out[0][3] == 0 and out[1][3] == depth + 1
    assert out[1][1] <= out[0][1] + 1, "multigrid CG(S_m) must not cost outer iterations"
    assert out[1][2] * 4 <= out[0][2], (out[0][2], out[1][2])
    # both updates solve the same system to the reference's 1e-4 ||rhs||: they differ by that much at most
    b = _get(s, capi.VEC_RHS, nt)
    x0, x1 = out[0][0], out[1][0]
    for x in (x0, x1):
        xt = np.ascontiguousarray(x)
        assert s.L.ifem_vec_set(s.ctx, capi.VEC_TMP, xt.ctypes.data_as(C.c_void_p)) == 0
        assert s.L.ifem_system_vmult(s.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0
        r = b - _get(s, capi.VEC_UPDATE, nt)
        cd, _ = s.constraints()
        r[cd] = 0
        assert np.linalg.norm(r) <= 1.05e-4 * np.linalg.norm(b)
    s.close()


def test_multigrid_solve_against_the_oracle_matrix():
    # small enough for the oracle: residual of the multigrid-preconditioned solve with the ORACLE's assembled matrix
    from openifem_amd import capi
    n = (8, 8, 8)
    s = _hierarchy(n)
    assert s.L.ifem_mg_depth(s.ctx) >= 1
    s.channel_state()
    s.opts.ainv_kind = 3
    s.assemble(False)
    st = s.solve(False)
    assert st.sm_mg_levels >= 2
    _, n_u, n_p = s.sizes()
    upd = _get(s, capi.VEC_UPDATE, n_u + n_p)
    t = s.partition_tables()
    g = np.concatenate([(t["l2g_u"][:, None] * 3 + np.arange(3)[None, :]).ravel(), 3 * t["n_unodes_global"] + t["l2g_p"]])
    m = BoxMesh(n, (0, 0, 0), EXTENT, kv=2)
    dofs, vals, _, _, kw = channel3d_state(m)
    # the state the host mirror seeded (its own generator), in the oracle mesh's lattice numbering
    present, ev = np.zeros(m.n_dofs), np.zeros(m.n_dofs)
    present[g], ev[g] = _get(s, capi.VEC_PRESENT, n_u + n_p), _get(s, capi.VEC_EVAL, n_u + n_p)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.assemble(orc.make_params(**kw), False, ev, present)
    A, b = S.csr("A"), S.rhs()
    x = np.zeros(m.n_dofs)
    x[g] = upd  # Morton numbering of the host mirror -> lattice numbering of the oracle's mesh
    assert np.linalg.norm(A @ x - b) <= 1.05e-4 * np.linalg.norm(b)
    s.close()


def test_multigrid_iteration_counts_do_not_grow_with_the_mesh():
    counts = {}
    for n in ((16, 16, 16), (32, 32, 32)):
        s = _hierarchy(n)
        s.channel_state()
        s.opts.ainv_kind = 3
        s.opts.inner_restart = 16
        s.assemble(False)
        st = s.solve(False)
        counts[n[0]] = (st.cg_sm_iters, st.precond_applies)
        s.close()
    per16, per32 = counts[16][0] / counts[16][1], counts[32][0] / counts[32][1]
    assert per32 <= 1.3 * per16 + 1, counts
    assert per32 <= 12, counts


def test_multigrid_on_virtual_ranks_matches_single_context():
    from openifem_amd import capi
    L = capi.load()
    n, P = (8, 8, 8), (2, 1, 1)
    world = 2
    # single context on the same global mesh
    s1 = _hierarchy((16, 8, 8))
    s1.channel_state()
    s1.opts.ainv_kind = 3
    s1.opts.fgmres_rel = 1e-9
    s1.opts.inner_rel = 1e-4
    s1.assemble(False)
    st1 = s1.solve(False)
    t1 = s1.partition_tables()
    _, n_u, n_p = s1.sizes()
    u1 = _get(s1, capi.VEC_UPDATE, n_u + n_p)
    g1 = np.concatenate([(t1["l2g_u"][:, None] * 3 + np.arange(3)[None, :]).ravel(), 3 * t1["n_unodes_global"] + t1["l2g_p"]])
    x1 = np.zeros(n_u + n_p)
    x1[g1] = u1
    depth = s1.L.ifem_mg_depth(s1.ctx)
    s1.close()
    worlds = [C.c_void_p(L.ifem_local_world_create(world)) for _ in range(depth + 1)]
    out, errs = [None] * world, []

    def work(rank):
        try:
            s = _hierarchy(n, P, rank, worlds)
            s.channel_state()
            assert L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL) == 0
            s.opts.ainv_kind = 3
            s.opts.fgmres_rel = 1e-9
            s.opts.inner_rel = 1e-4
            s.assemble(False)
            st = s.solve(False)
            t = s.partition_tables()
            no = 3 * t["n_unodes_owned"] + t["n_pnodes_owned"]
            out[rank] = (t, _get(s, capi.VEC_UPDATE, no), st.cg_sm_iters, st.sm_mg_levels, st.fgmres_iters)
            s.close()
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs
    xN = np.full(len(x1), np.nan)
    for t, u, _, _, _ in out:
        nuo, npo = t["n_unodes_owned"], t["n_pnodes_owned"]
        xN[(t["l2g_u"][:nuo, None] * 3 + np.arange(3)[None, :]).ravel()] = u[:3 * nuo]
        xN[3 * t["n_unodes_global"] + t["l2g_p"][:npo]] = u[3 * nuo:]
    assert not np.isnan(xN).any()
    assert out[0][3] == depth + 1 and out[1][3] == depth + 1, "multigrid was not used on the partitioned contexts"
    assert out[0][2] == out[1][2] and out[0][4] == out[1][4]
    assert np.linalg.norm(xN - x1) <= 1e-6 * np.linalg.norm(x1)
    assert abs(out[0][2] - st1.cg_sm_iters) <= max(3, st1.cg_sm_iters // 4), (out[0][2], st1.cg_sm_iters)
    for w in worlds:
        L.ifem_local_world_destroy(w)


def test_insimex_step_with_multigrid():
    # InsIMEX solves to min(1e-9, 1e-8 ||rhs||) (mpi_insimex.cpp:369-370): the later Krylov vectors are rough and plain
    # CG(S_m) needs hundreds of iterations per application; with the V-cycle the count stays in the tens
    from openifem_amd import capi
    res = {}
    for mg in (0, 1):
        s = _hierarchy((16, 16, 16), kind="InsIMEX")
        s.opts.ainv_kind = 3
        s.opts.inner_rel = 1e-2
        s.opts.sm_mg = mg
        s.channel_state()
        assert s.L.ifem_vec_copy(s.ctx, capi.VEC_PRESENT, capi.VEC_EVAL) == 0
        s.run_one_step(True, True)
        s.run_one_step(False, True)
        s.run_one_step(False, False)
        v, p = s.get_current_solution()
        res[mg] = (v.copy(), p.copy())
        s.close()
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-6 * np.abs(res[0][0]).max()
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-6 * np.abs(res[0][1]).max()


@pytest.mark.parametrize("dim,kv,reps", [(3, 2, (3, 2, 2)), (2, 2, (5, 3)), (2, 1, (6, 4)), (3, 1, (3, 3, 2))])
@pytest.mark.parametrize("use_nonzero", [False, True])
def test_matrix_free_block_diagonal_equals_the_assembled_one(dim, kv, reps, use_nonzero):
    """mg.hip::k_uu_diag (the block-Jacobi data of the coarse multigrid levels, integrated without a matrix) against the
    diagonal node blocks of the assembled A_uu on distorted cells with both constraint sets"""
    from openifem_amd import capi
    rng = np.random.default_rng(11 + dim + kv)
    m = BoxMesh(reps, (0,) * dim, (1.0, 0.6, 0.4)[:dim], kv=kv)
    m.vcoords = m.vcoords.copy()
    m.vcoords += 0.02 * rng.standard_normal(m.vcoords.shape)
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim), 3: (1, [0.05])})
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.vec_set(capi.VEC_PRESENT, rng.standard_normal(m.n_dofs))
    ctx.vec_set(capi.VEC_EVAL, rng.standard_normal(m.n_dofs))
    ctx.assemble(capi.make_params(mu=0.7, rho=1.3, gamma=0.2, dt=0.01, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5}), use_nonzero)
    # round 5: the integrals run in the arithmetic of the V-cycle they serve (ifem_tuning::mf_f32, default single precision: the smoother
    # applies the inverse blocks in single precision anyway); with mf_f32 = 0 they are double precision
    a, b = ctx.uu_block_diag(0), ctx.uu_block_diag(1)
    assert np.abs(a - b).max() <= 5e-6 * np.abs(a).max()
    ctx.set_tuning(mf_f32=0)
    b64 = ctx.uu_block_diag(1)
    assert np.abs(a - b64).max() <= 1e-10 * np.abs(a).max()
    assert np.abs(ctx.uu_block_diag(0) - a).max() == 0.0  # the hook restored the assembled blocks
    ctx.close()


@pytest.mark.parametrize("n", [(16, 16, 16), (24, 16, 16)])
def test_multigrid_ainv_keeps_the_reference_stopping_rule(n):
    """IFEM_AINV_MG: the V-cycle only preconditions the inner solve that stands in for MUMPS; the outer FGMRES on the
    assembled operator still stops at 1e-4 ||rhs|| and needs no more iterations than with the Jacobi-preconditioned inner solve"""
    from openifem_amd import capi
    s = _hierarchy(n)
    s.channel_state()
    s.opts.inner_restart = 16
    s.assemble(False)
    _, n_u, n_p = s.sizes()
    nt = n_u + n_p
    b = _get(s, capi.VEC_RHS, nt)
    cd, _ = s.constraints()
    out = {}
    for kind in (3, 4):
        s.opts.ainv_kind = kind
        st = s.solve(False)
        x = _get(s, capi.VEC_UPDATE, nt)
        assert s.L.ifem_vec_set(s.ctx, capi.VEC_TMP, x.ctypes.data_as(C.c_void_p)) == 0
        assert s.L.ifem_system_vmult(s.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0
        r = b - _get(s, capi.VEC_UPDATE, nt)
        r[cd] = 0
        assert np.linalg.norm(r) <= 1.05e-4 * np.linalg.norm(b), kind
        out[kind] = (st.fgmres_iters, st.inner_iters)
    assert out[4][0] <= out[3][0] + 1
    assert out[4][1] * 2 <= out[3][1], out
    # an asymmetric cycle (fewer smoothing steps after the coarse correction) is admissible under the flexible inner GMRES
    s.opts.ainv_kind = 4
    s.opts.mg_smooth_u, s.opts.mg_smooth_u_post = 3, 1
    st = s.solve(False)
    x = _get(s, capi.VEC_UPDATE, nt)
    assert s.L.ifem_vec_set(s.ctx, capi.VEC_TMP, x.ctypes.data_as(C.c_void_p)) == 0
    assert s.L.ifem_system_vmult(s.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0
    r = b - _get(s, capi.VEC_UPDATE, nt)
    r[cd] = 0
    assert np.linalg.norm(r) <= 1.05e-4 * np.linalg.norm(b)
    s.close()


def test_tight_first_inner_solve_saves_outer_iterations_only_on_velocity_dominated_residuals():
    """ifem_solver_opts::inner_rel_first (the bench setting): the first preconditioner application of a solve resolves A~^-1
    further when the residual is velocity-dominated; the true residual still meets 1e-4 ||rhs||, with no more outer
    iterations than without it.  On a residual that is mostly continuity equation the option must change nothing."""
    from openifem_amd import capi
    s = _hierarchy((16, 16, 16))
    s.channel_state()
    s.opts.ainv_kind, s.opts.inner_restart = 4, 16
    s.assemble(False)
    _, n_u, n_p = s.sizes()
    nt = n_u + n_p
    cd, _ = s.constraints()

    def solve(first):
        s.opts.inner_rel_first = first
        st = s.solve(False)
        x = _get(s, capi.VEC_UPDATE, nt)
        assert s.L.ifem_vec_set(s.ctx, capi.VEC_TMP, x.ctypes.data_as(C.c_void_p)) == 0
        assert s.L.ifem_system_vmult(s.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0
        r = _get(s, capi.VEC_RHS, nt) - _get(s, capi.VEC_UPDATE, nt)
        r[cd] = 0
        return st.fgmres_iters, st.inner_iters, np.linalg.norm(r) / np.linalg.norm(_get(s, capi.VEC_RHS, nt)), x

    b = _get(s, capi.VEC_RHS, nt)
    share = np.linalg.norm(b[n_u:]) / np.linalg.norm(b)
    assert share < 1e-3, share  # the bench state: momentum residual of the perturbed Poiseuille flow
    it0, in0, res0, _ = solve(0.0)
    it1, in1, res1, _ = solve(5e-5)
    assert res0 <= 1.05e-4 and res1 <= 1.05e-4
    assert it1 <= it0 and in1 > 0, (it0, in0, it1, in1)
    # a pressure-dominated right-hand side: same iterations, same update, with or without the option
    rhs = np.zeros(nt)
    rhs[n_u:] = np.random.default_rng(3).standard_normal(n_p)
    rhs[n_u:] -= rhs[n_u:].mean()
    assert s.L.ifem_vec_set(s.ctx, capi.VEC_RHS, rhs.ctypes.data_as(C.c_void_p)) == 0
    ita, ina, resa, xa = solve(0.0)
    itb, inb, resb, xb = solve(5e-5)
    assert (ita, ina) == (itb, inb) and np.abs(xa - xb).max() <= 1e-12 * np.abs(xa).max()
    s.close()


def test_tight_first_inner_solve_backs_off_after_a_miss():
    """inner_rel_first is self-correcting: a solve that used it and still needed a second outer iteration has bought nothing
    with the extra inner iterations, so the context leaves it off for its next 8 qualifying solves (16, 32, 64 after repeated
    misses) and then tries again; a solve that ends at its first check keeps it on.  The miss is forced here with an outer
    tolerance no single iteration reaches."""
    s = _hierarchy((16, 16, 16))
    s.channel_state()
    s.opts.ainv_kind, s.opts.inner_restart = 4, 16
    s.opts.inner_rel_first = 5e-5
    s.opts.inner_first_pshare = 1e-2  # (the default, 10 fgmres_rel, moves with the outer tolerance this test plays with)
    s.assemble(False)
    s.opts.fgmres_rel = 1e-9
    used = []
    for _ in range(11):
        st = s.solve(False)
        assert st.fgmres_iters > 1
        used.append(st.inner_first_tight)
    assert used == [1] + [0] * 8 + [1, 0], used  # miss -> 8 solves without it -> retry -> second miss: 16 solves off
    # a solve that does end at its first check (outer tolerance 0.5) switches it back on for good
    s.opts.fgmres_rel = 0.5
    seen = []
    for _ in range(20):
        st = s.solve(False)
        assert st.fgmres_iters == 1
        seen.append(st.inner_first_tight)
    assert seen[:15] == [0] * 15 and seen[15:] == [1] * 5, seen
    s.close()


@pytest.mark.parametrize("P,replica_cells", [((2, 1, 1), 0), ((2, 2, 1), 0), ((2, 2, 2), 0), ((2, 1, 1), 32768), ((2, 2, 2), 32768),
                                             ((2, 2, 2), 600)])
def test_multigrid_ainv_on_virtual_ranks(P, replica_cells):
    """the bench configuration (IFEM_AINV_MG + multigrid CG(S_m), halo overlap on) on the partitions bench.py uses for 2, 4 and
    8 GPUs: face, edge and corner neighbours on every level, transfers across rank boundaries.  replica_cells 0: every level
    partitioned; 32768: every coarse level of these small meshes is a replicated single-rank context (ifem_mg_attach's replicated
    coarse level: restrictions summed by a vector all-reduce, no exchange below the finest level); 600: one partitioned coarse level,
    replicas below it"""
    from openifem_amd import capi
    L = capi.load()
    n, world = (8, 8, 8), int(np.prod(P))
    s1 = _hierarchy(tuple(n[d] * P[d] for d in range(3)))
    s1.channel_state()
    s1.opts.ainv_kind = 4
    s1.opts.fgmres_rel = 1e-9
    s1.opts.inner_rel = 1e-4
    s1.assemble(False)
    st1 = s1.solve(False)
    t1 = s1.partition_tables()
    _, n_u, n_p = s1.sizes()
    g1 = np.concatenate([(t1["l2g_u"][:, None] * 3 + np.arange(3)[None, :]).ravel(), 3 * t1["n_unodes_global"] + t1["l2g_p"]])
    x1 = np.zeros(n_u + n_p)
    x1[g1] = _get(s1, capi.VEC_UPDATE, n_u + n_p)
    depth = s1.L.ifem_mg_depth(s1.ctx)
    s1.close()
    worlds = [C.c_void_p(L.ifem_local_world_create(world)) for _ in range(depth + 1)]
    out, errs, levels = [None] * world, [], [None] * world

    def work(rank):
        try:
            s = _hierarchy(n, P, rank, worlds, replica_cells=replica_cells)
            s.channel_state()
            assert L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL) == 0
            s.opts.ainv_kind = 4
            s.opts.fgmres_rel = 1e-9
            s.opts.inner_rel = 1e-4
            s.assemble(False)
            capi.comm_stats(L, s.ctx, reset=True)
            st = s.solve(False)
            t = s.partition_tables()
            no = 3 * t["n_unodes_owned"] + t["n_pnodes_owned"]
            out[rank] = (t, _get(s, capi.VEC_UPDATE, no), st.inner_iters, st.fgmres_iters)
            levels[rank] = capi.comm_stats_levels(L, s.ctx)
            s.close()
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs
    xN = np.full(len(x1), np.nan)
    for t, u, _, _ in out:
        nuo, npo = t["n_unodes_owned"], t["n_pnodes_owned"]
        xN[(t["l2g_u"][:nuo, None] * 3 + np.arange(3)[None, :]).ravel()] = u[:3 * nuo]
        xN[3 * t["n_unodes_global"] + t["l2g_p"][:npo]] = u[3 * nuo:]
    assert np.linalg.norm(xN - x1) <= 1e-6 * np.linalg.norm(x1)
    assert len({o[2] for o in out}) == 1 and len({o[3] for o in out}) == 1, "ranks disagree on the iteration counts"
    assert abs(out[0][2] - st1.inner_iters) <= max(3, st1.inner_iters // 4), (out[0][2], st1.inner_iters)
    # per-level communication of the solve (ifem_comm_stats_level): replicated levels are single-rank contexts and never communicate;
    # the level above the first replica hands over by vector all-reduces
    lv = levels[0]
    n_part = sum(1 for k in lv if k["nranks"] > 1)
    assert n_part == {0: len(lv), 32768: 1, 600: 2}[replica_cells], lv
    for k in lv[n_part:]:
        assert k["nranks"] == 1 and k["halo_exchanges"] == k["allreduce_dev"] == k["allreduce_host"] == k["allreduce_vec"] == 0, lv
    assert (lv[n_part - 1]["allreduce_vec"] > 0) == (replica_cells > 0), lv
    assert all(k["allreduce_vec"] == 0 for k in lv[:n_part - 1]), lv
    for w in worlds:
        L.ifem_local_world_destroy(w)


@pytest.mark.parametrize("reps", [(9, 7, 5), (4, 4, 4), (3, 1, 2)])
def test_single_precision_operator_matches_the_fp64_operator(reps):
    """the inner solve's single-precision matrix-free A_uu (ifem_uu_vmult variant IFEM_AINV_MG) against the fp64 matrix-free operator
    (itself 1e-12 against the assembled matrix, test_gpu_parity.py) on distorted meshes whose cell counts leave every kind of tail
    (315 cells, 64, 6), with constraints and a convective state"""
    from openifem_amd import capi
    from boxmesh import BoxMesh
    rng = np.random.default_rng(11)
    m = BoxMesh(reps, (0, 0, 0), (1.8, 0.7, 0.5), kv=2)
    m.vcoords = m.vcoords.copy()
    m.vcoords += 0.01 * rng.standard_normal(m.vcoords.shape)
    dofs, vals = m.dirichlet({0: (7, [0.3, -0.2, 0.1]), 2: (7, [0.0, 0.0, 0.0])})
    ctx = capi.Context(3, 2, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.vec_set(capi.VEC_PRESENT, rng.standard_normal(m.n_dofs))
    ctx.vec_set(capi.VEC_EVAL, rng.standard_normal(m.n_dofs))
    ctx.assemble(capi.make_params(mu=0.7, rho=1.3, gamma=0.2, dt=0.01), False)
    n_u = 3 * m.n_unodes
    x = rng.standard_normal(m.n_dofs)
    y64, y32 = ctx.uu_vmult(x, 3)[:n_u], ctx.uu_vmult(x, 4)[:n_u]
    assert np.abs(y32 - y64).max() <= 3e-5 * np.abs(y64).max()
    ctx.close()


def _cylinder_run(refinements, multigrid, steps=1, partition=None):
    """partition: (world, rank, local world handle) -- the strip partition of partition_unstructured on virtual ranks"""
    import os
    import re
    from openifem_amd import host
    prm = open(os.path.join(os.path.dirname(__file__), "golden", "prm", "fluid_cylinder_mpi.prm")).read()
    prm = re.sub(r"set Global refinements\s*=\s*\d+", f"set Global refinements = {refinements}", prm)
    flow = host.InsIM(prm, mesh="cylinder")
    flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0)
    if partition is not None:
        flow.set_partition((partition[0], 1, 1), partition[1], local_world=partition[2])
    flow.set_multigrid(multigrid)
    if not multigrid:
        flow.opts.inner_rel = 1e-3
        flow.opts.inner_maxit = 4000
    flow.setup(refinements)
    n_levels = len(flow.mg_levels())
    flow.run_one_step(True)
    st = flow.last_stats()
    v, p = flow.get_current_solution()
    if partition is not None:
        t = flow.partition_tables()
        v, p = v[:2 * t["n_unodes_owned"]], p[:t["n_pnodes_owned"]]
    out = dict(vmax=v.max(), pmax=p.max(), levels=n_levels, fgmres=st.fgmres_iters, inner=st.inner_iters / max(st.precond_applies, 1),
               cg_sm=st.cg_sm_iters / max(st.precond_applies, 1), n_dofs=len(v) + len(p), ainv=flow.opts.ainv_kind)
    flow.close()
    return out


def test_cylinder_level_chain_from_the_refinement_history():
    """round 4 (missing #3 / item 6): InsIM on the cylinder mesh (tests/fluid_cylinder_mpi) attaches the levels the mesh
    passed through under refine_global (host/insim.cpp::attach_nested_levels): the headline inner solver (matrix-free A_uu + V-cycle,
    multigrid-preconditioned CG(S_m)) on an unstructured mesh.  Three refinements = the reference's test: the regression constants
    0.374235 / 46.5226 are those of the Jacobi-preconditioned run; four and five refinements (0.87 M DoF): the inner iteration
    counts do not grow with the mesh."""
    from openifem_amd import capi
    ref = _cylinder_run(3, False)
    got = {r: _cylinder_run(r, True) for r in (3, 4, 5)}
    assert got[3]["levels"] == 3 and got[5]["levels"] == 5 and got[3]["ainv"] == capi.AINV_MG
    assert abs(got[3]["vmax"] - ref["vmax"]) < 1e-5 * ref["vmax"] and abs(got[3]["pmax"] - ref["pmax"]) < 1e-4 * ref["pmax"]
    assert abs(got[3]["vmax"] - 0.374235) / 0.374235 < 1e-3 and abs(got[3]["pmax"] - 46.5226) / 46.5226 < 1e-3
    assert got[5]["n_dofs"] > 800000
    # (convection-dominated operator, curved ring: the V-cycle is weaker than on the channel -- 13 / 18 / 20 inner iterations per
    # application -- but the counts level off, where Jacobi alone doubles per refinement)
    assert got[4]["inner"] <= 1.5 * got[3]["inner"] and got[5]["inner"] <= 1.2 * got[4]["inner"] + 1, got
    for r in (4, 5):
        assert got[r]["cg_sm"] <= got[3]["cg_sm"] + 2, got
        assert got[r]["fgmres"] <= got[3]["fgmres"] + 4, got
    assert got[3]["inner"] < 0.25 * ref["inner"], (got, ref)


@pytest.mark.parametrize("world", [2, 4])
def test_cylinder_level_chain_below_a_partitioned_mesh(world):
    """... and on several ranks: the strips of partition_unstructured cannot be cut consistently per level, so the refinement history hangs
    below the partitioned mesh as REPLICATED single-rank levels (attach_nested_levels, ifem_mg_attach's replicated coarse level).  Five
    refinements (0.87 M DoF) on 2 / 4 virtual ranks: the same constants and -- within the block-Jacobi effects of the partition -- the
    same counts as the single context"""
    from openifem_amd import capi
    L = capi.load()
    one = _cylinder_run(5, True)
    w = C.c_void_p(L.ifem_local_world_create(world))
    out, errs = [None] * world, []

    def work(rank):
        try:
            out[rank] = _cylinder_run(5, True, partition=(world, rank, w))
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not errs, errs
    L.ifem_local_world_destroy(w)
    assert all(o["levels"] == 5 and o["ainv"] == capi.AINV_MG for o in out), out
    assert sum(o["n_dofs"] for o in out) == one["n_dofs"]
    vmax, pmax = max(o["vmax"] for o in out), max(o["pmax"] for o in out)
    assert abs(vmax - one["vmax"]) < 1e-4 * one["vmax"] and abs(pmax - one["pmax"]) < 1e-3 * one["pmax"], (vmax, pmax, one)
    assert len({o["fgmres"] for o in out}) == 1 and abs(out[0]["fgmres"] - one["fgmres"]) <= 3, (out, one)
    assert out[0]["inner"] <= 1.3 * one["inner"] + 2, (out, one)
    # CG(S_m): the strips have no 2-deep pressure halo, so the finest level applies S_m as two SpMVs -- the V-cycle smooths with that
    # operator and the diagonal the rows of B give (linalg.hip::sm_diag_from_blocks); plain CG needed hundreds of iterations here
    assert out[0]["cg_sm"] <= one["cg_sm"] + 3, (out, one)


def _cylinder3d_run(refinements, multigrid, partition=None):
    """tests/fluid_cylinder_mpi's dim == 3 branch (fluid_cylinder_mpi.cpp:98-104) on the host mirror's extruded cylinder mesh"""
    import os
    import re
    from openifem_amd import host
    prm = open(os.path.join(os.path.dirname(__file__), "golden", "prm", "fluid_cylinder_mpi.prm")).read()
    prm = re.sub(r"set Dimension = 2", "set Dimension = 3", prm)
    prm = re.sub(r"set Global refinements = 3, 0", f"set Global refinements = {refinements}, 0", prm)
    prm = re.sub(r"set Gravity = 0.0, 0.0", "set Gravity = 0.0, 0.0, 0.0", prm)
    prm = re.sub(r"set Initial velocity = 0.0, 0.0", "set Initial velocity = 0.0, 0.0, 0.0", prm)
    prm = re.sub(r"set Number of Dirichlet BCs = 4", "set Number of Dirichlet BCs = 6", prm)
    prm = re.sub(r"set Dirichlet boundary id = 0, 2, 3, 4", "set Dirichlet boundary id = 0, 2, 3, 4, 5, 6", prm)
    prm = re.sub(r"set Dirichlet boundary components = 3, 3, 3, 3", "set Dirichlet boundary components = 7, 7, 7, 7, 7, 7", prm)
    prm = re.sub(r"set Dirichlet boundary values = 0.2, 0, 0, 0, 0, 0, 0, 0", "set Dirichlet boundary values = " + ", ".join(["0"] * 18), prm)

    from cylmesh import inflow_bc_3d  # parabolic in y and z at the inlet x = -0.3 (fluid_cylinder_mpi.cpp:56-75)
    flow = host.InsIM(prm, mesh="cylinder")
    flow.add_hard_coded_boundary_condition(0, lambda p, c, t: inflow_bc_3d(p, c))
    if partition is not None:
        flow.set_partition((partition[0], 1, 1), partition[1], local_world=partition[2])
    flow.set_multigrid(multigrid)
    if not multigrid:
        flow.opts.inner_rel = 1e-3
        flow.opts.inner_maxit = 4000
    flow.setup(refinements)
    n_levels = len(flow.mg_levels())
    flow.run_one_step(True)
    st = flow.last_stats()
    v, p = flow.get_current_solution()
    if partition is not None:
        t = flow.partition_tables()
        v, p = v[:3 * t["n_unodes_owned"]], p[:t["n_pnodes_owned"]]
    out = dict(vmax=v.max(), pmax=p.max(), vsum=np.abs(v).sum(), levels=n_levels, fgmres=st.fgmres_iters,
               inner=st.inner_iters / max(st.precond_applies, 1), cg_sm=st.cg_sm_iters / max(st.precond_applies, 1),
               n_dofs=len(v) + len(p), ainv=flow.opts.ainv_kind)
    flow.close()
    return out


def test_extruded_cylinder_level_chain():
    """the 3D (extruded) cylinder mesh of Utils::GridCreator<3>::flow_around_cylinder refined once: its unrefined mesh hangs below it as a
    level (parent-child tables of the hexahedral generator, multigrid.cpp::nested_prolongation in 3D), the matrix-core cell kernel and
    the matrix-free operator run on an unstructured 3D mesh, and the multigrid-preconditioned inner solves reproduce the
    Jacobi-preconditioned run"""
    from openifem_amd import capi
    ref = _cylinder3d_run(1, False)
    got = _cylinder3d_run(1, True)
    assert ref["levels"] == 0 and got["levels"] == 1 and got["ainv"] == capi.AINV_MG
    assert got["n_dofs"] == ref["n_dofs"]
    assert abs(got["vmax"] - ref["vmax"]) <= 1e-4 * abs(ref["vmax"]), (got, ref)
    assert abs(got["pmax"] - ref["pmax"]) <= 1e-3 * abs(ref["pmax"]), (got, ref)
    assert abs(got["vsum"] - ref["vsum"]) <= 1e-4 * ref["vsum"], (got, ref)
    assert got["inner"] < 0.5 * ref["inner"], (got, ref)
    assert got["cg_sm"] < 0.5 * ref["cg_sm"], (got, ref)


def test_extruded_cylinder_level_chain_on_virtual_ranks():
    """... and cut into two strips: the unrefined 3D mesh as a replicated level below the partitioned one"""
    from openifem_amd import capi
    L = capi.load()
    one = _cylinder3d_run(1, True)
    world = 2
    w = C.c_void_p(L.ifem_local_world_create(world))
    out, errs = [None] * world, []

    def work(rank):
        try:
            out[rank] = _cylinder3d_run(1, True, partition=(world, rank, w))
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs
    L.ifem_local_world_destroy(w)
    assert all(o["levels"] == 1 and o["ainv"] == capi.AINV_MG for o in out), out
    assert sum(o["n_dofs"] for o in out) == one["n_dofs"]
    assert abs(max(o["vmax"] for o in out) - one["vmax"]) <= 1e-4 * abs(one["vmax"]), (out, one)
    assert abs(max(o["pmax"] for o in out) - one["pmax"]) <= 1e-3 * abs(one["pmax"]), (out, one)
    assert abs(sum(o["vsum"] for o in out) - one["vsum"]) <= 1e-4 * one["vsum"], (out, one)
    assert out[0]["inner"] <= 1.5 * one["inner"] + 2 and out[0]["cg_sm"] <= one["cg_sm"] + 3, (out, one)


def test_captured_vcycle_graph_reproduces_the_eager_cycle():
    """ifem_tuning::vcycle_graph_cells: on a small single-rank chain the A_uu V-cycle is captured into a hipGraph once per state and
    replayed.  Same kernels in the same order on the same buffers: two time steps of tests/fluid_cylinder_mpi must come out as with eager
    launches (to rounding: the coarse levels' block diagonals are summed with atomics, so two eager runs differ in the last bits too), with
    the same iteration counts, and the graph must actually have been used."""
    import ctypes as C
    import os
    from openifem_amd import host, capi
    prm = open(os.path.join(os.path.dirname(__file__), "golden", "prm", "fluid_cylinder_mpi.prm")).read()

    def run(cells):
        flow = host.InsIM(prm, mesh="cylinder")
        flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0)
        flow.setup(3)
        tun = capi.Tuning()
        flow.L.ifem_default_tuning(C.byref(tun))
        assert tun.vcycle_graph_cells == 262144
        tun.vcycle_graph_cells = cells
        for c in flow.all_ctxs():
            assert flow.L.ifem_set_tuning(c, C.byref(tun)) == 0
        flow.run_one_step(True)
        flow.run_one_step(False)
        v, p = flow.get_current_solution()
        nit = flow.last_newton()
        stats = capi.vcycle_graph_stats(flow.L, flow.ctx)
        ainv = flow.opts.ainv_kind
        flow.close()
        return v, p, nit, stats, ainv

    v0, p0, n0, s0, k0 = run(0)
    v1, p1, n1, s1, k1 = run(262144)
    assert k0 == k1 == capi.AINV_MG
    assert s0 == (0, 0)
    assert s1[0] >= 1 and s1[1] > 10 * s1[0], s1  # captured a few times (the bounds move with the first assemblies), replayed many times
    assert n0 == n1
    assert np.abs(v0 - v1).max() <= 1e-9 * np.abs(v0).max() and np.abs(p0 - p1).max() <= 1e-9 * np.abs(p0).max()


def test_single_reduction_cg_gives_the_iterates_of_the_textbook_recurrence():
    """ifem_tuning::cg_single_reduction: the Chronopoulos / Gear form of the device-resident pressure CGs (one fused reduction per iteration:
    linalg.hip::cg1_*) against the two-reduction recurrence on the same solve -- same iteration counts (the checks come in bursts of four
    either way), the same Newton update to rounding, and the reference's stopping rule on the true residual"""
    import ctypes as C
    from openifem_amd import capi
    out = {}
    for single in (0, 1):
        s = _hierarchy((16, 16, 16))
        tun = capi.Tuning()
        s.L.ifem_default_tuning(C.byref(tun))
        assert tun.cg_single_reduction == 1
        tun.cg_single_reduction = single
        for c in s.all_ctxs():
            assert s.L.ifem_set_tuning(c, C.byref(tun)) == 0
        s.channel_state()
        s.opts.sm_mg = 0  # plain CG on S_m too: both pressure solves go through cg_device
        s.assemble(False)
        st = s.solve(False)
        _, n_u, n_p = s.sizes()
        x = _get(s, capi.VEC_UPDATE, n_u + n_p)
        res, bn = s.true_residual()
        assert res <= 1.05e-4 * bn
        out[single] = (x, st.fgmres_iters, st.cg_mp_iters, st.cg_sm_iters)
        s.close()
    (x0, f0, m0, s0), (x1, f1, m1, s1) = out[0], out[1]
    assert f0 == f1 and m0 == m1 and abs(s0 - s1) <= 4 * f0, out
    assert np.abs(x0 - x1).max() <= 1e-6 * np.abs(x0).max()


def test_inner_gmres_lengthens_its_restart_cycle_when_it_stagnates():
    """an application of A~^-1 that needs more than two restart cycles doubles the restart length of the context's inner GMRES for the
    applications that follow (solver.hip::precond_vmult; the refined cylinder needed it: 147 -> 59 inner iterations per application).  Forced
    here with GMRES(2) and a tight inner tolerance on a small channel: the length grows, the outer solve keeps the reference's stopping rule
    and the second solve needs no more inner iterations than the first"""
    s = _hierarchy((16, 16, 16))
    s.channel_state()
    s.opts.inner_restart = 2
    s.opts.inner_rel = 1e-7
    s.opts.inner_rel_first = 0.0
    s.assemble(False)
    assert s.L.ifem_inner_restart_length(s.ctx) == 0
    st1 = s.solve(False)
    grown = s.L.ifem_inner_restart_length(s.ctx)
    assert grown >= 4, grown
    res, bn = s.true_residual()
    assert res <= 1.05e-4 * bn
    st2 = s.solve(False)
    res, bn = s.true_residual()
    assert res <= 1.05e-4 * bn
    assert s.L.ifem_inner_restart_length(s.ctx) >= grown
    assert st2.inner_iters <= st1.inner_iters, (st1.inner_iters, st2.inner_iters)
    s.close()


def test_stored_uu_0_newton_step_against_the_oracles_assembled_matrix():
    """ifem_tuning::stored_uu = 0: no A_uu values are kept -- the assembly integrates the right-hand side, the outer
    operator applies A_uu matrix-free in fp64.  Same right-hand side as the stored assembly (1e-12), and the Newton update meets the
    reference's stopping rule on the ORACLE's assembled matrix; the same solve with the block CSR gives the same update."""
    import ctypes as C
    import orc
    from boxmesh import BoxMesh
    from cases import channel3d_state
    from openifem_amd import capi, host
    n = 8
    upd = {}
    rhs = {}
    for stored in (1, 0):
        s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), EXTENT)
        s.set_node_order(morton=False)  # lexicographic: the oracle's numbering
        s.setup(0)
        tun = capi.Tuning()
        s.L.ifem_default_tuning(C.byref(tun))
        tun.stored_uu = stored
        for c_ in s.all_ctxs():
            assert s.L.ifem_set_tuning(c_, C.byref(tun)) == 0
        s.channel_state()
        s.opts.inner_rel_first = 0.0
        s.assemble(False)
        st = s.solve(False)
        nl = sum(s.sizes()[1:])
        b, u = np.zeros(nl), np.zeros(nl)
        assert s.L.ifem_vec_get(s.ctx, capi.VEC_RHS, b.ctypes.data_as(C.c_void_p)) == 0
        assert s.L.ifem_vec_get(s.ctx, capi.VEC_UPDATE, u.ctypes.data_as(C.c_void_p)) == 0
        ev, pr = np.zeros(nl), np.zeros(nl)
        assert s.L.ifem_vec_get(s.ctx, capi.VEC_EVAL, ev.ctypes.data_as(C.c_void_p)) == 0
        assert s.L.ifem_vec_get(s.ctx, capi.VEC_PRESENT, pr.ctypes.data_as(C.c_void_p)) == 0
        rhs[stored], upd[stored] = b, u
        res, bn = s.true_residual()  # with the operator the solve used
        assert res <= 1.05e-4 * bn
        if stored == 0:
            assert s.L.ifem_nnz(s.ctx, 0) > 0  # the pattern exists, the values were never allocated:
            y = np.zeros(nl)
            assert s.L.ifem_uu_vmult(s.ctx, capi.VEC_UPDATE, capi.VEC_RHS, 0) < 0 and b"stored" in s.L.ifem_last_error()
        s.close()
    assert np.abs(rhs[0] - rhs[1]).max() <= 1e-12 * np.abs(rhs[1]).max()
    assert np.abs(upd[0] - upd[1]).max() <= 5e-4 * np.abs(upd[1]).max()
    # the oracle's matrix at the same state
    m = BoxMesh([n] * 3, (0, 0, 0), EXTENT, kv=2)
    dofs, vals, present, _, kw = channel3d_state(m)
    assert np.abs(present - pr).max() <= 1e-14  # same numbering, same analytic state; the perturbed point comes from the mirror's generator
    evalp = ev
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    S.assemble(orc.make_params(**kw), False, evalp, present)
    A, bo = S.csr("A"), S.rhs()
    assert np.abs(bo - rhs[0]).max() <= 1e-11 * np.abs(bo).max()
    free = np.ones(S.n, bool)
    free[dofs] = False
    assert np.linalg.norm((A @ upd[0] - bo)[free]) <= 1.05e-4 * np.linalg.norm(bo)


def test_stored_uu_0_refuses_what_it_cannot_do():
    import ctypes as C
    from openifem_amd import capi, host
    s = host.InsIM(host.channel_prm(3), (4, 4, 4), (0, 0, 0), EXTENT)
    s.setup(0)
    tun = capi.Tuning()
    s.L.ifem_default_tuning(C.byref(tun))
    tun.stored_uu = 0
    for c_ in s.all_ctxs():
        assert s.L.ifem_set_tuning(c_, C.byref(tun)) == 0
    s.channel_state()
    s.assemble(False)
    s.opts.ainv_kind = 0  # needs the stored block
    with pytest.raises(host.HostError) as e:
        s.solve(False)
    assert "stored_uu" in str(e.value)
    s.close()


def test_stored_uu_0_newton_loop_with_inhomogeneous_boundary_values():
    """tests/fluid_cylinder_mpi (parabolic inflow: non-zero constraint values) with ifem_tuning::stored_uu = 0: the first Newton iteration of the
    step -- the one assembled with nonzero_constraints -- takes the stored path by itself, the others run matrix-free; the reference's constants
    come out as with the block CSR"""
    import ctypes as C
    import os
    from openifem_amd import capi, host
    prm = open(os.path.join(os.path.dirname(__file__), "golden", "prm", "fluid_cylinder_mpi.prm")).read()
    out = {}
    for stored in (1, 0):
        flow = host.InsIM(prm, mesh="cylinder")
        flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0)
        flow.setup(3)
        tun = capi.Tuning()
        flow.L.ifem_default_tuning(C.byref(tun))
        tun.stored_uu = stored
        for c_ in flow.all_ctxs():
            assert flow.L.ifem_set_tuning(c_, C.byref(tun)) == 0
        flow.run_one_step(True)
        v, p = flow.get_current_solution()
        out[stored] = (v.max(), p.max(), flow.last_newton())
        flow.close()
    assert abs(out[0][0] - 0.374235) / 0.374235 < 1e-3 and abs(out[0][1] - 46.5226) / 46.5226 < 1e-3, out
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * out[1][0] and abs(out[0][1] - out[1][1]) <= 1e-5 * out[1][1], out
    assert out[0][2][0] == out[1][2][0], out  # the same number of Newton iterations


def test_restart_lengthening_is_one_decision_of_all_ranks():
    """the self-lengthening inner restart reads the free device memory, which differs between ranks -- ranks that chose
    different restart lengths would restart at different iterations and their all-reduces / halo exchanges would no longer pair up.  Two virtual
    ranks, one of which can afford 5 column pairs only (ifem_test_restart_fits): both end with the same length (the smaller wish), the solve
    completes and meets the stopping rule"""
    from openifem_amd import capi
    L = capi.load()
    P, n, world = (2, 1, 1), (8, 8, 8), 2
    probe = _hierarchy((16, 8, 8))
    depth = probe.L.ifem_mg_depth(probe.ctx)
    probe.close()
    worlds = [C.c_void_p(L.ifem_local_world_create(world)) for _ in range(depth + 1)]
    out, errs = [None] * world, []

    def work(rank):
        try:
            s = _hierarchy(n, P, rank, worlds)
            s.channel_state()
            assert L.ifem_halo_exchange(s.ctx, capi.VEC_EVAL) == 0
            s.opts.inner_restart = 2
            s.opts.inner_rel = 1e-7
            s.opts.inner_rel_first = 0.0
            assert L.ifem_test_restart_fits(s.ctx, 5 if rank == 0 else 100) == 0
            s.assemble(False)
            st = s.solve(False)
            res, bn = s.true_residual()
            out[rank] = (L.ifem_inner_restart_length(s.ctx), st.inner_iters, st.fgmres_iters, res / bn)
            s.close()
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs and all(o is not None for o in out), (errs, out)
    assert out[0][0] == out[1][0] and 2 < out[0][0] <= 5, out  # rank 1 alone would have doubled to 4, 8, ...: the smaller wish wins
    assert out[0][1:3] == out[1][1:3] and out[0][3] <= 1.05e-4, out
    for w in worlds:
        L.ifem_local_world_destroy(w)
