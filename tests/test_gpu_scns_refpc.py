"""The SCnsIM / SUPG block preconditioner in the reference's structure (mpi_supg_solver.cpp:19-192): ILU(0) of A_vv,
T_pp as an operator, B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp and its ILU(0) -- every piece against the oracle's restatement
(oracle.c::supg_pc_setup, Euclid = ILU(0) in the natural order), and the iteration counts of the whole solve beside the oracle's."""
import ctypes as C

import numpy as np
import pytest

import orc
from boxmesh import BoxMesh

pytestmark = pytest.mark.gpu


def _capi():
    import openifem_amd.capi as capi
    return capi


def _ctx(m):
    capi = _capi()
    return capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)


def _tune(ctx, **kw):
    capi = _capi()
    t = capi.Tuning()
    ctx.L.ifem_default_tuning(C.byref(t))
    for k, v in kw.items():
        setattr(t, k, v)
    assert ctx.L.ifem_set_tuning(ctx.h, C.byref(t)) == 0, ctx.L.ifem_last_error()


def _probe(ctx, which, x):
    x = np.ascontiguousarray(x, float)
    y = np.zeros(len(x))
    rc = ctx.L.ifem_scns_pc_probe(ctx.h, which, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    assert rc == 0, ctx.L.ifem_last_error()
    return y


def _box_case(dim, kv, reps, seed=5):
    capi = _capi()
    rng = np.random.default_rng(seed)
    m = BoxMesh(reps, (0,) * dim, (1.0, 0.6, 0.4)[:dim], kv=kv)
    m.vcoords = m.vcoords + 0.01 * rng.standard_normal(m.vcoords.shape)
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim), 3: (flag, [0.0] * dim)})
    ev, pr = 0.3 * rng.standard_normal(m.n_dofs), 0.3 * rng.standard_normal(m.n_dofs)
    kw = dict(mu=0.03, rho=1.2, dt=0.01)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    S.scns_assemble(orc.make_scns_params(**kw), True, ev, pr)
    ctx = _ctx(m)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    ctx.vec_set(capi.VEC_EVAL, ev)
    ctx.scns_assemble(capi.make_scns_params(**kw), True)
    return m, S, ctx, rng


@pytest.mark.parametrize("dim,kv,reps", [(2, 1, (9, 7)), (3, 1, (4, 4, 3)), (2, 2, (5, 4)), (3, 2, (3, 2, 2))])
def test_pieces_match_the_oracle(dim, kv, reps):
    """exact substitution on the device (sweeps < 0) = Euclid's ILU(0) of the oracle, entry for entry of the result: the block
    ILU(0) on velocity nodes IS the scalar ILU(0) on the reference's pattern of full dim x dim blocks"""
    m, S, ctx, rng = _box_case(dim, kv, reps)
    _tune(ctx, pvv_sweeps=-1, b2pp_sweeps=-1)
    n_u, n_p = m.n_u, m.n_pnodes
    xu, xp = rng.standard_normal(n_u), rng.standard_normal(n_p)
    for which, x in ((2, xp), (0, xu), (1, xp), (3, xp)):
        yo = S.scns_pc_probe(which, x)
        y = _probe(ctx, which, x)
        assert np.abs(y - yo).max() <= 1e-9 * np.abs(yo).max(), (which, np.abs(y - yo).max() / np.abs(yo).max())
    ctx.close()


def test_jacobi_sweeps_converge_to_the_exact_substitution():
    """k sweeps reproduce the first k terms of the (finite) Neumann series of each triangular inverse: as many sweeps as
    elimination levels give the exact solve; a handful are within a fraction of it"""
    m, S, ctx, rng = _box_case(2, 1, (9, 7))
    n_u, n_p = m.n_u, m.n_pnodes
    xu, xp = rng.standard_normal(n_u), rng.standard_normal(n_p)
    _tune(ctx, pvv_sweeps=-1, b2pp_sweeps=-1)
    eu, ep = _probe(ctx, 0, xu), _probe(ctx, 1, xp)
    err = {}
    for k in (1, 3, 5, 40):
        _tune(ctx, pvv_sweeps=k, b2pp_sweeps=k)
        err[k] = (np.abs(_probe(ctx, 0, xu) - eu).max() / np.abs(eu).max(), np.abs(_probe(ctx, 1, xp) - ep).max() / np.abs(ep).max())
    assert err[40][0] < 1e-12 and err[40][1] < 1e-12, err
    assert err[5][0] < err[3][0] < err[1][0] and err[5][1] < err[3][1] < err[1][1], err
    assert err[3][0] < 0.2 and err[5][1] < 0.2, err
    ctx.close()


def _cylinder(refinements):
    from cylmesh import CylinderMesh
    capi = _capi()
    m = CylinderMesh(refinements, kv=1)

    def inflow(p, c):
        return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0

    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    ctx = _ctx(m)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    kw = dict(mu=1.8e-4, rho=1.3e-3, dt=1e-2)
    ctx.scns_assemble(capi.make_scns_params(**kw), True)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    zero = np.zeros(S.n)
    S.scns_assemble(orc.make_scns_params(**kw), True, zero, zero)
    return m, S, ctx, dofs, vals


@pytest.mark.parametrize("refinements", [1, 3])
def test_solve_iteration_counts_beside_the_oracle(refinements):
    """tests/fluid_cylinder_mpi_scnsim, first Newton iteration: the HIP solve with the reference's preconditioner structure needs
    the outer iterations of the oracle's restatement of it (+ 2 at most: Jacobi-sweep substitutions, right- instead of
    left-preconditioned inner GMRES), and its inner count stays within 1.6 x; the solution meets 1e-6 ||rhs||"""
    capi = _capi()
    m, S, ctx, dofs, vals = _cylinder(refinements)
    st = ctx.scns_solve(True)
    upd = ctx.vec_get(capi.VEC_UPDATE)
    rc, upd_o, counts, res = S.scns_solve(True)
    assert rc == 0
    A, b = S.csr("A"), S.rhs()
    free = np.ones(S.n, bool)
    free[dofs] = False
    assert np.linalg.norm((A @ upd - b)[free]) <= 1.05e-6 * np.linalg.norm(b)
    assert st.fgmres_iters <= counts[0] + 2, (st.fgmres_iters, counts)
    assert st.inner_iters <= 1.6 * counts[1] + 10, (st.inner_iters, counts)
    # both are solutions of the same system to 1e-6 ||rhs||
    assert np.abs(upd - upd_o).max() <= 2e-4 * np.abs(upd_o).max()
    if refinements == 3:
        assert st.fgmres_iters <= 15, st.fgmres_iters  # (the node-block Jacobi P_vv of rounds 2-5 needed 53)
    ctx.close()


def test_legacy_structure_is_still_selectable():
    """ifem_tuning::scns_pc = 1: node-block Jacobi P_vv, explicit T_pp (rounds 2-5) -- same solution, more outer iterations"""
    capi = _capi()
    m, S, ctx, dofs, vals = _cylinder(1)
    st2 = ctx.scns_solve(True)
    u2 = ctx.vec_get(capi.VEC_UPDATE)
    _tune(ctx, scns_pc=1)
    st1 = ctx.scns_solve(True)
    u1 = ctx.vec_get(capi.VEC_UPDATE)
    assert np.abs(u1 - u2).max() <= 2e-4 * np.abs(u2).max()
    assert st2.fgmres_iters < st1.fgmres_iters, (st2.fgmres_iters, st1.fgmres_iters)
    ctx.close()
