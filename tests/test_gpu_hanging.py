"""Hanging-node constraints through the C ABI (ifem_set_hanging_constraints) against the oracle's literal restatement of
AffineConstraints::distribute_local_to_global with hanging lines (oracle.c::orc_ins_assemble_affine_dense) on the
smallest one-irregular meshes (tests/hangmesh.py): condensed operator, condensed right-hand side, and the Newton update
with constraints.distribute applied -- reference: DoFTools::make_hanging_node_constraints, mpi_fluid_solver.cpp:182-184;
mpi_insim.cpp:343-355,390."""
import numpy as np
import pytest

import orc
from hangmesh import HangingMesh

pytestmark = pytest.mark.gpu


def _capi():
    from openifem_amd import capi
    return capi


def _mesh(dim, kv):
    if dim == 2:
        return HangingMesh((3, 2), (0, 0), (1.5, 0.8), {(0, 0), (2, 1)}, kv=kv)
    return HangingMesh((2, 2, 2), (0, 0, 0), (1.0, 0.8, 0.6), {(0, 0, 0)}, kv=kv)


def _setup(dim, kv, use_nonzero, seed):
    capi = _capi()
    m = _mesh(dim, kv)
    assert len(m.hang_dof) > 0
    rng = np.random.default_rng(seed)
    flag = 3 if dim == 2 else 7
    # inflow profile on x- (some masters of hanging lines on the boundary carry a value), no-slip on y-
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)},
                             {0: lambda p, c: 0.3 + 0.5 * p[1] if c == 0 else 0.1 * p[1]})
    ev, pr = 0.3 * rng.standard_normal(m.n_dofs), 0.3 * rng.standard_normal(m.n_dofs)
    kw = dict(mu=0.7, rho=1.3, gamma=0.1, dt=0.05, g=(0.2, -9.8, 0.4)[:dim], neumann={1: 2.0})
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    Ao, bo = S.assemble_affine_dense(orc.make_params(**kw), use_nonzero, ev, pr, m)
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.set_hanging_constraints(m.hang_dof, m.hang_ptr, m.hang_master, m.hang_weight)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    ctx.vec_set(capi.VEC_EVAL, ev)
    ctx.assemble(capi.make_params(**kw), use_nonzero)
    return m, ctx, capi, kw, Ao, bo, (dofs, vals), rng


@pytest.mark.parametrize("dim,kv", [(2, 2), (2, 1), (3, 2), (3, 1)])
@pytest.mark.parametrize("use_nonzero", [True, False])
def test_condensed_operator_and_rhs_match_distribute_local_to_global(dim, kv, use_nonzero):
    m, ctx, capi, kw, Ao, bo, _, rng = _setup(dim, kv, use_nonzero, 5 + dim + kv)
    hang = m.hang_dof
    reg = np.setdiff1d(np.arange(m.n_dofs), hang)
    # right-hand side: every row, the hanging rows carry diag * inhomogeneity
    b = ctx.vec_get(capi.VEC_RHS)
    assert np.abs(b[reg] - bo[reg]).max() <= 1e-11 * np.abs(bo).max()
    # operator: columns by unit-free probing with random vectors
    for _ in range(3):
        x = rng.standard_normal(m.n_dofs)
        y = ctx.system_vmult(x)
        yo = Ao @ x
        assert np.abs(y[reg] - yo[reg]).max() <= 1e-11 * np.abs(yo).max()
        # hanging rows: decoupled, positive diagonal (deal.II: |Ke_hh| summed over the cells; this build: the diagonal of
        # the unconstrained A_uu for velocity dofs, the mean of those for pressure dofs)
        d = y[hang] / x[hang]
        assert np.all(d > 0)
        hu = hang < m.n_u
        assert np.abs(d[hu] - Ao[hang[hu], hang[hu]]).max() <= 1e-10 * np.abs(Ao.diagonal()).max()
        assert np.abs(Ao[hang][:, reg]).max() == 0 and np.abs(Ao[reg][:, hang]).max() == 0
    ctx.close()


@pytest.mark.parametrize("dim,kv", [(2, 2), (3, 2)])
def test_newton_update_with_hanging_nodes_matches_dense_solve(dim, kv):
    m, ctx, capi, kw, Ao, bo, (dofs, vals), rng = _setup(dim, kv, True, 11 + dim + kv)
    ctx.opts.fgmres_rel = 1e-10
    ctx.opts.inner_rel = 1e-3
    st = ctx.solve(capi.make_params(**kw), True)
    upd = ctx.vec_get(capi.VEC_UPDATE)
    xo = np.linalg.solve(Ao, bo)
    # constraints.distribute: Dirichlet entries are their values already; hanging entries from all masters
    xo = m.prolongation() @ xo
    assert st.fgmres_iters < 200
    assert np.abs(upd - xo).max() <= 1e-6 * np.abs(xo).max()
    # the update is conforming: hanging values equal the interpolation of their masters
    assert np.abs(upd - m.prolongation() @ upd).max() <= 1e-12 * np.abs(upd).max()
    ctx.close()


def test_conforming_poiseuille_is_reproduced_on_a_hanging_node_mesh():
    # plane Poiseuille (tests/fluid_pressure_driven) is quadratic: exactly representable in Q2 on ANY conforming
    # mesh, so with the hanging lines in place the time loop must land on Umax = dP H^2 / (8 mu L) = 2.5e-2 although the
    # mesh is refined non-uniformly; without the lines the discrete space is non-conforming and the answer is off
    capi = _capi()
    m = HangingMesh((4, 2), (0, 0), (2.0, 0.2), {(1, 0), (2, 1)}, kv=2)
    dofs, vals = m.dirichlet({2: (3, [0, 0]), 3: (3, [0, 0])})
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.set_hanging_constraints(m.hang_dof, m.hang_ptr, m.hang_master, m.hang_weight)
    ctx.opts.fgmres_rel = 1e-8
    P = capi.make_params(mu=1.0, rho=1.0, gamma=0.1, dt=1e-3, neumann={0: 10.0})
    x = np.zeros(m.n_dofs)
    ctx.vec_set(capi.VEC_PRESENT, x)
    for step in range(80):
        rc, _ = ctx.newton_step(P, step == 0)
        assert rc > 0
    v = ctx.vec_get(capi.VEC_PRESENT)[:m.n_u].reshape(-1, 2)
    y = m.unode_coords[:, 1]
    exact = 10.0 / (2 * 1.0 * 2.0) * y * (0.2 - y)
    assert abs(v[:, 0].max() - 2.5e-2) / 2.5e-2 < 1e-6
    assert np.abs(v[:, 0] - exact).max() < 1e-7 and np.abs(v[:, 1]).max() < 1e-7
    ctx.close()


@pytest.mark.parametrize("dim,form", [(2, 0), (3, 0), (2, 1)])
def test_scnsim_with_hanging_nodes(dim, form):
    # SCnsIM / SUPGInsIM (Q1/Q1, BASELINE config 5 is such a mesh): the condensed system is C^T A^ C with the Dirichlet
    # masters closed away -- shown above to be what distribute_local_to_global produces, independently of the integrand --
    # so the oracle's ordinary SCnsIM assembly (hanging dofs as regular ones) condensed in numpy is the reference here
    capi = _capi()
    m = _mesh(dim, 1)
    rng = np.random.default_rng(31 + dim + form)
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)},
                             {0: lambda p, c: 0.3 + 0.5 * p[1] if c == 0 else 0.1 * p[1]})
    ev, pr = 0.3 * rng.standard_normal(m.n_dofs), 0.3 * rng.standard_normal(m.n_dofs)
    kw = dict(mu=0.05, rho=1.2, dt=0.01, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5})
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    S.scns_assemble(orc.make_scns_params(formulation=form, **kw), True, ev, pr)
    Ah, bh = S.csr("A").toarray(), S.rhs()
    Cm = m.prolongation()
    isc = np.zeros(m.n_dofs, bool)
    isc[dofs] = True
    cv = np.zeros(m.n_dofs)
    cv[dofs] = vals
    Cc, c0 = Cm.copy(), np.zeros(m.n_dofs)
    for d in m.hang_dof:
        c0[d] = Cm[d, isc] @ cv[isc]
        Cc[d, isc] = 0
    Ao, bo = Cc.T @ Ah @ Cc, Cc.T @ (bh - Ah @ c0)
    hang = m.hang_dof
    reg = np.setdiff1d(np.arange(m.n_dofs), hang)
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.set_hanging_constraints(m.hang_dof, m.hang_ptr, m.hang_master, m.hang_weight)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    ctx.vec_set(capi.VEC_EVAL, ev)
    ctx.update_stress(kw["mu"])
    S2 = orc.System(m)  # the projected stress enters the SCnsIM integrand: same input on both sides
    st_o = S2.update_stress(kw["mu"], pr)
    S.scns_assemble(orc.make_scns_params(formulation=form, stress=st_o, **kw), True, ev, pr)
    Ah, bh = S.csr("A").toarray(), S.rhs()
    Ao, bo = Cc.T @ Ah @ Cc, Cc.T @ (bh - Ah @ c0)
    ctx.scns_assemble(capi.make_scns_params(formulation=form, **kw), True)
    b = ctx.vec_get(capi.VEC_RHS)
    assert np.abs(b[reg] - bo[reg]).max() <= 1e-11 * np.abs(bo).max()
    x = rng.standard_normal(m.n_dofs)
    y, yo = ctx.system_vmult(x), Ao @ x
    assert np.abs(y[reg] - yo[reg]).max() <= 1e-11 * np.abs(yo).max()
    # solve at the reference tolerance 1e-6 ||rhs|| (mpi_supg_solver.cpp:303-305): residual with the ORACLE's condensed
    # matrix on the regular rows, the update against the dense solve, hanging entries interpolated (distribute)
    ctx.scns_solve(True)
    upd = ctx.vec_get(capi.VEC_UPDATE)
    r = bo[reg] - Ao[np.ix_(reg, reg)] @ upd[reg]
    assert np.linalg.norm(r) <= 1.01e-6 * np.linalg.norm(b)
    xo = np.zeros(m.n_dofs)
    xo[reg] = np.linalg.solve(Ao[np.ix_(reg, reg)], bo[reg])
    xo = Cm @ xo
    assert np.abs(upd - xo).max() <= 1e-4 * np.abs(xo).max()
    assert np.abs(upd - Cm @ upd).max() <= 1e-12 * np.abs(upd).max()
    ctx.close()


def test_insimex_with_hanging_nodes():
    # InsIMEX (mpi_insimex.cpp:150-355): matrix assembled once, then right-hand-side-only assemblies; both go through the
    # same condensation.  Reference here: the oracle's ordinary IMEX assembly condensed in numpy (see the SCnsIM test)
    capi = _capi()
    m = _mesh(2, 2)
    rng = np.random.default_rng(77)
    dofs, vals = m.dirichlet({0: (3, [0.3, -0.2]), 2: (3, [0.0, 0.0])}, {0: lambda p, c: 0.3 + 0.5 * p[1] if c == 0 else 0.1 * p[1]})
    pr = 0.3 * rng.standard_normal(m.n_dofs)
    kw = dict(mu=0.7, rho=1.3, gamma=0.1, dt=0.05, g=(0.2, -9.8), neumann={1: 2.0})
    Cm = m.prolongation()
    isc = np.zeros(m.n_dofs, bool)
    isc[dofs] = True
    cv = np.zeros(m.n_dofs)
    cv[dofs] = vals
    Cc, c0 = Cm.copy(), np.zeros(m.n_dofs)
    for d in m.hang_dof:
        c0[d] = Cm[d, isc] @ cv[isc]
        Cc[d, isc] = 0
    reg = np.setdiff1d(np.arange(m.n_dofs), m.hang_dof)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    ctx = capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.set_hanging_constraints(m.hang_dof, m.hang_ptr, m.hang_master, m.hang_weight)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    for use_nonzero, assemble_system in ((True, True), (False, True), (False, False)):
        S.imex_assemble(orc.make_params(**kw), use_nonzero, assemble_system, pr)
        Ah, bh = S.csr("A").toarray(), S.rhs()
        off = c0 if use_nonzero else 0 * c0
        Ao, bo = Cc.T @ Ah @ Cc, Cc.T @ (bh - Ah @ off)
        ctx.imex_assemble(capi.make_params(**kw), use_nonzero, assemble_system)
        b = ctx.vec_get(capi.VEC_RHS)
        assert np.abs(b[reg] - bo[reg]).max() <= 1e-11 * np.abs(bo).max()
        x = rng.standard_normal(m.n_dofs)
        y, yo = ctx.system_vmult(x), Ao @ x
        assert np.abs(y[reg] - yo[reg]).max() <= 1e-11 * np.abs(yo).max()
    ctx.imex_solve(capi.make_params(**kw), False)
    upd = ctx.vec_get(capi.VEC_UPDATE)
    xo = np.zeros(m.n_dofs)
    xo[reg] = np.linalg.solve(Ao[np.ix_(reg, reg)], bo[reg])
    xo = Cm @ xo
    assert np.abs(upd - xo).max() <= 1e-6 * np.abs(xo).max()
    ctx.close()
