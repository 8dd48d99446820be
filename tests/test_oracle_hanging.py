"""The oracle's restatement of AffineConstraints::distribute_local_to_global with hanging-node lines
(oracle.c::orc_ins_assemble_affine_dense): structural identity with the global condensation C^T A^ C and the analytic
Poiseuille answer on a one-irregular mesh (the reference holds no test that dumps matrices: parity of this piece is
pinned by these two properties)."""
import numpy as np
import pytest

import orc
from hangmesh import HangingMesh


def _closed(m, dofs, vals):
    Cm = m.prolongation()
    isc = np.zeros(m.n_dofs, bool)
    isc[dofs] = True
    cv = np.zeros(m.n_dofs)
    cv[dofs] = vals
    Cc, c0 = Cm.copy(), np.zeros(m.n_dofs)
    for d in m.hang_dof:
        c0[d] = Cm[d, isc] @ cv[isc]
        Cc[d, isc] = 0
    return Cm, Cc, c0


@pytest.mark.parametrize("dim,kv", [(2, 2), (2, 1), (3, 2), (3, 1)])
def test_cellwise_distribution_equals_global_condensation(dim, kv):
    m = (HangingMesh((3, 2), (0, 0), (1.5, 0.8), {(0, 0), (2, 1)}, kv=kv) if dim == 2
         else HangingMesh((2, 2, 2), (0, 0, 0), (1.0, 0.8, 0.6), {(0, 0, 0)}, kv=kv))
    assert np.allclose(np.add.reduceat(m.hang_weight, m.hang_ptr[:-1]), 1.0)  # interpolation reproduces constants
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)},
                             {0: lambda p, c: 0.3 + 0.5 * p[1] if c == 0 else 0.1 * p[1]})
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    rng = np.random.default_rng(dim * 10 + kv)
    ev, pr = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
    P = orc.make_params(mu=0.7, rho=1.3, gamma=0.1, dt=0.05, g=(0.2, -9.8, 0.4)[:dim], neumann={1: 2.0})
    for use_nonzero in (True, False):
        A, b = S.assemble_affine_dense(P, use_nonzero, ev, pr, m)
        S.assemble(P, use_nonzero, ev, pr)
        Ah, bh = S.csr("A").toarray(), S.rhs()
        _, Cc, c0 = _closed(m, dofs, vals if use_nonzero else 0 * vals)
        reg = np.setdiff1d(np.arange(m.n_dofs), m.hang_dof)
        A2, b2 = Cc.T @ Ah @ Cc, Cc.T @ (bh - Ah @ c0)
        assert np.abs(A[np.ix_(reg, reg)] - A2[np.ix_(reg, reg)]).max() < 1e-13 * np.abs(A).max()
        assert np.abs(b[reg] - b2[reg]).max() < 1e-13 * np.abs(b).max()
        # hanging rows and columns are decoupled, with a positive diagonal
        h = m.hang_dof
        off = A[h].copy()
        off[np.arange(len(h)), h] = 0
        assert np.abs(off).max() == 0 and np.abs(A[np.ix_(reg, h)]).max() == 0 and np.all(A[h, h] > 0)


def test_poiseuille_on_a_hanging_node_mesh():
    # plane Poiseuille is quadratic, hence exact in Q2 on a conforming space: with the hanging lines the non-uniformly
    # refined channel lands on Umax = dP H^2 / (8 mu L) = 2.5e-2 (tests/fluid_pressure_driven's known answer)
    m = HangingMesh((4, 2), (0, 0), (2.0, 0.2), {(1, 0), (2, 1)}, kv=2)
    dofs, vals = m.dirichlet({2: (3, [0, 0]), 3: (3, [0, 0])})
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    P = orc.make_params(mu=1.0, rho=1.0, gamma=0.1, dt=1e-3, neumann={0: 10.0})
    Cm = m.prolongation()
    x = np.zeros(m.n_dofs)
    for step in range(80):
        ev = x.copy()
        for it in range(3):
            A, b = S.assemble_affine_dense(P, step == 0 and it == 0, ev, x, m)
            ev += Cm @ np.linalg.solve(A, b)  # constraints.distribute
        x = ev
    v = x[:m.n_u].reshape(-1, 2)
    y = m.unode_coords[:, 1]
    assert abs(v[:, 0].max() - 2.5e-2) / 2.5e-2 < 1e-6
    assert np.abs(v[:, 0] - 10.0 / 4.0 * y * (0.2 - y)).max() < 1e-8
