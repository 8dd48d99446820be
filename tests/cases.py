"""Shared synthetic cases: the 3D pressure-driven channel used by the bench and parity tests."""
import numpy as np

CHANNEL_BCS = {2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (4, [0]), 5: (4, [0])}
CHANNEL_KW = dict(mu=1.0, rho=1.0, gamma=0.1, dt=1e-3, neumann={0: 10.0})


def channel3d_state(m, seed=1234, rel=1e-3):
    """present = analytic plane Poiseuille, evaluation point = present + seeded perturbation."""
    dofs, vals = m.dirichlet(CHANNEL_BCS)
    n_u = m.dim * m.n_unodes
    y = m.unode_coords[:, 1]
    present = np.zeros(m.n_dofs)
    present[0:n_u:3] = 10.0 / (2 * 2.0) * y * (0.2 - y)
    present[n_u:] = 10.0 * (1 - m.pnode_coords[:, 0] / 2.0)
    rng = np.random.default_rng(seed)
    pert = rng.uniform(-1, 1, m.n_dofs)
    ev = present.copy()
    ev[:n_u] += rel * 0.025 * pert[:n_u]
    ev[n_u:] += rel * 10.0 * pert[n_u:]
    ev[dofs] = present[dofs]
    return dofs, vals, present, ev, dict(CHANNEL_KW)
