"""Generates tests/golden/cells.npz and tests/golden/newton_8.npz with the CPU oracle ("fixtures to generate here"):
single-cell element matrices / vectors of InsIM, InsIMEX-free SCnsIM and SUPGInsIM on affine and distorted 2D / 3D cells
with seeded inputs, the assembled 4 x 4 (x 4) system with Dirichlet elimination, and one Newton update of the 8^3
channel with an exact A_uu solve.  Run from the repository root:  python tests/golden/make_golden.py
The fixtures freeze the oracle: tests/test_oracle_golden.py compares the live oracle with them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from boxmesh import BoxMesh  # noqa: E402
from cases import channel3d_state  # noqa: E402


def cell_cases():
    out = {}
    for dim, kv, distort in [(2, 2, 0.0), (2, 2, 0.04), (3, 2, 0.0), (3, 2, 0.03), (2, 1, 0.04), (3, 1, 0.03)]:
        rng = np.random.default_rng(1000 * dim + 10 * kv + int(distort > 0))
        m = BoxMesh((2,) * dim, (0,) * dim, (1.0, 0.7, 0.5)[:dim], kv=kv)
        m.vcoords = m.vcoords + distort * rng.standard_normal(m.vcoords.shape)
        ev, pr, acc = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
        m.indicator = np.array([0, 1] * (m.n_cells // 2), np.int32)
        S = orc.System(m)
        tag = f"d{dim}k{kv}{'x' if distort else 'a'}"
        P = orc.make_params(mu=0.7, rho=1.3, gamma=0.2, dt=0.01, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5})
        for cell in (0, 1):
            Ke, Me, fe = S.cell(P, cell, ev, pr, acc)
            out[f"ins_{tag}_c{cell}_Ke"], out[f"ins_{tag}_c{cell}_Me"], out[f"ins_{tag}_c{cell}_fe"] = Ke, Me, fe
        nq = (kv + 1) ** dim
        st = S.update_stress(0.03, pr)
        fs = rng.standard_normal((dim * (dim + 1) // 2, m.n_unodes))
        sig, bf = rng.uniform(0, 3, (m.n_cells, nq)), rng.standard_normal((m.n_cells, nq, dim))
        out[f"stress_{tag}"] = st
        for form in (0, 1):
            Ps = orc.make_scns_params(mu=0.03, rho=1.2, dt=0.01, solid_rho=3.0, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5},
                                      stress=st, fsi_stress=fs, sigma_pml=sig, body_force=bf, formulation=form)
            for cell in (0, 1):
                Ke, fe = S.scns_cell(Ps, cell, ev * np.where(np.arange(m.n_dofs) >= m.n_u, 50.0, 1.0), pr, acc)
                out[f"scns{form}_{tag}_c{cell}_Ke"], out[f"scns{form}_{tag}_c{cell}_fe"] = Ke, fe
        out[f"inputs_{tag}"] = np.concatenate([ev, pr, acc, fs.ravel(), sig.ravel(), bf.ravel(), m.vcoords.ravel()])
    return out


def assembled_cases():
    out = {}
    for dim in (2, 3):
        rng = np.random.default_rng(77 + dim)
        m = BoxMesh((4, 4) if dim == 2 else (3, 3, 2), (0,) * dim, (1.0, 0.6, 0.4)[:dim], kv=2)
        flag = 3 if dim == 2 else 7
        dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)})
        ev, pr = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
        S = orc.System(m)
        S.set_constraints(0, dofs, None)
        S.set_constraints(1, dofs, vals)
        S.assemble(orc.make_params(mu=0.7, rho=1.3, gamma=0.2, dt=0.01, neumann={1: 2.5}), True, ev, pr)
        A = S.csr("A")
        out[f"asm{dim}_indptr"], out[f"asm{dim}_indices"], out[f"asm{dim}_data"] = A.indptr, A.indices, A.data
        out[f"asm{dim}_rhs"], out[f"asm{dim}_ev"], out[f"asm{dim}_pr"] = S.rhs(), ev, pr
    return out


def newton_case():
    m = BoxMesh((8, 8, 8), (0, 0, 0), (2.0, 0.2, 0.2), kv=2)
    dofs, vals, present, ev, kw = channel3d_state(m)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    P = orc.make_params(**kw)
    S.assemble(P, False, ev, present)
    S.opts.fgmres_rel = 1e-12
    rc, upd, it, res = S.solve(P, False, ainv=orc.SpluAinv())
    assert rc == 0
    return {"newton8_update": upd.astype(np.float64), "newton8_rhs_norm": np.array([np.linalg.norm(S.rhs())])}


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "cells.npz"), **cell_cases())
    np.savez_compressed(os.path.join(HERE, "assembled.npz"), **assembled_cases())
    np.savez_compressed(os.path.join(HERE, "newton_8.npz"), **newton_case())
    for f in ("cells.npz", "assembled.npz", "newton_8.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
