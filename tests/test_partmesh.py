"""CPU checks of the test-side partitioner (tests/partmesh.py) that feeds ifem_partition / local hanging lines to the
virtual-rank GPU tests: ownership is a partition of the dofs, halo plans are symmetric and ordered as the receiver expects,
masters of every local hanging dof are local."""
import numpy as np
import pytest

from hangmesh import HangingMesh
from partmesh import local_dirichlet, partition_mesh


@pytest.mark.parametrize("dim,kv,nranks", [(2, 1, 4), (2, 2, 2), (3, 2, 4)])
def test_partition_invariants(dim, kv, nranks):
    if dim == 2:
        m = HangingMesh((6, 4), (0, 0), (3.0, 1.6), {(1, 1), (2, 1), (2, 2), (4, 0), (3, 3)}, kv=kv)
    else:
        m = HangingMesh((3, 2, 2), (0, 0, 0), (1.5, 0.8, 0.6), {(0, 0, 0), (2, 1, 1)}, kv=kv)
    c = m.vcoords.mean(axis=1)
    mid = 0.5 * (c.min(axis=0) + c.max(axis=0))
    rank = (c[:, 0] > mid[0]).astype(int) + (2 * (c[:, 1] > mid[1]).astype(int) if nranks == 4 else 0)
    parts = partition_mesh(m, rank, nranks)
    owned = np.concatenate([P.own_gdof for P in parts])
    assert sorted(owned) == list(range(m.n_dofs))
    for P in parts:
        assert (P.ext_gdof[:dim * P.n_unodes_owned] == P.own_gdof[:dim * P.n_unodes_owned]).all()
        for kind in "up":
            sp_, si, rp = getattr(P, "send_%s_ptr" % kind), getattr(P, "send_%s_idx" % kind), getattr(P, "recv_%s_ptr" % kind)
            l2g, n_own = getattr(P, "l2g_" + kind), getattr(P, "n_%snodes_owned" % kind)
            assert rp[-1] == len(l2g) - n_own and (si < n_own).all()
            for k, q in enumerate(P.neighbors):
                Q = parts[q]
                me = list(Q.neighbors).index(P.rank)
                qrp, ql2g, qn = getattr(Q, "recv_%s_ptr" % kind), getattr(Q, "l2g_" + kind), getattr(Q, "n_%snodes_owned" % kind)
                assert (l2g[si[sp_[k]:sp_[k + 1]]] == ql2g[qn + qrp[me]:qn + qrp[me + 1]]).all()
        # every cell touching an owned node is assembled here; hanging lines are local and closed
        touched = set(int(x) for x in np.nonzero((np.isin(m.cell_unodes, P.owned_u)).any(axis=1))[0])
        assert touched <= set(int(x) for x in P.cells)
        assert (P.hang_master >= 0).all() and not set(P.hang_dof) & set(P.hang_master)
        g = {int(P.ext_gdof[d]): i for i, d in enumerate(P.hang_dof)}
        for gd, i in g.items():
            j = list(m.hang_dof).index(gd)
            assert (P.ext_gdof[P.hang_master[P.hang_ptr[i]:P.hang_ptr[i + 1]]] == m.hang_master[m.hang_ptr[j]:m.hang_ptr[j + 1]]).all()
    dofs, vals = m.dirichlet({0: (2 ** dim - 1, [0.3] * dim)})
    n_local_lines = sum(len(local_dirichlet(P, dofs, vals)[0]) for P in parts)
    assert n_local_lines >= len(dofs)
