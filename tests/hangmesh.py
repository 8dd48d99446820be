"""A box mesh whose coarse cells are refined once where asked: the smallest meshes with hanging nodes (one-irregular, as
p4est / deal.II keep them).  Test-side stand-in for parallel::distributed::Triangulation + execute_coarsening_and_
refinement and for DoFTools::make_hanging_node_constraints: same attributes as BoxMesh plus the hanging lines
(dof, ptr, master, weight) in the local dof numbering of the block vector [dim * unode + c | n_u + pnode]."""
import itertools

import numpy as np


def _lagrange(k, t):
    """values at t of the k+1 Lagrange polynomials on the equidistant nodes j/k of [0,1]"""
    xs = np.arange(k + 1) / k
    out = np.ones(k + 1)
    for i in range(k + 1):
        for j in range(k + 1):
            if j != i:
                out[i] *= (t - xs[j]) / (xs[i] - xs[j])
    return out


class HangingMesh:
    def __init__(self, reps, p0, p1, refine, kv=2):
        """reps: coarse cells per direction; refine: set of coarse cell index tuples (ix, iy[, iz]) refined once"""
        dim = len(reps)
        self.dim, self.kv = dim, kv
        reps = tuple(reps)
        p0, p1 = np.array(p0, float), np.array(p1, float)
        h = (p1 - p0) / np.array(reps)
        refine = {tuple(r) for r in refine}
        # integer lattice: one coarse cell = 2*kv velocity units = 2 pressure units per direction
        cells = []  # (origin in half-cell units, size in half-cell units)
        for ci in itertools.product(*[range(r) for r in reps[::-1]]):
            ci = ci[::-1]
            if ci in refine:
                for ch in itertools.product(*[range(2)] * dim):
                    ch = ch[::-1]
                    cells.append((tuple(2 * ci[d] + ch[d] for d in range(dim)), 1))
            else:
                cells.append((tuple(2 * c for c in ci), 2))
        self.n_cells = len(cells)
        nv, nu = 2 ** dim, (kv + 1) ** dim
        loc_v = [tuple(reversed(t)) for t in itertools.product(*[range(2)] * dim)]       # x fastest
        loc_u = [tuple(reversed(t)) for t in itertools.product(*[range(kv + 1)] * dim)]
        uid, pid = {}, {}
        cell_unodes = np.zeros((self.n_cells, nu), np.int32)
        cell_pnodes = np.zeros((self.n_cells, nv), np.int32)
        vcoords = np.zeros((self.n_cells, nv, dim))
        bid = -np.ones((self.n_cells, 2 * dim), np.int32)
        for c, (org, size) in enumerate(cells):
            for a, l in enumerate(loc_u):  # velocity lattice: kv units per half cell
                key = tuple(org[d] * kv + l[d] * size for d in range(dim))
                cell_unodes[c, a] = uid.setdefault(key, len(uid))
            for a, l in enumerate(loc_v):
                key = tuple(org[d] + l[d] * size for d in range(dim))
                cell_pnodes[c, a] = pid.setdefault(key, len(pid))
                vcoords[c, a] = p0 + np.array(key) * h / 2
            for d in range(dim):
                if org[d] == 0:
                    bid[c, 2 * d] = 2 * d
                if org[d] + size == 2 * reps[d]:
                    bid[c, 2 * d + 1] = 2 * d + 1
        self.cell_unodes, self.cell_pnodes = cell_unodes, cell_pnodes
        self.vcoords, self.cell_face_bid = vcoords, bid
        self.n_unodes, self.n_pnodes = len(uid), len(pid)
        self.n_u = dim * self.n_unodes
        self.n_dofs = self.n_u + self.n_pnodes
        self.indicator = None
        ukeys = np.zeros((self.n_unodes, dim), int)
        for k, i in uid.items():
            ukeys[i] = k
        pkeys = np.zeros((self.n_pnodes, dim), int)
        for k, i in pid.items():
            pkeys[i] = k
        self.unode_lattice, self.pnode_lattice = ukeys, pkeys
        self.unode_coords = p0 + ukeys * h / (2 * kv)
        self.pnode_coords = p0 + pkeys * h / 2
        self._ext_u = np.array([2 * kv * r for r in reps])
        # hanging nodes: a node inside the closure of an UNREFINED coarse cell that is not one of that cell's nodes;
        # its line = the coarse cell's shape functions at the node
        lines = {}
        for c, (org, size) in enumerate(cells):
            if size != 2:
                continue
            for (keys, k, unit, table, cell_nodes, off, ncomp) in (
                    (ukeys, kv, kv, uid, cell_unodes[c], 0, dim), (pkeys, 1, 1, pid, cell_pnodes[c], self.n_u, 1)):
                lo = np.array(org) * unit
                hi = lo + 2 * unit
                inside = np.all((keys >= lo) & (keys <= hi), axis=1)
                mine = set(int(x) for x in cell_nodes)
                loc = loc_u if k == kv and ncomp == dim else loc_v
                for nd in np.nonzero(inside)[0]:
                    if int(nd) in mine:
                        continue
                    t = (keys[nd] - lo) / (2.0 * unit)
                    w1 = [_lagrange(k, t[d]) for d in range(dim)]
                    ms, ws = [], []
                    for a, l in enumerate(loc):
                        w = np.prod([w1[d][l[d]] for d in range(dim)])
                        if abs(w) > 1e-13:
                            ms.append(int(cell_nodes[a]))
                            ws.append(float(w))
                    for cpt in range(ncomp):
                        dof = off + (dim * int(nd) + cpt if ncomp == dim else int(nd))
                        lines.setdefault(dof, ([off + (dim * m + cpt if ncomp == dim else m) for m in ms], ws))
        dofs = sorted(lines)
        self.hang_dof = np.array(dofs, np.int32)
        ptr, master, weight = [0], [], []
        for d in dofs:
            master += lines[d][0]
            weight += lines[d][1]
            ptr.append(len(master))
        self.hang_ptr = np.array(ptr, np.int32)
        self.hang_master = np.array(master, np.int32)
        self.hang_weight = np.array(weight, float)
        assert not set(dofs) & set(master), "lines are closed on a one-irregular mesh"

    def boundary_unodes(self, bid):
        d, side = bid // 2, bid % 2
        target = 0 if side == 0 else self._ext_u[d]
        return np.nonzero(self.unode_lattice[:, d] == target)[0]

    def dirichlet(self, bcs, fields=None):
        """as BoxMesh.dirichlet; hanging dofs keep their hanging line (interpolate_boundary_values skips them)"""
        hanging = set(int(x) for x in self.hang_dof)
        dofs, vals, seen = [], [], set()
        for bid in sorted(bcs):
            flag, value = bcs[bid]
            comps = [c for c in range(self.dim) if flag & (1 << c)]
            for nd in self.boundary_unodes(bid):
                for k, c in enumerate(comps):
                    dof = self.dim * int(nd) + c
                    if dof in seen or dof in hanging:
                        continue
                    seen.add(dof)
                    dofs.append(dof)
                    vals.append(fields[bid](self.unode_coords[nd], c) if fields and bid in fields else value[k])
        return np.array(dofs, np.int32), np.array(vals, float)

    def prolongation(self):
        """dense C: x_full = C x with the hanging entries interpolated from ALL their masters"""
        Cm = np.eye(self.n_dofs)
        for i, d in enumerate(self.hang_dof):
            Cm[d, :] = 0
            for k in range(self.hang_ptr[i], self.hang_ptr[i + 1]):
                Cm[d, self.hang_master[k]] = self.hang_weight[k]
        return Cm
