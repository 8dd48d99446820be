"""Independent numpy builder of box-mesh / DoF tables for the oracle (test infrastructure).

Mirrors GridGenerator::subdivided_hyper_rectangle(tria, reps, p0, p1, colorize=true) as used by the
reference tests (tests/fluid_pressure_driven/fluid_pressure_driven.cpp:33-34): boundary ids
0/1 = x-/x+, 2/3 = y-/y+, 4/5 = z-/z+.  Node numbering here is plain lexicographic over the
(kv*n+1)^dim lattice; it is deliberately NOT shared with the product host layer so that the two can be
cross-checked (tests/test_host_layer.py).
"""
import numpy as np


class BoxMesh:
    def __init__(self, reps, p0, p1, kv=2):
        reps = list(reps)
        dim = len(reps)
        self.dim, self.kv, self.reps = dim, kv, reps
        self.p0, self.p1 = np.array(p0, float), np.array(p1, float)
        n_cells = int(np.prod(reps))
        self.n_cells = n_cells
        nu1 = [kv * r + 1 for r in reps]
        np1 = [r + 1 for r in reps]
        self.nu1, self.np1 = nu1, np1
        self.n_unodes = int(np.prod(nu1))
        self.n_pnodes = int(np.prod(np1))
        nv = 2 ** dim
        nu = (kv + 1) ** dim
        # cell lattice indices (x fastest)
        ci = np.stack(np.unravel_index(np.arange(n_cells), reps[::-1]), axis=-1)[:, ::-1]  # [n_cells, dim] (ix,iy,iz)
        h = (self.p1 - self.p0) / np.array(reps)
        # vertices, lexicographic local order
        vloc = np.stack(np.unravel_index(np.arange(nv), (2,) * dim), axis=-1)[:, ::-1]  # [nv, dim]
        vidx = ci[:, None, :] + vloc[None, :, :]
        self.vcoords = np.ascontiguousarray(self.p0 + vidx * h)  # [n_cells, nv, dim]
        strides_p = np.cumprod([1] + np1[:-1])
        self.cell_pnodes = np.ascontiguousarray((vidx * strides_p).sum(-1).astype(np.int32))
        uloc = np.stack(np.unravel_index(np.arange(nu), (kv + 1,) * dim), axis=-1)[:, ::-1]
        uidx = kv * ci[:, None, :] + uloc[None, :, :]
        strides_u = np.cumprod([1] + nu1[:-1])
        self.cell_unodes = np.ascontiguousarray((uidx * strides_u).sum(-1).astype(np.int32))
        bid = -np.ones((n_cells, 2 * dim), np.int32)
        for d in range(dim):
            bid[ci[:, d] == 0, 2 * d] = 2 * d
            bid[ci[:, d] == reps[d] - 1, 2 * d + 1] = 2 * d + 1
        self.cell_face_bid = np.ascontiguousarray(bid)
        # node coordinates
        ui = np.stack(np.unravel_index(np.arange(self.n_unodes), nu1[::-1]), axis=-1)[:, ::-1]
        self.unode_lattice = ui
        self.unode_coords = self.p0 + ui * (h / kv)
        pi = np.stack(np.unravel_index(np.arange(self.n_pnodes), np1[::-1]), axis=-1)[:, ::-1]
        self.pnode_lattice = pi
        self.pnode_coords = self.p0 + pi * h
        self.n_u = dim * self.n_unodes
        self.n_dofs = self.n_u + self.n_pnodes
        self.indicator = None

    def boundary_unodes(self, bid):
        d, side = bid // 2, bid % 2
        target = 0 if side == 0 else self.nu1[d] - 1
        return np.nonzero(self.unode_lattice[:, d] == target)[0]

    def dirichlet(self, bcs, fields=None):
        """bcs: {boundary id: (component flag 1..7, [values...])} as in the .prm
        (mpi_fluid_solver.cpp:185-243); fields: {id: f(point, component) -> value} hard-coded BC.
        Returns (dofs, values): later ids do not override earlier ones (AffineConstraints keeps the
        first line, interpolate_boundary_values skips already-constrained dofs)."""
        dofs, vals, seen = [], [], set()
        for bid in sorted(bcs):
            flag, value = bcs[bid]
            comps = [c for c in range(self.dim) if flag & (1 << c)]
            for nd in self.boundary_unodes(bid):
                for k, c in enumerate(comps):
                    dof = self.dim * nd + c
                    if dof in seen:
                        continue
                    seen.add(dof)
                    dofs.append(dof)
                    if fields and bid in fields:
                        vals.append(fields[bid](self.unode_coords[nd], c))
                    else:
                        vals.append(value[k])
        return np.array(dofs, np.int32), np.array(vals, float)


def block_partition(reps, n_parts):
    """cell -> subdomain for a box mesh (cells numbered x fastest): n_parts lattice blocks, the factorisation of n_parts
    chosen so that the blocks are as cubic as possible (what p4est's Morton partition of the reference amounts to on a
    box).  Returns (cell_part [n_cells] int32, number of subdomains actually used)."""
    reps = list(reps)
    dim = len(reps)
    best, best_cost = None, None
    def factorisations(n, k):
        if k == 1:
            yield (n,)
            return
        for f in range(1, n + 1):
            if n % f == 0:
                for rest in factorisations(n // f, k - 1):
                    yield (f,) + rest
    for P in factorisations(n_parts, dim):
        if any(P[d] > reps[d] for d in range(dim)):
            continue
        side = [reps[d] / P[d] for d in range(dim)]
        cost = sum(np.prod(side) / side[d] for d in range(dim))  # surface of a block
        if best_cost is None or cost < best_cost:
            best, best_cost = P, cost
    if best is None:
        best = tuple(min(reps[d], 1) for d in range(dim))
    ci = np.stack(np.unravel_index(np.arange(int(np.prod(reps))), reps[::-1]), axis=-1)[:, ::-1]  # (ix, iy, iz)
    part = np.zeros(len(ci), np.int64)
    stride = 1
    for d in range(dim):
        b = (ci[:, d] * best[d]) // reps[d]
        part += b * stride
        stride *= best[d]
    return part.astype(np.int32), int(np.prod(best))
