"""Independent numpy restatement of Utils::GridCreator<2>::flow_around_cylinder + refine_global(L)
(reference source/utilities.cpp:345-524), test infrastructure.

Coarse mesh: 22 x 4 cells on [0,2.2] x [0,0.41] with the four cells around (0.2,0.2) removed and replaced by
hyper_cube_with_cylindrical_hole (8 cells, inner radius 0.05) shifted by (0.2, 0.205); merge_triangulations keeps
the bulk's vertex coordinates on the seam; the circle's vertices are re-centred to (0.2, 0.2).  Refinement places
new vertices with the PolarManifold on the cylinder and the TransfiniteInterpolationManifold in the 8 ring cells,
which for a cell with one curved (inner) edge reduces to x(xi, eta) = (1 - xi) Arc(eta) + xi Out(eta) evaluated at
the dyadic points; bulk cells refine uniformly.  Boundary ids: 0 inflow x=0, 1 outflow x=2.2, 2 y=0, 3 y=0.41,
4 cylinder.  Q2 nodes: vertices, edge midpoints, cell centres (cell integrals use the Q1 vertex mapping only).
"""
import numpy as np


class CylinderMesh:
    dim = 2

    def __init__(self, refinements=3, kv=2, for_3d=False):
        # for_3d: the variant GridCreator<3> extrudes (utilities.cpp:348-355): channel from x = -0.3, 25 bulk columns
        self.kv = kv
        s = 2 ** refinements
        left, ncol, hole = (-0.3, 25, (4, 5)) if for_3d else (0.0, 22, (1, 2))
        self.left = left
        hx, hy = 2.2 / 22, 0.41 / 4
        c = np.array([0.2, 0.2])
        r_in = 0.05
        sq_c = np.array([0.2, 0.205])
        O = [sq_c + np.array(v) for v in ((0.1, 0), (0.1, 0.1025), (0, 0.1025), (-0.1, 0.1025), (-0.1, 0), (-0.1, -0.1025),
                                          (0, -0.1025), (0.1, -0.1025))]
        t = np.arange(s + 1) / s
        XI, ETA = np.meshgrid(t, t, indexing="ij")  # [a, b] -> xi = a/s, eta = b/s
        patches = []  # each: array [s+1, s+1, 2] of vertex coordinates in the patch's (xi, eta) frame
        for j in range(4):
            for i in range(ncol):
                if i in hole and j in (1, 2):
                    continue
                X = np.stack([left + (i + XI) * hx, (j + ETA) * hy], -1)
                patches.append(X)
        for k in range(8):
            th = 2 * np.pi * k / 8 + ETA * (np.pi / 4)
            arc = c + r_in * np.stack([np.cos(th), np.sin(th)], -1)
            out = (1 - ETA)[..., None] * O[k] + ETA[..., None] * O[(k + 1) % 8]
            # frame: xi radial (inner -> outer), eta angular (counter-clockwise): positive Jacobian
            patches.append((1 - XI)[..., None] * arc + XI[..., None] * out)
        # global vertex numbering by rounded coordinates
        vid, vcoord = {}, []

        def vertex(p):
            key = (round(p[0] * 1e9), round(p[1] * 1e9))
            if key not in vid:
                vid[key] = len(vcoord)
                vcoord.append(p)
            return vid[key]

        cells = []
        for X in patches:
            ids = np.array([[vertex(X[a, b]) for b in range(s + 1)] for a in range(s + 1)])
            for b in range(s):
                for a in range(s):
                    cells.append([ids[a, b], ids[a + 1, b], ids[a, b + 1], ids[a + 1, b + 1]])
        cells = np.array(cells, np.int64)
        vcoord = np.array(vcoord)
        self.n_cells = len(cells)
        self.cell_vertices = cells
        self.vertex_coords = vcoord
        self.vcoords = np.ascontiguousarray(vcoord[cells])  # [n_cells, 4, 2]
        self.cell_pnodes = cells.astype(np.int32)
        self.n_pnodes = len(vcoord)
        self.pnode_coords = vcoord
        # edges
        local_edges = {(1, 0): (0, 1), (0, 1): (0, 2), (2, 1): (1, 3), (1, 2): (2, 3)}  # (i,j) lattice -> local vertices
        eid, ecoord, ecount = {}, [], {}
        cu = np.zeros((self.n_cells, 9), np.int64)
        nV = len(vcoord)
        for ci, cv in enumerate(cells):
            for (i, j), (a, b) in local_edges.items():
                key = (min(cv[a], cv[b]), max(cv[a], cv[b]))
                if key not in eid:
                    eid[key] = len(ecoord)
                    ecoord.append(0.5 * (vcoord[cv[a]] + vcoord[cv[b]]))
                    ecount[key] = 0
                ecount[key] += 1
                cu[ci, i + 3 * j] = nV + eid[key]
            cu[ci, 0], cu[ci, 2], cu[ci, 6], cu[ci, 8] = cv[0], cv[1], cv[2], cv[3]
        nE = len(ecoord)
        cu[:, 4] = nV + nE + np.arange(self.n_cells)
        centers = self.vcoords.mean(1)
        self.unode_coords = np.concatenate([vcoord, np.array(ecoord), centers])
        self.n_unodes = len(self.unode_coords)
        if kv == 2:
            self.cell_unodes = cu.astype(np.int32)
        else:
            self.cell_unodes = cells.astype(np.int32)
            self.unode_coords = vcoord
            self.n_unodes = nV
        # boundary ids by face centre (utilities.cpp:493-523); faces x-, x+, y-, y+ = (v0,v2), (v1,v3), (v0,v1), (v2,v3)
        faces = ((0, 2), (1, 3), (0, 1), (2, 3))
        bid = -np.ones((self.n_cells, 4), np.int32)
        for ci, cv in enumerate(cells):
            for f, (a, b) in enumerate(faces):
                key = (min(cv[a], cv[b]), max(cv[a], cv[b]))
                if ecount[key] != 1:
                    continue
                m = 0.5 * (vcoord[cv[a]] + vcoord[cv[b]])
                if abs(m[0] - 2.2) < 1e-12:
                    bid[ci, f] = 1
                elif abs(m[0] - left) < 1e-12:
                    bid[ci, f] = 0
                elif abs(m[1] - 0.41) < 1e-12:
                    bid[ci, f] = 3
                elif abs(m[1]) < 1e-12:
                    bid[ci, f] = 2
                else:
                    bid[ci, f] = 4
        self.cell_face_bid = bid
        self.n_u = 2 * self.n_unodes
        self.n_dofs = self.n_u + self.n_pnodes
        self.indicator = None

    def dirichlet(self, bcs, fields=None):
        """Same contract as BoxMesh.dirichlet: ids ascending, the first line of a dof wins."""
        n1 = self.kv + 1
        dofs, vals, seen = [], [], set()
        for bid in sorted(bcs):
            flag, value = bcs[bid]
            comps = [c for c in range(2) if flag & (1 << c)]
            for ci, f in zip(*np.nonzero(self.cell_face_bid == bid)):
                nd_, side = f // 2, (f % 2) * self.kv
                for a in range(n1 * n1):
                    idx = (a % n1, a // n1)
                    if idx[nd_] != side:
                        continue
                    node = self.cell_unodes[ci, a]
                    for k, c in enumerate(comps):
                        dof = 2 * node + c
                        if dof in seen:
                            continue
                        seen.add(dof)
                        dofs.append(dof)
                        vals.append(fields[bid](self.unode_coords[node], c) if fields and bid in fields else value[k])
        return np.array(dofs, np.int32), np.array(vals, float)


def inflow_bc(p, component):
    # tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:32-52 (2D): parabolic profile, Umax = 0.3
    if component == 0 and abs(p[0]) < 1e-10:
        return 4 * 0.3 * p[1] * (0.41 - p[1]) / (0.41 * 0.41)
    return 0.0


class CylinderMesh3D:
    """Utils::GridCreator<3>::flow_around_cylinder (utilities.cpp:526-570): the x in [-0.3, 2.2] variant of the 2D mesh
    extruded to z in [0, 0.41] in 8 * 2^refinements layers; boundary ids 0 / 1 (x), 2 / 3 (y), 4 / 5 (z), 6 the cylinder.
    Q2 nodes are numbered by the set of vertices they span (vertex, edge, face, cell), independently of the host mirror."""
    dim = 3

    def __init__(self, refinements=0, kv=2):
        self.kv = kv
        m2 = CylinderMesh(refinements, kv=1, for_3d=True)
        layers = 8 * 2 ** refinements
        z = np.linspace(0.0, 0.41, layers + 1)
        nv2 = len(m2.vertex_coords)
        self.vertex_coords = np.concatenate([np.column_stack([m2.vertex_coords, np.full(nv2, zk)]) for zk in z])
        cells, bid = [], []
        for k in range(layers):
            for ci, cv in enumerate(m2.cell_vertices):
                cells.append(list(cv + k * nv2) + list(cv + (k + 1) * nv2))
                fb = [6 if b == 4 else b for b in m2.cell_face_bid[ci]]
                bid.append(fb + [4 if k == 0 else -1, 5 if k == layers - 1 else -1])
        cells = np.array(cells, np.int64)
        self.n_cells = len(cells)
        self.cell_vertices = cells
        self.vcoords = np.ascontiguousarray(self.vertex_coords[cells])  # [n_cells, 8, 3]
        self.cell_pnodes = cells.astype(np.int32)
        self.n_pnodes = len(self.vertex_coords)
        self.pnode_coords = self.vertex_coords
        self.cell_face_bid = np.array(bid, np.int32)
        if kv == 1:
            self.cell_unodes, self.unode_coords = self.cell_pnodes, self.vertex_coords
        else:
            ent, coords = {}, [p for p in self.vertex_coords]
            cu = np.zeros((self.n_cells, 27), np.int64)
            for ci, cv in enumerate(cells):
                for a in range(27):
                    idx = (a % 3, (a // 3) % 3, a // 9)
                    span = [cv[v] for v in range(8)
                            if all(idx[d] == 1 or ((v >> d) & 1) == idx[d] // 2 for d in range(3))]
                    if len(span) == 1:
                        cu[ci, a] = span[0]
                        continue
                    key = frozenset(span)
                    if key not in ent:
                        ent[key] = len(coords)
                        coords.append(self.vertex_coords[span].mean(0))
                    cu[ci, a] = ent[key]
            self.cell_unodes = cu.astype(np.int32)
            self.unode_coords = np.array(coords)
        self.n_unodes = len(self.unode_coords)
        self.n_u = 3 * self.n_unodes
        self.n_dofs = self.n_u + self.n_pnodes
        self.indicator = None

    def dirichlet(self, bcs, fields=None):
        n1 = self.kv + 1
        dofs, vals, seen = [], [], set()
        for bid in sorted(bcs):
            flag, value = bcs[bid]
            comps = [c for c in range(3) if flag & (1 << c)]
            for ci, f in zip(*np.nonzero(self.cell_face_bid == bid)):
                nd_, side = f // 2, (f % 2) * self.kv
                for a in range(n1 ** 3):
                    idx = (a % n1, (a // n1) % n1, a // (n1 * n1))
                    if idx[nd_] != side:
                        continue
                    node = self.cell_unodes[ci, a]
                    for k, c in enumerate(comps):
                        dof = 3 * node + c
                        if dof in seen:
                            continue
                        seen.add(dof)
                        dofs.append(dof)
                        vals.append(fields[bid](self.unode_coords[node], c) if fields and bid in fields else value[k])
        return np.array(dofs, np.int32), np.array(vals, float)


def inflow_bc_3d(p, component):
    # parabolic in y and z at the inlet x = -0.3 (the shape of fluid_cylinder_mpi.cpp:56-75 with Umax = 9/4 * 0.2)
    if component == 0 and abs(p[0] + 0.3) < 1e-10:
        return 0.45 * (4 * p[1] * (0.41 - p[1]) / 0.41 ** 2) * (4 * p[2] * (0.41 - p[2]) / 0.41 ** 2)
    return 0.0
