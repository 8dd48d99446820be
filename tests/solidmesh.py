"""Q1 solid meshes as MPI::FSI sees them (test infrastructure): vertices at their current position, cells in deal.II's
lexicographic vertex order, the boundary faces of collect_solid_boundaries (mpi_fsi.cpp:78-94) and the localized nodal
fields velocity / acceleration / stress (mpi_fsi.cpp:350-362).  Stand-in for the solid solver's Triangulation + DoFHandler
(Solid degree 1 in every FSI test of the reference)."""
import itertools

import numpy as np


class SolidMesh:
    def __init__(self, dim, vertices, cells):
        self.dim = dim
        self.vertices = np.ascontiguousarray(vertices, float)
        self.cells = np.ascontiguousarray(cells, np.int32)
        self.bfaces = self._boundary_faces()
        n = len(self.vertices)
        self.velocity = np.zeros((n, dim))
        self.acceleration = np.zeros((n, dim))
        self.stress = np.zeros((dim * (dim + 1) // 2, n))

    def _boundary_faces(self):
        """faces that belong to one cell only; 2D: vertex pairs in the order of GeometryInfo<2>::face_to_cell_vertices
        (faces x-, x+, y-, y+ = (0,2), (1,3), (0,1), (2,3)); 3D: quadruples (unused by point_in_solid)"""
        if self.dim == 2:
            loc = [(0, 2), (1, 3), (0, 1), (2, 3)]
        else:
            loc = [(0, 2, 4, 6), (1, 3, 5, 7), (0, 1, 4, 5), (2, 3, 6, 7), (0, 1, 2, 3), (4, 5, 6, 7)]
        count, faces = {}, []
        for c in self.cells:
            for l in loc:
                f = tuple(int(c[i]) for i in l)
                faces.append(f)
                count[tuple(sorted(f))] = count.get(tuple(sorted(f)), 0) + 1
        return np.array([f for f in faces if count[tuple(sorted(f))] == 1], np.int32)

    def set_fields(self, vel, acc, stress):
        """callables of the vertex coordinates [n, dim] -> [n, dim], [n, dim], [ncomp, n]"""
        self.velocity = np.ascontiguousarray(vel(self.vertices), float)
        self.acceleration = np.ascontiguousarray(acc(self.vertices), float)
        self.stress = np.ascontiguousarray(stress(self.vertices), float)
        return self

    def moved(self, shift=None, rot=0.0, about=None):
        """a copy at another position: rigid rotation by `rot` (about the z axis in 3D) around `about`, then a shift"""
        v = self.vertices.copy()
        about = v.mean(axis=0) if about is None else np.asarray(about, float)
        c, s = np.cos(rot), np.sin(rot)
        r = v - about
        x, y = r[:, 0].copy(), r[:, 1].copy()
        r[:, 0], r[:, 1] = c * x - s * y, s * x + c * y
        v = about + r + (0 if shift is None else np.asarray(shift, float))
        out = SolidMesh(self.dim, v, self.cells)
        out.velocity, out.acceleration, out.stress = self.velocity.copy(), self.acceleration.copy(), self.stress.copy()
        return out


def lattice_solid(reps, p0, p1, mask=None, mapping=None):
    """cells of the lattice reps[0] x reps[1] (x reps[2]) on the box [p0, p1] that `mask(ix, iy[, iz])` keeps, vertices
    optionally mapped by `mapping(points [n, dim]) -> [n, dim]` (rotation, distortion); only the vertices in use are kept"""
    dim = len(reps)
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    nv1 = [r + 1 for r in reps]
    strides = np.cumprod([1] + nv1[:-1])
    vid, cells = {}, []
    loc = [tuple(reversed(t)) for t in itertools.product(*[range(2)] * dim)]  # x fastest
    for ci in itertools.product(*[range(r) for r in reps[::-1]]):
        ci = ci[::-1]
        if mask is not None and not mask(*ci):
            continue
        cells.append([vid.setdefault(int(sum((ci[d] + l[d]) * strides[d] for d in range(dim))), len(vid)) for l in loc])
    keys = np.array(sorted(vid, key=vid.get))
    idx = np.stack([(keys // strides[d]) % nv1[d] for d in range(dim)], axis=1)
    pts = p0 + idx * (p1 - p0) / np.array(reps)
    if mapping is not None:
        pts = mapping(pts)
    return SolidMesh(dim, pts, np.array(cells, np.int32))


def rotation(angle, about):
    about = np.asarray(about, float)
    c, s = np.cos(angle), np.sin(angle)

    def f(p):
        ab = np.concatenate([about, np.zeros(p.shape[1] - len(about))])
        r = p - ab
        out = r.copy()
        out[:, 0], out[:, 1] = c * r[:, 0] - s * r[:, 1], s * r[:, 0] + c * r[:, 1]
        return ab + out
    return f


def wobble(amp, freq):
    """smooth distortion that keeps cells valid for amp * freq well below 1: non-parallelogram quads / non-affine hexes"""
    def f(p):
        out = p.copy()
        d = p.shape[1]
        for k in range(d):
            out[:, k] += amp * np.sin(freq * p[:, (k + 1) % d] + 0.3 * k) * np.cos(0.7 * freq * p[:, (k + 2) % d] if d == 3 else 1.0)
        return out
    return f
