"""Pins of oracle/oracle_fsi.c (the CPU restatement of MPI::FSI's fluid-side inputs, source/mpi_fsi.cpp:96-127,142-223,
291-663).  The reference's tests hold no vector for these functions, so they are pinned by what must hold whatever the
implementation: the crossing-number test agrees with the analytic inside test of convex and non-convex polygons, the
unit-cell inversion undoes the forward map, Q1 interpolation reproduces affine fields, and find_fluid_bc on affine
solid / fluid fields gives the closed-form (v_s - v)/dt + grad(v) v - a_s."""
import numpy as np
import pytest

import orc
from boxmesh import BoxMesh
from hangmesh import HangingMesh
from solidmesh import lattice_solid, rotation, wobble

RNG = np.random.default_rng(7)


def _lshape_mask(ix, iy):
    return not (ix >= 3 and iy >= 2)  # 6 x 4 lattice with the upper right 3 x 2 block removed


def _rect(angle=0.35):
    rot = rotation(angle, (0.5, 0.4))
    return lattice_solid((6, 4), (0.2, 0.25), (0.8, 0.55), mapping=rot), rot


def _unrotate(p, angle, about):
    return rotation(-angle, about)(p)


@pytest.mark.parametrize("shape", ["rect", "lshape"])
def test_crossing_number_matches_analytic_inside(shape):
    angle, about = 0.35, (0.5, 0.4)
    mask = _lshape_mask if shape == "lshape" else None
    s = lattice_solid((6, 4), (0.2, 0.25), (0.8, 0.55), mask=mask, mapping=rotation(angle, about))
    assert len(s.bfaces) == (20 if shape == "rect" else 20)  # the L has the same perimeter count: 6+4+3+2+3+2
    S = orc.FsiSolid(s)
    pts = RNG.uniform((0.0, 0.0), (1.0, 0.8), (4000, 2))
    q = _unrotate(pts, angle, about)
    u = (q - (0.2, 0.25)) / ((0.8 - 0.2) / 6, (0.55 - 0.25) / 4)  # lattice coordinates
    inside = (u[:, 0] > 0) & (u[:, 0] < 6) & (u[:, 1] > 0) & (u[:, 1] < 4)
    if shape == "lshape":
        inside &= ~((u[:, 0] > 3) & (u[:, 1] > 2))
    edge = np.minimum(np.abs(u - np.round(u)).min(axis=1), 1.0) < 1e-9  # nobody sits on a lattice line
    assert not edge.any()
    got = S.point_in_solid(pts)
    assert inside.sum() > 300 and (~inside).sum() > 300
    assert (got == inside).all()


def test_crossing_number_special_branches():
    """the branches of mpi_fsi.cpp:175-208 for points level with a boundary vertex, on a horizontal face, on a vertex"""
    s = lattice_solid((2, 2), (0.0, 0.0), (1.0, 1.0))  # axis-aligned square, boundary vertices at y = 0, 0.5, 1
    S = orc.FsiSolid(s)
    pts = np.array([[0.25, 0.5],   # level with the vertices (0, 0.5) and (1, 0.5): two half crossings on the right -> inside
                    [0.25, 0.0],   # on the bottom face
                    [0.0, 0.0],    # on a corner vertex
                    [1.0, 0.25],   # on the right face (r2 + p2 == point)
                    [0.5, 1.0],    # on the top of the box AT a boundary vertex: the literal algorithm says outside (neither
                                   # horizontal face has x_diff1 * x_diff2 < 0 and half crossings are not counted at box(3))
                    [1.5, 0.5], [-0.1, 0.5], [0.5, 1.0001]])
    assert S.point_in_solid(pts).tolist() == [True, True, True, True, False, False, False, False]


@pytest.mark.parametrize("dim", [2, 3])
def test_real_to_unit_inverts_the_dlinear_map(dim):
    s = lattice_solid((3,) * dim, (0.0,) * dim, (1.0,) * dim, mapping=wobble(0.04, 5.0))
    L = orc.lib()
    for c in range(len(s.cells)):
        X = np.ascontiguousarray(s.vertices[s.cells[c]])
        xi = RNG.uniform(0, 1, dim)
        N = np.array([np.prod([xi[d] if (v >> d) & 1 else 1 - xi[d] for d in range(dim)]) for v in range(2 ** dim)])
        p = np.ascontiguousarray(N @ X)
        out = np.zeros(dim)
        assert L.orc_fsi_real_to_unit(dim, orc._ptr(X), orc._ptr(p), orc._ptr(out)) == 1
        assert np.abs(out - xi).max() < 1e-12


@pytest.mark.parametrize("dim", [2, 3])
def test_locate_and_interpolate_reproduce_affine_fields(dim):
    reps = (5, 4) if dim == 2 else (4, 3, 3)
    s = lattice_solid(reps, (0.1,) * dim, (0.9,) * dim, mapping=lambda p: wobble(0.015, 6.0)(rotation(0.4, (0.5, 0.5))(p)))
    S = orc.FsiSolid(s)
    G = RNG.normal(size=(dim, dim))
    f = lambda x: x @ G.T + 0.3  # noqa: E731
    pts = RNG.uniform(0.0, 1.0, (1500, dim))
    inside = S.point_in_solid(pts)
    cells, xi = S.locate(pts)
    assert inside.sum() > 200
    # the two independent decisions agree (2D: crossing number vs cell search; 3D: point_inside vs the 1e-10 search)
    assert ((cells >= 0) == inside).all()
    for i in np.nonzero(inside)[0][:400]:
        N = np.array([np.prod([xi[i, d] if (v >> d) & 1 else 1 - xi[i, d] for d in range(dim)]) for v in range(2 ** dim)])
        x = N @ s.vertices[s.cells[cells[i]]]
        assert np.abs(x - pts[i]).max() < 1e-12             # the located unit point maps back to the point
        assert np.abs(N @ f(s.vertices[s.cells[cells[i]]]) - f(pts[i:i + 1])[0]).max() < 1e-12


def test_update_indicator_is_all_vertices_inside():
    s, _ = _rect()
    m = BoxMesh((20, 16), (0, 0), (1.0, 0.8), kv=1)
    ind = orc.fsi_update_indicator(m, s)
    inside = orc.FsiSolid(s).point_in_solid(m.vcoords.reshape(-1, 2)).reshape(m.n_cells, 4)
    assert ind.sum() > 10 and (ind == inside.all(axis=1)).all()
    assert (inside.any(axis=1) & ~inside.all(axis=1)).sum() > 10  # cut cells exist and are real fluid


@pytest.mark.parametrize("kind", ["box2_q1", "box2_q2", "hang2_q1", "box3_q2"])
@pytest.mark.parametrize("use_dirichlet_bc", [False, True])
def test_find_fluid_bc_closed_form_on_affine_fields(kind, use_dirichlet_bc):
    dim = 3 if kind.startswith("box3") else 2
    kv = 2 if kind.endswith("q2") else 1
    if kind == "hang2_q1":
        m = HangingMesh((10, 8), (0, 0), (1.0, 0.8), {(i, j) for i in range(3, 7) for j in range(2, 6)}, kv=1)
    elif dim == 2:
        m = BoxMesh((20, 16), (0, 0), (1.0, 0.8), kv=kv)
    else:
        m = BoxMesh((8, 8, 6), (0, 0, 0), (1.0, 0.8, 0.6), kv=kv)
    if dim == 2:
        s, _ = _rect()
    else:
        s = lattice_solid((4, 3, 3), (0.21, 0.2, 0.13), (0.8, 0.61, 0.51), mapping=rotation(0.3, (0.47, 0.41)))  # no fluid node on a solid face
    Gv, Ga, Gf = RNG.normal(size=(dim, dim)), RNG.normal(size=(dim, dim)), RNG.normal(size=(dim, dim))
    ncomp = dim * (dim + 1) // 2
    Gs = RNG.normal(size=(ncomp, dim))
    s.set_fields(lambda x: x @ Gv.T + 0.2, lambda x: x @ Ga.T - 0.1, lambda x: Gs @ x.T + 0.5)
    dt = 0.01
    ind = orc.fsi_update_indicator(m, s)
    assert ind.sum() > 3
    present = np.zeros(m.n_dofs)
    present[:m.n_u] = (m.unode_coords @ Gf.T + 0.05).reshape(-1)  # affine: reproduced exactly by Q_k, gradient Gf
    present[m.n_u:] = RNG.normal(size=m.n_pnodes)
    fl = RNG.normal(size=(dim, dim, m.n_unodes))
    fsi_stress = np.full((ncomp, m.n_unodes), 7.0)
    acc, flag, val, nf = orc.fsi_find_fluid_bc(m, s, ind, dt, use_dirichlet_bc, present, fl, fsi_stress)
    assert nf == 0
    node_in = orc.FsiSolid(s).point_in_solid(m.unode_coords)
    in_ind = np.zeros(m.n_unodes, bool)
    in_ind[np.unique(m.cell_unodes[ind == 1])] = True
    sel = node_in & in_ind
    assert sel.sum() > 8
    x = m.unode_coords
    # nodal fsi_stress: assigned at the nodes of indicator cells inside the solid, untouched elsewhere (:469-471)
    k = 0
    for i in range(dim):
        for j in range(i + 1):
            want = np.where(sel, fl[i, j] - (Gs[k] @ x.T + 0.5), 7.0)
            assert np.abs(fsi_stress[k] - want).max() < 1e-11
            k += 1
    v = present[:m.n_u].reshape(-1, dim)
    if not use_dirichlet_bc:
        want = ((x @ Gv.T + 0.2) - v) / dt + v @ Gf.T - (x @ Ga.T - 0.1)
        want[~sel] = 0.0
        assert np.abs(acc[:m.n_u].reshape(-1, dim) - want).max() < 1e-9 * np.abs(want).max()
        assert not acc[m.n_u:].any() and not flag.any()
    else:
        interior = np.zeros(m.n_unodes, bool)
        if kv == 2:  # in-cell support points carry no line (:588-600)
            interior[m.cell_unodes[:, (3 ** dim) // 2]] = True
        lines = node_in & ~interior
        assert (flag.reshape(-1, dim) == lines[:, None]).all()
        want = (x @ Gv.T + 0.2) - v
        assert np.abs(val.reshape(-1, dim)[lines] - want[lines]).max() < 1e-12
        assert not acc.any()


@pytest.mark.parametrize("kind", ["box2_q2", "hang2_q1", "box3_q2"])
def test_fluid_point_values_reproduce_the_fe_space(kind):
    """the fluid solution at points of the solid (find_solid_bc, mpi_fsi.cpp:727-760): a field of the FE space itself
    (componentwise quadratic for Q2 velocity / stress, affine for Q1 and the pressure) is returned exactly at any point,
    points outside the mesh come back as not found with zero values"""
    dim = 3 if kind.startswith("box3") else 2
    kv = 2 if kind.endswith("q2") else 1
    if kind == "hang2_q1":
        m = HangingMesh((10, 8), (0, 0), (1.0, 0.8), {(i, j) for i in range(3, 7) for j in range(2, 6)}, kv=1)
    elif dim == 2:
        m = BoxMesh((7, 5), (0, 0), (1.0, 0.8), kv=kv)
        m.vcoords = wobble(0.01, 5.0)(m.vcoords.reshape(-1, 2)).reshape(m.vcoords.shape)  # non-affine cells ...
    else:
        m = BoxMesh((4, 3, 3), (0, 0, 0), (1.0, 0.8, 0.6), kv=kv)
    A, b = RNG.normal(size=(dim + 1, dim)), RNG.normal(size=dim + 1)
    if kind == "box2_q2":  # ... on which only isoparametric images are exact: use the unit-cell picture of the nodes
        x_u = np.zeros((m.n_unodes, 2))
        loc = np.stack(np.unravel_index(np.arange(9), (3, 3)), axis=-1)[:, ::-1] / 2.0
        for c in range(m.n_cells):
            N = np.stack([(1 - loc[:, 0]) * (1 - loc[:, 1]), loc[:, 0] * (1 - loc[:, 1]), (1 - loc[:, 0]) * loc[:, 1], loc[:, 0] * loc[:, 1]], axis=1)
            x_u[m.cell_unodes[c]] = N @ m.vcoords[c]
        x_p = np.zeros((m.n_pnodes, 2))
        x_p[m.cell_pnodes.ravel()] = m.vcoords.reshape(-1, 2)
    else:
        x_u, x_p = m.unode_coords, m.pnode_coords
    f = lambda x: x @ A.T + b  # noqa: E731  affine: in every space, exact under the d-linear map
    present = np.concatenate([f(x_u)[:, :dim].ravel(), f(x_p)[:, dim]])
    G = RNG.normal(size=(dim * dim, dim))
    stress = (G @ x_u.T + 0.3).reshape(dim, dim, m.n_unodes)
    pts = RNG.uniform(-0.05, 1.05, (300, dim)) * np.array([1.0, 0.8, 0.6][:dim])
    vals, st, cell = orc.fsi_fluid_at_points(m, present, stress, pts)
    hi = np.array([1.0, 0.8, 0.6][:dim])
    if kind == "box2_q2":
        inside = cell >= 0
        assert inside.sum() > 200
    else:
        inside = ((pts >= 0) & (pts <= hi)).all(axis=1)
        assert ((cell >= 0) == inside).all() and inside.sum() > 150 and (~inside).sum() > 10
    assert np.abs(vals[inside] - f(pts[inside])).max() < 1e-11
    assert np.abs(st[inside].reshape(-1, dim * dim) - (pts[inside] @ G.T + 0.3)).max() < 1e-11
    assert not vals[~inside].any() and not st[~inside].any()


def test_bench_solid_and_its_analytic_inside_test_agree_with_the_oracle():
    """tools/fsibench.py: the 24x12x12 rotated block of the bench leg / of the full-size GPU test, its closed-form inside test
    (what tests/test_gpu_fullsize.py checks the device against at 128^3) and the oracle's point_in_solid say the same"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fsibench
    s = fsibench.make_solid3d()
    X = s["vertices"][s["cells"]]
    det = np.einsum("ij,ij->i", np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), X[:, 4] - X[:, 0])
    assert len(s["cells"]) == 3456 and det.min() > 0  # right-handed cells in deal.II vertex order
    pts = RNG.uniform((0.5, 0.0, 0.0), (1.2, 0.2, 0.2), (4000, 3))
    inside, gap = fsibench.inside_solid3d(pts)
    got = orc.FsiSolid(fsibench._Solid(s)).point_in_solid(pts)
    assert 300 < inside.sum() < 3700 and gap.min() > 1e-9 and (got == inside).all()
    _, gap_v = fsibench.inside_solid3d(s["vertices"])
    assert (gap_v < 1e-12).sum() == 25 * 13 * 13 - 23 * 11 * 11  # the boundary vertices sit on the faces, the others do not
