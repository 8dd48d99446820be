"""The fluid-side inputs of MPI::FSI produced on the device (ifem_fsi_set_solid / _update_indicator /
_find_fluid_bc, csrc/fsi.hip) against the CPU restatement of source/mpi_fsi.cpp:96-127,142-223,291-663 (oracle/oracle_fsi.c).

Decisions (indicator, which dofs receive a line, which nodes are inside) must agree exactly; values to 1e-12 of their
scale -- both sides run the same Newton inversion of the d-linear map, the device with fused multiply-adds.  Fields are
random nodal values (nothing an affine shortcut would reproduce), solids are rotated and distorted."""
import numpy as np
import pytest

import orc
from boxmesh import BoxMesh
from hangmesh import HangingMesh
from partmesh import partition_mesh, run_virtual_ranks
from solidmesh import lattice_solid, rotation, wobble

pytestmark = pytest.mark.gpu


def _case(kind):
    rng = np.random.default_rng(11)
    if kind == "box2_q1":
        m = BoxMesh((40, 32), (0, 0), (1.0, 0.8), kv=1)
    elif kind == "box2_q2":
        m = BoxMesh((24, 20), (0, 0), (1.0, 0.8), kv=2)
    elif kind == "hang2_q1":
        m = HangingMesh((20, 16), (0, 0), (1.0, 0.8), {(i, j) for i in range(5, 13) for j in range(4, 11)}, kv=1)
    elif kind == "box3_q1":
        m = BoxMesh((12, 10, 8), (0, 0, 0), (1.0, 0.8, 0.6), kv=1)
    else:
        m = BoxMesh((9, 8, 6), (0, 0, 0), (1.0, 0.8, 0.6), kv=2)
    dim = m.dim
    if dim == 2:  # 150 cells, 50 boundary faces: more than one LDS tile of cells
        s = lattice_solid((15, 10), (0.22, 0.21), (0.81, 0.58), mapping=lambda p: wobble(0.006, 9.0)(rotation(0.35, (0.5, 0.4))(p)))
    else:  # 90 cells: two tiles
        s = lattice_solid((6, 5, 3), (0.21, 0.2, 0.13), (0.8, 0.61, 0.51),
                          mapping=lambda p: wobble(0.01, 6.0)(rotation(0.3, (0.47, 0.41))(p)))
    n = len(s.vertices)
    ncomp = dim * (dim + 1) // 2
    s.velocity, s.acceleration, s.stress = rng.normal(size=(n, dim)), rng.normal(size=(n, dim)), rng.normal(size=(ncomp, n))
    present = rng.normal(size=m.n_dofs)
    return m, s, present, rng


def _boundary_lines(m, rng):
    """some boundary lines that exist before find_fluid_bc (make_constraints): the x- face, all components"""
    nodes = np.nonzero(np.isclose(m.unode_coords[:, 0], 0.0))[0]
    dofs = (m.dim * nodes[:, None] + np.arange(m.dim)[None, :]).ravel().astype(np.int32)
    return dofs, rng.normal(size=len(dofs))


def _oracle(m, s, present, fluid_stress, fsi_stress0, dt, use_dirichlet_bc):
    ind = orc.fsi_update_indicator(m, s)
    fs = fsi_stress0.copy()
    acc, flag, val, nf = orc.fsi_find_fluid_bc(m, s, ind, dt, use_dirichlet_bc, present, fluid_stress, fs)
    assert nf == 0
    return ind, fs, acc, flag, val


@pytest.mark.parametrize("kind", ["box2_q1", "box2_q2", "hang2_q1", "box3_q1", "box3_q2"])
@pytest.mark.parametrize("use_dirichlet_bc", [False, True])
def test_device_fsi_inputs_match_oracle(kind, use_dirichlet_bc):
    from openifem_amd import capi
    m, s, present, rng = _case(kind)
    dim, dt = m.dim, 0.013
    ncomp = dim * (dim + 1) // 2
    ctx = capi.Context(dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.vec_set(capi.VEC_PRESENT, present)
    mu = 0.7
    fluid_stress = ctx.update_stress(mu)  # the projected stress the previous step left behind
    assert np.abs(fluid_stress - orc.System(m).update_stress(mu, present)).max() < 1e-10 * np.abs(fluid_stress).max()
    fsi_stress0 = rng.normal(size=(ncomp, m.n_unodes))
    ctx.set_scns_fields(fsi_stress=fsi_stress0)
    bdofs, bvals = _boundary_lines(m, rng)
    hang = getattr(m, "hang_dof", np.zeros(0, np.int32))
    if len(hang):
        ctx.set_hanging_constraints(m.hang_dof, m.hang_ptr, m.hang_master, m.hang_weight)
        keep = ~np.isin(bdofs, hang)
        bdofs, bvals = bdofs[keep], bvals[keep]
    ctx.set_constraints(1, bdofs, bvals)
    ctx.set_constraints(0, bdofs, None)

    ind_o, fs_o, acc_o, flag_o, val_o = _oracle(m, s, present, fluid_stress, fsi_stress0, dt, use_dirichlet_bc)
    assert ind_o.sum() >= 4 and (ind_o == 0).sum() > 0

    ctx.fsi_set_solid(s.vertices, s.cells, None if dim == 3 else s.bfaces, s.velocity, s.acceleration, s.stress)
    ind, n_art = ctx.fsi_update_indicator(m.n_cells)
    assert (ind == ind_o).all() and n_art == ind_o.sum()
    st = ctx.fsi_find_fluid_bc(dt, use_dirichlet_bc)
    assert st.n_not_found == 0 and st.n_inside > 0 and st.n_candidates >= st.n_inside

    fs = ctx.fsi_get_stress()
    changed = fs_o != fsi_stress0
    assert changed.any() and ((fs != fsi_stress0) == changed).all()       # the same entries were assigned
    assert np.abs(fs - fs_o).max() < 1e-12 * np.abs(fs_o).max()
    acc = ctx.vec_get(capi.VEC_FSI_ACC)
    f0, v0 = ctx.get_constraints(0)
    f1, v1 = ctx.get_constraints(1)
    if not use_dirichlet_bc:
        assert (acc != 0).sum() == (acc_o != 0).sum() > 0 and ((acc != 0) == (acc_o != 0)).all()
        assert np.abs(acc - acc_o).max() < 1e-12 * np.abs(acc_o).max()
        want = np.zeros(m.n_dofs, np.uint8)
        want[bdofs] = 1
        assert (f0 == want).all() and (f1 == want).all() and st.n_lines == 0  # the constraint objects are untouched
    else:
        assert not acc.any()
        taken = np.zeros(m.n_dofs, bool)
        taken[bdofs] = True
        taken[hang] = True
        new = (flag_o == 1) & ~taken[:m.n_u]  # left_object_wins (:641-651)
        assert new.sum() > 0 and st.n_lines == new.sum()
        if len(hang):
            assert ((flag_o == 1) & taken[:m.n_u]).sum() > 0, "no hanging / boundary dof inside the solid: the merge rule is not exercised"
        want_f = np.zeros(m.n_dofs, np.uint8)
        want_f[bdofs] = 1
        want_f[:m.n_u][new] = 1
        assert (f0 == want_f).all() and (f1 == want_f).all()
        want_v = np.zeros(m.n_dofs)
        want_v[bdofs] = bvals
        want_v[:m.n_u][new] = val_o[new]
        assert np.abs(v1 - want_v).max() < 1e-12 * np.abs(want_v).max()
        want_v[bdofs] = 0.0
        want_v[:m.n_u][new] = 0.0
        assert not (v0 - want_v).any()
    ctx.close()


def test_solid_outside_the_fluid_and_missing_solid():
    from openifem_amd import capi
    m, s, present, rng = _case("box2_q1")
    ctx = capi.Context(2, 1, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    with pytest.raises(capi.IfemError):
        ctx.fsi_update_indicator(m.n_cells)  # no solid yet
    far = s.moved(shift=(5.0, 0.0))
    ctx.fsi_set_solid(far.vertices, far.cells, far.bfaces, far.velocity, far.acceleration, far.stress)
    ind, n_art = ctx.fsi_update_indicator(m.n_cells)
    assert n_art == 0 and not ind.any()
    ctx.vec_set(capi.VEC_PRESENT, present)
    st = ctx.fsi_find_fluid_bc(0.01, True)
    assert st.n_candidates == 0 and st.n_lines == 0
    assert not ctx.vec_get(capi.VEC_FSI_ACC).any()
    ctx.close()


@pytest.mark.parametrize("kind", ["hang2_q1", "box3_q2"])
@pytest.mark.parametrize("use_dirichlet_bc", [False, True])
def test_device_fsi_inputs_on_four_virtual_ranks(kind, use_dirichlet_bc):
    """partitioned contexts: the first-touch cell is chosen by the global cell index, so every rank reproduces the
    single-rank values on the dofs it owns, and ghosts carry the owners' values after the call"""
    from openifem_amd import capi
    m, s, present, rng = _case(kind)
    dim, dt, mu = m.dim, 0.013, 0.7
    ncomp = dim * (dim + 1) // 2
    fluid_stress = orc.System(m).update_stress(mu, present)
    fsi_stress0 = rng.normal(size=(ncomp, m.n_unodes))
    ind_o, fs_o, acc_o, flag_o, val_o = _oracle(m, s, present, fluid_stress, fsi_stress0, dt, use_dirichlet_bc)
    c = m.vcoords.mean(axis=1)
    cell_rank = (c[:, 0] > 0.52).astype(int) + 2 * (c[:, 1] > 0.41).astype(int)  # the cut goes through the solid
    parts = partition_mesh(m, cell_rank, 4)
    hang = getattr(m, "hang_dof", np.zeros(0, np.int32))

    def work(rank, P, ctx):
        if len(P.hang_dof):
            ctx.set_hanging_constraints(P.hang_dof, P.hang_ptr, P.hang_master, P.hang_weight)
        ctx.vec_set(capi.VEC_PRESENT, present[P.ext_gdof])
        st_dev = ctx.update_stress(mu)
        assert np.abs(st_dev - fluid_stress[:, :, P.l2g_u]).max() < 1e-10 * np.abs(fluid_stress).max()
        ctx.set_scns_fields(fsi_stress=fsi_stress0[:, P.l2g_u])
        ctx.fsi_set_solid(s.vertices, s.cells, None if dim == 3 else s.bfaces, s.velocity, s.acceleration, s.stress)
        ind, _ = ctx.fsi_update_indicator(len(P.cells))
        assert (ind == ind_o[P.cells]).all()
        st = ctx.fsi_find_fluid_bc(dt, use_dirichlet_bc, cell_order=P.cells)
        assert st.n_not_found == 0
        fs = ctx.fsi_get_stress()
        # owned AND ghost entries: the fluid stress the difference is taken from is nodal, the solid part depends on the point only
        assert np.abs(fs - fs_o[:, P.l2g_u]).max() < 1e-12 * np.abs(fs_o).max()
        acc = ctx.vec_get(capi.VEC_FSI_ACC)
        assert np.abs(acc - acc_o[P.ext_gdof]).max() <= 1e-12 * max(np.abs(acc_o).max(), 1e-300)
        if use_dirichlet_bc:
            f1, v1 = ctx.get_constraints(1)
            f0, v0 = ctx.get_constraints(0)
            taken = np.zeros(m.n_dofs, bool)
            taken[hang] = True
            want = np.zeros(m.n_dofs, bool)
            want[:m.n_u] = (flag_o == 1) & ~taken[:m.n_u]
            wv = np.zeros(m.n_dofs)
            wv[:m.n_u] = np.where(want[:m.n_u], val_o, 0.0)
            assert (f1.astype(bool) == want[P.ext_gdof]).all() and (f0 == f1).all()
            assert np.abs(v1 - wv[P.ext_gdof]).max() < 1e-12 * np.abs(wv).max() and not v0.any()
        return int(st.n_inside)

    res = run_virtual_ranks(capi, parts, work)
    assert sum(res) > 0


@pytest.mark.parametrize("kind", ["box2_q1", "box2_q2", "hang2_q1", "box3_q1", "box3_q2"])
def test_fluid_values_at_solid_points_match_oracle(kind):
    """ifem_fsi_fluid_at_points: (u, p) and the projected viscous stress at the solid's vertices plus points outside the
    fluid mesh (find_solid_bc, mpi_fsi.cpp:727-760; update_solid_displacement, :268-271) -- same cell, values to 1e-12"""
    from openifem_amd import capi
    m, s, present, rng = _case(kind)
    dim = m.dim
    ctx = capi.Context(dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.vec_set(capi.VEC_PRESENT, present)
    hi = np.array([1.0, 0.8, 0.6][:dim])
    pts = np.concatenate([s.vertices, rng.uniform(-0.1, 1.1, (200, dim)) * hi])
    # before any update_stress the projected stress is zero, as fluid_solver.stress is
    v0, st0, c0 = ctx.fsi_fluid_at_points(pts)
    assert not st0.any()
    stress = ctx.update_stress(0.7)
    vo, so, co = orc.fsi_fluid_at_points(m, present, stress, pts)
    v, st, c = ctx.fsi_fluid_at_points(pts)
    assert (c == co).all() and (c0 == co).all() and (co >= 0).sum() > len(s.vertices) and (co < 0).sum() > 10
    assert np.abs(v - vo).max() < 1e-12 * np.abs(vo).max() and np.abs(v0 - vo).max() < 1e-12 * np.abs(vo).max()
    assert np.abs(st - so).max() < 1e-12 * np.abs(so).max()
    assert not v[co < 0].any() and not st[co < 0].any()
    ctx.close()


def test_fluid_values_at_solid_points_on_four_virtual_ranks():
    """every point of the mesh is found by at least one rank, with the single-context cell and values"""
    from openifem_amd import capi
    m, s, present, rng = _case("hang2_q1")
    stress = orc.System(m).update_stress(0.7, present)
    pts = np.concatenate([s.vertices, rng.uniform(-0.1, 1.1, (150, 2)) * np.array([1.0, 0.8])])
    vo, so, co = orc.fsi_fluid_at_points(m, present, stress, pts)
    c = m.vcoords.mean(axis=1)
    parts = partition_mesh(m, (c[:, 0] > 0.52).astype(int) + 2 * (c[:, 1] > 0.41).astype(int), 4)

    def work(rank, P, ctx):
        if len(P.hang_dof):
            ctx.set_hanging_constraints(P.hang_dof, P.hang_ptr, P.hang_master, P.hang_weight)
        ctx.vec_set(capi.VEC_PRESENT, present[P.ext_gdof])
        ctx.update_stress(0.7)
        v, st, cl = ctx.fsi_fluid_at_points(pts)
        found = cl >= 0
        gc = np.where(found, P.cells[np.maximum(cl, 0)], -1)
        # a point found here lies in that cell on the single context too; unless two local cells tie (a shared face), it is the same
        assert np.abs(v[found] - vo[found]).max() < 1e-12 * np.abs(vo).max()
        assert np.abs(st[found] - so[found]).max() < 1e-11 * np.abs(so).max()
        assert (co[found] >= 0).all() and not v[~found].any()
        return gc

    res = run_virtual_ranks(capi, parts, work)
    found_any = np.stack([r >= 0 for r in res]).any(axis=0)
    assert (found_any == (co >= 0)).all()
    lowest = np.min(np.where(np.stack(res) >= 0, np.stack(res), m.n_cells), axis=0)
    assert (lowest[co >= 0] == co[co >= 0]).all()  # the lowest cell over the ranks is the single-context cell


def test_set_constraints_from_the_line_list():
    """ifem_set_constraints builds flags / inhomogeneities on the device from the line list: a dof listed twice keeps its last
    line (what the sequential host loop of round 1 did), an empty list clears the set, bad dofs are refused"""
    from openifem_amd import capi
    m, s, present, rng = _case("box2_q1")
    ctx = capi.Context(2, 1, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    dofs = np.array([5, 9, 5, 40, 9, 5], np.int32)
    vals = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    ctx.set_constraints(1, dofs, vals)
    f, v = ctx.get_constraints(1)
    assert set(np.nonzero(f)[0]) == {5, 9, 40} and (v[[5, 9, 40]] == [6.0, 5.0, 4.0]).all() and np.count_nonzero(v) == 3
    ctx.set_constraints(0, dofs, None)
    f0, v0 = ctx.get_constraints(0)
    assert (f0 == f).all() and not v0.any()
    ctx.set_constraints(1, np.zeros(0, np.int32), None)
    f, v = ctx.get_constraints(1)
    assert not f.any() and not v.any()
    with pytest.raises(capi.IfemError):
        ctx.set_constraints(1, np.array([m.n_dofs], np.int32), None)
    with pytest.raises(capi.IfemError):
        ctx.set_constraints(1, np.array([m.n_u], np.int32), None)  # a pressure dof
    f0b, _ = ctx.get_constraints(0)
    assert (f0b == f0).all()  # a refused call leaves the other set alone
    ctx.close()
