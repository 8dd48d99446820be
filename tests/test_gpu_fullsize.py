"""Bench-scale correctness of BASELINE config 3 (3D channel 128^3 Q2/Q1, one GPU) through the C ABI.

The entrywise oracle comparisons of test_gpu_parity.py run on a few hundred cells.  At 128^3 the assembled A_uu holds
9.8e9 doubles (offsets beyond 2^31, a million workgroups through the XCD remap, a tail block), which those tests never
touch.  The oracle cannot assemble this size in seconds, so the checks are size-independent properties:

  * assembled A_uu (MFMA cell kernel + atomic scatter) against the matrix-free A_uu (sum-factorised cell kernel, an
    independent code path) on random vectors: 1e-10 of the largest entry;
  * the true residual ||b - A x|| of the bench solve, recomputed with the assembled operator (not FGMRES's recurrence),
    against the reference's stopping rule 1e-4 ||b|| (mpi_insim.cpp:379-380);
  * closed forms of the uniform box mesh: row sums of M_p, diag(M_u), and the right-hand side of the state u = 0 under
    gravity (rho g_c int N_a), from the 1D Q2 / Q1 integrals.

IFEM_TEST_FULL_N (default 128) shrinks the mesh for a quick local run.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

P0, P1 = (0.0, 0.0, 0.0), (2.0, 0.2, 0.2)
GRAVITY = (0.3, 0.0, 0.0)


def _vec_set(S, vec, x):
    rc = S.L.ifem_vec_set(S.ctx, vec, x.ctypes.data_as(C.c_void_p))
    assert rc == 0, S.L.ifem_last_error()


def _vec_get(S, vec, n):
    x = np.empty(n)
    rc = S.L.ifem_vec_get(S.ctx, vec, x.ctypes.data_as(C.c_void_p))
    assert rc == 0, S.L.ifem_last_error()
    return x


def _int_1d(lat, n, h, vertex, mid):
    """per-node 1D integral on a uniform Q2 lattice (index 0..2n): `vertex` per adjacent cell, `mid` for cell midpoints"""
    adj = np.where((lat == 0) | (lat == 2 * n), 1.0, 2.0)
    return np.where(lat % 2 == 0, vertex * adj, mid) * h


@pytest.fixture(scope="module")
def solver():
    from openifem_amd import host
    n = int(os.environ.get("IFEM_TEST_FULL_N", "128"))
    prm = host.channel_prm(3).replace("set Gravity = 0.0, 0.0, 0.0", "set Gravity = %g, %g, %g" % GRAVITY)
    assert "Gravity = 0.3" in prm
    S = host.InsIM(prm, (n, n, n), P0, P1, verbose=False)
    S.setup(0)
    S.n = n
    yield S
    S.close()


def test_closed_form_mass_blocks_and_gravity_rhs(solver):
    from openifem_amd import capi
    S, n = solver, solver.n
    n_cells, n_u, n_p = S.sizes()
    assert n_cells == n ** 3 and n_u == 3 * (2 * n + 1) ** 3 and n_p == (n + 1) ** 3
    h = (np.array(P1) - np.array(P0)) / n
    uc, pc = S.node_coords()
    ulat = np.rint((uc - np.array(P0)) / (h / 2)).astype(np.int64)
    plat = np.rint((pc - np.array(P0)) / h).astype(np.int64)
    cdofs, _ = S.constraints()
    # state u = 0, p = 0: rhs_(a,c) = rho g_c int N_a on unconstrained rows away from the Neumann inlet, 0 on constrained rows
    assert S.L.ifem_vec_zero(S.ctx, capi.VEC_PRESENT) == 0 and S.L.ifem_vec_zero(S.ctx, capi.VEC_EVAL) == 0
    S.assemble(False)
    b = _vec_get(S, capi.VEC_RHS, n_u + n_p)
    lump = np.ones(n_u // 3)
    sq = np.ones(n_u // 3)
    for d in range(3):
        lump *= _int_1d(ulat[:, d], n, h[d], 1.0 / 6.0, 4.0 / 6.0)    # int N_i
        sq *= _int_1d(ulat[:, d], n, h[d], 4.0 / 30.0, 16.0 / 30.0)   # int N_i^2
    bu = b[:n_u].reshape(-1, 3)
    free = np.ones(n_u, bool)
    free[cdofs] = False
    free = free.reshape(-1, 3)
    interior = ulat[:, 0] > 0  # the inlet face carries the Neumann term as well
    for c in range(3):
        sel = free[:, c] & interior
        err = np.abs(bu[sel, c] - GRAVITY[c] * lump[sel]).max()
        assert err < 1e-12 * lump.max(), (c, err)
    assert np.abs(b[:n_u][cdofs]).max() == 0.0
    assert np.abs(b[n_u:]).max() < 1e-18
    # the inlet rows: -int p n.N over the face with n = -e_x, p = 10: + 10 * (2D lumped integral) on the x component
    inlet = (ulat[:, 0] == 0) & free[:, 0]
    face = _int_1d(ulat[:, 1], n, h[1], 1.0 / 6.0, 4.0 / 6.0) * _int_1d(ulat[:, 2], n, h[2], 1.0 / 6.0, 4.0 / 6.0)
    expect = GRAVITY[0] * lump[inlet] + 10.0 * face[inlet]
    assert np.abs(bu[inlet, 0] - expect).max() < 1e-12 * np.abs(expect).max()
    # mass blocks: M_p 1 = int psi_i, diag(M_u) = int N_a^2 for every component
    x = np.ones(n_u + n_p)
    _vec_set(S, capi.VEC_TMP, x)
    assert S.L.ifem_mass_vmult(S.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0, S.L.ifem_last_error()
    y = _vec_get(S, capi.VEC_UPDATE, n_u + n_p)
    plump = np.ones(n_p)
    for d in range(3):
        plump *= np.where((plat[:, d] == 0) | (plat[:, d] == n), 0.5, 1.0) * h[d]
    assert np.abs(y[n_u:] - plump).max() < 1e-12 * plump.max()
    assert abs(y[n_u:].sum() - np.prod(np.array(P1) - np.array(P0))) < 1e-12
    assert np.abs(y[:n_u].reshape(-1, 3) - sq[:, None]).max() < 1e-12 * sq.max()


def test_assembled_auu_equals_matrix_free_and_solve_residual(solver):
    from openifem_amd import capi
    S = solver
    _, n_u, n_p = S.sizes()
    nt = n_u + n_p
    S.channel_state()  # the bench state: Poiseuille + seeded 1e-3 perturbation
    S.opts.ainv_kind = 3
    S.opts.inner_rel = 1e-2
    S.opts.inner_restart = 16
    S.assemble(False)
    rng = np.random.default_rng(20260930)
    for k in range(3):
        x = rng.standard_normal(nt)
        _vec_set(S, capi.VEC_TMP, x)
        assert S.L.ifem_uu_vmult(S.ctx, capi.VEC_UPDATE, capi.VEC_TMP, 0) == 0, S.L.ifem_last_error()
        ya = _vec_get(S, capi.VEC_UPDATE, nt)[:n_u]
        assert S.L.ifem_uu_vmult(S.ctx, capi.VEC_UPDATE, capi.VEC_TMP, 3) == 0, S.L.ifem_last_error()
        ym = _vec_get(S, capi.VEC_UPDATE, nt)[:n_u]
        assert np.isfinite(ya).all()
        err = np.abs(ya - ym).max() / np.abs(ya).max()
        assert err < 1e-10, (k, err)
    # the bench solve, then the true residual with the assembled operator
    b = _vec_get(S, capi.VEC_RHS, nt)
    st = S.solve(False)
    x = _vec_get(S, capi.VEC_UPDATE, nt)
    _vec_set(S, capi.VEC_TMP, x)
    assert S.L.ifem_system_vmult(S.ctx, capi.VEC_UPDATE, capi.VEC_TMP) == 0, S.L.ifem_last_error()
    ax = _vec_get(S, capi.VEC_UPDATE, nt)
    cdofs, _ = S.constraints()
    # constraints.distribute() zeroed the constrained entries after the solve: leave those rows out (their equations
    # d_r x_r = 0 hold by construction) and the columns contribute nothing since x_r = 0
    r = b - ax
    r[cdofs] = 0.0
    true_res = np.linalg.norm(r)
    assert true_res <= 1.05e-4 * np.linalg.norm(b), (true_res, np.linalg.norm(b), st.fgmres_iters, st.fgmres_res)
    assert abs(true_res - st.fgmres_res) <= 0.5 * st.fgmres_res + 1e-12 * np.linalg.norm(b)


def test_fsi_inputs_closed_form_on_affine_fields(solver):
    """the device-side FSI inputs (csrc/fsi.hip) at bench size against what must hold on ANY mesh: with affine solid fields
    and an affine fluid velocity, update_indicator marks exactly the cells whose vertices lie in the (analytically known)
    block, and find_fluid_bc writes (v_s - v)/dt + (grad v) v - a_s and -(solid stress) at exactly the nodes of those
    cells that lie in the block -- 2.1 M cells, 17 M nodes, a 3456-cell rotated solid (mpi_fsi.cpp:291-319, :415-556)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fsibench
    from openifem_amd import capi
    S, n = solver, solver.n
    L, ctx = S.L, S.ctx
    n_cells, n_u, n_p = S.sizes()
    rng = np.random.default_rng(5)
    solid = fsibench.make_solid3d()
    Gv, Ga, Gf = rng.normal(size=(3, 3)), rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
    Gs = rng.normal(size=(6, 3))
    xs = solid["vertices"]
    solid["velocity"] = np.ascontiguousarray(xs @ Gv.T + 0.2)
    solid["acceleration"] = np.ascontiguousarray(xs @ Ga.T - 0.1)
    solid["stress"] = np.ascontiguousarray(Gs @ xs.T + 0.5)
    uc, pc = S.node_coords()
    cu, cp, fb, vc = S.cell_tables()
    present = np.concatenate([(uc @ Gf.T + 0.05).ravel(), rng.normal(size=n_p)])
    _vec_set(S, capi.VEC_PRESENT, present)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fs = capi.FsiSolid(len(xs), len(solid["cells"]), 0, p(xs), p(solid["cells"]), None, p(solid["velocity"]), p(solid["acceleration"]),
                       p(solid["stress"]))
    assert L.ifem_fsi_set_solid(ctx, C.byref(fs)) == 0, L.ifem_last_error()
    ind = np.zeros(n_cells, np.int32)
    cnt = C.c_int64()
    assert L.ifem_fsi_update_indicator(ctx, p(ind), C.byref(cnt)) == 0, L.ifem_last_error()
    v_in, v_gap = fsibench.inside_solid3d(vc.reshape(-1, 3))
    # 17 M points come within ~1e-10 of the block's faces; the analytic test and the Newton inversion agree down to
    # rounding, so only points closer than 1e-12 to a face would be ambiguous
    assert v_gap.min() > 1e-12, "a fluid vertex sits on the solid's surface: the closed form is ambiguous there"
    want_ind = v_in.reshape(n_cells, 8).all(axis=1)
    assert want_ind.sum() > 1000 * (n / 128.0) ** 3 and (ind == want_ind).all() and cnt.value == want_ind.sum()
    dt = 1e-3
    st = capi.FsiStats()
    try:
        assert L.ifem_fsi_find_fluid_bc(ctx, dt, 0, None, C.byref(st)) == 0, L.ifem_last_error()
        node_in, gap = fsibench.inside_solid3d(uc)
        assert gap.min() > 1e-12
        in_ind = np.zeros(len(uc), bool)
        in_ind[np.unique(cu[want_ind])] = True
        sel = node_in & in_ind
        assert st.n_inside == sel.sum() > 0 and st.n_not_found == 0
        acc = _vec_get(S, capi.VEC_FSI_ACC, n_u + n_p)
        v = present[:n_u].reshape(-1, 3)
        want = ((uc @ Gv.T + 0.2) - v) / dt + v @ Gf.T - (uc @ Ga.T - 0.1)
        want[~sel] = 0.0
        assert not acc[n_u:].any()
        assert np.abs(acc[:n_u].reshape(-1, 3) - want).max() < 1e-9 * np.abs(want).max()
        fsi_stress = np.zeros((6, len(uc)))
        assert L.ifem_fsi_get_stress(ctx, p(fsi_stress)) == 0, L.ifem_last_error()
        want_s = np.where(sel[None, :], -(Gs @ uc.T + 0.5), 0.0)  # no projected fluid stress on this context yet
        assert np.abs(fsi_stress - want_s).max() < 1e-11 * np.abs(want_s).max()
        # the way back (find_solid_bc): the fluid solution at the solid's vertices; the velocity is in the FE space
        nv_s = len(xs)
        vals, cl = np.zeros((nv_s, 4)), np.zeros(nv_s, np.int32)
        assert L.ifem_fsi_fluid_at_points(ctx, nv_s, p(xs), p(vals), None, p(cl)) == 0, L.ifem_last_error()
        assert (cl >= 0).all() and np.abs(vals[:, :3] - (xs @ Gf.T + 0.05)).max() < 1e-11
    finally:  # leave the shared context as the other tests expect it
        assert L.ifem_set_cell_fields(ctx, None) == 0
        assert L.ifem_vec_zero(ctx, capi.VEC_FSI_ACC) == 0
        assert L.ifem_set_scns_fields(ctx, None, None, None) == 0
