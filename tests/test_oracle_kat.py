"""Pins the CPU oracle against the known answers the reference's own tests assert.

The reference cannot be built here (deal.II/PETSc absent), so these analytic / regression constants
are the only reference artefacts available; A_uu^-1 is an exact sparse LU (scipy splu) standing in for
MUMPS / UMFPACK exactly as the reference does (mpi_insim.cpp:124-127, insim.cpp).
"""
import numpy as np
import pytest

import orc
from boxmesh import BoxMesh


def _run(mesh, bcs, params, n_steps, fields=None):
    S = orc.System(mesh)
    dofs, vals = mesh.dirichlet(bcs, fields)
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    ainv = orc.SpluAinv()
    for step in range(n_steps):
        rc, _ = S.run_one_step(params, step == 0, x, ainv=ainv)
        assert rc > 0, f"Newton failed at step {step}: rc={rc}"
    return S, x


def test_fluid_pressure_driven_poiseuille():
    # tests/fluid_pressure_driven/fluid_pressure_driven.cpp:33-45 (+ .prm): vmax = dP D^2 / (8 mu L) = 2.5e-2 at 1e-3.
    # Poiseuille is exact in Q2, so the coarser 50x5 mesh (reference: 200x20) carries the same answer.
    m = BoxMesh([50, 5], (0, 0), (2.0, 0.2), kv=2)
    P = orc.make_params(mu=1, rho=1, gamma=0.1, dt=1e-3, neumann={0: 10.0})
    S, x = _run(m, {2: (3, [0, 0]), 3: (3, [0, 0])}, P, 80)
    vmax = x[:S.n_u].max()
    assert abs(vmax - 2.5e-2) / 2.5e-2 < 1e-3
    # stronger than the reference: full parabolic profile u = dP/(2 mu L) y (D - y), v = 0
    y = m.unode_coords[:, 1]
    u_exact = 10.0 / (2 * 2.0) * y * (0.2 - y)
    assert np.abs(x[0:S.n_u:2] - u_exact).max() < 1e-7
    assert np.abs(x[1:S.n_u:2]).max() < 1e-7


def test_fluid_gravity_hydrostatic():
    # tests/fluid_gravity/fluid_gravity.cpp:31-41: pmax - pmin = rho g L = 20 at 1e-3, one step dt = 0.1
    m = BoxMesh([100, 10], (0, 0), (2.0, 0.2), kv=2)
    P = orc.make_params(mu=0.002, rho=1, gamma=0.1, dt=1e-1, g=(10.0, 0.0))
    S, x = _run(m, {0: (3, [0, 0]), 2: (3, [0, 0]), 3: (3, [0, 0])}, P, 1)
    p = x[S.n_u:]
    assert abs((p.max() - p.min()) - 20) / 20 < 1e-3


def test_fluid_pipe_mpi_developed_profile():
    # tests/fluid_pipe_mpi/fluid_pipe_mpi.cpp:38-55: uniform inflow 1 -> vmax = 1.5 at 1e-2, 20 steps dt = 0.1
    m = BoxMesh([100, 10], (0, 0), (2.0, 0.2), kv=2)
    P = orc.make_params(mu=0.002, rho=1, gamma=0.1, dt=1e-1)
    S, x = _run(m, {0: (3, [1, 0]), 2: (3, [0, 0]), 3: (3, [0, 0])}, P, 20)
    vmax = x[:S.n_u].max()
    assert abs(vmax - 1.5) / 1.5 < 1e-2


def test_poiseuille_3d_extension():
    # the bench workload -- plane Poiseuille in a 3D box, z-walls constrained in z only (flag 4).
    # Exact in Q2 => Umax = 2.5e-2 at every resolution.
    m = BoxMesh([6, 3, 2], (0, 0, 0), (2.0, 0.2, 0.2), kv=2)
    P = orc.make_params(mu=1, rho=1, gamma=0.1, dt=1e-3, neumann={0: 10.0})
    bcs = {2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (4, [0]), 5: (4, [0])}
    S, x = _run(m, bcs, P, 80)
    y = m.unode_coords[:, 1]
    u_exact = 10.0 / (2 * 2.0) * y * (0.2 - y)
    assert abs(x[:S.n_u].max() - 2.5e-2) / 2.5e-2 < 1e-3
    assert np.abs(x[0:S.n_u:3] - u_exact).max() < 1e-7
    assert np.abs(x[1:S.n_u:3]).max() < 1e-7 and np.abs(x[2:S.n_u:3]).max() < 1e-7
    # pressure is linear: p = 10 (1 - x/L)
    px = m.pnode_coords[:, 0]
    assert np.abs(x[S.n_u:] - 10.0 * (1 - px / 2.0)).max() < 1e-5


@pytest.mark.parametrize("dim,kv", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_fe_tables_textbook(dim, kv):
    # FE_Q / QGauss are third-party (deal.II) arithmetic: check partition of unity, Kronecker property of the
    # gradients' sum and quadrature exactness to degree 2n-1 (SURVEY 8c iii).
    import ctypes as C
    nq1 = kv + 1
    nn, nq = (kv + 1) ** dim, nq1 ** dim
    phi, dphi, w, qp = np.zeros((nq, nn)), np.zeros((nq, nn, dim)), np.zeros(nq), np.zeros((nq, dim))
    n = orc.lib().orc_fe_tables(dim, kv, nq1, phi.ctypes.data_as(C.c_void_p), dphi.ctypes.data_as(C.c_void_p),
                                w.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p))
    assert n == nq
    assert np.allclose(phi.sum(1), 1, atol=1e-14)
    assert np.allclose(dphi.sum(1), 0, atol=1e-13)
    assert abs(w.sum() - 1) < 1e-14
    deg = 2 * nq1 - 1
    for d in range(dim):
        assert abs((w * qp[:, d] ** deg).sum() - 1.0 / (deg + 1)) < 1e-14
    # interpolation of a degree-kv polynomial is exact
    lat = np.stack(np.unravel_index(np.arange(nn), (kv + 1,) * dim), -1)[:, ::-1] / kv
    f = lambda x: (1 + x[..., 0]) ** kv * (2 - x[..., -1]) ** kv
    assert np.allclose(phi @ f(lat), f(qp), atol=1e-13)


def test_cell_matrix_structure():
    # structural identities of the restated integrand (SURVEY 8c iii): Me symmetric, Ke - Ke^T = convection only,
    # zero velocity => Ke symmetric; constant pressure mode: sum_j Ke[u_i, p_j] = -int div(phi_i)
    rng = np.random.default_rng(0)
    m = BoxMesh([2, 2, 2], (0, 0, 0), (1.0, 0.7, 0.4), kv=2)
    m.vcoords = m.vcoords + 0.03 * rng.standard_normal(m.vcoords.shape)  # trilinear-distorted cells
    S = orc.System(m)
    P = orc.make_params(mu=0.7, rho=1.3, gamma=0.2, dt=0.01)
    zero = np.zeros(S.n)
    Ke, Me, fe = S.cell(P, 3, zero, zero)
    assert np.allclose(Me, Me.T, atol=1e-15)
    assert np.allclose(Ke, Ke.T, atol=1e-13)
    assert np.abs(fe).max() == 0
    ev = rng.standard_normal(S.n)
    Ke2, _, _ = S.cell(P, 3, ev, zero)
    nu = 27 * 3
    assert np.allclose(Ke2[nu:, :], Ke[nu:, :]) and np.allclose(Ke2[:, nu:], Ke[:, nu:])
    assert not np.allclose(Ke2[:nu, :nu], Ke2[:nu, :nu].T)
    assert np.allclose(Ke[nu:, nu:], 0)


def test_fluid_cylinder_mpi_regression_constants():
    # tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:89-96: MPI::InsIM<2>, cylinder mesh + 3 global refinements
    # (5 888 cells, 54 192 DoF = 48 064 + 6 128), one step dt = 1e-2, parabolic inflow Umax = 0.3:
    # vmax = 0.374235, pmax = 46.5226 at 1e-3 -- the only regression constants of the reference for mpi_insim.
    # The oracle reproduces them to < 1e-6, which pins assembly on non-affine cells, inhomogeneous constraints
    # and the Newton loop against real reference output.
    from cylmesh import CylinderMesh, inflow_bc
    m = CylinderMesh(3)
    assert (m.n_cells, m.n_u, m.n_pnodes) == (5888, 48064, 6128)
    P = orc.make_params(mu=0.001, rho=1, gamma=0.1, dt=1e-2)
    bcs = {0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}
    S, x = _run(m, bcs, P, 1, fields={0: inflow_bc})
    vmax, pmax = x[:S.n_u].max(), x[S.n_u:].max()
    assert abs(vmax - 0.374235) / 0.374235 < 1e-5
    assert abs(pmax - 46.5226) / 46.5226 < 1e-5


def test_fluid_cylinder_serial_regression_constants():
    # tests/fluid_cylinder/fluid_cylinder.cpp:83-85 (serial InsIM<2>, same integrals as MPI::InsIM): cylinder mesh with
    # 1 refinement (368 cells), 100 time steps dt = 1e-2: vmax = 0.4064759, pmax = 0.1539404 at 1e-3.
    # Pins the time loop (Newton restarts, zero constraints after the first step) over 100 steps: oracle error ~3e-8.
    from cylmesh import CylinderMesh, inflow_bc
    m = CylinderMesh(1)
    P = orc.make_params(mu=0.001, rho=1, gamma=0.1, dt=1e-2)
    bcs = {0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}
    S, x = _run(m, bcs, P, 100, fields={0: inflow_bc})
    vmax, pmax = x[:S.n_u].max(), x[S.n_u:].max()
    assert abs(vmax - 0.4064759) / 0.4064759 < 1e-6
    assert abs(pmax - 0.1539404) / 0.1539404 < 1e-6


def test_fluid_cylinder_mpi_scnsim_regression_constant():
    # tests/fluid_cylinder_mpi_scnsim/fluid_cylinder_mpi_scnsim.cpp:81-88: MPI::SCnsIM<2> (slightly compressible NS with
    # SUPG/PSPG/LSIC), cylinder mesh, Q1/Q1 (18 384 DoF), one step dt = 1e-2, inflow pulse Umax = 4.5 (time = dt < 2 dt):
    # vmax = 4.5, pmax = 1.03544 at 1e-3.  Oracle: 1.0354357 -- pins the SCnsIM integrand incl. the UGN length-scale quirk.
    from cylmesh import CylinderMesh
    m = CylinderMesh(3, kv=1)
    assert m.n_dofs == 18384
    S = orc.System(m)

    def inflow(p, c):
        return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0

    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    rc, _ = S.scns_run_one_step(orc.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2), True, x)
    assert rc > 0
    vmax, pmax = x[:S.n_u].max(), x[S.n_u:].max()
    assert abs(vmax - 4.5) / 4.5 < 1e-9
    assert abs(pmax - 1.03544) / 1.03544 < 2e-5


def test_update_stress_reproduces_linear_fields():
    # FluidSolver::update_stress (mpi_fluid_solver.cpp:716-811): for a velocity field linear in x the viscous stress
    # 2 mu sym(grad u) is constant, the per-cell projection and the nodal average must return it exactly
    m = BoxMesh([3, 2, 2], (0, 0, 0), (1.0, 0.7, 0.4), kv=2)
    rng = np.random.default_rng(1)
    m.vcoords = m.vcoords + 0.02 * rng.standard_normal(m.vcoords.shape)
    S = orc.System(m)
    Amat = rng.standard_normal((3, 3))
    x = np.zeros(S.n)
    # nodal values of u = A x at the (straight-sided) support points are not isoparametric on distorted cells, so use
    # the undistorted lattice coordinates for the field and undistorted cells for exactness
    m2 = BoxMesh([3, 2, 2], (0, 0, 0), (1.0, 0.7, 0.4), kv=2)
    S2 = orc.System(m2)
    x[:S2.n_u] = (m2.unode_coords @ Amat.T).ravel()
    st = S2.update_stress(0.7, x)
    tau = 2 * 0.7 * 0.5 * (Amat + Amat.T)
    assert np.abs(st - tau[:, :, None]).max() < 1e-12


def test_fluid_initial_condition_mpi_known_answer():
    # tests/fluid_initial_condition_mpi/fluid_initial_condition_mpi.cpp:32-62: MPI::SCnsIM<2> on 150 x 20 cells of
    # [0,15] x [0,2], Q1/Q1, pressure ramp as initial condition, one step of dt = 1e-11: pmax = 1e4 to 1e-8
    m = BoxMesh([150, 20], (0, 0), (15.0, 2.0), kv=1)
    S = orc.System(m)
    dofs, vals = m.dirichlet({0: (1, [0]), 1: (1, [0]), 2: (2, [0]), 3: (2, [0])})
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    px = m.pnode_coords[:, 0]
    x[S.n_u:] = np.where((px > 4.0) & (px < 5.0), 1e4 * (px - 4.0), np.where((px >= 5.0) & (px < 12.0), 1e4, 0.0))
    rc, _ = S.scns_run_one_step(orc.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-11), True, x)
    assert rc > 0
    assert abs(x[S.n_u:].max() - 1e4) / 1e4 < 1e-8


def _supg_insim_run(m, bcs, mu, n_steps, neumann=None):
    S = orc.System(m)
    dofs, vals = m.dirichlet(bcs)
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    P = orc.make_scns_params(mu=mu, rho=1.0, dt=1e-2, neumann=neumann, formulation=1)
    for step in range(n_steps):
        rc, _ = S.scns_run_one_step(P, step == 0, x)
        assert rc > 0
    return S, x


def test_fluid_pressure_driven_mpi_insim_supg_known_answer():
    # tests/fluid_pressure_driven_mpi_insim_supg: MPI::SUPGInsIM<2> (mpi_insim_supg.cpp), 100 x 10 cells refined once,
    # Q1/Q1, 10 steps of 1e-2: vmax within 2 % of P D^2 / (8 mu L) = 2.5e-2, the 30th largest value within 1e-3
    m = BoxMesh([200, 20], (0, 0), (2.0, 0.2), kv=1)
    S, x = _supg_insim_run(m, {2: (3, [0, 0]), 3: (3, [0, 0])}, 1.0, 10, neumann={0: 10.0})
    v = np.sort(x[:S.n_u])[::-1]
    assert abs(v[0] - 2.5e-2) / 2.5e-2 < 2e-2
    assert abs(v[29] - 2.5e-2) / 2.5e-2 < 1e-3


def test_fluid_plane_wall_driven_mpi_insim_supg_regression_constant():
    # tests/fluid_plane_wall_driven_mpi_insim_supg: 20 x 16 cells of [0,2] x [0,0.4], top wall u = (1, 0), mu = 0.002,
    # 10 steps: the l2 norm of the velocity block is 4.7112 at 1e-3
    m = BoxMesh([20, 16], (0, 0), (2.0, 0.4), kv=1)
    S, x = _supg_insim_run(m, {2: (3, [0, 0]), 3: (3, [1, 0])}, 0.002, 10)
    l2 = np.linalg.norm(x[:S.n_u])
    assert abs(l2 - 4.7112) / 4.7112 < 1e-3


def test_fluid_cylinder_mpi_insimex_regression_constants():
    # tests/fluid_cylinder_mpi_insimex/fluid_cylinder_mpi_insimex.cpp:65-77: MPI::InsIMEX<2> (mpi_insimex.cpp), cylinder
    # mesh refined 3 times, Q2/Q1, one step dt = 1e-2: vmax = 0.374062, pmax = 46.5308 at 1e-3
    from cylmesh import CylinderMesh, inflow_bc
    m = CylinderMesh(3)
    S = orc.System(m)
    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow_bc})
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    rc, it, res = S.imex_run_one_step(orc.make_params(mu=0.001, rho=1, gamma=0.1, dt=1e-2), True, True, x, ainv=orc.SpluAinv())
    assert rc == 0
    vmax, pmax = x[:S.n_u].max(), x[S.n_u:].max()
    assert abs(vmax - 0.374062) / 0.374062 < 1e-4
    assert abs(pmax - 46.5308) / 46.5308 < 1e-4


def test_fluid_cylinder_insimex_serial_time_loop_constants():
    # tests/fluid_cylinder_insimex (serial InsIMEX, same integrand and the same run() schedule as mpi_insimex.cpp:455-470):
    # 1 refinement, 100 steps of 1e-2, matrix assembled in the first two steps only (nonzero, then zero constraints),
    # rhs-only afterwards: vmax = 0.4081072, pmax = 0.1539 at 1e-3.  Pins assemble_system = false and the time loop.
    from cylmesh import CylinderMesh, inflow_bc
    m = CylinderMesh(1)
    S = orc.System(m)
    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow_bc})
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    P = orc.make_params(mu=0.001, rho=1, gamma=0.1, dt=1e-2)
    ainv = orc.SpluAinv()
    for step in range(100):
        rc, _, _ = S.imex_run_one_step(P, step == 0, step < 2, x, ainv=ainv)
        assert rc == 0
    vmax, pmax = x[:S.n_u].max(), x[S.n_u:].max()
    assert abs(vmax - 0.4081072) / 0.4081072 < 1e-3
    assert abs(pmax - 0.1539) / 0.1539 < 1e-3


def _acoustic_run(m, n_steps, dt, t0, width, sigma_of_x=None):
    """tests/acoustic_duct_wave_mpi/*.cpp:33-59 and tests/acoustic_pml_mpi/*.cpp:33-79: MPI::SCnsIM<2>, Q1/Q1, air
    (mu 1.8e-4, rho 1.3e-3), Gaussian velocity pulse 6 exp(-((t - t0)/width)^2 / 2) on the inlet as a hard-coded,
    time-dependent Dirichlet field; u_x = 0 at the outlet, u_y = 0 on the walls.  The Field returns the INCREMENT over
    the step (the Newton update is what the nonzero constraints act on); SUPGFluidSolver::run advances the Field time by
    dt before every step and re-makes the constraints (mpi_supg_solver.cpp:438-480), update_stress feeds the next step."""
    mu, rho = 1.8e-4, 1.3e-3
    S = orc.System(m)
    sigma = None
    if sigma_of_x is not None:  # PML damping at the quadrature points [n_cells][n_q]
        import ctypes as C
        nq1 = m.kv + 1
        qp = np.zeros((nq1 ** 2, 2))
        scratch = np.zeros(4096)
        orc.lib().orc_fe_tables(2, m.kv, nq1, scratch.ctypes.data_as(C.c_void_p), scratch.ctypes.data_as(C.c_void_p),
                                scratch.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p))
        x0, hx = m.vcoords[:, 0, 0], m.vcoords[:, 1, 0] - m.vcoords[:, 0, 0]
        sigma = sigma_of_x(x0[:, None] + qp[None, :, 0] * hx[:, None])
    pulse = lambda t: 6.0 * np.exp(-0.5 * ((t - t0) / width) ** 2)  # noqa: E731
    unit = {0: lambda p, c: 1.0 if (c == 0 and abs(p[0]) < 1e-10) else 0.0}
    dofs, shape = m.dirichlet({0: (1, [100]), 1: (1, [0]), 2: (2, [0]), 3: (2, [0])}, unit)
    S.set_constraints(0, dofs, None)
    x = np.zeros(S.n)
    stress = None
    for k in range(1, n_steps + 1):
        t = k * dt
        inc = pulse(t) - (0.0 if k < 2 else pulse(t - dt))
        S.set_constraints(1, dofs, shape * inc)
        P = orc.make_scns_params(mu=mu, rho=rho, dt=dt, stress=stress, sigma_pml=sigma)
        rc, _ = S.scns_run_one_step(P, True, x)
        assert rc > 0
        stress = S.update_stress(mu, x)
    return S, x


def test_acoustic_duct_wave_mpi_regression_constant():
    # tests/acoustic_duct_wave_mpi/acoustic_duct_wave_mpi.cpp:52-68: 8 x 2 cells of [0,4] x [0,1] refined 3 times, 1000
    # steps of 1e-7: the pulse travels down the duct at the isentropic speed of sound; max velocity 5.93 (the peak 6 with
    # the scheme's dispersion) at 1e-3 -- pins the compressibility terms of the SCnsIM integrand
    m = BoxMesh([64, 16], (0, 0), (4.0, 1.0), kv=1)
    S, x = _acoustic_run(m, 1000, 1e-7, 0.5e-4, 0.15e-4)
    vmax = x[:S.n_u].max()
    assert abs(vmax - 5.93) / 5.93 < 1e-3


def test_acoustic_pml_mpi_known_answer():
    # tests/acoustic_pml_mpi/acoustic_pml_mpi.cpp:33-84: 7 x 2 cells of [0,1.4] x [0,0.4] refined 3 times, quartic PML
    # sigma = 340000 ((x - 0.2) / 1.2)^4 for x > 0.2, 500 steps of 1e-7: the pulse is absorbed, |vmax| < 5e-2
    m = BoxMesh([56, 16], (0, 0), (1.4, 0.4), kv=1)
    S, x = _acoustic_run(m, 500, 1e-7, 0.5e-6, 0.15e-6,
                         sigma_of_x=lambda xq: np.where(xq > 0.2, 340000.0 * ((xq - 0.2) / 1.2) ** 4, 0.0))
    assert abs(x[:S.n_u].max()) < 5e-2
    assert np.abs(x[:S.n_u]).max() > 0  # the run did move the fluid


def test_subdomain_assembly_without_atomics_equals_the_cell_loop():
    """orc_ins_assemble_subdomains (owner computes row, one subdomain per thread: the CPU baseline leg of bench.py) against
    orc_ins_assemble on a distorted 3D Q2/Q1 mesh with inhomogeneous constraints and a Neumann face"""
    from boxmesh import BoxMesh, block_partition
    rng = np.random.default_rng(5)
    m = BoxMesh((5, 4, 3), (0, 0, 0), (1.0, 0.8, 0.6), kv=2)
    m.vcoords = m.vcoords + 0.01 * rng.standard_normal(m.vcoords.shape)
    dofs, vals = m.dirichlet({0: (7, [0.3, -0.2, 0.1]), 2: (7, [0, 0, 0]), 3: (1, [0.05])})
    kw = dict(mu=0.7, rho=1.3, gamma=0.2, dt=0.01, g=(0.3, -9.8, 0.5), neumann={1: 2.5})
    ev, pr = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    P = orc.make_params(**kw)
    for use_nonzero in (True, False):
        S.assemble(P, use_nonzero, ev, pr)
        A0, M0, b0 = S.csr("A"), S.csr("M"), S.rhs()
        for n_parts in (1, 6, 12):
            part, used = block_partition(m.reps, n_parts)
            assert used == n_parts and part.max() == n_parts - 1
            S.assemble_subdomains(P, use_nonzero, ev, pr, part, used, n_threads=4)
            A1, M1, b1 = S.csr("A"), S.csr("M"), S.rhs()
            assert abs(A1 - A0).max() <= 1e-13 * abs(A0).max()
            assert abs(M1 - M0).max() <= 1e-13 * abs(M0).max()
            assert np.abs(b1 - b0).max() <= 1e-13 * np.abs(b0).max()


def test_supg_preconditioner_restatement_solves_the_scnsim_system():
    """oracle.c::orc_scns_solve -- SUPGFluidSolver::solve with BlockIncompSchurPreconditioner as the reference builds it
    (mpi_supg_solver.cpp:19-192,297-328) -- reaches 1e-6 ||rhs|| on the assembled SCnsIM cylinder system and agrees with the
    direct solve the regression constant above was computed with; its pieces obey their definitions (L U x of an ILU(0) equals
    the matrix on the pattern's diagonal blocks; T_pp and B2pp differ only through P_vv^-1 vs rowsum^-1)"""
    import scipy.sparse.linalg as spl
    from cylmesh import CylinderMesh
    m = CylinderMesh(1, kv=1)
    S = orc.System(m)

    def inflow(p, c):
        return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0

    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    S.set_constraints(1, dofs, vals)
    S.set_constraints(0, dofs, None)
    zero = np.zeros(S.n)
    S.scns_assemble(orc.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2), True, zero, zero)
    A, b = S.csr("A").tocsc(), S.rhs()
    rc, upd, counts, res = S.scns_solve(True)
    assert rc == 0 and counts[0] > 0 and counts[2] == counts[0] and counts[1] > counts[0]
    free = np.ones(S.n, bool)
    free[dofs] = False
    assert np.linalg.norm((A @ upd - b)[free]) <= 1.05e-6 * np.linalg.norm(b)
    exact = spl.splu(A).solve(b)
    assert np.abs(upd - exact).max() <= 1e-4 * np.abs(exact).max()
    # P_vv^-1 is an approximate inverse of A_vv, B2pp_inverse of B2pp: one application leaves a residual well below 1
    nu = S.n_u
    rng = np.random.default_rng(0)
    xu, xp = rng.standard_normal(nu), rng.standard_normal(S.n - nu)
    Avv = A.tocsr()[:nu, :nu]
    assert np.linalg.norm(Avv @ S.scns_pc_probe(0, xu) - xu) < 0.9 * np.linalg.norm(xu)
    yp = S.scns_pc_probe(1, xp)
    assert np.linalg.norm(S.scns_pc_probe(2, yp) - xp) < 0.9 * np.linalg.norm(xp)
    # T_pp x = A_pp x - A_pv Pvv^-1 A_vp x, from its parts
    Ac = A.tocsr()
    t = Ac[nu:, nu:] @ xp - Ac[nu:, :nu] @ S.scns_pc_probe(0, Ac[:nu, nu:] @ xp)
    assert np.abs(S.scns_pc_probe(3, xp) - t).max() <= 1e-12 * np.abs(t).max()
