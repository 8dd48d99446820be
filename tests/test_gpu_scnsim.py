"""GPU parity of the SCnsIM path (slightly compressible NS, SUPG/PSPG/LSIC) against the oracle."""
import numpy as np
import pytest

import orc
from boxmesh import BoxMesh

pytestmark = pytest.mark.gpu


def _capi():
    import openifem_amd.capi as capi
    return capi


def _ctx(m):
    capi = _capi()
    return capi.Context(m.dim, m.kv, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)


@pytest.mark.parametrize("dim,kv,reps", [(2, 1, (5, 4)), (3, 1, (3, 3, 2)), (2, 2, (3, 2))])
@pytest.mark.parametrize("use_nonzero", [False, True])
def test_scns_assembly_matches_oracle_with_every_term(dim, kv, reps, use_nonzero):
    # distorted cells, inhomogeneous Dirichlet, Neumann pressure, gravity, PML, body force, artificial-fluid cells with
    # FSI acceleration / nodal FSI stress, projected viscous stress from update_stress -- all terms of mpi_scnsim.cpp:307-512
    capi = _capi()
    rng = np.random.default_rng(11 + dim + kv)
    m = BoxMesh(reps, (0,) * dim, (1.0, 0.6, 0.4)[:dim], kv=kv)
    m.vcoords = m.vcoords + 0.02 * rng.standard_normal(m.vcoords.shape)
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)})
    nq = (kv + 1) ** dim
    ev, pr = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
    ev[m.n_u:] *= 50.0
    pr[m.n_u:] *= 50.0
    ind = (rng.uniform(size=m.n_cells) < 0.35).astype(np.int32)
    acc = rng.standard_normal(m.n_dofs)
    sigma = rng.uniform(0, 3, (m.n_cells, nq))
    bf = rng.standard_normal((m.n_cells, nq, dim))
    fsi_stress = rng.standard_normal((dim * (dim + 1) // 2, m.n_unodes))
    eddy = 0.02 * rng.standard_normal(m.n_unodes)  # both signs: only the positive part of nu_t(q) counts
    kw = dict(mu=0.03, rho=1.2, dt=0.01, solid_rho=3.0, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5})
    # projected stress of the present solution: oracle vs HIP
    S = orc.System(m)
    ctx = _ctx(m)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    ctx.vec_set(capi.VEC_EVAL, ev)
    st_o = S.update_stress(kw["mu"], pr)
    st_g = ctx.update_stress(kw["mu"])
    assert np.abs(st_g - st_o).max() / np.abs(st_o).max() < 1e-12
    # assembly
    m.indicator = ind
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    Po = orc.make_scns_params(stress=st_o, fsi_stress=fsi_stress, sigma_pml=sigma, body_force=bf, eddy_viscosity=eddy, **kw)
    S.scns_assemble(Po, use_nonzero, ev, pr, acc)
    Ao, bo = S.csr("A"), S.rhs()
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.set_indicator(ind)
    ctx.vec_set(capi.VEC_FSI_ACC, acc)
    ctx.set_scns_fields(sigma, bf, fsi_stress)
    ctx.set_eddy_viscosity(eddy)
    ctx.scns_assemble(capi.make_scns_params(**kw), use_nonzero)
    A, b = ctx.export_csr(0), ctx.vec_get(capi.VEC_RHS)
    assert abs(A - Ao).max() / abs(Ao).max() < 1e-11
    assert np.abs(b - bo).max() / np.abs(bo).max() < 1e-11
    # system_vmult includes the A_pp block
    x = rng.standard_normal(m.n_dofs)
    y = ctx.system_vmult(x)
    assert np.abs(y - Ao @ x).max() / np.abs(y).max() < 1e-11
    m.indicator = None


def test_scns_solve_reaches_reference_tolerance():
    capi = _capi()
    from cylmesh import CylinderMesh
    m = CylinderMesh(1, kv=1)
    rng = np.random.default_rng(2)

    def inflow(p, c):
        return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0

    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    ctx = _ctx(m)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    P = capi.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2)
    ctx.scns_assemble(P, True)
    st = ctx.scns_solve(True)
    upd = ctx.vec_get(capi.VEC_UPDATE)
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    zero = np.zeros(S.n)
    S.scns_assemble(orc.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2), True, zero, zero)
    A, b = S.csr("A"), S.rhs()
    # FGMRES stops at 1e-6 ||rhs|| (mpi_supg_solver.cpp:311-312); constrained entries are overwritten afterwards
    free = np.ones(S.n, bool)
    free[dofs] = False
    r = (A @ upd - b)[free]
    assert np.linalg.norm(r) <= 1.05e-6 * np.linalg.norm(b)
    assert np.abs(upd[dofs] - vals).max() == 0.0
    assert st.fgmres_iters > 0


def test_kat_fluid_cylinder_mpi_scnsim_on_gpu():
    # config "tests/fluid_cylinder_mpi_scnsim": vmax = 4.5, pmax = 1.03544 (1e-3) through the HIP Newton loop
    capi = _capi()
    from cylmesh import CylinderMesh
    m = CylinderMesh(3, kv=1)

    def inflow(p, c):
        return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41) if (c == 0 and abs(p[0]) < 1e-10) else 0.0

    dofs, vals = m.dirichlet({0: (3, [0.2, 0]), 2: (3, [0, 0]), 3: (3, [0, 0]), 4: (3, [0, 0])}, {0: inflow})
    ctx = _ctx(m)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    n_it, log = ctx.scns_newton_step(capi.make_scns_params(mu=1.8e-4, rho=1.3e-3, dt=1e-2), True)
    assert n_it > 0
    _, vmax = ctx.minmax(capi.VEC_PRESENT, 0)
    _, pmax = ctx.minmax(capi.VEC_PRESENT, 1)
    assert abs(vmax - 4.5) / 4.5 < 1e-3
    assert abs(pmax - 1.03544) / 1.03544 < 1e-3


def _prm(name):
    import os
    return open(os.path.join(os.path.dirname(__file__), "golden", "prm", name)).read()


def test_reference_driver_fluid_cylinder_mpi_scnsim_through_host_mirror():
    # tests/fluid_cylinder_mpi_scnsim/fluid_cylinder_mpi_scnsim.cpp:31-88 on the C++ host mirror with the reference's own
    # .prm: AllParameters -> flow_around_cylinder -> SCnsIM<2> -> add_hard_coded_boundary_condition -> run()
    from openifem_amd import host
    prm = _prm("fluid_cylinder_mpi_scnsim.prm")
    dt = 1e-2

    def inflow_bc(p, component, time):  # pulse: active only while time < 2 dt (the Field time is dt at the first step)
        if component == 0 and abs(p[0]) < 1e-10 and time < 2 * dt:
            return 4 * 4.5 * p[1] * (0.41 - p[1]) / (0.41 * 0.41)
        return 0.0

    flow = host.SCnsIM(prm, mesh="cylinder")
    flow.add_hard_coded_boundary_condition(0, inflow_bc)
    flow.run()
    v, p = flow.get_current_solution()
    assert abs(v.max() - 4.5) / 4.5 < 1e-3
    assert abs(p.max() - 1.03544) / 1.03544 < 1e-3
    # update_stress ran at the end of the step (mpi_supg_solver.cpp:411): the nodal stress matches the oracle's
    cu, cp, fb, vc = flow.cell_tables(kv=1)
    m = type("M", (), {})()
    m.dim, m.kv, m.n_cells, m.n_unodes, m.n_pnodes = 2, 1, cu.shape[0], len(v) // 2, len(p)
    m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.indicator = vc, cu, cp, fb, None
    st_o = orc.System(m).update_stress(1.8e-4, np.concatenate([v, p]))
    st_g = flow.update_stress()
    assert np.abs(st_g - st_o).max() / np.abs(st_o).max() < 1e-11


def test_reference_driver_fluid_initial_condition_mpi():
    # tests/fluid_initial_condition_mpi/fluid_initial_condition_mpi.cpp:32-62: pressure ramp as initial condition, one
    # step of dt = 1e-11, pmax stays 1e4 to 1e-8
    from openifem_amd import host

    def initial_condition(pt, component):
        if component == 2:
            if 4.0 < pt[0] < 5.0:
                return 1e4 * (pt[0] - 4.0)
            if 5.0 <= pt[0] < 12.0:
                return 1e4
        return 0.0

    flow = host.SCnsIM(_prm("fluid_initial_condition_mpi.prm"), (150, 20), (0, 0), (15, 2))
    flow.set_initial_condition(initial_condition)
    flow.run()
    _, p = flow.get_current_solution()
    assert abs(p.max() - 1e4) / 1e4 < 1e-8


@pytest.mark.slow
def test_reference_driver_fluid_body_force_mpi():
    # tests/fluid_body_force_mpi/fluid_body_force_mpi.cpp:30-87: body force strip + PML, 500 steps of 1e-7; the pressure
    # jump across the strip is 1e3 to 1e-3
    from openifem_amd import host

    def body_force(pt, component):
        return 1.0e3 / 1.3e-3 if (3.5 - 5e-4 < pt[0] < 4.5 + 5e-4 and component == 0) else 0.0

    def sigma_pml(pt, component):
        s = 0.0
        for b in (0.0, 8.0):
            if abs(pt[0] - b) < 3.0:
                s = 340000 * ((3.0 - abs(pt[0] - b)) / 3.0) ** 4
        return s

    flow = host.SCnsIM(_prm("fluid_body_force_mpi.prm"), (160, 30), (0, 0), (8, 2))
    flow.set_body_force(body_force)
    flow.set_sigma_pml_field(sigma_pml)
    flow.run()
    _, p = flow.get_current_solution()
    assert abs((p.max() - p.min()) - 1e3) / 1e3 < 1e-3


def _gaussian_pulse(dt, t0, width):
    # the reference drivers' hard-coded inlet Field: the increment of 6 exp(-((t - t0)/width)^2 / 2) over the step
    import math

    def tv(t):
        return 6.0 * math.exp(-0.5 * ((t - t0) / width) ** 2)

    def field(pt, component, time):
        if component == 0 and abs(pt[0]) < 1e-10:
            return tv(time) - (0.0 if time < 2 * dt else tv(time - dt))
        return 0.0
    return field


@pytest.mark.slow
def test_reference_driver_acoustic_duct_wave_mpi(tmp_path):
    # tests/acoustic_duct_wave_mpi/acoustic_duct_wave_mpi.cpp:33-68 on the host mirror with the reference's .prm: 1000
    # steps of 1e-7, time-dependent hard-coded inlet pulse; max velocity 5.93 at 1e-3
    from openifem_amd import host
    flow = host.SCnsIM(_prm("acoustic_duct_wave_mpi.prm"), (8, 2), (0, 0), (4, 1))
    flow.add_hard_coded_boundary_condition(0, _gaussian_pulse(1e-7, 0.5e-4, 0.15e-4))
    flow.set_output_dir(str(tmp_path))  # Save interval 1e-6: the run writes checkpoints like the reference's does
    flow.run()
    v, _ = flow.get_current_solution()
    assert abs(v.max() - 5.93) / 5.93 < 1e-3


@pytest.mark.slow
def test_reference_driver_acoustic_pml_mpi(tmp_path):
    # tests/acoustic_pml_mpi/acoustic_pml_mpi.cpp:33-84: quartic PML over x > 0.2 of [0,1.4] x [0,0.4], 500 steps of 1e-7;
    # the pulse is absorbed: |vmax| < 5e-2
    from openifem_amd import host

    def sigma_pml(pt, component):
        return 340000 * ((pt[0] + 1.2 - 1.4) / 1.2) ** 4 if pt[0] > 1.4 - 1.2 else 0.0

    flow = host.SCnsIM(_prm("acoustic_pml_mpi.prm"), (7, 2), (0, 0), (1.4, 0.4))
    flow.add_hard_coded_boundary_condition(0, _gaussian_pulse(1e-7, 0.5e-6, 0.15e-6))
    flow.set_sigma_pml_field(sigma_pml)
    flow.set_output_dir(str(tmp_path))
    flow.run()
    import glob
    assert len(glob.glob(str(tmp_path) + "/*.fluid_checkpoint")) == 2  # the two latest of the 50 written
    v, _ = flow.get_current_solution()
    assert abs(v.max()) < 5e-2 and abs(v).max() > 0


@pytest.mark.parametrize("dim,kv,reps", [(2, 1, (5, 4)), (3, 1, (3, 3, 2)), (2, 2, (3, 2))])
def test_supg_insim_assembly_matches_oracle(dim, kv, reps):
    # IFEM_FORM_SUPG_INSIM (mpi_insim_supg.cpp:100-262): the incompressible SUPG/PSPG/LSIC integrand; indicator, PML and
    # stress inputs are present on purpose and must be ignored by this formulation
    capi = _capi()
    rng = np.random.default_rng(23 + dim + kv)
    m = BoxMesh(reps, (0,) * dim, (1.0, 0.6, 0.4)[:dim], kv=kv)
    m.vcoords = m.vcoords + 0.02 * rng.standard_normal(m.vcoords.shape)
    flag = 3 if dim == 2 else 7
    dofs, vals = m.dirichlet({0: (flag, [0.3, -0.2, 0.1][:dim]), 2: (flag, [0.0] * dim)})
    nq = (kv + 1) ** dim
    ev, pr = rng.standard_normal(m.n_dofs), rng.standard_normal(m.n_dofs)
    bf = rng.standard_normal((m.n_cells, nq, dim))
    kw = dict(mu=0.03, rho=1.2, dt=0.01, g=(0.3, -9.8, 0.5)[:dim], neumann={1: 2.5})
    S = orc.System(m)
    S.set_constraints(0, dofs, None)
    S.set_constraints(1, dofs, vals)
    S.scns_assemble(orc.make_scns_params(body_force=bf, formulation=1, **kw), True, ev, pr)
    Ao, bo = S.csr("A"), S.rhs()
    ctx = _ctx(m)
    ctx.set_constraints(0, dofs, None)
    ctx.set_constraints(1, dofs, vals)
    ctx.vec_set(capi.VEC_PRESENT, pr)
    ctx.vec_set(capi.VEC_EVAL, ev)
    ctx.set_indicator((rng.uniform(size=m.n_cells) < 0.5).astype(np.int32))
    ctx.set_scns_fields(rng.uniform(0, 3, (m.n_cells, nq)), bf, None)
    ctx.update_stress(kw["mu"])
    ctx.scns_assemble(capi.make_scns_params(formulation=capi.FORM_SUPG_INSIM, **kw), True)
    A, b = ctx.export_csr(0), ctx.vec_get(capi.VEC_RHS)
    assert abs(A - Ao).max() / abs(Ao).max() < 1e-11
    assert np.abs(b - bo).max() / np.abs(bo).max() < 1e-11


def test_reference_driver_fluid_pressure_driven_mpi_insim_supg():
    # tests/fluid_pressure_driven_mpi_insim_supg/*.cpp:31-56 on the host mirror with the reference's .prm
    from openifem_amd import host
    flow = host.SUPGInsIM(_prm("fluid_pressure_driven_mpi_insim_supg.prm"), (100, 10), (0, 0), (2.0, 0.2))
    flow.run()
    v, _ = flow.get_current_solution()
    v = np.sort(v)[::-1]
    assert abs(v[0] - 2.5e-2) / 2.5e-2 < 2e-2
    assert abs(v[29] - 2.5e-2) / 2.5e-2 < 1e-3


def test_reference_driver_fluid_plane_wall_driven_mpi_insim_supg():
    # tests/fluid_plane_wall_driven_mpi_insim_supg/*.cpp:31-50: |v|_2 = 4.7112 at 1e-3
    from openifem_amd import host
    flow = host.SUPGInsIM(_prm("fluid_plane_wall_driven_mpi_insim_supg.prm"), (20, 16), (0, 0), (2.0, 0.4))
    flow.run()
    v, _ = flow.get_current_solution()
    assert abs(np.linalg.norm(v) - 4.7112) / 4.7112 < 1e-3
