"""The FSI caller's contract with the fluid step (BASELINE config 5, tests/fsi_leaflet_mpi: SCnsIM<2> Q1/Q1 on a locally
refined, distributed mesh driven by MPI::FSI), exercised through the C ABI.

MPI::FSI::run (source/mpi_fsi.cpp:1186-1212) does, EVERY time step, on the fluid side:
  update_indicator()                (:291-321)   cell indicator = all vertices inside the solid
  fluid_solver.make_constraints()   (:1192)      boundary lines re-made; after the first step nonzero_ := copy of zero_ (:1193-1198)
  find_fluid_bc()                   (:323-663)   nodal fsi_stress[k] = fluid stress - solid stress at the nodes inside the solid
                                                 of indicator cells (:469-471, persistent elsewhere); use_dirichlet_bc = false:
                                                 fsi_acceleration = (v_s - v)/dt + ... - a_s at those velocity dofs (:548-556);
                                                 use_dirichlet_bc = true (fsi_leaflet_mpi.cpp:91): Dirichlet lines
                                                 v_s - present on the velocity dofs inside the solid, merged with
                                                 left_object_wins into both AffineConstraints (:615-650)
  fluid_solver.run_one_step(true)   (:1208)      Newton loop with apply_nonzero_constraints = true, then update_stress
The solid solver, the point location and the interpolation are the FSI side (outside the path): a rigid disc
with prescribed motion and an analytic stress field stands in for them here; both sides of the comparison receive the
same numbers.  What is under test is the fluid side of that loop: per-step changes of the constrained-dof set, of the
indicator, of fsi_acceleration and of the nodal fsi_stress TOGETHER, on a hanging-node mesh, on one context and on four
virtual ranks, against the oracle running the same loop (oracle assembly of SCnsIM::assemble + the condensation that
tests/test_oracle_hanging.py shows to be distribute_local_to_global + an exact sparse solve).
"""
import threading

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

import orc
from hangmesh import HangingMesh
from partmesh import local_dirichlet, partition_mesh, run_virtual_ranks

pytestmark = pytest.mark.gpu

L_, H_, HC = 4.0, 1.0, 0.125  # channel, coarse cell size
A_ = 0.25                    # "leaflet" scale of fsi_leaflet_mpi.cpp:66-76: cells with centre in [L/4 - 2a, L/4 + 3a] are refined
KW = dict(mu=0.1, rho=1.0, dt=0.01, solid_rho=2.5, g=(0.0, 0.0), neumann={})
R_DISC, C0, AS = 0.3, np.array([0.72, 0.52]), np.array([0.3, -0.2])
V_PATH = np.array([9.0, 1.0])   # the disc crosses 1.5 fine cells per step (its path is prescribed, not integrated from V_S)
V_S = np.array([1.2, 0.15])     # rigid-body velocity handed to the fluid
OMEGA = 0.8
NEWTON_TOL, NEWTON_MAXIT = 1e-8, 10


def _mesh():
    reps = (int(L_ / HC), int(H_ / HC))
    refine = {(i, j) for i in range(reps[0]) for j in range(reps[1])
              if L_ / 4 - 2 * A_ <= (i + 0.5) * HC <= L_ / 4 + 3 * A_}
    m = HangingMesh(reps, (0, 0), (L_, H_), refine, kv=1)
    assert len(m.hang_dof) > 0
    return m


def _solid(step):
    """rigid disc: centre, velocity / acceleration / stress fields at points x [n, 2]"""
    t = step * KW["dt"]
    c = C0 + V_PATH * t
    def vel(x):
        r = x - c
        return V_S + OMEGA * np.stack([-r[:, 1], r[:, 0]], axis=1)
    def acc(x):
        return np.broadcast_to(AS, x.shape) - OMEGA ** 2 * (x - c)
    def stress(x):  # symmetric, components ordered (0,0), (1,0), (1,1) as the loops of mpi_fsi.cpp:459-474
        return np.stack([0.4 + 0.3 * x[:, 0], 0.05 * x[:, 1] - 0.1 * x[:, 0], -0.2 + 0.1 * x[:, 1] * x[:, 0]], axis=0)
    inside = lambda x: ((x - c) ** 2).sum(axis=1) <= R_DISC ** 2  # noqa: E731
    return inside, vel, acc, stress


def fsi_inputs(m, step, present, stress, fsi_stress, boundary, use_dirichlet_bc):
    """what MPI::FSI hands to the fluid solver before run_one_step(true) of time step `step` (0-based): indicator per cell,
    fsi_acceleration (global block vector), fsi_stress (updated IN PLACE where the reference assigns it), and the two
    constraint sets (dofs, nonzero values)."""
    dim, n_u = 2, m.n_u
    inside, vel, acc, sstress = _solid(step)
    v_in = inside(m.vcoords.reshape(-1, dim)).reshape(m.n_cells, -1)
    indicator = v_in.all(axis=1).astype(np.int32)
    assert indicator.sum() > 0, "the solid covers no fluid cell: the test exercises nothing"
    node_in = inside(m.unode_coords)
    in_ind_cell = np.zeros(m.n_unodes, bool)
    in_ind_cell[np.unique(m.cell_unodes[indicator == 1])] = True
    # nodal fsi_stress (scalar Q_k space = the velocity nodes): fluid stress - solid stress (:469-471)
    sel = in_ind_cell & node_in
    fl = np.stack([stress[0, 0], stress[1, 0], stress[1, 1]], axis=0)
    fsi_stress[:, sel] = fl[:, sel] - sstress(m.unode_coords)[:, sel]
    fsi_acc = np.zeros(m.n_dofs)
    bdofs, bvals = boundary
    taken = set(int(d) for d in bdofs) | set(int(d) for d in m.hang_dof)
    dofs, vals = list(bdofs), list(bvals if step == 0 else np.zeros(len(bvals)))  # nonzero_ := zero_ after the first step
    v = present[:n_u].reshape(-1, dim)
    if not use_dirichlet_bc:  # (:489-556) only dofs of indicator cells; the convective part grad_v v needs FE gradients at
        nodes = np.nonzero(sel)[0]  # the support points (FSI-side code): the test feeds (v_s - v)/dt - a_s
        a = (vel(m.unode_coords[nodes]) - v[nodes]) / KW["dt"] - acc(m.unode_coords[nodes])
        for c in range(dim):
            fsi_acc[dim * nodes + c] = a[:, c]
    else:  # (:569-650) every non-artificial cell; Q1: every support point is a vertex, none is skipped as in-cell
        nodes = np.nonzero(node_in)[0]
        vs = vel(m.unode_coords[nodes])
        for k, nd in enumerate(nodes):
            for c in range(dim):
                dof = dim * int(nd) + c
                if dof in taken:  # left_object_wins: boundary and hanging lines stay
                    continue
                dofs.append(dof)
                vals.append(vs[k, c] - present[dof])
    return indicator, fsi_acc, np.array(dofs, np.int32), np.array(vals, float)


def _boundary(m):
    inflow = lambda p, c: 6.0 * p[1] * (H_ - p[1]) / H_ ** 2 if c == 0 else 0.0  # noqa: E731
    return m.dirichlet({0: (3, [0, 0]), 2: (3, [0, 0]), 3: (3, [0, 0])}, {0: inflow})


def oracle_loop(m, n_steps, use_dirichlet_bc, fsi_inputs=None):
    """the same loop on the CPU oracle; returns [(present, newton iterations)] per step"""
    fsi_inputs = fsi_inputs or globals()["fsi_inputs"]
    n = m.n_dofs
    Cm = sp.csr_matrix(m.prolongation())
    hang = m.hang_dof
    reg = np.setdiff1d(np.arange(n), hang)
    m.indicator = np.zeros(m.n_cells, np.int32)
    S = orc.System(m)  # borrows m.indicator: updated in place below
    ind_buf = S._keep[4]
    present = np.zeros(n)
    stress = np.zeros((2, 2, m.n_unodes))
    fsi_stress = np.zeros((3, m.n_unodes))
    boundary = _boundary(m)
    out = []
    for step in range(n_steps):
        indicator, fsi_acc, dofs, vals = fsi_inputs(m, step, present, stress, fsi_stress, boundary, use_dirichlet_bc)
        ind_buf[:] = indicator
        S.set_constraints(0, dofs, None)
        S.set_constraints(1, dofs, vals)
        isc = np.zeros(n, bool)
        isc[dofs] = True
        cv = np.zeros(n)
        cv[dofs] = vals
        # C with the Dirichlet masters closed away, and their inhomogeneity c0 (AffineConstraints::close)
        Cc = Cm.tolil(copy=True)
        c0 = np.zeros(n)
        for d in hang:
            row = Cm.getrow(d)
            for k, w in zip(row.indices, row.data):
                if isc[k]:
                    c0[d] += w * cv[k]
                    Cc[d, k] = 0.0
        Cc = Cc.tocsr()
        evalp = present.copy()
        cur = init = rel = 1.0
        it = 0
        while rel > NEWTON_TOL and cur > 1e-14:
            assert it < NEWTON_MAXIT, "oracle Newton loop does not converge"
            nz = it == 0
            P = orc.make_scns_params(stress=stress, fsi_stress=fsi_stress, **KW)
            S.scns_assemble(P, nz, evalp, present, fsi_acc)
            Ah, bh = S.csr("A"), S.rhs()
            off = c0 if nz else np.zeros(n)
            Ao = (Cc.T @ Ah @ Cc).tocsr()
            bo = Cc.T @ (bh - Ah @ off)
            x = np.zeros(n)
            x[reg] = spl.spsolve(Ao[reg][:, reg].tocsc(), bo[reg])
            x = Cm @ x  # constraints.distribute: hanging entries from all masters (Dirichlet masters hold their values)
            # system_rhs.l2_norm(): regular rows + the hanging rows, which carry diag * inhomogeneity
            hrow = np.abs(Ah.diagonal()[hang]) * off[hang]
            cur = np.sqrt((bo[reg] ** 2).sum() + (hrow ** 2).sum())
            evalp += x
            if it == 0:
                init = cur
            rel = cur / init
            it += 1
        present = evalp
        stress = S.update_stress(KW["mu"], present)
        out.append((present.copy(), it))
    m.indicator = None
    return out


def hip_loop(m, nranks, n_steps, use_dirichlet_bc):
    from openifem_amd import capi
    c = m.vcoords.mean(axis=1)
    if nranks == 1:
        cell_rank = np.zeros(m.n_cells, int)
    else:  # 2 x 2 blocks cutting through the refined band and through the path of the disc
        cell_rank = (c[:, 0] > 1.02).astype(int) + 2 * (c[:, 1] > 0.5).astype(int)
    parts = partition_mesh(m, cell_rank, nranks)
    n = m.n_dofs
    shared = {"present": np.zeros(n), "stress": np.zeros((2, 2, m.n_unodes))}
    fsi_stress = [np.zeros((3, m.n_unodes)) for _ in range(nranks)]  # every rank keeps its own (identical) copy
    barrier = threading.Barrier(nranks)
    boundary = _boundary(m)
    out = [[] for _ in range(nranks)]

    def work(rank, P, ctx):
        ctx.set_hanging_constraints(P.hang_dof, P.hang_ptr, P.hang_master, P.hang_weight)
        ctx.vec_set(capi.VEC_PRESENT, np.zeros(P.n_local))
        prm = capi.make_scns_params(**KW)
        nuo, nul = P.n_unodes_owned, P.n_unodes
        for step in range(n_steps):
            # the FSI side works on localized (global) vectors, as the reference does (Vector<double> localized_*, :350-362)
            indicator, fsi_acc, dofs, vals = fsi_inputs(m, step, shared["present"], shared["stress"], fsi_stress[rank], boundary,
                                                        use_dirichlet_bc)
            barrier.wait()  # everybody has read the shared state of the previous step
            ctx.set_indicator(indicator[P.cells])
            ctx.vec_set(capi.VEC_FSI_ACC, fsi_acc[P.ext_gdof])
            ctx.set_scns_fields(fsi_stress=fsi_stress[rank][:, P.l2g_u])
            ld, lv = local_dirichlet(P, dofs, vals)
            ctx.set_constraints(1, ld, lv)
            ctx.set_constraints(0, ld, None)
            its, log = ctx.scns_newton_step(prm, True, tol=NEWTON_TOL, maxit=NEWTON_MAXIT)
            x = ctx.vec_get(capi.VEC_PRESENT)
            st = ctx.update_stress(KW["mu"])  # the projected stress the step left behind (FluidSolver::update_stress)
            shared["present"][P.own_gdof] = np.concatenate([x[:2 * nuo], x[2 * nul:2 * nul + P.n_pnodes_owned]])
            shared["stress"][:, :, P.owned_u] = st[:, :, :nuo]
            barrier.wait()
            out[rank].append((shared["present"].copy() if rank == 0 else None, its))
        return None

    run_virtual_ranks(capi, parts, work)
    assert all([o[1] for o in out[r]] == [o[1] for o in out[0]] for r in range(nranks)), "ranks disagree on Newton counts"
    return out[0]


@pytest.mark.parametrize("use_dirichlet_bc", [True, False])
@pytest.mark.parametrize("nranks", [1, 4])
def test_fsi_caller_loop_matches_oracle(nranks, use_dirichlet_bc):
    m = _mesh()
    n_steps = 4
    ref = oracle_loop(m, n_steps, use_dirichlet_bc)
    got = hip_loop(m, nranks, n_steps, use_dirichlet_bc)
    # the set of constrained dofs and the indicator really change from step to step
    sets = [set(int(d) for d in fsi_inputs(m, s, ref[max(s - 1, 0)][0], np.zeros((2, 2, m.n_unodes)), np.zeros((3, m.n_unodes)),
                                           _boundary(m), use_dirichlet_bc)[2]) for s in range(n_steps)]
    inds = [fsi_inputs(m, s, ref[0][0], np.zeros((2, 2, m.n_unodes)), np.zeros((3, m.n_unodes)), _boundary(m), False)[0]
            for s in range(n_steps)]
    assert any((inds[s] != inds[s + 1]).any() for s in range(n_steps - 1))
    if use_dirichlet_bc:
        assert any(sets[s] != sets[s + 1] for s in range(n_steps - 1))
    n_u = m.n_u
    for s in range(n_steps):
        xr, itr = ref[s]
        xg, itg = got[s]
        # velocity and pressure separately, relative to their own scale; 1e-6 = what Newton to 1e-8 on an FGMRES solve to
        # 1e-6 ||rhs|| (mpi_supg_solver.cpp:311-312) leaves against the oracle's exact linear solves
        ev = np.abs(xg[:n_u] - xr[:n_u]).max() / np.abs(xr[:n_u]).max()
        ep = np.abs(xg[n_u:] - xr[n_u:]).max() / max(np.abs(xr[n_u:]).max(), 1e-300)
        assert ev < 1e-6 and ep < 1e-6, (s, ev, ep, itr, itg)
        assert abs(itg - itr) <= 1, (s, itr, itg)


# ---- the same loop with the FSI side's work done from a MESHED solid: on the CPU by oracle/oracle_fsi.c (the restatement of
# update_indicator / find_fluid_bc), on the GPU by ifem_fsi_update_indicator / ifem_fsi_find_fluid_bc -- nothing
# but the solid's vertices and nodal fields crosses the boundary per step.

def _meshed_solid(step):
    """a distorted Q1 'leaflet' block in rigid motion: position, nodal velocity / acceleration / stress of time step `step`"""
    from solidmesh import lattice_solid, wobble
    t = step * KW["dt"]
    base = lattice_solid((14, 8), (0.47, 0.36), (1.01, 0.68), mapping=wobble(0.004, 11.0))
    c0 = base.vertices.mean(axis=0)
    s = base.moved(shift=V_PATH * t, rot=0.2 + OMEGA * 4 * t, about=c0)
    c = c0 + V_PATH * t
    r = s.vertices - c
    s.velocity = V_S + OMEGA * np.stack([-r[:, 1], r[:, 0]], axis=1)
    s.acceleration = np.broadcast_to(AS, r.shape) - OMEGA ** 2 * r
    x = s.vertices
    s.stress = np.stack([0.4 + 0.3 * x[:, 0], 0.05 * x[:, 1] - 0.1 * x[:, 0], -0.2 + 0.1 * x[:, 1] * x[:, 0]], axis=0)
    return s


def fsi_inputs_meshed(m, step, present, stress, fsi_stress, boundary, use_dirichlet_bc):
    """fsi_inputs through the oracle's restatement of mpi_fsi.cpp:291-663 (fsi_stress is updated in place)"""
    s = _meshed_solid(step)
    indicator = orc.fsi_update_indicator(m, s)
    assert indicator.sum() > 0
    fsi_acc, flag, val, nf = orc.fsi_find_fluid_bc(m, s, indicator, KW["dt"], use_dirichlet_bc, present, stress, fsi_stress)
    assert nf == 0
    bdofs, bvals = boundary
    taken = np.zeros(m.n_dofs, bool)
    taken[bdofs] = True
    taken[m.hang_dof] = True
    new = np.nonzero((flag == 1) & ~taken[:m.n_u])[0]  # merge with left_object_wins (:641-651)
    dofs = np.concatenate([bdofs, new]).astype(np.int32)
    vals = np.concatenate([bvals if step == 0 else np.zeros(len(bvals)), val[new]])
    return indicator, fsi_acc, dofs, vals


def hip_loop_device_inputs(m, nranks, n_steps, use_dirichlet_bc):
    from openifem_amd import capi
    c = m.vcoords.mean(axis=1)
    cell_rank = np.zeros(m.n_cells, int) if nranks == 1 else (c[:, 0] > 1.02).astype(int) + 2 * (c[:, 1] > 0.5).astype(int)
    parts = partition_mesh(m, cell_rank, nranks)
    shared = {"present": np.zeros(m.n_dofs)}
    barrier = threading.Barrier(nranks)
    bdofs, bvals = _boundary(m)
    out = [[] for _ in range(nranks)]

    def work(rank, P, ctx):
        ctx.set_hanging_constraints(P.hang_dof, P.hang_ptr, P.hang_master, P.hang_weight)
        ctx.vec_set(capi.VEC_PRESENT, np.zeros(P.n_local))
        prm = capi.make_scns_params(**KW)
        nuo, nul = P.n_unodes_owned, P.n_unodes
        moved = []
        for step in range(n_steps):
            s = _meshed_solid(step)
            ctx.fsi_set_solid(s.vertices, s.cells, s.bfaces, s.velocity, s.acceleration, s.stress)  # update_solid_box (:1189)
            ind, _ = ctx.fsi_update_indicator(len(P.cells))                                           # update_indicator (:1190)
            moved.append(ind)
            ld, lv = local_dirichlet(P, bdofs, bvals if step == 0 else np.zeros(len(bvals)))          # make_constraints (:1191-1197)
            ctx.set_constraints(1, ld, lv)
            ctx.set_constraints(0, ld, None)
            st = ctx.fsi_find_fluid_bc(KW["dt"], use_dirichlet_bc, cell_order=P.cells)                # find_fluid_bc (:1203)
            assert st.n_not_found == 0
            its, log = ctx.scns_newton_step(prm, True, tol=NEWTON_TOL, maxit=NEWTON_MAXIT)            # run_one_step(true) (:1208)
            x = ctx.vec_get(capi.VEC_PRESENT)
            # find_solid_bc (:727-760): fluid stress sigma = -p I + viscous stress at the solid's vertices, from the device
            v, tau, cl = ctx.fsi_fluid_at_points(s.vertices)
            shared.setdefault(("sigma", step), {})[rank] = (cl >= 0, tau - v[:, 2, None, None] * np.eye(2))
            barrier.wait()
            shared["present"][P.own_gdof] = np.concatenate([x[:2 * nuo], x[2 * nul:2 * nul + P.n_pnodes_owned]])
            barrier.wait()
            sigma = None
            if rank == 0:  # a vertex is found by the rank(s) whose local cells hold it
                parts_ = shared[("sigma", step)]
                sigma = np.full((len(s.vertices), 2, 2), np.nan)
                for r in range(nranks):
                    found, sg = parts_[r]
                    sigma[found] = sg[found]
            out[rank].append((shared["present"].copy() if rank == 0 else None, its, sigma))
        assert any((moved[k] != moved[k + 1]).any() for k in range(n_steps - 1)) or nranks > 1
        return None

    run_virtual_ranks(capi, parts, work)
    assert all([o[1] for o in out[r]] == [o[1] for o in out[0]] for r in range(nranks)), "ranks disagree on Newton counts"
    return out[0]


@pytest.mark.parametrize("use_dirichlet_bc", [True, False])
@pytest.mark.parametrize("nranks", [1, 4])
def test_fsi_caller_loop_with_device_produced_inputs(nranks, use_dirichlet_bc):
    m = _mesh()
    n_steps = 3
    ref = oracle_loop(m, n_steps, use_dirichlet_bc, fsi_inputs=fsi_inputs_meshed)
    got = hip_loop_device_inputs(m, nranks, n_steps, use_dirichlet_bc)
    n_u = m.n_u
    for s in range(n_steps):
        xr, itr = ref[s]
        xg, itg, sigma = got[s]
        ev = np.abs(xg[:n_u] - xr[:n_u]).max() / np.abs(xr[:n_u]).max()
        ep = np.abs(xg[n_u:] - xr[n_u:]).max() / max(np.abs(xr[n_u:]).max(), 1e-300)
        assert ev < 1e-6 and ep < 1e-6, (s, ev, ep, itr, itg)
        assert abs(itg - itr) <= 1, (s, itr, itg)
        # the traction data the solid side takes back (find_solid_bc): the oracle's interpolation of ITS solution and stress
        sol = _meshed_solid(s)
        vo, tauo, co = orc.fsi_fluid_at_points(m, xr, orc.System(m).update_stress(KW["mu"], xr), sol.vertices)
        assert (co >= 0).all() and not np.isnan(sigma).any()
        sigo = tauo - vo[:, 2, None, None] * np.eye(2)
        assert np.abs(sigma - sigo).max() < 1e-5 * np.abs(sigo).max(), (s, np.abs(sigma - sigo).max(), np.abs(sigo).max())


def mirror_loop_device_inputs(n_steps, use_dirichlet_bc):
    """BASELINE config 5's 2D fluid set-up on the C++ host mirror, line for line with tests/fsi_leaflet_mpi/fsi_leaflet_mpi.cpp:
    56-78 -- subdivided_hyper_rectangle, the refinement loop over the band [L/4 - 2a, L/4 + 3a], SCnsIM<2>,
    add_hard_coded_boundary_condition -- and the per-step calls of MPI::FSI::run (mpi_fsi.cpp:1186-1212) on the device.  The
    mesh, its hanging-node lines and the boundary lines are the mirror's own (csrc/host/grid.cpp), no Python mesh helper."""
    from openifem_amd import capi, host
    prm = f"""
subsection Simulation
  set Simulation type = FSI
  set Dimension = 2
  set Global refinements = 0, 0
  set End time = 1
  set Time step size = {KW["dt"]}
  set Output interval = 1
  set Refinement interval = 1000
  set Save interval = 1000
  set Gravity = 0.0, 0.0
end
subsection Fluid finite element system
  set Pressure degree = 1
  set Velocity degree = 1
end
subsection Fluid material properties
  set Dynamic viscosity = {KW["mu"]}
  set Fluid density = {KW["rho"]}
end
subsection Fluid solver control
  set Grad-Div stabilization = 0.1
  set Max Newton iterations = {NEWTON_MAXIT}
  set Nonlinear system tolerance = {NEWTON_TOL}
end
subsection Fluid Dirichlet BCs
  set Use hard-coded boundary values = 1
  set Number of Dirichlet BCs = 3
  set Dirichlet boundary id = 0, 2, 3
  set Dirichlet boundary components = 3, 3, 3
  set Dirichlet boundary values = 0, 0, 0, 0, 0, 0
end
subsection Fluid Neumann BCs
  set Number of Neumann BCs = 0
end
subsection Solid material properties
  set Solid density = {KW["solid_rho"]}
end
"""
    reps = (int(L_ / HC), int(H_ / HC))
    flow = host.SCnsIM(prm, reps, (0, 0), (L_, H_))
    assert flow.refine_band(0, L_ / 4 - 2 * A_, L_ / 4 + 3 * A_) > 0           # (fsi_leaflet_mpi.cpp:65-75)
    flow.add_hard_coded_boundary_condition(0, lambda p, c, t: 6.0 * p[1] * (H_ - p[1]) / H_ ** 2 if c == 0 else 0.0)
    flow.setup(0)
    n_cells, n_u, n_p = flow.sizes()
    assert len(flow.hanging_lines()[0]) > 0
    ctx = capi.Context.borrow(flow.ctx, 2, 1, n_u // 2, n_p)
    out = []
    for step in range(n_steps):
        s = _meshed_solid(step)
        ctx.fsi_set_solid(s.vertices, s.cells, s.bfaces, s.velocity, s.acceleration, s.stress)  # update_solid_box (:1189)
        ind, _ = ctx.fsi_update_indicator(n_cells)                                               # update_indicator (:1190)
        flow.make_constraints(zero_inhomogeneities=step > 0)                                      # make_constraints (:1191-1198)
        st = ctx.fsi_find_fluid_bc(KW["dt"], use_dirichlet_bc)                                    # find_fluid_bc (:1203)
        assert st.n_not_found == 0
        flow.run_one_step(True)                                                                   # run_one_step(true) (:1208)
        v, p = flow.get_current_solution()
        out.append((np.concatenate([v, p]), ind))
    flow.close()
    return out


@pytest.mark.parametrize("use_dirichlet_bc", [True, False])
def test_fsi_caller_loop_on_the_host_mirror_with_its_own_locally_refined_mesh(use_dirichlet_bc):
    m = _mesh()  # the oracle's side keeps the independent builder; tests/test_host_layer.py shows the two meshes are equal
    n_steps = 3
    ref = oracle_loop(m, n_steps, use_dirichlet_bc, fsi_inputs=fsi_inputs_meshed)
    got = mirror_loop_device_inputs(n_steps, use_dirichlet_bc)
    n_u = m.n_u
    assert any((got[k][1] != got[k + 1][1]).any() for k in range(n_steps - 1)), "the indicator never moved"
    for s in range(n_steps):
        xr, _ = ref[s]
        xg, _ = got[s]
        ev = np.abs(xg[:n_u] - xr[:n_u]).max() / np.abs(xr[:n_u]).max()
        ep = np.abs(xg[n_u:] - xr[n_u:]).max() / max(np.abs(xr[n_u:]).max(), 1e-300)
        assert ev < 1e-6 and ep < 1e-6, (s, ev, ep)


def test_fsi_loop_3d_insimex_with_device_produced_inputs():
    """the 3D coupling of the reference runs InsIMEX<3> under MPI::FSI with the penalty form (tests/fsi_leaflet_mpi/
    fsi_leaflet_mpi.cpp:105, use_dirichlet_bc = false): per step update_solid_box / update_indicator / make_constraints /
    find_fluid_bc on the device, then InsIMEX::run_one_step(true) (matrix re-assembled: the indicator moved) -- against the
    oracle doing the same with its restatements, on a Q2/Q1 box with a rotated hexahedral solid in rigid motion."""
    from openifem_amd import capi
    from boxmesh import BoxMesh
    from solidmesh import lattice_solid, rotation, wobble
    m = BoxMesh((8, 6, 5), (0, 0, 0), (1.0, 0.8, 0.6), kv=2)
    kw = dict(mu=0.05, rho=1.0, gamma=0.1, dt=0.01, g=(0.0, 0.0, 0.0), neumann={})
    bdofs, bvals = m.dirichlet({0: (7, [0.3, 0.0, 0.0]), 2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (7, [0, 0, 0]), 5: (7, [0, 0, 0])})
    base = lattice_solid((4, 3, 3), (0.23, 0.21, 0.13), (0.71, 0.62, 0.49),
                         mapping=lambda p: wobble(0.008, 6.0)(rotation(0.3, (0.47, 0.41))(p)))
    c0 = base.vertices.mean(axis=0)
    vs, om = np.array([0.8, 0.1, -0.05]), 0.6

    def solid(step):
        t = step * kw["dt"]
        s = base.moved(shift=np.array([4.0, 0.5, 0.3]) * t, rot=2 * om * t, about=c0)
        r = s.vertices - (c0 + np.array([4.0, 0.5, 0.3]) * t)
        s.velocity = vs + om * np.stack([-r[:, 1], r[:, 0], 0 * r[:, 0]], axis=1)
        s.acceleration = np.array([0.2, -0.1, 0.3]) - om ** 2 * r * np.array([1, 1, 0])
        s.stress = None
        return s

    n_steps = 3
    # oracle
    m.indicator = np.zeros(m.n_cells, np.int32)
    S = orc.System(m)
    ind_buf = S._keep[4]
    x = np.zeros(m.n_dofs)
    ref, inds = [], []
    ainv = orc.SpluAinv()
    for step in range(n_steps):
        s = solid(step)
        ind = orc.fsi_update_indicator(m, s)
        inds.append(ind)
        acc, _, _, nf = orc.fsi_find_fluid_bc(m, s, ind, kw["dt"], False, x, None, None)
        assert nf == 0 and ind.sum() > 0 and np.abs(acc).max() > 0
        ind_buf[:] = ind
        S.set_constraints(1, bdofs, bvals if step == 0 else np.zeros(len(bvals)))
        S.set_constraints(0, bdofs, None)
        rc, _, _ = S.imex_run_one_step(orc.make_params(**kw), True, True, x, ainv=ainv, fsi_acc=acc)
        assert rc == 0
        ref.append(x.copy())
    m.indicator = None
    assert any((inds[k] != inds[k + 1]).any() for k in range(n_steps - 1))
    # device
    ctx = capi.Context(3, 2, m.vcoords, m.cell_unodes, m.cell_pnodes, m.cell_face_bid, m.n_unodes, m.n_pnodes)
    ctx.vec_set(capi.VEC_PRESENT, np.zeros(m.n_dofs))
    ctx.opts.inner_rel = 1e-4
    P = capi.make_params(**kw)
    for step in range(n_steps):
        s = solid(step)
        ctx.fsi_set_solid(s.vertices, s.cells, None, s.velocity, s.acceleration, None)
        ind, _ = ctx.fsi_update_indicator(m.n_cells)
        assert (ind == inds[step]).all()
        ctx.set_constraints(1, bdofs, bvals if step == 0 else np.zeros(len(bvals)))
        ctx.set_constraints(0, bdofs, None)
        st = ctx.fsi_find_fluid_bc(kw["dt"], False)
        assert st.n_not_found == 0
        ctx.imex_step(P, True, True)
        got = ctx.vec_get(capi.VEC_PRESENT)
        n_u = m.n_u
        ev = np.abs(got[:n_u] - ref[step][:n_u]).max() / np.abs(ref[step][:n_u]).max()
        ep = np.abs(got[n_u:] - ref[step][n_u:]).max() / np.abs(ref[step][n_u:]).max()
        assert ev < 1e-6 and ep < 1e-6, (step, ev, ep)
    ctx.close()


def test_insim_with_attached_multigrid_levels_under_fsi_dirichlet_lines():
    """InsIM::initialize_system attaches the multigrid chain on box meshes and makes IFEM_AINV_MG the default inner solver --
    also when the solver is the fluid side of MPI::FSI, whose find_fluid_bc (use_dirichlet_bc = true) merges Dirichlet lines of the
    artificial fluid into the FINE context's constraint sets every step while the coarse levels keep the boundary lines only
    (include/ifem_hip.h, ifem_mg_attach).  The levels then are a weaker preconditioner, nothing else: the per-step loop converges
    and gives the solution of the same loop without levels (Jacobi-preconditioned inner GMRES)."""
    from openifem_amd import capi, host
    from solidmesh import lattice_solid, rotation
    reps, ext = (16, 8, 8), (2.0, 0.4, 0.4)
    base = lattice_solid((4, 3, 3), (0.55, 0.12, 0.11), (0.95, 0.29, 0.30), mapping=rotation(0.2, (0.7, 0.2)))
    dt = 1e-3

    def solid(step):
        s = base.moved(shift=np.array([0.5, 0.02, 0.01]) * step * dt * 40, rot=0.0, about=base.vertices.mean(axis=0))
        s.velocity = np.tile(np.array([0.02, 0.001, 0.0005]), (len(s.vertices), 1))
        s.acceleration = np.zeros_like(s.velocity)
        s.stress = None
        return s

    sols, its = {}, {}
    for mg in (True, False):
        flow = host.InsIM(host.channel_prm(3, dt=dt), reps, (0, 0, 0), ext)
        flow.set_multigrid(mg)
        flow.setup(0)
        assert (len(flow.mg_levels()) > 0) == mg
        n_cells, n_u, n_p = flow.sizes()
        if not mg:
            flow.opts.inner_rel = 1e-3
            flow.opts.inner_maxit = 4000
        ctx = capi.Context.borrow(flow.ctx, 3, 2, n_u // 3, n_p)
        n_lines = []
        for step in range(2):
            s = solid(step)
            ctx.fsi_set_solid(s.vertices, s.cells, None, s.velocity, s.acceleration, None)
            ind, n_art = ctx.fsi_update_indicator(n_cells)
            flow.make_constraints(zero_inhomogeneities=step > 0)
            st = ctx.fsi_find_fluid_bc(dt, True)
            assert st.n_not_found == 0 and st.n_lines > 0
            n_lines.append(st.n_lines)
            flow.run_one_step(True)
        v, p = flow.get_current_solution()
        sols[mg] = np.concatenate([v, p])
        its[mg] = flow.last_stats().fgmres_iters
        flow.close()
    a, b = sols[True], sols[False]
    n_u3 = len(a) - n_p
    assert np.abs(a[:n_u3] - b[:n_u3]).max() <= 1e-5 * np.abs(b[:n_u3]).max()
    assert np.abs(a[n_u3:] - b[n_u3:]).max() <= 1e-4 * np.abs(b[n_u3:]).max()
    assert its[True] <= 30, its


def test_insim_run_one_step_leaves_the_projected_stress_of_the_new_solution():
    """InsIM::run_one_step ends with update_stress() (mpi_insim.cpp:474-475): an FSI caller on InsIM<3> (the reference's
    fsi_gravity_mpi) reads the projected viscous stress of the step's solution through ifem_fsi_fluid_at_points, not a stale or
    zero field.  Checked against an explicit ifem_update_stress of the same solution and against mu du/dy of the flow."""
    from openifem_amd import capi, host
    import ctypes as C
    s = host.InsIM(host.channel_prm(3, dt=1.0), (4, 4, 4), (0, 0, 0), (2.0, 0.2, 0.2))  # dt = 1: one step lands near steady Poiseuille
    s.setup(0)
    s.run_one_step(True)
    L = s.L
    pts = np.array([[0.7, 0.05, 0.1], [1.3, 0.15, 0.06], [0.31, 0.1, 0.13]])
    n, dim = len(pts), 3

    def at_points():
        vals, st, cell = np.zeros((n, dim + 1)), np.zeros((n, dim, dim)), np.zeros(n, np.int32)
        assert L.ifem_fsi_fluid_at_points(s.ctx, n, pts.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
                                          st.ctypes.data_as(C.c_void_p), cell.ctypes.data_as(C.c_void_p)) == 0
        return vals, st, cell
    v1, st1, c1 = at_points()
    assert (c1 >= 0).all() and np.abs(st1).max() > 0
    assert L.ifem_update_stress(s.ctx, C.c_double(1.0), None) == 0  # the same projection once more, explicitly
    v2, st2, c2 = at_points()
    assert np.abs(st1 - st2).max() <= 1e-12 * np.abs(st1).max() and np.array_equal(v1, v2)  # (the nodal average sums atomically: equal to rounding)
    # the flow is (close to) plane Poiseuille: T_xy = mu du/dy = dP / (2 L) (H - 2 y)
    exact = 10.0 / (2 * 2.0) * (0.2 - 2 * pts[:, 1])
    assert np.abs(st1[:, 0, 1] - exact).max() < 0.05 * np.abs(exact).max() + 0.02
    s.close()
