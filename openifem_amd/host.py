"""ctypes binding of the C++ host mirror (csrc/host/facade.cpp): Fluid::MPI::InsIM driven like a reference test."""
import ctypes as C

import numpy as np

from . import capi


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ifem host error {code}: {msg}")
        self.code = code


BC_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_uint, C.c_double)
FIELD_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_uint)


def _lib():
    L = capi.load()
    if getattr(L, "_ifemx_bound", False):
        return L
    L.ifemx_last_error.restype = C.c_char_p
    L.ifemx_insim_create_box.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p)]
    L.ifemx_solver_create_box.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.POINTER(C.c_void_p)]
    L.ifemx_solver_create_cylinder.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    for f in (L.ifemx_set_body_force, L.ifemx_set_sigma_pml_field, L.ifemx_set_initial_condition):
        f.argtypes = [C.c_void_p, FIELD_FN]
    L.ifemx_update_stress.argtypes = [C.c_void_p, C.c_void_p]
    L.ifemx_output_results.argtypes = [C.c_void_p, C.c_char_p, C.c_uint]
    L.ifemx_save_checkpoint.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.ifemx_load_checkpoint.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.ifemx_set_output_dir.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.ifemx_time.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_double)]
    L.ifemx_write_vtu.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ifemx_destroy.argtypes = [C.c_void_p]
    L.ifemx_insim_create_cylinder.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.ifemx_add_hard_coded_boundary_condition.argtypes = [C.c_void_p, C.c_int, BC_FN]
    L.ifemx_run.argtypes = [C.c_void_p]
    L.ifemx_setup.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_run_one_step.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_run_one_step2.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ifemx_setup_host_only.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.ifemx_assemble.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_solve.argtypes = [C.c_void_p, C.c_int, C.POINTER(capi.SolveStats)]
    L.ifemx_last_stats.argtypes = [C.c_void_p, C.POINTER(capi.SolveStats)]
    L.ifemx_last_newton.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ifemx_solver_opts.restype = C.POINTER(capi.SolverOpts)
    L.ifemx_solver_opts.argtypes = [C.c_void_p]
    L.ifemx_ctx.restype = C.c_void_p
    L.ifemx_ctx.argtypes = [C.c_void_p]
    L.ifemx_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ifemx_get_solution.argtypes = [C.c_void_p, C.c_void_p]
    L.ifemx_node_coords.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifemx_cell_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifemx_set_partition.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ifemx_set_node_order.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_partition_sizes.argtypes = [C.c_void_p, C.c_void_p]
    L.ifemx_partition_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    L.ifemx_sm_plan_sizes.argtypes = [C.c_void_p, C.c_void_p]
    L.ifemx_sm_plan_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    L.ifemx_refine_band.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int64)]
    L.ifemx_hanging_lines.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.ifemx_make_constraints.argtypes = [C.c_void_p, C.c_int]
    L.ifemx_set_multigrid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.ifemx_set_mg_replica_cells.argtypes = [C.c_void_p, C.c_int64]
    L.ifemx_mg_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    L.ifemx_coarse_level_chain.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    L.ifemx_box_prolongation.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifemx_box_injection.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.ifemx_box_injection_partial.argtypes = L.ifemx_box_injection.argtypes
    L.ifemx_nested_transfer_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.ifemx_channel_state.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_double]
    L._ifemx_bound = True
    return L


class FluidSolver:
    """Fluid::MPI::InsIM<dim> / SCnsIM<dim> on a colorised subdivided_hyper_rectangle (or the cylinder mesh of
    Utils::GridCreator), configured by a .prm text and driven like the reference's test drivers."""
    KIND = "InsIM"

    def __init__(self, prm_text, reps=None, p0=None, p1=None, device=0, verbose=False, mesh="box"):
        self.L = _lib()
        self._bc_keep = []
        if mesh == "cylinder":  # Utils::GridCreator<dim>::flow_around_cylinder, dim from the .prm (3: the extruded mesh)
            import re
            mdim = re.search(r"set\s+Dimension\s*=\s*(\d)", prm_text)
            self.dim = int(mdim.group(1)) if mdim else 2
            self.h = C.c_void_p()
            self._chk(self.L.ifemx_solver_create_cylinder(self.KIND.encode(), prm_text.encode(), device, int(verbose),
                                                          C.byref(self.h)))
            return
        self.dim = len(reps)
        self.reps = tuple(int(v) for v in reps)
        r = np.ascontiguousarray(reps, np.uint32)
        a, b = np.ascontiguousarray(p0, float), np.ascontiguousarray(p1, float)
        self.h = C.c_void_p()
        self._chk(self.L.ifemx_solver_create_box(self.KIND.encode(), prm_text.encode(), self.dim,
                                                 r.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                                 b.ctypes.data_as(C.c_void_p), device, int(verbose), C.byref(self.h)))

    def _chk(self, rc):
        if rc < 0:
            raise HostError(rc, self.L.ifemx_last_error().decode())

    def close(self):
        if self.h:
            self.L.ifemx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_hard_coded_boundary_condition(self, bid, fn):
        """fn(point: tuple, component: int, time: float) -> float, as FluidSolver::add_hard_coded_boundary_condition"""
        dim = self.dim
        cb = BC_FN(lambda p, c, t: float(fn(tuple(p[i] for i in range(dim)), int(c), float(t))))
        self._bc_keep.append(cb)
        self._chk(self.L.ifemx_add_hard_coded_boundary_condition(self.h, bid, cb))

    def _field(self, setter, fn):
        dim = self.dim
        cb = FIELD_FN(lambda p, c: float(fn(tuple(p[i] for i in range(dim)), int(c))))
        self._bc_keep.append(cb)
        self._chk(setter(self.h, cb))

    def set_body_force(self, fn):
        """fn(point, component) -> float, FluidSolver::set_body_force"""
        self._field(self.L.ifemx_set_body_force, fn)

    def set_sigma_pml_field(self, fn):
        self._field(self.L.ifemx_set_sigma_pml_field, fn)

    def set_initial_condition(self, fn):
        self._field(self.L.ifemx_set_initial_condition, fn)

    def update_stress(self):
        _, n_u, _ = self.sizes()
        out = np.zeros((self.dim, self.dim, n_u // self.dim))
        self._chk(self.L.ifemx_update_stress(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def output_results(self, directory, index):
        """FluidSolver::output_results: fluid_<index>.<rank>.vtu (+ .pvtu, fluid.pvd) into `directory`"""
        d = directory if directory.endswith("/") else directory + "/"
        self._chk(self.L.ifemx_output_results(self.h, d.encode(), index))

    def save_checkpoint(self, directory, index):
        """FluidSolver::save_checkpoint: <index>.fluid_checkpoint (+ .info, _fixed.data) into `directory`"""
        d = directory if directory.endswith("/") else directory + "/"
        self._chk(self.L.ifemx_save_checkpoint(self.h, d.encode(), index))

    def load_checkpoint(self, directory):
        """FluidSolver::load_checkpoint: restore the newest checkpoint of `directory` (sets the system up); False if none"""
        d = directory if directory.endswith("/") else directory + "/"
        found = C.c_int(0)
        self._chk(self.L.ifemx_load_checkpoint(self.h, d.encode(), C.byref(found)))
        return bool(found.value)

    def set_output_dir(self, directory, enable_output=False):
        """directory of run()'s results and checkpoints (the reference's working directory)"""
        d = directory if directory.endswith("/") else directory + "/"
        self._chk(self.L.ifemx_set_output_dir(self.h, d.encode(), int(enable_output)))

    def time(self):
        """(time step, current time) of the solver's Utils::Time"""
        n, t = C.c_uint(0), C.c_double(0)
        self._chk(self.L.ifemx_time(self.h, C.byref(n), C.byref(t)))
        return n.value, t.value

    def write_vtu(self, filename, solution, fsi_acc=None, stress=None, subdomain=0):
        sol = np.ascontiguousarray(solution, float)
        acc = None if fsi_acc is None else np.ascontiguousarray(fsi_acc, float)
        st = None if stress is None else np.ascontiguousarray(stress, float)
        self._chk(self.L.ifemx_write_vtu(self.h, filename.encode(), sol.ctypes.data_as(C.c_void_p),
                                         None if acc is None else acc.ctypes.data_as(C.c_void_p),
                                         None if st is None else st.ctypes.data_as(C.c_void_p), subdomain))

    def set_partition(self, P, rank, nccl_unique_id=None, local_world=None):
        """rank `rank` of a P[0] x P[1] x P[2] block partition; call before setup()."""
        Pa = np.ascontiguousarray(list(P) + [1] * (3 - len(P)), np.int32)
        idbuf = None if nccl_unique_id is None else np.ascontiguousarray(nccl_unique_id, np.uint8)
        self._chk(self.L.ifemx_set_partition(self.h, Pa.ctypes.data_as(C.c_void_p), rank,
                                             None if idbuf is None else idbuf.ctypes.data_as(C.c_void_p), local_world))

    def refine_band(self, direction, lo, hi):
        """cell->set_refine_flag() on the coarse cells whose centre lies in [lo, hi] along `direction`, then
        execute_coarsening_and_refinement (tests/fsi_leaflet_mpi/fsi_leaflet_mpi.cpp:65-75); before setup().  Returns the
        number of flagged cells."""
        n = C.c_int64(0)
        self._chk(self.L.ifemx_refine_band(self.h, int(direction), float(lo), float(hi), C.byref(n)))
        return n.value

    def hanging_lines(self):
        """(dof, ptr, master, weight) of the hanging-node lines the host mirror made for a locally refined mesh"""
        z = np.zeros(2, np.int64)
        self._chk(self.L.ifemx_hanging_lines(self.h, z.ctypes.data_as(C.c_void_p), None, None, None, None))
        dof, ptr = np.zeros(z[0], np.int32), np.zeros(z[0] + 1, np.int32)
        master, weight = np.zeros(z[1], np.int32), np.zeros(z[1])
        self._chk(self.L.ifemx_hanging_lines(self.h, None, *[a.ctypes.data_as(C.c_void_p) for a in (dof, ptr, master, weight)]))
        return dof, ptr, master, weight

    def make_constraints(self, zero_inhomogeneities=False):
        """FluidSolver::make_constraints again (every step of MPI::FSI::run); zero_inhomogeneities: nonzero_constraints :=
        zero_constraints, as the FSI driver does after the first step (mpi_fsi.cpp:1193-1198)"""
        self._chk(self.L.ifemx_make_constraints(self.h, int(zero_inhomogeneities)))

    def set_multigrid(self, on=True, min_cells=0, level_worlds=None):
        """FluidSolver::multigrid / mg_min_cells / mg_local_worlds of the C++ host mirror (insim.hpp); call before
        setup().  The hierarchy itself -- level chain, transfers, ifem_mg_attach -- is built by
        FluidSolver::attach_multigrid_levels inside initialize_system()."""
        worlds = list(level_worlds or [])
        arr = (C.c_void_p * max(len(worlds), 1))(*worlds)
        self._chk(self.L.ifemx_set_multigrid(self.h, int(on), int(min_cells), arr if worlds else None, len(worlds)))

    def set_mg_replica_cells(self, cells):
        """FluidSolver::mg_replica_cells: on several ranks, coarse meshes of at most this many cells become replicated single-rank
        levels (ifem_mg_attach's replicated coarse level) instead of partitioned ones; 0 keeps every level partitioned.  Before setup()."""
        self._chk(self.L.ifemx_set_mg_replica_cells(self.h, int(cells)))

    def mg_levels(self):
        """[(global repetitions, context handle)] of the levels attached below this solver's context, finest first"""
        n = C.c_int(0)
        reps = np.zeros((16, 3), np.int32)
        ctxs = (C.c_void_p * 16)()
        self._chk(self.L.ifemx_mg_levels(self.h, C.byref(n), reps.ctypes.data_as(C.c_void_p), ctxs))
        return [(tuple(int(v) for v in reps[k][:self.dim]), C.c_void_p(ctxs[k])) for k in range(n.value)]

    def all_ctxs(self):
        """this solver's context and those of its multigrid levels"""
        return [self.ctx] + [c for _, c in self.mg_levels()]

    def set_node_order(self, morton=True):
        self._chk(self.L.ifemx_set_node_order(self.h, int(morton)))

    def partition_sizes(self):
        out = np.zeros(10, np.int64)
        self._chk(self.L.ifemx_partition_sizes(self.h, out.ctypes.data_as(C.c_void_p)))
        keys = ["n_unodes_owned", "n_unodes_local", "n_pnodes_owned", "n_pnodes_local", "n_neighbors", "n_send_u",
                "n_send_p", "n_unodes_global", "n_pnodes_global", "n_cells_local"]
        return dict(zip(keys, out.tolist()))

    def partition_tables(self):
        z = self.partition_sizes()
        nn = z["n_neighbors"]
        t = dict(l2g_u=np.zeros(z["n_unodes_local"], np.int64), l2g_p=np.zeros(z["n_pnodes_local"], np.int64),
                 neighbors=np.zeros(nn, np.int32), send_u_ptr=np.zeros(nn + 1, np.int32),
                 send_u_idx=np.zeros(z["n_send_u"], np.int32), recv_u_ptr=np.zeros(nn + 1, np.int32),
                 send_p_ptr=np.zeros(nn + 1, np.int32), send_p_idx=np.zeros(z["n_send_p"], np.int32),
                 recv_p_ptr=np.zeros(nn + 1, np.int32))
        order = ["l2g_u", "l2g_p", "neighbors", "send_u_ptr", "send_u_idx", "recv_u_ptr", "send_p_ptr", "send_p_idx",
                 "recv_p_ptr"]
        self._chk(self.L.ifemx_partition_tables(self.h, *[t[k].ctypes.data_as(C.c_void_p) for k in order]))
        t.update(z)
        return t

    def sm_plan(self):
        """2-deep pressure halo plan of the distributed explicit S_m (None when a block is narrower than two cells)"""
        z = np.zeros(11, np.int64)
        self._chk(self.L.ifemx_sm_plan_sizes(self.h, z.ctypes.data_as(C.c_void_p)))
        if z[3] == 0:
            return None
        nn = self.partition_sizes()["n_neighbors"]
        t = dict(box_lo=z[0:3].copy(), box_n=z[3:6].copy(), lattice_n=z[6:9].copy(), n_far=int(z[10]),
                 box_id=np.zeros(int(z[3] * z[4] * z[5]), np.int32), send_s_ptr=np.zeros(nn + 1, np.int32),
                 send_s_idx=np.zeros(int(z[9]), np.int32), recv_s_ptr=np.zeros(nn + 1, np.int32))
        self._chk(self.L.ifemx_sm_plan_tables(self.h, *[t[k].ctypes.data_as(C.c_void_p) for k in
                                                        ("box_id", "send_s_ptr", "send_s_idx", "recv_s_ptr")]))
        return t

    def global_dofs(self):
        z = self.partition_sizes()
        return self.dim * z["n_unodes_global"] + z["n_pnodes_global"]

    def run(self):
        self._chk(self.L.ifemx_run(self.h))

    def setup(self, global_refinements=0):
        self._chk(self.L.ifemx_setup(self.h, global_refinements))

    def setup_host_only(self, global_refinements=0):
        self._chk(self.L.ifemx_setup_host_only(self.h, global_refinements))

    def constraints(self):
        n = C.c_int64()
        self._chk(self.L.ifemx_constraints(self.h, None, None, C.byref(n)))
        d, v = np.zeros(n.value, np.int32), np.zeros(n.value)
        self._chk(self.L.ifemx_constraints(self.h, d.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.byref(n)))
        return d, v

    def run_one_step(self, apply_nonzero, assemble_system=True):
        self._chk(self.L.ifemx_run_one_step2(self.h, int(apply_nonzero), int(assemble_system)))

    def last_stats(self):
        """counters of the most recent solve (the one inside run_one_step included)"""
        st = capi.SolveStats()
        self._chk(self.L.ifemx_last_stats(self.h, C.byref(st)))
        return st

    def last_newton(self):
        """(Newton iterations, summed FGMRES iterations) of the most recent run_one_step"""
        a, b = C.c_int32(), C.c_int32()
        self._chk(self.L.ifemx_last_newton(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def assemble(self, use_nonzero):
        self._chk(self.L.ifemx_assemble(self.h, int(use_nonzero)))

    def solve(self, use_nonzero):
        st = capi.SolveStats()
        self._chk(self.L.ifemx_solve(self.h, int(use_nonzero), C.byref(st)))
        return st

    @property
    def opts(self):
        return self.L.ifemx_solver_opts(self.h).contents

    @property
    def ctx(self):
        return C.c_void_p(self.L.ifemx_ctx(self.h))

    def sizes(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.L.ifemx_sizes(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_current_solution(self):
        _, n_u, n_p = self.sizes()
        x = np.zeros(n_u + n_p)
        self._chk(self.L.ifemx_get_solution(self.h, x.ctypes.data_as(C.c_void_p)))
        return x[:n_u], x[n_u:]

    def node_coords(self):
        _, n_u, n_p = self.sizes()
        uc, pc = np.zeros((n_u // self.dim, self.dim)), np.zeros((n_p, self.dim))
        self._chk(self.L.ifemx_node_coords(self.h, uc.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p)))
        return uc, pc

    def cell_tables(self, kv=2):
        n_cells, _, _ = self.sizes()
        nu, nv = (kv + 1) ** self.dim, 2 ** self.dim
        cu, cp = np.zeros((n_cells, nu), np.int32), np.zeros((n_cells, nv), np.int32)
        fb, vc = np.zeros((n_cells, 2 * self.dim), np.int32), np.zeros((n_cells, nv, self.dim))
        self._chk(self.L.ifemx_cell_tables(self.h, *[x.ctypes.data_as(C.c_void_p) for x in (cu, cp, fb, vc)]))
        return cu, cp, fb, vc

    def channel_state(self, L=2.0, H=0.2, dP=10.0, mu=1.0, seed=1234, rel=1e-3):
        self._chk(self.L.ifemx_channel_state(self.h, L, H, dP, mu, seed, rel))

    def true_residual(self):
        """(||b - A x||, ||b||) of the last solve, recomputed with the assembled operator (ifem_true_residual)"""
        r, b = C.c_double(0), C.c_double(0)
        rc = self.L.ifem_true_residual(self.ctx, C.byref(r), C.byref(b))
        if rc < 0:
            raise HostError(rc, self.L.ifem_last_error().decode())
        return r.value, b.value

    def comm_stats(self, reset=False):
        return capi.comm_stats(self.L, self.ctx, reset)

    def set_profiling(self, on=True):
        self.L.ifem_set_profiling(self.ctx, int(on))

    def kprof_begin(self):
        """start the per-kernel-family event log of the level chain (ifem_kprof_begin)"""
        rc = self.L.ifem_kprof_begin(self.ctx)
        if rc < 0:
            raise HostError(rc, self.L.ifem_last_error().decode())

    def kprof_end(self):
        """stop it: {family: {scopes, ms, bytes, flops}} (ifem_kprof_end)"""
        return capi.kprof_end(self.L, self.ctx)

    def synchronize(self):
        rc = self.L.ifem_synchronize(self.ctx)
        if rc < 0:
            raise HostError(rc, self.L.ifem_last_error().decode())

    def timing(self):
        t = capi.Timing()
        rc = self.L.ifem_get_timing(self.ctx, C.byref(t))
        if rc < 0:
            raise HostError(rc, self.L.ifem_last_error().decode())
        return t


def coarse_level_chain(cells_per_rank, P, extent, min_cells=4):
    """multigrid.hpp::coarse_level_chain of the C++ host mirror: per-rank cell counts of the coarser levels"""
    L = _lib()
    dim = len(cells_per_rank)
    n, p, e = (np.ascontiguousarray(v, t) for v, t in ((cells_per_rank, np.int32), (P, np.int32), (extent, float)))
    out, cnt = np.zeros((16, 3), np.int32), C.c_int(0)
    rc = L.ifemx_coarse_level_chain(dim, n.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                    min_cells, out.ctypes.data_as(C.c_void_p), C.byref(cnt))
    if rc < 0:
        raise HostError(rc, L.ifemx_last_error().decode())
    return [tuple(int(v) for v in out[k][:dim]) for k in range(cnt.value)]


def box_prolongation(reps_fine, reps_coarse, degree, l2g_fine, l2g_coarse):
    """multigrid.hpp::box_prolongation of the C++ host mirror as a scipy CSR matrix (tests compare it with the numpy
    restatement capi.box_prolongation)"""
    import scipy.sparse as sp
    L = _lib()
    dim = len(reps_fine)
    rf, rc_ = np.ascontiguousarray(reps_fine, np.int32), np.ascontiguousarray(reps_coarse, np.int32)
    lf, lc = np.ascontiguousarray(l2g_fine, np.int64), np.ascontiguousarray(l2g_coarse, np.int64)
    ptr = np.zeros(len(lf) + 1, np.int64)
    args = [dim, rf.ctypes.data_as(C.c_void_p), rc_.ctypes.data_as(C.c_void_p), degree, lf.ctypes.data_as(C.c_void_p), len(lf),
            lc.ctypes.data_as(C.c_void_p), len(lc), ptr.ctypes.data_as(C.c_void_p)]
    rc = L.ifemx_box_prolongation(*args, None, None)
    if rc < 0:
        raise HostError(rc, L.ifemx_last_error().decode())
    col, w = np.zeros(ptr[-1], np.int32), np.zeros(ptr[-1])
    rc = L.ifemx_box_prolongation(*args, col.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise HostError(rc, L.ifemx_last_error().decode())
    return sp.csr_matrix((w, col, ptr), shape=(len(lf), len(lc)))


def nested_transfer_check(dim, level, kv):
    """host/multigrid.cpp::nested_prolongation / nested_injection between the cylinder meshes of refinement `level` and level - 1:
    (max |row sum - 1|, coarse nodes whose fine twin does not interpolate from them alone, error of a linear function on the
    straight patches, share of the fine nodes on straight patches)"""
    L = _lib()
    out = np.zeros(4)
    rc = L.ifemx_nested_transfer_check(dim, level, kv, out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise HostError(rc, L.ifemx_last_error().decode())
    return tuple(out)


def box_injection(reps_fine, reps_coarse, degree, l2g_coarse, l2g_fine, partial=False):
    """partial: the fine list is one rank's owned nodes under a replicated coarse level -- -1 where another rank owns the node"""
    L = _lib()
    dim = len(reps_fine)
    rf, rc_ = np.ascontiguousarray(reps_fine, np.int32), np.ascontiguousarray(reps_coarse, np.int32)
    lf, lc = np.ascontiguousarray(l2g_fine, np.int64), np.ascontiguousarray(l2g_coarse, np.int64)
    out = np.zeros(len(lc), np.int32)
    rc = (L.ifemx_box_injection_partial if partial else L.ifemx_box_injection)(dim, rf.ctypes.data_as(C.c_void_p), rc_.ctypes.data_as(C.c_void_p), degree,
                               lc.ctypes.data_as(C.c_void_p), len(lc), lf.ctypes.data_as(C.c_void_p), len(lf),
                               out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise HostError(rc, L.ifemx_last_error().decode())
    return out


class InsIM(FluidSolver):
    KIND = "InsIM"


class InsIMEX(FluidSolver):
    KIND = "InsIMEX"


class SCnsIM(FluidSolver):
    KIND = "SCnsIM"


class SUPGInsIM(FluidSolver):
    KIND = "SUPGInsIM"


def channel_prm(dim=3, dt=1e-3, end_time=8e-2, refinements=0):
    """tests/fluid_pressure_driven/fluid_pressure_driven.prm transcribed for the 2D case and extended to the 3D
    channel of SURVEY 8(d): no-slip on y-walls, w = 0 on z-walls, inlet pressure 10."""
    if dim == 2:
        dirichlet = ("  set Number of Dirichlet BCs = 2\n  set Dirichlet boundary id = 2, 3\n"
                     "  set Dirichlet boundary components = 3, 3\n  set Dirichlet boundary values = 0, 0, 0, 0\n")
        grav = "0.0, 0.0"
    else:
        dirichlet = ("  set Number of Dirichlet BCs = 4\n  set Dirichlet boundary id = 2, 3, 4, 5\n"
                     "  set Dirichlet boundary components = 7, 7, 4, 4\n"
                     "  set Dirichlet boundary values = 0, 0, 0, 0, 0, 0, 0, 0\n")
        grav = "0.0, 0.0, 0.0"
    return f"""
subsection Simulation
  set Simulation type = Fluid
  set Dimension = {dim}
  set Global refinements = {refinements}, 0
  set End time = {end_time}
  set Time step size = {dt}
  set Output interval = 1e-2
  set Refinement interval = 1000
  set Save interval = 100
  set Gravity = {grav}
end
subsection Fluid finite element system
  set Pressure degree = 1
  set Velocity degree = 2
end
subsection Fluid material properties
  set Dynamic viscosity = 1
  set Fluid density = 1
end
subsection Fluid solver control
  set Grad-Div stabilization = 0.1
  set Max Newton iterations = 8
  set Nonlinear system tolerance = 1e-6
end
subsection Fluid Dirichlet BCs
  set Use hard-coded boundary values = 0
{dirichlet}end
subsection Fluid Neumann BCs
  set Number of Neumann BCs = 1
  set Neumann boundary id = 0
  set Neumann boundary values = 10
end
"""
