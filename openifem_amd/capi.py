"""ctypes binding of libifem_hip.so -- exactly the entry points declared in include/ifem_hip.h.

This is plumbing for tests and bench.py; the product is the shared library.  There is no CPU fallback:
loading works anywhere (hipcc cross-compiles), every compute call needs a HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the host driver of the MI355X boxes only supports dmabuf IPC: RCCL between processes needs this before the HSA runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
LIB_PATH = os.path.join(_HERE, "lib", "libifem_hip.so")

VEC_PRESENT, VEC_EVAL, VEC_FSI_ACC, VEC_UPDATE, VEC_RHS, VEC_INCREMENT, VEC_TMP = range(7)
E_BADPARAM, E_NODEVICE, E_HIP, E_KRYLOV_NOCONV, E_NEWTON_MAXIT, E_COMM = -1, -2, -3, -4, -5, -6  # IFEM_E_*
E_NODEVICE_EXIT = 66  # exit status of a bench rank that stopped at IFEM_E_NODEVICE (distinct from a Python traceback's 1)
AINV_GMRES_BJACOBI, AINV_GMRES_BJACOBI_F32, AINV_SCALAR_GMRES, AINV_GMRES_BJACOBI_MF, AINV_MG = range(5)  # IFEM_AINV_*


class MeshDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("kv", C.c_int32), ("n_cells", C.c_int32),
                ("n_unodes_owned", C.c_int32), ("n_unodes_local", C.c_int32),
                ("n_pnodes_owned", C.c_int32), ("n_pnodes_local", C.c_int32),
                ("vcoords", C.c_void_p), ("cell_unodes", C.c_void_p), ("cell_pnodes", C.c_void_p),
                ("cell_face_bid", C.c_void_p)]


class Partition(C.Structure):
    """ifem_partition, member for member (tests/test_host_layer.py checks the size against ifem_abi_sizeof)"""
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("n_neighbors", C.c_int32),
                ("neighbor_rank", C.c_void_p), ("send_u_ptr", C.c_void_p), ("send_u_idx", C.c_void_p),
                ("recv_u_ptr", C.c_void_p), ("send_p_ptr", C.c_void_p), ("send_p_idx", C.c_void_p),
                ("recv_p_ptr", C.c_void_p), ("nccl_unique_id", C.c_void_p), ("local_world", C.c_void_p),
                ("p_lattice_n", C.c_int64 * 3), ("sm_box_lo", C.c_int64 * 3), ("sm_box_n", C.c_int64 * 3),
                ("sm_box_id", C.c_void_p), ("l2g_p", C.c_void_p),
                ("send_s_ptr", C.c_void_p), ("send_s_idx", C.c_void_p), ("recv_s_ptr", C.c_void_p)]


def make_partition(rank, nranks, neighbors, send_u_ptr, send_u_idx, recv_u_ptr, send_p_ptr, send_p_idx, recv_p_ptr,
                   nccl_unique_id=None, local_world=None):
    """ifem_partition from numpy tables (general meshes: no 2-deep pressure halo).  Returns (struct, keep-alive list)."""
    keep = [np.ascontiguousarray(a, np.int32) for a in (neighbors, send_u_ptr, send_u_idx, recv_u_ptr, send_p_ptr,
                                                        send_p_idx, recv_p_ptr)]
    uid = None if nccl_unique_id is None else np.ascontiguousarray(nccl_unique_id, np.uint8)
    P = Partition()
    P.rank, P.nranks, P.n_neighbors = rank, nranks, len(keep[0])
    (P.neighbor_rank, P.send_u_ptr, P.send_u_idx, P.recv_u_ptr, P.send_p_ptr, P.send_p_idx, P.recv_p_ptr) = [_ptr(a) for a in keep]
    P.nccl_unique_id = _ptr(uid)
    P.local_world = local_world
    return P, keep + [uid]


class InsParams(C.Structure):
    _fields_ = [("viscosity", C.c_double), ("rho", C.c_double), ("grad_div", C.c_double), ("dt", C.c_double),
                ("gravity", C.c_double * 3), ("n_neumann", C.c_int32), ("neumann_id", C.c_int32 * 8),
                ("neumann_p", C.c_double * 8)]


class ScnsParams(C.Structure):
    _fields_ = [("viscosity", C.c_double), ("rho", C.c_double), ("dt", C.c_double), ("solid_rho", C.c_double),
                ("gravity", C.c_double * 3), ("n_neumann", C.c_int32), ("neumann_id", C.c_int32 * 8),
                ("neumann_p", C.c_double * 8), ("formulation", C.c_int32)]


FORM_SCNSIM, FORM_SUPG_INSIM = 0, 1


class SolverOpts(C.Structure):
    _fields_ = [("fgmres_restart", C.c_int32), ("fgmres_maxit", C.c_int32), ("fgmres_rel", C.c_double),
                ("fgmres_abs", C.c_double), ("mp_rel", C.c_double), ("mp_abs", C.c_double),
                ("sm_rel", C.c_double), ("sm_abs", C.c_double), ("ainv_kind", C.c_int32),
                ("inner_restart", C.c_int32), ("inner_maxit", C.c_int32), ("inner_rel", C.c_double),
                ("explicit_schur", C.c_int32), ("verbose", C.c_int32), ("device_cg", C.c_int32), ("outer_matrix_free", C.c_int32),
                ("sm_mg", C.c_int32), ("mg_smooth", C.c_int32), ("mg_cheb_ratio", C.c_double),
                ("mg_smooth_u", C.c_int32), ("mg_smooth_u_post", C.c_int32), ("mg_cheb_ratio_u", C.c_double),
                ("inner_rel_first", C.c_double), ("inner_first_pshare", C.c_double)]


class SolveStats(C.Structure):
    _fields_ = [("fgmres_iters", C.c_uint32), ("fgmres_res", C.c_double), ("precond_applies", C.c_uint32),
                ("cg_mp_iters", C.c_uint32), ("cg_sm_iters", C.c_uint32), ("inner_iters", C.c_uint32),
                ("t_schur_setup_ms", C.c_double), ("t_cg_mp_ms", C.c_double), ("t_cg_sm_ms", C.c_double),
                ("t_ainv_ms", C.c_double), ("t_spmv_ms", C.c_double), ("t_total_ms", C.c_double),
                ("sm_mg_levels", C.c_uint32), ("inner_first_tight", C.c_uint32)]


class MgTransfer(C.Structure):
    _fields_ = [("n_fine_p_owned", C.c_int64), ("n_coarse_p_local", C.c_int64),
                ("pp_ptr", C.c_void_p), ("pp_col", C.c_void_p), ("pp_w", C.c_void_p),
                ("rp_ptr", C.c_void_p), ("rp_col", C.c_void_p), ("rp_w", C.c_void_p),
                ("n_fine_u_owned", C.c_int64), ("n_coarse_u_local", C.c_int64),
                ("pu_ptr", C.c_void_p), ("pu_col", C.c_void_p), ("pu_w", C.c_void_p),
                ("ru_ptr", C.c_void_p), ("ru_col", C.c_void_p), ("ru_w", C.c_void_p), ("inj_u", C.c_void_p)]


class Tuning(C.Structure):
    _fields_ = [("geo_cache", C.c_int32), ("xcd_swizzle", C.c_int32), ("asm_skip", C.c_int32), ("spmv_lanes", C.c_int32),
                ("sm_lanes", C.c_int32), ("mf_f32", C.c_int32), ("tpp_operator", C.c_int32), ("spmv_pipe", C.c_int32), ("halo_overlap", C.c_int32),
                ("asm3_variant", C.c_int32), ("cg_single_reduction", C.c_int32), ("asm3_cpb", C.c_int32), ("tpp_milu_permille", C.c_int32), ("tpp_ilu_order", C.c_int64), ("basis_pad", C.c_int64), ("tpp_tri_sweeps", C.c_int32), ("uu_row_order", C.c_int32), ("eig_steps", C.c_int32), ("vcycle_graph_cells", C.c_int32),
                ("scns_pc", C.c_int32), ("pvv_sweeps", C.c_int32), ("b2pp_sweeps", C.c_int32), ("scns_inner_reorth", C.c_int32), ("scns_inner_left", C.c_int32), ("scns_graph", C.c_int32), ("stored_uu", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("assemble_ms", C.c_double), ("assemble_kernel_ms", C.c_double), ("spmv_uu_ms_avg", C.c_double),
                ("spmv_uu_calls", C.c_uint64), ("spmv_uu_bytes", C.c_double), ("mf_ms_avg", C.c_double),
                ("mf_calls", C.c_uint64)]


class KprofEntry(C.Structure):
    """ifem_kprof_entry"""
    _fields_ = [("family", C.c_int32), ("scopes", C.c_uint32), ("ms", C.c_double), ("bytes", C.c_double), ("flops", C.c_double)]


KC_COUNT = 17  # IFEM_KC_COUNT


def kprof_end(L, ctx):
    """ifem_kprof_end as a dict family name -> {scopes, ms, bytes, flops}"""
    buf = (KprofEntry * KC_COUNT)()
    n = L.ifem_kprof_end(ctx, buf, KC_COUNT)
    if n < 0:
        raise IfemError(n, L.ifem_last_error().decode())
    return {L.ifem_kprof_family_name(e.family).decode(): {"scopes": int(e.scopes), "ms": e.ms, "bytes": e.bytes, "flops": e.flops}
            for e in buf[:n]}


EXPORTS = ["ifem_last_error", "ifem_device_count", "ifem_default_solver_opts", "ifem_comm_unique_id",
           "ifem_local_world_create", "ifem_local_world_destroy", "ifem_comm_selftest",
           "ifem_ctx_create", "ifem_ctx_destroy", "ifem_n_local_dofs", "ifem_nnz", "ifem_set_constraints",
           "ifem_set_cell_fields", "ifem_vec_set", "ifem_vec_get", "ifem_vec_copy", "ifem_vec_zero", "ifem_vec_axpy",
           "ifem_vec_norm2", "ifem_vec_minmax", "ifem_halo_exchange", "ifem_ins_assemble", "ifem_solve",
           "ifem_rhs_norm", "ifem_ins_newton_step", "ifem_system_vmult", "ifem_uu_vmult", "ifem_precond_vmult", "ifem_export_csr",
           "ifem_get_timing", "ifem_set_profiling", "ifem_synchronize", "ifem_set_hanging_constraints", "ifem_set_ainv_kind", "ifem_set_scns_fields", "ifem_update_stress",
           "ifem_scns_assemble", "ifem_scns_solve", "ifem_scns_newton_step", "ifem_imex_assemble", "ifem_imex_solve",
           "ifem_imex_step", "ifem_set_eddy_viscosity", "ifem_default_tuning", "ifem_set_tuning", "ifem_abi_sizeof",
           "ifem_mass_vmult", "ifem_mg_attach", "ifem_mg_depth", "ifem_uu_block_diag",
           "ifem_fsi_set_solid", "ifem_fsi_update_indicator", "ifem_fsi_find_fluid_bc", "ifem_fsi_get_stress",
           "ifem_get_constraints", "ifem_fsi_fluid_at_points", "ifem_comm_stats_get", "ifem_comm_stats_level", "ifem_true_residual", "ifem_tpp_ilu_probe", "ifem_tpp_override", "ifem_scns_pc_probe", "ifem_test_restart_fits",
           "ifem_kprof_begin", "ifem_kprof_end", "ifem_kprof_family_name", "ifem_export_rows", "ifem_export_uu_pattern", "ifem_vcycle_graph_stats", "ifem_inner_restart_length"]

# ifem_abi_sizeof(which): the ctypes mirror of every struct of the header
ABI_STRUCTS = None  # filled below (needs every class defined)

_lib = None


class IfemError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ifem error {code}: {msg}")
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    L.ifem_last_error.restype = C.c_char_p
    L.ifem_n_local_dofs.restype = C.c_int64
    L.ifem_n_local_dofs.argtypes = [C.c_void_p]
    L.ifem_nnz.restype = C.c_int64
    L.ifem_nnz.argtypes = [C.c_void_p, C.c_int]
    L.ifem_default_solver_opts.argtypes = [C.POINTER(SolverOpts)]
    L.ifem_ctx_create.argtypes = [C.POINTER(MeshDesc), C.POINTER(Partition), C.c_int, C.POINTER(C.c_void_p)]
    L.ifem_ctx_destroy.argtypes = [C.c_void_p]
    L.ifem_set_constraints.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p]
    L.ifem_set_cell_fields.argtypes = [C.c_void_p, C.c_void_p]
    L.ifem_vec_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ifem_vec_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ifem_vec_copy.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ifem_vec_zero.argtypes = [C.c_void_p, C.c_int]
    L.ifem_vec_axpy.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    L.ifem_vec_norm2.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.ifem_vec_minmax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ifem_halo_exchange.argtypes = [C.c_void_p, C.c_int]
    L.ifem_ins_assemble.argtypes = [C.c_void_p, C.POINTER(InsParams), C.c_int]
    L.ifem_solve.argtypes = [C.c_void_p, C.POINTER(InsParams), C.POINTER(SolverOpts), C.c_int, C.POINTER(SolveStats)]
    L.ifem_rhs_norm.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.ifem_ins_newton_step.argtypes = [C.c_void_p, C.POINTER(InsParams), C.POINTER(SolverOpts), C.c_int, C.c_double,
                                       C.c_int, C.c_void_p]
    L.ifem_system_vmult.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ifem_uu_vmult.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.ifem_mass_vmult.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ifem_precond_vmult.argtypes = [C.c_void_p, C.POINTER(InsParams), C.POINTER(SolverOpts), C.c_int, C.c_int]
    L.ifem_export_csr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifem_export_rows.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifem_export_uu_pattern.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.ifem_vcycle_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ifem_inner_restart_length.argtypes = [C.c_void_p]
    L.ifem_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.ifem_comm_unique_id.argtypes = [C.c_void_p]
    L.ifem_local_world_create.restype = C.c_void_p
    L.ifem_local_world_create.argtypes = [C.c_int]
    L.ifem_local_world_destroy.argtypes = [C.c_void_p]
    L.ifem_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.ifem_synchronize.argtypes = [C.c_void_p]
    L.ifem_set_hanging_constraints.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifem_set_ainv_kind.argtypes = [C.c_void_p, C.c_int]
    L.ifem_set_scns_fields.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifem_set_eddy_viscosity.argtypes = [C.c_void_p, C.c_void_p]
    L.ifem_update_stress.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    L.ifem_scns_assemble.argtypes = [C.c_void_p, C.POINTER(ScnsParams), C.c_int]
    L.ifem_imex_assemble.argtypes = [C.c_void_p, C.POINTER(InsParams), C.c_int, C.c_int]
    L.ifem_imex_solve.argtypes = [C.c_void_p, C.POINTER(InsParams), C.POINTER(SolverOpts), C.c_int, C.POINTER(SolveStats)]
    L.ifem_imex_step.argtypes = [C.c_void_p, C.POINTER(InsParams), C.POINTER(SolverOpts), C.c_int, C.c_int, C.POINTER(SolveStats)]
    L.ifem_scns_solve.argtypes = [C.c_void_p, C.POINTER(SolverOpts), C.c_int, C.POINTER(SolveStats)]
    L.ifem_scns_newton_step.argtypes = [C.c_void_p, C.POINTER(ScnsParams), C.POINTER(SolverOpts), C.c_int, C.c_double,
                                        C.c_int, C.c_void_p]
    L.ifem_default_tuning.argtypes = [C.POINTER(Tuning)]
    L.ifem_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
    L.ifem_mg_attach.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(MgTransfer)]
    L.ifem_mg_depth.argtypes = [C.c_void_p]
    L.ifem_comm_stats_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ifem_comm_stats_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ifem_tpp_ilu_probe.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.ifem_tpp_override.argtypes = [C.c_void_p, C.c_void_p]
    L.ifem_scns_pc_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ifem_test_restart_fits.argtypes = [C.c_void_p, C.c_int]
    L.ifem_true_residual.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ifem_kprof_begin.argtypes = [C.c_void_p]
    L.ifem_kprof_end.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.ifem_kprof_family_name.argtypes = [C.c_int32]
    L.ifem_kprof_family_name.restype = C.c_char_p
    L.ifem_uu_block_diag.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ifem_fsi_set_solid.argtypes = [C.c_void_p, C.POINTER(FsiSolid)]
    L.ifem_fsi_update_indicator.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.ifem_fsi_find_fluid_bc.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.POINTER(FsiStats)]
    L.ifem_fsi_get_stress.argtypes = [C.c_void_p, C.c_void_p]
    L.ifem_fsi_fluid_at_points.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifem_get_constraints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ifem_abi_sizeof.argtypes = [C.c_int]
    L.ifem_abi_sizeof.restype = C.c_int64
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FsiSolid(C.Structure):  # ifem_fsi_solid
    _fields_ = [("n_vertices", C.c_int32), ("n_cells", C.c_int32), ("n_boundary_faces", C.c_int32),
                ("vertices", C.c_void_p), ("cell_vertices", C.c_void_p), ("boundary_face_vertices", C.c_void_p),
                ("velocity", C.c_void_p), ("acceleration", C.c_void_p), ("stress", C.c_void_p)]


class FsiStats(C.Structure):  # ifem_fsi_stats
    _fields_ = [("n_candidates", C.c_int64), ("n_inside", C.c_int64), ("n_lines", C.c_int64), ("n_not_found", C.c_int64)]


class CommStats(C.Structure):  # ifem_comm_stats
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32), ("n_neighbors", C.c_int32), ("transport", C.c_int32),
                ("rccl_nranks", C.c_int32), ("halo_stream", C.c_int32), ("levels", C.c_int32), ("reserved_", C.c_int32),
                ("halo_exchanges", C.c_uint64), ("allreduce_dev", C.c_uint64), ("allreduce_host", C.c_uint64),
                ("allreduce_vec", C.c_uint64)]


ABI_STRUCTS = [MeshDesc, Partition, InsParams, SolverOpts, SolveStats, ScnsParams, Timing, Tuning, MgTransfer, FsiSolid, FsiStats,
               CommStats, KprofEntry]


def export_rows(L, ctx, row0, nrows, which=0):
    """ifem_export_rows: (rowptr, col, val) of the rows [row0, row0 + nrows) of the local [u|p] system"""
    rp = np.zeros(nrows + 1, np.int64)
    rc = L.ifem_export_rows(ctx, which, row0, nrows, _ptr(rp), None, None)
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    col, val = np.zeros(rp[-1], np.int32), np.zeros(rp[-1])
    rc = L.ifem_export_rows(ctx, which, row0, nrows, _ptr(rp), _ptr(col), _ptr(val))
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    return rp, col, val


def vcycle_graph_stats(L, ctx):
    """(captures, launches) of the hipGraph of the A_uu V-cycle of this context"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    rc = L.ifem_vcycle_graph_stats(ctx, C.byref(a), C.byref(b))
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    return a.value, b.value


def export_uu_pattern(L, ctx, node0, n_nodes):
    """ifem_export_uu_pattern: (absolute block offsets, block columns in storage order) of the A_uu rows of these nodes"""
    rp = np.zeros(n_nodes + 1, np.int64)
    rc = L.ifem_export_uu_pattern(ctx, node0, n_nodes, _ptr(rp), None)
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    col = np.zeros(rp[-1] - rp[0], np.int32)
    rc = L.ifem_export_uu_pattern(ctx, node0, n_nodes, _ptr(rp), _ptr(col))
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    return rp, col


def comm_stats(L, ctx, reset=False):
    """ifem_comm_stats_get as a dict"""
    st = CommStats()
    rc = L.ifem_comm_stats_get(ctx, C.byref(st), int(reset))
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())
    return {k: int(getattr(st, k)) for k, _ in CommStats._fields_ if k != "reserved_"}


def comm_stats_levels(L, ctx):
    """ifem_comm_stats_level of every level of the chain below ctx (a list of dicts, level 0 first)"""
    out = []
    for k in range(L.ifem_mg_depth(ctx) + 1):
        st = CommStats()
        rc = L.ifem_comm_stats_level(ctx, k, C.byref(st))
        if rc < 0:
            raise IfemError(rc, L.ifem_last_error().decode())
        out.append({f: int(getattr(st, f)) for f in ("nranks", "halo_exchanges", "allreduce_dev", "allreduce_host", "allreduce_vec")})
    return out


def make_params(mu=1.0, rho=1.0, gamma=0.1, dt=1e-3, g=(0, 0, 0), neumann=None):
    p = InsParams()
    p.viscosity, p.rho, p.grad_div, p.dt = mu, rho, gamma, dt
    for i in range(3):
        p.gravity[i] = g[i] if i < len(g) else 0.0
    neumann = neumann or {}
    p.n_neumann = len(neumann)
    for k, (bid, val) in enumerate(sorted(neumann.items())):
        p.neumann_id[k] = bid
        p.neumann_p[k] = val
    return p


def make_scns_params(mu, rho, dt, solid_rho=1.0, g=(0, 0, 0), neumann=None, formulation=FORM_SCNSIM):
    p = ScnsParams()
    p.formulation = formulation
    p.viscosity, p.rho, p.dt, p.solid_rho = mu, rho, dt, solid_rho
    for i in range(3):
        p.gravity[i] = g[i] if i < len(g) else 0.0
    neumann = neumann or {}
    p.n_neumann = len(neumann)
    for k, (bid, val) in enumerate(sorted(neumann.items())):
        p.neumann_id[k] = bid
        p.neumann_p[k] = val
    return p


def _csr_pair(Pm):
    Pm = Pm.tocsr()
    Pm.sort_indices()
    Rm = Pm.T.tocsr()
    Rm.sort_indices()
    return [np.ascontiguousarray(Pm.indptr, np.int64), np.ascontiguousarray(Pm.indices, np.int32), np.ascontiguousarray(Pm.data, float),
            np.ascontiguousarray(Rm.indptr, np.int64), np.ascontiguousarray(Rm.indices, np.int32), np.ascontiguousarray(Rm.data, float)]


def mg_attach(L, fine_h, coarse_h, P_p, P_u=None, inj_u=None):
    """ifem_mg_attach from scipy.sparse prolongations (rows: owned fine nodes, columns: local coarse nodes); P_u / inj_u
    (velocity nodes, coincident fine node of every owned coarse node) are optional"""
    keep = _csr_pair(P_p)
    t = MgTransfer()
    t.n_fine_p_owned, t.n_coarse_p_local = P_p.shape
    (t.pp_ptr, t.pp_col, t.pp_w, t.rp_ptr, t.rp_col, t.rp_w) = [_ptr(a) for a in keep]
    if P_u is not None:
        ku = _csr_pair(P_u) + [np.ascontiguousarray(inj_u, np.int32)]
        t.n_fine_u_owned, t.n_coarse_u_local = P_u.shape
        (t.pu_ptr, t.pu_col, t.pu_w, t.ru_ptr, t.ru_col, t.ru_w, t.inj_u) = [_ptr(a) for a in ku]
        keep += ku
    rc = L.ifem_mg_attach(fine_h, coarse_h, C.byref(t))
    if rc < 0:
        raise IfemError(rc, L.ifem_last_error().decode())


def lattice_prolongation_1d(n_fine, n_coarse, degree):
    """1D nodal interpolation between nested uniform lattices of Q_degree nodes: n_fine cells <- n_coarse cells,
    n_fine = n_coarse (identity) or 2 n_coarse.  Returns (idx [N_f, degree + 1], w [N_f, degree + 1]) with zero-weight
    padding: fine lattice point i = sum_k w[i, k] * coarse point idx[i, k]."""
    k = degree
    Nf = k * n_fine + 1
    i = np.arange(Nf)
    idx = np.zeros((Nf, k + 1), np.int64)
    w = np.zeros((Nf, k + 1))
    if n_fine == n_coarse:
        idx[:, 0], w[:, 0] = i, 1.0
        return idx, w
    assert n_fine == 2 * n_coarse, "levels must be nested with ratio 1 or 2 per direction"
    cell = np.minimum(i // (2 * k), n_coarse - 1)        # coarse cell holding the point (2k fine steps per coarse cell)
    t = (i - 2 * k * cell) / (2.0 * k)                   # position in the coarse cell, in [0, 1]
    nodes = np.arange(k + 1) / k
    for a in range(k + 1):                               # Lagrange polynomial a on the coarse cell's nodes
        la = np.ones(Nf)
        for b in range(k + 1):
            if b != a:
                la *= (t - nodes[b]) / (nodes[a] - nodes[b])
        idx[:, a] = k * cell + a
        w[:, a] = np.where(np.abs(la) < 1e-14, 0.0, la)
    return idx, w


def box_prolongation(reps_fine, reps_coarse, degree, l2g_fine_owned, l2g_coarse_local):
    """scipy CSR prolongation between two box meshes of the same domain (nested, ratio 1 or 2 per direction) on the Q_degree
    node lattice: rows = the given fine nodes (global lattice ids, x fastest), columns = positions in l2g_coarse_local."""
    import scipy.sparse as sp
    dim = len(reps_fine)
    Nf = [degree * r + 1 for r in reps_fine]
    Nc = [degree * r + 1 for r in reps_coarse]
    g = np.asarray(l2g_fine_owned, np.int64)
    coords, rem = [], g.copy()
    for d in range(dim):
        coords.append(rem % Nf[d])
        rem //= Nf[d]
    one = [lattice_prolongation_1d(reps_fine[d], reps_coarse[d], degree) for d in range(dim)]
    n_glob_c = int(np.prod(Nc))
    g2l = -np.ones(n_glob_c, np.int64)
    g2l[np.asarray(l2g_coarse_local, np.int64)] = np.arange(len(l2g_coarse_local))
    rows, cols, vals = [], [], []
    import itertools
    for combo in itertools.product(range(degree + 1), repeat=dim):
        wgt = np.ones(len(g))
        cid = np.zeros(len(g), np.int64)
        stride = 1
        for d in range(dim):
            idx, w = one[d]
            wgt = wgt * w[coords[d], combo[d]]
            cid = cid + idx[coords[d], combo[d]] * stride
            stride *= Nc[d]
        sel = wgt != 0.0
        lc = g2l[cid[sel]]
        assert (lc >= 0).all(), "a coarse node of the interpolation stencil is not local on this rank"
        rows.append(np.nonzero(sel)[0])
        cols.append(lc)
        vals.append(wgt[sel])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(len(g), len(l2g_coarse_local)))


def box_injection(reps_fine, reps_coarse, degree, l2g_coarse_owned, l2g_fine_owned):
    """for every given coarse lattice node the position in l2g_fine_owned of the fine node at the same point"""
    dim = len(reps_fine)
    Nf = [degree * r + 1 for r in reps_fine]
    Nc = [degree * r + 1 for r in reps_coarse]
    rem = np.asarray(l2g_coarse_owned, np.int64).copy()
    gid, stride = np.zeros(len(rem), np.int64), 1
    for d in range(dim):
        ratio = reps_fine[d] // reps_coarse[d]
        gid += (rem % Nc[d]) * ratio * stride
        rem //= Nc[d]
        stride *= Nf[d]
    g2l = -np.ones(int(np.prod(Nf)), np.int64)
    g2l[np.asarray(l2g_fine_owned, np.int64)] = np.arange(len(l2g_fine_owned))
    out = g2l[gid]
    assert (out >= 0).all(), "the fine node under an owned coarse node is not owned by the same rank"
    return out.astype(np.int32)


class Context:
    """RAII wrapper of one ifem_ctx built from plain numpy mesh tables."""

    def __init__(self, dim, kv, vcoords, cell_unodes, cell_pnodes, cell_face_bid, n_unodes, n_pnodes,
                 n_unodes_owned=None, n_pnodes_owned=None, partition=None, device=0):
        self.L = load()
        self._keep = [np.ascontiguousarray(vcoords, float), np.ascontiguousarray(cell_unodes, np.int32),
                      np.ascontiguousarray(cell_pnodes, np.int32),
                      None if cell_face_bid is None else np.ascontiguousarray(cell_face_bid, np.int32)]
        m = MeshDesc(dim, kv, len(self._keep[1]), n_unodes if n_unodes_owned is None else n_unodes_owned, n_unodes,
                     n_pnodes if n_pnodes_owned is None else n_pnodes_owned, n_pnodes, *[_ptr(a) for a in self._keep])
        self.dim, self.kv = dim, kv
        self.h = C.c_void_p()
        self._chk(self.L.ifem_ctx_create(C.byref(m), None if partition is None else C.byref(partition), device, C.byref(self.h)))
        self.n_local = self.L.ifem_n_local_dofs(self.h)
        self.n_u = dim * m.n_unodes_local
        self.n_owned = dim * m.n_unodes_owned + m.n_pnodes_owned
        self.n_unodes_owned = m.n_unodes_owned
        self.opts = SolverOpts()
        self.L.ifem_default_solver_opts(C.byref(self.opts))

    @classmethod
    def borrow(cls, handle, dim, kv, n_unodes, n_pnodes):
        """the methods of this class on a context somebody else owns (the C++ host mirror's, host.FluidSolver.ctx):
        single-rank contexts; close() leaves the context alone"""
        self = cls.__new__(cls)
        self.L = load()
        self._keep = []
        self.dim, self.kv = dim, kv
        self.h = C.c_void_p(handle.value if isinstance(handle, C.c_void_p) else handle)
        self._borrowed = True
        self.n_local = self.L.ifem_n_local_dofs(self.h)
        self.n_u = dim * n_unodes
        self.n_owned = dim * n_unodes + n_pnodes
        self.n_unodes_owned = n_unodes
        self.opts = SolverOpts()
        self.L.ifem_default_solver_opts(C.byref(self.opts))
        return self

    def _chk(self, rc):
        if rc < 0:
            raise IfemError(rc, self.L.ifem_last_error().decode())
        return rc

    def close(self):
        if self.h and not getattr(self, "_borrowed", False):
            self.L.ifem_ctx_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, **kw):
        """ifem_set_tuning: e.g. set_tuning(geo_cache=0)"""
        t = Tuning()
        self.L.ifem_default_tuning(C.byref(t))
        if getattr(self, "_tuning", None) is not None:
            t = self._tuning
        for k, v in kw.items():
            if not hasattr(t, k):
                raise AttributeError(k)
            setattr(t, k, v)
        self._chk(self.L.ifem_set_tuning(self.h, C.byref(t)))
        self._tuning = t

    def set_constraints(self, which, dofs, vals=None):
        dofs = np.ascontiguousarray(dofs, np.int32)
        vals = None if vals is None else np.ascontiguousarray(vals, float)
        self._chk(self.L.ifem_set_constraints(self.h, which, len(dofs), _ptr(dofs), _ptr(vals)))

    def set_hanging_constraints(self, dof, ptr, master, weight):
        """hanging-node lines x[dof[i]] = sum_k weight[k] x[master[k]] (ifem_set_hanging_constraints); empty dof clears"""
        dof, ptr = np.ascontiguousarray(dof, np.int32), np.ascontiguousarray(ptr, np.int32)
        master, weight = np.ascontiguousarray(master, np.int32), np.ascontiguousarray(weight, float)
        self._chk(self.L.ifem_set_hanging_constraints(self.h, len(dof), _ptr(dof), _ptr(ptr), _ptr(master), _ptr(weight)))

    def set_indicator(self, ind):
        ind = None if ind is None else np.ascontiguousarray(ind, np.int32)
        self._chk(self.L.ifem_set_cell_fields(self.h, _ptr(ind)))

    # ---- fluid-side inputs of MPI::FSI on the device (mpi_fsi.cpp:291-663)
    def fsi_set_solid(self, vertices, cells, bfaces=None, velocity=None, acceleration=None, stress=None):
        keep = [np.ascontiguousarray(vertices, float), np.ascontiguousarray(cells, np.int32),
                None if bfaces is None else np.ascontiguousarray(bfaces, np.int32),
                None if velocity is None else np.ascontiguousarray(velocity, float),
                None if acceleration is None else np.ascontiguousarray(acceleration, float),
                None if stress is None else np.ascontiguousarray(stress, float)]
        s = FsiSolid(len(keep[0]), len(keep[1]), 0 if keep[2] is None else len(keep[2]), *[_ptr(a) for a in keep])
        self._chk(self.L.ifem_fsi_set_solid(self.h, C.byref(s)))

    def fsi_update_indicator(self, n_cells):
        out = np.zeros(n_cells, np.int32)
        cnt = C.c_int64()
        self._chk(self.L.ifem_fsi_update_indicator(self.h, _ptr(out), C.byref(cnt)))
        return out, cnt.value

    def fsi_find_fluid_bc(self, dt, use_dirichlet_bc, cell_order=None):
        order = None if cell_order is None else np.ascontiguousarray(cell_order, np.int32)
        st = FsiStats()
        self._chk(self.L.ifem_fsi_find_fluid_bc(self.h, dt, int(use_dirichlet_bc), _ptr(order), C.byref(st)))
        return st

    def fsi_fluid_at_points(self, points, with_stress=True):
        """(u, p) [n, dim+1], viscous stress [n, dim, dim] and the local cell [n] at `points` (find_solid_bc, mpi_fsi.cpp:727-760)"""
        pts = np.ascontiguousarray(points, float).reshape(-1, self.dim)
        n = len(pts)
        vals, cell = np.zeros((n, self.dim + 1)), np.zeros(n, np.int32)
        st = np.zeros((n, self.dim, self.dim)) if with_stress else None
        self._chk(self.L.ifem_fsi_fluid_at_points(self.h, n, _ptr(pts), _ptr(vals), _ptr(st), _ptr(cell)))
        return vals, st, cell

    def fsi_get_stress(self):
        out = np.zeros((self.dim * (self.dim + 1) // 2, self.n_u // self.dim))
        self._chk(self.L.ifem_fsi_get_stress(self.h, _ptr(out)))
        return out

    def get_constraints(self, which):
        flags, vals = np.zeros(self.n_local, np.uint8), np.zeros(self.n_local)
        self._chk(self.L.ifem_get_constraints(self.h, which, _ptr(flags), _ptr(vals)))
        return flags, vals

    def _len(self, vec):
        return self.n_local if vec in (VEC_PRESENT, VEC_EVAL, VEC_FSI_ACC, VEC_INCREMENT) else self.n_owned

    def vec_set(self, vec, x):
        x = np.ascontiguousarray(x, float)
        assert x.size == self._len(vec)
        self._chk(self.L.ifem_vec_set(self.h, vec, _ptr(x)))

    def vec_get(self, vec):
        x = np.zeros(self._len(vec))
        self._chk(self.L.ifem_vec_get(self.h, vec, _ptr(x)))
        return x

    def assemble(self, params, use_nonzero):
        self._chk(self.L.ifem_ins_assemble(self.h, C.byref(params), int(use_nonzero)))

    def solve(self, params, use_nonzero):
        st = SolveStats()
        self._chk(self.L.ifem_solve(self.h, C.byref(params), C.byref(self.opts), int(use_nonzero), C.byref(st)))
        return st

    def set_scns_fields(self, sigma_pml=None, body_force=None, fsi_stress=None):
        keep = [None if a is None else np.ascontiguousarray(a, float) for a in (sigma_pml, body_force, fsi_stress)]
        self._chk(self.L.ifem_set_scns_fields(self.h, *[_ptr(a) for a in keep]))

    def update_stress(self, mu):
        out = np.zeros((self.dim, self.dim, self.n_u // self.dim))
        self._chk(self.L.ifem_update_stress(self.h, mu, _ptr(out)))
        return out

    def imex_assemble(self, params, use_nonzero, assemble_system=True):
        self._chk(self.L.ifem_imex_assemble(self.h, C.byref(params), int(use_nonzero), int(assemble_system)))

    def imex_solve(self, params, use_nonzero):
        st = SolveStats()
        self._chk(self.L.ifem_imex_solve(self.h, C.byref(params), C.byref(self.opts), int(use_nonzero), C.byref(st)))
        return st

    def imex_step(self, params, apply_nonzero, assemble_system=True):
        st = SolveStats()
        self._chk(self.L.ifem_imex_step(self.h, C.byref(params), C.byref(self.opts), int(apply_nonzero), int(assemble_system), C.byref(st)))
        return st

    def set_eddy_viscosity(self, nodal):
        a = None if nodal is None else np.ascontiguousarray(nodal, float)
        self._chk(self.L.ifem_set_eddy_viscosity(self.h, None if a is None else a.ctypes.data_as(C.c_void_p)))

    def scns_assemble(self, params, use_nonzero):
        self._chk(self.L.ifem_scns_assemble(self.h, C.byref(params), int(use_nonzero)))

    def scns_solve(self, use_nonzero):
        st = SolveStats()
        self._chk(self.L.ifem_scns_solve(self.h, C.byref(self.opts), int(use_nonzero), C.byref(st)))
        return st

    def scns_newton_step(self, params, apply_nonzero, tol=1e-6, maxit=8):
        log = np.zeros((maxit + 1, 4))
        rc = self._chk(self.L.ifem_scns_newton_step(self.h, C.byref(params), C.byref(self.opts), int(apply_nonzero), tol,
                                                    maxit, _ptr(log)))
        return rc, log[:rc]

    def rhs_norm(self):
        v = C.c_double()
        self._chk(self.L.ifem_rhs_norm(self.h, C.byref(v)))
        return v.value

    def newton_step(self, params, apply_nonzero, tol=1e-6, maxit=8):
        log = np.zeros((maxit + 1, 4))
        rc = self._chk(self.L.ifem_ins_newton_step(self.h, C.byref(params), C.byref(self.opts), int(apply_nonzero), tol,
                                                   maxit, _ptr(log)))
        return rc, log[:rc]

    def minmax(self, vec, block):
        a, b = C.c_double(), C.c_double()
        self._chk(self.L.ifem_vec_minmax(self.h, vec, block, C.byref(a), C.byref(b)))
        return a.value, b.value

    def system_vmult(self, x):
        self.vec_set(VEC_TMP, x)
        self._chk(self.L.ifem_system_vmult(self.h, VEC_UPDATE, VEC_TMP))
        return self.vec_get(VEC_UPDATE)

    def uu_vmult(self, x, variant=0):
        """y_u = A_uu x_u; variant 0 stored fp64, 1 fp32 copy, 3 matrix-free"""
        self.vec_set(VEC_TMP, x)
        self._chk(self.L.ifem_uu_vmult(self.h, VEC_UPDATE, VEC_TMP, variant))
        return self.vec_get(VEC_UPDATE)

    def uu_block_diag(self, which):
        """inverse diagonal node blocks of A_uu [n_unodes_owned, dim, dim]: 0 from the assembled matrix, 1 matrix-free"""
        out = np.zeros((self.n_unodes_owned, self.dim, self.dim))
        self._chk(self.L.ifem_uu_block_diag(self.h, which, _ptr(out)))
        return out

    def mass_vmult(self, x):
        """[diag(M_u) x_u ; M_p x_p]"""
        self.vec_set(VEC_TMP, x)
        self._chk(self.L.ifem_mass_vmult(self.h, VEC_UPDATE, VEC_TMP))
        return self.vec_get(VEC_UPDATE)

    def precond_vmult(self, params, x):
        self.vec_set(VEC_TMP, x)
        self._chk(self.L.ifem_precond_vmult(self.h, C.byref(params), C.byref(self.opts), VEC_UPDATE, VEC_TMP))
        return self.vec_get(VEC_UPDATE)

    def export_csr(self, which=0):
        import scipy.sparse as sp
        n = self.n_owned
        rp = np.zeros(n + 1, np.int64)
        self._chk(self.L.ifem_export_csr(self.h, which, _ptr(rp), None, None))
        col = np.zeros(rp[-1], np.int32)
        val = np.zeros(rp[-1])
        self._chk(self.L.ifem_export_csr(self.h, which, _ptr(rp), _ptr(col), _ptr(val)))
        return sp.csr_matrix((val, col, rp), shape=(n, self.n_local))

    def export_rows(self, row0, nrows, which=0):
        return export_rows(self.L, self.h, row0, nrows, which)

    def timing(self):
        t = Timing()
        self._chk(self.L.ifem_get_timing(self.h, C.byref(t)))
        return t
