"""ctypes binding of oracle/liboracle.so (test infrastructure: the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")


class _Mesh(C.Structure):
    _fields_ = [("dim", C.c_int32), ("kv", C.c_int32), ("n_cells", C.c_int32), ("n_unodes", C.c_int32),
                ("n_pnodes", C.c_int32), ("vcoords", C.c_void_p), ("cell_unodes", C.c_void_p),
                ("cell_pnodes", C.c_void_p), ("cell_face_bid", C.c_void_p), ("indicator", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("mu", C.c_double), ("rho", C.c_double), ("gamma", C.c_double), ("dt", C.c_double),
                ("g", C.c_double * 3), ("n_neumann", C.c_int32), ("neumann_id", C.c_int32 * 8),
                ("neumann_p", C.c_double * 8)]


class Opts(C.Structure):
    _fields_ = [("fgmres_restart", C.c_int32), ("fgmres_maxit", C.c_int32), ("fgmres_rel", C.c_double),
                ("fgmres_abs", C.c_double), ("inner_restart", C.c_int32), ("inner_maxit", C.c_int32),
                ("inner_rel", C.c_double), ("n_threads", C.c_int32)]


class ScnsParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("rho", C.c_double), ("dt", C.c_double), ("solid_rho", C.c_double),
                ("g", C.c_double * 3), ("n_neumann", C.c_int32), ("neumann_id", C.c_int32 * 8),
                ("neumann_p", C.c_double * 8), ("stress", C.c_void_p), ("fsi_stress", C.c_void_p),
                ("sigma_pml", C.c_void_p), ("body_force", C.c_void_p), ("eddy_viscosity", C.c_void_p),
                ("formulation", C.c_int32)]


class _Solid(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_vertices", C.c_int32), ("n_cells", C.c_int32), ("n_bfaces", C.c_int32),
                ("vertices", C.c_void_p), ("cell_vertices", C.c_void_p), ("bface_vertices", C.c_void_p),
                ("velocity", C.c_void_p), ("acceleration", C.c_void_p), ("stress", C.c_void_p)]


FULL_SOLVE = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                         C.POINTER(C.c_double), C.POINTER(C.c_double))

AINV = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

_lib = None


def lib(native=False):
    global _lib
    name = "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(_ODIR, name)
    if native:  # host-tuned build for the CPU baseline leg: always rebuilt on the box that runs it (-march=native)
        subprocess.check_call(["make", "-C", _ODIR, "native"], stdout=subprocess.DEVNULL)
        return _bind(C.CDLL(path))
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", _ODIR], stdout=subprocess.DEVNULL)
    if _lib is None:
        _lib = _bind(C.CDLL(path))
    return _lib


def _bind(L):
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(_Mesh)]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_n_dofs.argtypes = [C.c_void_p]
    L.orc_n_u.argtypes = [C.c_void_p]
    L.orc_set_constraints.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.orc_default_opts.argtypes = [C.POINTER(Opts)]
    for f, t in (("orc_rowptr", C.c_int64), ("orc_col", C.c_int32), ("orc_A", C.c_double), ("orc_M", C.c_double),
                 ("orc_rhs", C.c_double)):
        getattr(L, f).restype = C.POINTER(t)
        getattr(L, f).argtypes = [C.c_void_p]
    L.orc_ins_assemble.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_ins_assemble_subdomains.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int32, C.c_int32]
    L.orc_ins_assemble_affine_dense.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_ins_cell.argtypes = [C.POINTER(_Mesh), C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_ins_solve.restype = C.c_int32
    L.orc_ins_solve.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.POINTER(Opts), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.orc_imex_run_one_step.restype = C.c_int32
    L.orc_ins_run_one_step.restype = C.c_int32
    L.orc_ins_run_one_step.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_double, C.c_int32,
                                       C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_precond_vmult.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Opts), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    L.orc_scns_assemble.argtypes = [C.c_void_p, C.POINTER(ScnsParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_scns_cell.argtypes = [C.POINTER(_Mesh), C.POINTER(ScnsParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]
    L.orc_scns_pc_probe.restype = C.c_int32
    L.orc_scns_pc_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.orc_scns_solve.restype = C.c_int32
    L.orc_scns_solve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.orc_scns_run_one_step.restype = C.c_int32
    L.orc_scns_run_one_step.argtypes = [C.c_void_p, C.POINTER(ScnsParams), C.c_int32, C.c_double, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_update_stress.argtypes = [C.POINTER(_Mesh), C.c_double, C.c_void_p, C.c_void_p]
    L.orc_fsi_solid_box.argtypes = [C.POINTER(_Solid), C.c_void_p]
    L.orc_fsi_point_in_solid.restype = C.c_int32
    L.orc_fsi_point_in_solid.argtypes = [C.POINTER(_Solid), C.c_void_p, C.c_void_p]
    L.orc_fsi_real_to_unit.restype = C.c_int32
    L.orc_fsi_real_to_unit.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_fsi_locate.restype = C.c_int32
    L.orc_fsi_locate.argtypes = [C.POINTER(_Solid), C.c_void_p, C.c_void_p]
    L.orc_fsi_update_indicator.argtypes = [C.POINTER(_Mesh), C.POINTER(_Solid), C.c_void_p]
    L.orc_fsi_find_fluid_bc.restype = C.c_int32
    L.orc_fsi_find_fluid_bc.argtypes = [C.POINTER(_Mesh), C.POINTER(_Solid), C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_fsi_fluid_at_points.argtypes = [C.POINTER(_Mesh), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
    L.orc_fe_tables.restype = C.c_int32
    L.orc_fe_tables.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def make_params(mu=1.0, rho=1.0, gamma=0.1, dt=1e-3, g=(0, 0, 0), neumann=None):
    p = Params()
    p.mu, p.rho, p.gamma, p.dt = mu, rho, gamma, dt
    for i in range(3):
        p.g[i] = g[i] if i < len(g) else 0.0
    neumann = neumann or {}
    p.n_neumann = len(neumann)
    for k, (bid, val) in enumerate(sorted(neumann.items())):
        p.neumann_id[k] = bid
        p.neumann_p[k] = val
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_scns_params(mu, rho, dt, solid_rho=1.0, g=(0, 0, 0), neumann=None, stress=None, fsi_stress=None,
                     sigma_pml=None, body_force=None, formulation=0, eddy_viscosity=None):
    """formulation 0: SCnsIM, 1: SUPGInsIM (mpi_insim_supg.cpp)"""
    p = ScnsParams()
    p.formulation = formulation
    p.mu, p.rho, p.dt, p.solid_rho = mu, rho, dt, solid_rho
    for i in range(3):
        p.g[i] = g[i] if i < len(g) else 0.0
    neumann = neumann or {}
    p.n_neumann = len(neumann)
    for k, (bid, val) in enumerate(sorted(neumann.items())):
        p.neumann_id[k] = bid
        p.neumann_p[k] = val
    p._keep = [None if a is None else np.ascontiguousarray(a, float) for a in (stress, fsi_stress, sigma_pml, body_force, eddy_viscosity)]
    p.stress, p.fsi_stress, p.sigma_pml, p.body_force, p.eddy_viscosity = [_ptr(a) for a in p._keep]
    return p


class SpluFullSolve:
    """Exact solve of the whole Newton system (stands in for FGMRES + the Euclid-ILU block preconditioner of
    SUPGFluidSolver::solve, mpi_supg_solver.cpp:297-328: the preconditioner only changes iteration counts)."""

    def __init__(self):
        self.cb = FULL_SOLVE(self._call)

    def _call(self, user, n, rowptr, col, val, rhs, x):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        rp = np.ctypeslib.as_array(rowptr, (n + 1,))
        nnz = int(rp[-1])
        A = sp.csr_matrix((np.ctypeslib.as_array(val, (nnz,)).copy(), np.ctypeslib.as_array(col, (nnz,)).copy(), rp.copy()),
                          shape=(n, n))
        np.ctypeslib.as_array(x, (n,))[:] = spl.spsolve(A.tocsc(), np.ctypeslib.as_array(rhs, (n,)))


class SpluAinv:
    """Exact A_uu^-1 through scipy splu: stands in for MUMPS (mpi_insim.cpp:124-127) in parity runs."""

    def __init__(self):
        self.lu = None
        self.cb = AINV(self._call)
        self.n_factor = 0

    def _call(self, user, refresh, n, rowptr, col, val, x, y):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        if refresh or self.lu is None:
            rp = np.ctypeslib.as_array(rowptr, (n + 1,))
            nnz = int(rp[-1])
            A = sp.csr_matrix((np.ctypeslib.as_array(val, (nnz,)).copy(), np.ctypeslib.as_array(col, (nnz,)).copy(),
                               rp.copy()), shape=(n, n))
            self.lu = spl.splu(A.tocsc())
            self.n_factor += 1
        xv = np.ctypeslib.as_array(x, (n,))
        yv = np.ctypeslib.as_array(y, (n,))
        yv[:] = self.lu.solve(xv)


class System:
    """Thin owner of an orc_system plus the numpy arrays it borrows."""

    def __init__(self, mesh, native=False):
        self.L = lib(native)
        self.mesh = mesh
        self._keep = [np.ascontiguousarray(mesh.vcoords, float), np.ascontiguousarray(mesh.cell_unodes, np.int32),
                      np.ascontiguousarray(mesh.cell_pnodes, np.int32),
                      np.ascontiguousarray(mesh.cell_face_bid, np.int32),
                      None if getattr(mesh, "indicator", None) is None else np.ascontiguousarray(mesh.indicator, np.int32)]
        m = _Mesh(mesh.dim, mesh.kv, mesh.n_cells, mesh.n_unodes, mesh.n_pnodes, *[_ptr(a) for a in self._keep])
        self.cmesh = m
        self.h = C.c_void_p(self.L.orc_create(C.byref(m)))
        self.n = self.L.orc_n_dofs(self.h)
        self.n_u = self.L.orc_n_u(self.h)
        self.opts = Opts()
        self.L.orc_default_opts(C.byref(self.opts))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_constraints(self, which, dofs, vals=None):
        dofs = np.ascontiguousarray(dofs, np.int32)
        vals = None if vals is None else np.ascontiguousarray(vals, float)
        self.L.orc_set_constraints(self.h, which, len(dofs), _ptr(dofs), _ptr(vals))

    def assemble(self, params, use_nonzero, evalp, present, fsi_acc=None):
        self.L.orc_ins_assemble(self.h, C.byref(params), int(use_nonzero), _ptr(evalp), _ptr(present), _ptr(fsi_acc))

    def assemble_subdomains(self, params, use_nonzero, evalp, present, cell_part, n_parts, n_threads=0, fsi_acc=None):
        """orc_ins_assemble_subdomains: owner-computes rows over the subdomains cell_part names, no atomics"""
        cp = np.ascontiguousarray(cell_part, np.int32)
        self.L.orc_ins_assemble_subdomains(self.h, C.byref(params), int(use_nonzero), _ptr(evalp), _ptr(present), _ptr(fsi_acc),
                                           _ptr(cp), int(n_parts), int(n_threads))

    def assemble_affine_dense(self, params, use_nonzero, evalp, present, mesh, fsi_acc=None):
        """assembly through constraints that also hold the hanging lines of `mesh` (tests/hangmesh.py): dense (A, rhs)"""
        A, b = np.zeros((self.n, self.n)), np.zeros(self.n)
        self.L.orc_ins_assemble_affine_dense(self.h, C.byref(params), int(use_nonzero), _ptr(evalp), _ptr(present), _ptr(fsi_acc),
                                             len(mesh.hang_dof), _ptr(mesh.hang_dof), _ptr(mesh.hang_ptr), _ptr(mesh.hang_master),
                                             _ptr(mesh.hang_weight), _ptr(A), _ptr(b))
        return A, b

    def csr(self, which="A"):
        import scipy.sparse as sp
        rp = np.ctypeslib.as_array(self.L.orc_rowptr(self.h), (self.n + 1,)).copy()
        nnz = int(rp[-1])
        col = np.ctypeslib.as_array(self.L.orc_col(self.h), (nnz,)).copy()
        val = np.ctypeslib.as_array(getattr(self.L, "orc_" + which)(self.h), (nnz,)).copy()
        return sp.csr_matrix((val, col, rp), shape=(self.n, self.n))

    def rhs(self):
        return np.ctypeslib.as_array(self.L.orc_rhs(self.h), (self.n,)).copy()

    def cell(self, params, cell, evalp, present, fsi_acc=None):
        nd = self.mesh.dim * self.mesh.cell_unodes.shape[1] + self.mesh.cell_pnodes.shape[1]
        Ke, Me, fe = np.zeros((nd, nd)), np.zeros((nd, nd)), np.zeros(nd)
        self.L.orc_ins_cell(C.byref(self.cmesh), C.byref(params), cell, _ptr(evalp), _ptr(present), _ptr(fsi_acc),
                            _ptr(Ke), _ptr(Me), _ptr(fe))
        return Ke, Me, fe

    def solve(self, params, use_nonzero, ainv=None):
        upd = np.zeros(self.n)
        it, res = C.c_int32(0), C.c_double(0)
        cb = C.cast(ainv.cb, C.c_void_p) if ainv is not None else None
        rc = self.L.orc_ins_solve(self.h, C.byref(params), int(use_nonzero), C.byref(self.opts), cb, None, _ptr(upd),
                                  C.byref(it), C.byref(res))
        return rc, upd, it.value, res.value

    def run_one_step(self, params, apply_nonzero, present, newton_tol=1e-6, newton_maxit=8, ainv=None, fsi_acc=None):
        log = np.zeros((newton_maxit + 1, 4))
        cb = C.cast(ainv.cb, C.c_void_p) if ainv is not None else None
        rc = self.L.orc_ins_run_one_step(self.h, C.byref(params), int(apply_nonzero), newton_tol, newton_maxit,
                                         C.byref(self.opts), cb, None, _ptr(present), _ptr(fsi_acc), _ptr(log))
        return rc, log[:max(rc, 0)]

    def imex_assemble(self, params, use_nonzero, assemble_system, present, fsi_acc=None):
        self.L.orc_imex_assemble(self.h, C.byref(params), int(use_nonzero), int(assemble_system), _ptr(present), _ptr(fsi_acc))

    def imex_run_one_step(self, params, apply_nonzero, assemble_system, present, ainv=None, fsi_acc=None):
        """InsIMEX::run_one_step on `present` (updated in place); returns (rc, fgmres iterations, residual)"""
        cb = C.cast(ainv.cb, C.c_void_p) if ainv is not None else None
        it, res = C.c_int32(), C.c_double()
        rc = self.L.orc_imex_run_one_step(self.h, C.byref(params), int(apply_nonzero), int(assemble_system),
                                          C.byref(self.opts), cb, None, _ptr(present), _ptr(fsi_acc), C.byref(it), C.byref(res))
        return rc, it.value, res.value

    def scns_assemble(self, params, use_nonzero, evalp, present, fsi_acc=None):
        self.L.orc_scns_assemble(self.h, C.byref(params), int(use_nonzero), _ptr(evalp), _ptr(present), _ptr(fsi_acc))

    def scns_cell(self, params, cell, evalp, present, fsi_acc=None):
        nd = self.mesh.dim * self.mesh.cell_unodes.shape[1] + self.mesh.cell_pnodes.shape[1]
        Ke, fe = np.zeros((nd, nd)), np.zeros(nd)
        self.L.orc_scns_cell(C.byref(self.cmesh), C.byref(params), cell, _ptr(evalp), _ptr(present), _ptr(fsi_acc),
                             _ptr(Ke), _ptr(fe))
        return Ke, fe

    def scns_run_one_step(self, params, apply_nonzero, present, newton_tol=1e-6, newton_maxit=8, solver=None, fsi_acc=None):
        solver = solver or SpluFullSolve()
        log = np.zeros((newton_maxit + 1, 4))
        rc = self.L.orc_scns_run_one_step(self.h, C.byref(params), int(apply_nonzero), newton_tol, newton_maxit,
                                          C.cast(solver.cb, C.c_void_p), None, _ptr(present), _ptr(fsi_acc), _ptr(log))
        return rc, log[:max(rc, 0)]

    def scns_solve(self, use_nonzero, fgmres_restart=30, perm_v=None, perm_p=None):
        """SUPGFluidSolver::solve with the reference's BlockIncompSchurPreconditioner (ILU(0)(A_vv), operator T_pp, ILU(0)(B2pp)) on
        the last scns_assemble; returns (rc, update, (FGMRES its, Tpp_itr, preconditioner applications, Pvv applications), residual)"""
        upd, counts, res = np.zeros(self.n), np.zeros(4, np.int64), C.c_double()
        pv = None if perm_v is None else np.ascontiguousarray(perm_v, np.int32)
        pp = None if perm_p is None else np.ascontiguousarray(perm_p, np.int32)
        rc = self.L.orc_scns_solve(self.h, int(use_nonzero), fgmres_restart, _ptr(pv), _ptr(pp), _ptr(upd), _ptr(counts), C.byref(res))
        return rc, upd, tuple(int(v) for v in counts), res.value

    def scns_pc_probe(self, which, x):
        """pieces of the reference-structure SUPG preconditioner: 0 Pvv^-1 x, 1 B2pp_inverse x, 2 B2pp x, 3 T_pp x"""
        x = np.ascontiguousarray(x, float)
        y = np.zeros(len(x))
        assert self.L.orc_scns_pc_probe(self.h, which, _ptr(x), _ptr(y)) == 0
        return y

    def update_stress(self, mu, present):
        out = np.zeros((self.mesh.dim, self.mesh.dim, self.mesh.n_unodes))
        self.L.orc_update_stress(C.byref(self.cmesh), mu, _ptr(present), _ptr(out))
        return out

    def precond(self, params, v, ainv=None):
        z = np.zeros(self.n)
        cb = C.cast(ainv.cb, C.c_void_p) if ainv is not None else None
        v = np.ascontiguousarray(v, float)
        self.L.orc_precond_vmult(self.h, C.byref(params), C.byref(self.opts), cb, None, _ptr(v), _ptr(z))
        return z


class FsiSolid:
    """the solid as MPI::FSI sees it on every rank (tests/solidmesh.py objects): oracle_fsi.c"""

    def __init__(self, solid):
        self.L = lib()
        dim = solid.dim
        self._keep = [np.ascontiguousarray(solid.vertices, float), np.ascontiguousarray(solid.cells, np.int32),
                      None if dim == 3 else np.ascontiguousarray(solid.bfaces, np.int32),
                      np.ascontiguousarray(solid.velocity, float), np.ascontiguousarray(solid.acceleration, float),
                      None if solid.stress is None else np.ascontiguousarray(solid.stress, float)]
        self.dim = dim
        self.c = _Solid(dim, len(self._keep[0]), len(self._keep[1]), 0 if dim == 3 else len(self._keep[2]),
                        *[_ptr(a) for a in self._keep])

    def box(self):
        b = np.zeros(2 * self.dim)
        self.L.orc_fsi_solid_box(C.byref(self.c), _ptr(b))
        return b

    def point_in_solid(self, pts):
        pts = np.ascontiguousarray(pts, float).reshape(-1, self.dim)
        b = self.box()
        return np.array([self.L.orc_fsi_point_in_solid(C.byref(self.c), _ptr(b), _ptr(p)) for p in pts], bool)

    def locate(self, pts):
        pts = np.ascontiguousarray(pts, float).reshape(-1, self.dim)
        cells, xi = np.zeros(len(pts), np.int32), np.zeros((len(pts), self.dim))
        for i, p in enumerate(pts):
            cells[i] = self.L.orc_fsi_locate(C.byref(self.c), _ptr(p), _ptr(xi[i]))
        return cells, xi


def _cmesh(mesh, indicator=None):
    keep = [np.ascontiguousarray(mesh.vcoords, float), np.ascontiguousarray(mesh.cell_unodes, np.int32),
            np.ascontiguousarray(mesh.cell_pnodes, np.int32), np.ascontiguousarray(mesh.cell_face_bid, np.int32),
            None if indicator is None else np.ascontiguousarray(indicator, np.int32)]
    m = _Mesh(mesh.dim, mesh.kv, mesh.n_cells, mesh.n_unodes, mesh.n_pnodes, *[_ptr(a) for a in keep])
    m._keep = keep
    return m


def fsi_update_indicator(mesh, solid):
    """FSI::update_indicator (mpi_fsi.cpp:291-319) -> int32 [n_cells]"""
    S = solid if isinstance(solid, FsiSolid) else FsiSolid(solid)
    m = _cmesh(mesh)
    out = np.zeros(mesh.n_cells, np.int32)
    S.L.orc_fsi_update_indicator(C.byref(m), C.byref(S.c), _ptr(out))
    return out


def fsi_find_fluid_bc(mesh, solid, indicator, dt, use_dirichlet_bc, present, fluid_stress, fsi_stress):
    """FSI::find_fluid_bc (mpi_fsi.cpp:323-663) on one rank: fsi_stress [ncomp][n_unodes] is updated in place; returns
    (fsi_acc [n_dofs], line_flag, line_val [dim*n_unodes] before the left_object_wins merge, n_not_found)"""
    S = solid if isinstance(solid, FsiSolid) else FsiSolid(solid)
    m = _cmesh(mesh, indicator)
    n_u = mesh.dim * mesh.n_unodes
    acc = np.zeros(n_u + mesh.n_pnodes)
    flag, val = np.zeros(n_u, np.int32), np.zeros(n_u)
    present = np.ascontiguousarray(present, float)
    fl = None if fluid_stress is None else np.ascontiguousarray(fluid_stress, float)
    assert fsi_stress is None or (fsi_stress.flags.c_contiguous and fsi_stress.dtype == np.float64)
    nf = S.L.orc_fsi_find_fluid_bc(C.byref(m), C.byref(S.c), dt, int(use_dirichlet_bc), _ptr(present), _ptr(fl), _ptr(fsi_stress),
                                   _ptr(acc), _ptr(flag), _ptr(val))
    return acc, flag, val, nf


def fsi_fluid_at_points(mesh, present, fluid_stress, points):
    """(u, p) [n, dim+1], viscous stress [n, dim, dim] and the fluid cell [n] at `points` (find_solid_bc, mpi_fsi.cpp:727-760)"""
    L = lib()
    m = _cmesh(mesh)
    pts = np.ascontiguousarray(points, float).reshape(-1, mesh.dim)
    n = len(pts)
    vals, st, cell = np.zeros((n, mesh.dim + 1)), np.zeros((n, mesh.dim, mesh.dim)), np.zeros(n, np.int32)
    present = np.ascontiguousarray(present, float)
    fl = None if fluid_stress is None else np.ascontiguousarray(fluid_stress, float)
    L.orc_fsi_fluid_at_points(C.byref(m), _ptr(present), _ptr(fl), n, _ptr(pts), _ptr(vals), _ptr(st), _ptr(cell))
    return vals, st, cell
