"""math_audio_amd — ctypes view of libmathaudio_hip.so (the C-ABI in include/mathaudio_hip.h).

This package is plumbing: it loads the in-tree shared library (built by
`__graft_entry__.build()` / `make -C math_audio_amd/csrc`) and exposes the C entry points
with NumPy / raw-device-pointer arguments for the tests and bench.py. The compiled host
mirror of the reference's Rust API lives in math_audio_amd/host/*.hpp (C++), because the
reference is compiled code; see INTEGRATION.md.

There is no CPU fallback: a missing library raises at import of `lib()`, and every compute
call returns MA_ERR_NO_DEVICE on a host without a gfx950 GPU.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MA_LIB_PATH") or os.path.join(_HERE, "lib", "libmathaudio_hip.so")     # MA_LIB_PATH: a diagnostic build (tools/)

MA_OK, MA_ERR_SINGULAR, MA_ERR_DIM, MA_ERR_INVALID, MA_ERR_UNSUPPORTED, MA_ERR_HIP, MA_ERR_NO_DEVICE, MA_ERR_NOMEM, MA_ERR_RETRY = range(9)
_STATUS_NAMES = ["MA_OK", "MA_ERR_SINGULAR", "MA_ERR_DIM", "MA_ERR_INVALID", "MA_ERR_UNSUPPORTED", "MA_ERR_HIP",
                 "MA_ERR_NO_DEVICE", "MA_ERR_NOMEM", "MA_ERR_RETRY"]


class MaError(RuntimeError):
    def __init__(self, status, text):
        self.status = status
        name = _STATUS_NAMES[status] if 0 <= status < len(_STATUS_NAMES) else str(status)
        super().__init__("%s: %s" % (name, text))


class ma_c64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class ma_mesh_t(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("nodes", C.c_void_p), ("n_elem", C.c_int32), ("conn", C.c_void_p),
                ("center", C.c_void_p), ("normal", C.c_void_p), ("area", C.c_void_p), ("dof", C.c_void_p),
                ("bc_type", C.c_void_p), ("bc_values", C.c_void_p), ("bc_len", C.c_void_p), ("is_eval", C.c_void_p)]


class ma_clusters_t(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("center", C.c_void_p), ("elem_ptr", C.c_void_p), ("elem_idx", C.c_void_p),
                ("near_ptr", C.c_void_p), ("near_idx", C.c_void_p), ("far_ptr", C.c_void_p), ("far_idx", C.c_void_p)]


class ma_physics_t(C.Structure):
    _fields_ = [("wave_number", C.c_double), ("harmonic_factor", C.c_double), ("tau", C.c_double), ("gamma", C.c_double)]


_lib = None


def lib():
    """Load libmathaudio_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ma_last_error_string.restype = C.c_char_p
        L.ma_version.restype = C.c_char_p
        vp, dbl, i32, i64 = C.c_void_p, C.c_double, C.c_int32, C.c_int64
        P = C.POINTER
        sig = {
            "ma_device_count": [P(C.c_int)],
            "ma_bem_assemble_tbem": [P(ma_mesh_t), P(ma_physics_t), dbl, dbl, vp, vp],
            "ma_bem_plan_create": [P(ma_mesh_t), C.c_int, P(vp)],
            "ma_bem_plan_destroy": [vp],
            "ma_bem_plan_num_dofs": [vp, P(i32)],
            "ma_bem_plan_num_near_pairs": [vp, P(i64)],
            "ma_bem_plan_assemble_dev": [vp, P(ma_physics_t), dbl, dbl, vp, vp, vp],
            "ma_bem_plan_incident_rhs_dev": [vp, P(ma_physics_t), dbl, dbl, C.c_int, vp, dbl, dbl, C.c_int, vp, vp],
            "ma_bem_incident_rhs": [C.c_int, vp, vp, P(ma_physics_t), dbl, dbl, C.c_int, vp, dbl, dbl, vp],
            "ma_bem_plan_probe_pairs": [vp, P(ma_physics_t), i64, vp, vp],
            "ma_bem_plan_probe_self": [vp, P(ma_physics_t), vp],
            "ma_bem_plan_get_near_pairs": [vp, vp],
            "ma_bem_plan_set_timing": [vp, C.c_int],
            "ma_bem_plan_last_timing": [vp, vp],
            "ma_zgesv": [i32, vp, vp, vp],
            "ma_lu_plan_create": [i32, C.c_int, P(vp)],
            "ma_lu_plan_create_pivoting": [i32, C.c_int, i32, P(vp)],
            "ma_lu_plan_pivoting": [vp, P(i32)],
            "ma_lu_plan_speculation_stats": [vp, P(i64), P(i64), P(i64)],
            "ma_lu_plan_set_speculation": [vp, i32],
            "ma_lu_plan_speculation": [vp, P(i32)],
            "ma_zgesv_pivoting": [i32, vp, vp, vp, i32],
            "ma_bem_sweep_create_pivoting": [vp, i32, i32, i32, P(vp)],
            "ma_lu_plan_destroy": [vp],
            "ma_lu_plan_factor_solve_dev": [vp, vp, vp, i32, vp],
            "ma_lu_plan_solve_dev": [vp, vp, vp, i32, vp],
            "ma_lu_solve": [i32, vp, vp, vp],
            "ma_bem_solve_sweep": [vp, i32, vp, dbl, dbl, dbl, dbl, C.c_int, vp, dbl, dbl, i32, vp, vp],
            "ma_bem_sweep_create": [vp, i32, i32, P(vp)],
            "ma_bem_sweep_destroy": [vp],
            "ma_device_copy": [vp, vp, i64, vp],
            "ma_bem_sweep_run": [vp, i32, vp, dbl, dbl, dbl, dbl, C.c_int, vp, dbl, dbl, vp, vp],
            "ma_bem_sweep_solutions_dev": [vp, P(vp), P(i32)],
            "ma_bem_sweep_set_timing": [vp, C.c_int],
            "ma_bem_sweep_last_timing": [vp, vp],
            "ma_bem_sweep_info": [vp, P(i32), P(i32), P(i32), P(i32), P(i32)],
            "ma_bem_sweep_lu_plan": [vp, P(vp)],
            "ma_bem_sweep_stream": [vp, P(vp)],
            "ma_bem_solve_sweep_multi": [P(ma_mesh_t), vp, i32, i32, vp, dbl, dbl, dbl, dbl, C.c_int, vp, dbl, dbl, i32, vp, vp],
            "ma_bem_solve_sweep_multi_timed": [P(ma_mesh_t), vp, i32, i32, vp, dbl, dbl, dbl, dbl, C.c_int, vp, dbl, dbl, i32, vp, vp, vp, vp, vp],
            "ma_bem_plan_assemble_multi_dev": [vp, i32, vp, vp, vp, vp, vp, vp],
            "ma_bem_plan_assemble_multi_part_dev": [vp, i32, vp, vp, vp, vp, vp, i32, i32, vp],
            "ma_lu_plan_main_stream": [vp, P(vp)],
            "ma_lu_plan_stage_spacing": [vp, i32, P(i32)],
            "ma_lu_plan_dump_intervals": [vp, i32, vp, i32, P(i32)],
            "ma_sweep_owner": [i32, i32],
            "ma_bem_plan_device": [vp, P(C.c_int)],
            "ma_lu_factorize": [i32, vp, P(vp)],
            "ma_lu_factorization_solve": [vp, vp, vp],
            "ma_lu_factorization_destroy": [vp],
            "ma_lu_plan_factor_solve_batch_dev": [vp, i32, vp, vp, i32, vp],
            "ma_lu_plan_num_blocks": [vp, P(i32)],
            "ma_lu_plan_reserve_events": [vp, C.c_int64],
            "ma_lu_plan_stage_reset": [vp, vp],
            "ma_lu_plan_slot_stream": [vp, i32, P(vp)],
            "ma_lu_plan_stage_begin": [vp, i32, vp, vp, i32, vp],
            "ma_lu_plan_stage_round": [vp, i32, vp, vp, vp],
            "ma_lu_plan_stage_finish": [vp, i32, vp],
            "ma_lu_plan_stage_info_dev": [vp, i32, vp, vp],
            "ma_lu_plan_status": [vp, vp],
            "ma_lu_plan_set_timing": [vp, C.c_int],
            "ma_lu_plan_last_timing": [vp, vp],
            "ma_lu_plan_last_update_stats": [vp, P(dbl), P(dbl), P(dbl)],
            "ma_lu_plan_last_big_update_stats": [vp, P(dbl), P(dbl)],
            "ma_csr_create": [i64, vp, vp, vp, C.c_int, P(vp)],
            "ma_csr_create_helmholtz": [i64, vp, vp, vp, vp, C.c_int, P(vp)],
            "ma_csr_create_rect": [i64, i64, vp, vp, vp, C.c_int, P(vp)],
            "ma_csr_num_cols": [vp, P(i64)],
            "ma_precond_create_amg": [i32, vp, vp, vp, i32, dbl, i32, i32, i32, P(vp)],
            "ma_csr_destroy": [vp],
            "ma_csr_num_rows": [vp, P(i64), P(i64)],
            "ma_csr_set_wavenumber": [vp, dbl, dbl],
            "ma_csr_add_boundary": [vp, i32, vp],
            "ma_csr_assemble": [vp, dbl, dbl, i32, vp, vp, vp],
            "ma_csr_spmv": [vp, vp, vp],
            "ma_csr_residual": [vp, vp, vp, vp],
            "ma_csr_jacobi": [vp, vp, vp, dbl, C.c_int],
            "ma_csr_l1jacobi": [vp, vp, vp, C.c_int],
            "ma_csr_spmv_dev": [vp, vp, vp, vp],
            "ma_csr_residual_dev": [vp, vp, vp, vp, vp],
            "ma_csr_jacobi_dev": [vp, vp, vp, dbl, C.c_int, vp, vp],
            "ma_csr_l1jacobi_dev": [vp, vp, vp, C.c_int, vp, vp],
            "ma_csr_sym_gauss_seidel": [vp, vp, vp, C.c_int],
            "ma_csr_sym_gauss_seidel_dev": [vp, vp, vp, C.c_int, vp],
            "ma_csr_gauss_seidel_sweep_dev": [vp, vp, vp, C.c_int, C.c_int, vp],
            "ma_csr_gauss_seidel_levels": [vp, P(C.c_int64), P(C.c_int64)],
            "ma_fem_matrix_create": [i64, i64, vp, vp, vp, C.c_int, P(vp)],
            "ma_fem_smooth": [vp, vp, vp, C.c_int, C.c_int, dbl],
            "ma_fem_residual": [vp, vp, vp, vp],
            "ma_op_create_dense": [i64, vp, C.c_int, P(vp)],
            "ma_op_create_dense_dev": [i64, vp, C.c_int, P(vp)],
            "ma_op_create_csr": [vp, P(vp)],
            "ma_op_create_tbem": [vp, P(ma_physics_t), dbl, dbl, i32, i32, P(vp)],
            "ma_op_create_tbem_multi": [P(ma_mesh_t), P(ma_physics_t), dbl, dbl, vp, i32, P(vp)],
            "ma_op_num_shards": [vp, P(i32), vp, vp],
            "ma_op_create_slfmm": [vp, P(ma_clusters_t), P(ma_physics_t), i32, i32, i32, P(vp)],
            "ma_op_slfmm_near_matrix": [vp, vp],
            "ma_precond_create_ilu0": [vp, P(vp)],
            "ma_precond_create_ilu_fixed_point": [vp, i32, P(vp)],
            "ma_precond_create_schwarz": [vp, i32, i32, P(vp)],
            "ma_precond_schwarz_stats": [vp, P(i64), P(i64), P(i64), P(dbl)],
            "ma_csr_get": [vp, vp, vp, vp],
            "ma_amg_config_preset": [i32, vp],
            "ma_precond_create_amg_from_csr": [vp, vp, P(vp)],
            "ma_precond_amg_info": [vp, P(i32), P(dbl), P(dbl), P(dbl)],
            "ma_precond_amg_level": [vp, i32, P(vp), P(vp), P(vp)],
            "ma_bicgstab": [vp, vp, i32, dbl, vp, vp],
            "ma_cgs": [vp, vp, i32, dbl, vp, vp],
            "ma_cg": [vp, vp, i32, dbl, vp, vp],
            "ma_cluster_tree_build": [P(ma_mesh_t), i32, dbl, P(vp)],
            "ma_cluster_tree_destroy": [vp],
            "ma_cluster_tree_num_levels": [vp, P(i32)],
            "ma_cluster_tree_level_info": [vp, i32, P(i32), P(i32), P(i32), P(i32), P(i64), P(i64), P(i64), P(i64)],
            "ma_cluster_tree_level_get": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
            "ma_op_create_mlfmm": [vp, vp, P(ma_physics_t), P(vp)],
            "ma_op_destroy": [vp],
            "ma_op_num_rows": [vp, P(i64)],
            "ma_op_apply": [vp, vp, vp],
            "ma_op_apply_dev": [vp, vp, vp, vp],
            "ma_op_apply_transpose": [vp, vp, vp],
            "ma_op_apply_hermitian": [vp, vp, vp],
            "ma_op_apply_transpose_dev": [vp, vp, vp, vp],
            "ma_op_apply_hermitian_dev": [vp, vp, vp, vp],
            "ma_csr_transpose": [vp, P(vp)],
            "ma_gmres": [vp, vp, vp, i32, i32, dbl, vp, vp],
            "ma_precond_create_jacobi": [vp, dbl, i32, P(vp)],
            "ma_precond_create_l1jacobi": [vp, i32, P(vp)],
            "ma_precond_create_sym_gauss_seidel": [vp, i32, P(vp)],
            "ma_precond_create_diagonal": [vp, P(vp)],
            "ma_precond_destroy": [vp],
            "ma_precond_apply_dev": [vp, vp, vp, vp],
            "ma_precond_apply": [vp, vp, vp],
            "ma_gmres_preconditioned": [vp, vp, vp, vp, i32, i32, dbl, vp, vp],
            "ma_gmres_pipelined": [vp, vp, vp, vp, i32, i32, dbl, vp, vp],
            "ma_bem_plan_scattered_field": [vp, P(ma_physics_t), i32, vp, vp, vp, vp],
            "ma_room_build_matrix": [i32, vp, vp, vp, dbl, vp],
            "ma_room_build_matrix_dev": [i32, vp, vp, vp, dbl, vp, vp],
            "ma_bem_incident_evaluate": [C.c_int, vp, vp, P(ma_physics_t), C.c_int, vp, dbl, dbl, vp, vp],
            "ma_room_element_data": [i32, vp, vp, vp, vp, vp, vp],
            "ma_room_build_matrix_adaptive": [i32, vp, i32, vp, dbl, C.c_int, vp],
            "ma_room_incident_derivative": [i32, vp, vp, i32, vp, vp, C.c_int, dbl, vp],
            "ma_room_field_pressure": [i32, vp, vp, vp, vp, i32, vp, vp, C.c_int, i32, vp, dbl, vp],
            "ma_zgemm_sub": [i32, i32, i32, vp, vp, vp],
            "ma_probe_mfma_f64": [C.c_int, P(dbl)],
            "ma_diag_zgemm_dev": [i32, i32, i32, vp, vp, vp, i32, vp],
            "ma_diag_mfma_burn": [vp, i32, i32, i32, vp],
        }
        for name, args in sig.items():
            if hasattr(L, name):
                f = getattr(L, name)
                f.argtypes = args
                f.restype = C.c_int
        _lib = L
    return _lib


def check(status):
    if status != MA_OK:
        raise MaError(status, lib().ma_last_error_string().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().ma_device_count(C.byref(n)))
    return n.value


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MeshArrays:
    """Keeps the NumPy arrays behind an ma_mesh_t alive (the C side only borrows them)."""

    def __init__(self, nodes, conn, center, normal, area, dof=None, bc_type=None, bc_values=None, bc_len=None, is_eval=None):
        n = int(np.asarray(conn).shape[0])
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        self.conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.center = np.ascontiguousarray(center, dtype=np.float64)
        self.normal = np.ascontiguousarray(normal, dtype=np.float64)
        self.area = np.ascontiguousarray(area, dtype=np.float64)
        self.dof = np.ascontiguousarray(np.arange(n) if dof is None else dof, dtype=np.int32)
        self.bc_type = np.ascontiguousarray(np.zeros(n) if bc_type is None else bc_type, dtype=np.uint8)
        self.bc_values = None if bc_values is None else np.ascontiguousarray(bc_values, dtype=np.complex128)
        self.bc_len = None if bc_len is None else np.ascontiguousarray(bc_len, dtype=np.int32)
        self.is_eval = None if is_eval is None else np.ascontiguousarray(is_eval, dtype=np.uint8)
        self.n_elem = n
        self.c = ma_mesh_t(self.nodes.shape[0], _vp(self.nodes), n, _vp(self.conn), _vp(self.center), _vp(self.normal),
                           _vp(self.area), _vp(self.dof), _vp(self.bc_type), _vp(self.bc_values), _vp(self.bc_len),
                           _vp(self.is_eval))


def physics(k, harmonic=1.0, tau=1.0, gamma=1.0):
    return ma_physics_t(float(k), float(harmonic), float(tau), float(gamma))


def assemble_tbem(mesh, k, beta, harmonic=1.0, tau=1.0):
    """Host-buffer drop-in of build_tbem_system_with_beta (tbem.rs:96): returns (A, rhs)."""
    nd = mesh.n_elem if mesh.is_eval is None else int((mesh.is_eval == 0).sum())
    A = np.empty((nd, nd), dtype=np.complex128)
    rhs = np.empty(nd, dtype=np.complex128)
    ph = physics(k, harmonic, tau)
    beta = complex(beta)
    check(lib().ma_bem_assemble_tbem(C.byref(mesh.c), C.byref(ph), beta.real, beta.imag, _vp(A), _vp(rhs)))
    return A, rhs


def incident_rhs(centers, normals, k, beta, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, harmonic=1.0, tau=1.0):
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    normals = np.ascontiguousarray(normals, dtype=np.float64)
    v = np.ascontiguousarray(vec, dtype=np.float64)
    out = np.empty(centers.shape[0], dtype=np.complex128)
    ph = physics(k, harmonic, tau)
    beta = complex(beta); amp = complex(amp)
    check(lib().ma_bem_incident_rhs(centers.shape[0], _vp(centers), _vp(normals), C.byref(ph), beta.real, beta.imag,
                                    kind, _vp(v), amp.real, amp.imag, _vp(out)))
    return out


class BemPlan:
    """ma_bem_plan_t: geometry + near-pair plan resident in HBM; assemble per frequency."""

    def __init__(self, mesh, device=0):
        self.mesh = mesh
        self.h = C.c_void_p()
        check(lib().ma_bem_plan_create(C.byref(mesh.c), device, C.byref(self.h)))
        n = C.c_int32()
        check(lib().ma_bem_plan_num_dofs(self.h, C.byref(n)))
        self.num_dofs = n.value
        m = C.c_int64()
        check(lib().ma_bem_plan_num_near_pairs(self.h, C.byref(m)))
        self.num_near_pairs = m.value
        d = C.c_int()
        check(lib().ma_bem_plan_device(self.h, C.byref(d)))
        self.device = d.value

    def close(self):
        if self.h:
            lib().ma_bem_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def assemble_dev(self, k, beta, d_A, d_rhs, stream=0, harmonic=1.0, tau=1.0):
        ph = physics(k, harmonic, tau); beta = complex(beta)
        check(lib().ma_bem_plan_assemble_dev(self.h, C.byref(ph), beta.real, beta.imag, C.c_void_p(d_A), C.c_void_p(d_rhs),
                                             C.c_void_p(stream)))

    def assemble_multi_dev(self, ks, betas, d_As, d_rhss, stream=0, harmonic=1.0, tau=1.0):
        """ma_bem_plan_assemble_multi_dev: the systems of several wavenumbers in one call (far pairs of up to three per pass)."""
        nf = len(ks)
        phs = (type(physics(1.0)) * nf)(*[physics(k, harmonic, tau) for k in ks])
        br = (C.c_double * nf)(*[complex(b).real for b in betas]); bi = (C.c_double * nf)(*[complex(b).imag for b in betas])
        pa = (C.c_void_p * nf)(*[int(a) for a in d_As]); pr = (C.c_void_p * nf)(*[int(r) for r in d_rhss])
        check(lib().ma_bem_plan_assemble_multi_dev(self.h, nf, phs, br, bi, pa, pr, C.c_void_p(stream)))

    def assemble_multi_part_dev(self, ks, betas, d_As, d_rhss, part, nparts, stream=0, harmonic=1.0, tau=1.0):
        """ma_bem_plan_assemble_multi_part_dev: piece `part` of `nparts` of assemble_multi_dev (issue them in order on one stream)."""
        nf = len(ks)
        phs = (type(physics(1.0)) * nf)(*[physics(k, harmonic, tau) for k in ks])
        br = (C.c_double * nf)(*[complex(b).real for b in betas]); bi = (C.c_double * nf)(*[complex(b).imag for b in betas])
        pa = (C.c_void_p * nf)(*[int(a) for a in d_As]); pr = (C.c_void_p * nf)(*[int(r) for r in d_rhss])
        check(lib().ma_bem_plan_assemble_multi_part_dev(self.h, nf, phs, br, bi, pa, pr, int(part), int(nparts), C.c_void_p(stream)))

    def incident_rhs_dev(self, k, beta, d_rhs, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=0,
                         harmonic=1.0, tau=1.0):
        ph = physics(k, harmonic, tau); beta = complex(beta); amp = complex(amp)
        v = np.ascontiguousarray(vec, dtype=np.float64)
        check(lib().ma_bem_plan_incident_rhs_dev(self.h, C.byref(ph), beta.real, beta.imag, kind, _vp(v), amp.real, amp.imag,
                                                 1 if accumulate else 0, C.c_void_p(d_rhs), C.c_void_p(stream)))

    def probe_pairs(self, k, pairs, harmonic=1.0, tau=1.0):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty((pairs.shape[0], 5), dtype=np.complex128)
        ph = physics(k, harmonic, tau)
        check(lib().ma_bem_plan_probe_pairs(self.h, C.byref(ph), pairs.shape[0], _vp(pairs), _vp(out)))
        return out

    def probe_self(self, k, harmonic=1.0, tau=1.0):
        out = np.empty((self.num_dofs, 5), dtype=np.complex128)
        ph = physics(k, harmonic, tau)
        check(lib().ma_bem_plan_probe_self(self.h, C.byref(ph), _vp(out)))
        return out

    def near_pairs(self):
        out = np.empty((self.num_near_pairs, 2), dtype=np.int32)
        check(lib().ma_bem_plan_get_near_pairs(self.h, _vp(out)))
        return out

    def set_timing(self, on=True):
        check(lib().ma_bem_plan_set_timing(self.h, 1 if on else 0))

    def last_timing(self):
        out = np.zeros(3)
        check(lib().ma_bem_plan_last_timing(self.h, _vp(out)))
        return out


PIVOT_PARTIAL, PIVOT_TOURNAMENT = 0, 1     # MA_LU_PIVOT_* of include/mathaudio_hip.h


def _pivoting(p):
    if p is None:
        return None
    if isinstance(p, str):
        return {"partial": PIVOT_PARTIAL, "tournament": PIVOT_TOURNAMENT}[p]
    return int(p)


def zgesv(A, b, return_pivots=False, pivoting=None, return_factors=False):
    """Host-buffer drop-in of lu_solve (lu.rs:142): returns x; raises MaError(MA_ERR_SINGULAR / MA_ERR_DIM).
    return_pivots: also the 0-based row interchanges (LAPACK ipiv - 1). pivoting: None / "partial" (ma_zgesv) or "tournament"
    (ma_zgesv_pivoting). return_factors: also L \\ U as LAPACK stores them (with return_pivots)."""
    A = np.array(A, dtype=np.complex128, order="C")
    x = np.array(b, dtype=np.complex128)
    if A.ndim != 2 or A.shape[0] != A.shape[1] or x.shape != (A.shape[0],):
        raise MaError(MA_ERR_DIM, "A must be n x n and b of length n")
    piv = np.zeros(A.shape[0], dtype=np.int32)
    pv = _pivoting(pivoting)
    want_piv = return_pivots or return_factors
    if pv is None:
        check(lib().ma_zgesv(A.shape[0], _vp(A), _vp(x), _vp(piv) if want_piv else None))
    else:
        check(lib().ma_zgesv_pivoting(A.shape[0], _vp(A), _vp(x), _vp(piv) if want_piv else None, pv))
    if return_factors:
        return x, piv, A
    return (x, piv) if return_pivots else x


class LuPlan:
    """ma_lu_plan_t: workspace for device-resident factor+solve of an n x n complex128 system."""

    def __init__(self, n, device=0, pivoting=None):
        """pivoting: None (ma_lu_plan_create: partial), "partial" or "tournament" (ma_lu_plan_create_pivoting)."""
        self.n = n
        self.h = C.c_void_p()
        pv = _pivoting(pivoting)
        if pv is None:
            check(lib().ma_lu_plan_create(n, device, C.byref(self.h)))
        else:
            check(lib().ma_lu_plan_create_pivoting(n, device, pv, C.byref(self.h)))

    def set_speculation(self, mode):
        """ma_lu_plan_set_speculation: "off" / 0, "verified" / 1 (fallback in line), "optimistic" / 2 (status MA_ERR_RETRY on a rejected panel)."""
        m = {"off": 0, "verified": 1, "optimistic": 2}.get(mode, mode)
        check(lib().ma_lu_plan_set_speculation(self.h, int(m)))

    def speculation(self):
        v = C.c_int32(0)
        check(lib().ma_lu_plan_speculation(self.h, C.byref(v)))
        return ("off", "verified", "optimistic")[v.value]

    def speculation_stats(self):
        """ma_lu_plan_speculation_stats: half-panels (accepted at the first attempt, accepted by the widened attempt, rejected) since the plan was made."""
        a = C.c_int64(0); w = C.c_int64(0); r = C.c_int64(0)
        check(lib().ma_lu_plan_speculation_stats(self.h, C.byref(a), C.byref(w), C.byref(r)))
        return a.value, w.value, r.value

    def pivoting(self):
        v = C.c_int32(0)
        check(lib().ma_lu_plan_pivoting(self.h, C.byref(v)))
        return "tournament" if v.value == PIVOT_TOURNAMENT else "partial"

    def close(self):
        if self.h:
            lib().ma_lu_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def factor_solve_dev(self, d_A, d_B, nrhs=1, stream=0):
        check(lib().ma_lu_plan_factor_solve_dev(self.h, C.c_void_p(d_A), C.c_void_p(d_B), nrhs, C.c_void_p(stream)))

    def solve_dev(self, d_A_factored, d_B, nrhs=1, stream=0):
        """ma_lu_plan_solve_dev: further right-hand sides with the factors the plan's last factor_solve_dev left in d_A."""
        check(lib().ma_lu_plan_solve_dev(self.h, C.c_void_p(d_A_factored), C.c_void_p(d_B), nrhs, C.c_void_p(stream)))

    def factor_solve_batch_dev(self, d_As, d_Bs, nrhs=1, stream=0):
        """Interleaved factor+solve of len(d_As) independent systems (device pointers)."""
        m = len(d_As)
        pa = (C.c_void_p * m)(*[C.c_void_p(p) for p in d_As]); pb = (C.c_void_p * m)(*[C.c_void_p(p) for p in d_Bs])
        check(lib().ma_lu_plan_factor_solve_batch_dev(self.h, m, pa, pb, nrhs, C.c_void_p(stream)))

    # staged (pipelined) use: see include/mathaudio_hip.h
    def num_blocks(self):
        g = C.c_int32()
        check(lib().ma_lu_plan_num_blocks(self.h, C.byref(g)))
        return g.value

    def reserve_events(self, count):
        check(lib().ma_lu_plan_reserve_events(self.h, int(count)))

    def slot_stream(self, slot):
        p = C.c_void_p()
        check(lib().ma_lu_plan_slot_stream(self.h, int(slot), C.byref(p)))
        return p.value

    def main_stream(self):
        """ma_lu_plan_main_stream: the CU-masked stream of the big updates (MA_LU_CU_SPLIT), or None."""
        p = C.c_void_p()
        check(lib().ma_lu_plan_main_stream(self.h, C.byref(p)))
        return p.value

    def cu_split(self):
        """ma_lu_plan_cu_split: (CUs the big updates stay off, CUs of the chip)."""
        a = C.c_int32(0); b = C.c_int32(0)
        lib().ma_lu_plan_cu_split.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        check(lib().ma_lu_plan_cu_split(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dump_intervals(self, phase, capacity=200000):
        """ma_lu_plan_dump_intervals: (start, end) ms of the timed intervals of one phase of the last timed staged run."""
        out = np.zeros((capacity, 2), dtype=np.float64)
        c = C.c_int32(0)
        check(lib().ma_lu_plan_dump_intervals(self.h, int(phase), _vp(out), int(capacity), C.byref(c)))
        return out[:min(c.value, capacity)]

    def stage_spacing(self, slots):
        """ma_lu_plan_stage_spacing: rounds between the starts of two slots of the staged schedule."""
        v = C.c_int32(0)
        check(lib().ma_lu_plan_stage_spacing(self.h, int(slots), C.byref(v)))
        return v.value

    def stage_reset(self, stream=0):
        check(lib().ma_lu_plan_stage_reset(self.h, C.c_void_p(stream)))

    def stage_begin(self, slot, d_A, d_B, nrhs=1, stream=0):
        check(lib().ma_lu_plan_stage_begin(self.h, int(slot), C.c_void_p(d_A), C.c_void_p(d_B), nrhs, C.c_void_p(stream)))

    def stage_round(self, slots, blocks, stream=0):
        m = len(slots)
        a = (C.c_int32 * m)(*slots); b = (C.c_int32 * m)(*blocks)
        check(lib().ma_lu_plan_stage_round(self.h, m, a, b, C.c_void_p(stream)))

    def stage_finish(self, slot, stream=0):
        check(lib().ma_lu_plan_stage_finish(self.h, int(slot), C.c_void_p(stream)))

    def status(self, stream=0):
        return lib().ma_lu_plan_status(self.h, C.c_void_p(stream))

    def set_timing(self, on=True):
        """True / 1: every phase is bracketed by HIP events; 2: only the trailing-update launches; False / 0: off."""
        check(lib().ma_lu_plan_set_timing(self.h, int(on)))

    def last_timing(self):
        out = np.zeros(8)
        check(lib().ma_lu_plan_last_timing(self.h, _vp(out)))
        return out

    def last_big_update_stats(self):
        """(launches, algorithmic flops) of the big trailing updates on the caller's stream in the last call."""
        a, b = C.c_double(), C.c_double()
        check(lib().ma_lu_plan_last_big_update_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_update_stats(self):
        """(launches, algorithmic flops, algorithmic C bytes) of every update launch (caller's stream and lanes) of the last call."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(lib().ma_lu_plan_last_update_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


def zgemm_sub(A, B, Cm):
    """ma_zgemm_sub: C - A @ B on host arrays through the LU's trailing-update kernel (f64 matrix cores)."""
    A = np.ascontiguousarray(A, dtype=np.complex128); B = np.ascontiguousarray(B, dtype=np.complex128)
    out = np.array(Cm, dtype=np.complex128, order="C")
    check(lib().ma_zgemm_sub(A.shape[0], B.shape[1], A.shape[1], _vp(A), _vp(B), _vp(out)))
    return out


def probe_mfma_f64(device=0):
    t = C.c_double(0)
    check(lib().ma_probe_mfma_f64(device, C.byref(t)))
    return t.value


class CsrOperator:
    """ma_csr_t: CsrMatrix<Complex64> (values=...) or the Helmholtz K/M pair of one pattern (K=..., M=...)."""

    def __init__(self, row_ptrs, col_indices, values=None, K=None, M=None, device=0):
        self.rp = np.ascontiguousarray(row_ptrs, dtype=np.int64)
        self.ci = np.ascontiguousarray(col_indices, dtype=np.int64)
        self.n = len(self.rp) - 1
        self.h = C.c_void_p()
        if values is not None:
            v = np.ascontiguousarray(values, dtype=np.complex128)
            check(lib().ma_csr_create(self.n, _vp(self.rp), _vp(self.ci), _vp(v), device, C.byref(self.h)))
        else:
            k = np.ascontiguousarray(K, dtype=np.float64); m = np.ascontiguousarray(M, dtype=np.float64)
            check(lib().ma_csr_create_helmholtz(self.n, _vp(self.rp), _vp(self.ci), _vp(k), _vp(m), device, C.byref(self.h)))

    @staticmethod
    def rect(nrows, ncols, row_ptrs, col_indices, values, device=0):
        """ma_csr_create_rect: a rectangular operator (the AMG transfer operators P and R); matvec only."""
        self = CsrOperator.__new__(CsrOperator)
        rp = np.ascontiguousarray(row_ptrs, dtype=np.int64); ci = np.ascontiguousarray(col_indices, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.complex128)
        self.n = int(nrows); self.ncols = int(ncols); self.nnz = int(rp[-1]); self.h = C.c_void_p()
        check(lib().ma_csr_create_rect(self.n, self.ncols, _vp(rp), _vp(ci), _vp(v), device, C.byref(self.h)))
        return self

    @staticmethod
    def from_coo(n, rows, cols, values, device=0):
        """HelmholtzMatrix triplets (helmholtz.rs:22-33) -> operator; duplicates are summed, zero-diagonal rows are skipped by sweeps."""
        rows = np.ascontiguousarray(rows, dtype=np.int64); cols = np.ascontiguousarray(cols, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.complex128)
        self = CsrOperator.__new__(CsrOperator)
        self.n = int(n); self.rp = None; self.ci = None
        self.h = C.c_void_p()
        check(lib().ma_fem_matrix_create(self.n, len(rows), _vp(rows), _vp(cols), _vp(values), device, C.byref(self.h)))
        return self

    def fem_smooth(self, x, b, kind=0, iterations=2, omega=2.0 / 3.0):
        """smooth() of math-fem/src/multigrid/smoother.rs:44-68: kind 0 Gauss-Seidel (default), 1 Jacobi, 2 symmetric GS."""
        x = np.array(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        check(lib().ma_fem_smooth(self.h, _vp(x), _vp(b), int(kind), int(iterations), float(omega)))
        return x

    def fem_residual(self, x, b):
        x = np.ascontiguousarray(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        r = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_fem_residual(self.h, _vp(x), _vp(b), _vp(r)))
        return r

    def close(self):
        if self.h:
            lib().ma_csr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_wavenumber(self, k):
        k = complex(k)
        check(lib().ma_csr_set_wavenumber(self.h, k.real, k.imag))

    def add_boundary(self, tag, values):
        """HelmholtzAssembler.boundary_values[tag] (assembler.rs:19-32)."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        check(lib().ma_csr_add_boundary(self.h, int(tag), _vp(v)))

    def assemble(self, k, boundary_coeffs=None, stream=0):
        """HelmholtzAssembler::assemble(wavenumber, boundary_coeffs) (assembler.rs:216-257)."""
        k = complex(k)
        bc = boundary_coeffs or {}
        tags = np.ascontiguousarray(list(bc.keys()), dtype=np.int32); co = np.ascontiguousarray([complex(v) for v in bc.values()], dtype=np.complex128)
        check(lib().ma_csr_assemble(self.h, k.real, k.imag, len(tags), _vp(tags) if len(tags) else None, _vp(co) if len(tags) else None, C.c_void_p(stream)))

    def matvec(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128); y = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_csr_spmv(self.h, _vp(x), _vp(y)))
        return y

    def residual(self, x, b):
        x = np.ascontiguousarray(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        r = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_csr_residual(self.h, _vp(x), _vp(b), _vp(r)))
        return r

    def jacobi(self, x, b, omega, sweeps):
        x = np.array(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        check(lib().ma_csr_jacobi(self.h, _vp(x), _vp(b), float(omega), int(sweeps)))
        return x

    def l1_jacobi(self, x, b, sweeps):
        x = np.array(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        check(lib().ma_csr_l1jacobi(self.h, _vp(x), _vp(b), int(sweeps)))
        return x

    def sym_gauss_seidel(self, x, b, sweeps):
        """smooth_sym_gauss_seidel (amg.rs:932-978), level-scheduled on the device."""
        x = np.array(x, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
        check(lib().ma_csr_sym_gauss_seidel(self.h, _vp(x), _vp(b), int(sweeps)))
        return x

    def gauss_seidel_levels(self):
        f = C.c_int64(); bk = C.c_int64()
        check(lib().ma_csr_gauss_seidel_levels(self.h, C.byref(f), C.byref(bk)))
        return f.value, bk.value

    def sym_gauss_seidel_dev(self, d_x, d_b, sweeps, stream=0):
        check(lib().ma_csr_sym_gauss_seidel_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_b), int(sweeps), C.c_void_p(stream)))

    def spmv_dev(self, d_x, d_y, stream=0):
        check(lib().ma_csr_spmv_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream)))

    def residual_dev(self, d_x, d_b, d_r, stream=0):
        check(lib().ma_csr_residual_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_b), C.c_void_p(d_r), C.c_void_p(stream)))

    def jacobi_dev(self, d_x, d_b, omega, sweeps, d_tmp, stream=0):
        check(lib().ma_csr_jacobi_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_b), float(omega), int(sweeps), C.c_void_p(d_tmp), C.c_void_p(stream)))

    def l1_jacobi_dev(self, d_x, d_b, sweeps, d_tmp, stream=0):
        check(lib().ma_csr_l1jacobi_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_b), int(sweeps), C.c_void_p(d_tmp), C.c_void_p(stream)))


class ClusterTree:
    """build_cluster_tree(elements, target_elements_per_leaf, physics) (mlfmm.rs:979-1038), built on the host by the library."""

    def __init__(self, mesh, target_elements_per_leaf, k):
        self.h = C.c_void_p()
        check(lib().ma_cluster_tree_build(C.byref(mesh.c), int(target_elements_per_leaf), float(k), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ma_cluster_tree_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_levels(self):
        n = C.c_int32()
        check(lib().ma_cluster_tree_num_levels(self.h, C.byref(n)))
        return n.value

    def level(self, l):
        """dict of the level's parameters and lists (centres, radii, element / near / far / son lists as offset + index arrays, fathers)."""
        nc, terms, theta, phi = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        ne, nn, nf, ns = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().ma_cluster_tree_level_info(self.h, l, C.byref(nc), C.byref(terms), C.byref(theta), C.byref(phi), C.byref(ne), C.byref(nn), C.byref(nf), C.byref(ns)))
        n = nc.value
        out = {"n_clusters": n, "expansion_terms": terms.value, "theta_points": theta.value, "phi_points": phi.value,
               "center": np.zeros((n, 3)), "radius": np.zeros(n), "father": np.zeros(n, dtype=np.int32)}
        for nm, cnt in (("elem", ne.value), ("near", nn.value), ("far", nf.value), ("son", ns.value)):
            out[nm + "_ptr"] = np.zeros(n + 1, dtype=np.int32); out[nm + "_idx"] = np.zeros(max(cnt, 1), dtype=np.int32)[:cnt]
        check(lib().ma_cluster_tree_level_get(self.h, l, _vp(out["center"]), _vp(out["radius"]), _vp(out["elem_ptr"]), _vp(out["elem_idx"]), _vp(out["near_ptr"]), _vp(out["near_idx"]),
                                              _vp(out["far_ptr"]), _vp(out["far_idx"]), _vp(out["son_ptr"]), _vp(out["son_idx"]), _vp(out["father"])))
        return out


class GmresInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("restarts", C.c_int32), ("converged", C.c_int32), ("residual", C.c_double)]


class LinearOperator:
    """ma_op_t: the LinearOperator<Complex64> boundary (traits.rs:316-327). Build with dense(), csr() or tbem()."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep           # borrowed objects (plan / CSR handle) must outlive the operator
        n = C.c_int64()
        check(lib().ma_op_num_rows(self.h, C.byref(n)))
        self.n = n.value

    @staticmethod
    def dense(A, device=0):
        A = np.ascontiguousarray(A, dtype=np.complex128)
        h = C.c_void_p()
        check(lib().ma_op_create_dense(A.shape[0], _vp(A), device, C.byref(h)))
        return LinearOperator(h)

    @staticmethod
    def dense_dev(n, d_A, device=0, keep=None):
        h = C.c_void_p()
        check(lib().ma_op_create_dense_dev(n, C.c_void_p(d_A), device, C.byref(h)))
        return LinearOperator(h, keep)

    @staticmethod
    def csr(csr_operator):
        h = C.c_void_p()
        check(lib().ma_op_create_csr(csr_operator.h, C.byref(h)))
        return LinearOperator(h, csr_operator)

    @staticmethod
    def tbem(plan, k, beta, rows=None, harmonic=1.0, tau=1.0):
        ph = physics(k, harmonic, tau); beta = complex(beta)
        r0, r1 = (0, plan.num_dofs) if rows is None else rows
        h = C.c_void_p()
        check(lib().ma_op_create_tbem(plan.h, C.byref(ph), beta.real, beta.imag, r0, r1, C.byref(h)))
        return LinearOperator(h, plan)

    @staticmethod
    def gathered(inner, row0, row1, gather):
        """ma_op_create_gathered: `inner` owns rows [row0, row1) of y = A x; gather(d_y, n, row0, row1, stream) -> 0 completes y in
        place with the other ranks' rows (an all-gather on the caller's communicator), ordered on `stream`."""
        fn_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)

        def _cb(user, d_y, n, r0, r1, stream):
            try:
                return int(gather(int(d_y), int(n), int(r0), int(r1), int(stream or 0)))
            except Exception as e:          # never unwind through the C frames
                import sys
                sys.stderr.write("gather callback failed: %r\n" % (e,))
                return -1
        cb = fn_t(_cb)
        h = C.c_void_p()
        lib().ma_op_create_gathered.argtypes = [C.c_void_p, C.c_int64, C.c_int64, fn_t, C.c_void_p, C.c_void_p]
        check(lib().ma_op_create_gathered(inner.h, int(row0), int(row1), cb, None, C.byref(h)))
        return LinearOperator(h, (inner, cb))

    @staticmethod
    def gathered_rccl(inner, comm, nranks, rank):
        """ma_op_create_gathered_rccl: the row exchange as ncclAllGather INSIDE the library (rank `rank` of `nranks` owns rows
        [rank * per, (rank + 1) * per), per = ceil(n / nranks)); `comm` is an RcclComm (or a raw ncclComm_t of the process's librccl)."""
        h = C.c_void_p()
        lib().ma_op_create_gathered_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        check(lib().ma_op_create_gathered_rccl(inner.h, C.c_void_p(comm.h.value if hasattr(comm, "h") else int(comm)), int(nranks), int(rank), C.byref(h)))
        return LinearOperator(h, (inner, comm))

    @staticmethod
    def tbem_multi(mesh, k, beta, devices, harmonic=1.0, tau=1.0):
        """ma_op_create_tbem_multi: the matrix-free operator row-sharded over `devices` (vectors live on devices[0])."""
        ph = physics(k, harmonic, tau); beta = complex(beta)
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(lib().ma_op_create_tbem_multi(C.byref(mesh.c), C.byref(ph), beta.real, beta.imag, _vp(dv), len(dv), C.byref(h)))
        return LinearOperator(h, mesh)

    @staticmethod
    def slfmm(plan, clusters, k, n_theta, n_phi, n_terms, harmonic=1.0, tau=1.0):
        """ma_op_create_slfmm: build_slfmm_system + SlfmmSystem::matvec / matvec_transpose (assembly/slfmm.rs). `clusters` carries
        center [nc, 3], elem_ptr / elem_idx, near_ptr / near_idx, far_ptr / far_idx (int32 arrays)."""
        ph = physics(k, harmonic, tau)
        arrs = [np.ascontiguousarray(clusters.center, dtype=np.float64)] + [np.ascontiguousarray(getattr(clusters, f), dtype=np.int32)
                                                                           for f in ("elem_ptr", "elem_idx", "near_ptr", "near_idx", "far_ptr", "far_idx")]
        cs = ma_clusters_t(len(arrs[1]) - 1, *[a.ctypes.data if a.size else None for a in arrs])
        h = C.c_void_p()
        check(lib().ma_op_create_slfmm(plan.h, C.byref(cs), C.byref(ph), int(n_theta), int(n_phi), int(n_terms), C.byref(h)))
        return LinearOperator(h, (plan, arrs))

    @staticmethod
    def mlfmm(plan, tree, k, harmonic=1.0, tau=1.0):
        """ma_op_create_mlfmm: build_mlfmm_system + MlfmmSystem::matvec (assembly/mlfmm.rs) over a ClusterTree."""
        ph = physics(k, harmonic, tau)
        h = C.c_void_p()
        check(lib().ma_op_create_mlfmm(plan.h, tree.h, C.byref(ph), C.byref(h)))
        return LinearOperator(h, plan)

    def slfmm_phase_mode(self):
        """ma_op_slfmm_phase_mode: 2 phases recomputed (fast), 1 stored table, 0 libm."""
        v = C.c_int32(-1)
        lib().ma_op_slfmm_phase_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        check(lib().ma_op_slfmm_phase_mode(self.h, C.byref(v)))
        return v.value

    def slfmm_near_matrix(self):
        """SlfmmSystem::extract_near_field_matrix (slfmm.rs:104-132)."""
        A = np.empty((self.n, self.n), dtype=np.complex128)
        check(lib().ma_op_slfmm_near_matrix(self.h, _vp(A)))
        return A

    def shards(self):
        """(first rows, devices) of the operator's shards."""
        ns = C.c_int32()
        check(lib().ma_op_num_shards(self.h, C.byref(ns), None, None))
        rows = np.zeros(ns.value, dtype=np.int32); devs = np.zeros(ns.value, dtype=np.int32)
        check(lib().ma_op_num_shards(self.h, C.byref(ns), _vp(rows), _vp(devs)))
        return rows, devs

    def close(self):
        if self.h:
            lib().ma_op_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128); y = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_op_apply(self.h, _vp(x), _vp(y)))
        return y

    def apply_dev(self, d_x, d_y, stream=0):
        check(lib().ma_op_apply_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream)))

    def apply_transpose_dev(self, d_x, d_y, stream=0):
        check(lib().ma_op_apply_transpose_dev(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream)))

    def apply_transpose(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128); y = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_op_apply_transpose(self.h, _vp(x), _vp(y)))
        return y

    def apply_hermitian(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex128); y = np.empty(self.n, dtype=np.complex128)
        check(lib().ma_op_apply_hermitian(self.h, _vp(x), _vp(y)))
        return y


def gmres(op, b, x0=None, restart=30, max_iterations=100, tol=1e-6):
    """gmres / gmres_with_guess (gmres.rs:96-277) on the device: returns (x, GmresInfo)."""
    b = np.ascontiguousarray(b, dtype=np.complex128)
    x = np.empty(op.n, dtype=np.complex128)
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.complex128)
    info = GmresInfo()
    check(lib().ma_gmres(op.h, _vp(b), _vp(x0a), restart, max_iterations, float(tol), _vp(x), C.byref(info)))
    return x, info


def _krylov(fn, op, b, max_iterations, tol):
    b = np.ascontiguousarray(b, dtype=np.complex128)
    x = np.empty_like(b); info = GmresInfo()
    check(fn(op.h, _vp(b), int(max_iterations), float(tol), _vp(x), C.byref(info)))
    return x, info


def bicgstab(op, b, max_iterations=1000, tol=1e-6):
    """bicgstab(operator, b, config) (math-solvers/src/iterative/bicgstab.rs:46-182)."""
    return _krylov(lib().ma_bicgstab, op, b, max_iterations, tol)


def cgs(op, b, max_iterations=1000, tol=1e-6):
    """cgs(operator, b, config) (cgs.rs:46-139)."""
    return _krylov(lib().ma_cgs, op, b, max_iterations, tol)


def cg(op, b, max_iterations=1000, tol=1e-6):
    """cg(operator, b, config) (cg.rs:49-138)."""
    return _krylov(lib().ma_cg, op, b, max_iterations, tol)


class Preconditioner:
    """ma_precond_t: the Preconditioner<Complex64> boundary (traits.rs:370-375); Jacobi / l1-Jacobi / symmetric GS sweeps from zero."""

    def __init__(self, csr_operator, kind="jacobi", omega=2.0 / 3.0, sweeps=2):
        self.h = C.c_void_p()
        self._keep = csr_operator
        if kind == "diagonal":         # csr_operator is a LinearOperator of any kind here (fmm_interface.rs:177-212)
            check(lib().ma_precond_create_diagonal(csr_operator.h, C.byref(self.h)))
        elif kind == "jacobi":
            check(lib().ma_precond_create_jacobi(csr_operator.h, float(omega), int(sweeps), C.byref(self.h)))
        elif kind in ("sgs", "sym_gauss_seidel"):
            check(lib().ma_precond_create_sym_gauss_seidel(csr_operator.h, int(sweeps), C.byref(self.h)))
        else:
            check(lib().ma_precond_create_l1jacobi(csr_operator.h, int(sweeps), C.byref(self.h)))

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.complex128); z = np.empty_like(r)
        check(lib().ma_precond_apply(self.h, _vp(r), _vp(z)))
        return z

    def close(self):
        if self.h:
            lib().ma_precond_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gmres_preconditioned(op, precond, b, x0=None, restart=30, max_iterations=100, tol=1e-6):
    """gmres_preconditioned(_with_guess) (gmres.rs:282-585) on the device: returns (x, GmresInfo)."""
    b = np.ascontiguousarray(b, dtype=np.complex128)
    x = np.empty(op.n, dtype=np.complex128)
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.complex128)
    info = GmresInfo()
    check(lib().ma_gmres_preconditioned(op.h, precond.h, _vp(b), _vp(x0a), restart, max_iterations, float(tol), _vp(x), C.byref(info)))
    return x, info


def scattered_field(plan, k, eval_points, surface_pressure, surface_velocity=None, harmonic=1.0, tau=1.0):
    """compute_scattered_field (postprocess/pressure.rs:81-137) on the device."""
    ep = np.ascontiguousarray(eval_points, dtype=np.float64)
    ps = np.ascontiguousarray(surface_pressure, dtype=np.complex128)
    vs = None if surface_velocity is None else np.ascontiguousarray(surface_velocity, dtype=np.complex128)
    out = np.empty(ep.shape[0], dtype=np.complex128)
    ph = physics(k, harmonic, tau)
    check(lib().ma_bem_plan_scattered_field(plan.h, C.byref(ph), ep.shape[0], _vp(ep), _vp(ps), _vp(vs), _vp(out)))
    return out


def room_build_matrix(center, normal, area, k):
    """build_bem_matrix_parallel (room_acoustics/solver.rs:448-493) on the device."""
    c = np.ascontiguousarray(center, dtype=np.float64); nr = np.ascontiguousarray(normal, dtype=np.float64)
    a = np.ascontiguousarray(area, dtype=np.float64)
    n = len(a)
    A = np.empty((n, n), dtype=np.complex128)
    check(lib().ma_room_build_matrix(n, _vp(c), _vp(nr), _vp(a), float(k), _vp(A)))
    return A


def _conn4(conn):
    conn = np.asarray(conn, dtype=np.int32)
    if conn.ndim == 2 and conn.shape[1] == 3:
        conn = np.concatenate([conn, -np.ones((conn.shape[0], 1), dtype=np.int32)], axis=1)
    return np.ascontiguousarray(conn, dtype=np.int32)


def room_element_data(nodes, conn):
    """element_center_and_normal, element_area, element_characteristic_length (room_acoustics/solver.rs:38-122, 600-611):
    returns (center, normal, area, char_length). Host arithmetic inside the library; needs no device."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64); conn = _conn4(conn)
    n = conn.shape[0]
    c = np.empty((n, 3)); nr = np.empty((n, 3)); a = np.empty(n); cl = np.empty(n)
    check(lib().ma_room_element_data(n, _vp(nodes), _vp(conn), _vp(c), _vp(nr), _vp(a), _vp(cl)))
    return c, nr, a, cl


def room_build_matrix_adaptive(nodes, conn, k, use_adaptive=True):
    """build_bem_matrix_adaptive (room_acoustics/solver.rs:500-597) on the device."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float64); conn = _conn4(conn)
    n = conn.shape[0]
    A = np.empty((n, n), dtype=np.complex128)
    check(lib().ma_room_build_matrix_adaptive(nodes.shape[0], _vp(nodes), n, _vp(conn), float(k), 1 if use_adaptive else 0, _vp(A)))
    return A


def _amp(amp, nsrc, npts):
    amp = np.ascontiguousarray(amp, dtype=np.float64)
    per_point = 1 if amp.ndim == 2 else 0
    if per_point and amp.shape != (nsrc, npts):
        raise ValueError("per-point amplitudes must be (n_sources, n_points)")
    return amp, per_point


def room_incident_derivative(center, normal, src_pos, amp, k):
    """calculate_incident_field_derivative_parallel (solver.rs:638-678): -sum_s dG/dn amp."""
    c = np.ascontiguousarray(center, dtype=np.float64); nr = np.ascontiguousarray(normal, dtype=np.float64)
    sp = np.ascontiguousarray(src_pos, dtype=np.float64).reshape(-1, 3)
    amp, pp = _amp(amp, sp.shape[0], c.shape[0])
    out = np.empty(c.shape[0], dtype=np.complex128)
    check(lib().ma_room_incident_derivative(c.shape[0], _vp(c), _vp(nr), sp.shape[0], _vp(sp), _vp(amp), pp, float(k), _vp(out)))
    return out


def room_field_pressure(center, normal, area, surface_pressure, src_pos, amp, points, k):
    """calculate_field_pressure_bem_parallel (solver.rs:687-748)."""
    c = np.ascontiguousarray(center, dtype=np.float64); nr = np.ascontiguousarray(normal, dtype=np.float64); a = np.ascontiguousarray(area, dtype=np.float64)
    ps = np.ascontiguousarray(surface_pressure, dtype=np.complex128)
    sp = np.ascontiguousarray(src_pos, dtype=np.float64).reshape(-1, 3); pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    amp, pp = _amp(amp, sp.shape[0], pts.shape[0])
    out = np.empty(pts.shape[0], dtype=np.complex128)
    check(lib().ma_room_field_pressure(len(a), _vp(c), _vp(nr), _vp(a), _vp(ps), sp.shape[0], _vp(sp), _vp(amp), pp, pts.shape[0], _vp(pts), float(k), _vp(out)))
    return out


def incident_evaluate(points, k, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, normals=None, harmonic=1.0):
    """IncidentField::evaluate_pressure (and ::evaluate_normal_derivative when normals are given): returns p or (p, dp/dn)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
    ph = physics(k, harmonic)
    p = np.empty(pts.shape[0], dtype=np.complex128)
    if normals is None:
        check(lib().ma_bem_incident_evaluate(pts.shape[0], _vp(pts), None, C.byref(ph), kind, _vp(v), amp.real, amp.imag, _vp(p), None))
        return p
    nr = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    d = np.empty(pts.shape[0], dtype=np.complex128)
    check(lib().ma_bem_incident_evaluate(pts.shape[0], _vp(pts), _vp(nr), C.byref(ph), kind, _vp(v), amp.real, amp.imag, _vp(p), _vp(d)))
    return p, d


def total_field(plan, k, eval_points, surface_pressure, surface_velocity=None, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0):
    """compute_total_field (postprocess/pressure.rs:273-311): (p_incident, p_scattered) at the evaluation points."""
    return incident_evaluate(eval_points, k, kind, vec, amp), scattered_field(plan, k, eval_points, surface_pressure, surface_velocity)


def lu_solve(A, b):
    """lu_solve(&a, &b) -> x (math-solvers/src/direct/lu.rs:142-153): inputs untouched, factors stay on the device."""
    A = np.ascontiguousarray(A, dtype=np.complex128); b = np.ascontiguousarray(b, dtype=np.complex128)
    if A.ndim != 2 or A.shape[0] != A.shape[1] or b.shape != (A.shape[0],):
        raise MaError(MA_ERR_DIM, "lu_solve: A must be n x n and b of length n")
    x = np.empty_like(b)
    check(lib().ma_lu_solve(A.shape[0], _vp(A), _vp(b), _vp(x)))
    return x


class LuFactorization:
    """lu_factorize(&a) (lu.rs:83-137) / LuFactorization::solve(&b) (lu.rs:38-78): the factors live in HBM."""

    def __init__(self, A):
        A = np.ascontiguousarray(A, dtype=np.complex128)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise MaError(MA_ERR_DIM, "lu_factorize: A must be square")
        self.n = A.shape[0]
        self.h = C.c_void_p()
        check(lib().ma_lu_factorize(self.n, _vp(A), C.byref(self.h)))

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.complex128)
        if b.shape != (self.n,):
            raise MaError(MA_ERR_DIM, "solve: b must have n entries")
        x = np.empty_like(b)
        check(lib().ma_lu_factorization_solve(self.h, _vp(b), _vp(x)))
        return x

    def close(self):
        if self.h:
            lib().ma_lu_factorization_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_sweep(plan, frequencies_hz, speed_of_sound=343.0, beta_scale=4.0, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, slots=3, harmonic=1.0, tau=1.0):
    """The BEM drivers' frequency loop as one device-resident call (ma_bem_solve_sweep): returns (X[n_freq, n], status[n_freq])."""
    f = np.ascontiguousarray(frequencies_hz, dtype=np.float64); v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
    X = np.empty((len(f), plan.num_dofs), dtype=np.complex128); st = np.zeros(len(f), dtype=np.int32)
    rc = lib().ma_bem_solve_sweep(plan.h, len(f), _vp(f), float(speed_of_sound), float(harmonic), float(tau), float(beta_scale), int(kind), _vp(v),
                                  amp.real, amp.imag, int(slots), _vp(X), _vp(st))
    if rc not in (MA_OK, MA_ERR_SINGULAR):
        check(rc)
    return X, st


def rccl_unique_id():
    """ma_rccl_get_unique_id: 128 bytes made on rank 0 and handed to every rank of the communicator by the host's own means."""
    buf = (C.c_ubyte * 128)()
    lib().ma_rccl_get_unique_id.argtypes = [C.c_void_p]
    check(lib().ma_rccl_get_unique_id(buf))
    return bytes(buf)


class RcclComm:
    """ma_rccl_comm_create / _destroy: an ncclComm_t of the librccl the library bound (the copy the process already holds, if any)."""

    def __init__(self, nranks, rank, unique_id, device=0):
        self.h = C.c_void_p(); self.nranks = nranks; self.rank = rank
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        lib().ma_rccl_comm_create.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]
        check(lib().ma_rccl_comm_create(int(nranks), int(rank), buf, int(device), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ma_rccl_comm_destroy.argtypes = [C.c_void_p]
            lib().ma_rccl_comm_destroy(self.h)
            self.h = C.c_void_p()


def memcpy_dtod(d_dst, d_src, nbytes, stream=0):
    """ma_device_copy: device-to-device copy, complete on return."""
    check(lib().ma_device_copy(C.c_void_p(d_dst), C.c_void_p(d_src), int(nbytes), C.c_void_p(stream)))


class BemSweep:
    """ma_bem_sweep_t: the frequency loop (room_simulator_bem.rs:328-360) behind a reusable handle -- LU plan, streams, the systems in
    flight, the spares of the assembly-ahead and the parked solutions are allocated once. The plan is borrowed."""

    def __init__(self, plan, max_frequencies, slots=3, pivoting=None):
        """pivoting: None (the sweep's default: tournament), "partial" or "tournament" (ma_bem_sweep_create_pivoting)."""
        self.plan = plan; self.n = plan.num_dofs; self.h = C.c_void_p()
        pv = _pivoting(pivoting)
        check(lib().ma_bem_sweep_create_pivoting(plan.h, int(slots), int(max_frequencies), -1 if pv is None else pv, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().ma_bem_sweep_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, frequencies_hz, speed_of_sound=343.0, beta_scale=4.0, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, harmonic=1.0, tau=1.0, to_host=True):
        """ma_bem_sweep_run: (X[n_freq, n] or None with to_host=False, status[n_freq])."""
        f = np.ascontiguousarray(frequencies_hz, dtype=np.float64); v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
        X = np.empty((len(f), self.n), dtype=np.complex128) if to_host else None
        st = np.zeros(len(f), dtype=np.int32)
        rc = lib().ma_bem_sweep_run(self.h, len(f), _vp(f), float(speed_of_sound), float(harmonic), float(tau), float(beta_scale), int(kind), _vp(v),
                                    amp.real, amp.imag, _vp(X) if to_host else None, _vp(st))
        if rc not in (MA_OK, MA_ERR_SINGULAR):
            check(rc)
        return X, st

    def solutions_dev(self):
        """(device pointer, count): the parked solutions of the last run, row i = its i-th frequency."""
        p = C.c_void_p(); c = C.c_int32(0)
        check(lib().ma_bem_sweep_solutions_dev(self.h, C.byref(p), C.byref(c)))
        return p.value, c.value

    def set_timing(self, enable=True):
        check(lib().ma_bem_sweep_set_timing(self.h, 1 if enable else 0))

    def last_timing(self):
        o = np.zeros(8)
        check(lib().ma_bem_sweep_last_timing(self.h, _vp(o)))
        return {"wall_s": o[0], "device_ms": o[1], "assembly_ms": o[2], "assembly_pieces": int(o[3]), "big_update_ms": o[4], "big_update_launches": int(o[5]),
                "big_update_flops": o[6], "frequencies": int(o[7])}

    def info(self):
        v = [C.c_int32(0) for _ in range(5)]
        check(lib().ma_bem_sweep_info(self.h, *[C.byref(t) for t in v]))
        return {"slots": v[0].value, "blocks": v[1].value, "spacing": v[2].value, "systems_ahead": v[3].value, "staged": bool(v[4].value)}

    def lu_plan(self):
        """The handle's LU plan (borrowed: do not close it) for the diagnostics of LuPlan."""
        p = C.c_void_p()
        check(lib().ma_bem_sweep_lu_plan(self.h, C.byref(p)))
        lu = LuPlan.__new__(LuPlan); lu.n = self.n; lu.h = p; lu.close = lambda: None
        return lu

    def stream(self):
        p = C.c_void_p()
        check(lib().ma_bem_sweep_stream(self.h, C.byref(p)))
        return p.value


def sweep_owner(frequency_index, ndev):
    """Index into `devices` of the device that solves a frequency in ma_bem_solve_sweep_multi (f mod ndev)."""
    return lib().ma_sweep_owner(int(frequency_index), int(ndev))


def solve_sweep_multi(mesh, devices, frequencies_hz, speed_of_sound=343.0, beta_scale=4.0, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, slots=3, harmonic=1.0, tau=1.0):
    """ma_bem_solve_sweep_multi: the frequency loop over the GPUs of one node, frequency f on devices[f mod ndev], one host
    thread per device inside the library. Returns (X[n_freq, n], status[n_freq])."""
    f = np.ascontiguousarray(frequencies_hz, dtype=np.float64); v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    nd = int((mesh.is_eval == 0).sum()) if mesh.is_eval is not None else mesh.n_elem
    X = np.empty((len(f), nd), dtype=np.complex128); st = np.zeros(len(f), dtype=np.int32)
    rc = lib().ma_bem_solve_sweep_multi(C.byref(mesh.c), _vp(dv), len(dv), len(f), _vp(f), float(speed_of_sound), float(harmonic), float(tau), float(beta_scale),
                                        int(kind), _vp(v), amp.real, amp.imag, int(slots), _vp(X), _vp(st))
    if rc not in (MA_OK, MA_ERR_SINGULAR):
        check(rc)
    return X, st


class BemSweepMulti:
    """ma_bem_sweep_multi_t: the multi-device frequency loop behind a reusable handle -- one BEM plan and one sweep handle per device, made
    once (room_simulator_bem.rs runs one sweep per source position over one mesh). Frequency f on devices[f mod ndev]."""

    def __init__(self, mesh, devices, max_frequencies, slots=3):
        self.devices = [int(d) for d in devices]
        self.n = int((mesh.is_eval == 0).sum()) if mesh.is_eval is not None else mesh.n_elem
        dv = np.ascontiguousarray(self.devices, dtype=np.int32)
        self.h = C.c_void_p()
        L = lib()
        L.ma_bem_sweep_multi_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.ma_bem_sweep_multi_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.ma_bem_sweep_multi_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ma_bem_sweep_multi_destroy.argtypes = [C.c_void_p]
        check(L.ma_bem_sweep_multi_create(C.byref(mesh.c), _vp(dv), len(dv), int(slots), int(max_frequencies), C.byref(self.h)))

    def run(self, frequencies_hz, speed_of_sound=343.0, beta_scale=4.0, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, harmonic=1.0, tau=1.0):
        """ma_bem_sweep_multi_run: (X[n_freq, n], status[n_freq])."""
        f = np.ascontiguousarray(frequencies_hz, dtype=np.float64); v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
        X = np.empty((len(f), self.n), dtype=np.complex128); st = np.zeros(len(f), dtype=np.int32)
        rc = lib().ma_bem_sweep_multi_run(self.h, len(f), _vp(f), float(speed_of_sound), float(harmonic), float(tau), float(beta_scale), int(kind), _vp(v),
                                          amp.real, amp.imag, _vp(X), _vp(st))
        if rc not in (MA_OK, MA_ERR_SINGULAR):
            check(rc)
        return X, st

    def last_timing(self):
        """(wall seconds, frequencies) per device of the last run."""
        secs = np.zeros(len(self.devices)); cnt = np.zeros(len(self.devices), dtype=np.int32)
        check(lib().ma_bem_sweep_multi_last_timing(self.h, _vp(secs), _vp(cnt)))
        return secs, cnt

    def close(self):
        if self.h:
            lib().ma_bem_sweep_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_sweep_multi_timed(mesh, devices, frequencies_hz, speed_of_sound=343.0, beta_scale=4.0, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, slots=3, harmonic=1.0, tau=1.0):
    """ma_bem_solve_sweep_multi_timed: (X, status, device_seconds, device_setup_seconds, device_frequencies)."""
    f = np.ascontiguousarray(frequencies_hz, dtype=np.float64); v = np.ascontiguousarray(vec, dtype=np.float64); amp = complex(amp)
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    nd = int((mesh.is_eval == 0).sum()) if mesh.is_eval is not None else mesh.n_elem
    X = np.empty((len(f), nd), dtype=np.complex128); st = np.zeros(len(f), dtype=np.int32)
    secs = np.zeros(len(dv)); setup = np.zeros(len(dv)); cnt = np.zeros(len(dv), dtype=np.int32)
    rc = lib().ma_bem_solve_sweep_multi_timed(C.byref(mesh.c), _vp(dv), len(dv), len(f), _vp(f), float(speed_of_sound), float(harmonic), float(tau), float(beta_scale),
                                              int(kind), _vp(v), amp.real, amp.imag, int(slots), _vp(X), _vp(st), _vp(secs), _vp(setup), _vp(cnt))
    if rc not in (MA_OK, MA_ERR_SINGULAR):
        check(rc)
    return X, st, secs, setup, cnt


class IluPreconditioner:
    """IluPreconditioner::from_csr (math-solvers/src/preconditioners/ilu.rs): ILU(0), factorised on the host by the library, applied on
    the device; usable wherever a Preconditioner is (gmres_preconditioned, gmres_pipelined)."""

    def __init__(self, csr_operator):
        self.h = C.c_void_p(); self._keep = csr_operator
        check(lib().ma_precond_create_ilu0(csr_operator.h, C.byref(self.h)))
        self.n = csr_operator.n

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.complex128); z = np.empty_like(r)
        check(lib().ma_precond_apply(self.h, _vp(r), _vp(z)))
        return z

    def close(self):
        if self.h:
            lib().ma_precond_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IluFixedPointPreconditioner(IluPreconditioner):
    """IluFixedPointPreconditioner::from_csr(matrix, iterations) (ilu_parallel.rs:397-495; default 3): Jacobi-style sweeps over the ILU(0)
    factors instead of the two triangular solves."""

    def __init__(self, csr_operator, iterations=3):
        self.h = C.c_void_p(); self._keep = csr_operator
        check(lib().ma_precond_create_ilu_fixed_point(csr_operator.h, int(iterations), C.byref(self.h)))
        self.n = csr_operator.n


class AdditiveSchwarzPreconditioner(IluPreconditioner):
    """AdditiveSchwarzPreconditioner::from_csr(matrix, num_subdomains, overlap) (schwarz.rs:84-145): overlapping index blocks, ILU(0) per
    block, weighted sum of the local solutions; all blocks solved together on the device."""

    def __init__(self, csr_operator, num_subdomains=8, overlap=2):
        self.h = C.c_void_p(); self._keep = csr_operator
        check(lib().ma_precond_create_schwarz(csr_operator.h, int(num_subdomains), int(overlap), C.byref(self.h)))
        self.n = csr_operator.n

    def stats(self):
        """(num_subdomains, min_size, max_size, avg_size), schwarz.rs:148-170."""
        a = C.c_int64(); b = C.c_int64(); c = C.c_int64(); d = C.c_double()
        check(lib().ma_precond_schwarz_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value


class AmgPreconditioner:
    """ma_precond_create_amg: AmgPreconditioner::apply (amg.rs:981-1103) on the device over a host-built hierarchy.
    levels: list of dicts {"A": CsrOperator[, "P": CsrOperator.rect, "R": CsrOperator.rect]} (the coarsest without P / R);
    smoother "jacobi" | "l1" | "sgs"; cycle "V" | "W" | "F". The handles are borrowed and kept alive here."""

    def __init__(self, levels, smoother="jacobi", jacobi_weight=0.6667, num_pre_smooth=1, num_post_smooth=1, cycle="V"):
        self._keep = levels
        L = len(levels)
        A = (C.c_void_p * L)(*[lv["A"].h.value for lv in levels])
        Pm = (C.c_void_p * L)(*[lv["P"].h.value if lv.get("P") is not None else None for lv in levels])
        R = (C.c_void_p * L)(*[lv["R"].h.value if lv.get("R") is not None else None for lv in levels])
        sm = {"jacobi": 0, "l1": 1, "sgs": 2}[smoother]; cy = {"V": 0, "W": 1, "F": 2}[cycle]
        self.h = C.c_void_p()
        check(lib().ma_precond_create_amg(L, C.cast(A, C.c_void_p), C.cast(Pm, C.c_void_p), C.cast(R, C.c_void_p), sm, float(jacobi_weight),
                                          int(num_pre_smooth), int(num_post_smooth), cy, C.byref(self.h)))

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.complex128); z = np.empty_like(r)
        check(lib().ma_precond_apply(self.h, _vp(r), _vp(z)))
        return z

    def close(self):
        if self.h:
            lib().ma_precond_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AmgConfig(C.Structure):
    """ma_amg_config_t = AmgConfig (amg.rs:104-146), enums as integers in their declaration order:
    coarsening 0 RugeStuben / 1 Pmis / 2 Hmis, interpolation 0 Standard / 1 Extended / 2 Direct,
    smoother 0 Jacobi / 1 L1Jacobi / 2 SymmetricGaussSeidel / 3 Chebyshev, cycle 0 V / 1 W / 2 F."""
    _fields_ = [("coarsening", C.c_int32), ("interpolation", C.c_int32), ("smoother", C.c_int32), ("cycle", C.c_int32),
                ("strong_threshold", C.c_double), ("max_levels", C.c_int32), ("coarse_size", C.c_int32), ("num_pre_smooth", C.c_int32),
                ("num_post_smooth", C.c_int32), ("jacobi_weight", C.c_double), ("trunc_factor", C.c_double), ("max_interp_elements", C.c_int32),
                ("aggressive_coarsening_levels", C.c_int32)]
    _PRESETS = {"default": 0, "for_bem": 1, "for_fem": 2, "for_parallel": 3, "for_difficult_problems": 4}

    @staticmethod
    def preset(name="default", **overrides):
        c = AmgConfig()
        check(lib().ma_amg_config_preset(AmgConfig._PRESETS[name], C.byref(c)))
        for k, v in overrides.items():
            if k not in dict(AmgConfig._fields_):
                raise KeyError(k)
            setattr(c, k, v)
        return c


def _csr_arrays(h):
    n = C.c_int64(); nnz = C.c_int64(); nc = C.c_int64()
    check(lib().ma_csr_num_rows(h, C.byref(n), C.byref(nnz))); check(lib().ma_csr_num_cols(h, C.byref(nc)))
    rp = np.empty(n.value + 1, dtype=np.int64); ci = np.empty(max(nnz.value, 1), dtype=np.int64); v = np.empty(max(nnz.value, 1), dtype=np.complex128)
    check(lib().ma_csr_get(h, _vp(rp), _vp(ci), _vp(v)))
    return {"shape": (n.value, nc.value), "row_ptrs": rp, "col_indices": ci[:nnz.value], "values": v[:nnz.value]}


class AmgFromCsr:
    """ma_precond_create_amg_from_csr: AmgPreconditioner::from_csr(matrix, config) (amg.rs:276-372) -- hierarchy built on the host inside
    the library with the reference's steps, V / W / F cycle on the device. `csr_operator` (level 0) is borrowed."""

    def __init__(self, csr_operator, config=None):
        self._keep = csr_operator
        self.config = config if config is not None else AmgConfig.preset()
        self.h = C.c_void_p()
        check(lib().ma_precond_create_amg_from_csr(csr_operator.h, C.byref(self.config), C.byref(self.h)))

    def info(self):
        nl = C.c_int32(); gc = C.c_double(); oc = C.c_double(); ms = C.c_double()
        check(lib().ma_precond_amg_info(self.h, C.byref(nl), C.byref(gc), C.byref(oc), C.byref(ms)))
        return {"num_levels": nl.value, "grid_complexity": gc.value, "operator_complexity": oc.value, "setup_time_ms": ms.value}

    def level(self, l):
        """{A, P, R}: each {"shape", "row_ptrs", "col_indices", "values"} read back from the level's handles (P, R None on the coarsest)."""
        a = C.c_void_p(); p_ = C.c_void_p(); r = C.c_void_p()
        check(lib().ma_precond_amg_level(self.h, int(l), C.byref(a), C.byref(p_), C.byref(r)))
        return {"A": _csr_arrays(a), "P": _csr_arrays(p_) if p_ else None, "R": _csr_arrays(r) if r else None}

    def diagnostics(self):
        """AmgDiagnostics (amg.rs:1107-1133)."""
        d = self.info()
        lv = [self.level(l)["A"] for l in range(d["num_levels"])]
        d.update(level_dofs=[a["shape"][0] for a in lv], level_nnz=[len(a["values"]) for a in lv])
        return d

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.complex128); z = np.empty_like(r)
        check(lib().ma_precond_apply(self.h, _vp(r), _vp(z)))
        return z

    def close(self):
        if self.h:
            lib().ma_precond_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gmres_pipelined(op, b, precond=None, x0=None, restart=30, max_iterations=100, tol=1e-6):
    """gmres_pipelined (gmres_pipelined.rs:18-250) on the device: returns (x, GmresInfo); precond None = identity."""
    b = np.ascontiguousarray(b, dtype=np.complex128)
    x = np.empty(op.n, dtype=np.complex128); info = GmresInfo()
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.complex128)
    check(lib().ma_gmres_pipelined(op.h, precond.h if precond is not None else None, _vp(b), _vp(x0a), restart, max_iterations, float(tol), _vp(x), C.byref(info)))
    return x, info
