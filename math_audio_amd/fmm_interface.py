"""Host-side mirror of the solver wrappers and mesh utilities of math-bem/src/core/solver/fmm_interface.rs (:356-600). The operators
themselves (DenseOperator, SlfmmOperator, MlfmmOperator, CsrOperator, :25-170) are `ma.LinearOperator.dense / slfmm / mlfmm / csr`;
every solve below runs on the device through the C-ABI, with the reference's argument meaning and its quirks kept:
`solve_with_ilu` and `solve_with_ilu_operator` run CGS WITHOUT the preconditioner they are named after (:389-439), the hierarchical
FMM preconditioner is the identity (:326-355) and `SparseNearfieldIlu` the diagonal of the self blocks (:249-297)."""
import math
import numpy as np
import math_audio_amd as ma


class KrylovConfig:
    """CgsConfig / BiCgstabConfig (max_iterations, tolerance) and GmresConfig (+ restart) as the wrappers read them."""

    def __init__(self, max_iterations=1000, tolerance=1e-6, restart=30, print_interval=0):
        self.max_iterations, self.tolerance, self.restart, self.print_interval = int(max_iterations), float(tolerance), int(restart), int(print_interval)


def solve_cgs(operator, b, config):             # :360-366
    return ma.cgs(operator, b, config.max_iterations, config.tolerance)


def solve_bicgstab(operator, b, config):        # :369-375
    return ma.bicgstab(operator, b, config.max_iterations, config.tolerance)


def solve_gmres(operator, b, config):           # :378-384
    return ma.gmres(operator, b, restart=config.restart, max_iterations=config.max_iterations, tol=config.tolerance)


def solve_with_ilu(matrix, b, config):          # :389-418: builds an ILU it does not use, runs plain CGS on the dense operator
    op = ma.LinearOperator.dense(matrix)
    try:
        return ma.cgs(op, b, config.max_iterations, config.tolerance)
    finally:
        op.close()


def solve_with_ilu_operator(operator, nearfield_matrix, b, config):     # :420-439: plain CGS, the near-field matrix is ignored
    return ma.cgs(operator, b, config.max_iterations, config.tolerance)


def solve_tbem_with_ilu(matrix, b, config):     # :441-447
    return solve_with_ilu(matrix, b, config)


def csr_from_dense(matrix, threshold=1e-15):
    """CsrMatrix::from_dense(matrix, threshold) (sparse/csr.rs:104-132): entries of norm > threshold, row by row."""
    A = np.asarray(matrix, dtype=np.complex128)
    keep = np.sqrt(A.real * A.real + A.imag * A.imag) > threshold
    rp = np.concatenate(([0], np.cumsum(keep.sum(axis=1)))).astype(np.int64)
    rows, cols = np.nonzero(keep)
    return rp, cols.astype(np.int64), A[rows, cols]


def gmres_solve_with_ilu_operator(operator, nearfield_matrix, b, config):   # :462-474
    rp, ci, v = csr_from_dense(nearfield_matrix)
    csr = ma.CsrOperator(rp, ci, values=v)
    pre = ma.IluPreconditioner(csr)
    try:
        return ma.gmres_preconditioned(operator, pre, b, restart=config.restart, max_iterations=config.max_iterations, tol=config.tolerance)
    finally:
        pre.close(); csr.close()


def gmres_solve_with_ilu(matrix, b, config):    # :450-459
    op = ma.LinearOperator.dense(matrix)
    try:
        return gmres_solve_with_ilu_operator(op, matrix, b, config)
    finally:
        op.close()


def gmres_solve_tbem_with_ilu(matrix, b, config):   # :477-483
    return gmres_solve_with_ilu(matrix, b, config)


def gmres_solve_fmm_hierarchical(fmm_operator, b, config):
    """gmres_solve_with_hierarchical_precond / gmres_solve_fmm_hierarchical (:490-513): HierarchicalFmmPreconditioner::apply is
    r.clone() (:350-354), so this is GMRES left-preconditioned by the identity = plain GMRES (same iterates, same tolerance base ||b||)."""
    return ma.gmres(fmm_operator, b, restart=config.restart, max_iterations=config.max_iterations, tol=config.tolerance)


gmres_solve_with_hierarchical_precond = gmres_solve_fmm_hierarchical


def gmres_solve_fmm_batched(fmm_operator, b, config):   # :516-524
    return ma.gmres(fmm_operator, b, restart=config.restart, max_iterations=config.max_iterations, tol=config.tolerance)


def gmres_solve_fmm_batched_with_ilu(fmm_operator, b, config):   # :527-538: ILU(0) of extract_near_field_matrix()
    return gmres_solve_with_ilu_operator(fmm_operator, fmm_operator.slfmm_near_matrix(), b, config)


def sparse_nearfield_ilu(fmm_operator):
    """SparseNearfieldIlu::from_slfmm (:249-297): z_i = r_i / d_i with d the diagonal of the self blocks (1 where its norm <= 1e-15):
    the diagonal preconditioner of the operator."""
    return ma.Preconditioner(fmm_operator, "diagonal")


def recommended_mesh_resolution(frequency, speed_of_sound, elements_per_wavelength):    # :544-551
    return float(elements_per_wavelength) / (speed_of_sound / frequency)


def mesh_resolution_for_frequency_range(min_freq, max_freq, speed_of_sound, elements_per_wavelength):   # :554-561
    return recommended_mesh_resolution(max_freq, speed_of_sound, elements_per_wavelength)


def estimate_element_count(room_dimensions, mesh_resolution):   # :564-570
    w, d, h = room_dimensions
    surface_area = 2.0 * (w * d + w * h + d * h)
    element_size = 1.0 / mesh_resolution
    return int(math.ceil(surface_area / (element_size * element_size)))


class AdaptiveMeshConfig:                       # :573-603
    def __init__(self, base_resolution, source_refinement, source_refinement_radius):
        self.base_resolution, self.source_refinement, self.source_refinement_radius = float(base_resolution), float(source_refinement), float(source_refinement_radius)

    @staticmethod
    def for_frequency_range(min_freq, max_freq):
        return AdaptiveMeshConfig(mesh_resolution_for_frequency_range(min_freq, max_freq, 343.0, 6), 1.5, 0.5)

    @staticmethod
    def from_resolution(resolution):
        return AdaptiveMeshConfig(resolution, 1.0, 0.0)
