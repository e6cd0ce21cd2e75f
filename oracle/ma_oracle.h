/* ma_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference algorithm for the acoustic Helmholtz
 * hot path of pierreaubert/math-audio (Rust). Every function cites the
 * reference file:line it follows. Nothing under math_audio_amd/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / timed CPU baseline.
 *
 * Parity pin: the reference is Rust and cannot be built in this image (no
 * cargo/rustc), and its tests hold no stored matrix entries. The oracle is
 * pinned by the reference's own known answers for this path:
 * quadrature weight sums, planar self-term invariants, Mie closed form, LU
 * residuals, CSR/Jacobi known answers and the QA-suite acceptance thresholds
 * (BEM vs Mie L2 < 5 % / 30 %). Entry-level matrix parity is otherwise
 * UNPINNED by the reference ("parity unpinned" at matrix-entry level).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: Rust never contracts
 * a*b+c into an fma, so neither does the oracle).
 */
#ifndef MA_ORACLE_H
#define MA_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } mao_c64;

/* IntegrationResult (reference math-bem/src/core/types.rs:722-734) */
typedef struct {
  mao_c64 g;        /* g_integral          */
  mao_c64 dg_dn;    /* dg_dn_integral      (H,  d/dn_y) */
  mao_c64 dg_dnx;   /* dg_dnx_integral     (H^T, d/dn_x) */
  mao_c64 d2g;      /* d2g_dnxdny_integral (E) */
  mao_c64 rhs;      /* rhs_contribution    */
} mao_integration_result;

/* Subelement (reference integration/singular.rs:468-481) */
typedef struct {
  double xi_center, eta_center, factor;
  int gauss_order;
  int has_tri;          /* tri_vertices: Some/None */
  double tri[6];        /* xi0,eta0, xi1,eta1, xi2,eta2 */
} mao_subelement;

#define MAO_MAX_SUBELEMENTS 110

/* ---- quadrature (gauss.rs:15-105) ---- */
int mao_gauss_legendre(int order, double* x, double* w);          /* returns count */
int mao_triangle_quadrature(int order, double* xi_eta_w);         /* returns count; weights already x0.5 */
int mao_quad_quadrature(int order, double* xi_eta_w);             /* returns count */

/* ---- physics (types.rs:39-219) ---- */
double  mao_wave_number(double frequency, double speed_of_sound);
mao_c64 mao_burton_miller_beta(double k, double harmonic, double tau);
mao_c64 mao_burton_miller_beta_scaled(double k, double harmonic, double tau, double scale);
mao_c64 mao_burton_miller_beta_adaptive(double k, double harmonic, double tau, double radius, double* scale_out);

/* ---- meshes (mesh/generators.rs:29-228, 434-602) ---- */
void mao_icosphere_counts(int subdivisions, int* n_nodes, int* n_elem);
void mao_icosphere(double radius, int subdivisions, double* nodes, int* conn4);
void mao_uv_sphere_counts(int n_theta, int n_phi, int* n_nodes, int* n_elem);
void mao_uv_sphere(double radius, int n_theta, int n_phi, double* nodes, int* conn4);
/* conn4: n_elem x 4, 4th = -1 for triangles. Fills center[3n], normal[3n], area[n]. */
void mao_element_geometry(int n_elem, const double* nodes, const int* conn4,
                          double* center, double* normal, double* area);

/* ---- panel integrals ---- */
int mao_generate_subelements(const double* x, const double* coords /*nn*3*/, int num_nodes,
                             double area, mao_subelement* out /*cap 110*/);
void mao_regular_integration(const double* x, const double* nx, const double* coords, int num_nodes,
                             double area, double k, double harmonic, double tau,
                             const mao_c64* bc_values, int bc_len, int bc_type, int compute_rhs,
                             mao_integration_result* out);
void mao_singular_integration(const double* x, const double* nx, const double* coords, int num_nodes,
                              double k, double harmonic, double tau,
                              const mao_c64* bc_values, int bc_len, int bc_type, int compute_rhs,
                              mao_integration_result* out);
void mao_singular_integration_with_params(const double* x, const double* nx, const double* coords, int num_nodes,
                              double k, double harmonic, double tau,
                              const mao_c64* bc_values, int bc_len, int bc_type, int compute_rhs,
                              int edge_gauss_order, int subelement_gauss_order, int edge_sections,
                              int subtriangles_per_section, mao_integration_result* out);

/* ---- assembly (assembly/tbem.rs:96-345) ----
 * bc_type: 0 velocity, 1 pressure, 2 other. bc_values: n_elem*4 (first bc_len[e] used).
 * Rows [row_begin,row_end) of source elements are assembled (whole range = full matrix);
 * A is the full N x N row-major buffer (rows outside the range untouched), rhs length N.
 * nthreads > 1 distributes source rows over threads (what rayon's par_iter over rows
 * would do, tbem.rs:382) with the SEQUENTIAL function's arithmetic (tbem.rs:96-222). */
int mao_build_tbem_system_with_beta(int n_elem, const double* nodes, const int* conn4,
        const double* center, const double* normal, const double* area,
        const int* dof, const unsigned char* bc_type, const mao_c64* bc_values, const int* bc_len,
        const unsigned char* is_eval,
        double k, double harmonic, double tau, double beta_re, double beta_im,
        mao_c64* A, mao_c64* rhs, int num_dofs, int row_begin, int row_end, int nthreads);
/* packed != 0: A is a (row_end - row_begin) x num_dofs strip and rhs has row_end - row_begin entries (sampled rows of
 * systems too large to hold whole, e.g. 50 172 panels) */
int mao_build_tbem_rows(int n_elem, const double* nodes, const int* conn4,
        const double* center, const double* normal, const double* area,
        const int* dof, const unsigned char* bc_type, const mao_c64* bc_values, const int* bc_len,
        const unsigned char* is_eval,
        double k, double harmonic, double tau, double beta_re, double beta_im,
        mao_c64* A, mao_c64* rhs, int num_dofs, int row_begin, int row_end, int nthreads, int packed);

/* ---- incident field (incident.rs:93-342); kind 0 plane wave (dir, amplitude), 1 point source (pos, strength) */
void mao_incident_pressure(int kind, const double* vec3, mao_c64 amp, int n, const double* points, double k, mao_c64* out);
void mao_incident_normal_derivative(int kind, const double* vec3, mao_c64 amp, int n, const double* points,
                                    const double* normals, double k, mao_c64* out);
void mao_compute_rhs_with_beta(int kind, const double* vec3, mao_c64 amp, int n, const double* centers,
                               const double* normals, double k, double tau, mao_c64 beta, mao_c64* rhs);

/* ---- dense solve (math-solvers/src/direct/lu.rs:142-153 -> LAPACK zgesv semantics) ----
 * A row-major N x N (destroyed: holds LU), b in/out. ipiv (0-based, LAPACK style) may be NULL.
 * Returns 0, 1 = singular (LuError::SingularMatrix), 2 = dimension mismatch. */
int mao_zgesv(int n, mao_c64* A, mao_c64* b, int* ipiv, int nthreads);
/* pure-Rust fallback path restated literally (lu.rs:83-137 + 38-78), including its pivot replay */
int mao_lu_solve_fallback(int n, const mao_c64* A, const mao_c64* b, mao_c64* x);

/* ---- Mie oracle (math-wave/src/analytical/solutions_3d.rs:56-273) ---- */
double mao_spherical_bessel_j(int n, double x);
double mao_spherical_bessel_y(int n, double x);
double mao_legendre_p(int n, double x);
double mao_sphere_rcs_3d(double k, double radius, int num_terms);
void   mao_sphere_scattering_3d(double k, double radius, int num_terms, int nr, const double* r,
                                int nt, const double* theta, mao_c64* pressure /* nr*nt */);

/* ---- field post-processing (postprocess/pressure.rs:81-258) ---- */
void mao_compute_scattered_field(int n_eval, const double* eval_points, int n_elem, const double* nodes,
        const int* conn4, const unsigned char* is_eval, const mao_c64* surface_pressure,
        const mao_c64* surface_velocity /* may be NULL */, double k, double harmonic, mao_c64* out);

/* ---- CSR / smoothers / GMRES (math-solvers) ---- */
void mao_csr_matvec(int n_rows, const long long* row_ptr, const long long* col, const mao_c64* val,
                    const mao_c64* x, mao_c64* y, int nthreads);
void mao_helmholtz_values(long long nnz, const double* K, const double* M, double k_re, double k_im, mao_c64* val);
void mao_amg_jacobi(int n, const long long* row_ptr, const long long* col, const mao_c64* val,
                    mao_c64* x, const mao_c64* b, double omega, int sweeps, int nthreads);
void mao_amg_l1_jacobi(int n, const long long* row_ptr, const long long* col, const mao_c64* val,
                    mao_c64* x, const mao_c64* b, int sweeps, int nthreads);
void mao_amg_sym_gauss_seidel(int n, const long long* row_ptr, const long long* col, const mao_c64* val,
                    mao_c64* x, const mao_c64* b, int sweeps);
/* math-fem geometric-MG smoothers on COO triplets (multigrid/smoother.rs:44-176); kind 0 GS, 1 Jacobi, 2 SGS */
void mao_fem_smooth(int n, long long nnz, const long long* rows, const long long* cols, const mao_c64* vals,
                    mao_c64* x, const mao_c64* b, int kind, int iterations, double omega);
void mao_fem_residual(int n, long long nnz, const long long* rows, const long long* cols, const mao_c64* vals,
                    const mao_c64* x, const mao_c64* b, mao_c64* r);
/* dense or CSR operator GMRES (iterative/gmres.rs:105-277). op_kind 0 dense row-major, 1 CSR. */
typedef struct { int iterations, restarts, converged; double residual; } mao_gmres_info;
void mao_gmres(int n, int op_kind, const mao_c64* dense, const long long* row_ptr, const long long* col,
               const mao_c64* val, const mao_c64* b, const mao_c64* x0 /*NULL ok*/, int restart, int max_iterations,
               double tol, mao_c64* x, mao_gmres_info* info);

/* gmres_preconditioned(_with_guess) (gmres.rs:282-585) on a CSR operator with a one-level smoother as the
 * Preconditioner: pkind 0 identity, 1 Jacobi(omega, sweeps), 2 l1-Jacobi(sweeps), each applied from z = 0. */
void mao_gmres_preconditioned(int n, const long long* row_ptr, const long long* col, const mao_c64* val, int pkind, double omega, int sweeps,
                              const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol, mao_c64* x, mao_gmres_info* info);

/* AmgPreconditioner::v_cycle / apply (preconditioners/amg.rs:981-1065, 1068-1103) over a hierarchy given level by level: operator
 * a_*[l] (n[l] x n[l]); prolongation p_*[l] (n[l] x n[l+1]) and restriction r_*[l] (n[l+1] x n[l]) for l < nlevels-1 (NULL on the
 * coarsest). smoother 0 Jacobi(jacobi_weight), 1 l1-Jacobi, 2 symmetric Gauss-Seidel; cycle 0 V, 1 W, 2 F. */
typedef struct {
  int nlevels; const int* n;
  const long long* const* a_rp; const long long* const* a_col; const mao_c64* const* a_val;
  const long long* const* p_rp; const long long* const* p_col; const mao_c64* const* p_val;
  const long long* const* r_rp; const long long* const* r_col; const mao_c64* const* r_val;
  int smoother; double jacobi_weight; int num_pre_smooth, num_post_smooth, cycle;
} mao_amg_hierarchy;
void mao_amg_apply(const mao_amg_hierarchy* H, const mao_c64* r, mao_c64* z);
void mao_gmres_amg(const mao_amg_hierarchy* H, const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol,
                   mao_c64* x, mao_gmres_info* info);

/* gmres_pipelined (iterative/gmres_pipelined.rs:18-250): op_kind 0 dense / 1 CSR; pkind 0 identity, 1 Jacobi(omega, sweeps),
 * 2 l1-Jacobi(sweeps) (CSR operators), the _amg form preconditions with the hierarchy's V-cycle */
void mao_gmres_pipelined(int n, int op_kind, const mao_c64* dense, const long long* row_ptr, const long long* col, const mao_c64* val,
                         int pkind, double omega, int sweeps, const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol,
                         mao_c64* x, mao_gmres_info* info);
void mao_gmres_pipelined_amg(const mao_amg_hierarchy* H, const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol,
                             mao_c64* x, mao_gmres_info* info);

/* ---- single-level fast multipole operator (assembly/slfmm.rs): build_slfmm_system + SlfmmSystem::matvec / matvec_transpose /
 * extract_near_field_matrix. Clusters arrive as CSR-style lists (element_indices, near_clusters, far_clusters, centres). */
void mao_fmm_near_block(const double* nodes, const int* conn, const double* center, const double* normal, const double* area,
                        int ns, const int* src_idx, int nf, const int* fld_idx, int is_self, double k, double harmonic, double tau, mao_c64* out);
typedef struct mao_slfmm mao_slfmm;
void mao_spherical_hankel_first_kind(int order, double x, double harmonic, mao_c64* result);
int mao_unit_sphere_quadrature(int n_theta, int n_phi, double* coords, double* weights);
mao_slfmm* mao_slfmm_build(int n_elem, const double* nodes, const int* conn, const double* center, const double* normal, const double* area,
                           const int* dof, const unsigned char* bc_type, int n_clusters, const double* cluster_center, const int* elem_ptr, const int* elem_idx,
                           const int* near_ptr, const int* near_idx, const int* far_ptr, const int* far_idx,
                           double k, double harmonic, double tau, int n_theta, int n_phi, int n_terms);
void mao_slfmm_free(mao_slfmm* S);
void mao_slfmm_matvec(const mao_slfmm* S, int transpose, const mao_c64* x, mao_c64* y);
void mao_slfmm_near_matrix(const mao_slfmm* S, mao_c64* A);

/* ---- room-acoustics collocation assembly (room_acoustics/solver.rs:448-493) ---- */
void mao_room_build_matrix(int n_elem, const double* center, const double* normal, const double* area,
                           double k, mao_c64* A, int nthreads);
/* the rest of the room path: element data (solver.rs:38-122, 600-611), adaptive assembly (:500-597), incident
 * derivative (:638-678), field pressure (:687-748) */
void mao_room_element_data(int n_elem, const double* nodes, const int* conn, double* center, double* normal, double* area, double* charlen);
void mao_room_build_matrix_adaptive(int n_elem, const double* nodes, const int* conn, double k, int use_adaptive, mao_c64* A);
void mao_room_incident_derivative(int n, const double* center, const double* normal, int nsrc, const double* src_pos, const double* amp,
                                  int per_point, double k, mao_c64* out);
void mao_room_field_pressure(int n, const double* center, const double* normal, const double* area, const mao_c64* surface_pressure,
                             int nsrc, const double* src_pos, const double* amp, int per_point, int npts, const double* pts, double k, mao_c64* out);

#ifdef __cplusplus
}
#endif
#endif
