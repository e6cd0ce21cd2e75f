/* mathaudio_hip.h — C-ABI of libmathaudio_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of pierreaubert/math-audio: the math-bem dense
 * Burton–Miller (TBEM) assembly, the math-solvers dense complex solve it feeds, and the
 * CSR SpMV / Jacobi smoothers of the FEM side. The reference has no FFI today; each entry
 * point below names the Rust function or trait method it replaces (paths relative to the
 * reference root). The Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions:
 *  - Complex64 is #[repr(C)] {re: f64, im: f64} == ma_c64 == C `double _Complex`.
 *  - Dense matrices are ndarray C-order: row-major, A[i*n + j].
 *  - All host pointers are caller-owned borrows; the library never retains or frees them.
 *  - Device memory lives behind opaque handles (or caller-provided device pointers in the
 *    *_dev variants, passed as void*; streams are hipStream_t passed as void*).
 *  - No unwinding across the boundary: every call returns an int status; the text of the last
 *    error on the calling thread is available from ma_last_error_string().
 *  - Every entry point is re-entrant; a handle may be used from one thread at a time.
 *  - There is NO CPU fallback: without a usable gfx950 device every compute call fails with
 *    MA_ERR_NO_DEVICE.
 */
#ifndef MATHAUDIO_HIP_H
#define MATHAUDIO_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } ma_c64;

/* status codes; the mapping onto the reference's error enums is part of the contract */
enum {
  MA_OK              = 0,
  MA_ERR_SINGULAR    = 1,  /* -> LuError::SingularMatrix            (math-solvers/src/direct/lu.rs:16-21)  */
  MA_ERR_DIM         = 2,  /* -> LuError::DimensionMismatch / SolverError::DimensionMismatch               */
  MA_ERR_INVALID     = 3,  /* null pointer, negative size, bad enum  -> BemError::InvalidConfiguration      */
  MA_ERR_UNSUPPORTED = 4,  /* input class not handled by the device path yet (listed per function)         */
  MA_ERR_HIP         = 5,  /* a HIP runtime call failed; text in ma_last_error_string()                    */
  MA_ERR_NO_DEVICE   = 6,  /* no gfx950 device visible                                                     */
  MA_ERR_NOMEM       = 7,  /* device or host allocation failed                                             */
  MA_ERR_RETRY       = 8   /* an LU plan in the optimistic speculation mode met a panel it cannot vouch for: the system
                              must be solved again in the verified mode (ma_lu_plan_set_speculation). Never returned
                              by the drop-in entries (ma_zgesv, ma_lu_solve, ma_lu_factorize, the sweeps): they retry. */
};

const char* ma_last_error_string(void);
const char* ma_version(void);
int ma_device_count(int* count);                 /* MA_OK and *count = 0 on a host without GPUs */

/* ------------------------------------------------------------------------------------------
 * Mesh and physics — flattened `&[Element]` + `nodes: Array2<f64>` (math-bem/src/core/types.rs:330-351)
 * A Rust shim fills this from the AoS `Element`s (connectivity, center, normal, area,
 * dof_addresses[0], boundary_condition, property).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t        n_nodes;
  const double*  nodes;      /* n_nodes*3, row-major (Array2<f64>)                                   */
  int32_t        n_elem;
  const int32_t* conn;       /* n_elem*4 node indices; 4th = -1 for Tri3                             */
  const double*  center;     /* n_elem*3  Element.center (collocation point)                         */
  const double*  normal;     /* n_elem*3  Element.normal (stored, outward-flipped: generators.rs:590) */
  const double*  area;       /* n_elem    Element.area                                               */
  const int32_t* dof;        /* n_elem    Element.dof_addresses[0]; must enumerate 0..num_dofs-1 over
                                          the non-evaluation elements                                */
  const uint8_t* bc_type;    /* n_elem    0 Velocity, 1 Pressure, 2 other (tbem.rs:234-244)          */
  const ma_c64*  bc_values;  /* n_elem*4  BoundaryCondition values (first bc_len[e] are used); may be
                                          NULL = all zero                                            */
  const int32_t* bc_len;     /* n_elem    length of the BC Vec (>=1); may be NULL = all 1            */
  const uint8_t* is_eval;    /* n_elem    ElementProperty::Evaluation flag; may be NULL = none       */
} ma_mesh_t;

typedef struct {
  double wave_number;        /* PhysicsParams.wave_number   (types.rs:24)  */
  double harmonic_factor;    /* PhysicsParams.harmonic_factor (+1)         */
  double tau;                /* +1 exterior, -1 interior    (types.rs:45)  */
  double gamma;              /* PhysicsParams::gamma() = 1  (types.rs:216) */
} ma_physics_t;

/* ------------------------------------------------------------------------------------------
 * TBEM assembly.
 * Replaces: build_tbem_system_with_beta(&[Element], &Array2<f64>, &PhysicsParams, Complex64)
 *           -> TbemSystem        math-bem/src/core/assembly/tbem.rs:96-222
 * (and through it build_tbem_system :45, _bounded :64, _scaled :85; callers bem_solver.rs:367,
 *  bin/qa_suite.rs:228,355).
 * A: num_dofs*num_dofs row-major [[source_dof, field_dof]] (tbem.rs:340), overwritten.
 * rhs: num_dofs, overwritten (TbemSystem.rhs: free-term shares and rhs_contributions of the elements that carry
 *      non-zero boundary values, tbem.rs:273-304, regular.rs:157-177, singular.rs:360-392; zero for rigid scatterers).
 * Tri3 and Quad4 elements (conn row = 3 node ids and -1, or 4 node ids), also mixed.
 * ------------------------------------------------------------------------------------------ */
int ma_bem_assemble_tbem(const ma_mesh_t* mesh, const ma_physics_t* physics,
                         double beta_re, double beta_im, ma_c64* A, ma_c64* rhs);

/* Device-resident form for frequency sweeps: geometry, the near-pair list and all
 * frequency-independent tables stay in HBM behind the plan. */
typedef struct ma_bem_plan ma_bem_plan_t;
int ma_bem_plan_create(const ma_mesh_t* mesh, int device, ma_bem_plan_t** out);
int ma_bem_plan_destroy(ma_bem_plan_t* plan);
int ma_bem_plan_num_dofs(const ma_bem_plan_t* plan, int32_t* num_dofs);
int ma_bem_plan_device(const ma_bem_plan_t* plan, int* device);   /* the device the plan's arrays live on */
int ma_bem_plan_num_near_pairs(const ma_bem_plan_t* plan, int64_t* n);
/* d_A (num_dofs^2 ma_c64) and d_rhs (num_dofs) are DEVICE pointers; work is enqueued on `stream`. */
int ma_bem_plan_assemble_dev(ma_bem_plan_t* plan, const ma_physics_t* physics, double beta_re, double beta_im,
                             void* d_A, void* d_rhs, void* stream);
/* build_tbem_system_with_beta (tbem.rs:96-222) for nf (1..16) wavenumbers of the plan's mesh in one call -- the next systems of a
 * frequency sweep (room_simulator_bem.rs:329 walks the frequencies one by one): physics[f], beta[f] -> d_A[f], d_rhs[f] (device
 * pointers, nf distinct matrices). The far pairs of up to three systems share one pass over the quadrature points. */
int ma_bem_plan_assemble_multi_dev(ma_bem_plan_t* plan, int32_t nf, const ma_physics_t* physics, const double* beta_re, const double* beta_im,
                                   void* const* d_A, void* const* d_rhs, void* stream);
/* The same assembly in `nparts` pieces issued one at a time (part = 0 .. nparts - 1, in that order, all on one stream): every
 * part a slice of the far pairs' rows, the first also the right-hand sides' preparation, the last also the near and self pairs.
 * For a sweep that assembles ahead and feeds the parts to its stream where it would otherwise wait (sweep_plan.hip). */
int ma_bem_plan_assemble_multi_part_dev(ma_bem_plan_t* plan, int32_t nf, const ma_physics_t* physics, const double* beta_re, const double* beta_im,
                                        void* const* d_A, void* const* d_rhs, int32_t part, int32_t nparts, void* stream);

/* Incident field RHS.
 * Replaces: IncidentField::compute_rhs_with_beta(centers, normals, physics, beta)
 *           math-bem/src/core/incident.rs:317-342 (PlaneWave / PointSource :93-280).
 * kind 0: plane wave, vec3 = unit direction; kind 1: point source, vec3 = position.
 * d_rhs[i] (+)= -(gamma p_inc + beta tau dp_inc/dn); accumulate != 0 adds to d_rhs (the QA suite's
 * `&system.rhs + &rhs`, bin/qa_suite.rs:247). */
int ma_bem_plan_incident_rhs_dev(ma_bem_plan_t* plan, const ma_physics_t* physics, double beta_re, double beta_im,
                                 int kind, const double* vec3, double amp_re, double amp_im,
                                 int accumulate, void* d_rhs, void* stream);
int ma_bem_incident_rhs(int n, const double* centers, const double* normals, const ma_physics_t* physics,
                        double beta_re, double beta_im, int kind, const double* vec3,
                        double amp_re, double amp_im, ma_c64* rhs);
/* IncidentField::evaluate_pressure (incident.rs:93-166) / ::evaluate_normal_derivative (:177-280) at n points; either
 * output may be NULL (normals may be NULL when dpdn_out is). compute_total_field (postprocess/pressure.rs:273-311) is
 * this p plus ma_bem_plan_scattered_field. */
int ma_bem_incident_evaluate(int n, const double* points, const double* normals, const ma_physics_t* physics, int kind, const double* vec3,
                             double amp_re, double amp_im, ma_c64* p_out, ma_c64* dpdn_out);

/* ------------------------------------------------------------------------------------------
 * Dense complex solve.
 * Replaces: lu_solve(&Array2<Complex64>, &Array1<Complex64>) -> Result<Array1<_>, LuError>
 *           math-solvers/src/direct/lu.rs:142-153 (native: LAPACK zgesv through ndarray-linalg).
 * A_rowmajor is destroyed (holds the LU factors), b_inout receives x. ipiv may be NULL.
 * Returns MA_OK, MA_ERR_SINGULAR or MA_ERR_DIM.
 * ------------------------------------------------------------------------------------------ */
int ma_zgesv(int32_t n, ma_c64* A_rowmajor, ma_c64* b_inout, int32_t* ipiv_or_null);

/* The frequency loop of the BEM drivers (math-bem/bin/room_simulator_bem.rs:329-360; BemSolver::solve, bem_solver.rs:355-480)
 * as one device-resident call: per frequency k = 2 pi f / c, beta = i h scale / k (burton_miller_beta_scaled), TBEM assembly,
 * incident right-hand side (kind 0 plane wave: vec3 = direction; 1 point source: vec3 = position), dense solve. `slots`
 * systems (1..4, 0 = default 3) are kept in HBM and factored as one interleaved batch; only the n_freq x num_dofs
 * solutions come back. status_or_null[f]: MA_OK or MA_ERR_SINGULAR per frequency. */
int ma_bem_solve_sweep(ma_bem_plan_t* plan, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                       double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im, int32_t slots,
                       ma_c64* X_out, int32_t* status_or_null);

/* The loop behind a reusable handle (round 4): what ma_bem_solve_sweep allocates per call -- the LU plan and its streams, the
 * `slots` systems in flight, the spare systems of the assembly-ahead (2 x 3 more matrices when they fit), the parked solutions --
 * is allocated once for up to max_frequencies frequencies per run and reused by every ma_bem_sweep_run. The reference's driver
 * sweeps once per source (room_simulator_bem.rs:243-256 builds the mesh once, :328-360 loops over the frequencies; a Rust caller
 * holds the handle in the struct that holds its mesh and implements Drop with ma_bem_sweep_destroy). The plan is borrowed.
 * ma_bem_sweep_run: arguments as ma_bem_solve_sweep; X_out_or_null = NULL leaves the solutions on the device
 * (ma_bem_sweep_solutions_dev: row i = the i-th frequency of the last run, valid until the next run). Runs of one handle must
 * not overlap; handles of different plans / devices are independent (one host thread per device: ma_bem_solve_sweep_multi). */
typedef struct ma_bem_sweep ma_bem_sweep_t;
typedef struct ma_lu_plan ma_lu_plan_t;           /* workspace for device-resident solves of size n (ma_lu_plan_create below) */
int ma_bem_sweep_create(ma_bem_plan_t* plan, int32_t slots, int32_t max_frequencies, ma_bem_sweep_t** out);
/* the same with the pivoting of the sweep's LU plan chosen by the caller (MA_LU_PIVOT_PARTIAL / MA_LU_PIVOT_TOURNAMENT, see
 * ma_lu_plan_create_pivoting; -1 = the sweep's default, which is what ma_bem_sweep_create and ma_bem_solve_sweep[_multi] use:
 * tournament, MA_SWEEP_PIVOTING=partial overrides) */
int ma_bem_sweep_create_pivoting(ma_bem_plan_t* plan, int32_t slots, int32_t max_frequencies, int32_t pivoting, ma_bem_sweep_t** out);
int ma_bem_sweep_destroy(ma_bem_sweep_t* sweep);
int ma_bem_sweep_run(ma_bem_sweep_t* sweep, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                     double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im,
                     ma_c64* X_out_or_null, int32_t* status_or_null);
int ma_bem_sweep_solutions_dev(ma_bem_sweep_t* sweep, void** d_X, int32_t* count);
/* device-to-device copy on the current device, complete on return (for callers that mix the library's device pointers with their own buffers) */
int ma_device_copy(void* d_dst, const void* d_src, int64_t bytes, void* stream);
/* Measurement (bench.py; no reference counterpart): with timing on, HIP events bracket the run, every piece of assembly and every
 * big trailing update ON THE STREAM THEY RUN ON. ma_bem_sweep_last_timing, out8 = { wall seconds of the last run (entry to
 * solutions on the host), device ms between the run's first and last event, ms summed over the assembly pieces (incident
 * right-hand sides included), pieces, ms summed over the big trailing updates, their launches, their algorithmic flops (8 M N K
 * each), frequencies }. ma_bem_sweep_info: slots, blocks per factorisation, rounds between two slots' starts, systems assembled
 * ahead (1 = none), 1 if the staged pipeline runs (any pointer may be NULL). The LU plan and the stream are borrowed. */
int ma_bem_sweep_set_timing(ma_bem_sweep_t* sweep, int enable);
int ma_bem_sweep_last_timing(ma_bem_sweep_t* sweep, double* out8);
int ma_bem_sweep_info(ma_bem_sweep_t* sweep, int32_t* slots, int32_t* blocks, int32_t* spacing, int32_t* systems_ahead, int32_t* staged);
int ma_bem_sweep_lu_plan(ma_bem_sweep_t* sweep, ma_lu_plan_t** plan);
int ma_bem_sweep_stream(ma_bem_sweep_t* sweep, void** stream);

/* The same loop over the GPUs of one node (BASELINE.json configs[2]): frequency f is solved on devices[f mod ndev]
 * (ma_sweep_owner), one host thread, BEM plan, LU plan and stream per device, no collective on the data path; X_out
 * (n_freq x num_dofs) and status_or_null are indexed by f as above. A Rust caller replaces the `for freq` loop of
 * math-bem/bin/room_simulator_bem.rs:329 by this one call. ma_bem_solve_sweep itself runs on its plan's device, whatever
 * device the calling thread has selected, and leaves the thread's current device unchanged. */
int ma_bem_solve_sweep_multi(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz,
                             double speed_of_sound, double harmonic_factor, double tau, double beta_scale, int incident_kind,
                             const double* incident_vec3, double amp_re, double amp_im, int32_t slots, ma_c64* X_out, int32_t* status_or_null);
/* The same with a per-device account (any of the three may be NULL): device_seconds[d] = wall time of device d's sweep,
 * device_setup_seconds[d] = its plan creation (geometry upload, near-pair list), device_frequencies[d] = frequencies it solved. */
int ma_bem_solve_sweep_multi_timed(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz,
                                   double speed_of_sound, double harmonic_factor, double tau, double beta_scale, int incident_kind,
                                   const double* incident_vec3, double amp_re, double amp_im, int32_t slots, ma_c64* X_out, int32_t* status_or_null,
                                   double* device_seconds, double* device_setup_seconds, int32_t* device_frequencies);
int ma_sweep_owner(int32_t frequency_index, int32_t ndev);      /* index into devices[] of the owner of a frequency */
/* The multi-device loop behind a reusable handle (round 5): one BEM plan and one sweep handle per device -- geometry, LU plan, streams, the
 * systems in flight, the spares of the assembly-ahead, the parked solutions -- made once for runs of up to max_frequencies frequencies
 * (over all devices) and kept across ma_bem_sweep_multi_run calls (arguments as ma_bem_solve_sweep_multi; frequency f on
 * devices[f mod ndev], one host thread per device, no collective). The reference's driver sweeps once per source position over one mesh
 * (room_simulator_bem.rs:243-256, :328-360): a Rust caller keeps this handle beside its mesh and implements Drop with _destroy.
 * ma_bem_solve_sweep_multi[_timed] is create + run + destroy. _last_timing: wall seconds and frequencies per device of the last run. */
typedef struct ma_bem_sweep_multi ma_bem_sweep_multi_t;
int ma_bem_sweep_multi_create(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t slots, int32_t max_frequencies, ma_bem_sweep_multi_t** out);
int ma_bem_sweep_multi_run(ma_bem_sweep_multi_t* handle, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                           double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im, ma_c64* X_out, int32_t* status_or_null);
int ma_bem_sweep_multi_last_timing(ma_bem_sweep_multi_t* handle, double* device_seconds, int32_t* device_frequencies);
int ma_bem_sweep_multi_destroy(ma_bem_sweep_multi_t* handle);
/* the order in which the staged frequency loop (room_simulator_bem.rs:328-360 as `slots` staggered factorisations of `blocks` blocks, `spacing`
 * rounds apart) begins its n_freq frequencies: order_out[q] = frequency of the q-th begin. Systems are assembled ahead only when this is the
 * identity. Pure host arithmetic (no device). */
int ma_sweep_begin_order(int32_t blocks, int32_t slots, int32_t spacing, int32_t n_freq, int32_t* order_out);

/* The same solve with the reference's own signature, lu_solve(&a, &b) -> x: inputs untouched, factors not copied back. */
int ma_lu_solve(int32_t n, const ma_c64* A_rowmajor, const ma_c64* b, ma_c64* x);
/* lu_factorize(&a) -> LuFactorization (lu.rs:83-137) and LuFactorization::solve(&b) (lu.rs:38-78): the factors stay in HBM. */
typedef struct ma_lu_factorization ma_lu_factorization_t;
int ma_lu_factorize(int32_t n, const ma_c64* A_rowmajor, ma_lu_factorization_t** out);
int ma_lu_factorization_solve(ma_lu_factorization_t* f, const ma_c64* b, ma_c64* x);
int ma_lu_factorization_destroy(ma_lu_factorization_t* f);

int ma_lu_plan_create(int32_t n, int device, ma_lu_plan_t** out);
/* Pivoting of a plan's factorisations (round 5). MA_LU_PIVOT_PARTIAL: LAPACK's zgetrf pivots -- what every entry that hands
 * pivots or factors across this boundary uses (ma_zgesv, ma_lu_factorize: lu.rs:83-137 exposes `pivots`). MA_LU_PIVOT_TOURNAMENT:
 * the pivot rows of every 32-column panel by a tournament (communication-avoiding LU: local eliminations, then eliminations of the
 * winners' rows), one chip-wide decision per panel instead of one per column; admissible behind lu_solve (lu.rs:142-153 returns x
 * only) and the default of the frequency sweep. Same storage of L, U and of the interchange sequence; solutions agree with LAPACK's to
 * rounding (tests/test_lu_gpu.py). ma_lu_plan_create makes a partial-pivoting plan. */
#define MA_LU_PIVOT_PARTIAL 0
#define MA_LU_PIVOT_TOURNAMENT 1
int ma_lu_plan_create_pivoting(int32_t n, int device, int32_t pivoting, ma_lu_plan_t** out);
int ma_lu_plan_pivoting(ma_lu_plan_t* plan, int32_t* pivoting);
/* Every plan factors its 64-column panels as two 32-column half-panels and tries each half-panel SPECULATIVELY first
 * (lu_spec.hip): partial pivoting among the panel's top 32 rows, then the check that no row below holds a larger entry in any column --
 * if it passes, zgetrf would have chosen the same rows, and the panel is done in two short launches without any exchange between
 * workgroups. Rows that fail the check (at most 32) join the candidates of a WIDENED attempt, checked the same way (two such attempts,
 * the second with the first's new violators); a panel all attempts give up is restored and factored by the plan's own panel kernel. Boundary operators of the Burton-Miller form (tbem.rs:96-222)
 * pass at the first attempt except near mesh singularities (the poles of a UV sphere: 6 % of the half-panels of BASELINE config #3),
 * which the widened attempt takes. Counts since the plan was made (synchronises the device); MA_LU_SPECULATE=0 switches it off. */
int ma_lu_plan_speculation_stats(ma_lu_plan_t* plan, int64_t* accepted, int64_t* accepted_widened, int64_t* rejected);
/* MA_LU_SPECULATE_VERIFIED (what such a plan starts with): the plan's own panel kernel is launched behind every speculative panel and
 * runs where the check failed -- every result is final. MA_LU_SPECULATE_OPTIMISTIC: nothing is launched behind it; a system that met a
 * rejected panel carries status -1 (ma_lu_plan_stage_info_dev) / MA_ERR_RETRY (ma_lu_plan_status) and is the CALLER's to solve again in
 * the verified mode -- what ma_bem_sweep_run does with the (never yet observed) frequencies whose operator is not diagonally dominant.
 * MA_LU_SPECULATE_OFF: the plan's own panel kernel only. */
#define MA_LU_SPECULATE_OFF 0
#define MA_LU_SPECULATE_VERIFIED 1
#define MA_LU_SPECULATE_OPTIMISTIC 2
int ma_lu_plan_set_speculation(ma_lu_plan_t* plan, int32_t mode);
int ma_lu_plan_speculation(ma_lu_plan_t* plan, int32_t* mode);
/* ma_zgesv with the pivoting named (ma_zgesv itself: partial). With MA_LU_PIVOT_TOURNAMENT the factors and ipiv that come back are
 * those of the tournament: P A = L U holds with them as it does with LAPACK's, the rows chosen differ. */
int ma_zgesv_pivoting(int32_t n, ma_c64* A_rowmajor, ma_c64* b, int32_t* ipiv_or_null, int32_t pivoting);
int ma_lu_plan_destroy(ma_lu_plan_t* plan);
/* Factor d_A in place (row-major, device) and solve for nrhs right-hand sides stored as
 * d_B[nrhs][n] (each contiguous). Asynchronous on `stream`; the singularity flag is reported by
 * ma_lu_plan_status(), which synchronises the stream. */
int ma_lu_plan_factor_solve_dev(ma_lu_plan_t* plan, void* d_A, void* d_B, int32_t nrhs, void* stream);
/* solve further right-hand sides with the factors a previous ma_lu_plan_factor_solve_dev call on this plan left in d_A */
int ma_lu_plan_solve_dev(ma_lu_plan_t* plan, void* d_A_factored, void* d_B, int32_t nrhs, void* stream);
/* The same for nmat (1..4) independent systems of size n kept in flight together (d_As[m], d_Bs[m] device
 * pointers), in lock step: every system has its own look-ahead stream, so that one system's latency-bound
 * panel chain runs underneath the others' trailing updates. Results per system are bit-identical to separate
 * ma_lu_plan_factor_solve_dev calls. */
int ma_lu_plan_factor_solve_batch_dev(ma_lu_plan_t* plan, int32_t nmat, void* const* d_As, void* const* d_Bs, int32_t nrhs, void* stream);
/* Staged use of a plan: a pipeline over a long sequence of systems (the frequencies of a sweep). factor_solve_batch moves
 * its systems in lock step; here every slot (0..3) is at its own block of columns, so a driver can start slot s a quarter of a
 * factorisation after slot s-1: each round then carries one big, one medium and one small trailing update, and every slot's
 * latency-bound panel chain has the time of all of them to finish. Per slot: stage_begin when A and b of its next system are
 * ready on `stream` (it starts the first block column), then one stage_round per block 0..num_blocks-1 -- together with the
 * other slots' current blocks, in one call per round --, then stage_finish (backward substitution; `stream` waits for it,
 * x is in b). stage_reset clears the status words and the timing accumulators before a run; ma_lu_plan_status reports as
 * for the batch. Same kernels and arithmetic as ma_lu_plan_factor_solve_batch_dev. */
int ma_lu_plan_reserve_events(ma_lu_plan_t* plan, int64_t count);   /* timing events created ahead of a timed run */
int ma_lu_plan_num_blocks(ma_lu_plan_t* plan, int32_t* blocks);
int ma_lu_plan_stage_reset(ma_lu_plan_t* plan, void* stream);
int ma_lu_plan_slot_stream(ma_lu_plan_t* plan, int32_t slot, void** stream);
/* A plan for 4 096..16 384 rows (or MA_LU_CU_SPLIT=<0 | 32 | 64>) runs its big trailing updates on a stream whose CU mask leaves
 * P CUs (P / 8 per XCD: 64 for a partial-pivoting plan, 32 for a tournament plan) to the latency-bound kernels of its lanes (no reference
 * counterpart: a schedule detail behind lu_solve, math-solvers/src/direct/lu.rs:142-153). *stream = that stream, or NULL when the plan does not split the chip. A driver of the
 * staged schedule passes it as ITS stream (assemblies included): one hardware queue less. The masked stream is a blocking stream (the
 * only kind the runtime makes with a CU mask): stage_reset / stage_begin refuse the NULL stream on such a plan (MA_ERR_INVALID) -- work on
 * the NULL stream would serialise against every big update. */
int ma_lu_plan_main_stream(ma_lu_plan_t* plan, void** stream);
/* CUs the plan leaves to its panel kernels (0: the chip is not split) and the chip's CU count */
int ma_lu_plan_cu_split(ma_lu_plan_t* plan, int32_t* panel_cus, int32_t* total_cus);
/* rounds between the starts of two of `slots` slots that the plan's kernels were measured best with (a driver of the staged
 * schedule starts slot s at round s * spacing; ma_bem_solve_sweep and bench.py do) */
int ma_lu_plan_stage_spacing(ma_lu_plan_t* plan, int32_t slots, int32_t* spacing);
int ma_lu_plan_stage_begin(ma_lu_plan_t* plan, int32_t slot, void* d_A, void* d_B, int32_t nrhs, void* stream);
int ma_lu_plan_stage_round(ma_lu_plan_t* plan, int32_t count, const int32_t* slots, const int32_t* blocks, void* stream);
int ma_lu_plan_stage_finish(ma_lu_plan_t* plan, int32_t slot, void* stream);
/* after stage_finish: the slot's status word (0, or 1 + the column of the first zero pivot) copied to a device int on `stream` */
int ma_lu_plan_stage_info_dev(ma_lu_plan_t* plan, int32_t slot, int32_t* d_out, void* stream);
int ma_lu_plan_status(ma_lu_plan_t* plan, void* stream);
/* Status values of the dense solve: MA_ERR_SINGULAR when a pivot column's largest entry is below 1e-30 in modulus (lu.rs:106-110,
 * which includes zgetf2's exact zero) or holds no comparable value (NaN); MA_ERR_HIP when a panel kernel was abandoned (its
 * co-resident workgroups did not complete an exchange within the limit: the plan is poisoned until the next factorisation starts,
 * no rows are moved with pivots of an abandoned panel). The rule that keeps panel kernels co-resident, as a pure function
 * (DESIGN.md 4 "Residency"): a CU can always take `slots` spinning workgroups of lds_bytes / regs registers per lane. */
int ma_lu_panel_slots_per_cu(int64_t lds_bytes, int32_t regs, int32_t* slots);

/* ------------------------------------------------------------------------------------------
 * Sparse FEM side: complex CSR operator, SpMV, residual, Jacobi-type smoother sweeps.
 * Replaces: CsrMatrix<Complex64>{num_rows, values, col_indices: Vec<usize>, row_ptrs: Vec<usize>} and
 *           CsrMatrix::matvec(&x) -> y                 math-solvers/src/sparse/csr.rs:21-33, 240-292
 *           HelmholtzAssembler::assemble(k, ..)        math-fem/src/assembly/assembler.rs:216-257
 *             (ma_csr_create_helmholtz keeps the real K and M values of the shared pattern resident and
 *              forms a_ij = K_ij - k^2 M_ij inside the kernels; ma_csr_set_wavenumber replaces the
 *              per-frequency value rewrite)
 *           AmgPreconditioner::smooth_jacobi / smooth_l1_jacobi / compute_diag_inv
 *                                                     math-solvers/src/preconditioners/amg.rs:855-929, 400-413
 *           the residual r = b - A x of the V-cycle    amg.rs:1028
 * Indices are the Rust side's usize widened to int64 (they are range-checked and narrowed to 32 bits
 * on the device). Column indices of a row must be unique (CSR from from_triplets / the assembler).
 * The *_dev forms take device pointers to n complex128 values each and a hipStream_t.
 * ------------------------------------------------------------------------------------------ */
typedef struct ma_csr ma_csr_t;
int ma_csr_create(int64_t n, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr_t** out);
int ma_csr_create_helmholtz(int64_t n, const int64_t* row_ptrs, const int64_t* col_indices, const double* K, const double* M,
                            int device, ma_csr_t** out);
/* A rectangular CSR operator (nrows x ncols): the AMG transfer operators P and R (amg.rs:236-243). SpMV only (x: ncols entries,
 * y: nrows); the diagonal-based sweeps and the transpose need a square handle. */
int ma_csr_create_rect(int64_t nrows, int64_t ncols, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr_t** out);
int ma_csr_num_cols(const ma_csr_t* h, int64_t* ncols);
int ma_csr_device(const ma_csr_t* h, int* device);
/* the operator's CSR arrays back on the host (row_ptrs n+1, col_indices nnz, values nnz; a K / M handle gives K - k^2 M at its wavenumber) */
int ma_csr_get(ma_csr_t* h, int64_t* row_ptrs, int64_t* col_indices, ma_c64* values);
int ma_csr_destroy(ma_csr_t* h);
int ma_csr_num_rows(const ma_csr_t* h, int64_t* n, int64_t* nnz);
int ma_csr_set_wavenumber(ma_csr_t* h, double k_re, double k_im);
/* HelmholtzAssembler with boundary matrices (math-fem/src/assembly/assembler.rs:19-32, 216-257): boundary_values[tag] are
 * real values on the operator's pattern; assemble(wavenumber, boundary_coeffs) forms A = K - k^2 M + sum_t c_t B_t over the
 * tags present in both. Without matching tags it equals ma_csr_set_wavenumber. */
int ma_csr_add_boundary(ma_csr_t* h, int32_t tag, const double* values_nnz);
int ma_csr_assemble(ma_csr_t* h, double k_re, double k_im, int32_t ncoef, const int32_t* tags, const ma_c64* coeffs, void* stream);
int ma_csr_spmv(ma_csr_t* h, const ma_c64* x, ma_c64* y);
int ma_csr_residual(ma_csr_t* h, const ma_c64* x, const ma_c64* b, ma_c64* r);
int ma_csr_jacobi(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, double omega, int sweeps);
int ma_csr_l1jacobi(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int sweeps);
int ma_csr_spmv_dev(ma_csr_t* h, const void* d_x, void* d_y, void* stream);
int ma_csr_residual_dev(ma_csr_t* h, const void* d_x, const void* d_b, void* d_r, void* stream);
int ma_csr_jacobi_dev(ma_csr_t* h, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, void* stream);
int ma_csr_l1jacobi_dev(ma_csr_t* h, void* d_x, const void* d_b, int sweeps, void* d_tmp, void* stream);
/* AmgPreconditioner::smooth_sym_gauss_seidel(matrix, x, b, num_sweeps)   math-solvers/src/preconditioners/amg.rs:932-978:
 * forward then backward Gauss-Seidel sweep over the rows, num_sweeps times (a row without a stored diagonal uses 1, rows
 * with |a_ii| <= 1e-15 are left alone). The sweeps run level by level over the dependency levels of the sparsity pattern
 * (built on first use), which reproduces the sequential sweep in index order; one launch per level.
 * ma_csr_gauss_seidel_sweep_dev is one sweep: mode 0 = math-fem smoother.rs:71-117 (x_i = (b_i - sigma)/a_ii, rows with
 * |a_ii| < 1e-15 skipped), mode 1 = the amg.rs form; backward = 0 ascending rows, 1 descending rows. */
int ma_csr_sym_gauss_seidel(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int sweeps);
int ma_csr_sym_gauss_seidel_dev(ma_csr_t* h, void* d_x, const void* d_b, int sweeps, void* stream);
int ma_csr_gauss_seidel_sweep_dev(ma_csr_t* h, void* d_x, const void* d_b, int mode, int backward, void* stream);
int ma_csr_gauss_seidel_levels(ma_csr_t* h, int64_t* forward, int64_t* backward);   /* dependency levels per sweep (diagnostic) */
/* A sweep runs as ONE persistent launch: the new iterate goes to a second array of the handle that starts as a sentinel, a row polls
 * the new values it depends on until they have left it, old values come from the untouched input (same arithmetic per row as a
 * launch per dependency level: bit-identical results). MA_CSR_GS_FLAGS=0: a launch per level; MA_CSR_GS_PERSISTENT=1: one
 * persistent launch with a device-wide barrier per level (measured slower). Every wait is bounded; ma_csr_status synchronises and
 * reports MA_ERR_HIP if a sweep ever gave up waiting. A handle's sweeps must not run on two streams at once. */
int ma_csr_status(ma_csr_t* h);

/* math-fem geometric-multigrid smoothers on the COO HelmholtzMatrix (math-fem/src/assembly/helmholtz.rs:22-33,
 * multigrid/smoother.rs:44-68, 120-160, 163-176). The triplets are summed once into a CSR operator (an ma_csr_t: all
 * ma_csr_* calls work on it); rows with |a_ii| < 1e-15 are skipped by the sweeps as in the reference.
 * kind: 0 Gauss-Seidel (SmootherConfig's default, smoother.rs:31-39, 71-117), 1 Jacobi, 2 symmetric Gauss-Seidel (forward
 * then backward sweep per iteration); the Gauss-Seidel sweeps are level-scheduled (see ma_csr_sym_gauss_seidel). */
/* the transposed operator as a new handle (apply_transpose of CsrMatrix, csr.rs:420-440) */
int ma_csr_transpose(ma_csr_t* h, ma_csr_t** out);
/* value epoch of a handle (bumped by set_wavenumber / assemble) and the cheap re-sync of a transposed copy after the
 * source changed frequency (*rebuild = 1: the copy has to be built again with ma_csr_transpose) */
unsigned long long ma_csr_epoch(const ma_csr_t* h);
int ma_csr_refresh_transpose(const ma_csr_t* src, ma_csr_t* dst, int* rebuild);
int ma_fem_matrix_create(int64_t n, int64_t nnz, const int64_t* rows, const int64_t* cols, const ma_c64* values, int device, ma_csr_t** out);
int ma_fem_smooth(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int kind, int iterations, double omega);
int ma_fem_residual(ma_csr_t* h, const ma_c64* x, const ma_c64* b, ma_c64* r);

/* ------------------------------------------------------------------------------------------
 * Operator boundary and GMRES.
 * Replaces: trait LinearOperator<Complex64> { num_rows, apply(&x) -> y }   math-solvers/src/traits.rs:316-327
 *           DenseOperator (matrix.dot(x))                               math-bem/src/core/solver/fmm_interface.rs:25-53
 *           impl LinearOperator for CsrMatrix                           math-solvers/src/sparse/csr.rs:420-440
 *           gmres / gmres_with_guess(operator, b, x0, config)            math-solvers/src/iterative/gmres.rs:96-277
 * ma_op_create_tbem is the matrix-free TBEM operator of BASELINE.json configs[4] (new; it must equal A x of
 * the dense matrix): rows [row0, row1) of y are produced; the plan is borrowed and must outlive the operator.
 * Tri3, Quad4 and mixed meshes (ElementType, types.rs) are all streamed.
 * A Rust struct holding the handle implements the trait; Drop calls ma_op_destroy.
 * ------------------------------------------------------------------------------------------ */
typedef struct ma_op ma_op_t;
typedef struct { int32_t iterations, restarts, converged; double residual; } ma_gmres_info_t;   /* GmresSolution, gmres.rs:72-85 */
int ma_op_create_dense(int64_t n, const ma_c64* A_rowmajor, int device, ma_op_t** out);
int ma_op_create_dense_dev(int64_t n, const void* d_A, int device, ma_op_t** out);
int ma_op_create_csr(ma_csr_t* csr, ma_op_t** out);
int ma_op_create_tbem(ma_bem_plan_t* plan, const ma_physics_t* physics, double beta_re, double beta_im,
                      int32_t row0, int32_t row1, ma_op_t** out);
/* The same operator ROW-SHARDED over the GPUs of one node (BASELINE.json configs[4], the memory-capped
 * dense-free path): devices[g] owns collocation rows [g N / ndev, (g+1) N / ndev) with its own BEM plan (the geometry is O(N)),
 * applies its row block against all field panels, and the y slices are gathered on devices[0] by peer copies inside the library
 * (one exchange per apply, 16 B N in total, no reduction). Every vector the caller passes lives on devices[0], so ma_gmres,
 * ma_gmres_preconditioned and ma_precond_create_diagonal drive it like any other ma_op_t. apply_transpose / apply_hermitian sum
 * the shards' contributions on devices[0] in shard order. The mesh is only borrowed during the call. */
int ma_op_create_tbem_multi(const ma_mesh_t* mesh, const ma_physics_t* physics, double beta_re, double beta_im,
                            const int32_t* devices, int32_t ndev, ma_op_t** out);
/* Single-level fast multipole operator A = [N] + [S][D][T].
 * Replaces: build_slfmm_system(elements, nodes, clusters, physics, n_theta, n_phi, n_terms) -> SlfmmSystem and its
 *           LinearOperator impl: apply = SlfmmSystem::matvec, apply_transpose = matvec_transpose
 *           math-bem/src/core/assembly/slfmm.rs:417-470, 150-257, 262-376, 378-395 (callers: bem_solver.rs:371-392,
 *           room_acoustics/solver.rs:754-1148); extract_near_field_matrix :104-132.
 * Clusters are the caller's (the reference builds them with its octree, which stays on the host): per cluster its centre,
 * its element indices, its near_clusters and far_clusters (types.rs:445-488) as offset / index lists. n_theta must be a
 * tabulated Gauss-Legendre order (1..8, 10, 12, 16, 20). The near blocks are integrated by the TBEM kernels with
 * compute_near_block's coefficient; the far field is the reference's (its D entry is h_0(k r) i k on the whole diagonal,
 * slfmm.rs:707-710). Velocity-type boundary conditions, no evaluation elements, every element in at most one cluster. */
typedef struct {
  int32_t        n_clusters;
  const double*  center;      /* n_clusters*3  Cluster.center                                   */
  const int32_t* elem_ptr;    /* n_clusters+1  offsets into elem_idx                            */
  const int32_t* elem_idx;    /*               Cluster.element_indices, cluster after cluster   */
  const int32_t* near_ptr;    /* n_clusters+1 */
  const int32_t* near_idx;    /*               Cluster.near_clusters                            */
  const int32_t* far_ptr;     /* n_clusters+1 */
  const int32_t* far_idx;     /*               Cluster.far_clusters                             */
} ma_clusters_t;
int ma_op_create_slfmm(ma_bem_plan_t* plan, const ma_clusters_t* clusters, const ma_physics_t* physics,
                       int32_t n_theta, int32_t n_phi, int32_t n_terms, ma_op_t** out);
int ma_op_slfmm_near_matrix(ma_op_t* op, ma_c64* A_rowmajor);
/* how the operator's upward / downward passes get the phases w_p exp(i k s_p.(x_j - C)): 2 recomputed with the bounded-argument sin / cos
 * (sphere rules of <= 1024 points), 1 from a stored table (larger rules, or MA_FMM_STORE_PHASES=1), 0 recomputed with libm (=0: the first version) */
int ma_op_slfmm_phase_mode(ma_op_t* op, int32_t* mode);
/* Multi-level fast multipole operator.
 * Replaces: build_cluster_tree(elements, target_elements_per_leaf, physics) -> Vec<ClusterLevel>   math-bem/src/core/assembly/mlfmm.rs:979-1038
 *           (estimate_num_levels :954-974, subdivide_level :1056-1180, compute_near_far_lists :1183-1223): host code, the reference's
 *           arithmetic in the reference's order; ma_cluster_tree_level_get hands the lists back (any pointer may be NULL; sizes from
 *           ma_cluster_tree_level_info: n_clusters(+1 for the *_ptr arrays), n_elem_listed, n_near, n_far, n_sons);
 *           build_mlfmm_system(elements, nodes, cluster_tree, physics) -> MlfmmSystem (:483-558) and MlfmmOperator's LinearOperator impl
 *           (solver/fmm_interface.rs:98-135): apply = MlfmmSystem::matvec (:128-460); apply_transpose is unimplemented!() there and
 *           MA_ERR_UNSUPPORTED here.
 * The reference's operator is its "simplified model" (:838-840), taken as it stands: near blocks for (i, i) and (i, j > i) with the
 * (j, i) block applied as the transpose, no free term, element centres ON an octant boundary listed in every octant they touch (their
 * dofs are summed once per leaf). Levels above the first one with a far pair contribute nothing and are skipped; every level from
 * there down must have a tabulated theta_points (4..8, 10, 12, 16, 20), else MA_ERR_UNSUPPORTED: gauss.rs:27-60 would give the
 * reference a longer sphere rule than theta_points * phi_points and its length guards would drop stages. Velocity-type boundary
 * conditions, no evaluation elements. */
typedef struct ma_cluster_tree ma_cluster_tree_t;
int ma_cluster_tree_build(const ma_mesh_t* mesh, int32_t target_elements_per_leaf, double wave_number, ma_cluster_tree_t** out);
int ma_cluster_tree_destroy(ma_cluster_tree_t* tree);
int ma_cluster_tree_num_levels(const ma_cluster_tree_t* tree, int32_t* levels);
int ma_cluster_tree_level_info(const ma_cluster_tree_t* tree, int32_t level, int32_t* n_clusters, int32_t* expansion_terms, int32_t* theta_points, int32_t* phi_points,
                               int64_t* n_elem_listed, int64_t* n_near, int64_t* n_far, int64_t* n_sons);
int ma_cluster_tree_level_get(const ma_cluster_tree_t* tree, int32_t level, double* center, double* radius, int32_t* elem_ptr, int32_t* elem_idx, int32_t* near_ptr, int32_t* near_idx,
                              int32_t* far_ptr, int32_t* far_idx, int32_t* son_ptr, int32_t* son_idx, int32_t* father);
int ma_op_create_mlfmm(ma_bem_plan_t* plan, const ma_cluster_tree_t* tree, const ma_physics_t* physics, ma_op_t** out);
/* number of shards of an operator (1 unless row-sharded) and, optionally, their first rows and devices */
int ma_op_num_shards(const ma_op_t* op, int32_t* shards, int32_t* row_begin_or_null, int32_t* device_or_null);
int ma_op_destroy(ma_op_t* op);
int ma_op_num_rows(const ma_op_t* op, int64_t* n);
int ma_op_apply(ma_op_t* op, const ma_c64* x, ma_c64* y);
/* A LinearOperator (traits.rs:316-327) whose ROWS are spread over processes (one rank per GPU; SURVEY 8e.2, BASELINE config #5):
 * `inner` = this rank's operator restricted to rows [row0, row1) (ma_op_create_tbem with a row range). After it has written its
 * rows of y = A x, `gather(user, d_y, n, row0, row1, stream)` must complete d_y in place with the other ranks' rows, ordered on
 * `stream` -- an all-gather on the caller's communicator (RCCL under torch.distributed's "nccl" backend); it returns 0 on
 * success. The handle drives ma_gmres* like any other; apply_transpose / apply_hermitian are not defined for it. `inner` is
 * not owned. */
typedef int (*ma_gather_fn)(void* user, void* d_y, int64_t n, int64_t row0, int64_t row1, void* stream);
int ma_op_create_gathered(ma_op_t* inner, int64_t row0, int64_t row1, ma_gather_fn gather, void* user, ma_op_t** out);
/* The same with the exchange INSIDE the library (north_star: RCCL over xGMI for the block exchange; traits.rs:316-327 `apply`): rank `rank` of
 * `nranks` owns rows [rank * per, min(n, (rank + 1) * per)), per = ceil(n / nranks), and every apply ends with one ncclAllGather of `per`
 * rows per rank on the apply's stream -- no callback, nothing for a Rust host to glue. nccl_comm is an ncclComm_t of the librccl the
 * process holds (the library binds librccl at first use with dlopen and so shares the copy already loaded); ma_rccl_get_unique_id /
 * ma_rccl_comm_create / ma_rccl_comm_destroy wrap ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy of that same copy for hosts
 * without an RCCL binding of their own (the 128-byte id travels from rank 0 to the others by the host's own means).
 * Every rank must take the same decisions (same iteration counts): the collective has no timeout. Each rank's block carries its
 * "a kernel abandoned a wait" word, and every rank raises its own if any rank's was set: the Krylov drivers of ALL ranks then return
 * MA_ERR_HIP at their end. Unmeasured on > 1 GPU. */
int ma_op_create_gathered_rccl(ma_op_t* inner, void* nccl_comm, int32_t nranks, int32_t rank, ma_op_t** out);
int ma_rccl_get_unique_id(void* id128);
int ma_rccl_comm_create(int32_t nranks, int32_t rank, const void* id128, int device, void** comm);
int ma_rccl_comm_destroy(void* comm);
int ma_op_apply_dev(ma_op_t* op, const void* d_x, void* d_y, void* stream);
/* apply_transpose (y = A^T x) and apply_hermitian (y = A^H x), traits.rs:326-358, for every operator kind. A matrix-free
 * operator created over a row block [row0, row1) returns that block's contribution to all entries of y (the blocks' results
 * add up). */
int ma_op_apply_transpose(ma_op_t* op, const ma_c64* x, ma_c64* y);
int ma_op_apply_hermitian(ma_op_t* op, const ma_c64* x, ma_c64* y);
int ma_op_apply_transpose_dev(ma_op_t* op, const void* d_x, void* d_y, void* stream);
int ma_op_apply_hermitian_dev(ma_op_t* op, const void* d_x, void* d_y, void* stream);
/* Preconditioner boundary: trait Preconditioner<T> { apply(&r) -> z }   math-solvers/src/traits.rs:370-375.
 * The device preconditioners are the AMG smoothers applied from z = 0 (one level of
 * AmgPreconditioner::apply, amg.rs:981-1005, 1068-1087): Jacobi(omega, sweeps), l1-Jacobi(sweeps) or symmetric
 * Gauss-Seidel(sweeps) on the CSR handle's current values (after ma_csr_set_wavenumber). The CSR handle is borrowed. */
typedef struct ma_precond ma_precond_t;
int ma_precond_create_jacobi(ma_csr_t* csr, double omega, int32_t sweeps, ma_precond_t** out);
int ma_precond_create_l1jacobi(ma_csr_t* csr, int32_t sweeps, ma_precond_t** out);
int ma_precond_create_sym_gauss_seidel(ma_csr_t* csr, int32_t sweeps, ma_precond_t** out);
/* DiagonalPreconditioner::from_diagonal of an operator's diagonal (math-bem/src/core/solver/fmm_interface.rs:177-212): any
 * operator kind; for the matrix-free TBEM operator the diagonal is the self terms, for the single-level fast multipole operator the
 * diagonal of its self blocks (SparseNearfieldIlu::from_slfmm, fmm_interface.rs:249-297); the multi-level operator is MA_ERR_UNSUPPORTED */
int ma_precond_create_diagonal(ma_op_t* op, ma_precond_t** out);
/* AmgPreconditioner::apply = v_cycle from z = 0 (math-solvers/src/preconditioners/amg.rs:981-1065, 1068-1103) on the device, over a
 * hierarchy the caller built (AmgPreconditioner::from_csr, amg.rs:276-372, stays on the host): level l = operator A[l] (square
 * ma_csr_t) and, for l < nlevels-1, prolongation P[l] (n_l x n_{l+1}) and restriction R[l] (n_{l+1} x n_l) from ma_csr_create_rect;
 * all handles are borrowed. smoother 0 Jacobi(jacobi_weight) (AmgSmoother::Jacobi / Chebyshev), 1 L1Jacobi, 2 SymmetricGaussSeidel;
 * pre / post sweeps per level, the coarsest level runs 20 / 20 / 10 sweeps (:986-1003); cycle 0 V, 1 W (two V-cycles), 2 F (a second
 * V-cycle on the residual). Use with ma_gmres_preconditioned (SolverType::GmresAmg, math-fem/src/solver/mod.rs:667). */
int ma_precond_create_amg(int32_t nlevels, ma_csr_t* const* A, ma_csr_t* const* P, ma_csr_t* const* R, int32_t smoother, double jacobi_weight,
                          int32_t num_pre_smooth, int32_t num_post_smooth, int32_t cycle, ma_precond_t** out);
/* AmgPreconditioner::from_csr(matrix, config) (math-solvers/src/preconditioners/amg.rs:276-372): the whole preconditioner from the matrix.
 * The hierarchy is built on the host inside the library with the reference's steps in their order -- compute_strength_matrix (:418-474),
 * coarsen_ruge_stuben (:477-532) or coarsen_pmis (:535-642), build_interpolation (:645-807: Standard / Extended / Direct, truncation),
 * R = transpose_csr(P) (:810-822), A_c = R (A P) with CsrMatrix::matmul (sparse/csr.rs:594-651, entries of norm <= 1e-15 dropped) -- and
 * uploaded level by level; the V / W / F cycle then runs on the device as for ma_precond_create_amg. The level operators are owned by
 * the preconditioner, `A` (level 0) is borrowed and read with its current values (after ma_csr_set_wavenumber).
 * AmgConfig (:104-146), enums in their declaration order:
 *   coarsening 0 RugeStuben, 1 Pmis, 2 Hmis (the reference runs Pmis for it); interpolation 0 Standard, 1 Extended, 2 Direct;
 *   smoother 0 Jacobi, 1 L1Jacobi, 2 SymmetricGaussSeidel, 3 Chebyshev (the reference smooths with Jacobi for it); cycle 0 V, 1 W, 2 F.
 * aggressive_coarsening_levels is carried and, as in the reference, not read. */
typedef struct ma_amg_config {
  int32_t coarsening, interpolation, smoother, cycle;
  double strong_threshold;
  int32_t max_levels, coarse_size, num_pre_smooth, num_post_smooth;
  double jacobi_weight, trunc_factor;
  int32_t max_interp_elements, aggressive_coarsening_levels;
} ma_amg_config_t;
/* which: 0 AmgConfig::default() (:148-167), 1 for_bem (:173-181), 2 for_fem (:184-191), 3 for_parallel (:194-203),
 * 4 for_difficult_problems (:206-218) */
int ma_amg_config_preset(int32_t which, ma_amg_config_t* cfg);
int ma_precond_create_amg_from_csr(ma_csr_t* A, const ma_amg_config_t* cfg, ma_precond_t** out);
/* num_levels / grid_complexity / operator_complexity / setup_time_ms (:375-392); any pointer may be NULL */
int ma_precond_amg_info(ma_precond_t* M, int32_t* num_levels, double* grid_complexity, double* operator_complexity, double* setup_time_ms);
/* the handles of one level (borrowed; P and R are NULL on the coarsest): AmgDiagnostics (:1107-1133) reads their sizes, the tests
 * their entries (ma_csr_get) */
int ma_precond_amg_level(ma_precond_t* M, int32_t level, ma_csr_t** A, ma_csr_t** P, ma_csr_t** R);
/* IluPreconditioner::from_csr(matrix) (math-solvers/src/preconditioners/ilu.rs:36-140) and its apply (:143-175): ILU(0) on the matrix' own
 * pattern. The factorisation runs on the host with the reference's loops (setup, like the AMG hierarchy), the two triangular solves of
 * every apply on the device (level-scheduled). Use with ma_gmres_preconditioned: gmres_solve_with_ilu(_operator)
 * (math-bem/src/core/solver/fmm_interface.rs:450-474) is ma_op + this preconditioner over the (near-field) matrix as CSR. */
int ma_precond_create_ilu0(ma_csr_t* csr, ma_precond_t** out);
/* IluFixedPointPreconditioner::from_csr(matrix, iterations) (math-solvers/src/preconditioners/ilu_parallel.rs:397-495; from_csr_default: 3)
 * and its apply (:510-590): x = D^-1 r, then `iterations` times x <- D^-1 (r - (L + U_off) x) over the ILU(0) factors. Factorisation on the
 * host, the sweeps on the device. IluColoringPreconditioner (:52-148, level-scheduled solves of the same factors) is ma_precond_create_ilu0. */
int ma_precond_create_ilu_fixed_point(ma_csr_t* csr, int32_t iterations, ma_precond_t** out);
/* AdditiveSchwarzPreconditioner::from_csr(matrix, num_subdomains, overlap) (math-solvers/src/preconditioners/schwarz.rs:84-145) and its apply
 * (:394-408): contiguous index blocks grown `overlap` times along the matrix graph, ILU(0) of every local matrix, the local solutions added
 * back with weights 1 / (subdomains holding the row). Setup on the host; on the device the subdomains are stacked into one block-diagonal
 * operator, so all local solves run together. stats (:148-170): subdomains, smallest, largest and mean size. */
int ma_precond_create_schwarz(ma_csr_t* csr, int32_t num_subdomains, int32_t overlap, ma_precond_t** out);
int ma_precond_schwarz_stats(ma_precond_t* M, int64_t* num_subdomains, int64_t* min_size, int64_t* max_size, double* avg_size);
int ma_precond_destroy(ma_precond_t* M);
int ma_precond_apply_dev(ma_precond_t* M, const void* d_r, void* d_z, void* stream);
int ma_precond_apply(ma_precond_t* M, const ma_c64* r_host, ma_c64* z_host);   /* host buffers */
/* DiagonalPreconditioner::from_csr (preconditioners/diagonal.rs:27-37, 57-75) is jacobi(omega = 1, sweeps = 1). */
/* gmres_preconditioned(_with_guess)(operator, precond, b, x0, config): left preconditioning, tolerance relative to
 * ||M^-1 b||   math-solvers/src/iterative/gmres.rs:282-585 */
int ma_gmres_preconditioned(ma_op_t* op, ma_precond_t* M, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations,
                            double tol, ma_c64* x_out, ma_gmres_info_t* info);
/* gmres_pipelined(operator, precond, b, x0, config) (math-solvers/src/iterative/gmres_pipelined.rs:18-250, p-GMRES): auxiliary
 * basis Z = M^-1 A V, classical Gram-Schmidt; the inner products of a step run on a second stream beside the operator apply of
 * the next (the reference's rayon::join). M may be NULL (IdentityPreconditioner). Left preconditioning, tolerance relative to
 * ||M^-1 (b - A x0)||. */
int ma_gmres_pipelined(ma_op_t* op, ma_precond_t* M_or_null, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations,
                       double tol, ma_c64* x_out, ma_gmres_info_t* info);
/* The other Krylov solvers behind the same operator boundary: bicgstab(operator, b, config) (math-solvers/src/iterative/bicgstab.rs:46-182; what
 * BemSolver uses for the fast multipole methods, math-bem/src/core/bem_solver.rs), cgs (cgs.rs:46-139), cg (cg.rs:49-138; correct for Hermitian
 * positive definite operators only, as the reference says). x starts at 0; converged when ||r|| / ||b|| < tol; breakdown (|rho|, |<r0, v>|,
 * |<t, t>|, |omega| < 1e-30) returns converged = 0 with the iterate reached; info->restarts is 0. Defaults of the configs: 1000 iterations, 1e-6. */
int ma_bicgstab(ma_op_t* op, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info);
int ma_cgs(ma_op_t* op, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info);
int ma_cg(ma_op_t* op, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info);
/* Restarted GMRES(m), relative tolerance on ||b||; defaults of GmresConfig: restart 30, tol 1e-6, 100 restarts
 * (gmres.rs:27-35). x0 may be NULL. Non-convergence is reported in info->converged, not as an error. */
int ma_gmres(ma_op_t* op, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations, double tol,
             ma_c64* x_out, ma_gmres_info_t* info);

/* ------------------------------------------------------------------------------------------
 * Field evaluation and the room-acoustics collocation matrix (the O(M N) / O(N^2) loops either side of the solve).
 * Replaces: compute_scattered_field(eval_points, elements, nodes, surface_pressure, surface_velocity, physics)
 *           math-bem/src/core/postprocess/pressure.rs:81-137 (7-point rule per element, :154-258); surface values are
 *           one per boundary element in plan order; surface_velocity may be NULL.
 *           build_bem_matrix_parallel(mesh, k)   math-bem/src/room_acoustics/solver.rs:448-493, from the element
 *           centres, normals and areas of element_center_and_normal / element_area (:38-122).
 * ------------------------------------------------------------------------------------------ */
int ma_bem_plan_scattered_field(ma_bem_plan_t* plan, const ma_physics_t* physics, int32_t n_eval, const double* eval_points,
                                const ma_c64* surface_pressure, const ma_c64* surface_velocity, ma_c64* out);
int ma_room_build_matrix(int32_t n, const double* center, const double* normal, const double* area, double k, ma_c64* A_rowmajor);
int ma_room_build_matrix_dev(int32_t n, const void* d_center, const void* d_normal, const void* d_area, double k, void* d_A, void* stream);
/* The rest of math-bem/src/room_acoustics/solver.rs around the collocation matrix:
 * element_center_and_normal :38-67, element_area :70-122, element_characteristic_length :600-611 (host arithmetic,
 *   conn = 4 node ids per element, -1 in the fourth slot of a triangle);
 * build_bem_matrix_adaptive(mesh, k, use_adaptive) :500-597;
 * calculate_incident_field_derivative_parallel :638-678 and calculate_field_pressure_bem_parallel :687-748, where
 *   amp is [nsrc] (the same amplitude towards every point) or, with per_point != 0, [nsrc][n points] as
 *   Source::amplitude_towards (math-xem-common/src/source.rs:203-219) gives it. */
int ma_room_element_data(int32_t n_elem, const double* nodes, const int32_t* conn, double* center, double* normal, double* area, double* charlen);
int ma_room_build_matrix_adaptive(int32_t n_nodes, const double* nodes, int32_t n_elem, const int32_t* conn, double k, int use_adaptive, ma_c64* A_rowmajor);
int ma_room_incident_derivative(int32_t n, const double* center, const double* normal, int32_t nsrc, const double* src_pos, const double* amp, int per_point,
                                double k, ma_c64* out);
int ma_room_field_pressure(int32_t n, const double* center, const double* normal, const double* area, const ma_c64* surface_pressure, int32_t nsrc,
                           const double* src_pos, const double* amp, int per_point, int32_t npts, const double* pts, double k, ma_c64* out);

/* ------------------------------------------------------------------------------------------
 * Parity-test hooks (no counterpart in the reference API): raw panel integrals computed by the
 * device kernels, comparable to IntegrationResult of regular_integration / singular_integration
 * (math-bem/src/core/integration/regular.rs:33, singular.rs:123; types.rs:722-734).
 *  probe_pairs: pairs = npairs x (i, j), i != j, panel indices in plan order; out5 = npairs x 5
 *               complex {number of sub-elements, G, H, H^T, E}.
 *  probe_self:  out5 = num_dofs x 5 complex {number of kernel points, G, H, H^T, E}.
 *  get_near_pairs: the (i, j) list whose level-0 distance ratio is < 3 (singular.rs:553-556).
 * ------------------------------------------------------------------------------------------ */
int ma_bem_plan_probe_pairs(ma_bem_plan_t* plan, const ma_physics_t* physics, int64_t npairs,
                            const int32_t* pairs, ma_c64* out5);
int ma_bem_plan_probe_self(ma_bem_plan_t* plan, const ma_physics_t* physics, ma_c64* out5);
int ma_bem_plan_get_near_pairs(const ma_bem_plan_t* plan, int32_t* out_pairs);

/* C <- C - A B (row-major, tight leading dimensions, host buffers) on the f64 matrix cores: the kernel of the LU's trailing update
 * (three real products per complex product; what zgetrf's zgemm call becomes, lu.rs:142-153 -> LAPACK). */
int ma_zgemm_sub(int32_t M, int32_t N, int32_t K, const ma_c64* A, const ma_c64* B, ma_c64* C);
/* Measured issue rate of v_mfma_f64_16x16x4_f64 over the whole chip, TFLOP/s (roofline peak check). */
int ma_probe_mfma_f64(int device, double* tflops);
int ma_lu_plan_dump_intervals(ma_lu_plan_t* plan, int32_t phase, double* out_pairs, int32_t capacity, int32_t* count);   /* measurement: (start, end) ms of the timed intervals of one phase */
#ifdef MA_DIAGNOSTICS
/* The DIAGNOSTIC build only (make -C math_audio_amd/csrc diag -> lib/libmathaudio_hip_diag.so, loaded by tools and by the tests that
 * need a hook through MA_LIB_PATH): `repeat` launches of the trailing-update kernel on device buffers / of the matrix-core probe on
 * `stream` as background load. The shipped library exports neither these nor any switch that changes a result: the diagnostic build adds
 * MA_LU_TEST_ABORT_COL (a panel workgroup gives up a wait), MA_TEST_ALLOW_DUPLICATE_DEVICES (one GPU listed several times) and
 * MA_TEST_SWEEP_REJECT (a frequency of a sweep is treated as rejected by the speculative panels). */
int ma_diag_zgemm_dev(int32_t M, int32_t N, int32_t K, const void* dA, const void* dB, void* dC, int32_t repeat, void* stream);
int ma_diag_mfma_burn(void* d_out, int32_t blocks, int32_t iters, int32_t repeat, void* stream);
/* Virtual ranks for ma_op_create_gathered_rccl on ONE device (RCCL refuses two ranks on one GPU): nranks loopback communicators whose
 * all-gather is a host-barrier copy between ranks that are host threads of this process; it replaces ncclAllGather for the rest of the
 * process. _poison: that rank's status entry reads "a wait was abandoned" in every later exchange (-1: none). */
int ma_rccl_test_loopback_create(int32_t nranks, int device, void** comms);
int ma_rccl_test_loopback_poison(void* comm, int32_t rank);
int ma_rccl_test_loopback_destroy(void** comms, int32_t nranks);
#endif

/* ------------------------------------------------------------------------------------------
 * Diagnostics used by bench.py / tests: elapsed GPU time (ms) of the tagged phases of the most
 * recent call on the plan, measured with HIP events on the plan's stream when timing is enabled.
 * ------------------------------------------------------------------------------------------ */
int ma_bem_plan_set_timing(ma_bem_plan_t* plan, int enable);
/* out[0]=far kernel, out[1]=near kernel, out[2]=self kernel ms of the last assemble */
int ma_bem_plan_last_timing(ma_bem_plan_t* plan, double* out3);
int ma_lu_plan_set_timing(ma_lu_plan_t* plan, int enable);   /* 0 off, 1 every phase, 2 only the trailing-update launches (staged runs) */
/* out[0]=panel kernels (on the look-ahead streams, overlap out[3]), out[1]=row swaps, out[2]=trsm, out[3]=zgemm launches
 * of the main lane, out[4]=right-hand-side and triangular solves (ms); out[5]=number of zgemm launches (all lanes);
 * out[6]=whole factor+solve on the caller's stream (ms); out[7]=zgemm launches of the look-ahead lanes (ms) */
int ma_lu_plan_last_timing(ma_lu_plan_t* plan, double* out8);
/* the update (zgemm) launches out[3]+out[7] cover: count, algorithmic flops (8 M N K) and C read+write bytes (32 M N) */
int ma_lu_plan_last_update_stats(ma_lu_plan_t* plan, double* launches, double* flops, double* c_bytes);
int ma_lu_plan_last_big_update_stats(ma_lu_plan_t* plan, double* launches, double* flops);   /* of those: the big trailing updates on the caller's stream (kernel zgemm3m_dma_kernel<2, 2, true>) */

#ifdef __cplusplus
}
#endif
#endif /* MATHAUDIO_HIP_H */
