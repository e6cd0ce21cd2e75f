// bem_kernels.hpp — device-side views and launchers of the TBEM assembly kernels.
#pragma once
#include "ma_common.hpp"

namespace ma {

// Panel geometry in HBM, structure-of-arrays over the np non-evaluation Tri3 panels.
// Field side (integration over panel j): vertices p0,p1,p2 in connectivity order, edges
// e1 = p1-p0, e2 = p2-p0, n_y = (e1 x e2)/|e1 x e2| (NOT flipped, regular.rs:249-257),
// jac = |e1 x e2|. Collocation side (row i): stored centre c and stored outward-flipped
// normal nx (generators.rs:590-600), stored area (subdivision criterion, singular.rs:535).
struct dc;
struct BemGeom {
  int np;                 // panels
  int nd;                 // num_dofs (== np) = leading dimension of A
  const double* p0[3];
  const double* p1[3];
  const double* p2[3];
  const double* e1[3];
  const double* e2[3];
  const double* ny[3];
  const double* jac;
  const double* c[3];
  const double* nx[3];
  const double* area;
  const int* dof;
  const unsigned char* bc_type;
  // Quad4 panels (mixed meshes): 4th vertex, per-panel node count (3 / 4), and the list of the quad panels.
  // nquad == 0 (the sphere configurations) leaves every Tri3 kernel on its original path.
  const double* p3[3];
  const unsigned char* ptype;
  const int* quad_ids;
  int nquad;
  int all_velocity;       // every panel carries a velocity-type condition (bc_type 0: the rigid configurations): the far kernel then
                          // needs neither the single-layer sum G nor dG/dn_x (assemble_tbem reads only H and E, tbem.rs:311-330)
};

// boundary values of the panels (BoundaryCondition::{Velocity,Pressure} payloads, types.rs:330-351): 4 slots per panel
struct BemBc {
  const ::ma::dc* val;          // [4 np]
  const int* len;               // values present (1..4)
  const unsigned char* nz;      // has_nonzero_bc (tbem.rs:247-249)
};

struct BemPhys {
  double k, harmonic, tau, gamma;
  double beta_re, beta_im;
  double sign;            // dg_dn_sign (tbem.rs:108-123)
};

int bem_upload_tables(const double tri13_scaled[13][3], const double* glx, const double* glw, const int glidx[21][2]);
int bem_launch_near_list(const BemGeom& g, int pass, int* counts, const long long* offsets, int2* pairs, hipStream_t st);
int bem_launch_far(const BemGeom& g, const BemPhys& ph, c64* A, hipStream_t st);
// nf (1..3) systems of the SAME mesh at different wavenumbers in one pass: the geometry of a quadrature point is computed once
int bem_launch_far_multi(const BemGeom& g, int nf, const BemPhys* phs, c64* const* As, hipStream_t st, int blk0 = 0, int nblk = -1);
int bem_far_row_strips(const BemGeom& g);
int bem_launch_near(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* A, hipStream_t st);
int bem_launch_near_multi(const BemGeom& g, int cnt, const BemPhys* ph, c64* const* As, const int2* pairs, long long npairs, hipStream_t st);
int bem_launch_self(const BemGeom& g, const BemPhys& ph, c64* A, hipStream_t st);
int bem_launch_probe_pairs(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out5, hipStream_t st);
int bem_launch_probe_self(const BemGeom& g, const BemPhys& ph, c64* out5, hipStream_t st);
int bem_launch_near_list_values(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st);
int bem_launch_self_list_values(const BemGeom& g, const BemPhys& ph, c64* out, hipStream_t st);
// Quad4 columns of the matrix-free operator
int bem_quad_strips(const BemGeom& g);
int bem_launch_quad_matvec(const BemGeom& g, const BemPhys& ph, int row0, int row1, int rows_per_block, const c64* x, c64* partial_quad_strips, hipStream_t st);
int bem_launch_quad_matvec_t(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, int chunk_rows, const c64* x, c64* partial, hipStream_t st);
int bem_launch_quad_pairs_far(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st);
int bem_launch_zero(c64* v, int n, hipStream_t st);
int bem_launch_rhs_bc(const BemGeom& g, const BemPhys& ph, const BemBc& bc, const int2* pairs, const long long* pair_off, long long npairs,
                      c64* scratch, c64* rhs, hipStream_t st);
int bem_launch_incident(const BemGeom& g, const BemPhys& ph, int kind, const double* v, double are, double aim,
                        int accumulate, c64* rhs, hipStream_t st);

}  // namespace ma

// the plan object behind ma_bem_plan_t (shared by bem_plan.hip and op_plan.hip)
struct ma_bem_plan {
  int device = 0;
  int np = 0, nd = 0;
  double avg_radius = 0.0;         // tbem.rs:108-117
  double diameter = 0.0;           // diagonal of the nodes' bounding box: every distance the kernels see is below it
  void* pool = nullptr;            // one HBM allocation holding every SoA array
  ma::BemGeom geom{};
  int2* d_pairs = nullptr;         // near pairs, sorted by collocation row i then j
  long long* d_pair_off = nullptr; // np + 1 row offsets into d_pairs
  long long npairs = 0;
  bool has_bc = false;             // some panel carries a non-zero boundary value
  ma::BemBc bc{};                  // device arrays (inside `bcpool`)
  void* bcpool = nullptr;
  ma::c64* d_rhs_scratch = nullptr;  // far[np] | self[np] | self5[5 np] | near[npairs]
  bool timing = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  double last_ms[3] = {0, 0, 0};
  bool ev_valid = false;
};
int ma_bem_make_phys(const ma_bem_plan* P, const ma_physics_t* ph, double bre, double bim, ma::BemPhys* o);
