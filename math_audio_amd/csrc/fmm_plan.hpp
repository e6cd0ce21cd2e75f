// fmm_plan.hpp — the single-level fast multipole operator (math-bem/src/core/assembly/slfmm.rs) behind ma_op_t.
#pragma once
#include "bem_kernels.hpp"

struct ma_slfmm;
int slfmm_create(ma_bem_plan* plan, const ma_clusters_t* cl, const ma_physics_t* physics, int n_theta, int n_phi, int n_terms, ma_slfmm** out);
void slfmm_destroy(ma_slfmm* S);
// y = ([N] + [S][D][T]) x (transpose = 0: SlfmmSystem::matvec) or its transpose (matvec_transpose); device vectors of num_dofs entries
int slfmm_apply(ma_slfmm* S, const ma::c64* d_x, ma::c64* d_y, int transpose, hipStream_t st);
// extract_near_field_matrix (slfmm.rs:104-132): [N] as a dense num_dofs x num_dofs matrix on the device
int slfmm_near_matrix(ma_slfmm* S, ma::c64* d_A, hipStream_t st);
long long slfmm_num_dofs(const ma_slfmm* S);
int slfmm_device(const ma_slfmm* S);
