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
int slfmm_self_diagonal(ma_slfmm* S, ma::c64* d_diag, hipStream_t st);   // n entries, zero where a dof sits in no cluster
long long slfmm_num_dofs(const ma_slfmm* S);
int slfmm_phase_mode(const ma_slfmm* S);   // how the upward / downward passes get their phases: 2 recomputed (bounded-argument sin / cos), 1 stored table, 0 libm
int slfmm_device(const ma_slfmm* S);

// ---- multi-level operator (math-bem/src/core/assembly/mlfmm.rs)
struct ma_cluster_tree;
struct ma_mlfmm;
// build_cluster_tree(elements, target_elements_per_leaf, physics) (:979-1038) from the elements' centres: host code
int cluster_tree_build(int n_elem, const double* center, long long target_elements_per_leaf, double wave_number, ma_cluster_tree** out);
void cluster_tree_destroy(ma_cluster_tree* T);
int cluster_tree_num_levels(const ma_cluster_tree* T);
int cluster_tree_level_info(const ma_cluster_tree* T, int level, int* nclusters, int* terms, int* theta, int* phi, long long* n_elem_listed, long long* n_near, long long* n_far,
                            long long* n_sons);
int cluster_tree_level_get(const ma_cluster_tree* T, int level, double* center, double* radius, int* elem_ptr, int* elem_idx, int* near_ptr, int* near_idx, int* far_ptr, int* far_idx,
                           int* son_ptr, int* son_idx, int* father);
// build_mlfmm_system (:483-558) + MlfmmSystem::matvec (:128-460)
int mlfmm_create(ma_bem_plan* plan, const ma_cluster_tree* T, const ma_physics_t* physics, ma_mlfmm** out);
void mlfmm_destroy(ma_mlfmm* S);
int mlfmm_apply(ma_mlfmm* S, const ma::c64* d_x, ma::c64* d_y, hipStream_t st);
