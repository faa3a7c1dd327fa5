/*
 * oracle/d4est_oracle_mgmatrix.c -- TEST INFRASTRUCTURE ONLY (see d4est_oracle.h).
 *
 * CPU restatement of the multigrid MATRIX OPERATOR: the zeroth-order term of a linearised nonlinear problem kept as one dense
 * block per element, Galerkin-restricted level by level, and applied by the smoother's apply_lhs on every level below the finest
 * (SURVEY.md section 8, row a6 last column).
 *   d4est_quadrature_compute_mass_matrix                         Quadrature/d4est_quadrature.c:1143-1186
 *   d4est_quadrature_apply_fofufofvlilj, QUAD_COMPUTE_MATRIX      Quadrature/d4est_quadrature.c:593-774 (:748-760)
 *   d4est_solver_multigrid_matrix_setup_fofufofvlilj_operator     Solver/d4est_solver_multigrid_matrix_operator.c:160-245
 *   d4est_operators_compute_prolong_matrix                        dGMath/d4est_operators.c:572-605
 *   d4est_operators_compute_PT_mat_P                              dGMath/d4est_operators.c:608-667
 *   d4est_solver_multigrid_matrix_operator_restriction_callback   Solver/d4est_solver_multigrid_matrix_operator.c:6-48, driven by the
 *     coarse-grid walk of d4est_solver_multigrid_apply_restriction Solver/d4est_solver_multigrid_callbacks.h:113-208
 *   constant_density_star_apply_jac_add_nonlinear_term_using_matrix Problems/ConstantDensityStar/constant_density_star_fcns.h:485-527
 * Same operation order as the reference (unit-vector columns, mat * P then P^T * (mat P), children summed with axpy, per-element
 * dgemv then one axpy over the whole vector); BLAS calls are the oracle's naive row-major loops.
 */
#include "d4est_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double* dalloc0(size_t n) {
  double* p = (double*)calloc(n > 0 ? n : 1, sizeof(double));
  if (!p) { fprintf(stderr, "[ORACLE_ABORT] out of memory\n"); abort(); }
  return p;
}
static int nodes3(int deg) { return (deg + 1) * (deg + 1) * (deg + 1); }

/* LinearAlgebra/d4est_linalg.c:154-166 (the element-by-element scan of the reference collapses to this assignment) */
static void set_column(double* A, const double* column, int col, int N, int M) {
  for (int i = 0; i < N; i++) A[M * i + col] = column[i];
}

/* Quadrature/d4est_quadrature.c:1143-1186, volume objects, 3-D: out is (deg_lobatto+1)^3 x (deg_lobatto+1)^3, row-major */
void oracle_quadrature_compute_mass_matrix(int quad_type, int deg_lobatto, const double* jac_quad, int deg_quad, double* out) {
  const int n = nodes3(deg_lobatto);
  double* u = dalloc0(n);
  double* Mu = dalloc0(n);
  for (int i = 0; i < n; i++) {
    u[i] = 1.;
    oracle_quadrature_apply_mass_matrix(quad_type, u, deg_lobatto, jac_quad, deg_quad, Mu);
    set_column(out, Mu, i, n, n);
    u[i] = 0.;
  }
  free(Mu);
  free(u);
}

/* Quadrature/d4est_quadrature.c:593-774 with apply_or_compute_matrix = QUAD_COMPUTE_MATRIX, interpolate_f = 0 (:635-683 form
 * fofu_fofv_jac = jac * f(u) f(v) at the quadrature nodes; coeff_quad = that product of callbacks, NULL = 1) */
void oracle_quadrature_compute_fofufofvlilj_matrix(int quad_type, int deg_lobatto, const double* coeff_quad, const double* jac_quad,
                                                   int deg_quad, double* out) {
  const int nq = nodes3(deg_quad);
  double* fofu_fofv_jac = dalloc0(nq);
  for (int i = 0; i < nq; i++) {
    fofu_fofv_jac[i] = jac_quad[i];
    if (coeff_quad) fofu_fofv_jac[i] *= coeff_quad[i];
  }
  oracle_quadrature_compute_mass_matrix(quad_type, deg_lobatto, fofu_fofv_jac, deg_quad, out);
  free(fofu_fofv_jac);
}

/* Solver/d4est_solver_multigrid_matrix_operator.c:160-245: one block per element, consecutive in element order
 * (matrix_nodal_stride advances by volume_nodes^2); returns the number of doubles written (d4est_mesh_get_local_matrix_nodes) */
long long oracle_mg_matrix_setup_fofufofvlilj_operator(int quad_type, int n_elements, const int* deg, const int* deg_quad,
                                                       const int* quad_stride, const double* J_quad, const double* coeff_quad,
                                                       double* matrix_at0) {
  long long matrix_nodal_stride = 0;
  for (int e = 0; e < n_elements; e++) {
    const int volume_nodes = nodes3(deg[e]);
    const long long matrix_volume_nodes = (long long)volume_nodes * volume_nodes;
    if (matrix_at0)
      oracle_quadrature_compute_fofufofvlilj_matrix(quad_type, deg[e], coeff_quad ? coeff_quad + quad_stride[e] : NULL,
                                                    J_quad + quad_stride[e], deg_quad[e], matrix_at0 + matrix_nodal_stride);
    matrix_nodal_stride += matrix_volume_nodes;
  }
  return matrix_nodal_stride;
}

/* dGMath/d4est_operators.c:572-605: prolong_mat is (sum_i (degh_i+1)^3) x (degH+1)^3, row-major, column i = P e_i */
void oracle_compute_prolong_matrix(int degH, int dim, const int* degh, int children, double* prolong_mat) {
  int volume_nodes_h = 0;
  if (dim != 3) { fprintf(stderr, "[ORACLE_ABORT] compute_prolong_matrix: dim = %d\n", dim); abort(); }
  for (int i = 0; i < children; i++) volume_nodes_h += nodes3(degh[i]);
  const int volume_nodes_H = nodes3(degH);
  double* u = dalloc0(volume_nodes_H);
  double* Mu = dalloc0(volume_nodes_h);
  for (int i = 0; i < volume_nodes_H; i++) {
    u[i] = 1.;
    if (children == 8) oracle_apply_hp_prolong(u, degH, dim, degh, Mu);
    else oracle_apply_p_prolong(u, degH, dim, degh[0], Mu);
    set_column(prolong_mat, Mu, i, volume_nodes_h, volume_nodes_H);
    u[i] = 0.;
  }
  free(Mu);
  free(u);
}

/* dGMath/d4est_operators.c:608-667, LITERALLY (literal_window != 0): the stacked prolongation P ((sum_i nh_i) x nH) is transposed as a
 * whole into PT (nH x sum_i nh_i, :637) and child i's left factor is the window &PT[stride_P] READ AS an nH x nh_i row-major matrix
 * (:651).  With one child (p-coarsening, copies) that window is P_0^T.  With eight children it is NOT P_i^T: entry (r, c) of the window
 * is PT_flat[stride_P + r nh_i + c], which for equal child degrees is P_j[c][R] with R = (i nH + r) / 8, j = (i nH + r) % 8 -- rows of
 * different children interleaved.  The function's name, its header comment and the smoother's use of the result (a Galerkin coarse
 * operator) all say sum_i P_i^T mat_i P_i, which is what literal_window == 0 forms (P_i^T re-formed from P_i).  Both are kept: the
 * literal form is what a d4est build computes on h-coarsened levels, the exact form is the operator it means to compute
 * (DESIGN.md "MG matrix operator").  The children's blocks of mat come one after another. */
void oracle_compute_PT_mat_P(const double* mat, int degH, int dim, const int* degh, int children, int literal_window, double* PT_mat_P) {
  int volume_nodes_h[8];
  int total_volume_nodes_h = 0, max_volume_nodes_h = -1;
  for (int i = 0; i < children; i++) {
    volume_nodes_h[i] = nodes3(degh[i]);
    total_volume_nodes_h += volume_nodes_h[i];
    max_volume_nodes_h = (volume_nodes_h[i] > max_volume_nodes_h) ? volume_nodes_h[i] : max_volume_nodes_h;
  }
  const int volume_nodes_H = nodes3(degH);
  for (long long i = 0; i < (long long)volume_nodes_H * volume_nodes_H; i++) PT_mat_P[i] = 0.;
  double* P = dalloc0((size_t)total_volume_nodes_h * volume_nodes_H);
  double* PT = dalloc0((size_t)total_volume_nodes_h * volume_nodes_H);
  double* mat_P_i = dalloc0((size_t)max_volume_nodes_h * volume_nodes_H);
  double* PT_mat_P_i = dalloc0((size_t)volume_nodes_H * volume_nodes_H);
  double* PiT = dalloc0((size_t)max_volume_nodes_h * volume_nodes_H);
  oracle_compute_prolong_matrix(degH, dim, degh, children, P);
  oracle_linalg_mat_transpose_nonsqr(P, PT, total_volume_nodes_h, volume_nodes_H);
  long long stride_mat = 0, stride_P = 0;
  for (int i = 0; i < children; i++) {
    oracle_linalg_mat_multiply(&mat[stride_mat], &P[stride_P], mat_P_i, volume_nodes_h[i], volume_nodes_h[i], volume_nodes_H);
    const double* left = &PT[stride_P];
    if (!literal_window) {
      oracle_linalg_mat_transpose_nonsqr(&P[stride_P], PiT, volume_nodes_h[i], volume_nodes_H);
      left = PiT;
    }
    oracle_linalg_mat_multiply(left, mat_P_i, PT_mat_P_i, volume_nodes_H, volume_nodes_h[i], volume_nodes_H);
    oracle_linalg_vec_axpy(1., PT_mat_P_i, PT_mat_P, volume_nodes_H * volume_nodes_H);
    stride_mat += (long long)volume_nodes_h[i] * volume_nodes_h[i];
    stride_P += (long long)volume_nodes_h[i] * volume_nodes_H;
  }
  free(PiT);
  free(PT_mat_P_i);
  free(mat_P_i);
  free(PT);
  free(P);
}

/* child i's left factor of the literal form as a dense (degH+1)^3 x (degh_i+1)^3 row-major matrix (the window of :651), for hosts
 * that want to reproduce it */
void oracle_PT_window(int degH, const int* degh, int children, int child, double* window) {
  int total = 0;
  long long stride_P = 0;
  for (int i = 0; i < children; i++) {
    if (i < child) stride_P += (long long)nodes3(degh[i]) * nodes3(degH);
    total += nodes3(degh[i]);
  }
  const int nH = nodes3(degH), nh = nodes3(degh[child]);
  double* P = dalloc0((size_t)total * nH);
  double* PT = dalloc0((size_t)total * nH);
  oracle_compute_prolong_matrix(degH, 3, degh, children, P);
  oracle_linalg_mat_transpose_nonsqr(P, PT, total, nH);
  memcpy(window, &PT[stride_P], sizeof(double) * (size_t)nH * nh);
  free(PT);
  free(P);
}

/* The coarse-grid walk of the restriction (Solver/d4est_solver_multigrid_callbacks.h:113-208) calling the matrix operator's callback
 * (Solver/d4est_solver_multigrid_matrix_operator.c:6-48) per coarse element.  Item k: hrefine[k] = 0 (p-coarsening or copy, one child
 * of degree degh[8k]) or 1 (eight children degh[8k..8k+7]); the third case of the reference (an element that is not coarsened) is
 * hrefine = 0 with degh = degH, for which compute_PT_mat_P with the identity prolongation copies the block. */
void oracle_mg_matrix_restriction(int n_items, const int* hrefine, const int* degH, const int* degh, int literal_window,
                                  const double* fine_matrix, double* coarse_matrix) {
  long long fine_matrix_stride = 0, coarse_matrix_stride = 0;
  for (int k = 0; k < n_items; k++) {
    const int children = hrefine[k] == 1 ? 8 : 1;
    oracle_compute_PT_mat_P(&fine_matrix[fine_matrix_stride], degH[k], 3, &degh[8 * k], children, literal_window,
                            &coarse_matrix[coarse_matrix_stride]);
    for (int i = 0; i < children; i++) {
      const long long fine_volume_nodes = nodes3(degh[8 * k + i]);
      fine_matrix_stride += fine_volume_nodes * fine_volume_nodes;
    }
    const long long coarse_volume_nodes = nodes3(degH[k]);
    coarse_matrix_stride += coarse_volume_nodes * coarse_volume_nodes;
  }
}

/* Problems/ConstantDensityStar/constant_density_star_fcns.h:485-527: per element dgemv into a scratch vector, then ONE axpy into Au */
void oracle_apply_element_blocks_add(int n_elements, const int* deg, const int* nodal_stride, int local_nodes, const double* matrix,
                                     const double* u, double* Au) {
  double* Mu = dalloc0((size_t)local_nodes);
  long long matrix_stride = 0;
  for (int e = 0; e < n_elements; e++) {
    const int volume_nodes = nodes3(deg[e]);
    oracle_linalg_matvec_plus_vec(1., &matrix[matrix_stride], &u[nodal_stride[e]], 0., &Mu[nodal_stride[e]], volume_nodes, volume_nodes);
    matrix_stride += (long long)volume_nodes * volume_nodes;
  }
  oracle_linalg_vec_axpy(1.0, Mu, Au, local_nodes);
  free(Mu);
}
