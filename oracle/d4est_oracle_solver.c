/*
 * oracle/d4est_oracle_solver.c -- TEST INFRASTRUCTURE ONLY (see d4est_oracle.h).
 *
 * CPU restatement of the Chebyshev smoother inner loop and the CG-Lanczos spectral bound:
 *   d4est_solver_multigrid_smoother_cheby_iterate_aux   Solver/d4est_solver_multigrid_smoother_cheby.c:81-176
 *   cg_eigs + tridiag_gershgorin[_new]                  Solver/d4est_solver_cg_eigs.c:9-64, :116-275
 * apply_lhs is the weak Laplacian of d4est_oracle_flux.c (as in Problems/Poisson/poisson_sinx_fcns.h:110-128),
 * registered once with oracle_set_aij_operator (the pointers must stay alive).
 */
#include "d4est_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static struct {
  int set;
  int quad_type, n_elements;
  const int *deg, *deg_quad, *nodal_stride, *quad_stride;
  int local_nodes, local_nodes_quad;
  const double *J_quad, *rst_xyz_quad;
  int n_ghost;
  const int *ghost_deg, *ghost_deg_quad, *ghost_nodal_stride;
  int ghost_nodes;
  const int *side_nbr, *side_nbr_face, *side_reorder, *side_mortar_stride, *side_bndry_stride;
  const double *sj, *n, *drst_m, *drst_p, *hm, *hp;
  double penalty_prefactor;
  int penalty_fcn;
  int threads;
} g_op;

void oracle_set_aij_operator(int quad_type, int n_elements, const int* deg, const int* deg_quad, const int* nodal_stride,
                             const int* quad_stride, int local_nodes, int local_nodes_quad, const double* J_quad,
                             const double* rst_xyz_quad, const int* side_nbr, const int* side_nbr_face, const int* side_reorder,
                             const int* side_mortar_stride, const int* side_bndry_stride, const double* sj, const double* n,
                             const double* drst_m, const double* drst_p, const double* hm, const double* hp,
                             double penalty_prefactor, int penalty_fcn, int threads) {
  g_op.set = 1;
  g_op.quad_type = quad_type; g_op.n_elements = n_elements; g_op.deg = deg; g_op.deg_quad = deg_quad;
  g_op.nodal_stride = nodal_stride; g_op.quad_stride = quad_stride; g_op.local_nodes = local_nodes;
  g_op.local_nodes_quad = local_nodes_quad; g_op.J_quad = J_quad; g_op.rst_xyz_quad = rst_xyz_quad;
  g_op.n_ghost = 0; g_op.ghost_deg = g_op.ghost_deg_quad = g_op.ghost_nodal_stride = NULL; g_op.ghost_nodes = 0;
  g_op.side_nbr = side_nbr; g_op.side_nbr_face = side_nbr_face; g_op.side_reorder = side_reorder;
  g_op.side_mortar_stride = side_mortar_stride; g_op.side_bndry_stride = side_bndry_stride;
  g_op.sj = sj; g_op.n = n; g_op.drst_m = drst_m; g_op.drst_p = drst_p; g_op.hm = hm; g_op.hp = hp;
  g_op.penalty_prefactor = penalty_prefactor; g_op.penalty_fcn = penalty_fcn; g_op.threads = threads;
}

/* apply_lhs: Au = A u with homogeneous Dirichlet data (single rank: no ghosts) */
static void apply_lhs(const double* u, double* Au);
void oracle_apply_lhs(const double* u, double* Au) { apply_lhs(u, Au); }
/* zeroth-order term of a linearised nonlinear problem: + sum_e d4est_quadrature_apply_fofufofvlilj(u_e; coeff) added with axpy 1.0
 * (Problems/ConstantDensityStar/constant_density_star_fcns.h:528-603, :777-850); NULL = pure Laplacian */
static const double* g_lhs_coeff = NULL;
void oracle_set_lhs_coefficient(const double* coeff_quad) { g_lhs_coeff = coeff_quad; }
/* the same term on a coarse multigrid level, where the reference holds it as Galerkin-restricted dense element blocks
 * (constant_density_star_apply_jac, constant_density_star_fcns.h:806-850 with :485-527) */
static const double* g_lhs_blocks = NULL;
void oracle_set_lhs_element_blocks(const double* matrix) { g_lhs_blocks = matrix; }
void oracle_operator_info(int* n_elements, const int** deg, const int** nodal_stride, int* local_nodes) {
  if (!g_op.set) { fprintf(stderr, "[ORACLE_ABORT] operator not set\n"); abort(); }
  *n_elements = g_op.n_elements; *deg = g_op.deg; *nodal_stride = g_op.nodal_stride; *local_nodes = g_op.local_nodes;
}
static void apply_lhs(const double* u, double* Au) {
  if (!g_op.set) { fprintf(stderr, "[ORACLE_ABORT] operator not set\n"); abort(); }
  double dummy = 0.;
  oracle_laplacian_apply_aij(g_op.quad_type, g_op.n_elements, g_op.deg, g_op.deg_quad, g_op.nodal_stride, g_op.quad_stride,
                             g_op.local_nodes, g_op.local_nodes_quad, g_op.J_quad, g_op.rst_xyz_quad, 0, NULL, NULL, NULL, 0,
                             g_op.side_nbr, g_op.side_nbr_face, g_op.side_reorder, g_op.side_mortar_stride,
                             g_op.side_bndry_stride, g_op.sj, g_op.n, g_op.drst_m, g_op.drst_p, g_op.hm, g_op.hp,
                             g_op.penalty_prefactor, g_op.penalty_fcn, u, &dummy, NULL, Au, g_op.threads);
  if (g_lhs_coeff) {
    double* Mu = (double*)malloc(sizeof(double) * (size_t)g_op.local_nodes);
    oracle_elements_apply_weighted_mass_matrix(g_op.quad_type, g_op.n_elements, g_op.deg, g_op.deg_quad, g_op.nodal_stride, g_op.quad_stride,
                                               g_op.J_quad, g_lhs_coeff, u, Mu);
    oracle_linalg_vec_axpy(1.0, Mu, Au, g_op.local_nodes);
    free(Mu);
  }
  if (g_lhs_blocks) oracle_apply_element_blocks_add(g_op.n_elements, g_op.deg, g_op.nodal_stride, g_op.local_nodes, g_lhs_blocks, u, Au);
}

/* Solver/d4est_solver_multigrid_smoother_cheby.c:81-176 */
void oracle_cheby_iterate_aux(double* u, const double* rhs, double* Au, double* r, int iter, double lmin, double lmax,
                              int compute_residual_at_end) {
  const int local_nodes = g_op.local_nodes;
  double d = (lmax + lmin) * .5;
  double c = (lmax - lmin) * .5;
  double alpha = 0., beta;
  double* p = (double*)calloc(local_nodes > 0 ? local_nodes : 1, sizeof(double));
  for (int i = 0; i < iter; i++) {
    apply_lhs(u, Au);
    memcpy(r, Au, sizeof(double) * local_nodes);
    oracle_linalg_vec_xpby(rhs, -1., r, local_nodes);
    if (i == 0) alpha = 1. / d;
    else if (i == 1) alpha = 2. * d / (2 * d * d - c * c);
    else alpha = 1. / (d - (alpha * c * c / 4.));
    beta = alpha * d - 1.;
    oracle_linalg_vec_scale(alpha, r, local_nodes);
    oracle_linalg_vec_xpby(r, beta, p, local_nodes);
    oracle_linalg_vec_axpy(1., p, u, local_nodes);
  }
  if (compute_residual_at_end == 1) {
    apply_lhs(u, Au);
    memcpy(r, Au, sizeof(double) * local_nodes);
    oracle_linalg_vec_xpby(rhs, -1., r, local_nodes);
  }
  free(p);
}

/* Solver/d4est_solver_cg_eigs.c:9-33 */
static void tridiag_gershgorin(int i, int local_nodes, double a0, double b0, double a1, double b1, double* max, double* min) {
  double diag, offdiag_sum;
  if (i != 0 && i < local_nodes - 1) {
    diag = (1. / a1 + b0 / a0);
    offdiag_sum = fabs(sqrt(b1) / a1) + fabs(sqrt(b0) / a0);
  } else if (i == 0) {
    diag = 1. / a1;
    offdiag_sum = sqrt(b1) / a1;
  } else {
    diag = 1. / a1 + b0 / a0;
    offdiag_sum = fabs(sqrt(b0) / a0);
  }
  *max = diag + offdiag_sum;
  *min = diag - offdiag_sum;
}

/* Solver/d4est_solver_cg_eigs.c:36-64 */
static void tridiag_gershgorin_new(int i, int local_nodes, double a0, double b0, double a1, double b1, double* max, double* min) {
  double diag, offdiag_sum;
  (void)local_nodes;
  if (i != 0) {
    diag = (1. / a1 + b0 / a0);
    offdiag_sum = fabs(sqrt(b0) / a0);
  } else {
    diag = 1. / a1;
    offdiag_sum = sqrt(b1) / a1;
  }
  *max = diag + offdiag_sum;
  *min = diag - offdiag_sum;
}

/* Solver/d4est_solver_cg_eigs.c:116-275 (single rank: the three sc_allreduce are identities).
 * NB: as in the reference, the CG iteration advances u. */
void oracle_cg_eigs(double* u, const double* rhs, double* Au, int imax, int use_new, double* spectral_bound) {
  const int local_nodes = g_op.local_nodes;
  double delta_new, delta_old;
  double alpha = -1., beta = -1.;
  double* d = (double*)malloc(sizeof(double) * (local_nodes > 0 ? local_nodes : 1));
  double* r = (double*)malloc(sizeof(double) * (local_nodes > 0 ? local_nodes : 1));
  apply_lhs(u, Au);
  memcpy(r, Au, sizeof(double) * local_nodes);
  oracle_linalg_vec_xpby(rhs, -1., r, local_nodes);
  memcpy(d, r, sizeof(double) * local_nodes);
  delta_new = oracle_linalg_vec_dot(r, r, local_nodes);
  double alpha_old, beta_old, temp_max, temp_min;
  for (int i = 0; i < imax; i++) {
    apply_lhs(d, Au);
    double d_dot_Au = oracle_linalg_vec_dot(d, Au, local_nodes);
    alpha_old = alpha;
    alpha = delta_new / d_dot_Au;
    oracle_linalg_vec_axpy(alpha, d, u, local_nodes);
    oracle_linalg_vec_axpy(-alpha, Au, r, local_nodes);
    delta_old = delta_new;
    delta_new = oracle_linalg_vec_dot(r, r, local_nodes);
    beta_old = beta;
    beta = delta_new / delta_old;
    oracle_linalg_vec_xpby(r, beta, d, local_nodes);
    if (!use_new) tridiag_gershgorin(i, local_nodes, alpha_old, beta_old, alpha, beta, &temp_max, &temp_min);
    else tridiag_gershgorin_new(i, local_nodes, alpha_old, beta_old, alpha, beta, &temp_max, &temp_min);
    if (i > 0) *spectral_bound = (*spectral_bound > temp_max) ? *spectral_bound : temp_max;
    else *spectral_bound = temp_max;
  }
  free(d);
  free(r);
}

/* host-side bound from recorded (alpha_i, beta_i), i = 0..imax-1: shared with nothing in the product;
 * lets tests feed the device-recorded histories through the reference formula. */
double oracle_gershgorin_bound(const double* alpha_h, const double* beta_h, int imax, int local_nodes, int use_new) {
  double bound = 0., tmax, tmin, a_old = -1., b_old = -1.;
  for (int i = 0; i < imax; i++) {
    if (!use_new) tridiag_gershgorin(i, local_nodes, a_old, b_old, alpha_h[i], beta_h[i], &tmax, &tmin);
    else tridiag_gershgorin_new(i, local_nodes, a_old, b_old, alpha_h[i], beta_h[i], &tmax, &tmin);
    bound = (i > 0) ? ((bound > tmax) ? bound : tmax) : tmax;
    a_old = alpha_h[i];
    b_old = beta_h[i];
  }
  return bound;
}
