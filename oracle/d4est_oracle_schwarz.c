/*
 * oracle/d4est_oracle_schwarz.c -- TEST INFRASTRUCTURE ONLY (see d4est_oracle.h).
 *
 * CPU restatement of the additive Schwarz smoother (single rank, single field):
 *   restrictor / hat weights              Solver/d4est_solver_schwarz_operators.c:7-125, :170-262, :334-397
 *   nodal field -> restricted subdomains  Solver/d4est_solver_schwarz_helpers.c:62-122
 *   subdomain operator                    Solver/d4est_solver_schwarz_laplacian_ext.c:167-358
 *   subdomain CG                          Solver/d4est_solver_schwarz_subdomain_solver_cg.c:101-249
 *   weighting + correction                Solver/d4est_solver_schwarz_helpers.c:344-451,
 *                                         Solver/d4est_solver_schwarz_transfer_ghost_data.c:97-120
 *   d4est_solver_schwarz_iterate          Solver/d4est_solver_schwarz.c:172-285
 *
 * The subdomain operator of the reference applies the stiffness matrix on the subdomain's elements and the SIPG
 * mortar terms with zero_and_skip masks: an element outside the subdomain enters every mortar with u = du/dr = 0 and
 * receives nothing (dGMath/d4est_laplacian_flux.c:486-520, :563-600, :944-962).  That is, term by term, the registered
 * operator (oracle_apply_lhs, homogeneous Dirichlet data) applied to the subdomain field padded with zeros on all
 * other elements and read back on the subdomain's elements -- which is how it is evaluated here.
 */
#include "d4est_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* d4est_solver_schwarz_operators.c:7-40 */
static double quintic_poly(double r) { return (15. * r - 10. * r * r * r + 3. * r * r * r * r * r) / 8.; }
static double phi_fcn(double r) {
  if (r < -1 || r > 1) return (r > 0) - (r < 0);
  return quintic_poly(r);
}
static double poly_hat_weight_fcn(double r, double overlap_size) {
  double d0 = overlap_size;
  return .5 * (phi_fcn((r + 1) / d0) - phi_fcn((r - 1) / d0));
}

/* :42-60  restrictor_1d = [side 0: first restricted_size rows of I | side 1: last restricted_size rows of I], each rs x (deg+1) */
void oracle_schwarz_build_restrictor_1d(double* restrictor_1d, int deg, int restricted_size) {
  int original_size = deg + 1;
  memset(restrictor_1d, 0, sizeof(double) * 2 * original_size * restricted_size);
  for (int i = 0; i < restricted_size; i++) {
    restrictor_1d[(deg + 1) * i + i] = 1.;
    restrictor_1d[original_size * restricted_size + (deg + 1) * (i + 1) - restricted_size + i] = 1.;
  }
}

/* :78-105  [left element | right element | core], the hat centred on the core with the overlap as ramp width */
void oracle_schwarz_build_weights_1d(double* w, int deg, int restricted_size) {
  double r[64], wt[64];
  oracle_lobatto_nodes_and_weights(deg + 1, r, wt);
  double rmax = 1.;
  double rmin = r[deg + 1 - restricted_size];
  double overlap_size = rmax - rmin;
  for (int i = 0; i < restricted_size; i++) {
    w[i] = poly_hat_weight_fcn(r[i + deg + 1 - restricted_size] - 2, overlap_size);
    w[restricted_size + i] = poly_hat_weight_fcn(r[i] + 2, overlap_size);
  }
  for (int i = 0; i < deg + 1; i++) w[2 * restricted_size + i] = poly_hat_weight_fcn(r[i], overlap_size);
}

static void dir_and_side_of_face(int face, int* dir, int* side) { *dir = face / 2; *side = face % 2; } /* dGMath/d4est_reference.c:189-220 */

/* d4est_solver_schwarz_metadata.c:497-506 */
int oracle_schwarz_restricted_nodes(const int* faces, int deg, int restricted_size) {
  int n = 1;
  for (int k = 0; k < 3; k++) n *= (faces[k] != -1) ? restricted_size : (deg + 1);
  return n;
}

/* d4est_solver_schwarz_operators.c:170-262: Kronecker product of identity / restrictor (or its transpose) per direction */
void oracle_schwarz_apply_restrictor(const double* in, const int* faces, int deg, int restricted_size, int transpose, double* out) {
  int degp1 = deg + 1;
  int nodes_in = transpose ? restricted_size : degp1;
  int nodes_out = transpose ? degp1 : restricted_size;
  double* R = (double*)malloc(sizeof(double) * 2 * degp1 * restricted_size);
  double* S = (double*)malloc(sizeof(double) * 2 * degp1 * restricted_size);
  oracle_schwarz_build_restrictor_1d(R, deg, restricted_size);
  if (transpose) { /* :127-146 */
    oracle_linalg_mat_transpose_nonsqr(R, S, restricted_size, degp1);
    oracle_linalg_mat_transpose_nonsqr(R + restricted_size * degp1, S + restricted_size * degp1, restricted_size, degp1);
  } else {
    memcpy(S, R, sizeof(double) * 2 * degp1 * restricted_size);
  }
  double* eyes = (double*)calloc((size_t)degp1 * degp1, sizeof(double));
  for (int i = 0; i < degp1; i++) eyes[i * degp1 + i] = 1.;
  const double* operators[3] = {eyes, eyes, eyes};
  int op_rows[3] = {degp1, degp1, degp1}, op_cols[3] = {degp1, degp1, degp1};
  for (int i = 0; i < 3; i++) {
    if (faces[i] != -1) {
      int dir, side;
      dir_and_side_of_face(faces[i], &dir, &side);
      operators[dir] = &S[side * nodes_in * nodes_out];
      op_rows[dir] = nodes_out;
      op_cols[dir] = nodes_in;
    }
  }
  oracle_kron_A1A2A3x_nonsqr(out, operators[2], operators[1], operators[0], in, op_rows[2], op_cols[2], op_rows[1], op_cols[1],
                             op_rows[0], op_cols[0]);
  free(eyes); free(R); free(S);
}

/* :334-397 (+ Kron/d4est_kron.h:158-179): core weights in the unrestricted directions, left / right ramps in the restricted ones */
void oracle_schwarz_apply_weights(const double* in, const int* core_faces, int deg, int restricted_size, double* out) {
  int degp1 = deg + 1;
  double w[3 * 64];
  oracle_schwarz_build_weights_1d(w, deg, restricted_size);
  const double* vecs[3];
  int vec_sizes[3];
  for (int i = 0; i < 3; i++) { vecs[i] = &w[2 * restricted_size]; vec_sizes[i] = degp1; }
  for (int i = 0; i < 3; i++) {
    if (core_faces[i] != -1) {
      int dir, side;
      dir_and_side_of_face(core_faces[i], &dir, &side);
      vecs[dir] = &w[side * restricted_size];
      vec_sizes[dir] = restricted_size;
    }
  }
  for (int i = 0; i < vec_sizes[2]; i++)
    for (int k = 0; k < vec_sizes[1]; k++)
      for (int m = 0; m < vec_sizes[0]; m++) {
        int stride = (m + (k + i * vec_sizes[1]) * vec_sizes[0]);
        out[stride] = vecs[2][i] * vecs[1][k] * vecs[0][m] * in[stride];
      }
}

/* d4est_solver_schwarz_laplacian_ext.c:167-358 (see the header of this file for the zero-padding argument) */
void oracle_schwarz_apply_over_subdomain(int n_sub_elements, const int* elem, const int* faces, int restricted_size,
                                         const double* u_restricted, double* Au_restricted) {
  int n_elements, local_nodes;
  const int *deg, *nodal_stride;
  oracle_operator_info(&n_elements, &deg, &nodal_stride, &local_nodes);
  double* u = (double*)calloc((size_t)local_nodes, sizeof(double));
  double* Au = (double*)malloc(sizeof(double) * (size_t)local_nodes);
  int rstride = 0;
  for (int j = 0; j < n_sub_elements; j++) { /* restrict-transpose: d4est_solver_schwarz_helpers.c:210-239 */
    int e = elem[j];
    oracle_schwarz_apply_restrictor(&u_restricted[rstride], &faces[3 * j], deg[e], restricted_size, 1, &u[nodal_stride[e]]);
    rstride += oracle_schwarz_restricted_nodes(&faces[3 * j], deg[e], restricted_size);
  }
  oracle_apply_lhs(u, Au);
  rstride = 0;
  for (int j = 0; j < n_sub_elements; j++) { /* :330-337, helpers.c:157-182 */
    int e = elem[j];
    oracle_schwarz_apply_restrictor(&Au[nodal_stride[e]], &faces[3 * j], deg[e], restricted_size, 0, &Au_restricted[rstride]);
    rstride += oracle_schwarz_restricted_nodes(&faces[3 * j], deg[e], restricted_size);
  }
  free(u); free(Au);
}

/* Solver/d4est_solver_schwarz_subdomain_solver_cg.c:101-249 */
static void subdomain_solver_cg(int ne, const int* elem, const int* faces, int rs, int nodes, double* du, const double* rhs, int iter,
                                double atol, double rtol, int* final_iter, double* final_res) {
  double delta_new, delta_old, d_dot_Ad, alpha, beta;
  double* d = (double*)malloc(sizeof(double) * nodes);
  double* Ad = (double*)malloc(sizeof(double) * nodes);
  double* r = (double*)malloc(sizeof(double) * nodes);
  oracle_schwarz_apply_over_subdomain(ne, elem, faces, rs, du, Ad);
  memcpy(r, Ad, sizeof(double) * nodes);
  oracle_linalg_vec_xpby(rhs, -1., r, nodes);
  memcpy(d, r, sizeof(double) * nodes);
  delta_new = oracle_linalg_vec_dot(r, r, nodes);
  double delta_0 = delta_new;
  double tol_break = atol * atol + delta_0 * rtol * rtol;
  int i;
  for (i = 0; i < iter; i++) {
    oracle_schwarz_apply_over_subdomain(ne, elem, faces, rs, d, Ad);
    d_dot_Ad = oracle_linalg_vec_dot(d, Ad, nodes);
    alpha = delta_new / d_dot_Ad;
    oracle_linalg_vec_axpy(alpha, d, du, nodes);
    oracle_linalg_vec_axpy(-alpha, Ad, r, nodes);
    delta_old = delta_new;
    delta_new = oracle_linalg_vec_dot(r, r, nodes);
    beta = delta_new / delta_old;
    oracle_linalg_vec_xpby(r, beta, d, nodes);
    if (delta_new < tol_break) break;
  }
  free(Ad); free(d); free(r);
  *final_iter = i;
  *final_res = sqrt(delta_new);
}

/* Solver/d4est_solver_schwarz.c:172-285.  The subdomain solves only read r, so they are independent; they run on OpenMP threads
 * (register the operator with threads = 1) and the corrections are then added serially in the reference's order. */
void oracle_schwarz_iterate(int n_subdomains, const int* sub_first, const int* sub_elem, const int* sub_faces,
                            const int* sub_core_faces, int restricted_size, int subdomain_iter, double subdomain_atol,
                            double subdomain_rtol, double* u, const double* r, int* final_iter, double* final_res) {
  int n_elements, local_nodes;
  const int *deg, *nodal_stride;
  oracle_operator_info(&n_elements, &deg, &nodal_stride, &local_nodes);
  double** du_all = (double**)calloc((size_t)(n_subdomains > 0 ? n_subdomains : 1), sizeof(double*));
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < n_subdomains; i++) {
    int k0 = sub_first[i], ne = sub_first[i + 1] - sub_first[i];
    const int* elem = &sub_elem[k0];
    const int* faces = &sub_faces[3 * k0];
    int nodes = 0;
    for (int j = 0; j < ne; j++) nodes += oracle_schwarz_restricted_nodes(&faces[3 * j], deg[elem[j]], restricted_size);
    double* restricted_r = (double*)malloc(sizeof(double) * nodes);
    double* du = (double*)calloc((size_t)nodes, sizeof(double));
    int rstride = 0;
    for (int j = 0; j < ne; j++) { /* helpers.c:62-122 */
      int e = elem[j];
      oracle_schwarz_apply_restrictor(&r[nodal_stride[e]], &faces[3 * j], deg[e], restricted_size, 0, &restricted_r[rstride]);
      rstride += oracle_schwarz_restricted_nodes(&faces[3 * j], deg[e], restricted_size);
    }
    int it; double res;
    subdomain_solver_cg(ne, elem, faces, restricted_size, nodes, du, restricted_r, subdomain_iter, subdomain_atol, subdomain_rtol, &it, &res);
    if (final_iter) final_iter[i] = it;
    if (final_res) final_res[i] = res;
    free(restricted_r);
    du_all[i] = du;
  }
  for (int i = 0; i < n_subdomains; i++) {
    int k0 = sub_first[i], ne = sub_first[i + 1] - sub_first[i];
    const int* elem = &sub_elem[k0];
    const int* faces = &sub_faces[3 * k0];
    const int* core_faces = &sub_core_faces[3 * k0];
    const double* du = du_all[i];
    /* weights (helpers.c:344-370), restrict-transpose (:421-451), add in subdomain order (transfer_ghost_data.c:97-120) */
    int rstride = 0;
    for (int j = 0; j < ne; j++) {
      int e = elem[j], n3 = (deg[e] + 1) * (deg[e] + 1) * (deg[e] + 1);
      int rn = oracle_schwarz_restricted_nodes(&faces[3 * j], deg[e], restricted_size);
      double* wdu = (double*)malloc(sizeof(double) * rn);
      double* corr = (double*)malloc(sizeof(double) * n3);
      oracle_schwarz_apply_weights(&du[rstride], &core_faces[3 * j], deg[e], restricted_size, wdu);
      oracle_schwarz_apply_restrictor(wdu, &faces[3 * j], deg[e], restricted_size, 1, corr);
      for (int k = 0; k < n3; k++) u[nodal_stride[e] + k] += corr[k];
      free(wdu); free(corr);
      rstride += rn;
    }
    free(du_all[i]);
  }
  free(du_all);
}

/* d4est_solver_multigrid_smoother_schwarz, Solver/d4est_solver_multigrid_smoother_schwarz.c:98-196 */
void oracle_schwarz_smoother(int n_subdomains, const int* sub_first, const int* sub_elem, const int* sub_faces, const int* sub_core_faces,
                             int restricted_size, int subdomain_iter, double subdomain_atol, double subdomain_rtol, int iterations,
                             double* u, const double* rhs, double* r) {
  int n_elements, local_nodes;
  const int *deg, *nodal_stride;
  oracle_operator_info(&n_elements, &deg, &nodal_stride, &local_nodes);
  for (int i = 0; i < iterations; i++) {
    oracle_apply_lhs(u, r);
    oracle_linalg_vec_xpby(rhs, -1., r, local_nodes);
    oracle_schwarz_iterate(n_subdomains, sub_first, sub_elem, sub_faces, sub_core_faces, restricted_size, subdomain_iter, subdomain_atol,
                           subdomain_rtol, u, r, NULL, NULL);
  }
  oracle_apply_lhs(u, r);
  oracle_linalg_vec_xpby(rhs, -1., r, local_nodes);
}
