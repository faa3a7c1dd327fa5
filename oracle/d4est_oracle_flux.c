/*
 * oracle/d4est_oracle_flux.c -- TEST INFRASTRUCTURE ONLY (see d4est_oracle.h).
 *
 * CPU restatement of the face part of d4est's weak Laplacian on a FLAT side list
 * (one entry per (local element, face) "(-) side", the unit the reference's
 * p4est_iterate face callback hands to the flux function, Mesh/d4est_mortars.c:601-803):
 *   d4est_laplacian_flux_interface / _boundary   dGMath/d4est_laplacian_flux.c:232-1014, :23-230
 *   d4est_laplacian_flux_sipg_interface / _dirichlet  dGMath/d4est_laplacian_flux_sipg.c:494-942, :15-336
 *   d4est_laplacian_apply_aij                     dGMath/d4est_laplacian.c:318-417
 * Covered: conforming mortars (faces_m = faces_p = 1) with different p on the two
 * sides, p4est face re-orientation given as a (flip0, flip1, transpose) code
 * (dGMath/d4est_operators.c:2031-2081), local or ghost (+) side, Dirichlet boundaries
 * with the boundary function evaluated on the Lobatto face nodes
 * (EVAL_BNDRY_FCN_ON_LOBATTO, d4est_laplacian_flux_sipg.c:108-112), Robin boundaries, and hanging (1<->4)
 * mortars between local elements (flux_interface_general).
 */
#include "d4est_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FLUX_ABORT(msg) do { fprintf(stderr, "[ORACLE_ABORT] %s (%s:%d)\n", msg, __FILE__, __LINE__); abort(); } while (0)

static double* dalloc(size_t n) {
  double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
  if (!p) FLUX_ABORT("out of memory");
  return p;
}

/* d4est_laplacian_flux_sipg.c:945-1005 */
double oracle_sipg_penalty(int fcn, int deg_m, double h_m, int deg_p, double h_p, double prefactor) {
  if (fcn == 0) { /* maxp_sqr_over_minh (default) */
    double max_deg = (deg_m > deg_p) ? deg_m : deg_p;
    double min_h = (h_m < h_p) ? h_m : h_p;
    return (prefactor * (max_deg) * (max_deg)) / min_h;
  } else if (fcn == 1) { /* meanp_sqr_over_meanh */
    double mean_p = .5 * (deg_m + deg_p);
    double mean_h = .5 * (h_m + h_p);
    return (prefactor * mean_p * mean_p) / mean_h;
  } else if (fcn == 2) { /* maxpp1_sqr_over_minh */
    double max_deg = (deg_m > deg_p) ? deg_m : deg_p;
    double min_h = (h_m < h_p) ? h_m : h_p;
    return (prefactor * (max_deg + 1) * (max_deg + 1)) / min_h;
  } else if (fcn == 3) { /* mean_p_sqr_over_h */
    double mean_penalty = .5 * (deg_m * deg_m / h_m + deg_p * deg_p / h_p);
    return prefactor * mean_penalty;
  }
  FLUX_ABORT("unknown penalty function");
  return 0.;
}

/* d4est_operators.c:1993-2087 with the p4est_expand_face_transform result passed in as
 * code = flip0 | flip1 << 1 | (not aligned) << 2 ; flips per d4est_operators.c:1951-1991 */
void oracle_reorient_face_data(const double* in, int deg, int code, double* out) {
  int n = deg + 1;
  double* t0 = dalloc((size_t)n * n);
  double* t1 = dalloc((size_t)n * n);
  for (int b = 0; b < n; b++)
    for (int a = 0; a < n; a++) t0[a + n * b] = (code & 1) ? in[(deg - a) + n * b] : in[a + n * b];
  for (int b = 0; b < n; b++)
    for (int a = 0; a < n; a++) t1[a + n * b] = (code & 2) ? t0[a + n * (deg - b)] : t0[a + n * b];
  for (int b = 0; b < n; b++)
    for (int a = 0; a < n; a++) out[a + n * b] = (code & 4) ? t1[b + n * a] : t1[a + n * b];
  free(t0); free(t1);
}

/* 2-D versions of d4est_quadrature_interpolate (d4est_quadrature.c:966-1016, dim == 2 branch) and
 * d4est_quadrature_apply_galerkin_integral with jac = 1 (d4est_quadrature.c:197-201) */
static void interp2d(int quad_type, const double* in, int deg_lobatto, double* out, int deg_quad) {
  int nl = deg_lobatto + 1, nq = deg_quad + 1;
  double* I = dalloc((size_t)nq * nl);
  oracle_quad_interp(quad_type, deg_lobatto, deg_quad, I);
  oracle_kron_A1A2x_nonsqr(out, I, I, in, nq, nl, nq, nl);
  free(I);
}

static void galerkin2d(int quad_type, const double* in_quad, int deg_lobatto, int deg_quad, double* out) {
  int nl = deg_lobatto + 1, nq = deg_quad + 1;
  double* I = dalloc((size_t)nq * nl);
  double* It = dalloc((size_t)nq * nl);
  double* w = dalloc(nq);
  double* wx = dalloc((size_t)nq * nq);
  oracle_quad_interp(quad_type, deg_lobatto, deg_quad, I);
  oracle_linalg_mat_transpose_nonsqr(I, It, nq, nl);
  oracle_quad_weights(quad_type, deg_quad, w);
  /* d4est_kron_vec1_o_vec2_dot_xy with y = ones: w_i w_k x */
  for (int i = 0; i < nq; i++)
    for (int k = 0; k < nq; k++) wx[k + i * nq] = w[i] * w[k] * 1.0 * in_quad[k + i * nq];
  oracle_kron_A1A2x_nonsqr(out, It, It, wx, nl, nq, nl, nq);
  free(I); free(It); free(w); free(wx);
}

typedef struct {
  int quad_type;
  int n_elements;
  const int *deg, *deg_quad, *nodal_stride;
  int n_ghost;
  const int *ghost_deg, *ghost_deg_quad, *ghost_nodal_stride;
  const int *side_nbr, *side_nbr_face, *side_reorder, *side_mortar_stride, *side_bndry_stride;
  const double *sj, *n, *drst_m, *drst_p, *hm, *hp;
  double penalty_prefactor;
  int penalty_fcn;
} flux_ctx_t;

static int nbr_deg(const flux_ctx_t* c, int nbr) { return nbr >= 0 ? c->deg[nbr] : c->ghost_deg[-(nbr + 2)]; }
static int nbr_deg_quad(const flux_ctx_t* c, int nbr) { return nbr >= 0 ? c->deg_quad[nbr] : c->ghost_deg_quad[-(nbr + 2)]; }

/* one conforming interface side: dGMath/d4est_laplacian_flux.c:232-1014 + d4est_laplacian_flux_sipg.c:494-942 */
static void flux_interface_side(const flux_ctx_t* c, int e, int f_m, const double* u, const double* u_ghost,
                                double* const dudr_local[3], double* const dudr_ghost[3], double* Au) {
  const int s = 6 * e + f_m;
  const int nbr = c->side_nbr[s], f_p = c->side_nbr_face[s], code = c->side_reorder[s];
  const int deg_m = c->deg[e], deg_p = nbr_deg(c, nbr);
  const int deg_mq = (c->deg_quad[e] > nbr_deg_quad(c, nbr)) ? c->deg_quad[e] : nbr_deg_quad(c, nbr); /* deg_mortar_quad */
  const int deg_ml = (deg_m > deg_p) ? deg_m : deg_p;                                                 /* deg_mortar_lobatto */
  const int fm = (deg_m + 1) * (deg_m + 1), fp = (deg_p + 1) * (deg_p + 1);
  const int T = (deg_mq + 1) * (deg_mq + 1), TL = (deg_ml + 1) * (deg_ml + 1);
  const int S = c->side_mortar_stride[s];
  const double* sj = &c->sj[S];
  const double* hm = &c->hm[S];
  const double* hp = &c->hp[S];
  const double* nrm[3];
  const double *rm[3][3], *rp[3][3];
  for (int d = 0; d < 3; d++) nrm[d] = &c->n[(size_t)3 * S + (size_t)d * T];
  for (int d1 = 0; d1 < 3; d1++)
    for (int d2 = 0; d2 < 3; d2++) {
      rm[d1][d2] = &c->drst_m[(size_t)9 * S + (size_t)(d1 + 3 * d2) * T];
      rp[d1][d2] = &c->drst_p[(size_t)9 * S + (size_t)(d1 + 3 * d2) * T];
    }
  const double* um = &u[c->nodal_stride[e]];
  const double* up = nbr >= 0 ? &u[c->nodal_stride[nbr]] : &u_ghost[c->ghost_nodal_stride[-(nbr + 2)]];
  const int p_off = nbr >= 0 ? c->nodal_stride[nbr] : c->ghost_nodal_stride[-(nbr + 2)];
  double* const* dudr_p_src = nbr >= 0 ? dudr_local : dudr_ghost;

  double* u_m_f = dalloc(fm); double* u_p_f = dalloc(fp); double* tmp = dalloc(fp > T ? fp : T);
  double* u_m_mortar = dalloc(T); double* u_p_mortar = dalloc(T);
  double* u_m_q = dalloc(T); double* u_p_q = dalloc(T);
  double *dudr_m_q[3], *dudr_p_q[3], *dudx_m[3], *dudx_p_porder[3], *dudx_p[3];
  for (int d = 0; d < 3; d++) { dudr_m_q[d] = dalloc(T); dudr_p_q[d] = dalloc(T); dudx_m[d] = dalloc(T); dudx_p_porder[d] = dalloc(T); dudx_p[d] = dalloc(T); }

  /* traces of u, (+) side re-oriented to the (-) ordering (:575-633) */
  oracle_apply_slicer(um, f_m, deg_m, u_m_f);
  oracle_apply_slicer(up, f_p, deg_p, tmp);
  oracle_reorient_face_data(tmp, deg_p, code, u_p_f);
  /* project onto the mortar space (p-prolong to deg_mortar_quad) and interpolate to the quadrature nodes (:635-694) */
  oracle_apply_p_prolong(u_m_f, deg_m, 2, deg_mq, u_m_mortar);
  oracle_apply_p_prolong(u_p_f, deg_p, 2, deg_mq, u_p_mortar);
  interp2d(c->quad_type, u_m_mortar, deg_mq, u_m_q, deg_mq);
  interp2d(c->quad_type, u_p_mortar, deg_mq, u_p_q, deg_mq);
  /* dudr traces: (-) side in (-) order, (+) side in its OWN order (:698-815) */
  for (int d = 0; d < 3; d++) {
    double* a = dalloc(fm > fp ? fm : fp);
    double* b = dalloc(T);
    oracle_apply_slicer(&dudr_local[d][c->nodal_stride[e]], f_m, deg_m, a);
    oracle_apply_p_prolong(a, deg_m, 2, deg_mq, b);
    interp2d(c->quad_type, b, deg_mq, dudr_m_q[d], deg_mq);
    oracle_apply_slicer(&dudr_p_src[d][p_off], f_p, deg_p, a);
    oracle_apply_p_prolong(a, deg_p, 2, deg_mq, b);
    interp2d(c->quad_type, b, deg_mq, dudr_p_q[d], deg_mq);
    free(a); free(b);
  }
  /* du/dx_j = sum_i (dr_i/dx_j) du/dr_i (:816-856) */
  for (int j = 0; j < 3; j++) {
    for (int k = 0; k < T; k++) { dudx_m[j][k] = 0.; dudx_p_porder[j][k] = 0.; }
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < T; k++) {
        dudx_m[j][k] += rm[i][j][k] * dudr_m_q[i][k];
        dudx_p_porder[j][k] += rp[i][j][k] * dudr_p_q[i][k];
      }
  }
  for (int d = 0; d < 3; d++) oracle_reorient_face_data(dudx_p_porder[d], deg_mq, code, dudx_p[d]); /* :858-900 */

  /* SIPG terms at the mortar quadrature nodes (d4est_laplacian_flux_sipg.c:571-640) */
  double* term1 = dalloc(T); double* term3 = dalloc(T); double* term2[3];
  for (int l = 0; l < 3; l++) term2[l] = dalloc(T);
  for (int k = 0; k < T; k++) {
    double sigma = 1.0 * oracle_sipg_penalty(c->penalty_fcn, deg_m, hm[k], deg_p, hp[k], c->penalty_prefactor);
    term1[k] = 0.;
    for (int d = 0; d < 3; d++) term1[k] += -1. * nrm[d][k] * sj[k] * .5 * (dudx_p[d][k] + dudx_m[d][k]);
    for (int l = 0; l < 3; l++) {
      term2[l][k] = 0.;
      for (int d = 0; d < 3; d++) term2[l][k] += -.5 * rm[l][d][k] * sj[k] * nrm[d][k] * (u_m_q[k] - u_p_q[k]);
    }
    term3[k] = sj[k] * sigma * (u_m_q[k] - u_p_q[k]);
  }
  /* V^T W on the mortar, project onto the (-) side, lift, D^T, accumulate (:641-790, :896-927) */
  const int deg = deg_m, vn = (deg + 1) * (deg + 1) * (deg + 1);
  double* vt = dalloc(TL); double* proj = dalloc(fm); double* lifted = dalloc(vn); double* dt = dalloc(vn);
  double* Au_m = &Au[c->nodal_stride[e]];
  double* acc2 = dalloc(vn); double* acc3 = dalloc(vn); double* acc1 = dalloc(vn);
  double* t2sum = dalloc((size_t)3 * vn);
  for (int l = 0; l < 3; l++) {
    galerkin2d(c->quad_type, term2[l], deg_ml, deg_mq, vt);
    oracle_apply_p_prolong_transpose(vt, deg_ml, 2, deg_m, proj);
    oracle_apply_lift(proj, deg, f_m, lifted);
    oracle_apply_dij_transpose(lifted, deg, l, &t2sum[(size_t)l * vn]);
  }
  galerkin2d(c->quad_type, term1, deg_ml, deg_mq, vt);
  oracle_apply_p_prolong_transpose(vt, deg_ml, 2, deg_m, proj);
  oracle_apply_lift(proj, deg, f_m, acc1);
  galerkin2d(c->quad_type, term3, deg_ml, deg_mq, vt);
  oracle_apply_p_prolong_transpose(vt, deg_ml, 2, deg_m, proj);
  oracle_apply_lift(proj, deg, f_m, acc3);
  for (int i = 0; i < vn; i++) {
    for (int d = 0; d < 3; d++) Au_m[i] += t2sum[(size_t)d * vn + i];
    Au_m[i] += acc3[i];
    Au_m[i] += acc1[i];
  }
  free(u_m_f); free(u_p_f); free(tmp); free(u_m_mortar); free(u_p_mortar); free(u_m_q); free(u_p_q);
  for (int d = 0; d < 3; d++) { free(dudr_m_q[d]); free(dudr_p_q[d]); free(dudx_m[d]); free(dudx_p_porder[d]); free(dudx_p[d]); free(term2[d]); }
  free(term1); free(term3); free(vt); free(proj); free(lifted); free(dt); free(acc1); free(acc2); free(acc3); free(t2sum);
}

/* ---------------------------------------------------------------------------------------------------------
 * Non-conforming (hanging, 1 <-> 4) interfaces: the general form of d4est_laplacian_flux_interface
 * (dGMath/d4est_laplacian_flux.c:232-1014) + d4est_laplacian_flux_sipg_interface (d4est_laplacian_flux_sipg.c:494-942)
 * for faces_m, faces_p in {1, 4}.  Local elements only.
 * --------------------------------------------------------------------------------------------------------- */
static const int* g_side_hang = NULL;        /* 0 conforming, 1 big side (faces_m = 1, faces_p = 4), 2 small side (faces_m = 4, faces_p = 1) */
static const int* g_side_sub = NULL;         /* small side: index of the element in its group */
static const int* g_side_nbr4 = NULL;        /* big side: e_p_oriented[0..3]; small side: the group e_m[0..3] */
static const int* g_side_orientation = NULL; /* p4est orientation */
void oracle_flux_set_hanging(const int* side_hang, const int* side_sub, const int* side_nbr4, const int* side_orientation) {
  g_side_hang = side_hang; g_side_sub = side_sub; g_side_nbr4 = side_nbr4; g_side_orientation = side_orientation;
}

/* the integer tables this file works with, in one place so that tests/test_topology_tables.py can hold them to the reference's own data
 * (oracle_topology_table): dGMath/d4est_reference.c:3-12 and p4est-2.8 src/p8est_connectivity.c:57-63 */
static const int FToF_code[6][6] = {{0, 1, 1, 0, 0, 1}, {2, 0, 0, 1, 1, 0}, {2, 0, 0, 1, 1, 0},
                                    {0, 2, 2, 0, 0, 1}, {0, 2, 2, 0, 0, 1}, {2, 0, 0, 2, 2, 0}};
static const int code_to_perm[3][4] = {{1, 2, 5, 6}, {0, 3, 4, 7}, {0, 4, 3, 7}};
static const int perm_to_order[8][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {1, 0, 3, 2}, {1, 3, 0, 2},
                                        {2, 0, 3, 1}, {2, 3, 0, 1}, {3, 1, 2, 0}, {3, 2, 1, 0}};
static const int face_permutation_refs[6][6] = {{0, 1, 1, 0, 0, 1}, {2, 0, 0, 1, 1, 0}, {2, 0, 0, 1, 1, 0},
                                                {0, 2, 2, 0, 0, 1}, {0, 2, 2, 0, 0, 1}, {2, 0, 0, 2, 2, 0}};
/* ids as d4est_hip_topology_table: 4 p8est_face_permutation_refs, 10 / 11 / 12 the d4est_reference.c tables */
int oracle_topology_table(int id, int* out) {
  const int* src = NULL;
  int n = 0;
  if (id == 4) { src = &face_permutation_refs[0][0]; n = 36; }
  else if (id == 10) { src = &FToF_code[0][0]; n = 36; }
  else if (id == 11) { src = &code_to_perm[0][0]; n = 12; }
  else if (id == 12) { src = &perm_to_order[0][0]; n = 32; }
  else return -1;
  for (int i = 0; out && i < n; i++) out[i] = src[i];
  return n;
}

/* dGMath/d4est_reference.c:3-12 (tables), :84-110 (face_dim == 2 branch) */
int oracle_reorient_face_order(int f_m, int f_p, int o, int i) {
  int perm = code_to_perm[FToF_code[f_m][f_p]][o];
  return perm_to_order[perm][i];
}

/* p4est 2.8, src/p4est_connectivity.c:2877-2944 (p4est_expand_face_transform, P4_TO_P8 branch) -- third-party dependency of the
 * reference (third_party/p4est-2.8.tar.gz, un-vendored); restated from its published algorithm:
 * ftransform[0..2] = my_axis, [3..5] = target_axis, [6..8] = edge_reverse. */
void oracle_expand_face_transform(int iface, int nface, int ftransform[9]) {
  const int target_face = nface % 6, orientation = nface / 6;
  int reverse;
  ftransform[0] = iface < 2 ? 1 : 0;
  ftransform[1] = iface < 4 ? 2 : 1;
  ftransform[2] = iface / 2;
  reverse = face_permutation_refs[0][iface] ^ face_permutation_refs[0][target_face] ^ (orientation == 0 || orientation == 3);
  ftransform[3 + reverse] = target_face < 2 ? 1 : 0;
  ftransform[3 + !reverse] = target_face < 4 ? 2 : 1;
  ftransform[5] = target_face / 2;
  reverse = (face_permutation_refs[iface][target_face] == 1);
  ftransform[6 + reverse] = (orientation & 1);
  ftransform[6 + !reverse] = (orientation >> 1);
  ftransform[8] = 2 * (iface & 1) + (target_face & 1);
}

/* d4est_operators.c:2031-2050: the (flip0, flip1, !aligned) triple of d4est_operators_reorient_face_data as the code
 * oracle_reorient_face_data takes */
int oracle_face_reorder_code(int f_m, int f_p, int o) {
  int ftransform[9];
  oracle_expand_face_transform(((f_m <= f_p) ? f_m : f_p), 6 * o + ((f_m <= f_p) ? f_p : f_m), ftransform);
  int flip0 = ftransform[6];
  int flip1 = ftransform[7];
  int aligned = ((ftransform[1] - ftransform[0]) * (ftransform[4] - ftransform[3]) > 0);
  return flip0 | (flip1 << 1) | ((aligned == 0) << 2);
}

/* Mesh/d4est_mortars.c:550-598 */
static void project_side_onto_mortar_space(const double* in_side, int faces_side, const int* deg_side, double* out_mortar,
                                           int faces_mortar, const int* deg_mortar) {
  if (faces_side == 1 && faces_mortar == 1) oracle_apply_p_prolong(in_side, deg_side[0], 2, deg_mortar[0], out_mortar);
  else if (faces_side == 1 && faces_mortar == 4) oracle_apply_hp_prolong(in_side, deg_side[0], 2, deg_mortar, out_mortar);
  else if (faces_side == 4 && faces_mortar == 4) {
    int ss = 0, sm = 0;
    for (int i = 0; i < 4; i++) {
      oracle_apply_p_prolong(&in_side[ss], deg_side[i], 2, deg_mortar[i], &out_mortar[sm]);
      ss += (deg_side[i] + 1) * (deg_side[i] + 1);
      sm += (deg_mortar[i] + 1) * (deg_mortar[i] + 1);
    }
  } else FLUX_ABORT("project_side_onto_mortar_space");
}

/* Mesh/d4est_mortars.c:510-547 */
static void project_mass_mortar_onto_side(const double* in_mortar, int faces_mortar, const int* deg_mortar, double* out_side,
                                          int faces_side, const int* deg_side) {
  if (faces_side == 1 && faces_mortar == 1) oracle_apply_p_prolong_transpose(in_mortar, deg_mortar[0], 2, deg_side[0], out_side);
  else if (faces_side == 1 && faces_mortar == 4) oracle_apply_hp_prolong_transpose(in_mortar, deg_mortar, 2, deg_side[0], out_side);
  else if (faces_side == 4 && faces_mortar == 4) {
    int ss = 0, sm = 0;
    for (int i = 0; i < 4; i++) {
      oracle_apply_p_prolong_transpose(&in_mortar[sm], deg_mortar[i], 2, deg_side[i], &out_side[ss]);
      ss += (deg_side[i] + 1) * (deg_side[i] + 1);
      sm += (deg_mortar[i] + 1) * (deg_mortar[i] + 1);
    }
  } else FLUX_ABORT("project_mass_mortar_onto_side");
}

/* element references: local id >= 0, or -(g + 2) for ghost element g (e_m_is_ghost / d4est_mesh_get_field_on_element,
 * dGMath/d4est_laplacian_flux.c:513-557, :700-768; a ghost (-) element receives nothing, d4est_laplacian_flux_sipg.c:896-927) */
static int ref_deg(const flux_ctx_t* c, int r) { return r >= 0 ? c->deg[r] : c->ghost_deg[-(r + 2)]; }
static int ref_deg_quad(const flux_ctx_t* c, int r) { return r >= 0 ? c->deg_quad[r] : c->ghost_deg_quad[-(r + 2)]; }
static const double* ref_field(const flux_ctx_t* c, int r, const double* local, const double* ghost) {
  return r >= 0 ? &local[c->nodal_stride[r]] : &ghost[c->ghost_nodal_stride[-(r + 2)]];
}

static void flux_interface_general(const flux_ctx_t* c, const int* e_m, int faces_m, int f_m, const int* e_p_oriented, int faces_p,
                                   int f_p, int orientation, int code, int S, const double* u, const double* u_ghost,
                                   double* const dudr_local[3], double* const dudr_ghost[3], double* Au) {
  const int faces_mortar = (faces_m > faces_p) ? faces_m : faces_p;
  int e_p[4];
  int deg_m_lobatto[4], deg_m_quad[4], deg_p_lobatto[4], deg_p_quad[4], deg_p_lobatto_porder[4];
  int face_nodes_m_lobatto[4], face_nodes_p_lobatto[4];
  int deg_mortar_quad[4], deg_mortar_lobatto[4], nodes_mortar_quad[4], nodes_mortar_lobatto[4];
  int deg_mortar_quad_porder[4], nodes_mortar_quad_porder[4];
  /* e_p_oriented[i] = e_p[reorient(i)]  (Mesh/d4est_element_data.c:130-150) */
  if (faces_p == 1) e_p[0] = e_p_oriented[0];
  else for (int i = 0; i < 4; i++) e_p[oracle_reorient_face_order(f_m, f_p, orientation, i)] = e_p_oriented[i];
  int total_side_nodes_m_lobatto = 0, total_side_nodes_p_lobatto = 0;
  for (int i = 0; i < faces_m; i++) {                                        /* :277-288 */
    deg_m_lobatto[i] = ref_deg(c, e_m[i]); deg_m_quad[i] = ref_deg_quad(c, e_m[i]);
    face_nodes_m_lobatto[i] = (deg_m_lobatto[i] + 1) * (deg_m_lobatto[i] + 1);
    total_side_nodes_m_lobatto += face_nodes_m_lobatto[i];
  }
  for (int i = 0; i < faces_p; i++) {                                        /* :293-305 */
    deg_p_lobatto[i] = ref_deg(c, e_p_oriented[i]); deg_p_quad[i] = ref_deg_quad(c, e_p_oriented[i]);
    deg_p_lobatto_porder[i] = ref_deg(c, e_p[i]);
    face_nodes_p_lobatto[i] = (deg_p_lobatto[i] + 1) * (deg_p_lobatto[i] + 1);
    total_side_nodes_p_lobatto += face_nodes_p_lobatto[i];
  }
  int total_nodes_mortar_quad = 0, total_nodes_mortar_lobatto = 0;
  for (int i = 0; i < faces_m; i++)
    for (int j = 0; j < faces_p; j++) {                                      /* :310-323 */
      deg_mortar_quad[i + j] = (deg_m_quad[i] > deg_p_quad[j]) ? deg_m_quad[i] : deg_p_quad[j];
      deg_mortar_lobatto[i + j] = (deg_m_lobatto[i] > deg_p_lobatto[j]) ? deg_m_lobatto[i] : deg_p_lobatto[j];
      nodes_mortar_quad[i + j] = (deg_mortar_quad[i + j] + 1) * (deg_mortar_quad[i + j] + 1);
      nodes_mortar_lobatto[i + j] = (deg_mortar_lobatto[i + j] + 1) * (deg_mortar_lobatto[i + j] + 1);
      total_nodes_mortar_quad += nodes_mortar_quad[i + j];
      total_nodes_mortar_lobatto += nodes_mortar_lobatto[i + j];
    }
  for (int i = 0; i < faces_mortar; i++) {                                   /* :329-337 */
    int inew = (faces_mortar == 4) ? oracle_reorient_face_order(f_m, f_p, orientation, i) : i;
    deg_mortar_quad_porder[inew] = deg_mortar_quad[i];
    nodes_mortar_quad_porder[inew] = nodes_mortar_quad[i];
  }
  const int TT = total_nodes_mortar_quad;
  const double* sj = &c->sj[S];
  const double* hm = &c->hm[S];
  const double* hp = &c->hp[S];
  const double* nrm[3];
  const double *rm[3][3], *rp[3][3];
  for (int d = 0; d < 3; d++) nrm[d] = &c->n[(size_t)3 * S + (size_t)d * TT];  /* :417-449 */
  for (int d1 = 0; d1 < 3; d1++)
    for (int d2 = 0; d2 < 3; d2++) {
      rm[d1][d2] = &c->drst_m[(size_t)9 * S + (size_t)(d1 + 3 * d2) * TT];
      rp[d1][d2] = &c->drst_p[(size_t)9 * S + (size_t)(d1 + 3 * d2) * TT];
    }
  double* u_m_on_f_m = dalloc(total_side_nodes_m_lobatto);
  double* u_p_on_f_p = dalloc(total_side_nodes_p_lobatto);
  double* u_m_mortar = dalloc(TT); double* u_m_q = dalloc(TT); double* u_p_mortar = dalloc(TT); double* u_p_q = dalloc(TT);
  double* tmp = dalloc(total_side_nodes_p_lobatto + 1);
  int stride = 0;
  for (int i = 0; i < faces_m; i++) {                                        /* :513-557 */
    oracle_apply_slicer(ref_field(c, e_m[i], u, u_ghost), f_m, deg_m_lobatto[i], &u_m_on_f_m[stride]);
    stride += face_nodes_m_lobatto[i];
  }
  stride = 0;
  for (int i = 0; i < faces_p; i++) {                                        /* :575-633 */
    oracle_apply_slicer(ref_field(c, e_p_oriented[i], u, u_ghost), f_p, deg_p_lobatto[i], tmp);
    oracle_reorient_face_data(tmp, deg_p_lobatto[i], code, &u_p_on_f_p[stride]);
    stride += face_nodes_p_lobatto[i];
  }
  project_side_onto_mortar_space(u_m_on_f_m, faces_m, deg_m_lobatto, u_m_mortar, faces_mortar, deg_mortar_quad);   /* :635-645 */
  project_side_onto_mortar_space(u_p_on_f_p, faces_p, deg_p_lobatto, u_p_mortar, faces_mortar, deg_mortar_quad);   /* :647-657 */
  stride = 0;
  for (int f = 0; f < faces_mortar; f++) {                                   /* :659-694 */
    interp2d(c->quad_type, &u_m_mortar[stride], deg_mortar_quad[f], &u_m_q[stride], deg_mortar_quad[f]);
    interp2d(c->quad_type, &u_p_mortar[stride], deg_mortar_quad[f], &u_p_q[stride], deg_mortar_quad[f]);
    stride += nodes_mortar_quad[f];
  }
  double *dudr_m_q[3], *dudr_p_q_porder[3], *dudx_m[3], *dudx_p_porder[3], *dudx_p[3];
  for (int d = 0; d < 3; d++) {
    dudr_m_q[d] = dalloc(TT); dudr_p_q_porder[d] = dalloc(TT); dudx_m[d] = dalloc(TT); dudx_p_porder[d] = dalloc(TT); dudx_p[d] = dalloc(TT);
    double* a_m = dalloc(total_side_nodes_m_lobatto); double* a_p = dalloc(total_side_nodes_p_lobatto);
    double* b = dalloc(TT);
    stride = 0;
    for (int f = 0; f < faces_m; f++) {                                      /* :700-731 */
      oracle_apply_slicer(ref_field(c, e_m[f], dudr_local[d], dudr_ghost[d]), f_m, deg_m_lobatto[f], &a_m[stride]);
      stride += face_nodes_m_lobatto[f];
    }
    stride = 0;
    for (int f = 0; f < faces_p; f++) {                                      /* :733-768: (+) side in ITS order */
      oracle_apply_slicer(ref_field(c, e_p[f], dudr_local[d], dudr_ghost[d]), f_p, deg_p_lobatto_porder[f], &a_p[stride]);
      stride += (deg_p_lobatto_porder[f] + 1) * (deg_p_lobatto_porder[f] + 1);
    }
    project_side_onto_mortar_space(a_p, faces_p, deg_p_lobatto_porder, b, faces_mortar, deg_mortar_quad_porder);   /* :770-780 */
    stride = 0;
    for (int f = 0; f < faces_mortar; f++) {                                 /* :812-830 */
      interp2d(c->quad_type, &b[stride], deg_mortar_quad_porder[f], &dudr_p_q_porder[d][stride], deg_mortar_quad_porder[f]);
      stride += nodes_mortar_quad_porder[f];
    }
    project_side_onto_mortar_space(a_m, faces_m, deg_m_lobatto, b, faces_mortar, deg_mortar_quad);                 /* :782-792 */
    stride = 0;
    for (int f = 0; f < faces_mortar; f++) {                                 /* :794-810 */
      interp2d(c->quad_type, &b[stride], deg_mortar_quad[f], &dudr_m_q[d][stride], deg_mortar_quad[f]);
      stride += nodes_mortar_quad[f];
    }
    free(a_m); free(a_p); free(b);
  }
  for (int j = 0; j < 3; j++) {                                              /* :833-856 */
    for (int k = 0; k < TT; k++) { dudx_m[j][k] = 0.; dudx_p_porder[j][k] = 0.; }
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < TT; k++) {
        dudx_m[j][k] += rm[i][j][k] * dudr_m_q[i][k];
        dudx_p_porder[j][k] += rp[i][j][k] * dudr_p_q_porder[i][k];
      }
  }
  int face_mortar_stride = 0;
  for (int face = 0; face < faces_mortar; face++) {                          /* :858-900 */
    int face_p = (faces_mortar == 4) ? oracle_reorient_face_order(f_m, f_p, orientation, face) : face;
    int oriented_face_mortar_stride = 0;
    for (int b = 0; b < face_p; b++) oriented_face_mortar_stride += nodes_mortar_quad_porder[b];
    for (int d = 0; d < 3; d++)
      oracle_reorient_face_data(&dudx_p_porder[d][oriented_face_mortar_stride], deg_mortar_quad[face], code, &dudx_p[d][face_mortar_stride]);
    face_mortar_stride += nodes_mortar_quad[face];
  }
  if (faces_m != faces_mortar)                                               /* :905-915 "another dr/dx factor" */
    for (int d = 0; d < 3; d++) for (int k = 0; k < TT; k++) dudx_m[d][k] *= .5;
  if (faces_p != faces_mortar)
    for (int d = 0; d < 3; d++) for (int k = 0; k < TT; k++) dudx_p[d][k] *= .5;

  /* ---- d4est_laplacian_flux_sipg_interface_aux (d4est_laplacian_flux_sipg.c:494-833) */
  double* term1 = dalloc(TT); double* term3 = dalloc(TT); double* term2[3];
  double* vt1 = dalloc(total_nodes_mortar_lobatto); double* vt3 = dalloc(total_nodes_mortar_lobatto); double* vt2[3];
  for (int l = 0; l < 3; l++) { term2[l] = dalloc(TT); vt2[l] = dalloc(total_nodes_mortar_lobatto); }
  stride = 0;
  int stride_lobatto = 0;
  for (int f = 0; f < faces_mortar; f++) {
    for (int k = 0; k < nodes_mortar_quad[f]; k++) {
      int ks = k + stride;
      double sigma = 1.0 * oracle_sipg_penalty(c->penalty_fcn, (faces_m == faces_mortar) ? deg_m_lobatto[f] : deg_m_lobatto[0], hm[ks],
                                               (faces_p == faces_mortar) ? deg_p_lobatto[f] : deg_p_lobatto[0], hp[ks],
                                               c->penalty_prefactor);                                        /* :577-592 */
      term1[ks] = 0.;
      for (int d = 0; d < 3; d++) term1[ks] += -1. * nrm[d][ks] * sj[ks] * .5 * (dudx_p[d][ks] + dudx_m[d][ks]);
      for (int l = 0; l < 3; l++) {
        term2[l][ks] = 0.;
        for (int d = 0; d < 3; d++) term2[l][ks] += -.5 * rm[l][d][ks] * sj[ks] * nrm[d][ks] * (u_m_q[ks] - u_p_q[ks]);
      }
      term3[ks] = sj[ks] * sigma * (u_m_q[ks] - u_p_q[ks]);
    }
    galerkin2d(c->quad_type, &term1[stride], deg_mortar_lobatto[f], deg_mortar_quad[f], &vt1[stride_lobatto]);  /* :641-690 */
    for (int d = 0; d < 3; d++) galerkin2d(c->quad_type, &term2[d][stride], deg_mortar_lobatto[f], deg_mortar_quad[f], &vt2[d][stride_lobatto]);
    galerkin2d(c->quad_type, &term3[stride], deg_mortar_lobatto[f], deg_mortar_quad[f], &vt3[stride_lobatto]);
    stride += nodes_mortar_quad[f];
    stride_lobatto += nodes_mortar_lobatto[f];
  }
  double* proj1 = dalloc(total_side_nodes_m_lobatto); double* proj3 = dalloc(total_side_nodes_m_lobatto); double* proj2[3];
  for (int d = 0; d < 3; d++) {                                              /* :732-768 */
    proj2[d] = dalloc(total_side_nodes_m_lobatto);
    project_mass_mortar_onto_side(vt2[d], faces_mortar, deg_mortar_lobatto, proj2[d], faces_m, deg_m_lobatto);
  }
  project_mass_mortar_onto_side(vt1, faces_mortar, deg_mortar_lobatto, proj1, faces_m, deg_m_lobatto);
  project_mass_mortar_onto_side(vt3, faces_mortar, deg_mortar_lobatto, proj3, faces_m, deg_m_lobatto);
  stride = 0;
  for (int f = 0; f < faces_m; f++) {                                        /* :770-826, :896-927 */
    const int deg = deg_m_lobatto[f], vn = (deg + 1) * (deg + 1) * (deg + 1);
    if (e_m[f] < 0) { stride += face_nodes_m_lobatto[f]; continue; }         /* e_m_is_ghost: handled by its owner */
    double* lifted = dalloc(vn); double* dt = dalloc(vn); double* l1 = dalloc(vn); double* l3 = dalloc(vn);
    double* t2sum = dalloc((size_t)3 * vn);
    oracle_apply_lift(&proj1[stride], deg, f_m, l1);
    for (int d = 0; d < 3; d++) {
      oracle_apply_lift(&proj2[d][stride], deg, f_m, lifted);
      oracle_apply_dij_transpose(lifted, deg, d, &t2sum[(size_t)d * vn]);
      if (faces_m != faces_mortar) for (int i = 0; i < vn; i++) t2sum[(size_t)d * vn + i] *= .5;
    }
    oracle_apply_lift(&proj3[stride], deg, f_m, l3);
    double* Au_m = &Au[c->nodal_stride[e_m[f]]];
    for (int i = 0; i < vn; i++) {
      for (int d = 0; d < 3; d++) Au_m[i] += t2sum[(size_t)d * vn + i];
      Au_m[i] += l3[i];
      Au_m[i] += l1[i];
    }
    free(lifted); free(dt); free(l1); free(l3); free(t2sum);
    stride += face_nodes_m_lobatto[f];
  }
  free(u_m_on_f_m); free(u_p_on_f_p); free(u_m_mortar); free(u_m_q); free(u_p_mortar); free(u_p_q); free(tmp);
  for (int d = 0; d < 3; d++) { free(dudr_m_q[d]); free(dudr_p_q_porder[d]); free(dudx_m[d]); free(dudx_p_porder[d]); free(dudx_p[d]); free(term2[d]); free(vt2[d]); free(proj2[d]); }
  free(term1); free(term3); free(vt1); free(vt3); free(proj1); free(proj3);
}

/* Robin boundary data (BC_ROBIN): the callbacks robin_coeff / robin_rhs of d4est_laplacian_robin_bc_t evaluated at the
 * boundary mortar quadrature nodes (d4est_laplacian_flux_sipg.c:388-412), indexed like sj.  NULL = Dirichlet. */
static const double* g_robin_coeff = NULL;
static const double* g_robin_rhs = NULL;
void oracle_flux_set_robin(const double* coeff_quad, const double* rhs_quad) {
  g_robin_coeff = coeff_quad;
  g_robin_rhs = rhs_quad;
}

/* one Robin boundary side: d4est_laplacian_flux.c:23-230 (trace + interpolation) + d4est_laplacian_flux_sipg.c:339-489 */
static void flux_robin_side(const flux_ctx_t* c, int e, int f_m, const double* u, double* Au) {
  const int s = 6 * e + f_m;
  const int deg = c->deg[e], deg_q = c->deg_quad[e];
  const int fm = (deg + 1) * (deg + 1), T = (deg_q + 1) * (deg_q + 1), vn = fm * (deg + 1);
  const int S = c->side_mortar_stride[s];
  const double* sj = &c->sj[S];
  double* u_f = dalloc(fm); double* u_q = dalloc(T); double* term1 = dalloc(T);
  double* vt = dalloc(fm); double* lifted = dalloc(vn);
  oracle_apply_slicer(&u[c->nodal_stride[e]], f_m, deg, u_f);
  interp2d(c->quad_type, u_f, deg, u_q, deg_q);
  for (int k = 0; k < T; k++) term1[k] = sj[k] * (g_robin_coeff[S + k] * u_q[k] - g_robin_rhs[S + k]); /* :414-415 */
  galerkin2d(c->quad_type, term1, deg, deg_q, vt);                                                       /* :419-432 */
  oracle_apply_lift(vt, deg, f_m, lifted);                                                               /* :435-441 */
  double* Au_m = &Au[c->nodal_stride[e]];
  for (int i = 0; i < vn; i++) Au_m[i] += lifted[i];                                                     /* :480-483 */
  free(u_f); free(u_q); free(term1); free(vt); free(lifted);
}

/* one Dirichlet boundary side: d4est_laplacian_flux.c:23-230 + d4est_laplacian_flux_sipg.c:15-336 */
static void flux_boundary_side(const flux_ctx_t* c, int e, int f_m, const double* u, double* const dudr_local[3],
                               const double* bndry_lobatto, double* Au) {
  const int s = 6 * e + f_m;
  const int deg = c->deg[e], deg_q = c->deg_quad[e];
  const int fm = (deg + 1) * (deg + 1), T = (deg_q + 1) * (deg_q + 1), vn = fm * (deg + 1);
  const int S = c->side_mortar_stride[s];
  const double* sj = &c->sj[S];
  const double* h = &c->hm[S];
  const double* nrm[3];
  const double* r[3][3];
  for (int d = 0; d < 3; d++) nrm[d] = &c->n[(size_t)3 * S + (size_t)d * T];
  for (int d1 = 0; d1 < 3; d1++)
    for (int d2 = 0; d2 < 3; d2++) r[d1][d2] = &c->drst_m[(size_t)9 * S + (size_t)(d1 + 3 * d2) * T];
  double* u_f = dalloc(fm); double* u_q = dalloc(T); double* g_q = dalloc(T);
  double *dudr_q[3], *dudx[3], *term2[3];
  double* a = dalloc(fm);
  for (int d = 0; d < 3; d++) {
    dudr_q[d] = dalloc(T); dudx[d] = dalloc(T); term2[d] = dalloc(T);
    oracle_apply_slicer(&dudr_local[d][c->nodal_stride[e]], f_m, deg, a);
    interp2d(c->quad_type, a, deg, dudr_q[d], deg_q);
  }
  oracle_apply_slicer(&u[c->nodal_stride[e]], f_m, deg, u_f);
  for (int j = 0; j < 3; j++) {
    for (int k = 0; k < T; k++) dudx[j][k] = 0.;
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < T; k++) dudx[j][k] += r[i][j][k] * dudr_q[i][k];
  }
  interp2d(c->quad_type, u_f, deg, u_q, deg_q);
  /* boundary values on the Lobatto face nodes, interpolated to the quadrature nodes (sipg.c:80-107) */
  const double* g = (bndry_lobatto && c->side_bndry_stride) ? &bndry_lobatto[c->side_bndry_stride[s]] : NULL;
  if (g) interp2d(c->quad_type, g, deg, g_q, deg_q);
  else for (int k = 0; k < T; k++) g_q[k] = 0.;
  double* term1 = dalloc(T); double* term3 = dalloc(T);
  for (int k = 0; k < T; k++) {
    double sigma = 1.0 * oracle_sipg_penalty(c->penalty_fcn, deg, h[k], deg, h[k], c->penalty_prefactor);
    double du = u_q[k] - g_q[k];
    term1[k] = 0.;
    for (int d = 0; d < 3; d++) term1[k] += -1. * nrm[d][k] * sj[k] * (dudx[d][k]);
    for (int l = 0; l < 3; l++) {
      term2[l][k] = 0.;
      for (int d = 0; d < 3; d++) term2[l][k] += -.5 * r[l][d][k] * nrm[d][k] * sj[k] * 2. * du;
    }
    term3[k] = sj[k] * sigma * du;
  }
  double* vt = dalloc(fm); double* lifted = dalloc(vn); double* t2 = dalloc((size_t)3 * vn); double* l1 = dalloc(vn); double* l3 = dalloc(vn);
  galerkin2d(c->quad_type, term1, deg, deg_q, vt);
  oracle_apply_lift(vt, deg, f_m, l1);
  for (int d = 0; d < 3; d++) {
    galerkin2d(c->quad_type, term2[d], deg, deg_q, vt);
    oracle_apply_lift(vt, deg, f_m, lifted);
    oracle_apply_dij_transpose(lifted, deg, d, &t2[(size_t)d * vn]);
  }
  galerkin2d(c->quad_type, term3, deg, deg_q, vt);
  oracle_apply_lift(vt, deg, f_m, l3);
  double* Au_m = &Au[c->nodal_stride[e]];
  for (int i = 0; i < vn; i++) { /* sipg.c:318-326 */
    for (int d = 0; d < 3; d++) Au_m[i] += t2[(size_t)d * vn + i];
    Au_m[i] += l3[i];
    Au_m[i] += l1[i];
  }
  free(u_f); free(u_q); free(g_q); free(a); free(term1); free(term3); free(vt); free(lifted); free(t2); free(l1); free(l3);
  for (int d = 0; d < 3; d++) { free(dudr_q[d]); free(dudx[d]); free(term2[d]); }
}

/* d4est_laplacian_apply_aij (dGMath/d4est_laplacian.c:318-417): stiffness (overwrites Au), ghost data assumed
 * exchanged (u_ghost), dudr on local + ghost elements, then the mortar terms accumulated into Au. */
void oracle_laplacian_apply_aij(int quad_type, int n_elements, const int* deg, const int* deg_quad, const int* nodal_stride,
                                const int* quad_stride, int local_nodes, int local_nodes_quad, const double* J_quad,
                                const double* rst_xyz_quad, int n_ghost, const int* ghost_deg, const int* ghost_deg_quad,
                                const int* ghost_nodal_stride, int ghost_nodes, const int* side_nbr, const int* side_nbr_face,
                                const int* side_reorder, const int* side_mortar_stride, const int* side_bndry_stride,
                                const double* sj, const double* n, const double* drst_m, const double* drst_p,
                                const double* hm, const double* hp, double penalty_prefactor, int penalty_fcn,
                                const double* u, const double* u_ghost, const double* bndry_lobatto, double* Au,
                                int stiffness_threads) {
  flux_ctx_t c = {quad_type, n_elements, deg, deg_quad, nodal_stride, n_ghost, ghost_deg, ghost_deg_quad, ghost_nodal_stride,
                  side_nbr, side_nbr_face, side_reorder, side_mortar_stride, side_bndry_stride, sj, n, drst_m, drst_p, hm, hp,
                  penalty_prefactor, penalty_fcn};
  oracle_laplacian_apply_stiffness_matrix(quad_type, n_elements, deg, deg_quad, nodal_stride, quad_stride, local_nodes_quad,
                                          J_quad, rst_xyz_quad, u, Au, stiffness_threads);
  double *dl[3], *dg[3];
  for (int d = 0; d < 3; d++) { dl[d] = dalloc(local_nodes); dg[d] = dalloc(ghost_nodes); }
  oracle_laplacian_compute_dudr(n_elements, deg, nodal_stride, u, dl[0], dl[1], dl[2]);
  if (n_ghost > 0) oracle_laplacian_compute_dudr(n_ghost, ghost_deg, ghost_nodal_stride, u_ghost, dg[0], dg[1], dg[2]);
  for (int e = 0; e < n_elements; e++)
    for (int f = 0; f < 6; f++) {
      if (side_nbr[6 * e + f] == -1) {
        if (g_robin_coeff) flux_robin_side(&c, e, f, u, Au);
        else flux_boundary_side(&c, e, f, u, dl, bndry_lobatto, Au);
      }
      else if (g_side_hang && g_side_hang[6 * e + f] == 1) {
        int em[1] = {e};
        flux_interface_general(&c, em, 1, f, &g_side_nbr4[4 * (6 * e + f)], 4, side_nbr_face[6 * e + f], g_side_orientation[6 * e + f],
                               side_reorder[6 * e + f], side_mortar_stride[6 * e + f], u, u_ghost, dl, dg, Au);
      } else if (g_side_hang && g_side_hang[6 * e + f] == 2) {
        /* the reference's callback handles the 4 hanging elements of the face together: once, here at the group's first LOCAL
         * member (ghost members are read but receive nothing, Mesh/d4est_mortars.c:655-700) */
        const int* grp = &g_side_nbr4[4 * (6 * e + f)];
        int first_local = -1;
        for (int i = 0; i < 4 && first_local < 0; i++) if (grp[i] >= 0) first_local = grp[i];
        if (first_local == e) {
          int ep[1] = {side_nbr[6 * e + f]};
          flux_interface_general(&c, grp, 4, f, ep, 1, side_nbr_face[6 * e + f], g_side_orientation[6 * e + f],
                                 side_reorder[6 * e + f], side_mortar_stride[6 * e + f], u, u_ghost, dl, dg, Au);
        }
      }
      else flux_interface_side(&c, e, f, u, u_ghost, dl, dg, Au);
    }
  for (int d = 0; d < 3; d++) { free(dl[d]); free(dg[d]); }
}
