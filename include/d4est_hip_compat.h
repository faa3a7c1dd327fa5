/*
 * d4est_hip_compat.h -- the reference's OWN entry points for the hot path, exported by libd4est_hip_compat.so.
 *
 * "Functions a C-ABI replacement must export (same signatures, extern "C")".  Every prototype below is the
 * reference's, name and argument list unchanged (file:line beside each), so a d4est build can drop the object files that define
 * them (or link this library in front of them) and keep every caller untouched.  The struct arguments the reference passes by
 * pointer are forward-declared opaque here -- the shims never look inside them, with two exceptions stated below -- so this header
 * needs no p4est / d4est header.  A translation unit that already includes the reference's headers must NOT include this one
 * (the prototypes are identical; the opaque typedefs would collide): define D4EST_HIP_COMPAT_NO_TYPES to get only the binding API.
 *
 * Layout facts relied on (and nothing else):
 *   * d4est_quadrature_t starts with `d4est_quadrature_type_t quad_type` (an int-sized enum, 0 = Gauss-Legendre,
 *     1 = Gauss-Lobatto; src/Quadrature/d4est_quadrature.h:8-14, :117-119).  The compactified types abort, as unsupported.
 *   * d4est_elliptic_data_t is the struct of src/EllipticSystem/d4est_elliptic_data.h:6-37, mirrored below.
 *
 * Host pointers in, host pointers out, like the reference.  Element-level shims run on cached one-element plans with persistent
 * pinned staging and device buffers: no hipMalloc per call (one set per (deg, deg_quad) pair on first use).  They are the
 * compatibility path, PCIe- and launch-latency bound (tens of microseconds per element); the fast path is a whole-mesh plan
 * (include/d4est_hip.h) bound to the p4est with d4est_hip_compat_bind_mesh, which the operator-level shims use.
 * Volume objects, DIM = 3 only (the reference's d8est build); other `dim` values abort.  Errors abort (D4EST_ABORT convention).
 */
#ifndef D4EST_HIP_COMPAT_H
#define D4EST_HIP_COMPAT_H

#include "d4est_hip.h"

#ifdef __cplusplus
extern "C" {
#define D4EST_RESTRICT
#else
#define D4EST_RESTRICT restrict
#endif

#ifndef D4EST_HIP_COMPAT_NO_TYPES
/* opaque stand-ins for the reference's types (pointers only) */
typedef struct p8est p4est_t;                                   /* pXest.h: p4est_t == p8est_t when DIM = 3 */
typedef struct d4est_operators_opaque d4est_operators_t;        /* dGMath/d4est_operators.h:9-51 */
typedef struct d4est_geometry_opaque d4est_geometry_t;          /* Geometry/d4est_geometry.h */
typedef struct d4est_quadrature_opaque d4est_quadrature_t;      /* Quadrature/d4est_quadrature.h:117-130 */
typedef struct d4est_mesh_data_opaque d4est_mesh_data_t;        /* Mesh/d4est_mesh.h:123-169 */
typedef struct d4est_ghost_opaque d4est_ghost_t;                /* Mesh/d4est_ghost.h */
typedef struct d4est_ghost_data_opaque d4est_ghost_data_t;      /* Mesh/d4est_ghost_data.h */
typedef struct d4est_laplacian_flux_data_opaque d4est_laplacian_flux_data_t; /* dGMath/d4est_laplacian_flux.h */
typedef struct d4est_laplacian_with_opt_flux_data_opaque d4est_laplacian_with_opt_flux_data_t; /* dGMath/d4est_laplacian_with_opt_flux.h */
typedef int d4est_quadrature_object_type_t;                     /* enum {QUAD_OBJECT_MORTAR, QUAD_OBJECT_VOLUME}, d4est_quadrature.h:16-17 */
typedef int d4est_quadrature_integrand_type_t;                  /* enum, d4est_quadrature.h:21-29 */
typedef int d4est_field_type_t;                                 /* enum, Mesh/d4est_field.h */
#define QUAD_OBJECT_MORTAR 0
#define QUAD_OBJECT_VOLUME 1
#define QUAD_INTEGRAND_UNKNOWN 5

/* src/EllipticSystem/d4est_elliptic_data.h:6-37 (all four vectors are aliases the callee never owns) */
typedef struct {
  int mpirank;
  int local_nodes;
  int num_of_fields;
  d4est_field_type_t* field_types;
  double* Au;
  double* u;
  double* u0;
  double* rhs;
  void* user;
} d4est_elliptic_data_t;

/* src/EllipticSystem/d4est_elliptic_eqns.h:12-36 */
typedef void (*d4est_apply_operator_fcn_t)(p4est_t*, d4est_ghost_t*, d4est_ghost_data_t*, d4est_elliptic_data_t*, d4est_operators_t*,
                                           d4est_geometry_t*, d4est_quadrature_t*, d4est_mesh_data_t*, void*);
typedef struct {
  d4est_apply_operator_fcn_t apply_lhs;
  d4est_apply_operator_fcn_t build_residual;
  void* user;
} d4est_elliptic_eqns_t;

/* ---- element level: src/Quadrature/d4est_quadrature.h:132-141 ------------------------------------------------------------ */
void d4est_quadrature_apply_stiffness_matrix(d4est_operators_t *d4est_ops,d4est_quadrature_t *d4est_quadrature,d4est_geometry_t *d4est_geometry,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *in,int deg_lobatto,double *jac_quad,double *rst_xyz[3][3],int deg_quad,double *out);
void d4est_quadrature_apply_mass_matrix(d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geometry,d4est_quadrature_t *d4est_quadrature,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *in,int deg_lobatto,double *jac_quad,int deg_quad,double *out);
void d4est_quadrature_apply_galerkin_integral(d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geometry,d4est_quadrature_t *d4est_quadrature,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *in_quad,int deg_lobatto,double *jac_quad,int deg_quad,double *out);
void d4est_quadrature_interpolate(d4est_operators_t *d4est_ops,d4est_quadrature_t *d4est_quadrature,d4est_geometry_t *d4est_geometry,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *u_lobatto_in,int deg_lobatto,double *u_quad_out,int deg_quad);
void d4est_quadrature_apply_inverse_mass_matrix(d4est_operators_t *d4est_ops,double *in,int deg_Lobatto,double *jac_Gauss,int deg_Gauss,int dim,double *out);

/* ---- element level: src/dGMath/d4est_operators.h:69-126 -------------------------------------------------------------------- */
void d4est_operators_apply_dij(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,int dir,double *D4EST_RESTRICT out);
void d4est_operators_apply_dij_transpose(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,int dir,double *D4EST_RESTRICT out);
void d4est_operators_apply_lift(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,int face,double *D4EST_RESTRICT out);
void d4est_operators_apply_slicer(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int face,int deg,double *D4EST_RESTRICT out);
void d4est_operators_apply_mij(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,double *D4EST_RESTRICT out);
void d4est_operators_apply_invmij(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,double *D4EST_RESTRICT out);
void d4est_operators_apply_p_prolong(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int degH,int dim,int degh,double *D4EST_RESTRICT out);
void d4est_operators_apply_hp_prolong(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int degH,int dim,int *degh,double *D4EST_RESTRICT out);
void d4est_operators_apply_p_restrict(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int degh,int dim,int degH,double *D4EST_RESTRICT out);
void d4est_operators_apply_hp_restrict(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int *degh,int dim,int degH,double *D4EST_RESTRICT out);
void d4est_operators_apply_p_prolong_transpose(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int degh,int dim,int degH,double *D4EST_RESTRICT out);
void d4est_operators_apply_hp_prolong_transpose(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int *degh,int dim,int degH,double *D4EST_RESTRICT out);

/* ---- operator / smoother level ---------------------------------------------------------------------------------------------
 * These take the p4est and the mesh-data structs, which the shims cannot read.  The host glue builds a
 * whole-mesh plan from them once per d4est_mesh_update and binds it to the p4est pointer; the shims look the plan up, run the
 * host-pointer entries of d4est_hip.h on the caller's vectors and ignore the other arguments.  An unbound p4est aborts.
 *   d4est_laplacian_apply_stiffness_matrix     src/dGMath/d4est_laplacian.h:22   u, Au: &vec[which_field * local_nodes]
 *   d4est_laplacian_apply_aij                  src/dGMath/d4est_laplacian.h:24   vectors from d4est_elliptic_data; flux_fcn_data
 *                                              is not read: SIPG parameters and boundary data are the plan's
 *   ..._smoother_cheby_iterate_aux             src/Solver/d4est_solver_multigrid_smoother_cheby.h:33; `fcns` is NOT called: the
 *                                              operator is the bound plan's apply_lhs (Laplacian + the zeroth-order term of
 *                                              d4est_hip_plan_set_lhs_coefficient), the whole loop runs on the device
 *   cg_eigs                                    src/Solver/d4est_solver_cg_eigs.h:9 */
void d4est_laplacian_apply_stiffness_matrix(p4est_t *p4est,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,double *D4EST_RESTRICT u,double *D4EST_RESTRICT Au,int local_nodes,int which_field);
void d4est_laplacian_apply_aij(p4est_t *p4est,d4est_ghost_t *d4est_ghost,d4est_ghost_data_t *d4est_ghost_data,d4est_elliptic_data_t *d4est_elliptic_data,d4est_laplacian_flux_data_t *flux_fcn_data,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,int which_field);
/* the "_with_opt" twins (src/dGMath/d4est_laplacian_with_opt.h:22-23): the reference's second implementation of the SAME operator,
 * which visits every face once and accumulates into both sides (d4est_laplacian_with_opt_flux_sipg.c:1231-1245); here they are the
 * same plan-bound applies as the two functions above */
void d4est_laplacian_with_opt_apply_stiffness_matrix(p4est_t *p4est,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,double *D4EST_RESTRICT u,double *D4EST_RESTRICT Au,int local_nodes,int which_field);
void d4est_laplacian_with_opt_apply_aij(p4est_t *p4est,d4est_ghost_t *d4est_ghost,d4est_ghost_data_t *d4est_ghost_data,d4est_elliptic_data_t *d4est_elliptic_data,d4est_laplacian_with_opt_flux_data_t *flux_fcn_data,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,int which_field);
void d4est_solver_multigrid_smoother_cheby_iterate_aux(p4est_t *p4est,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,d4est_ghost_t *d4est_ghost,d4est_ghost_data_t *d4est_ghost_data,d4est_elliptic_data_t *vecs,d4est_elliptic_eqns_t *fcns,double *r,int iter,double lmin,double lmax,int print_residual_norm,int mg_level,int compute_residual_at_end);
void cg_eigs(p4est_t *p4est,d4est_elliptic_data_t *vecs,d4est_elliptic_eqns_t *fcns,d4est_ghost_t *ghost,d4est_ghost_data_t *ghost_data,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,int imax,int print_spectral_bound_iterations,int use_new,double *spectral_bound);
#endif /* D4EST_HIP_COMPAT_NO_TYPES */

/* ---- binding (not in the reference) ------------------------------------------------------------------------------------------ */
/* associate a whole-mesh plan with a p4est pointer (call after every d4est_mesh_update; re-binding replaces; plan = NULL unbinds).
 * The plan stays owned by the caller. */
void d4est_hip_compat_bind_mesh(const void* p4est, d4est_hip_plan_t* plan);
d4est_hip_plan_t* d4est_hip_compat_bound_plan(const void* p4est);
/* free the cached one-element plans, transfer objects and staging buffers of the element-level shims */
void d4est_hip_compat_release(void);

#ifdef __cplusplus
}
#endif
#endif /* D4EST_HIP_COMPAT_H */
