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
 * DIM = 3 only (the reference's d8est build); other `dim` values abort.  Volume objects run on the device; QUAD_OBJECT_MORTAR objects of
 * the quadrature entry points (interpolate, mass, galerkin, the two callback forms) are served on the host.  Errors abort (D4EST_ABORT).
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
typedef struct d4est_solver_schwarz_opaque d4est_solver_schwarz_t;   /* Solver/d4est_solver_schwarz.h:14-38 (not read by the shims) */
/* The CALLER-SET head of the two flux-data structs, mirrored member for member (dGMath/d4est_laplacian_flux.h:113-128,
 * dGMath/d4est_laplacian_with_opt_flux.h:125-139); the "internally set" tails are never touched here.  The operator-level shims read
 * flux_type, the SIPG prefactor (first member of d4est_laplacian_flux_sipg_params_t, dGMath/d4est_laplacian_flux_sipg.h:18-19) and
 * bc_type, and abort when they differ from what the bound plan was set up with (d4est_hip_compat_bind_flux). */
typedef struct d4est_laplacian_flux_data {
  int flux_type;                                   /* d4est_laplacian_flux_type_t: FLUX_SIPG 0, FLUX_NIPG 1, FLUX_IIPG 2, FLUX_NOT_SET 3 (d4est_laplacian_aux.h:6) */
  void (*interface_fcn)(void);                     /* d4est_laplacian_flux_interface_fcn_t */
  void (*boundary_fcn)(void);                      /* d4est_laplacian_flux_boundary_fcn_t */
  void* flux_data;                                 /* FLUX_SIPG: d4est_laplacian_flux_sipg_params_t* */
  int (*get_deg_mortar_quad)(void*, void*);
  void* get_deg_mortar_quad_ctx;
  int bc_type;                                     /* d4est_laplacian_bc_t: BC_ROBIN 0, BC_DIRICHLET 1, BC_NOT_SET 2 (d4est_laplacian_aux.h:7) */
  void* bc_data;
} d4est_laplacian_flux_data_t;
typedef struct d4est_laplacian_with_opt_flux_data {
  int flux_type;
  void (*interface_fcn)(void);
  void (*boundary_fcn)(void);
  void* flux_data;
  int (*get_deg_mortar_quad)(void*, void*);
  void* get_deg_mortar_quad_ctx;
  int skip_p_side;
  int last_mortar_side_id_m;
  int bc_type;
  void* bc_data;
} d4est_laplacian_with_opt_flux_data_t;
#define FLUX_SIPG 0
#define BC_ROBIN 0
#define BC_DIRICHLET 1
/* src/Mesh/d4est_xyz_functions.h:14-25 (DIM = 3): f(x, y, z, user); src/Mesh/d4est_mesh.h:19 */
typedef double (*d4est_xyz_fcn_t)(double, double, double, void*);
typedef int d4est_mesh_init_field_option_t;
#define INIT_FIELD_NOT_SET 0
#define INIT_FIELD_ON_LOBATTO 1
#define INIT_FIELD_ON_QUAD 2
/* the additive Schwarz metadata, mirrored member for member (src/Solver/d4est_solver_schwarz_metadata.h:19-90; p4est_qcoord_t = int32_t):
 * what d4est_hip_compat_flatten_schwarz_metadata walks */
typedef struct {
  int mpirank, tree, tree_quadid, id, deg;
  int faces[3];
  int core_faces[3];
  int is_core;
  int nodal_size, nodal_stride, restricted_nodal_size, restricted_nodal_stride;
} d4est_solver_schwarz_element_metadata_t;
typedef struct {
  int mpirank, subdomain_id, core_id;
  d4est_solver_schwarz_element_metadata_t* element_metadata;
  int core_deg, core_tree, num_elements, restricted_nodal_size, restricted_nodal_stride, nodal_size, nodal_stride, element_stride;
} d4est_solver_schwarz_subdomain_metadata_t;
typedef struct {
  int num_nodes_overlap;
  int restricted_nodal_size, nodal_size, num_subdomains, num_elements;
  d4est_solver_schwarz_subdomain_metadata_t* subdomain_metadata;
  d4est_solver_schwarz_element_metadata_t* element_metadata;
  void* subdomain_ghostdata;   /* d4est_ghost_data_ext_t* */
  void* element_ghostdata;
  d4est_ghost_t* d4est_ghost;
  const char* input_section;
} d4est_solver_schwarz_metadata_t;
typedef int d4est_quadrature_object_type_t;                     /* enum {QUAD_OBJECT_MORTAR, QUAD_OBJECT_VOLUME}, d4est_quadrature.h:16-17 */
typedef int d4est_quadrature_integrand_type_t;                  /* enum, d4est_quadrature.h:21-29 */
typedef int d4est_field_type_t;                                 /* enum, Mesh/d4est_field.h */
typedef int d4est_quadrature_apply_or_compute_matrix_t;         /* enum {QUAD_APPLY_MATRIX, QUAD_COMPUTE_MATRIX}, d4est_quadrature.h:19 */
#define QUAD_OBJECT_MORTAR 0
#define QUAD_OBJECT_VOLUME 1
#define QUAD_INTEGRAND_UNKNOWN 5
#define QUAD_APPLY_MATRIX 0
#define QUAD_COMPUTE_MATRIX 1
/* src/Mesh/d4est_xyz_functions.h:27-37 (DIM = 3): f(x, y, z, u, user) */
typedef double (*d4est_xyzu_fcn_t)(double, double, double, double, void*);

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
/* the mass terms of the nonlinear problems with their user callbacks (d4est_quadrature.h:135, :138; d4est_quadrature.c:776-936, :593-774),
 * called directly by the Problem files (src/Problems/ConstantDensityStar/constant_density_star_fcns.h:407, :575): the callbacks are host
 * function pointers, evaluated on the host at the quadrature nodes (or, interpolate_f, at the Lobatto nodes), the integrals run through
 * the shims above.  QUAD_COMPUTE_MATRIX (the dense element matrix the multigrid matrix operator asks for,
 * src/Solver/d4est_solver_multigrid_matrix_operator.c:215-238) is d4est_quadrature_compute_mass_matrix (:1143-1186) with jac * f(u) f(v): all
 * columns in one device call.  interpolate / apply_mass_matrix / apply_galerkin_integral
 * and these two also serve QUAD_OBJECT_MORTAR objects (dim - 1, on the host), as the reference's estimators and mesh update need. */
void d4est_quadrature_apply_fofufofvlj(d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *u,double *v,int deg_lobatto,double *jac_quad,double *xyz_quad[3],int deg_quad,double *out,d4est_xyzu_fcn_t fofu_fcn,void *fofu_ctx,d4est_xyzu_fcn_t fofv_fcn,void *fofv_ctx,int interpolate_f,double *xyz_lobatto[3]);
void d4est_quadrature_compute_mass_matrix(d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geometry,d4est_quadrature_t *d4est_quadrature,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,int deg_lobatto,double *jac_quad,int deg_quad,double *out);
void d4est_quadrature_apply_fofufofvlilj(d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,void *object,d4est_quadrature_object_type_t object_type,d4est_quadrature_integrand_type_t integrand_type,double *vec,double *u,double *v,int deg_lobatto,double *xyz_quad[3],double *jac_quad,int deg_quad,double *out,d4est_xyzu_fcn_t fofu_fcn,void *fofu_ctx,d4est_xyzu_fcn_t fofv_fcn,void *fofv_ctx,d4est_quadrature_apply_or_compute_matrix_t apply_or_compute_matrix,int interpolate_f,double *xyz_lobatto[3]);

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
/* src/dGMath/d4est_operators.h:104, :107: the dense prolongation and the Galerkin product of the multigrid matrix operator's restriction
 * callback (src/Solver/d4est_solver_multigrid_matrix_operator.c:29).  compute_PT_mat_P forms sum_i P_i^T mat_i P_i; with eight children
 * the reference's own code reads a window of the transposed STACKED prolongation as the left factor (d4est_operators.c:651), which is not
 * P_i^T -- D4EST_HIP_REFERENCE_PT_WINDOW=1 reproduces that arithmetic to the letter (see DESIGN.md "MG matrix operator") */
void d4est_operators_compute_prolong_matrix(d4est_operators_t *d4est_ops,int degH,int dim,int *degh,int children,double *D4EST_RESTRICT prolong_mat);
void d4est_operators_compute_PT_mat_P(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT mat,int degH,int dim,int *degh,int children,double *D4EST_RESTRICT PT_mat_P);
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
 *                                              d4est_hip_plan_set_lhs_coefficient), the whole loop runs on the device.  So that a
 *                                              caller with a DIFFERENT apply_lhs is not served the wrong operator silently, register
 *                                              the callback the plan stands for with d4est_hip_compat_bind_operator: the two shims
 *                                              then abort when fcns->apply_lhs is another function
 *   cg_eigs                                    src/Solver/d4est_solver_cg_eigs.h:9 */
/* also with the reference's own names and argument lists (registrations below supply what the opaque mesh structs would):
 *   d4est_laplacian_build_rhs_with_strong_bc   src/dGMath/d4est_laplacian.h:25  the source callback is evaluated on the host at the
 *                                              coordinates of d4est_hip_compat_bind_coordinates; boundary data as set on the plan
 *   d4est_solver_schwarz_iterate               src/Solver/d4est_solver_schwarz.h:42  on the handle of d4est_hip_compat_bind_schwarz
 *   d4est_operators_apply_flip, _reorient_face_data   src/dGMath/d4est_operators.h (index work, host)
 *   d4est_mortars_project_side_onto_mortar_space, _mass_mortar_onto_side   src/Mesh/d4est_mortars.h (face transfers, dim - 1, host)
 * apply_p_prolong & co. accept dim = 2 (faces: host) besides dim = 3 (device). */
void d4est_laplacian_build_rhs_with_strong_bc(p4est_t *p4est,d4est_ghost_t *ghost,d4est_ghost_data_t *ghost_data,d4est_operators_t *d4est_ops,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,d4est_elliptic_data_t *prob_vecs,d4est_laplacian_flux_data_t *flux_fcn_data_for_build_rhs,double *D4EST_RESTRICT rhs,d4est_xyz_fcn_t problem_rhs_fcn,d4est_mesh_init_field_option_t init_option,void *ctx,int which_field);
void d4est_solver_schwarz_iterate(p4est_t *p4est,d4est_geometry_t *d4est_geom,d4est_quadrature_t *d4est_quad,d4est_mesh_data_t *d4est_factors,d4est_ghost_t *ghost,d4est_solver_schwarz_t *schwarz,d4est_elliptic_data_t *vecs,double *r);
void d4est_operators_apply_flip(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int dim,int deg,int dir,double *out);
void d4est_operators_reorient_face_data(d4est_operators_t *d4est_ops,double *D4EST_RESTRICT in,int face_dim,int deg,int o,int f_m,int f_p,double *D4EST_RESTRICT out);
void d4est_mortars_project_side_onto_mortar_space(d4est_operators_t *d4est_ops,double *in_side,int faces_side,int *deg_side,double *out_mortar,int faces_mortar,int *deg_mortar);
void d4est_mortars_project_mass_mortar_onto_side(d4est_operators_t *dgmath,double *in_mortar,int faces_mortar,int *deg_mortar,double *out_side,int faces_side,int *deg_side);
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
#ifndef D4EST_HIP_COMPAT_NO_TYPES
/* the apply_lhs callback whose operator the bound plan applies (e.g. constant_density_star_apply_jac): cheby_iterate_aux / cg_eigs abort
 * on any other fcns->apply_lhs; NULL removes the registration (then fcns is not looked at) */
void d4est_hip_compat_bind_operator(const void* p4est, d4est_apply_operator_fcn_t apply_lhs);
/* what the bound plan was set up with: [flux] sipg_penalty_prefactor and the boundary-condition type (BC_ROBIN 0 / BC_DIRICHLET 1).  Once
 * registered, d4est_laplacian_apply_aij, its _with_opt twin and d4est_laplacian_build_rhs_with_strong_bc ABORT when the flux data the
 * caller passes says otherwise (other flux type, other prefactor, other bc_type) instead of silently applying the plan's operator */
void d4est_hip_compat_bind_flux(const void* p4est, double sipg_penalty_prefactor, int bc_type);
/* the node coordinates of the mesh (d4est_factors->xyz[d] at the Lobatto nodes, ->xyz_quad[d] at the quadrature nodes; host arrays that
 * stay the caller's; either may be NULL): where d4est_laplacian_build_rhs_with_strong_bc evaluates the source callback */
void d4est_hip_compat_bind_coordinates(const void* p4est, double* xyz_lobatto[3], double* xyz_quad[3]);
/* the Schwarz smoother of this mesh and its three [d4est_solver_schwarz] CG options, for d4est_solver_schwarz_iterate; NULL unbinds */
void d4est_hip_compat_bind_schwarz(const void* p4est, d4est_hip_schwarz_t* sz, int subdomain_iter, double subdomain_atol, double subdomain_rtol);
/* the reference's Schwarz metadata as the flat arrays d4est_hip_schwarz_create takes; outputs caller-allocated:
 * sub_first[num_subdomains + 1], sub_elem[num_elements], sub_faces / sub_core_faces[3 num_elements] */
void d4est_hip_compat_flatten_schwarz_metadata(const d4est_solver_schwarz_metadata_t* md, int* sub_first, int* sub_elem, int* sub_faces, int* sub_core_faces);
/* d4est_laplacian_build_rhs_with_strong_bc (src/dGMath/d4est_laplacian.c:16-140) on the bound plan, with the source values handed over: rhs[which_field] = M f - A(0).
 * The reference's function evaluates the source callback through d4est_mesh_init_field (it needs the p4est and the mesh data, which
 * the shims cannot read), so the caller does that step -- one call it already has -- and passes the values: f at the Lobatto nodes
 * (init_option 1 = INIT_FIELD_ON_LOBATTO) or at the quadrature nodes (2 = INIT_FIELD_ON_QUAD; the values of d4est_mesh_init_field_option_t).  Boundary data: set the inhomogeneous
 * values on the plan first (d4est_hip_plan_set_dirichlet_values), the homogeneous ones afterwards.  INTEGRATION.md section 3. */
void d4est_hip_compat_build_rhs_with_strong_bc(const void* p4est, d4est_elliptic_data_t* prob_vecs, double* rhs, const double* f, int init_option, int which_field);
#endif
/* free the cached one-element plans, transfer objects and staging buffers of the element-level shims */
void d4est_hip_compat_release(void);

#ifdef __cplusplus
}
#endif
#endif /* D4EST_HIP_COMPAT_H */
